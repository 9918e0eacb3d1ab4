"""Effective clock and MFMA-pipe occupancy per dispatch from one rocprofv3 --pmc pass that collected
GRBM_GUI_ACTIVE and SQ_VALU_MFMA_BUSY_CYCLES (+ optional SQ_WAVE_CYCLES / SQ_WAIT_INST_ANY) with --kernel-trace, csv output.
Usage: python tools/pmc_clock.py <dir with *_kernel_trace.csv and *_counter_collection.csv> [kernel substring]"""
import csv, glob, sys
d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
kt = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
    kt[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cc = {}
for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
    cc.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
for disp, c in cc.items():
    ns, name = kt[disp]
    if flt not in name or "GRBM_GUI_ACTIVE" not in c:
        continue
    cyc = c["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
    line = f"{disp:>5} {name[:44]:44s} {ns / 1e3:8.1f} us  {cyc / ns:5.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        line += f"  MFMA busy {c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc * 100:5.1f} %"      # 1024 SIMDs
    if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_INST_ANY" in c:
        line += f"  waves waiting {c['SQ_WAIT_INST_ANY'] / c['SQ_WAVE_CYCLES'] * 100:5.1f} %"
    print(line)
