"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (CPU box, no GPU).
Usage: python tools/kernel_resources.py [source.hip ...]"""
import os, re, subprocess, sys
HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(HERE, "omgsr_amd", "csrc")
srcs = sys.argv[1:] or ["igemm.hip", "igemm_dma.hip", "igemm_halo.hip", "attention.hip", "norm.hip", "elementwise.hip"]
for src in srcs:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", os.path.join(CSRC, src), "-o", "/dev/null",
                        "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+: +([A-Za-z \[\]/]+): +(\S+)", line) or re.search(r"remark: +([A-Za-z \[\]/]+): +(\S+)", line)
        if not m:
            m = re.search(r": +(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]): +(\S+)", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
        cur[k] = v
        if k.startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*", "", name)
            print(f"{src:18s} {name[:70]:70s} V={cur.get('VGPRs'):>4} A={cur.get('AGPRs'):>4} spillV={cur.get('VGPRs Spill')} scratch={cur.get('ScratchSize [bytes/lane]')} occ={cur.get('Occupancy [waves/SIMD]')} lds={v}")
