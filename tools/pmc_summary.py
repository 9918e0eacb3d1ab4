"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (short names). Usage: pmc_summary.py file.csv [substr]"""
import csv, sys, re, collections
rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for r in rows:
    name = re.sub(r"\(.*$", "", r["Kernel_Name"].replace("(anonymous namespace)::", ""))[:60]
    if flt not in name:
        continue
    key = (name, r["Dispatch_Id"])
    agg.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
for (name, d), c in agg.items():
    print(name, "dispatch", d, " ".join(f"{k}={v:.4g}" for k, v in sorted(c.items())))
