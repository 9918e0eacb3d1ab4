"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result: per-kernel launches / total / average / % —
the same table `--stats` prints. Usage: python tools/rocpd_summary.py results.db [out.md]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\(.*$", "", name)          # drop the argument list
    return name[:110]


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    total = sum(v[1] for v in agg.values())
    lines = ["| kernel | launches | total ms | avg us | % |", "|---|---|---|---|---|"]
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"| `{k}` | {c} | {ns / 1e6:.3f} | {ns / c / 1e3:.2f} | {100.0 * ns / total:.1f} |")
    lines.append(f"| **total** | {sum(v[0] for v in agg.values())} | {total / 1e6:.3f} | | 100 |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "a") as f:
            f.write(out + "\n")


if __name__ == "__main__":
    main()
