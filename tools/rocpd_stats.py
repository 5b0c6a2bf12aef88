"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average duration.
usage: python tools/rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    q = f"select {namecol}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by {namecol} order by 3 desc"
    rows = list(c.execute(q))
    tot = sum(r[2] for r in rows)
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct"]
    for n, cnt, s, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{cnt},{s/1e6:.3f},{a/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*s/tot:.2f}")
    lines.append(f"\"TOTAL\",{sum(r[1] for r in rows)},{tot/1e6:.3f},,,,100")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
