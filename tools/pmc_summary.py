"""Aggregate a rocprofv3 --pmc counter_collection.csv: mean counter value per dispatch, per kernel.
usage: python tools/pmc_summary.py <dir-or-csv> [kernel-substring]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    files = [src] if src.endswith(".csv") else glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(lambda: defaultdict(int))
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"][:60]
                if flt and flt not in k:
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                cnt[k][row["Counter_Name"]] += 1
    for k in acc:
        print(k)
        for c in sorted(acc[k]):
            print(f"   {c:34s} {acc[k][c] / cnt[k][c]:16.1f}  (n={cnt[k][c]})")


if __name__ == "__main__":
    main()
