"""Per-kernel counter summary from separate rocprofv3 --pmc passes over the same command
(`python3 bench.py --only --steps 3 --warmup 1 --no-graph --no-cpu-baseline --no-roofline`):

    python tools/pmc_kernels.py <fetch-dir> <write-dir> <mfma-dir> > profiles/rNN_pmc_traffic.json

FETCH_SIZE / WRITE_SIZE as /opt/skills/guides/MI355X_MICROARCH.md prescribes: counters are in KB, FETCH_SIZE is doubled
on gfx950 (128-byte requests tallied as 64 bytes).  SQ_VALU_MFMA_BUSY_CYCLES (summed over the chip's SIMDs) and
SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE from the third pass.  Values are means per launch of the kernel name."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def load(d):
    tot, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = row["Kernel_Name"].replace("(anonymous namespace)::", "")
                k = re.sub(r"^void ", "", k)
                k = re.sub(r"\(.*$", "", k).strip()                          # drop the argument list
                tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                n[k][row["Counter_Name"]] += 1
    return tot, n


def main():
    ft, fn = load(sys.argv[1])
    wt, wn = load(sys.argv[2])
    mt, mn = load(sys.argv[3]) if len(sys.argv) > 3 else ({}, {})
    out = {"method": "rocprofv3 --pmc in separate passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES "
                     "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) over `python3 bench.py --only --steps 3 --warmup 1 --no-graph "
                     "--no-cpu-baseline --no-roofline`; KB -> bytes; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md); means per launch",
           "kernels": {}}
    steps = max((fn[k]["FETCH_SIZE"] for k in fn if "adam_kernel" in k), default=1)
    tf = tw = 0.0
    for k in sorted(ft, key=lambda k: -ft[k].get("FETCH_SIZE", 0.0)):
        nl = fn[k]["FETCH_SIZE"]
        fetch = ft[k]["FETCH_SIZE"] * 2048.0 / nl
        write = wt.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0 / max(wn.get(k, {}).get("WRITE_SIZE", 1), 1)
        tf += ft[k]["FETCH_SIZE"] * 2048.0
        tw += wt.get(k, {}).get("WRITE_SIZE", 0.0) * 1024.0
        ent = {"launches_per_step": round(nl / steps, 2), "fetch_bytes_per_launch": int(fetch),
               "write_bytes_per_launch": int(write), "traffic_bytes_per_launch": int(fetch + write)}
        m = mt.get(k)
        if m:
            for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                if c in m:
                    ent[c + "_per_launch"] = int(m[c] / mn[k][c])
        out["kernels"][k] = ent
    out["step_total"] = {"steps_profiled": steps, "fetch_bytes": int(tf / steps), "write_bytes": int(tw / steps)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
