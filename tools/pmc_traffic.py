"""HBM traffic per convolution family from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
`python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline`.

usage: python tools/pmc_traffic.py <fetch-dir> <write-dir> > profiles/rNN_pmc_traffic.json

Corrections as /opt/skills/guides/MI355X_MICROARCH.md prescribes: the counters are in KB; FETCH_SIZE
is doubled on gfx950 (128-byte requests tallied as 64).  Kernels are attributed to bench.py's three
families by name; the 1x1 kernels shared by the forward and input-gradient directions and the
split-K reduction are split evenly between the two; "per launch" divides by the family's operator
calls (its primary kernels, not the reductions).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    tot, n = defaultdict(float), defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                tot[row["Kernel_Name"]] += float(row["Counter_Value"])
                n[row["Kernel_Name"]] += 1
    return tot, n


def family(name):
    """-> list of (family, weight, is_primary)"""
    if "wgrad" in name:
        return [("wgrad", 1.0, "reduce" not in name)]
    if "conv3x3_kernel<0>" in name or "igemm_kernel<0" in name:
        return [("igemm_xy", 1.0, True)]
    if "conv3x3_kernel<1>" in name or "conv3x3_kernel<2>" in name or "igemm_kernel<1" in name or "smalln" in name:
        return [("igemm_yx", 1.0, True)]
    if "gemm_rows" in name or "gemm_stream" in name:
        return [("igemm_xy", 0.5, True), ("igemm_yx", 0.5, True)]
    if "splitk_reduce" in name:
        return [("igemm_xy", 0.5, False), ("igemm_yx", 0.5, False)]
    return []


def main():
    fetch, nf = load(sys.argv[1], "FETCH_SIZE")
    write, _ = load(sys.argv[2], "WRITE_SIZE")
    out = {"method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 3 "
                     "--warmup 1 --no-graph`; FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B "
                     "requests at 64 B); KB -> bytes; families by kernel name (tools/pmc_traffic.py)"}
    fam = defaultdict(lambda: {"fetch": 0.0, "write": 0.0, "launches": 0.0})
    for k in fetch:
        for f, w, primary in family(k):
            fam[f]["fetch"] += w * fetch[k] * 2.0 * 1024.0
            fam[f]["write"] += w * write.get(k, 0.0) * 1024.0
            if primary:
                fam[f]["launches"] += w * nf[k]
    for f, v in fam.items():
        n = max(v["launches"], 1.0)
        out[f] = {"launches_profiled": int(round(n)),
                  "fetch_bytes_per_launch": int(v["fetch"] / n),
                  "write_bytes_per_launch": int(v["write"] / n),
                  "traffic_bytes_per_launch": int((v["fetch"] + v["write"]) / n)}
    adam = [k for k in fetch if "adam_kernel" in k]
    steps = nf[adam[0]] if adam else 1
    out["step_total"] = {"steps_profiled": steps,
                         "fetch_bytes": int(sum(fetch.values()) * 2.0 * 1024.0 / steps),
                         "write_bytes": int(sum(write.values()) * 1024.0 / steps)}
    if adam:
        out["check_adam_kernel"] = {"fetch_bytes": int(fetch[adam[0]] * 2048.0 / steps),
                                    "write_bytes": int(write.get(adam[0], 0.0) * 1024.0 / steps)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
