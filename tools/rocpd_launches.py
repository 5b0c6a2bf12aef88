"""Per-launch listing of one training step from a rocprofv3 (rocpd sqlite) kernel trace: kernel, grid, workgroup,
duration (STEP_BACK=n: the n-th step before the last), and how many rounds of (256 CUs x workgroups per CU) the grid needs - a grid of 258 single-occupancy
workgroups takes two rounds.  usage: python tools/rocpd_launches.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:70]


def main():
    c = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    want = [k for k in ("grid_x", "grid_y", "grid_z", "workgroup_x", "workgroup_y", "workgroup_z", "grid_size_x",
                        "grid_size_y", "grid_size_z", "workgroup_size_x", "workgroup_size_y", "workgroup_size_z",
                        "lds_size", "lds_block_size") if k in cols]
    if len(sys.argv) > 3:
        print(cols)
    rows = list(c.execute(f"select {namecol}, start, end, {', '.join(want)} from kernels order by start"))
    # one step = the span between the last two adam_kernel launches
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    import os
    back = int(os.environ.get("STEP_BACK", "0"))      # 0 = the last step, 1 = the one before, ...
    lo, hi = (adam[-2 - back] + 1, adam[-1 - back] + 1) if len(adam) >= 2 + back else (0, len(rows))
    out = ["kernel,grid,workgroup,lds,us"]
    for r in rows[lo:hi]:
        d = dict(zip(want, r[3:]))
        gx = d.get("grid_x", d.get("grid_size_x", 0))
        gy = d.get("grid_y", d.get("grid_size_y", 1)) or 1
        gz = d.get("grid_z", d.get("grid_size_z", 1)) or 1
        wx = d.get("workgroup_x", d.get("workgroup_size_x", 1)) or 1
        wy = d.get("workgroup_y", d.get("workgroup_size_y", 1)) or 1
        wz = d.get("workgroup_z", d.get("workgroup_size_z", 1)) or 1
        wgs = (gx * gy * gz) // (wx * wy * wz)
        lds = d.get("lds_size", d.get("lds_block_size", 0))
        out.append(f"\"{short(r[0])}\",{wgs},{wx * wy * wz},{lds},{(r[2] - r[1]) / 1e3:.2f}")
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    else:
        print(text)


main()
