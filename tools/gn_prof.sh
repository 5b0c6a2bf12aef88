#!/bin/bash
# usage (GPU box): bash tools/gn_prof.sh B  -> device-side duration of the GroupNorm kernels per variant (rocprofv3)
export TMPDIR=/tmp
B=${1:-16}
for v in plain film+act planes2 planes4 bwd rms; do
  rm -rf /tmp/gnp; rocprofv3 --kernel-trace --output-format rocpd -d /tmp/gnp -o r -- python3 tools/gn_bench.py $B $v > /dev/null 2>&1
  db=$(find /tmp/gnp -name '*.db' | head -1)
  echo "== variant $v (B=$B)"
  python3 - "$db" <<'PY'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
n = "name" if "name" in cols else "kernel_name"
for name, cnt, avg, mn in c.execute(f"select {n}, count(*), avg(end-start), min(end-start) from kernels group by {n} having count(*) > 50 order by 3 desc"):
    if re.search("gn_|rmsnorm", name):
        print(f"   {name.split('(')[0][:44]:44s} calls {cnt:4d}  avg {avg/1e3:6.2f} us  min {mn/1e3:6.2f} us")
PY
done
