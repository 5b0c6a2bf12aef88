#!/bin/bash
# usage (GPU box, from the repo root): bash tools/prof_workload.sh <workload> <tag> [extra bench.py flags]
# rocprofv3 kernel trace of an un-graphed bench run -> gpurun_out/<tag>_kernel_stats.csv
set -e
wl=$1; tag=$2; shift 2
export TMPDIR=/tmp
out=$PWD/gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -o r -- python3 bench.py --workload $wl --steps 12 --warmup 3 $( [ -n "$GRAPH" ] || echo --no-graph ) --only --no-cpu-baseline --no-roofline "$@" > $out/${tag}_prof.log 2>&1
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
python3 tools/rocpd_stats.py $db $out/${tag}_kernel_stats.csv > /dev/null
head -14 $out/${tag}_kernel_stats.csv | cut -c1-120
tail -1 $out/${tag}_kernel_stats.csv
