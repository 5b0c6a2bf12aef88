#!/bin/bash
# usage (GPU box): bash tools/prof_launches.sh <workload> <tag> [bench flags] -> gpurun_out/<tag>_launches.csv (one step)
set -e
wl=$1; tag=$2; shift 2
export TMPDIR=/tmp
out=$PWD/gpurun_out
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --output-format rocpd -d /tmp/prof_$tag -o r -- python3 bench.py --workload $wl --steps 4 --warmup 2 $( [ -n "$GRAPH" ] || echo --no-graph ) --only --no-cpu-baseline --no-roofline "$@" > $out/${tag}_prof.log 2>&1
db=$(find /tmp/prof_$tag -name '*.db' | head -1)
STEP_BACK=${STEP_BACK:-0} python3 tools/rocpd_launches.py $db $out/${tag}_launches.csv x | head -3
wc -l $out/${tag}_launches.csv
