#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_ab.sh "ENV=a" "ENV=b" ...   -> steady-state fetch / write bytes per step for each
# environment setting (two counter passes each), the largest kernels' rows into gpurun_out/pmc_ab_<n>.json
export TMPDIR=/tmp
out=$PWD/gpurun_out
i=0
for e in "$@"; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmcab_$c
    env $e true
    ( export $e; rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcab_$c --output-format csv -- python3 bench.py --only --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $out/pmcab.log 2>&1 )
  done
  python3 tools/pmc_kernels.py /tmp/pmcab_FETCH_SIZE /tmp/pmcab_WRITE_SIZE > $out/pmc_ab_$i.json
  echo "[$e] $(tail -c 200 $out/pmc_ab_$i.json | tr -d '\n')"
  i=$((i+1))
done
