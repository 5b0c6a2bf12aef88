#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc_workload.sh <workload> <tag>   -> gpurun_out/<tag>_pmc_traffic.json
# three separate counter passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) over a short eager run of one secondary workload, summarised
# per kernel over the steady-state steps (tools/pmc_kernels.py: the dispatches between the first and the last adam_kernel)
set -e
wl=$1; tag=$2
export TMPDIR=/tmp
out=$PWD/gpurun_out
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  t=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmcw_$t
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmcw_$t --output-format csv -- python3 bench.py --workload $wl --only --steps 7 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $out/pmcw_$t.log 2>&1
done
python3 tools/pmc_kernels.py /tmp/pmcw_FETCH_SIZE /tmp/pmcw_WRITE_SIZE /tmp/pmcw_SQ_VALU_MFMA_BUSY_CYCLES > $out/${tag}_pmc_traffic.json
tail -c 200 $out/${tag}_pmc_traffic.json
