#!/bin/bash
# A/B runs of the F(4x4) selection thresholds (lgm_conv3x3_wino4_preferred) on the GPU box:
#   bash tools/w4_threshold_sweep.sh small     headline workload at the per-rank batches 64 / 32 / 16
#   bash tools/w4_threshold_sweep.sh ddpm64    config 5 (B = 64, 64 x 64)
set -e
run() { # name, bench args..., then VAR=value...
  name=$1; shift
  args=(); while [[ $# -gt 0 && $1 != *=* ]]; do args+=("$1"); shift; done
  v=$(env "$@" timeout -k 10 200 python bench.py --only "${args[@]}" 2>/dev/null | python -c "import sys,json; [print(json.loads(l)['ms_per_step']) for l in sys.stdin if l.startswith('{')]")
  echo "$name ${args[*]} $* -> $v ms"
}
if [[ ${1:-small} == ddpm64 ]]; then
  a="--workload ddpm64 --steps 20 --warmup 5"
  run base $a LGM_X=0
  run no8 $a LGM_WINO4_NO8=1
  run noyx8 $a LGM_WINO4_NOYX8=1
  run c2_512 $a LGM_WINO4_C2_MINC=512
  run c2_1024 $a LGM_WINO4_C2_MINC=1024
  run base $a LGM_X=0
else
  for b in 64 32 16; do
    a="--batch $b --steps 30 --warmup 8"
    run base $a LGM_X=0
    run u1_64 $a LGM_WINO4_MIN_UNITS1=64
    run u0_64 $a LGM_WINO4_MIN_UNITS=64
    run both64 $a LGM_WINO4_MIN_UNITS=64 LGM_WINO4_MIN_UNITS1=64
  done
fi
