#!/bin/bash
# A/B of environment settings on the DDPM training step (graph replay): ms per step at the given batches.
#   tools/step_ab.sh "128 64 32 16" "LGM_WINO4_LIGHT=0" "LGM_WINO4_LIGHT=1" "LGM_WINO4_LIGHT=1 LGM_WINO4_FWD_ALL=1"
# One bench.py process per (setting, batch); nothing here retries.
batches="$1"; shift
for setting in "$@"; do
  for b in $batches; do
    out=$(env $setting timeout -k 10 200 python bench.py --only --batch $b --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1)
    ms=$(echo "$out" | python -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null)
    echo "[$setting] B=$b: $ms ms/step"
  done
done
