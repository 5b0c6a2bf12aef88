#!/bin/bash
# usage (GPU box): bash tools/plan_sweep.sh  -> step time of the headline workload under variations of the split planners'
# cost-model constants (pair plan: LGM_PLAN_*; forward plan: LGM_FPLAN_*), at the per-rank batches
for cfg in "" "LGM_FPLAN_OVH=2.5" "LGM_FPLAN_OVH=1.0" "LGM_FPLAN_SPLIT=0.5" "LGM_FPLAN_SPLIT=2" "LGM_FPLAN_SPLIT=0.25 LGM_FPLAN_OVH=2.5" ; do
  for b in 128 32 16; do
    r=$(env $cfg python bench.py --only --no-cpu-baseline --batch $b 2>&1 >/dev/null | grep "timed region" | sed "s/.*steps in//")
    echo "cfg=[$cfg] B=$b $r"
  done
done
