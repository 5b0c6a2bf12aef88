for cfg in "" "LGM_PLAN_SLAB=0.5" "LGM_PLAN_SLAB=2" "LGM_PLAN_TC=2.5" "LGM_PLAN_TW=4.5" "LGM_PLAN_WPH=1.2" "LGM_PLAN_TC=2.5 LGM_PLAN_TW=4.5" ; do
  for b in 128 16; do
    r=$(env $cfg python bench.py --only --no-cpu-baseline --batch $b 2>&1 >/dev/null | grep "timed region" | sed "s/.*steps in//")
    echo "cfg=[$cfg] B=$b $r"
  done
done
