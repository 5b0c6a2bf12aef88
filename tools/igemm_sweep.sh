for cfg in "" "LGM_IGEMM_TSMALL=128" "LGM_IGEMM_TSMALL=512" "LGM_IGEMM_TBIG=512" "LGM_IGEMM_CMIN=2" "LGM_IGEMM_CMIN=8" "LGM_IGEMM_TSMALL=64 LGM_IGEMM_TBIG=256"; do
  for b in 128 32 16; do
    r=$(env $cfg python bench.py --only --no-cpu-baseline --batch $b 2>&1 >/dev/null | grep "timed region" | sed "s/.*steps in//")
    echo "cfg=[$cfg] B=$b $r"
  done
done
