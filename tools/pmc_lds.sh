#!/bin/bash
# usage (GPU box): bash tools/pmc_lds.sh   -> LDS bank-conflict share per kernel of one DDPM step (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)
export TMPDIR=/tmp
rm -rf /tmp/pmc_lds
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/pmc_lds --output-format csv -- python3 bench.py --only --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/pmc_lds.log 2>&1
python3 - <<PY
import csv, glob, collections
tot=collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("/tmp/pmc_lds/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].replace("(anonymous namespace)::","").split("(")[0][-44:]
        tot[k][r["Counter_Name"]]+=float(r["Counter_Value"])
print("kernel, conflict/active, lds_active/wave_cycles, lds_insts")
for k,v in sorted(tot.items(), key=lambda kv:-kv[1].get("SQ_LDS_IDX_ACTIVE",0))[:14]:
    a=v.get("SQ_LDS_IDX_ACTIVE",1)
    print(f"{k}, {v.get('SQ_LDS_BANK_CONFLICT',0)/max(a,1):.3f}, {a/max(v.get('SQ_WAVE_CYCLES',1),1):.3f}, {int(v.get('SQ_INSTS_LDS',0))}")
PY
