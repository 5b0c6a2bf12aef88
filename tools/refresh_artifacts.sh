#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_artifacts.sh [rNN]   -> everything lands under gpurun_out/
# the driver's default bench line (headline + secondary configs + per-rank proxy), rocprofv3 kernel-trace summaries of
# the SAME commands (graph replay) at the headline and the per-rank batches, one step's per-launch listing (eager),
# the secondary workloads' kernel summaries, three PMC passes (FETCH_SIZE | WRITE_SIZE | MFMA busy) and the LDS pass
set -e
R=${1:-r06}
export TMPDIR=/tmp
out=$PWD/gpurun_out
# STAGE=1: bench line + kernel traces; STAGE=2: counter passes + engine numbers; unset: everything (may exceed one gpurun call)
if [ "${STAGE:-1}" = "1" ]; then
python3 bench.py 2> $out/${R}_bench.err | tail -1 > $out/${R}_bench.json
echo "bench line done"
GRAPH=1 bash tools/prof_workload.sh ddpm32 ${R}_bench_b128 > /dev/null
# the per-rank batches run the kernel selection of an N > 1 job (light F(4x4) workgroups, launch plans for 240 CUs), as
# bench.py's per_rank_proxy does
for b in 64 32 16; do LGM_WINO4_LIGHT=1 LGM_CU_MARGIN=16 GRAPH=1 bash tools/prof_workload.sh ddpm32 ${R}_bench_b$b --batch $b > /dev/null; done
for w in ddpm64 wgan_gp64 vqvae vqvae_ema; do GRAPH=1 bash tools/prof_workload.sh $w ${R}_bench_$w > /dev/null; done
bash tools/prof_launches.sh ddpm32 ${R}_b128 > /dev/null
LGM_WINO4_LIGHT=1 LGM_CU_MARGIN=16 bash tools/prof_launches.sh ddpm32 ${R}_b16 --batch 16 > /dev/null
bash tools/prof_launches.sh vqvae ${R}_vqvae > /dev/null
bash tools/prof_launches.sh wgan_gp64 ${R}_wgan_generator_step > /dev/null
echo "kernel traces done"
fi
if [ -z "$STAGE" ] || [ "$STAGE" = "2" ]; then
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$tag --output-format csv -- python3 bench.py --only --steps 4 --warmup 1 --no-graph --no-cpu-baseline --no-roofline > $out/pmc_$tag.log 2>&1
  echo "pmc $tag done"
done
python3 tools/pmc_kernels.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/${R}_pmc_traffic.json
bash tools/pmc_lds.sh > $out/${R}_pmc_lds.txt 2>&1 || true
tail -c 300 $out/${R}_pmc_traffic.json
# round 6: the non-fused Winograd engine's prototype numbers and the GEMM / 1x1 comparisons
python3 tools/weng_proto.py > $out/${R}_weng_proto.txt 2> /dev/null || true
python3 tools/weng_gemm_bench.py > $out/${R}_weng_gemm_bench.txt 2> /dev/null || true
python3 tools/gemm1x1_bench.py 128 > $out/${R}_gemm1x1_b128.txt 2> /dev/null || true
python3 tools/gemm1x1_bench.py 16 > $out/${R}_gemm1x1_b16.txt 2> /dev/null || true
GRAPH=1 bash tools/prof_launches.sh ddpm32 ${R}_b128_graph > /dev/null
echo "engine numbers done"
fi
if [ -n "$HOG" ]; then
# what a resident collective costs (1-GPU emulation): 1 / 16 foreign workgroups beside the replayed step, with the kernel
# selection of one GPU and of a rank (light workgroups; without and with the CU margin)
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/libcu_hog.so tools/cu_hog.hip
( for e in "LGM_WINO4_LIGHT=0 LGM_CU_MARGIN=0" "LGM_WINO4_LIGHT=1 LGM_CU_MARGIN=0" "LGM_WINO4_LIGHT=1 LGM_CU_MARGIN=16"; do
    echo "== $e"; env $e python3 tools/cu_hog_step.py 128 64 32 16 2>&1 | grep -v amdgpu.ids; done ) > $out/${R}_hog_step.txt
tail -4 $out/${R}_hog_step.txt
fi
