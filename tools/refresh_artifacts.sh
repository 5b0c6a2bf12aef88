#!/bin/bash
# usage (GPU box, repo root): bash tools/refresh_artifacts.sh   -> everything lands under gpurun_out/
# bench JSON lines of the four workloads + the per-rank-batch proxies, rocprofv3 kernel-trace summaries, three PMC passes
set -e
export TMPDIR=/tmp
out=$PWD/gpurun_out
python3 bench.py 2> $out/bench_ddpm32.err | tail -1 > $out/r02_bench_ddpm32.json
for w in ddpm64 wgan_gp64 vqvae; do python3 bench.py --workload $w 2> $out/bench_$w.err | tail -1 > $out/r02_bench_$w.json; done
python3 bench.py --workload vqvae --vq-ema 2> $out/bench_vqvae_ema.err | tail -1 > $out/r02_bench_vqvae_ema.json
for b in 64 32 16; do python3 bench.py --batch $b --no-cpu-baseline 2>/dev/null | tail -1 > $out/r02_bench_ddpm32_b$b.json; done
echo "bench lines done"
bash tools/prof_workload.sh ddpm32 r02_bench_b128 > /dev/null
for w in ddpm64 wgan_gp64 vqvae; do bash tools/prof_workload.sh $w r02_bench_$w > /dev/null; done
bash tools/prof_workload.sh ddpm32 r02_bench_b16 --batch 16 > /dev/null
bash tools/prof_workload.sh ddpm32 r02_bench_b64 --batch 64 > /dev/null
echo "kernel traces done"
for c in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pmc_$tag
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$tag --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $out/pmc_$tag.log 2>&1
  echo "pmc $tag done"
done
python3 tools/pmc_kernels.py /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE /tmp/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/r02_pmc_traffic.json
head -c 600 $out/r02_pmc_traffic.json | tail -c 300
