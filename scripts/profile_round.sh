#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel stats + PMC passes (HBM traffic, SQ), everything under gpurun_out/<tag>/
#   scripts/profile_round.sh <tag> [quick]
set -u
tag=${1:-r02}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import torch" || exit 1
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $out/bench_b12.json 2> $out/bench_b12.err || exit 1
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --workload b64_s2_320x1024_fwdbwd --no-cpu > $out/bench_hires.json 2> $out/bench_hires.err || exit 1
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --workload b12_s2_192x640_pyramid4_fwdbwd --no-cpu > $out/bench_pyramid.json 2> $out/bench_pyramid.err || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats_b12 -o p --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu --no-also --cold-steps 0 --pipelined 0 --spin-up 0 > $out/stats_b12.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats_hires -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --workload b64_s2_320x1024_fwdbwd --no-cpu --no-also --cold-steps 0 --pipelined 0 --spin-up 0 > $out/stats_hires.log 2>&1 || exit 1
find $out -name "*kernel_trace.csv" -delete   # per-dispatch traces are large; the stats files stay
[ "${2:-}" = quick ] && { echo done-quick; exit 0; }
for wl in b12_s2_192x640_fwdbwd b64_s2_320x1024_fwdbwd; do
  for c in FETCH_SIZE WRITE_SIZE; do
    scripts/pmc_pass.sh ${tag}_${c}_${wl} $wl $c || exit 1
  done
  scripts/pmc_pass.sh ${tag}_SQ1_${wl} $wl SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY || exit 1
  scripts/pmc_pass.sh ${tag}_SQ2_${wl} $wl SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 || exit 1
done
# the plane sweep (train.py:49-60 at the reference's shape): resize + sweep + resize, the sweep kernel dominates
wl=b12_s2_sweep80_96x320
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 --workload $wl --no-cpu > $out/bench_sweep.json 2> $out/bench_sweep.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  scripts/pmc_pass.sh ${tag}_${c}_${wl} $wl $c || exit 1
done
scripts/pmc_pass.sh ${tag}_SQ1_${wl} $wl SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY || exit 1
scripts/pmc_pass.sh ${tag}_SQ2_${wl} $wl SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 || exit 1
# the finest pyramid level with the image gradient (config 3's dominant launch: the scattering warp backward)
for c in FETCH_SIZE WRITE_SIZE; do
  PMC_EXTRA="--image-grad" scripts/pmc_pass.sh ${tag}_${c}_imggrad b12_s2_192x640_fwdbwd $c || exit 1
done
PMC_EXTRA="--image-grad" scripts/pmc_pass.sh ${tag}_SQ1_imggrad b12_s2_192x640_fwdbwd SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY || exit 1
echo done
