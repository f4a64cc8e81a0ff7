#!/bin/bash
# Round profile: bench lines + rocprofv3 kernel stats + HBM-traffic PMC passes, everything under gpurun_out/<tag>/
#   scripts/profile_round.sh <tag>
set -u
tag=${1:-r01}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import torch" || exit 1
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $out/bench_b12.json 2> $out/bench_b12.err || exit 1
timeout -k 10 300 python3 bench.py --steps 50 --warmup 10 --workload b64_s2_320x1024_fwdbwd --no-cpu > $out/bench_hires.json 2> $out/bench_hires.err || exit 1
timeout -k 10 300 python3 bench.py --steps 100 --warmup 10 --workload b12_s2_192x640_pyramid4_fwdbwd --no-cpu > $out/bench_pyramid.json 2> $out/bench_pyramid.err || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats_b12 -o p --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu --streams 1 > $out/stats_b12.log 2>&1 || exit 1
timeout -s KILL 240 rocprofv3 --kernel-trace --stats -d $out/stats_hires -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --workload b64_s2_320x1024_fwdbwd --no-cpu --streams 1 > $out/stats_hires.log 2>&1 || exit 1
for wl in b12_s2_192x640_fwdbwd b64_s2_320x1024_fwdbwd; do
  for c in FETCH_SIZE WRITE_SIZE; do
    scripts/pmc_pass.sh ${tag}_${c}_${wl} $wl $c || exit 1
  done
done
echo done
