#!/bin/bash
# One rocprofv3 counter pass over a short bench.py run (whole batch on one stream: every dispatch of a kernel is the same
# full-batch launch); each pass is its own process (counters are not split over passes).
#   [PMC_EXTRA="--image-grad"] scripts/pmc_pass.sh <tag> <workload> <counter> [<counter> ...]
# Output: gpurun_out/pmc_<tag>.txt (per-kernel means via scripts/pmc_summary.py; the raw CSV is deleted afterwards)
set -u
tag=$1; wl=$2; shift 2
out=gpurun_out/pmc_$tag
export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace -d "$out" -o p --output-format csv -- python3 bench.py --workload "$wl" --steps 3 --warmup 1 --no-cpu --no-also --cold-steps 0 --pipelined 0 --spin-up 0 ${PMC_EXTRA:-} > "$out.log" 2>&1
rc=$?
echo "pass $tag rc=$rc"
[ $rc -eq 0 ] || exit $rc
python3 scripts/pmc_summary.py "$out" > "$out.txt"
rm -rf "$out"   # the raw per-dispatch CSV is large (gpurun copies back at most 64 MiB); the per-kernel means are kept
