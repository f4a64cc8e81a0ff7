#!/bin/bash
# Plane-sweep profile: kernel stats + two SQ counter passes of scripts/ps_bench.py (24 sources of one target, then the
# 12x2 batch form with its resizes).  Output under gpurun_out/<tag>/.
#   scripts/profile_sweep.sh <tag>
set -u
tag=${1:-r01_sweep}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 -c "import torch" || exit 1
timeout -k 10 120 python3 scripts/ps_bench.py 24 > $out/ps_bench.txt 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $out/stats -o p --output-format csv -- python3 scripts/ps_bench.py 24 > $out/stats.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace -d $out/pmc1 -o p --output-format csv -- python3 scripts/ps_bench.py 24 > $out/pmc1.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $out/pmc1 > $out/pmc1.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH --kernel-trace -d $out/pmc2 -o p --output-format csv -- python3 scripts/ps_bench.py 24 > $out/pmc2.log 2>&1 || exit 1
python3 scripts/pmc_summary.py $out/pmc2 > $out/pmc2.txt
echo done
