#!/bin/bash
# A/B of one library with and without an environment switch, alternating: scripts/ab_env.sh VAR=1 [rounds]
v=$1; n=${2:-3}; l=depth_estimation_amd/libdepthsynth.so
for i in $(seq $n); do
  for e in "X_=0" "$v"; do
    x=$(env $e timeout -k 10 120 scripts/kbench $l 12 2 192 640 300 2>&1 | grep "one call" | sed 's/.*: //') || exit 1
    y=$(env $e timeout -k 10 120 scripts/kbench $l 64 2 320 1024 30 2>&1 | grep "one call" | sed 's/.*: //') || exit 1
    echo "$e   b12: $x   hires: $y"
  done
done
