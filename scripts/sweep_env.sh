#!/bin/bash
# scripts/sweep_env.sh VAR v1 v2 ...  -- kbench of the one-call step at both sizes for each value of an environment switch
var=$1; shift
l=depth_estimation_amd/libdepthsynth.so
for v in "$@"; do
  x=$(env $var=$v timeout -k 10 120 scripts/kbench $l 12 2 192 640 300 2>&1 | grep "one call" | sed 's/.*: //;s/ per step.*//') || exit 1
  y=$(env $var=$v timeout -k 10 120 scripts/kbench $l 64 2 320 1024 30 2>&1 | grep "one call" | sed 's/.*: //;s/ per step.*//') || exit 1
  echo "$var=$v   b12: $x   hires: $y"
done
