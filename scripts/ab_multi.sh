#!/bin/bash
# scripts/ab_multi.sh rounds "ENV=.. lib" "ENV=.. lib" ...   -- alternating kbench runs of several (environment, library) pairs on one box
n=$1; shift
for i in $(seq $n); do
  for spec in "$@"; do
    e=${spec%% *}; l=${spec##* }
    x=$(env $e timeout -k 10 120 scripts/kbench $l 12 2 192 640 300 2>&1 | grep "one call" | sed 's/.*: //;s/ per step.*//') || exit 1
    y=$(env $e timeout -k 10 120 scripts/kbench $l 64 2 320 1024 30 2>&1 | grep "one call" | sed 's/.*: //;s/ per step.*//') || exit 1
    echo "$e $l   b12: $x   hires: $y"
  done
done
