#!/bin/bash
# A/B of two builds of the library on one box, alternating: scripts/ab.sh <libA> <libB> [rounds]
a=$1; b=$2; n=${3:-3}
for i in $(seq $n); do
  for l in $a $b; do
    x=$(timeout -k 10 120 scripts/kbench $l 12 2 192 640 300 2>&1 | grep "one call" | sed 's/.*: //') || exit 1
    y=$(timeout -k 10 120 scripts/kbench $l 64 2 320 1024 30 2>&1 | grep "one call" | sed 's/.*: //') || exit 1
    echo "$l   b12: $x   hires: $y"
  done
done
