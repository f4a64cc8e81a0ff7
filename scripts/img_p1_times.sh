#!/bin/bash
# pass-1 / reducer kernel times of the image-gradient step for several library builds (rocprofv3 kernel trace, smooth depth)
export TMPDIR=/tmp
for l in "$@"; do
  rm -rf /tmp/p1t; timeout -s KILL 120 rocprofv3 --kernel-trace --stats -d /tmp/p1t -o p --output-format csv -- python3 scripts/img_lib_bench.py $l smooth > /tmp/p1t.log 2>&1 || exit 1
  python3 - "$l" <<'PY'
import csv, glob, sys
f = glob.glob('/tmp/p1t/**/*kernel_stats.csv', recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    if 'warp_bwd_kernel' in r['Name'] or 'img_reduce' in r['Name']:
        out.append(('pass1' if 'warp_bwd' in r['Name'] else 'reducer') + ' %.1f us' % (float(r['AverageNs']) / 1e3))
print(sys.argv[1], ' '.join(out))
PY
done
