"""Print the main figures of a bench.py JSON line: python scripts/show_line.py <file>"""
import json
import sys

d = json.load(open(sys.argv[1]))
print(d["config"]["workload"], d["value"], d["unit"], d["ms_per_step"], "ms", d["kernels_ms"],
      "one stream:", (d.get("levels_on_one_stream") or {}).get("value"))
