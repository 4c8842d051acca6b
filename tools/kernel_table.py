"""Per-kernel table of a bench line's in-loop measurement (roofline.all_kernels): python tools/kernel_table.py line.json [rows]"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 14
print("value %.0f  ms/step %.2f" % (d["value"], d["ms_per_step"]))
ak = d["roofline"]["all_kernels"]
tot = sum(v["ms_per_step"] for v in ak.values())
print("sum of kernel ms/step %.1f" % tot)
for k, v in sorted(ak.items(), key=lambda kv: -kv[1]["ms_per_step"])[:rows]:
    print("%-28s %5.1f x %8.1f us = %6.2f ms" % (k, v["launches_per_step"], v["avg_launch_us"], v["ms_per_step"]))
