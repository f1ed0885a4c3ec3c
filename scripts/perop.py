#!/usr/bin/env python3
"""Per-operator table of bench.py JSON lines side by side: scripts/perop.py a.json b.json ..."""
import json
import sys
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
keys = []
for r in runs:
    for o in r["roofline"]["per_op"]:
        k = (o["layer"], o["op"])
        if k not in keys:
            keys.append(k)
print("%-44s" % "layer op" + "".join("%26s" % f.split("/")[-1][:24] for f in sys.argv[1:]))
for k in keys:
    row = "%d %-42s" % (k[0], k[1][:42])
    for r in runs:
        o = [x for x in r["roofline"]["per_op"] if (x["layer"], x["op"]) == k]
        row += "%14.2f ms %6.1f TF" % (o[0]["ms"], o[0]["tflops"]) if o else "%26s" % "-"
    print(row)
print("%-44s" % "value evals/s" + "".join("%26.2f" % r["value"] for r in runs))
print("%-44s" % "all conv ms" + "".join("%26.2f" % r["roofline"]["all_conv_kernels"]["total_ms"] for r in runs))
