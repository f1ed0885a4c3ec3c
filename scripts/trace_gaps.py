#!/usr/bin/env python3
"""Where a generation's wall time goes on the device: scripts/trace_gaps.py kernel_trace.csv
Reads a rocprofv3 --kernel-trace CSV, takes the LAST generation (from the last cppn_render_kernel on) and prints per kernel:
launches, busy time, and the idle gaps between consecutive dispatches (end -> next start) -- launch latency / drain that a
hipGraph or a second stream could hide."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
starts = [i for i, k in enumerate(ks) if "cppn_render" in k[2]]
seg = ks[starts[-1]:] if starts else ks
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
gaps = [max(0, seg[i + 1][0] - max(x[1] for x in seg[:i + 1][-4:])) for i in range(len(seg) - 1)]
agg = defaultdict(lambda: [0, 0])
for s, e, n in seg:
    short = n.split("(")[0].replace("void eig::", "")[:70]
    agg[short][0] += 1
    agg[short][1] += e - s
print("last generation: %d dispatches, span %.3f ms, busy %.3f ms (%.1f %%), idle between dispatches %.3f ms; median gap %.1f us"
      % (len(seg), span / 1e6, busy / 1e6, 100.0 * busy / span, sum(gaps) / 1e6, sorted(gaps)[len(gaps) // 2] / 1e3))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
    print("%8.3f ms %5d x %7.1f us  %s" % (t / 1e6, c, t / c / 1e3, n))
