#!/usr/bin/env python3
"""Do kernels of two streams really run at the same time?  scripts/trace_overlap.py kernel_trace.csv
Reads a rocprofv3 --kernel-trace CSV, takes the LAST generation (from the last cppn_render_kernel on) and prints: dispatches per
queue, the span, the time at least one kernel runs, the time at least TWO run (true overlap), and the sum of kernel durations."""
import csv
import sys
from collections import Counter

rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
starts = [i for i, k in enumerate(ks) if "cppn_render" in k[2]]
seg = ks[starts[-1]:] if starts else ks
ev = sorted([(s, 1) for s, e, _, _ in seg] + [(e, -1) for s, e, _, _ in seg])
depth, last, t1, t2 = 0, ev[0][0], 0, 0
for t, d in ev:
    if depth >= 1:
        t1 += t - last
    if depth >= 2:
        t2 += t - last
    depth += d
    last = t
span = seg[-1][1] - seg[0][0]
tot = sum(e - s for s, e, _, _ in seg)
print("last generation: %d dispatches on queues %s" % (len(seg), dict(Counter(q for _, _, _, q in seg))))
print("span %.3f ms | >= 1 kernel running %.3f ms (%.1f %%) | >= 2 running %.3f ms (%.1f %% of span) | sum of kernel durations %.3f ms"
      % (span / 1e6, t1 / 1e6, 100.0 * t1 / span, t2 / 1e6, 100.0 * t2 / span, tot / 1e6))
