#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (scripts/pmc_passes.sh) per kernel -> JSON (committed under profiles/).

HBM bytes follow MI355X_MICROARCH.md section HBM: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B taken from
TCC_EA0_RDREQ/WRREQ; on gfx950 FETCH_SIZE reports HALF the bytes of wide (16 B/lane) coalesced reads -- the conv kernel's
loads are all 16 B/lane DMA, so the read side is doubled; WRITE_SIZE is used as reported (uncalibrated)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(d, out, pop):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(int)
    dur = collections.defaultdict(float)
    for name in ("sq", "fetch", "write", "lds"):
        try:
            rows = list(csv.DictReader(open("%s/%s/%s_counter_collection.csv" % (d, name, name))))
        except FileNotFoundError:
            continue
        for r in rows:
            agg[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    for r in csv.DictReader(open("%s/sq/sq_kernel_trace.csv" % d)):
        calls[r["Kernel_Name"]] += 1
        dur[r["Kernel_Name"]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    # stamp: bench.py reports `roofline.traffic` from this file only when the kernel sources are the ones it runs on
    import bench
    res = {"pop": pop, "kernel_sources_sha": bench.kernel_sources_sha(),
           "commit": os.environ.get("EIGEN_COMMIT") or bench.git_head(), "kernels": {}}
    for k in sorted(agg, key=lambda x: -dur[x])[:8]:
        v = agg[k]
        n = max(calls[k], 1)
        e = {"calls": calls[k], "total_ms": dur[k], "avg_ms": dur[k] / n, "counters": dict(v)}
        if "FETCH_SIZE" in v:
            e["hbm_read_bytes_per_launch"] = 2.0 * v["FETCH_SIZE"] * 1024 / n
        if "WRITE_SIZE" in v:
            e["hbm_write_bytes_per_launch"] = v["WRITE_SIZE"] * 1024 / n
        if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs on the chip
            e["mfma_pipe_busy_frac"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / (v["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            e["gfx_clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8.0 / (dur[k] * 1e-3) * 1e-9
        if "TCC_HIT_sum" in v:
            e["l2_hit_rate"] = v["TCC_HIT_sum"] / max(v["TCC_HIT_sum"] + v["TCC_MISS_sum"], 1.0)
        if "SQ_LDS_BANK_CONFLICT" in v:
            e["lds_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1.0)
        res["kernels"][k] = e
    json.dump(res, open(out, "w"), indent=1)
    for k, e in res["kernels"].items():
        print(k[:70], {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in e.items() if kk != "counters"})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 256)
