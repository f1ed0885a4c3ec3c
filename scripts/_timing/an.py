import numpy as np, glob, os
for path in sorted(glob.glob("gpurun_out/timeline_H*.bin")):
    r = np.fromfile(path, dtype=np.uint64).reshape(-1, 4, 8)
    r = r[r[:, 0, 0] != 0]
    te, t0, t1, t2 = (r[:, :, i].astype(np.int64) for i in range(4))
    ts, tp = r[:, :, 6].astype(np.int64), r[:, :, 7].astype(np.int64)
    print(os.path.basename(path), "entry->setup %.0f  setup->prewait(DMA issue, addr tables, acc zero) %.0f  wait+barrier(+up loads issue) %.0f | loop %.0f epi %.0f" % (
        np.median(ts - te), np.median(tp - ts), np.median(t0 - tp), np.median(t1 - t0), np.median(t2 - t1)))
