"""tbz_k2_lz77_dual trace (tools/trace/run.sh k2): 8 u64 per workgroup (the first 8192): t0, t1, front-end wave's barrier
wait, batches, resolve wave's barrier wait, its end, front end's last hand-over"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
life = (t1 - t0) / 100.0
k = a[:, 3].astype(np.float64)
w0 = a[:, 2].astype(np.float64) / 100.0
w1 = a[:, 4].astype(np.float64) / 100.0
fin = (a[:, 6].astype(np.int64) - t0) / 100.0   # when the front end had handed over its last batch
print("workgroups %d; lifetime mean %.1f us p50 %.1f p99 %.1f; batches mean %.1f; %.3f us per batch" % (len(a), life.mean(), *np.percentile(life, [50, 99]), k.mean(), life.sum() / k.sum()))
print("front end: waits %.1f us (%.0f%%); resolve wave: waits %.1f us (%.0f%%); final flush + rest after the last batch %.1f us (%.0f%%)" % (
    w0.mean(), 100 * w0.sum() / life.sum(), w1.mean(), 100 * w1.sum() / life.sum(), (life - fin).mean(), 100 * (life - fin).sum() / life.sum()))
span = (t1.max() - t0.min()) / 100.0
print("span of the sampled workgroups %.1f us" % span)
