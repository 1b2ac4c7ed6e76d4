"""tbz_k0b_validate trace (tools/trace/run.sh k0b): 8 u64 per tile: t0, t1, trips of the symbol loop, candidates, kept"""
import sys, numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] != 0]
t0 = a[:, 0].astype(np.int64); t1 = a[:, 1].astype(np.int64)
life = (t1 - t0) / 100.0
trips = a[:, 2].astype(np.int64)
print("tiles", len(a), "span %.1f us" % ((t1.max() - t0.min()) / 100.0), "life mean %.1f p50 %.1f p99 %.1f max %.1f" % (life.mean(), *np.percentile(life, [50, 99]), life.max()))
print("trips mean %.1f p50 %d p99 %d max %d; count mean %.1f max %d; kept %d" % (trips.mean(), *np.percentile(trips, [50, 99]), trips.max(), a[:, 3].mean(), a[:, 3].max(), a[:, 4].sum()))
o = np.argsort(-life)[:8]
for i in o: print("  tile", i, "life %.1f us trips %d count %d kept %d -> %.3f us/trip" % (life[i], trips[i], a[i, 3], a[i, 4], life[i] / max(1, trips[i])))
print("launch spread: first start 0, last start %.1f us" % ((t0.max() - t0.min()) / 100.0))
