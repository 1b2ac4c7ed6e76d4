#!/usr/bin/env python3
"""K1g wave trace (tools/trace/run.sh k1): 16 u64 per workgroup, written by the -DTBZ_WAVE_TRACE build:
t0, t1 (100 MHz), XCC|HW_ID, header+commit, build, rounds, round 1, number of rounds, round 1: wave trips / lane
iterations / phases, later rounds: wave trips / lane iterations"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16)
a = a[a[:, 0] != 0]
t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
org = t0.min()
t0 -= org
t1 -= org
life = (t1 - t0) / 100.0  # us at 100 MHz
print("workgroups %d; span %.1f us; lifetime us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (
    len(a), t1.max() / 100.0, life.mean(), *np.percentile(life, [50, 90, 99]), life.max()))
span = t1.max()
# concurrency over time
ev = np.zeros(span + 2, dtype=np.int64)
np.add.at(ev, t0, 1)
np.add.at(ev, t1, -1)
conc = np.cumsum(ev)
for k in range(0, 10):
    lo, hi = span * k // 10, span * (k + 1) // 10
    print("  %3d%%..%3d%%: mean resident workgroups %.0f" % (k * 10, k * 10 + 10, conc[lo:hi].mean()))
for name, col in (("hdr+commit", 3), ("build", 4), ("rounds", 5), ("round 1", 6)):
    v = a[:, col].astype(np.float64) / 100.0
    print("  %-7s mean %.1f us (%.0f%% of lifetime)" % (name, v.mean(), 100 * v.sum() / life.sum()))
print("  rounds per workgroup: mean %.2f max %d" % (a[:, 7].mean(), a[:, 7].max()))
hw = a[:, 2]
xcc = (hw >> 32) & 0xF
print("  XCC histogram:", np.bincount(xcc.astype(np.int64)))
order = np.argsort(t0)
print("  launch time of workgroup #k (us): ", [(int(k), round(t0[order[k]] / 100.0, 1)) for k in (0, len(a) // 8, len(a) // 4, len(a) // 2, 3 * len(a) // 4, len(a) - 1)])

wt1, li1, ph1, wt2, li2 = (a[:, k].astype(np.float64) for k in (8, 9, 10, 11, 12))
print("  round 1: wave trips mean %.0f, phases %.1f, lane-iterations %.0f (%.0f%% of 64 x trips); later rounds: trips %.0f, lane-iterations %.0f"
      % (wt1.mean(), ph1.mean(), li1.mean(), 100 * li1.sum() / (64 * wt1.sum()), wt2.mean(), li2.mean()))
r1 = a[:, 6].astype(np.float64) / 100.0
print("  round 1: %.3f us per wave trip; later rounds: %.3f us per wave trip" % (r1.sum() / wt1.sum(), (a[:, 5].astype(np.float64) / 100.0 - r1).sum() / max(1.0, wt2.sum())))

top = np.argsort(-life)[:8]
print("  longest workgroups: (us, rounds, round-1 trips, later trips, start us)")
for i in top:
    print("    %.1f us, %d rounds, %d + %d trips, started at %.1f us, hdr+commit %.1f build %.1f rounds %.1f" % (
        life[i], a[i, 7], a[i, 8], a[i, 11], t0[i] / 100.0, a[i, 3] / 100.0, a[i, 4] / 100.0, a[i, 5] / 100.0))
