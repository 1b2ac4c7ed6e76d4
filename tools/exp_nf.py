"""experiment: stage times of one no-flush stream (usage: python tools/exp_nf.py [MiB] [level])"""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import corpus as K
T = importlib.import_module("3bz_amd")
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lvl = int(sys.argv[2]) if len(sys.argv) > 2 else 6
p = K.enwik_like(mib << 20, 0x3B2)
s = zlib.compress(p, lvl)
eng = T.Engine(0)
d_in, d_out = eng.malloc(len(s) + 64), eng.malloc(len(p) + 64)
eng.h2d(d_in, s)
for it in range(3):
    t0 = time.perf_counter()
    r = eng.inflate_device(d_in, len(s), d_out, len(p), 1)
    dt = time.perf_counter() - t0
    t = eng.timings()
    print("status %d %.2f ms (%.1f GB/s) scan %.2f find %.2f huff %.2f lz %.2f resolve %.2f ck %.2f | items %d groups %d H %d cands %d gang %d" % (
        r.status, dt * 1e3, len(p) / dt / 1e9, t.scan_ms, t.find_ms, t.huff_ms, t.lz_ms, t.resolve_ms, t.cksum_ms, t.n_segments, t.n_groups, t.n_hgroups, t.n_candidates, t.k1_gang), flush=True)
out = bytearray(len(p))
eng.d2h(out, d_out)
print("ok", bytes(out) == p)
