"""timing of the other BASELINE shapes (parity is the tests' job; this only watches for performance cliffs)"""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0)
def run(name, s, fmt, U):
    d_in = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
    d_out = torch.empty(U + 64, dtype=torch.uint8, device="cuda")
    best = None
    for i in range(3):
        t0 = time.perf_counter()
        r = eng.inflate_device(d_in.data_ptr(), len(s), d_out.data_ptr(), U, fmt)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        t = eng.timings()
        if best is None or dt < best[0]: best = (dt, t.huff_ms, t.lz_ms, t.n_segments, t.n_groups)
    print("%-34s status %d  %.2f ms  %.1f GB/s  (huff %.2f lz %.2f, %d segs %d groups)" % (name, r.status, best[0] * 1e3, U / best[0] / 1e9, best[1], best[2], best[3], best[4]), flush=True)
U = 64 << 20
s, p, a = K.zlib_flush_stream(U, workers=16, want_plain=False)
run("config 2 shape 64 MiB", s, 1, U)
s, p, a = K.zlib_flush_stream(U, flush=zlib.Z_SYNC_FLUSH, workers=1, want_plain=False)
run("sync-flush 64 MiB (one group)", s, 1, U)
plain = K.enwik_like(U, seed=3)
z = zlib.compress(plain, 6)
run("one zlib stream, no flush, 64 MiB", z, 1, U)
s, p = K.adversarial_stream(total=32 << 20)
run("config 5 adversarial 32 MiB", s, 1, len(p))
s, p, a = K.zlib_flush_stream(U, block=1000, workers=16, want_plain=False)
run("1000-octet flush blocks 64 MiB", s, 1, U)
