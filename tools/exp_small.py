"""fixed costs: wall time per call vs device stage times for small inputs (TBZ_DEBUG-style breakdown)"""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0)
for mib in (1, 4, 16, 64, 256):
    U = mib << 20
    s, p, a = K.zlib_flush_stream(U, workers=16, want_plain=False)
    d_in = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
    d_out = torch.empty(U + 64, dtype=torch.uint8, device="cuda")
    for i in range(3):
        r = eng.inflate_device(d_in.data_ptr(), len(s), d_out.data_ptr(), U, 1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    N = 20
    for i in range(N):
        r = eng.inflate_device(d_in.data_ptr(), len(s), d_out.data_ptr(), U, 1)
    t1 = time.perf_counter()
    t = eng.timings()
    wall = (t1 - t0) / N * 1e3
    print("%4d MiB: wall %.3f ms (%.1f GB/s) | device span %.3f: scan %.3f huff %.3f lz %.3f ck %.3f | outside the stages %.3f" %
          (mib, wall, U / wall / 1e6, t.total_ms, t.scan_ms, t.huff_ms, t.lz_ms, t.cksum_ms,
           wall - t.scan_ms - t.huff_ms - t.lz_ms - t.cksum_ms), flush=True)
