import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from tools import corpus as K
T = importlib.import_module("3bz_amd")
U = int(os.environ.get("EXP_MIB", "256")) << 20
s, p, a = K.zlib_flush_stream(U, workers=16, want_plain=False)
d_in = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
d_out = torch.empty(U + 64, dtype=torch.uint8, device="cuda")
for lib in sys.argv[1:]:
    eng = T.Engine(0, lib_path=lib)
    best = None
    for i in range(4):
        r = eng.inflate_device(d_in.data_ptr(), len(s), d_out.data_ptr(), U, 1)
        t = eng.timings()
        if i: best = (t.huff_ms, t.lz_ms, t.scan_ms, t.cksum_ms, t.total_ms) if best is None or t.huff_ms < best[0] else best
    print(os.path.basename(lib), "status", r.status, "huff %.2f lz %.2f scan %.2f ck %.2f total %.2f" % best, flush=True)
    eng.close()
