"""feasibility: do two engines (two HIP streams) overlap on one GPU?  Two threads, each decoding its own stream."""
import importlib, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from tools import corpus as K
T = importlib.import_module("3bz_amd")
U = int(os.environ.get("EXP_MIB", "512")) << 20
NT = int(os.environ.get("EXP_THREADS", "2"))
ins = []
for i in range(NT):
    s, p, a = K.zlib_flush_stream(U, seed=0x3B2 + i, workers=16, want_plain=False)
    d_in = torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()).cuda()
    d_out = torch.empty(U + 64, dtype=torch.uint8, device="cuda")
    ins.append((len(s), d_in, d_out, T.Engine(0)))
def run(i, reps, offset_ms=0.0):
    C, d_in, d_out, eng = ins[i]
    if offset_ms: time.sleep(offset_ms / 1e3)
    for _ in range(reps):
        r = eng.inflate_device(d_in.data_ptr(), C, d_out.data_ptr(), U, 1)
        assert r.status == 0
for i in range(NT): run(i, 2)
torch.cuda.synchronize()
t0 = time.perf_counter(); run(0, 6); torch.cuda.synchronize(); t1 = time.perf_counter()
print("one engine alone: %.2f ms per %d MiB call" % ((t1 - t0) / 6 * 1e3, U >> 20))
for off in (0.0, 1.5, 3.0):
    ths = [threading.Thread(target=run, args=(i, 6, off * i)) for i in range(NT)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("%d engines concurrently (start offset %.1f ms): %.2f ms per round of %d x %d MiB -> %.1f GB/s aggregate" %
          (NT, off, (t1 - t0) / 6 * 1e3, NT, U >> 20, NT * U * 6 / (t1 - t0) / 1e9))
