"""config-3-shaped timing: N gzip members of 256 KiB decoded as a batch (device-resident), per K1 flavour"""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
from tools import corpus as K
T = importlib.import_module("3bz_amd")
n_members = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blob, offs, plains = K.gzip_members(n_members, 256 << 10, workers=16)
ends = offs[1:] + [len(blob)]
lens = [e - o for o, e in zip(offs, ends)]
U = sum(len(p) for p in plains)
d_in = torch.from_numpy(np.frombuffer(blob, dtype=np.uint8).copy()).cuda()
d_out = torch.empty(U + 64, dtype=torch.uint8, device="cuda")
out_offs = [i * (256 << 10) for i in range(n_members)]
caps = [256 << 10] * n_members
for mode in ("gang8", "gang16", "gang32", "gang64", "auto"):
    if mode != "auto": os.environ["TBZ_K1_MODE"] = mode
    else: os.environ.pop("TBZ_K1_MODE", None)
    eng = T.Engine(0)
    best = None
    for i in range(3):
        t0 = time.perf_counter()
        res = eng.inflate_batch_device(d_in.data_ptr(), offs, lens, d_out.data_ptr(), out_offs, caps, 2)
        dt = time.perf_counter() - t0
        t = eng.timings()
        if best is None or dt < best[0]: best = (dt, t.huff_ms, t.lz_ms, t.cksum_ms, t.scan_ms)
    ok = all(r.status == 0 for r in res) and res[3].crc32 == zlib.crc32(plains[3])
    got = d_out[:U].cpu().numpy()
    ok = ok and bytes(got[: 256 << 10]) == plains[0] and bytes(got[-(256 << 10):]) == plains[-1]
    print("%s: ok=%s %d members %.1f MiB: call %.2f ms (%.1f GB/s) huff %.2f lz %.2f ck %.2f scan %.2f" %
          (mode, ok, n_members, U / 2**20, best[0] * 1e3, U / best[0] / 1e9, best[1], best[2], best[3], best[4]), flush=True)
    eng.close()
