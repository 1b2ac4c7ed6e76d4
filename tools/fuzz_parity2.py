"""usage: [TBZ_K1_MODE=...] python tools/fuzz_parity2.py <seed> <seconds>
second fuzz: gzip and zlib containers, random output capacities (overflow), and BATCHES of independently corrupted
streams through tbz_inflate_batch — engine (lane-emulator build) against the oracle, stream by stream."""
import importlib, os, random, sys, zlib, time, gzip as pygzip
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity_cases as P
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0, lib_path=os.environ.get("EMU_LIB", os.path.join(ROOT, "tests", "emu", "libtbz_emu.so")))
rng = random.Random(int(sys.argv[1]))
plain = P._mixed_plain(40000, 21)
bases = {"zlib": [zlib.compress(plain, 6), K.zlib_flush_stream(30000, block=2048)[0], zlib.compress(plain[:9000], 0)],
         "gzip": [pygzip.compress(plain, 6, mtime=0), pygzip.compress(plain[:5000], 9, mtime=0)],
         "deflate": [zlib.compress(plain, 9)[2:-4]]}
def corrupt(b):
    b = bytearray(b)
    for _ in range(rng.randrange(0, 3)):
        mode = rng.randrange(4)
        if mode == 0: b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif mode == 1: b = b[: rng.randrange(1, max(2, len(b)))]
        elif mode == 2:
            i = rng.randrange(len(b)); b[i:i] = bytes([0, 0, 255, 255])
        else: b[rng.randrange(len(b))] = rng.randrange(256)
    return bytes(b)
t0 = time.time(); n = 0; nb = 0; stale = 0
while time.time() - t0 < float(sys.argv[2]):
    fmt = rng.choice(list(bases))
    if rng.random() < 0.5:
        b = corrupt(rng.choice(bases[fmt]))
        cap = rng.choice([50000, 50000, rng.randrange(0, 45000)])
        stale += P.same_or_stale_tables(eng, b, fmt, cap, what="fuzz2 %d" % n)[1]
        n += 1
    else:
        k = rng.randrange(2, 7)
        datas = [corrupt(rng.choice(bases[fmt])) for _ in range(k)]
        caps = [rng.choice([50000, rng.randrange(0, 45000)]) for _ in range(k)]
        outs = [bytearray(c) for c in caps]
        res = eng.inflate_batch(datas, P.FMT[fmt], outs)
        for i in range(k):
            r = res[i]
            flag = "error" if r.status < 0 else ("finished", "underrun", "overflow")[r.status]
            for fresh in (False, True):   # second try: the documented stale-table deviation (parity_cases.same_or_stale_tables)
                P.O.set_fresh_tables(fresh)
                try:
                    want = P.oracle_oneshot(datas[i], fmt, caps[i])
                finally:
                    P.O.set_fresh_tables(False)
                ok = flag == want["flag"] and (r.status == want["code"] if flag == "error" else
                                               r.out_len == want["offset"] and bytes(outs[i][: r.out_len]) == want["bytes"])
                if ok:
                    stale += fresh
                    break
            assert ok, ("batch", nb, i, flag, want["flag"], r.status, want["code"])
        nb += 1
print("fuzz2 single", n, "batches", nb, "clean; stale-table deviations (documented):", stale)
