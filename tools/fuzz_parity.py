"""usage: [TBZ_K1_MODE=...] python tools/fuzz_parity.py <seed> <seconds>\nrandom corruptions of four stream shapes, engine (lane-emulator build) against the oracle; see tests/parity_cases.py:case_fuzz"""
import importlib, os, random, sys, zlib, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity_cases as P
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0, lib_path=os.environ.get("EMU_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "emu", "libtbz_emu.so")))
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
plain = P._mixed_plain(60000, 9)
bases = [("zlib", zlib.compress(plain, 6)), ("zlib", K.zlib_flush_stream(50000, block=4096)[0]),
         ("deflate", zlib.compress(plain, 1)[2:-4]), ("zlib", K.zlib_flush_stream(30000, block=1000, flush=zlib.Z_SYNC_FLUSH)[0])]
t0 = time.time(); n = 0; kinds = {}
while time.time() - t0 < float(sys.argv[2]) if len(sys.argv) > 2 else 300:
    fmt, b = rng.choice(bases)
    b = bytearray(b)
    for _ in range(rng.randrange(1, 4)):
        mode = rng.randrange(4)
        if mode == 0: b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif mode == 1: b = b[: rng.randrange(1, max(2, len(b)))]
        elif mode == 2:
            i = rng.randrange(len(b)); b[i:i] = bytes([0, 0, 255, 255])
        else:
            i = rng.randrange(len(b)); b[i] = rng.randrange(256)
    w, stale = P.same_or_stale_tables(eng, bytes(b), fmt, 70000, what="fuzz %d" % n)
    kinds["stale-table deviation"] = kinds.get("stale-table deviation", 0) + (1 if stale else 0)
    kinds[w["flag"]] = kinds.get(w["flag"], 0) + 1
    n += 1
print("fuzz cases", n, kinds)
