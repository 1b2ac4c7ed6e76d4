import importlib, os, random, sys, time, zlib
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from tests import parity_cases as P
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0, lib_path=os.environ.get("EMU_LIB"))
rng = random.Random(41)
fp = K.enwik_like(70000, seed=9)
def flushed(wbits, mode, every, level):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    return b"".join(c.compress(fp[i:i + every]) + c.flush(mode) for i in range(0, len(fp), every)) + c.flush()
n = 0
while True:
    fmt, wbits = rng.choice([("zlib", 15), ("gzip", 31), ("deflate", -15)])
    mode = rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH])
    blob = flushed(wbits, mode, rng.choice([1000, 4096, 9000, 20000]), rng.choice([0, 1, 6]))
    if rng.random() < 0.2: blob = blob[:rng.randrange(1, len(blob))]
    steps = [rng.randrange(1, 15000) for _ in range(rng.randrange(1, 6))]
    sizes = [rng.randrange(1, 40000) for _ in range(rng.randrange(1, 5))]
    if n >= int(sys.argv[1]):
        try:
            P._chunked_lockstep(eng, blob, fmt, steps, sizes, "case %d" % n)
        except AssertionError as e:
            print("FAIL", n, e.args[0][:3], flush=True)
            for rep in range(2):
                out = bytearray(sizes[0]); r = eng.inflate(blob[:steps[0]], P.FMT[fmt], out)
                print(" same engine again:", r.status, r.out_len, r.out_total, r.in_consumed, r.segments)
            e2 = T.Engine(0, lib_path=os.environ.get("EMU_LIB"))
            out = bytearray(sizes[0]); r = e2.inflate(blob[:steps[0]], P.FMT[fmt], out)
            print(" fresh engine:", r.status, r.out_len, r.out_total)
            break
    n += 1
    if n > 990: break
print("done", n)
