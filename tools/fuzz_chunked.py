"""usage: [TBZ_...] [EMU_LIB=path] python tools/fuzz_chunked.py <seed> <seconds>
the chunked protocol (more input after input-underrun, a new buffer after output-overflow) in random input chunks and
output buffer sizes, engine against the oracle call by call (tests/parity_cases.py:_chunked_lockstep): full-flush,
sync-flush, no-flush, truncated and DAMAGED streams in the three containers — exercises the device-resident session
(tbz_session_*): resume points, the window carried across calls, pending output, errors surfacing in the reference's call."""
import importlib, os, random, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import parity_cases as P
from tools import corpus as K
T = importlib.import_module("3bz_amd")
eng = T.Engine(0, lib_path=os.environ.get("EMU_LIB", os.path.join(ROOT, "tests", "emu", "libtbz_emu.so")))
rng = random.Random(int(sys.argv[1]))
fp = K.enwik_like(70000, seed=9)
def flushed(wbits, mode, every, level):
    c = zlib.compressobj(level, zlib.DEFLATED, wbits)
    return b"".join(c.compress(fp[i:i + every]) + c.flush(mode) for i in range(0, len(fp), every)) + c.flush()
t0 = time.time(); n = 0; based = 0; errs = 0
while time.time() - t0 < float(sys.argv[2]):
    fmt, wbits = rng.choice([("zlib", 15), ("gzip", 31), ("deflate", -15)])
    mode = rng.choice([zlib.Z_FULL_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_SYNC_FLUSH, zlib.Z_NO_FLUSH])
    blob = flushed(wbits, mode, rng.choice([1000, 4096, 9000, 20000]), rng.choice([0, 1, 6]))
    if rng.random() < 0.2: blob = blob[:rng.randrange(1, len(blob))]
    if rng.random() < 0.25:   # damage: the error (or whatever it turns into) has to surface in the same call
        b = bytearray(blob)
        for _ in range(rng.randrange(1, 3)):
            b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        blob = bytes(b)
    steps = [rng.randrange(1, 15000) for _ in range(rng.randrange(1, 6))]
    if sum(steps) * 1500 < len(blob) * len(steps): steps.append(15000)   # (the lockstep harness allows 2000 chunks)
    sizes = [rng.randrange(1, 40000) for _ in range(rng.randrange(1, 5))]
    P._chunked_lockstep(eng, blob, fmt, steps, sizes, "chunk fuzz %d %s %s %s" % (n, fmt, steps, sizes))
    st = P._chunked_lockstep.last_state; n += 1
    based += bool(st.result is not None and st.result.boundary_out > 0); errs += bool(getattr(P._chunked_lockstep, "last_error", None))
print("chunk fuzz ok:", n, "streams; the resume point moved in", based, "; ended in the same error as the oracle in", errs)
