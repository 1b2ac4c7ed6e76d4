"""usage: python tools/bench_small.py            (on the GPU box)
The call floor and the crossover against one CPU core: one-shot decodes of small zlib streams
(enwik-style text, level 6; and stored data) of 1 KiB ... 1 MiB of output through
  (a) tbz_inflate_device — input and output resident on the device: the engine's own time per call,
  (b) tbz_inflate        — ordinary host buffers: what a 3bz caller sees,
  (c) the CPU oracle (oracle/tbz_oracle.c, one core; the checker — timed here as bench.py's cpu_baseline leg is),
with the one-launch path (tbz_small_fused) on and off.  Prints a table; the crossover is the size from which (b) beats (c)."""
import importlib, os, sys, time, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import corpus as K
from oracle import oracle as O
T = importlib.import_module("3bz_amd")
O.lib()


def engine(**env):
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return T.Engine(0)
    finally:
        for k in env:
            os.environ.pop(k, None)


def per_call(f, budget=0.25):
    f(); f()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget:
        f()
        n += 1
    return (time.perf_counter() - t0) / n


e1, e0 = engine(), engine(TBZ_SMALL_FUSED=0)
print("%-26s %9s | %-23s | %-23s | %-12s" % ("stream", "out", "device buffers us (fused / general)", "host buffers us (fused / general)", "oracle 1 core us"))
for kind in ("text", "stored"):
    for n in (1 << 10, 4 << 10, 16 << 10, 64 << 10, 128 << 10, 256 << 10, 1 << 20):
        p = K.enwik_like(n, seed=n) if kind == "text" else K.xorshift64star_bytes(n, n)
        s = zlib.compress(p, 6 if kind == "text" else 0)
        out = bytearray(n)
        row = []
        for e in (e1, e0):
            d_in, d_out = e.malloc(len(s) + 64), e.malloc(n + 64)
            e.h2d(d_in, s)
            r = e.inflate_device(d_in, len(s), d_out, n, 1)
            assert r.status == 0 and r.out_len == n and r.adler32 == zlib.adler32(p)
            row.append(per_call(lambda: e.inflate_device(d_in, len(s), d_out, n, 1)))
            e.free(d_in); e.free(d_out)
        for e in (e1, e0):
            r = e.inflate(s, 1, out)
            assert r.status == 0 and bytes(out) == p
            row.append(per_call(lambda: e.inflate(s, 1, out)))
        oo = bytearray(n)
        row.append(per_call(lambda: O.decompress_vector(s, "zlib", output=oo)))
        print("%-26s %9d | %10.1f / %-10.1f | %10.1f / %-10.1f | %10.1f" % ("%s, %d B compressed" % (kind, len(s)), n, row[0] * 1e6, row[1] * 1e6, row[2] * 1e6, row[3] * 1e6, row[4] * 1e6), flush=True)
e1.close(); e0.close()
