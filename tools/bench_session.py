#!/usr/bin/env python3
"""Sessions (tbz_session_*): calls per second and octets decoded again, for input that arrives in small chunks.

    python tools/bench_session.py [--mib 16] [--chunk 3] [--max-calls 20000] [--lib tests/emu/libtbz_emu.so]

The stream is ONE fixed-Huffman block of `--mib` MiB of output (a few literals, then matches of length 258: 13 bits per
token, so a 3-octet chunk is about two tokens) — the worst case for a resume point that can only stand at a block
start: every call would decode the block's prefix again, O(n^2).  With the token-granular resume point a call costs its
new input plus one block header: `re-decode factor` = input octets handed to the engine / octets fed stays flat, and
the second half of the calls takes as long as the first.  Output octets are checked against the plaintext."""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import corpus as K  # noqa: E402


def one_block_stream(n_out):
    w = K.FixedHuffmanWriter()
    out = bytearray()
    w.begin_block(True)
    for b in b"3bz on an MI355X: ":
        w.literal(b)
        out.append(b)
    k = 0
    while len(out) + 258 <= n_out:
        d = (1, 7, 18, 3)[k & 3]
        w.match(258, d)
        K._lz_apply(out, 258, d)
        k += 1
    w.end_block()
    w.align()
    return w.getvalue(), bytes(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=float, default=16)
    ap.add_argument("--chunk", type=int, default=3)
    ap.add_argument("--max-calls", type=int, default=20000)
    ap.add_argument("--lib", default=None)
    a = ap.parse_args()
    T = importlib.import_module("3bz_amd")
    A = T.api
    s, plain = one_block_stream(int(a.mib * (1 << 20)))
    eng = T.Engine(0, lib_path=a.lib)
    out = bytearray(len(plain))
    st = A.make_deflate_state(out)
    calls, pos, t0, half_t, marks = 0, 0, time.perf_counter(), None, []
    n_calls = min(a.max_calls, (len(s) + a.chunk - 1) // a.chunk)
    while pos < len(s) and calls < n_calls:
        end = min(len(s), pos + a.chunk)
        A.decompress(A.make_octet_vector_context(s, start=pos, end=end), st, engine=eng)
        pos = end
        calls += 1
        if calls == n_calls // 2:
            half_t = time.perf_counter() - t0
        if calls in (n_calls // 4, n_calls // 2, 3 * n_calls // 4, n_calls):
            marks.append((calls, eng.session_stats(st._sess)))
    dt = time.perf_counter() - t0
    n = st.output_offset
    assert bytes(out[:n]) == plain[:n], "octets differ"
    n_dec, in_dec = eng.session_stats(st._sess)
    print("stream: %d octets in one fixed-Huffman block -> %d octets; chunks of %d octets" % (len(s), len(plain), a.chunk))
    print("calls %d in %.2f s = %.0f calls/s; first half %.2f s, second half %.2f s (linear: equal; quadratic: 3x)"
          % (calls, dt, calls / dt, half_t, dt - half_t))
    print("octets fed %d, octets handed to the engine %d: re-decode factor %.1f (a block-start resume point would be %.0f)"
          % (pos, in_dec, in_dec / max(1, pos), pos / 2 / a.chunk))
    print("engine calls %d; decoded so far %d octets; per quarter (calls, (engine calls, octets)): %s" % (n_dec, n, marks))
    eng.close()


if __name__ == "__main__":
    main()
