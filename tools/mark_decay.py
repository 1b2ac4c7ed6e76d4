#!/usr/bin/env python3
"""How fast does the pointer plane of an H-group empty?  (profiles/README.md, round 3.)

A group that starts at a block boundary without its 32 KiB of history decodes against SYMBOLIC history: an output octet
copied — directly or through earlier matches — from before the group's start is a pointer until K6 resolves it.  This tool
measures, on the bench corpus, how many octets of each 32 KiB window after such a start are pointers: it inflates the
stream from a block boundary twice with system zlib, once with the true history as the preset dictionary and once with
every history octet complemented, and counts the output octets that differ (exactly the octets that derive from history).

    gcc -O2 tools/zlib_blocks.c -o /tmp/zlib_blocks -lz
    python tools/mark_decay.py [--mib 64] [--starts 24] [--span-kib 512]

Measurement helper: nothing in the product or the test suites depends on it."""
import argparse
import os
import random
import subprocess
import sys
import zlib

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tools import corpus  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mib", type=int, default=64)
    ap.add_argument("--starts", type=int, default=24)
    ap.add_argument("--span-kib", type=int, default=512)
    ap.add_argument("--blocks-tool", default="/tmp/zlib_blocks")
    a = ap.parse_args()
    plain = corpus.enwik_like(a.mib << 20, seed=0x3B2)
    comp = zlib.compress(plain, 6)
    path = "/tmp/mark_decay.z"
    with open(path, "wb") as f:
        f.write(comp)
    blocks = [tuple(map(int, ln.split())) for ln in subprocess.check_output([a.blocks_tool, path]).decode().split("\n") if ln]
    arr = np.frombuffer(comp, dtype=np.uint8)

    def from_bit(bit, nbytes):  # the stream from `bit` on, moved to an octet boundary
        b0, sh = bit >> 3, bit & 7
        w = arr[b0:b0 + nbytes + 1].astype(np.uint16)
        return (w[:-1] if sh == 0 else ((w[:-1] >> sh) | (w[1:] << (8 - sh))) & 0xFF).astype(np.uint8).tobytes()

    span = a.span_kib << 10
    random.seed(1)
    rows = []
    for k in random.sample(range(50, len(blocks) - 200), a.starts):
        bit, outp, _bf, _bt = blocks[k]
        hist = plain[outp - 32768:outp]
        data = from_bit(bit, span)
        x = zlib.decompressobj(-15, zdict=hist).decompress(data, span)
        y = zlib.decompressobj(-15, zdict=bytes(v ^ 0xFF for v in hist)).decompress(data, span)
        assert x == plain[outp:outp + len(x)]
        d = np.frombuffer(x, np.uint8) != np.frombuffer(y, np.uint8)
        rows.append(d[:len(d) // 32768 * 32768].reshape(-1, 32768).sum(1))
    n = min(len(r) for r in rows)
    mean = np.mean([r[:n] for r in rows], axis=0)
    print("%d blocks of %.0f KiB output on average; pointers per 32 KiB window after a group's start (mean of %d starts):"
          % (len(blocks), (a.mib << 10) / len(blocks), a.starts))
    print(" ".join("%d" % v for v in mean))
    print("as a share of the window: " + " ".join("%.2f" % (v / 32768) for v in mean))


if __name__ == "__main__":
    main()
