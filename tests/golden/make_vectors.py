#!/usr/bin/env python3
"""Generate tests/golden/deflate_vectors.json and tests/golden/test_deflated.* fixtures.

Run HERE (build container) only: reads the reference's own test DATA
  - /root/reference/deflate-test.lisp : the 37 `(deflate-test "<bits>" "<hex>" ['class])`
    known-answer forms (deflate-test.lisp:69-302; bit packing rule at :38-43,
    LSB-first within each octet)
  - /root/reference/test.deflated     : 8-byte LE length + raw deflate
    (test-chunked-input.lisp:7-25)
and writes inputs/expected outputs as data (hex strings, digests).  No reference
source text is copied: only the bit strings / hex strings (test vectors) are kept.

Nothing on the GPU box reads /root/reference; the committed fixtures are what
tests use.
"""
import hashlib
import json
import os
import re
import zlib

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def pack_bits(bits: str) -> bytes:
    """deflate-test.lisp:38-43: bit x goes to bit (x mod 8) of octet (x // 8)."""
    bits = bits.replace(" ", "")
    out = bytearray((len(bits) + 7) // 8)
    for x, c in enumerate(bits):
        if c == "1":
            out[x // 8] |= 1 << (x % 8)
    return bytes(out)


def main():
    src = open(os.path.join(REF, "deflate-test.lisp")).read().split("\n")
    vectors = []
    # forms start at column 0 with "(deflate-test"; may span several lines
    def add(start, bits, hexout, form):
        m = re.search(r"'(eof|format)\s*\)", form)
        klass = m.group(1) if m else "ok"
        bits = bits.replace(" ", "").replace("\n", "")
        data = pack_bits(bits)
        expect = bytes.fromhex(hexout.replace(" ", "").replace("\n", ""))
        vectors.append({
            "line": start,
            "class": klass,
            "nbits": len(bits),
            "input_hex": data.hex(),
            "expected_hex": expect.hex() if klass == "ok" else None,
        })

    def read_form(i):
        form = src[i]
        depth = form.count("(") - form.count(")")
        while depth > 0:
            i += 1
            form += "\n" + src[i]
            depth = form.count("(") - form.count(")")
        return form, i

    i = 0
    while i < len(src):
        if src[i].startswith("(deflate-test "):
            # direct form: (deflate-test "<bits>" "<hex>" ['class])
            start = i + 1  # 1-based line number
            form, i = read_form(i)
            strs = re.findall(r'"([^"]*)"', form)
            assert len(strs) == 2, (start, form)
            add(start, strs[0], strs[1], form)
        elif src[i].startswith("(let ("):
            # (let ((name "<bits>") ...) (deflate-test (concatenate 'string name ...) "<hex>" ['class]))
            start = i + 1
            form, i = read_form(i)
            form_nc = "\n".join(l.split(";")[0] for l in form.split("\n"))
            binds = dict(re.findall(r'\(\s*(\w+)\s+"([^"]*)"\s*\)', form_nc))
            m = re.search(r"\(concatenate 'string ([^)]*)\)\s*\"([^\"]*)\"", form_nc)
            assert m, (start, form)
            bits = "".join(binds[n] for n in m.group(1).split())
            add(start, bits, m.group(2), form_nc[m.start():])
        i += 1
    assert len(vectors) == 37, len(vectors)
    # cross-check the OK vectors against system zlib (independent implementation)
    for v in vectors:
        if v["class"] == "ok":
            d = zlib.decompressobj(-15).decompress(bytes.fromhex(v["input_hex"]))
            assert d.hex() == v["expected_hex"], v
    with open(os.path.join(HERE, "deflate_vectors.json"), "w") as f:
        json.dump({"source": "3bz deflate-test.lisp:69-302 (nayuki Simple-DEFLATE suite)",
                   "vectors": vectors}, f, indent=1)

    # test.deflated: compressed bytes are the fixture; plaintext is NOT stored
    raw = open(os.path.join(REF, "test.deflated"), "rb").read()
    n = int.from_bytes(raw[:8], "little")
    plain = zlib.decompressobj(-15).decompress(raw[8:])
    assert len(plain) == n
    with open(os.path.join(HERE, "test_deflated.bin"), "wb") as f:
        f.write(raw)
    meta = {
        "source": "3bz test.deflated (test-chunked-input.lisp:7-25): u64le length + raw deflate",
        "file_bytes": len(raw),
        "plain_len": n,
        "sha256": hashlib.sha256(plain).hexdigest(),
        "adler32": "%08x" % zlib.adler32(plain),
        "crc32": "%08x" % zlib.crc32(plain),
    }
    with open(os.path.join(HERE, "test_deflated.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("wrote", len(vectors), "vectors;", meta)

    # the exported API: the 14 symbol NAMES of package.lisp:13-27 and the lambda lists of the functions behind them
    # (api.lisp, io-common.lisp, io-mmap.lisp) — names and argument lists are the interface contract, kept as data
    pkg = open(os.path.join(REF, "package.lisp")).read()
    exports = re.findall(r"#:([^\s()]+)", pkg[pkg.index("(:export"):])
    with open(os.path.join(HERE, "package_exports.json"), "w") as f:
        json.dump({"source": "3bz package.lisp:13-27", "exports": exports}, f, indent=1)
    print("exports:", exports)


if __name__ == "__main__":
    main()
