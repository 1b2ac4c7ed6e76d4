"""GPU parity tests proper: the same cases as the CPU emulator run, through lib3bz_amd.so (C ABI)
on a real MI355X, at larger sizes, plus full-size properties.  Fails loudly if the HIP library is
missing — there is no fallback."""
import importlib
import os
import zlib

import pytest

from tests import parity_cases as P
from tools import corpus as K

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["auto", "findalways", "hostlayout", "k2single", "gang8", "gang16", "gang32", "gang64", "lane"])
def eng(request):
    """auto picks wide gangs (32/64 lanes per item) for these sizes; the other K1 flavours are forced through TBZ_K1_MODE"""
    T = importlib.import_module("3bz_amd")
    path = T._lib.default_path()
    assert os.path.exists(path), "lib3bz_amd.so missing: run __graft_entry__.build() (no CPU fallback exists)"
    if request.param == "hostlayout":  # chain walk + layout on the host even when the device could do it (K3)
        os.environ["TBZ_HOST_LAYOUT"] = "1"
    elif request.param == "findalways":  # K0b block-start finder on every stream, however small (default: large items only)
        os.environ["TBZ_FIND"] = "always"
    elif request.param == "k2single":  # one wave per group in K2 (default: front end and resolve on two waves)
        os.environ["TBZ_K2_MODE"] = "single"
    elif request.param != "auto":
        os.environ["TBZ_K1_MODE"] = request.param
    e = T.Engine(0)
    os.environ.pop("TBZ_K1_MODE", None)
    os.environ.pop("TBZ_HOST_LAYOUT", None)
    os.environ.pop("TBZ_K2_MODE", None)
    os.environ.pop("TBZ_FIND", None)
    yield e
    e.close()


@pytest.mark.parametrize("case", P.ALL_CASES, ids=lambda c: c.__name__)
def test_gpu_case(eng, case, request):
    flavour = request.node.callspec.params["eng"]
    if flavour in P.FLAVOUR_CASES:
        if case.__name__ not in P.FLAVOUR_CASES[flavour]:
            pytest.skip("this flavour runs the cases that can tell it from the default")
    elif flavour != "auto" and case not in P.K1_CASES:
        pytest.skip("does not depend on the K1 flavour")
    case(eng)


def test_gpu_reference_chunk_patterns_in_full(eng, request):
    """test.deflated in 3-octet chunks and into 3-octet buffers, to the end (the CPU suite bounds the call count)"""
    if request.node.callspec.params["eng"] not in ("auto", "findalways"):
        pytest.skip("host-side protocol over the same engine calls")
    P.case_reference_chunk_patterns(eng, max_calls=None, n_random=12)


def test_gpu_larger_sizes(eng):
    P.case_flush_streams(eng, n=8 << 20)
    P.case_noflush_streams(eng, n=6 << 20)
    P.case_block_starts_found(eng, n_blocks=160)
    P.case_chunked_resume(eng, n=600_000)
    P.case_containers_and_levels(eng, n=2_000_000)
    P.case_configs_1_3_5(eng, adv_total=8 << 20)
    P.case_device_buffers(eng, n=4 << 20)
    P.case_gzip_members(eng, n_members=40, max_len=400_000, n_false=3000)


def test_gpu_large_items_handed_to_wide_gangs():
    """a batch of many small streams and one large one: the launch's mean item picks gangs of 8, which hand the large
    stream's blocks back (SEG_WIDE) to gangs of 64 — here with a forced 20 Kbit threshold, and once with the engine's own"""
    import zlib
    T = importlib.import_module("3bz_amd")
    small = [K.enwik_like(3000 + 37 * i, 100 + i) for i in range(600)]
    big = K.enwik_like(6 << 20, 7)
    plains = small + [big]
    streams = [zlib.compress(p, 6) for p in plains]
    for forced in ("20000", None):
        if forced:
            os.environ["TBZ_WIDE_BITS"] = forced  # (read when the context is created)
        e = T.Engine(0)
        os.environ.pop("TBZ_WIDE_BITS", None)
        try:
            outs = [bytearray(len(p)) for p in plains]
            res = e.inflate_batch(streams, T.FORMATS["zlib"], outs)
            t = e.timings()
            for r, o, p in zip(res, outs, plains):
                assert r.status == 0 and bytes(o) == p and r.adler32 == zlib.adler32(p)
            if forced:
                assert t.huff_launches >= 2, t.huff_launches
        finally:
            e.close()


def test_gpu_config2_128mib_properties(eng):
    """size-independent properties at a size the oracle would take long on: adler32 of the output
    (a checksum of everything), exact length, segment count, and agreement of the engine's own
    adler with zlib's over the generated plaintext."""
    n = 128 << 20
    s, p, a = K.zlib_flush_stream(n, workers=min(16, os.cpu_count() or 1))
    d_in = eng.malloc(len(s) + 64)
    d_out = eng.malloc(n + 64)
    eng.h2d(d_in, s)
    res = eng.inflate_device(d_in, len(s), d_out, n, 1)
    got = bytearray(n)
    eng.d2h(got, d_out)
    eng.free(d_in)
    eng.free(d_out)
    assert res.status == 0 and res.out_len == n and res.segments == n // 16384
    assert res.adler32 == a == zlib.adler32(bytes(got))
    assert bytes(got) == p


def test_gpu_config4_batch_on_one_gpu(eng, request):
    """BASELINE config 4 at full size on ONE GPU: the 8 x 128 MiB batch in one tbz_inflate_batch_device call —
    status, length and adler32 of every stream (a checksum of everything, compared with zlib's over the
    generated plaintext and with the stream's own trailer), and a full octet compare of two of them."""
    import numpy as np
    if request.node.callspec.params["eng"] != "auto":
        pytest.skip("full-size batch: once, with the flavours the engine picks itself")
    each, n = 128 << 20, 8
    streams = [K.zlib_flush_stream(each, seed=0x3B2 + i, workers=min(16, os.cpu_count() or 1)) for i in range(n)]
    in_offs, out_offs, ip, op = [], [], 0, 0
    for s, p, a in streams:
        in_offs.append(ip)
        ip += (len(s) + 15) & ~15
        out_offs.append(op)
        op += each
    d_in, d_out = eng.malloc(ip + 64), eng.malloc(op + 64)
    try:
        for (s, _, _), o in zip(streams, in_offs):
            eng.h2d(d_in + o, s)
        res = eng.inflate_batch_device(d_in, in_offs, [len(s[0]) for s in streams], d_out, out_offs, [each] * n, 1)
        for i, r in enumerate(res):
            assert r.status == 0 and r.out_len == each and (r.flags & 1), (i, r.status, r.out_len)
            assert r.adler32 == streams[i][2] and r.segments == each // 16384, i
        for i in (0, n - 1):
            got = bytearray(each)
            eng.d2h(got, d_out + out_offs[i])
            assert np.array_equal(np.frombuffer(got, np.uint8), np.frombuffer(streams[i][1], np.uint8)), i
    finally:
        eng.free(d_in)
        eng.free(d_out)


def test_gpu_one_stream_across_ranks(eng):
    """SURVEY §8e row 2 on the card: the parts of ONE stream that 1/2/3/4/8 ranks would decode, decoded in turn
    by this GPU; verdict, offsets, combined checksum and octets (the two-process run is tests/test_multirank_gloo.py)"""
    from tests.test_sharded_stream import shard_cases
    shard_cases(eng, n=6 << 20)


_RCCL_WORKER = r'''
import importlib, os, sys, zlib
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from tools import corpus as K
T = importlib.import_module("3bz_amd")
M = importlib.import_module("3bz_amd.multi")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
eng = T.Engine(0)
s, p, a = K.zlib_flush_stream(4 << 20, seed=0x3B9, block=16384)
o = M.inflate_sharded(eng, s, 1, 0, 1, dist, torch, device="cuda")
assert o["sharded"] and o["status"] == 0 and o["total"] == len(p) and o["check"] == a and o["len"] == len(p), o
out = bytearray(len(p))
eng.d2h(out, o["d_out"], len(p))
eng.free(o["d_out"])
assert bytes(out) == p
bad = s[:-1] + bytes([s[-1] ^ 1])
o = M.inflate_sharded(eng, bad, 1, 0, 1, dist, torch, device="cuda")
assert not o["sharded"] and o["status"] == -11, o
recs = M.exchange_results([eng.inflate(s, 1, out)], [0], 0, 1, dist, torch, device="cuda")
assert recs[0].status == 0 and recs[0].adler32 == a
dist.destroy_process_group()
print("RCCL_OK")
'''


def test_gpu_record_exchange_over_rccl(tmp_path):
    """the two exchanges of 3bz_amd/multi.py (result records; one stream across ranks) with CUDA tensors over
    the "nccl" backend (= RCCL), one rank — the only world size a one-GPU box offers; world_size 2 runs over gloo
    in tests/test_multirank_gloo.py, N = 2/4/8 over xGMI in bench.py under the driver"""
    import subprocess
    import sys
    w = tmp_path / "rccl_worker.py"
    w.write_text(_RCCL_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(w), root], capture_output=True, text=True, env=env, timeout=600)
    assert "RCCL_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
