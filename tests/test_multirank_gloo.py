"""N>1 path on CPU: world_size 2 over gloo.  Each rank decodes ITS shard of a batch of independent
zlib streams (through the lane-emulator build of the engine — there is no GPU here) and the ranks
all_gather the 64-byte result records exactly as bench.py does over RCCL; then ONE flush-delimited
stream is decoded by both ranks together (contiguous segment ranges, seams proven from the records)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys, zlib
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from tools import corpus as K
T = importlib.import_module("3bz_amd")
M = importlib.import_module("3bz_amd.multi")
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n = 5
streams = [K.zlib_flush_stream(20000 + 3000 * i, seed=0x3B2 + i, block=4096) for i in range(n)]
owner = M.assign_streams([len(s[0]) for s in streams], world)
eng = T.Engine(0, lib_path=os.path.join(sys.argv[1], "tests", "emu", "libtbz_emu.so"))
mine = [i for i in range(n) if owner[i] == rank]
outs = [bytearray(len(streams[i][1])) for i in mine]
res = eng.inflate_batch([streams[i][0] for i in mine], 1, outs) if mine else []
for i, o, r in zip(mine, outs, res):
    assert r.status == 0 and bytes(o) == streams[i][1], i
allr = M.exchange_results(res, owner, rank, world, dist, torch)
assert len(allr) == n
for i, r in enumerate(allr):
    assert r.status == 0 and r.out_len == len(streams[i][1]) and r.adler32 == streams[i][2], (i, r.status)
# ONE flush-delimited stream across the ranks (SURVEY 8e row 2): each rank decodes its range, records are
# all_gathered, every rank reaches the same verdict; output parts are gathered here only to check them
s, p, a = K.zlib_flush_stream(150000, seed=0x3B9, block=4096)
o = M.inflate_sharded(eng, s, 1, rank, world, dist, torch)
assert o["sharded"] and o["status"] == 0 and o["total"] == len(p) and o["check"] == a, o
part = bytearray(o["len"])
if o["len"]:
    eng.d2h(part, o["d_out"], o["len"])
eng.free(o["d_out"])
assert bytes(part) == p[o["offset"]:o["offset"] + o["len"]] and 0 < o["len"] < len(p), (rank, o)
lens = [None] * world
dist.all_gather_object(lens, (o["offset"], o["len"]))
assert lens[0][0] == 0 and lens[1][0] == lens[0][1] and lens[1][0] + lens[1][1] == len(p), lens
# a stream that does not shard (sync flush: matches reach across the cut): rank 0 decodes it all, same answer
c = zlib.compressobj(6)
z = b"".join(c.compress(p[i:i + 8192]) + c.flush(zlib.Z_SYNC_FLUSH) for i in range(0, len(p), 8192)) + c.flush()
o = M.inflate_sharded(eng, z, 1, rank, world, dist, torch)
assert not o["sharded"] and o["status"] == 0 and o["total"] == len(p) and o["check"] == a, o
if rank == 0:
    whole = bytearray(o["len"])
    eng.d2h(whole, o["d_out"], o["len"])
    eng.free(o["d_out"])
    assert bytes(whole) == p
else:
    assert o["d_out"] is None
# ... and a damaged one: every rank learns the single-GPU status (adler32 mismatch, zlib.lisp:95)
bad = s[:-1] + bytes([s[-1] ^ 1])
o = M.inflate_sharded(eng, bad, 1, rank, world, dist, torch)
assert not o["sharded"] and o["status"] == -11, o
if o["d_out"]:
    eng.free(o["d_out"])
dist.barrier()
if rank == 0:
    print("MULTIRANK_OK", owner)
dist.destroy_process_group()
'''


def test_two_ranks_shard_streams_and_gather_records(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emu"), "libtbz_emu.so"],
                          stdout=subprocess.DEVNULL)
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(w), ROOT],
                       capture_output=True, text=True, env=env, timeout=600)
    assert "MULTIRANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])
