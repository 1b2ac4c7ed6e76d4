"""N>1 path on CPU: world_size 2 over gloo.  Each rank decodes ITS shard of a batch of independent
zlib streams (through the lane-emulator build of the engine — there is no GPU here) and the ranks
all_gather the 64-byte result records exactly as bench.py does over RCCL."""
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
