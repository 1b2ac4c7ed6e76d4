"""Multi-GPU: independent streams shard across ranks, one process per GPU.

A stream's decode never needs a collective (each deflate-state is self-contained,
deflate.lisp:4-62); the only exchange is an all_gather of the fixed 64-byte result records
(struct tbz_result) so every rank knows every stream's status / length / checksum — 8 x 64 B,
latency-bound, not bandwidth-bound.  `torch.distributed` is plumbing here (backend "nccl" is RCCL on
ROCm; the CPU tests use "gloo").
"""
import ctypes as C

from . import _lib


def assign_streams(compressed_sizes, world):
    """longest-processing-time-first on compressed size; round-robin when equal.
    Returns owner[i] = rank that decodes stream i."""
    order = sorted(range(len(compressed_sizes)), key=lambda i: (-compressed_sizes[i], i))
    load = [0] * world
    owner = [0] * len(compressed_sizes)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += compressed_sizes[i]
    return owner


def results_to_tensor(results, torch, device="cpu"):
    """pack a list of tbz_result into a [n, 64] uint8 tensor"""
    buf = bytearray()
    for r in results:
        buf += bytes(r)
    t = torch.frombuffer(buf if buf else bytearray(64), dtype=torch.uint8).clone()
    return t.reshape(-1, 64)[: len(results)].to(device)


def tensor_to_results(t):
    out = []
    raw = bytes(t.cpu().contiguous().numpy().tobytes())
    for i in range(len(raw) // 64):
        out.append(_lib.Result.from_buffer_copy(raw[i * 64:(i + 1) * 64]))
    return out


def exchange_results(local_results, owner, rank, world, dist, torch, device="cpu"):
    """X1: all_gather the result records.  `local_results` are this rank's streams in stream order;
    returns the records of ALL streams in stream order on every rank."""
    n = len(owner)
    per_rank = [[i for i in range(n) if owner[i] == r] for r in range(world)]
    width = max(1, max(len(p) for p in per_rank))
    mine = torch.zeros((width, 64), dtype=torch.uint8, device=device)
    if local_results:
        mine[: len(local_results)] = results_to_tensor(local_results, torch, device)
    gathered = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    out = [None] * n
    for r in range(world):
        recs = tensor_to_results(gathered[r])
        for k, i in enumerate(per_rank[r]):
            out[i] = recs[k]
    return out
