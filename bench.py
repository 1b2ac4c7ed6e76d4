#!/usr/bin/env python3
"""bench.py — BASELINE.json's headline metric, and every other config of SURVEY §8d.

A "step" is one pass of the hot path (K0 scan [+ K0b block-start finder] -> K1 Huffman decode -> K2 LZ77
[-> K6 cross-segment resolution] -> K4/K5 checksum -> trailer verify) over one batch of synthetic input,
input and output resident in HBM when the timed region starts.

    python bench.py [--config 2] [--gpus N] [--steps K] [--warmup W]

  --config 2   (default) :zlib, 1 GiB of enwik-style text, Z_FULL_FLUSH every 16 KiB — the config the metric is quoted on
  --config 2s  config 2's ONE stream (seed 0x3B2 on every rank) decoded by ALL ranks together: rank r takes the octets between
               cut r and cut r+1 (tbz_inflate_sharded_plan), one all_gather of 8 x int64 per rank, the seam proof and the
               combined checksum (tbz_inflate_sharded_verdict) inside the timed region  ("scaling": "strong")
  --config 2b  the same text with Z_SYNC_FLUSH (history crosses every flush point)
  --config nf  the same text, no flush at all (an ordinary zlib stream: ONE segment for a marker scanner)
  --config 1   :deflate, one stored block of 65 535 octets (plumbing / call floor)
  --config 3   :gzip, 4 096 members x 256 KiB, one batch call (crc32 path), member offsets GIVEN
  --config 3u  the same file as ONE blob, offsets NOT given: tbz_inflate_gzip_members_device finds the members (K0g)
  --config 4   the FIXED batch of 8 x 128 MiB independent :zlib streams, sharded over the ranks by
               multi.assign_streams, one tbz_inflate_batch_device call per rank  ("scaling": "strong")
  --config 5   adversarial LZ77 (256 MiB, fixed-Huffman blocks, distance 1 / 32 768), no flush markers
  --config 5f  the same with an empty stored block (history kept) every MiB of output

N > 1: one process per GPU.  Under torch.distributed.run (RANK/WORLD_SIZE in the environment) this process is
a rank; otherwise the parent — which never touches the GPU — starts `python -m torch.distributed.run
--nproc-per-node N bench.py …` as a CHILD process and relays its output.  Every config but 4 gives each rank
its own workload of the same shape ("weak"); the only exchange is an all_gather of the 64-byte result records
(RCCL), enqueued behind the decode — no stream octet crosses GPUs.

`--backend gloo --lib tests/emu/libtbz_emu.so` runs the same launch / sharding logic on the CPU lane emulator
at small sizes (tests/test_bench_launch.py); it is a test mode and its numbers mean nothing.
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CONFIGS = ("1", "2", "2s", "2b", "nf", "3", "3u", "4", "5", "5f")
DEFAULT_MIB = {"1": 0, "2": 1024, "2s": 1024, "2b": 256, "nf": 64, "3": 1024, "3u": 1024, "4": 1024, "5": 256, "5f": 256}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="2", choices=CONFIGS)
    ap.add_argument("--size-mib", type=float, default=0, help="decompressed MiB of the workload (default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the byte compare of the output (status/checksum stay)")
    ap.add_argument("--no-check", action="store_true", help="experiments with deliberately wrong kernels: time only")
    ap.add_argument("--gen-workers", type=int, default=0)
    ap.add_argument("--corpus-cache", default=None, help="directory: load the workload from it if it is there, else generate "
                    "and save it (profiler runs: generate first WITHOUT the profiler, whose library initialises the GPU before "
                    "Python starts — the generators fork)")
    ap.add_argument("--gen-only", action="store_true", help="generate (and cache) the workload, then exit")
    ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
    ap.add_argument("--lib", default=None, help="engine library (tests: the CPU lane-emulator build)")
    ap.add_argument("--no-h2h", action="store_true", help="skip the host-to-host leg (pageable host buffers through tbz_inflate)")
    ap.add_argument("--no-others", action="store_true", help="the default run also times every other config (one child "
                    "process each, before this process touches the GPU) and attaches their lines as `other_configs`: skip that")
    return ap.parse_args(argv)


OTHER_CONFIGS = ("1", "3", "3u", "2b", "nf", "5", "5f")


def run_other_configs():
    """Every other BASELINE config through this same script, one child process at a time (each generates its workload,
    decodes it, verifies status / length / checksum / every octet, prints its line), so that their throughput is witnessed
    by whoever runs the default command — not only the headline's.  Runs BEFORE this process initialises the GPU."""
    out = []
    import torch  # noqa: F401  (the first import on a fresh box pages the image in — minutes; it initialises no GPU state)
    t_all = time.time()
    for c in OTHER_CONFIGS:
        t0 = time.time()
        rec = {"config": c}
        if t0 - t_all > 240:  # the headline must not wait for side lines: four minutes in all, two per config
            rec.update({"failed": True, "error": "skipped: the side lines' time budget was spent"})
            out.append(rec)
            continue
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--config", c, "--steps", "5", "--warmup", "2",
                                "--no-cpu-baseline", "--no-others"], capture_output=True, text=True, timeout=120)
            j = json.loads(r.stdout.strip().split("\n")[-1])
            rec.update({"workload": j["config"]["workload"], "value": j["value"], "unit": j["unit"],
                        "ms_per_step": j["ms_per_step"], "steps": j["steps"], "verified": True,
                        "decompressed_bytes": j["config"]["decompressed_bytes"],
                        "compressed_bytes": j["config"]["compressed_bytes"], "kernel_ms": j["roofline"]["kernel_ms"]})
        except Exception as e:  # a failing side line must not take the headline with it
            rec.update({"failed": True, "error": repr(e)[:200]})
        rec["elapsed_s"] = round(time.time() - t0, 1)
        out.append(rec)
    return out


# --------------------------------------------------------------------------------------------------
# workloads: built BEFORE anything touches the GPU or the process group (the generators fork workers)
# --------------------------------------------------------------------------------------------------
class Workload:
    """streams: list of (compressed bytes, plain bytes, expected checksum or None); one engine call decodes all"""

    def __init__(self, name, fmt, streams, scaling="weak", note=None):
        self.name, self.fmt, self.streams, self.scaling, self.note = name, fmt, streams, scaling, note
        self.C = sum(len(s[0]) for s in streams)
        self.U = sum(len(s[1]) for s in streams)


def build_workload(cfg, mib, rank, world, workers):
    from tools import corpus as K
    U = int(mib * (1 << 20))
    seed = 0x3B2 + rank
    if cfg == "2s":
        s, p, a = K.zlib_flush_stream(U, seed=0x3B2, workers=workers)   # the SAME stream on every rank
        w = Workload("config 2s: :zlib, ONE stream of %g MiB enwik-style text, Z_FULL_FLUSH every 16 KiB (%d segments), decoded by "
                     "all ranks together: contiguous parts cut at flush markers, seams proven from the parts' own results, one "
                     "all_gather of 8 x int64 per rank" % (mib, U // 16384), "zlib", [(s, p, a)], scaling="strong")
        w.total_U = U
        return w
    if cfg in ("2", "2b", "nf"):
        if cfg == "2":
            s, p, a = K.zlib_flush_stream(U, seed=seed, workers=workers)
            what = "Z_FULL_FLUSH every 16 KiB (%d dynamic-Huffman segments)" % (U // 16384)
        elif cfg == "2b":
            s, p, a = K.zlib_flush_stream(U, seed=seed, flush=zlib.Z_SYNC_FLUSH)
            what = "Z_SYNC_FLUSH every 16 KiB (history crosses the flush points)"
        else:
            p = K.enwik_like(U, seed)
            s = zlib.compress(p, 6)
            a = zlib.adler32(p)
            what = "no flush (one ordinary zlib stream)"
        return Workload("config %s: :zlib, %g MiB enwik-style text per GPU, %s, zlib level 6, seed 0x3B2+rank" %
                        (cfg, mib, what), "zlib", [(s, p, a)])
    if cfg == "1":
        s, p = K.config1_stream()
        return Workload("config 1: :deflate, one stored block of 65535 uniform octets (xorshift64* 0x3B5A0001)",
                        "deflate", [(s, p, None)])
    if cfg in ("3", "3u"):
        member = 256 << 10
        n = max(1, U // member)
        blob, offs, plains = K.gzip_members(n, member, seed=seed, workers=workers)
        ends = offs[1:] + [len(blob)]
        streams = [(blob[o:e], pl, zlib.crc32(pl)) for o, e, pl in zip(offs, ends, plains)]
        if cfg == "3u":
            w = Workload("config 3u: :gzip, ONE file of %d members x 256 KiB of the same text, member starts found on the "
                         "device (tbz_inflate_gzip_members_device), per-member parity, seed 0x3B2+rank" % n, "gzip", streams)
            w.blob = blob
            return w
        return Workload("config 3: :gzip, %d members x 256 KiB of the same text, one batch call at GIVEN offsets, per-member "
                        "parity (3bz stops after member 1), seed 0x3B2+rank" % n, "gzip", streams)
    if cfg == "4":
        M = importlib.import_module("3bz_amd.multi")
        n_streams, each = 8, max(64 << 10, U // 8)
        owner = M.assign_streams([each] * n_streams, world)   # equal sizes: LPT degenerates to round-robin
        streams = []
        for i in range(n_streams):
            if owner[i] != rank:
                continue
            s, p, a = K.zlib_flush_stream(each, seed=0x3B2 + i, workers=workers)
            streams.append((s, p, a))
        w = Workload("config 4: FIXED batch of 8 x %g MiB independent :zlib streams (config-2 shape, seeds 0x3B2+i), "
                     "owner = multi.assign_streams, one tbz_inflate_batch_device call per rank" % (each / (1 << 20)),
                     "zlib", streams, scaling="strong")
        w.total_U = each * n_streams
        return w
    if cfg in ("5", "5f"):
        s, p = K.adversarial_stream(U, sync_flush_every=(1 << 20) if cfg == "5f" else 0)
        return Workload("config %s: :zlib, %g MiB adversarial LZ77 (fixed-Huffman blocks <= 64 KiB out; phase A dist 1 len 258; "
                        "phase B dist 32768 + short-period overlaps), %s" %
                        (cfg, mib, "no flush markers: one sequential segment for a marker scanner" if cfg == "5"
                         else "an empty stored block every MiB of output, history kept"), "zlib", [(s, p, zlib.adler32(p))])
    raise SystemExit("unknown config")


def save_workload(wl, d):
    os.makedirs(d, mode=0o700, exist_ok=True)
    idx = {"name": wl.name, "fmt": wl.fmt, "scaling": wl.scaling, "note": wl.note, "total_U": getattr(wl, "total_U", None),
           "blob": getattr(wl, "blob", None) is not None, "streams": []}
    with open(os.path.join(d, "octets.bin"), "wb") as f:
        for s, p, ck in wl.streams:
            idx["streams"].append([len(s), len(p), ck])
            f.write(s)
            f.write(p)
        if idx["blob"]:
            idx["blob_len"] = len(wl.blob)
            f.write(wl.blob)
    with open(os.path.join(d, "index.json"), "w") as f:
        json.dump(idx, f)


def load_workload(d):
    idx = json.load(open(os.path.join(d, "index.json")))
    streams = []
    with open(os.path.join(d, "octets.bin"), "rb") as f:
        for ls, lp, ck in idx["streams"]:
            streams.append((f.read(ls), f.read(lp), ck))
        blob = f.read(idx["blob_len"]) if idx["blob"] else None
    wl = Workload(idx["name"], idx["fmt"], streams, scaling=idx["scaling"], note=idx["note"])
    if idx["total_U"]:
        wl.total_U = idx["total_U"]
    if blob is not None:
        wl.blob = blob
    return wl


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def k1_name(g):
    return "tbz_k1_huff_decode" if g <= 1 else "tbz_k1h_headers+tbz_k1g%d_huff_decode" % g


def k2_names(kinds):
    names = [n for b, n in ((1, "tbz_k2_lz77_dual"), (2, "tbz_k2_lz77_small"), (4, "tbz_k2_lz77")) if kinds & b]
    if kinds & 8:
        names.append("tbz_k2_lz77[plane 1]")
    return "+".join(names) or "none"


def main(argv=None):
    args = parse_args(argv)
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        # the parent never initialises the GPU: it starts the ranks as a child process and relays rank 0's line
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + \
              (sys.argv[1:] if argv is None else list(argv))
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run(cmd, env=env)
        return r.returncode

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    emu = args.backend == "gloo"

    import numpy as np
    cfg = args.config
    mib = args.size_mib or DEFAULT_MIB[cfg]
    others = None
    if (cfg == "2" and world == 1 and not args.size_mib and not args.lib and not args.corpus_cache and not args.no_cpu_baseline
            and not args.no_others and not args.gen_only and not args.no_check and args.backend == "nccl"):
        others = run_other_configs()
    ncpu = os.cpu_count() or 1
    workers = args.gen_workers or max(1, min(16, ncpu // max(1, world)))
    t0 = time.time()
    wl = None
    cache = None
    if args.corpus_cache:
        # The cache holds raw octets and a JSON index — nothing that executes when it is read — in a directory only this
        # user can enter, and its name carries a hash of the generators (tools/corpus.py) and of this file's workload
        # table: a change to either makes a new entry instead of silently reusing a corpus the source no longer describes.
        import hashlib
        h = hashlib.sha256(open(os.path.join(ROOT, "tools", "corpus.py"), "rb").read())
        h.update(open(os.path.abspath(__file__), "rb").read())
        os.makedirs(args.corpus_cache, mode=0o700, exist_ok=True)
        st = os.stat(args.corpus_cache)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise SystemExit("--corpus-cache %s: not a private directory of this user" % args.corpus_cache)
        cache = os.path.join(args.corpus_cache, "wl_%s_%g_r%dof%d_%s" % (cfg, mib, rank, world, h.hexdigest()[:16]))
        if os.path.exists(os.path.join(cache, "index.json")):
            wl = load_workload(cache)
    if wl is None:
        wl = build_workload(cfg, mib, rank, world, workers)  # forks its worker pool here, before any GPU / RCCL state
        if cache:
            save_workload(wl, cache)
    gen_s = time.time() - t0
    if args.gen_only:
        return 0

    import torch
    dist = None
    dev = "cpu"
    if not emu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X (there is no CPU fallback for the product path)")
        torch.cuda.set_device(local_rank)
        dev = "cuda:%d" % local_rank
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)

    T = importlib.import_module("3bz_amd")
    M = importlib.import_module("3bz_amd.multi")
    eng = T.Engine(0 if emu else local_rank, lib_path=args.lib)
    fmt = T.FORMATS[wl.fmt]

    # ---- device-resident input (streams packed at 16-octet aligned offsets) and output
    n = len(wl.streams)
    in_offs, in_lens, out_offs, out_caps = [], [], [], []
    ipos = opos = 0
    for s, p, _ in wl.streams:
        in_offs.append(ipos)
        in_lens.append(len(s))
        ipos += (len(s) + 15) & ~15
        out_offs.append(opos)
        out_caps.append(len(p))
        opos += (len(p) + 15) & ~15
    blob = getattr(wl, "blob", None)
    if blob is not None:   # one file: members back to back, output ranges 16-octet aligned + room for members decoded alone
        ipos, opos = len(blob), opos + (4 << 20)
    d_in = eng.malloc(ipos + 64)
    d_out = eng.malloc(opos + 64)
    shard = None
    if cfg == "2s":   # this rank's part of the one stream, resident on its device; any part's output fits U
        s0 = wl.streams[0][0]
        cuts = M.shard_plan(s0, world, lib=eng.lib)
        shard = {"cuts": cuts, "lo": cuts[rank], "hi": cuts[rank + 1], "fmt": fmt if rank == 0 else T.FORMATS["deflate"]}
        eng.h2d(d_in, s0[shard["lo"]:shard["hi"]])
    elif blob is not None:
        eng.h2d(d_in, blob)
    else:
        for (s, _, _), o in zip(wl.streams, in_offs):
            eng.h2d(d_in + o, s)

    gathered = [None]
    # the batch entry point takes plain uint64 arrays: built once, as a caller decoding batches of one shape would
    c_in_offs, c_in_lens, c_out_offs, c_out_caps = (eng.u64_array(v) for v in (in_offs, in_lens, out_offs, out_caps))

    member_offs = [None]

    verdicts = [None]

    def step_sharded():
        n_in = shard["hi"] - shard["lo"]
        if n_in == 0 and rank > 0:
            rec, res = [1, 0, 0, 0, 0, 0, 0, 0], None
        else:
            res = eng.inflate_device(d_in, n_in, d_out, out_caps[0], shard["fmt"])
            got = int(res.out_len) if res.status >= 0 else 0
            ck = 0
            if res.status in (0, 1):
                s1, s2 = eng.adler32_device(d_out, got, 1, 0)
                ck = s1 | (s2 << 16)
            rec = [int(res.status), got, int(res.in_consumed), n_in, ck, int(res.flags), 0, 0]
        recs = [rec]
        if world > 1:
            mine = torch.tensor(rec, dtype=torch.int64, device=dev)
            out = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(out, mine)
            recs = [[int(x) for x in g.cpu().tolist()] for g in out]
        verdicts[0] = M.shard_verdict(recs, wl.streams[0][0], fmt, shard["cuts"], lib=eng.lib)
        assert verdicts[0]["ok"], verdicts[0]
        return [res] if res is not None else []

    def step():
        if shard is not None:
            return step_sharded()
        if blob is not None:
            res, _io, member_offs[0] = eng.inflate_gzip_members_device(d_in, len(blob), d_out, opos, n + 8)
            assert len(res) == n, (len(res), n)
        elif n == 0:
            res = []
        elif n == 1:
            res = [eng.inflate_device(d_in, in_lens[0], d_out, out_caps[0], fmt)]
        else:
            res = eng.inflate_batch_device(d_in, c_in_offs, c_in_lens, d_out, c_out_offs, c_out_caps, fmt, raw=True)
        if world > 1:
            # X1: every rank learns every stream's 64-byte result record — an all_gather over RCCL, enqueued
            # behind the decode and checked once after the timed loop; not on the data path
            width = max(1, -(-8 // world)) if cfg == "4" else 1
            mine = torch.zeros((width, 64), dtype=torch.uint8, device=dev)
            if res:
                mine[:len(res)] = M.results_to_tensor(res, torch, dev)
            out = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(out, mine)
            gathered[0] = out
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        if not emu:
            torch.cuda.synchronize()

    keys = ("scan", "find", "huff", "lz", "resolve", "cksum", "total")
    for _ in range(args.warmup):
        res = step()
    tim = dict.fromkeys(keys, 0.0)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
        if n:
            t = eng.timings()
            for k in keys:
                tim[k] += getattr(t, k + "_ms")
    if world > 1 and shard is None:  # inside the timed region: the last exchange has to have arrived and to be clean
        for r_, t_ in enumerate(gathered[0]):
            for rec in M.tensor_to_results(t_):
                assert rec.status == 0, (r_, rec.status)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    t = eng.timings() if n else None

    # ---- correctness of what was timed: status, length, checksum against what the generator computed from the
    # plaintext (and the engine's own trailer verdict), and a byte compare of the output
    if shard is not None and not args.no_check:
        # one stream across the ranks: the verdict of the last step (seams, total, combined checksum against the stream's own
        # trailer) and this rank's octets against its part of the plaintext
        v, (s0, p0, a0) = verdicts[0], wl.streams[0]
        assert v["ok"] and v["total"] == len(p0) and v["check"] == a0 and v["in_consumed"] == len(s0), v
        if not args.no_verify:
            lo = v["offsets"][rank]
            hi = v["offsets"][rank + 1] if rank + 1 < world else v["total"]
            got = bytearray(hi - lo)
            if hi > lo:
                eng.d2h(got, d_out, hi - lo)
            assert np.array_equal(np.frombuffer(got, dtype=np.uint8), np.frombuffer(p0, dtype=np.uint8)[lo:hi]), "part differs"
    for i, ((s, p, ck), r) in enumerate(zip(wl.streams, res)):
        if args.no_check or shard is not None:
            break
        assert r.status == 0 and r.out_len == len(p), (i, r.status, r.out_len, len(p))
        if wl.fmt == "zlib":
            assert r.adler32 == ck and (r.flags & 1), (i, hex(r.adler32), hex(ck))
        elif wl.fmt == "gzip":
            assert r.crc32 == ck and (r.flags & 1), (i, hex(r.crc32), hex(ck))
    if not args.no_verify and not args.no_check and n and shard is None:
        got = bytearray(opos)
        eng.d2h(got, d_out, opos)
        g = np.frombuffer(got, dtype=np.uint8)
        for (s, p, _), o in zip(wl.streams, member_offs[0] if blob is not None else out_offs):
            assert np.array_equal(g[o:o + len(p)], np.frombuffer(p, dtype=np.uint8)), "output differs from the plaintext"

    K_ = max(1, args.steps)
    ms = {k: v / K_ for k, v in tim.items()}
    U_job = getattr(wl, "total_U", None) or wl.U * world      # strong: the fixed batch; weak: every rank's own
    value = U_job / (dt / K_) / 1e6
    decode_ms = ms["huff"] + ms["lz"] + ms["resolve"]
    alg = wl.C + wl.U                                         # decode stage: C read + U written
    path_alg = wl.C + (2 * wl.U if wl.fmt != "deflate" else wl.U)
    roof = {
        "bound": "hbm",
        "kernel": ("%s + %s%s (the decode stage of SURVEY §8d: C read + U written)" %
                   (k1_name(t.k1_gang), k2_names(t.k2_kinds), " + tbz_k6_*" if t.n_hgroups else "")) if t else None,
        "achieved": alg / (decode_ms * 1e-3) / 1e9 if decode_ms > 0 else None,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "traffic": None,
        "algorithmic_bytes": alg,
        "kernel_ms": {"tbz_k0_scan(+k0b)": ms["scan"], "tbz_k0b_find": ms["find"], "tbz_k1_huff_decode": ms["huff"],
                      "tbz_k2_lz77": ms["lz"], "tbz_k6_resolve": ms["resolve"], "tbz_k4k5_checksum": ms["cksum"],
                      "call_device_span": ms["total"]},
        "path_achieved": path_alg / (dt / K_) / 1e9,
        "path_algorithmic_bytes": path_alg,
    }
    roof["frac"] = roof["achieved"] / HBM_PEAK_GBS if roof["achieved"] else None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and world == 1:
        try:
            j = json.load(open(pmc))
            # a rocprofv3 --pmc pass cannot run inside this process: the figure is the last committed pass over the
            # SAME config, stamped with the commit and date it was taken at (null for any other config)
            if str(j.get("config", "2")) == cfg and (mib == DEFAULT_MIB[cfg]):
                roof["traffic"] = j.get("decode_stage_hbm_bytes_per_launch")
                roof["traffic_source"] = {k: j.get(k) for k in ("commit", "date", "config", "how") if k in j}
        except Exception:
            pass

    # the issue roof of the decode stage: a kernel of table lookups and bit arithmetic is bound by how many instructions
    # the SIMDs can issue long before it is bound by HBM.  The roof is CALIBRATED (tools/issue_roof.hip -> profiles/
    # issue_roof.json: what a SIMD retires per ns for the decoders' own instruction mixes at 1 .. 8 waves per SIMD); a kernel's
    # roof is that table at its own SALU : VALU ratio and occupancy (tools/pmc_issue.py), so frac <= 1 by construction.
    # Like `traffic`, the counters are the last committed pass over the SAME config (tools/pmc_sq.sh), stamped; null otherwise
    roof["issue"] = None
    pmi = os.path.join(ROOT, "profiles", "pmc_issue.json")
    if os.path.exists(pmi) and world == 1:
        try:
            j = json.load(open(pmi))
            if str(j.get("config", "2")) == cfg and (mib == DEFAULT_MIB[cfg]) and j.get("unit"):
                ks = {k: v for k, v in j["kernels"].items() if k.startswith(("tbz_k1g", "tbz_k1_huff", "tbz_k2_")) and v.get("issue_roof")}
                tot = sum(v["duration_ns"] for v in ks.values())
                roof["issue"] = {
                    "bound": "issue", "unit": j["unit"],
                    # stage figures: the kernels' own, weighted by their durations
                    "achieved": sum(v["issue_achieved"] * v["duration_ns"] for v in ks.values()) / tot,
                    "peak": sum(v["issue_roof"] * v["duration_ns"] for v in ks.values()) / tot,
                    "what": "VALU + SALU + LDS wave-instructions per SIMD per ns of the decode-stage kernels against the MEASURED rate "
                            "of a stream of the same SALU : VALU mix at the kernel's own waves per SIMD (tools/issue_roof.hip); "
                            "valu_frac: the vector pipe alone against a pure vector stream",
                    "kernels": {k: {q: v[q] for q in ("issue_achieved", "issue_roof", "issue_frac", "valu_frac", "salu_per_valu",
                                                       "waves_per_simd", "parked", "issuing", "lds_bank_conflict_share", "lanes_active")}
                                for k, v in ks.items()},
                    "source": {q: j.get(q) for q in ("commit", "date", "config", "how")},
                }
                roof["issue"]["frac"] = roof["issue"]["achieved"] / roof["issue"]["peak"]
        except Exception:
            pass

    # ---- the path a 3bz caller takes ("report host-to-host separately"; api.lisp:23-65, bench.lisp:90-120):
    # input in an ordinary (pageable) host buffer, output into a preallocated ordinary host buffer, through tbz_inflate /
    # tbz_inflate_batch.  Never `value`.  The ceiling beside it: what hipMemcpy moves on this box between PINNED host
    # memory and the device (torch's pinned tensors), each direction alone.
    h2h = None
    if rank == 0 and world == 1 and n and blob is None and shard is None and not emu and not args.no_h2h:
        h_ins = [s for s, _, _ in wl.streams]
        h_outs = [bytearray(len(p)) for _, p, _ in wl.streams]
        def h2h_step():
            if n == 1:
                return [eng.inflate(h_ins[0], fmt, h_outs[0])]
            return eng.inflate_batch(h_ins, fmt, h_outs)
        hres = h2h_step()  # warm-up: staging buffers, copy threads, first touch of the output pages
        legs = {"h2d_ms": 0.0, "decode_ms": 0.0, "d2h_ms": 0.0}
        hs = max(1, min(5, args.steps))
        t1 = time.perf_counter()
        for _ in range(hs):
            hres = h2h_step()
            ht = eng.timings()
            legs["h2d_ms"] += ht.h2d_ms / hs
            legs["decode_ms"] += ht.host_decode_ms / hs
            legs["d2h_ms"] += ht.d2h_ms / hs
        hdt = (time.perf_counter() - t1) / hs
        for (s, p, _), r, o in zip(wl.streams, hres, h_outs):
            assert r.status == 0 and r.out_len == len(p), (r.status, r.out_len)
            if not args.no_verify:
                assert np.array_equal(np.frombuffer(o, dtype=np.uint8), np.frombuffer(p, dtype=np.uint8)), "host-to-host output differs"
        # the link's rate, each direction alone, pinned memory, 256 MiB
        nb = 256 << 20
        hp = torch.empty(nb, dtype=torch.uint8).pin_memory()
        dp = torch.empty(nb, dtype=torch.uint8, device=dev)
        rates = {}
        for name, fn in (("h2d", lambda: dp.copy_(hp, non_blocking=True)), ("d2h", lambda: hp.copy_(dp, non_blocking=True))):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                fn()
            e1.record()
            torch.cuda.synchronize()
            rates[name] = 4 * nb / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del hp, dp
        slower = min(rates.values())
        h2h = {"value": wl.U / hdt / 1e6, "unit": "MB/s", "ms_per_step": hdt * 1e3, "steps": hs, **{k: round(v, 3) for k, v in legs.items()},
               "bytes_up": wl.C, "bytes_down": wl.U,
               "hipMemcpy_pinned_GBs": {k: round(v, 2) for k, v in rates.items()},
               "frac_of_slower_copy_rate": (wl.U / hdt / 1e9) / slower,
               "what": "tbz_inflate%s on pageable host buffers: chunks through two pinned buffers per direction (copy threads fill one "
                       "while the DMA engine moves the other); h2d_ms / decode_ms / d2h_ms are the call's three legs (host wall clock)"
                       % ("_batch" if n > 1 else "")}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and n:
        from oracle import oracle as O
        O.lib()
        # a bounded sample: whole streams of the workload until ~1 GiB of output or the CPU budget is spent
        budget_s, done_u, done_n, cdt = 20.0, 0, 0, 0.0
        for s, p, _ in wl.streams:
            out = bytearray(len(p))
            t1 = time.perf_counter()
            _, cnt = O.decompress_vector(s, wl.fmt, output=out)
            cdt += time.perf_counter() - t1
            assert cnt == len(p)
            done_u += len(p)
            done_n += 1
            if cdt > budget_s:
                break
        cpu = {"value": done_u / cdt / 1e6, "unit": "MB/s", "cores": 1, "kind": "port",
               "sample": "%d of the workload's %d stream(s), %.1f MiB of output, oracle/tbz_oracle.c once (C restatement "
                         "of 3bz, not 3bz; 3bz itself is single-threaded Lisp and no Lisp exists on this box); host has "
                         "%d cores" % (done_n, n, done_u / (1 << 20), ncpu)}

    if rank == 0:
        line = {
            "metric": "decompressed MB/s on 1 GiB many-block zlib, 1/2/4/8 MI355X; % HBM roofline",
            "value": value,
            "unit": "MB/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / K_ * 1e3,
            "higher_is_better": True,
            "scaling": wl.scaling,
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": wl.name, "compressed_bytes": wl.C, "decompressed_bytes": wl.U,
                       "streams_per_gpu": n, "segments": int(t.n_segments) if t else 0,
                       "lz77_groups": int(t.n_groups) if t else 0, "history_groups": int(t.n_hgroups) if t else 0,
                       "block_start_candidates": int(t.n_candidates) if t else 0,
                       "token_words": int(t.token_words) if t else 0, "scratch_bytes": int(t.scratch_bytes) if t else 0,
                       "gen_seconds": round(gen_s, 1), "backend": args.backend},
            "roofline": roof,
            "cpu_baseline": cpu,
            "host_to_host": h2h,
        }
        if others is not None:
            line["other_configs"] = others
        print(json.dumps(line), flush=True)
    eng.free(d_in)
    eng.free(d_out)
    eng.close()
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
