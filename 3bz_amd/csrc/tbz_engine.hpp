// tbz_engine.hpp — host orchestration behind the C ABI of include/tbz_amd.h.
//
// One call = one pipeline over a batch of streams, all device-resident:
//
//   K0 scan markers + build items (device) ──> K1 huffman decode (a gang of lanes per item; tokens + run tables)
//        ──> K3: prove that every item landed on its successor, scan the sizes, write K2's tables (device)
//            [else, on the host: follow the landing chain, repair overshoots (K1 fix-up rounds), redo
//             declined items, prefix-sum the segment sizes, form LZ77 groups]
//        ──> K2 lz77 (one wave per group) ──> K4/K5 checksum partials + combine ──> per-stream status, results
//
// The host reads 8 bytes + one index per stream after K0, a 16-byte verdict after K1 and one record per
// stream while K2 runs; its general path is control flow over one record per segment.  It never reads
// stream octets and decodes nothing.  Correctness argument for the speculative markers: item 0 of
// a stream starts at a true block boundary; an item is on the chain iff its predecessor on the
// chain ENDED A BLOCK exactly at its start, so by induction every chained item starts at a true
// block boundary and the concatenation of their tokens is exactly what a sequential decoder
// (3bz's :block-end -> :start-of-block loop, deflate.lisp:719-722) would produce.  Markers that are
// not on the chain are never used; an item that runs into a false marker mid-block is re-decoded
// from that block's start by a fix-up item that ignores markers it crosses.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "../../include/tbz_amd.h"
#include "tbz_kernels.hpp"

static_assert(sizeof(tbz_result) == 64, "tbz_result is exchanged between ranks as 64-octet records");
namespace tbz {

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
};

}  // namespace tbz

// every grow-only device pool of a context, ONCE: the members of tbz_ctx and the list all_pools() walks (destroy, trim,
// scratch accounting) are both generated from this list, so that a pool cannot be forgotten in one of them
// (d_cold / d_cold2: the gang kernels' parked canonical lists — tbz::GangCold — of the main launch / of the gangs of 64 beside it)
#define TBZ_CTX_POOLS(X) \
  X(d_str_off) X(d_str_len) X(d_tile_first) X(d_tile_counts) X(d_tile_offsets) X(d_markers) X(d_items) X(d_res) \
  X(d_tok) X(d_scratch) X(d_runs) X(d_segs) X(d_groups) X(d_order) X(d_k3_fi) X(d_k3_ni) \
  X(d_k3_oo) X(d_k3_oc) X(d_k3_sums) X(d_k3_flags) X(d_k3_gscan) X(d_k3_gne) X(d_k3_streams) X(d_k3_glob) \
  X(d_redo_items) X(d_redo_res) X(d_k0_slots) X(d_k0_fm) X(d_hdr) X(d_gck) X(d_gchunks) X(d_ck_l1) \
  X(d_tok2) X(d_runs2) X(d_ck_chunks) X(d_ck_parts) X(d_ck_streams) X(d_ck_out) X(d_crc_tab) X(d_in_stage) \
  X(d_out_stage) X(d_kb_tf) X(d_kb_slots) X(d_kb_counts) X(d_kb_offsets) X(d_kb_cands) X(d_kb_fc) X(d_kb_head) \
  X(d_markers2) X(d_kb_fm2) X(d_mark) X(d_hg) X(d_k6s) X(d_bigs) X(d_recs) X(d_kc_tf) \
  X(d_kc_slots) X(d_kc_ends) X(d_kc_link) X(d_kc_fm2) X(d_markers3) X(d_kb_keep) X(d_kb_kcounts) X(d_gz_cands) \
  X(d_gz_count) X(d_gz_tmp) X(d_wide_res) X(d_cold) X(d_cold2) X(d_small_rec)

namespace tbz {
// A few host threads that copy between a caller's (pageable) buffer and the pinned staging buffers: one thread moves
// ~10 GB/s, PCIe Gen5 x16 ~55 — the staging copy must not be what a host-to-host call waits for.  Created on the first
// large staged copy of a context, parked on a condition variable between jobs.
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
  __builtin_ia32_pause();
#else
  std::this_thread::yield();
#endif
}
struct CopySeg {
  uint8_t* dst;
  const uint8_t* src;
  size_t len;
};
struct CopyPool {
  static constexpr size_t PIECE = 256u << 10;
  std::vector<std::thread> th;
  std::mutex mu;
  std::condition_variable cv;
  std::atomic<uint64_t> gen{0};
  bool stop = false;
  // the job (written under `mu`, and only while no worker is inside work())
  const CopySeg* segs = nullptr;
  size_t nsegs = 0;
  std::atomic<size_t> next{0};
  std::atomic<size_t> left{0};   // segments not yet copied
  std::atomic<int> inside{0};    // workers between taking a job and leaving work()
  std::vector<CopySeg> tmp;
  void work() {
    for (;;) {
      const size_t i = next.fetch_add(1, std::memory_order_relaxed);
      if (i >= nsegs) return;
      memcpy(segs[i].dst, segs[i].src, segs[i].len);
      left.fetch_sub(1, std::memory_order_release);
    }
  }
  void start(int nthreads) {
    for (int t = 0; t < nthreads; t++)
      th.emplace_back([this]() {
        uint64_t seen = 0;
        for (;;) {
          // a staged copy is a run of jobs a fraction of a millisecond apart: look for the next one for a while before
          // going to sleep (a condition-variable wake-up costs as much as copying a megabyte)
          for (int spin = 0; spin < 20000 && gen.load(std::memory_order_acquire) == seen; spin++) cpu_relax();
          {
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&]() { return stop || gen.load() != seen; });
            if (stop) return;
            seen = gen.load();
            inside.fetch_add(1);  // (under the lock: the job cannot change until this worker has left work())
          }
          work();
          inside.fetch_sub(1, std::memory_order_release);
        }
      });
  }
  // copy every segment (any sizes); the calling thread works too
  void run(const CopySeg* sg, size_t ns) {
    {
      std::lock_guard<std::mutex> lk(mu);
      while (inside.load(std::memory_order_acquire)) std::this_thread::yield();  // a late waker of the job before
      segs = sg;
      nsegs = ns;
      next.store(0);
      left.store(ns);
      gen.fetch_add(1, std::memory_order_release);
    }
    cv.notify_all();
    work();
    while (left.load(std::memory_order_acquire)) cpu_relax();
  }
  void copy(void* d, const void* s_, size_t bytes) {
    if (bytes < 4 * PIECE || th.empty()) {
      memcpy(d, s_, bytes);
      return;
    }
    tmp.clear();
    for (size_t off = 0; off < bytes; off += PIECE)
      tmp.push_back(CopySeg{(uint8_t*)d + off, (const uint8_t*)s_ + off, std::min(PIECE, bytes - off)});
    run(tmp.data(), tmp.size());
  }
  ~CopyPool() {
    {
      std::lock_guard<std::mutex> lk(mu);
      stop = true;
    }
    cv.notify_all();
    for (auto& t : th) t.join();
  }
};
}  // namespace tbz

struct tbz_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;  // the gangs of 64 that take a narrow launch's few LARGE items run beside it (launch_k1)
  hipEvent_t evw[2] = {};
  hipEvent_t ev[12] = {};
  std::string err;
  tbz_timings tim{};
  uint64_t gang_rounds = 0, gang_valid = 0;  // diagnostics of the last call (K1g)
  bool k1h = true;           // env TBZ_K1H=0: no header pre-pass (the gang leaders parse every header)
  bool k2_ring2 = false;     // env TBZ_K2_RING=2: the ring kernel on two waves (round 3) instead of three
  bool k2_single = false;    // env TBZ_K2_MODE=single: one wave per group for the linear-window groups too (default: two)
  bool host_layout = false;  // env TBZ_HOST_LAYOUT=1: always chain / lay out on the host (tests force both paths)
  void* h_pin = nullptr;     // pinned host scratch for small read-backs
  size_t h_pin_cap = 0;
  // host <-> device staging of the host-buffer entry points (stage_in / stage_out below): two pinned buffers per
  // direction, an event per buffer, a pool of copying threads
  void* h_stage[4] = {};     // [0], [1]: towards the device; [2], [3]: from it
  hipEvent_t ev_stage[4] = {};
  tbz::CopyPool* copy_pool = nullptr;
  tbz::CopyPool* copy_pool2 = nullptr;  // the pipelined path drains the output while the input is still arriving
  hipStream_t stream_in = nullptr, stream_out = nullptr;  // ... on streams of their own
  bool small_fused = true;   // env TBZ_SMALL_FUSED=0: no one-launch path for small streams (tbz_small_fused)
  size_t small_max_in = 80u << 10;  // ... which takes single streams up to this many octets (env TBZ_SMALL_MAX_KIB)
  size_t pipe_min = 64u << 20;  // host inputs from this size on are decoded part by part (env TBZ_PIPE_MIN_KIB; 0: never)
  size_t pipe_part = 32u << 20; // ... of about this many input octets (env TBZ_PIPE_PART_KIB)
  int copy_threads = 8;      // env TBZ_COPY_THREADS (1: the calling thread alone)
  size_t stage_chunk = 16u << 20;  // env TBZ_STAGE_CHUNK_KIB
  int k1_mode = 0;  // 0 auto, 1 lane-per-item, 4..64 gang of that many lanes (env TBZ_K1_MODE; tests force each)
  bool sym_hist = true;  // groups that need history they do not hold run against symbolic history + K6 (env TBZ_HIST=off:
                         // they join their predecessors' group instead, one workgroup per chain, as in round 1)
  uint64_t pool_cap = 96ull << 30;  // octets of token pool + run tables one pass may hold (env TBZ_POOL_CAP_MIB): a batch
                                    // whose streams need more is decoded in several passes over consecutive streams
  // thresholds and diagnostics switches (tests force the rare paths at small sizes): read ONCE, when the context is created
  struct Tun {
    long find_enough = 2048, find_min_len = 48 << 10;  // (128 KiB until round 3: one 256 KiB gzip member — 93 KB — decoded alone
                                                        // took 2.0 ms as ONE item, 1.3 ms as five: profiles/README.md)
    int k0b_pair = -1;        // -1: by launch size
    int sub_min = 0, ovl = 0; // 0: the defaults
    long wide_bits = -1;      // -1: by gang width
    int slice = 0, h_join = 0, k6_block = 0;
    long k0c_max_block = 0;
    long k6_lds_min = 24 << 10;  // mean range length from which tbz_k6_resolve_lds is used (TBZ_K6_LDS_MIN; tests: 0)
    bool k6_two_levels = false, no_fused_adler = false, debug = false, debug2 = false;
    bool tok_full = false;    // token pools of one word per input bit for the gang kernels too (TBZ_TOK_FULL=1; default: per two)
    std::string debug_cands;
  } tun;
  int find_mode = 1;  // K0b block-start finder: 0 never, 1 for streams whose items are large (default), 2 for every stream
                      // of at least one finder tile (env TBZ_FIND=off|auto|always; tests force it at small sizes)
  // device pools (grow-only): TBZ_CTX_POOLS above
#define TBZ_X(name) tbz::DevBuf name;
  TBZ_CTX_POOLS(TBZ_X)
#undef TBZ_X
  std::vector<tbz::DevBuf> dense;  // token regions of items the one-lane kernel decoded again (SEG_REDO, probes): one word per
                                   // bit of those items only; released when the next call starts
};

namespace tbz {

// every device pool of a context (destroy, trim, accounting)
static std::vector<DevBuf*> all_pools(tbz_ctx* ctx) {
  return {
#define TBZ_X(name) &ctx->name,
      TBZ_CTX_POOLS(TBZ_X)
#undef TBZ_X
  };
}
static uint64_t dense_total(tbz_ctx* ctx) {
  uint64_t t = 0;
  for (const DevBuf& b : ctx->dense) t += b.cap;
  return t;
}
static uint64_t scratch_total(tbz_ctx* ctx) {
  uint64_t t = 0;
  for (DevBuf* b : all_pools(ctx))
    if (b != &ctx->d_in_stage && b != &ctx->d_out_stage) t += b->cap;
  return t + dense_total(ctx);
}

#define TBZ_HIP(call)                                                                       \
  do {                                                                                      \
    hipError_t e_ = (call);                                                                 \
    if (e_ != hipSuccess) {                                                                 \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                         \
      return TBZ_E_HIP;                                                                     \
    }                                                                                       \
  } while (0)

static int ensure(tbz_ctx* ctx, DevBuf& b, size_t bytes) {
  if (bytes <= b.cap && b.p) return 0;
  if (b.p) TBZ_HIP(hipFree(b.p));
  b.p = nullptr;
  b.cap = 0;
  size_t want = bytes + std::min<size_t>(bytes / 8, 32u << 20) + 256;  // (some room to grow into; a large pool is sized exactly)
  hipError_t e = hipMalloc(&b.p, want);
  if (e != hipSuccess) {
    ctx->err = std::string("hipMalloc(") + std::to_string(want) + "): " + hipGetErrorString(e);
    b.p = nullptr;
    return TBZ_E_NOMEM;
  }
  b.cap = want;
  return 0;
}
template <class T>
static int upload(tbz_ctx* ctx, DevBuf& b, const std::vector<T>& v) {
  int r = ensure(ctx, b, std::max<size_t>(v.size() * sizeof(T), 16));
  if (r) return r;
  if (!v.empty()) TBZ_HIP(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
  return 0;
}

// pinned host scratch for the small read-backs (grow-only)
static int pinned(tbz_ctx* ctx, size_t bytes) {
  bytes += 64;
  if (bytes <= ctx->h_pin_cap) return 0;
  if (ctx->h_pin) TBZ_HIP(hipHostFree(ctx->h_pin));
  ctx->h_pin = nullptr;
  ctx->h_pin_cap = 0;
  TBZ_HIP(hipHostMalloc(&ctx->h_pin, bytes * 2));
  ctx->h_pin_cap = bytes * 2;
  return 0;
}

// ---- host <-> device staging for the host-buffer entry points (the path a 3bz caller takes: a Lisp vector in, a Lisp
// vector out — api.lisp:23-65, bench.lisp:90-120).  A caller's buffer is ordinary pageable memory: handing it to
// hipMemcpyAsync makes the runtime stage it through its own pinned bounce buffer at a fraction of the link's rate.  Here
// it goes through two pinned buffers of the context: while the DMA engine moves one, the copy pool fills (or drains) the
// other, so that the link is busy all the time and the host copy costs nothing on top.
static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
static int stage_setup(tbz_ctx* ctx) {
  if (ctx->h_stage[0]) return 0;
  for (int k = 0; k < 4; k++) {
    TBZ_HIP(hipHostMalloc(&ctx->h_stage[k], ctx->stage_chunk));
    TBZ_HIP(hipEventCreate(&ctx->ev_stage[k]));
  }
  TBZ_HIP(hipStreamCreate(&ctx->stream_in));
  TBZ_HIP(hipStreamCreate(&ctx->stream_out));
  if (ctx->copy_threads > 1) {
    ctx->copy_pool = new CopyPool();
    ctx->copy_pool->start(ctx->copy_threads - 1);
    ctx->copy_pool2 = new CopyPool();
    ctx->copy_pool2->start(ctx->copy_threads - 1);
  }
  return 0;
}
// which stream, which pair of pinned buffers and which copy pool a staged transfer uses: the context's own stream for
// the plain calls (the transfer is ordered with the decode by the stream); streams of their own in the pipelined path,
// where the input of part k+1, the decode of part k and the output of part k-1 move at the same time
struct StageLane {
  hipStream_t stream;
  int b0;  // h_stage[b0], h_stage[b0 + 1]
  CopyPool* pool;
};
constexpr size_t STAGE_MIN = 1u << 20;  // smaller transfers are left to the runtime (one bounce, no pipeline to fill)
// one piece of a staged transfer: `len` octets between host memory and device offset `dev_off` of the transfer's device
// buffer.  Pieces are in ascending device order and do not overlap (the streams of a batch, 16-octet aligned).
struct StagePiece {
  uint8_t* host;
  uint64_t dev_off;
  uint64_t len;
};
// the parts of pieces[*cursor ...] that fall into device range [lo, hi): copy segments between host and a pinned buffer
// whose octet 0 corresponds to device offset lo
static void stage_segments(const std::vector<StagePiece>& ps, size_t& cursor, uint64_t lo, uint64_t hi, uint8_t* pin, bool to_pin,
                           std::vector<CopySeg>& out) {
  out.clear();
  while (cursor < ps.size() && ps[cursor].dev_off + ps[cursor].len <= lo) cursor++;
  for (size_t i = cursor; i < ps.size() && ps[i].dev_off < hi; i++) {
    const uint64_t a = std::max(lo, ps[i].dev_off), b = std::min(hi, ps[i].dev_off + ps[i].len);
    for (uint64_t x = a; x < b; x += CopyPool::PIECE) {
      const size_t n = (size_t)std::min<uint64_t>(CopyPool::PIECE, b - x);
      uint8_t* h = ps[i].host + (x - ps[i].dev_off);
      uint8_t* p = pin + (x - lo);
      out.push_back(to_pin ? CopySeg{p, h, n} : CopySeg{h, p, n});
    }
  }
}
static void stage_run(CopyPool* pool, const std::vector<CopySeg>& sg) {
  if (sg.empty()) return;
  if (pool && sg.size() > 1) {
    pool->run(sg.data(), sg.size());
  } else {
    for (const CopySeg& c : sg) memcpy(c.dst, c.src, c.len);
  }
}
// host -> device: the pieces go to d_base + dev_off, enqueued on the context's stream (what follows on the stream sees
// the octets; the caller's buffers are free when this returns).  The gaps between pieces are transferred as they are.
static int stage_in(tbz_ctx* ctx, void* d_base, const std::vector<StagePiece>& ps, const StageLane* lane = nullptr) {
  if (ps.empty()) return 0;
  const uint64_t lo0 = ps.front().dev_off, hi0 = ps.back().dev_off + ps.back().len;
  if (hi0 - lo0 < STAGE_MIN && !lane) {
    for (const StagePiece& q : ps)
      if (q.len) TBZ_HIP(hipMemcpyAsync((uint8_t*)d_base + q.dev_off, q.host, q.len, hipMemcpyHostToDevice, ctx->stream));
    return 0;
  }
  int r = stage_setup(ctx);
  if (r) return r;
  const StageLane own{ctx->stream, 0, ctx->copy_pool};
  const StageLane& L = lane ? *lane : own;
  const uint64_t ch = ctx->stage_chunk;
  std::vector<CopySeg> sg;
  size_t cursor = 0, k = 0;
  for (uint64_t off = lo0; off < hi0; off += ch, k++) {
    const uint64_t len = std::min(ch, hi0 - off);
    const int b = L.b0 + (int)(k & 1);
    if (k >= 2) TBZ_HIP(hipEventSynchronize(ctx->ev_stage[b]));  // the buffer's previous chunk has left
    stage_segments(ps, cursor, off, off + len, (uint8_t*)ctx->h_stage[b], true, sg);
    stage_run(L.pool, sg);
    TBZ_HIP(hipMemcpyAsync((uint8_t*)d_base + off, ctx->h_stage[b], len, hipMemcpyHostToDevice, L.stream));
    TBZ_HIP(hipEventRecord(ctx->ev_stage[b], L.stream));
  }
  // the staging buffers are reused by the next transfer: nothing may still be reading them then (and with a lane of
  // its own: the octets are on the device when this returns)
  for (int b = 0; b < 2; b++)
    if (k > (size_t)b) TBZ_HIP(hipEventSynchronize(ctx->ev_stage[L.b0 + b]));
  return 0;
}
// device -> host after everything enqueued on the context's stream; complete when this returns
static int stage_out(tbz_ctx* ctx, const void* d_base, const std::vector<StagePiece>& ps, const StageLane* lane = nullptr) {
  if (ps.empty()) return 0;
  const uint64_t lo0 = ps.front().dev_off, hi0 = ps.back().dev_off + ps.back().len;
  if (hi0 - lo0 < STAGE_MIN && !lane) {
    for (const StagePiece& q : ps)
      if (q.len) TBZ_HIP(hipMemcpyAsync(q.host, (const uint8_t*)d_base + q.dev_off, q.len, hipMemcpyDeviceToHost, ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    return 0;
  }
  int r = stage_setup(ctx);
  if (r) return r;
  const StageLane own{ctx->stream, 2, ctx->copy_pool};
  const StageLane& L = lane ? *lane : own;
  const uint64_t ch = ctx->stage_chunk;
  const size_t nch = (size_t)((hi0 - lo0 + ch - 1) / ch);
  std::vector<CopySeg> sg;
  size_t cursor = 0;
  for (size_t k = 0; k <= nch; k++) {
    if (k < nch) {  // chunk k on its way (its buffer was drained two chunks ago)
      const int b = L.b0 + (int)(k & 1);
      const uint64_t off = lo0 + k * ch;
      TBZ_HIP(hipMemcpyAsync(ctx->h_stage[b], (const uint8_t*)d_base + off, std::min(ch, hi0 - off), hipMemcpyDeviceToHost, L.stream));
      TBZ_HIP(hipEventRecord(ctx->ev_stage[b], L.stream));
    }
    if (k >= 1) {  // chunk k - 1 has arrived: to the callers' buffers while chunk k moves
      const int b = L.b0 + (int)((k - 1) & 1);
      const uint64_t off = lo0 + (k - 1) * ch;
      TBZ_HIP(hipEventSynchronize(ctx->ev_stage[b]));
      stage_segments(ps, cursor, off, std::min(off + ch, hi0), (uint8_t*)ctx->h_stage[b], false, sg);
      stage_run(L.pool, sg);
    }
  }
  return 0;
}
static int stage_in(tbz_ctx* ctx, void* d_dst, const void* h_src, size_t n) {
  if (!n) return 0;
  return stage_in(ctx, d_dst, std::vector<StagePiece>{StagePiece{(uint8_t*)h_src, 0, n}});
}
static int stage_out(tbz_ctx* ctx, void* h_dst, const void* d_src, size_t n) {
  if (!n) return 0;
  return stage_out(ctx, d_src, std::vector<StagePiece>{StagePiece{(uint8_t*)h_dst, 0, n}});
}

// ---- CRC constant tables (host side of K5) ---------------------------------------------------
static uint32_t h_mulmod(uint32_t a, uint32_t b) {
  uint32_t p = 0;
  for (int i = 0; i < 32; i++) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b & 1) ? ((b >> 1) ^ 0xedb88320u) : (b >> 1);
  }
  return p;
}
static void build_crc_tables(std::vector<uint32_t>& t) {
  t.assign(CRC_WORDS, 0);
  for (uint32_t n = 0; n < 256; n++) {  // checksums.lisp:177-193
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
    t[CRC_T + n] = c;
  }
  // x^(2^k)
  uint32_t p = 0x40000000u;  // x^1
  for (int k = 0; k < 64; k++) {
    t[CRC_X2N + k] = p;
    p = h_mulmod(p, p);
  }
  auto pow_x8 = [&](uint64_t nbytes) {
    uint32_t r = 0x80000000u;
    uint32_t k = 3;
    while (nbytes) {
      if (nbytes & 1) r = h_mulmod(t[CRC_X2N + (k & 63)], r);
      nbytes >>= 1;
      k++;
    }
    return r;
  };
  // advance-by-256-octets operator applied to each state byte
  uint32_t z256 = pow_x8(256);
  for (uint32_t b = 0; b < 4; b++)
    for (uint32_t v = 0; v < 256; v++) t[CRC_K + b * 256 + v] = h_mulmod(z256, v << (8 * b));
  for (uint32_t j = 0; j < 64; j++) t[CRC_LANE + j] = pow_x8(256 - 4 * j);
}

struct StreamPlan {
  uint64_t in_off, in_len, out_off, out_cap;
  uint32_t first_item, n_items;   // round-0 items of this stream: head + one per marker
  uint32_t first_marker;          // index of the stream's first marker in the global list
  // chain walk state
  uint32_t cur_item;              // next item to consume (index into round-0 items) or fix-up slot
  bool pending_fixup = false, done = false, next_continues = false;
  uint64_t fix_start_bit = 0;
  int32_t status = 0;             // final per-stream status once done
  uint64_t total_out = 0;
  uint64_t in_end_bit = 0;
  uint64_t boundary_bit = 0;  // last flush boundary the chain is known to have landed on (resume / shard seam)
  uint64_t boundary_out = 0;  // output octets produced before it
  bool stored_cut = false;    // the input ran out inside a stored block's payload
  bool blk_known = false;     // not finished: where a resumed decode would start (CoreOpts)
  uint64_t blk_bit = 0, blk_out = 0;
  uint64_t hdr_bit = 0;       // ... exactly: the header of the block the input ran out in (K1 reports it with every underrun)
  uint32_t trailer0 = 0, trailer1 = 0, trailer_have = 0;
  bool saw_final = false;
  uint32_t seg_first = 0, seg_count = 0;
};

struct SegHost {
  Seg seg;
  Item item;  // the K1 item that produced it (re-decoded to locate a distance error exactly)
  uint32_t stream;
  uint32_t deficit;
  bool continues;  // must share the window with the previous segment (fix-up continuation)
};

static int record(tbz_ctx* ctx, int i) {
  TBZ_HIP(hipEventRecord(ctx->ev[i], ctx->stream));
  return 0;
}
static float elapsed(tbz_ctx* ctx, int a, int b) {
  float ms = 0;
  if (hipEventElapsedTime(&ms, ctx->ev[a], ctx->ev[b]) != hipSuccess) return 0;
  return ms;
}

// checksum of out[out_off .. +len) per stream on the device; kind 1 = adler32, 2 = crc32
static int run_checksums(tbz_ctx* ctx, int kind, const void* d_out, const std::vector<uint64_t>& offs,
                         const std::vector<uint64_t>& lens, const std::vector<uint32_t>& init,
                         std::vector<uint32_t>& out) {
  size_t n = offs.size();
  std::vector<CkChunk> chunks;
  std::vector<CkStream> cs(n);
  for (size_t s = 0; s < n; s++) {
    cs[s].first = (uint32_t)chunks.size();
    for (uint64_t o = 0; o < lens[s]; o += CK_CHUNK) {
      CkChunk c;
      c.out_abs = offs[s] + o;
      c.len = (uint32_t)std::min<uint64_t>(CK_CHUNK, lens[s] - o);
      c.stream = (uint32_t)s;
      chunks.push_back(c);
    }
    cs[s].count = (uint32_t)chunks.size() - cs[s].first;
    cs[s].init0 = init[s];
    cs[s].pad = 0;
  }
  int r;
  if ((r = upload(ctx, ctx->d_ck_chunks, chunks))) return r;
  if ((r = upload(ctx, ctx->d_ck_streams, cs))) return r;
  if ((r = ensure(ctx, ctx->d_ck_parts, std::max<size_t>(chunks.size(), 1) * sizeof(CkPartial)))) return r;
  if ((r = ensure(ctx, ctx->d_ck_out, std::max<size_t>(n, 1) * sizeof(uint32_t)))) return r;
  if (kind == 1) {
    if (!chunks.empty()) {
      K4Params p{(const u8*)d_out, (const CkChunk*)ctx->d_ck_chunks.p, (CkPartial*)ctx->d_ck_parts.p,
                 (u32)chunks.size()};
      TBZ_LAUNCH(tbz_k4_adler_partial, chunks.size(), ctx->stream, p);
    }
    K4cParams c{(const CkChunk*)ctx->d_ck_chunks.p, (const CkPartial*)ctx->d_ck_parts.p,
                (const CkStream*)ctx->d_ck_streams.p, (u32*)ctx->d_ck_out.p, (u32)n};
    TBZ_LAUNCH(tbz_k4_adler_combine, n, ctx->stream, c);
  } else {
    if (!chunks.empty()) {
      K5Params p{(const u8*)d_out, (const CkChunk*)ctx->d_ck_chunks.p, (CkPartial*)ctx->d_ck_parts.p,
                 (const u32*)ctx->d_crc_tab.p, (u32)chunks.size()};
      TBZ_LAUNCH(tbz_k5_crc_partial, chunks.size(), ctx->stream, p);
    }
    K5cParams c{(const CkChunk*)ctx->d_ck_chunks.p, (const CkPartial*)ctx->d_ck_parts.p,
                (const CkStream*)ctx->d_ck_streams.p, (const u32*)ctx->d_crc_tab.p, (u32*)ctx->d_ck_out.p,
                (u32)n};
    TBZ_LAUNCH(tbz_k5_crc_combine, n, ctx->stream, c);
  }
  TBZ_HIP(hipGetLastError());
  out.resize(n);
  TBZ_HIP(hipMemcpyAsync(out.data(), ctx->d_ck_out.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// adler32 from the per-group partials the two-wave K2 leaves behind (it takes them from its LDS window while the
// window is flushed): level 1 folds ranges of 256 groups, level 2 is the ordinary ordered combine per stream
static int run_adler_groups(tbz_ctx* ctx, const std::vector<uint32_t>& first_group, const std::vector<uint32_t>& n_groups,
                            const std::vector<uint32_t>& init, std::vector<uint32_t>& out) {
  size_t n = first_group.size();
  std::vector<CkStream> l1, l2(n);
  for (size_t s = 0; s < n; s++) {
    l2[s].first = (uint32_t)l1.size();
    for (uint32_t j = 0; j < n_groups[s]; j += 256) {
      CkStream r;
      r.first = first_group[s] + j;
      r.count = std::min<uint32_t>(256, n_groups[s] - j);
      r.init0 = 0;
      r.pad = 0;
      l1.push_back(r);
    }
    l2[s].count = (uint32_t)l1.size() - l2[s].first;
    l2[s].init0 = init[s];
    l2[s].pad = 0;
  }
  int r;
  if ((r = upload(ctx, ctx->d_ck_l1, l1))) return r;
  if ((r = upload(ctx, ctx->d_ck_streams, l2))) return r;
  if ((r = ensure(ctx, ctx->d_ck_chunks, std::max<size_t>(l1.size(), 1) * sizeof(CkChunk)))) return r;
  if ((r = ensure(ctx, ctx->d_ck_parts, std::max<size_t>(l1.size(), 1) * sizeof(CkPartial)))) return r;
  if ((r = ensure(ctx, ctx->d_ck_out, std::max<size_t>(n, 1) * sizeof(uint32_t)))) return r;
  if (!l1.empty()) {
    K4lParams p{(const CkChunk*)ctx->d_gchunks.p, (const CkPartial*)ctx->d_gck.p, (const CkStream*)ctx->d_ck_l1.p,
                (CkChunk*)ctx->d_ck_chunks.p, (CkPartial*)ctx->d_ck_parts.p, (u32)l1.size()};
    TBZ_LAUNCH(tbz_k4_adler_combine_l1, l1.size(), ctx->stream, p);
  }
  K4cParams c{(const CkChunk*)ctx->d_ck_chunks.p, (const CkPartial*)ctx->d_ck_parts.p,
              (const CkStream*)ctx->d_ck_streams.p, (u32*)ctx->d_ck_out.p, (u32)n};
  TBZ_LAUNCH(tbz_k4_adler_combine, n, ctx->stream, c);
  TBZ_HIP(hipGetLastError());
  out.resize(n);
  TBZ_HIP(hipMemcpyAsync(out.data(), ctx->d_ck_out.p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  return 0;
}

// what a resumable session (tbz_session_*) and the allocate-once entry point ask of the pipeline beyond a plain call;
// only for calls of ONE stream
struct CoreOpts {
  // ---- in
  uint32_t start_bit_off = 0;  // the stream's first block header sits at this bit of its first octet (a continuation:
                               // raw blocks entered at a block boundary, deflate.lisp:518-528 needs no other state)
  uint64_t resume_tok_bit = 0; // != 0: a continuation INSIDE a block — the block's header is parsed at start_bit_off, then the
                               // token loop is entered at this bit (from the stream's first octet; beyond the header)
  uint64_t hist_len = 0;       // octets of earlier output that precede this call's output in the buffer (at
                               // out_off - hist_len ...): matches may reach into them (deflate.lisp:343-352, the window)
  bool prefix_on_error = false;  // a stream that fails is still laid out and decoded up to the failing token
                                 // (tbz_result.out_total = that many octets), as a front-to-back decoder would have
  // the output buffer is obtained once the size is known: total octets -> device pointer of a buffer in which the
  // stream's output starts at octet `out_offs[0]` (and that holds total + 64 octets from there)
  std::function<void*(uint64_t total)> alloc;
  // ---- out (stream not finished)
  bool blk_known = false;
  uint64_t blk_bit = 0;        // start of the block in which the input ended (or the position where it ended, when that is
                               // where a block starts), in bits from the stream's first octet
  uint64_t blk_out = 0;        // octets produced before that block
  uint64_t hdr_bit = 0;        // the header of the block in which the input ended, exactly (also after blocks decoded through)
  uint64_t tok_bit = 0;        // start of the token (or header field) the input ended in: everything before it is decoded
  uint64_t end_bit = 0;        // finished: bit position after the final block (octet-aligned, before any trailer the
                               // engine did not parse because the format was raw deflate)
  int32_t first_error = 0;     // prefix_on_error: the status a plain call would have reported (0 if none)
};

// ---- one small stream in ONE launch (tbz_small_fused, tbz_kernels.hpp): the kernel decides the clean case only —
// finished, trailer present and matching, everything fits — and says "fall back" for anything else, so every other
// outcome is produced by the general path below.  One launch, one 128-octet read-back: the call floor falls from ~200 us
// (a dozen launches, three read-backs) to ~35.
static int small_fused_try(tbz_ctx* ctx, int format, const void* d_in, uint64_t in_len, void* d_out, uint64_t out_cap,
                           tbz_result* R, bool* handled) {
  *handled = false;
  int r;
  const uint64_t tok_words = (in_len * 8 >> 1) + 256, run_slots = (in_len * 8 >> RUN_SHIFT) + 4;
  if ((r = ensure(ctx, ctx->d_tok, tok_words * 2))) return r;
  if ((r = ensure(ctx, ctx->d_runs, run_slots * sizeof(RunRec)))) return r;
  if ((r = ensure(ctx, ctx->d_small_rec, sizeof(SmallRec)))) return r;
  if ((r = pinned(ctx, sizeof(SmallRec)))) return r;
  SmallParams sp{(const u8*)d_in, in_len, (u8*)d_out, out_cap, (u32)format, (u32)std::max<long>(0, ctx->tun.find_min_len),
                 (u16*)ctx->d_tok.p, (RunRec*)ctx->d_runs.p, (const u32*)ctx->d_crc_tab.p, (SmallRec*)ctx->d_small_rec.p};
  TBZ_LAUNCH(tbz_small_fused, 1, ctx->stream, sp);
  SmallRec* h = (SmallRec*)ctx->h_pin;
  TBZ_HIP(hipMemcpyAsync(h, ctx->d_small_rec.p, sizeof(SmallRec), hipMemcpyDeviceToHost, ctx->stream));
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->tun.debug) fprintf(stderr, "tbz: small fused: state %u why %u\n", h->state, h->why);
  if (h->state != SMALL_DONE) return 0;
  const SegResult& q = h->seg;
  memset(R, 0, sizeof *R);
  R->status = TBZ_FINISHED;
  R->segments = (q.tok_words || q.out_bytes) ? 1u : 0u;
  R->out_len = R->out_total = R->boundary_out = q.out_bytes;
  R->in_consumed = q.end_bit / 8;
  R->flags = 2u | (format != TBZ_FORMAT_DEFLATE ? 1u : 0u);
  if (format != TBZ_FORMAT_DEFLATE) {
    R->trailer_check = q.trailer0;
    R->trailer_isize = format == TBZ_FORMAT_GZIP ? q.trailer1 : 0u;
    (format == TBZ_FORMAT_ZLIB ? R->adler32 : R->crc32) = h->check;
  }
  ctx->tim.huff_launches = 1;
  ctx->tim.k1_gang = 64;
  ctx->tim.k2_kinds = q.tok_words ? 4u : 0u;
  ctx->tim.token_words = q.tok_words;
  ctx->tim.n_segments = ctx->tim.n_groups = R->segments;
  ctx->tim.scratch_bytes = scratch_total(ctx);
  ctx->gang_rounds = q.reserved >> 32;
  ctx->gang_valid = q.reserved & 0xffffffffu;
  *handled = true;
  return 0;
}

// An H-group is a group decoded in parallel with the groups before it: its matches may reach into output
// that does not exist yet.  hgs = the H-groups of the call in output order; the launch list holds n_small groups for the
// linear kernels (launched by the caller), then n_big plain large groups, then the n_h H-groups.
struct HGroupSpan { uint64_t start, end, floor; uint32_t stream; };
// The ring launch for the plain large groups and BOTH planes of the H-groups (octets -> the output, pointer high octets ->
// the mark plane over [mark_lo, mark_hi) of it), then K6: the work lists (blocks and superblocks of consecutive H-groups),
// the chains over the groups' tails, and the parallel resolve of everything else.  Events 10 / 11 bracket K6.
static int hgroups_launch(tbz_ctx* ctx, const std::vector<HGroupSpan>& hgs, K2Params k2, size_t n_small, size_t n_big,
                          size_t n_h, void* d_out, uint64_t mark_lo, uint64_t mark_hi) {
  int r;
  // ---- K6 work lists (see tbz_kernels.hpp): blocks = runs of consecutive H-groups of one stream, at most `bmax` long
  size_t max_per_stream = 0;
  for (size_t i = 0, j; i < hgs.size(); i = j) {
    for (j = i; j < hgs.size() && hgs[j].stream == hgs[i].stream; j++) {}
    max_per_stream = std::max(max_per_stream, j - i);
  }
  // three levels: groups -> blocks of at most bmax groups -> superblocks of at most bmax blocks, bmax = N^(1/3): the
  // dependent chain is 3 * bmax steps (two levels: 2 * sqrt(N))
  size_t bmax = 1;
  while (bmax * bmax * bmax < max_per_stream) bmax++;
  if (ctx->tun.k6_block) bmax = std::max(1, ctx->tun.k6_block);
  const bool three = !ctx->tun.k6_two_levels;
  if (!three) {
    bmax = 1;
    while (bmax * bmax < max_per_stream) bmax++;
    if (ctx->tun.k6_block) bmax = std::max(1, ctx->tun.k6_block);
  }
  std::vector<K6Range> ranges;
  std::vector<K6List> lists;
  // section 1: tails of the groups of multi-group blocks (tbz_k6_chain_sym, relative to the block afterwards);
  // section 1b: the last 32 KiB of the blocks of multi-block superblocks (tbz_k6_chain_sym again, relative to the
  // superblock afterwards); section 2: one range per superblock, one list per stream (tbz_k6_chain: final);
  // sections 3s / 3a / 3b: tbz_k6_resolve, in this order (what each refers to is final by then)
  std::vector<K6Range> r1, r1b, r2, r3s, r3a, r3b;
  std::vector<K6List> l1, l1b, l2;
  struct Blk { size_t j, e; uint64_t B0, E, last_lo, fl; };
  for (size_t i = 0, jj; i < hgs.size(); i = jj) {
    // ---- this stream's blocks
    std::vector<Blk> blks;
    for (jj = i; jj < hgs.size() && hgs[jj].stream == hgs[i].stream;) {
      size_t e = jj + 1;  // block [jj, e)
      while (e < hgs.size() && e - jj < bmax && hgs[e].stream == hgs[jj].stream && hgs[e].start == hgs[e - 1].end) e++;
      const uint64_t B0 = hgs[jj].start, E = hgs[e - 1].end, fl = hgs[jj].floor;
      const uint64_t last_lo = E - B0 > K6_W ? E - K6_W : B0;
      blks.push_back(Blk{jj, e, B0, E, last_lo, fl});
      if (e - jj > 1) {
        l1.push_back(K6List{(u32)r1.size(), (u32)(e - jj)});
        for (size_t k = jj; k < e; k++) {
          const uint64_t t_lo = hgs[k].end - hgs[k].start > K6_W ? hgs[k].end - K6_W : hgs[k].start;
          r1.push_back(K6Range{hgs[k].start, t_lo, hgs[k].end, fl});
          if (t_lo < std::min(hgs[k].end, last_lo)) r3a.push_back(K6Range{B0, t_lo, std::min(hgs[k].end, last_lo), fl});
        }
      }
      for (size_t k = jj; k < e; k++)  // a group's octets before its tail, in sub-ranges of at most 64 KiB
        if (hgs[k].end - hgs[k].start > K6_W)
          for (uint64_t x = hgs[k].start; x < hgs[k].end - K6_W; x += 65536)
            r3b.push_back(K6Range{hgs[k].start, x, std::min(x + 65536, hgs[k].end - K6_W), fl});
      jj = e;
    }
    // ---- its superblocks: runs of adjacent blocks
    // (a superblock that begins 32 KiB or more after the one before it ended sees nothing of it: everything in
    // between came out of groups that needed no history and is final in memory — such superblocks open a list of
    // their own, i.e. a workgroup of their own: config 5f alternates H-groups and plain ones, and ONE workgroup
    // walked its thousand superblocks for 2.4 ms)
    K6List ls{(u32)r2.size(), 0};
    uint64_t prev_SE = 0;
    for (size_t q = 0, qe; q < blks.size(); q = qe) {
      qe = q + 1;
      while (three && qe < blks.size() && qe - q < bmax && blks[qe].B0 == blks[qe - 1].E) qe++;
      const uint64_t S0 = blks[q].B0, SE = blks[qe - 1].E, fl = blks[q].fl;
      if (ls.count && S0 - prev_SE >= K6_W) {
        l2.push_back(ls);
        ls = K6List{(u32)r2.size(), 0};
      }
      prev_SE = SE;
      const uint64_t s_last_lo = SE - S0 > K6_W ? SE - K6_W : S0;
      if (qe - q > 1) {
        l1b.push_back(K6List{(u32)r1b.size(), (u32)(qe - q)});
        for (size_t k = q; k < qe; k++) {
          r1b.push_back(K6Range{blks[k].B0, blks[k].last_lo, blks[k].E, fl});
          if (blks[k].last_lo < std::min(blks[k].E, s_last_lo)) r3s.push_back(K6Range{S0, blks[k].last_lo, std::min(blks[k].E, s_last_lo), fl});
        }
      }
      r2.push_back(K6Range{S0, s_last_lo, SE, fl});
      ls.count++;
    }
    l2.push_back(ls);
  }
  const size_t o1b = r1.size(), o2 = o1b + r1b.size(), o3s = o2 + r2.size(), o3a = o3s + r3s.size(), o3b = o3a + r3a.size();
  for (auto& l : l1b) l.first += (u32)o1b;
  for (auto& l : l2) l.first += (u32)o2;
  ranges = r1;
  ranges.insert(ranges.end(), r1b.begin(), r1b.end());
  ranges.insert(ranges.end(), r2.begin(), r2.end());
  ranges.insert(ranges.end(), r3s.begin(), r3s.end());
  ranges.insert(ranges.end(), r3a.begin(), r3a.end());
  ranges.insert(ranges.end(), r3b.begin(), r3b.end());
  lists = l1;
  lists.insert(lists.end(), l1b.begin(), l1b.end());
  lists.insert(lists.end(), l2.begin(), l2.end());
  // the mark plane covers [mark_lo, mark_hi) of the output, at the same alignment (mod 16) as the output itself
  const uint32_t m0 = (uint32_t)(((uintptr_t)d_out + mark_lo) & 15);
  if ((r = ensure(ctx, ctx->d_mark, (mark_hi - mark_lo) + 64))) return r;
  if ((r = upload(ctx, ctx->d_hg, ranges))) return r;
  if ((r = upload(ctx, ctx->d_k6s, lists))) return r;
  // ONE launch of the ring kernel: the plain large groups, and both planes of every H-group (octets -> out, pointer
  // high octets -> the mark plane) as neighbouring workgroups, which read the same tokens
  k2.order = (const u32*)ctx->d_order.p + n_small;
  k2.n_groups = (u32)(n_big + n_h);
  k2.win_bytes = 0;
  k2.mixed = 1;
  k2.n_plain = (u32)n_big;
  k2.mark_base = (u8*)ctx->d_mark.p + m0;
  k2.mark_bias = mark_lo;
  ctx->tim.k2_kinds |= 4u | 8u;
  ctx->tim.n_hgroups = n_h;
  {
    const size_t nwg = n_big + 2 * n_h;
    if (ctx->k2_single) TBZ_LAUNCH(tbz_k2_lz77, nwg, ctx->stream, k2);
    else if (ctx->k2_ring2) TBZ_LAUNCH_WG(tbz_k2_lz77_ring2, nwg, 128, ctx->stream, k2);
    else TBZ_LAUNCH_WG(tbz_k2_lz77_ring3, nwg, 192, ctx->stream, k2);
  }
  TBZ_HIP(hipEventRecord(ctx->ev[10], ctx->stream));
  const K6Range* dr = (const K6Range*)ctx->d_hg.p;
  const K6List* dl = (const K6List*)ctx->d_k6s.p;
  K6Params k6{(u8*)d_out, (u8*)ctx->d_mark.p + m0, mark_lo, dr, dl, (u32)ranges.size(), 0, 0};
  if (!l1.empty()) {
    k6.n_lists = (u32)l1.size();
    TBZ_LAUNCH_WG(tbz_k6_chain_sym, l1.size(), K6_THREADS, ctx->stream, k6);
  }
  if (!l1b.empty()) {
    k6.lists = dl + l1.size();
    k6.n_lists = (u32)l1b.size();
    TBZ_LAUNCH_WG(tbz_k6_chain_sym, l1b.size(), K6_THREADS, ctx->stream, k6);
  }
  k6.lists = dl + l1.size() + l1b.size();
  k6.n_lists = (u32)l2.size();
  TBZ_LAUNCH_WG(tbz_k6_chain, l2.size(), K6_THREADS, ctx->stream, k6);
  auto resolve = [&](size_t first, const std::vector<K6Range>& rs, uint64_t maxlen) {
    const size_t count = rs.size();
    if (!count) return;
    K6Params q = k6;
    q.ranges = dr + first;
    q.n_ranges = (u32)count;
    uint64_t total = 0;
    for (const K6Range& g : rs) total += g.hi > g.lo ? g.hi - g.lo : 0;
    if (total / count >= (uint64_t)ctx->tun.k6_lds_min) {  // long ranges: the pointers' 32 KiB staged in LDS (1 GiB no-flush stream: K6 1.89 -> 1.44 ms)
      q.pieces = (u32)((maxlen + 15 + K6R_PIECE - 1) / K6R_PIECE);
      TBZ_LAUNCH_WG(tbz_k6_resolve_lds, count * (size_t)q.pieces, K6R_THREADS, ctx->stream, q);
    } else {
      q.pieces = (u32)((maxlen + 15 + K6_PIECE - 1) / K6_PIECE);
      TBZ_LAUNCH(tbz_k6_resolve, count * (size_t)q.pieces, ctx->stream, q);
    }
  };
  resolve(o3s, r3s, K6_W);
  resolve(o3a, r3a, K6_W);
  resolve(o3b, r3b, 65536);
  TBZ_HIP(hipEventRecord(ctx->ev[11], ctx->stream));
  return 0;
}

// the whole pipeline on device-resident buffers
static int inflate_core(tbz_ctx* ctx, int format, size_t n, const void* d_in, const uint64_t* in_offs,
                        const uint64_t* in_lens, void* d_out, const uint64_t* out_offs, const uint64_t* out_caps,
                        tbz_result* results, bool size_only, CoreOpts* opt = nullptr) {
  if (!ctx || !results || (n && (!in_offs || !in_lens))) return TBZ_E_ARG;
  if (format < 0 || format > 2) return TBZ_E_ARG;
  if (!size_only && n && (!out_offs || !out_caps)) return TBZ_E_ARG;
  if (opt && n != 1) return TBZ_E_ARG;
  const uint64_t hist_len = opt ? opt->hist_len : 0;
  if (hist_len && !ctx->sym_hist) return TBZ_E_UNSUPPORTED;  // (TBZ_HIST=off: no way to reach octets of an earlier call)
  const uint32_t bit_off = opt ? opt->start_bit_off : 0;
  const uint64_t resume_abs = (opt && opt->resume_tok_bit && n) ? in_offs[0] * 8 + opt->resume_tok_bit : 0;
  TBZ_HIP(hipSetDevice(ctx->device));
  ctx->tim = tbz_timings{};
  ctx->gang_rounds = ctx->gang_valid = 0;
  for (size_t i = 0; i < n; i++) memset(&results[i], 0, sizeof(tbz_result));
  if (n == 0) return 0;
  if (n > 0x7fffffffu) return TBZ_E_ARG;
  int r;
  // one small stream, an ordinary one-shot call, the engine's own flavours: the one-launch path first
  if (n == 1 && !opt && !size_only && ctx->small_fused && in_lens[0] && in_lens[0] <= ctx->small_max_in && d_in && (d_out || !out_caps[0]) &&
      !ctx->k1_mode && ctx->find_mode == 1 && !ctx->host_layout && !ctx->k2_single && ctx->sym_hist && !ctx->tun.tok_full) {
    bool handled = false;
    if ((r = small_fused_try(ctx, format, (const uint8_t*)d_in + in_offs[0], in_lens[0], d_out ? (uint8_t*)d_out + out_offs[0] : nullptr,
                             out_caps[0], &results[0], &handled)))
      return r;
    if (handled) return 0;
    ctx->tim = tbz_timings{};
  }

  // ---------------------------------------------------------------- stream table + tiles
  std::vector<StreamPlan> sp(n);
  std::vector<uint64_t> h_off(n), h_len(n);
  std::vector<uint32_t> tile_first(n + 1);
  uint64_t in_extent = 0, in_lo = ~0ull, tiles = 0, in_total_bits = 0;
  for (size_t s = 0; s < n; s++) {
    sp[s].in_off = in_offs[s];
    sp[s].in_len = in_lens[s];
    sp[s].out_off = size_only ? 0 : out_offs[s];
    sp[s].out_cap = size_only ? ~0ull : out_caps[s];
    h_off[s] = in_offs[s];
    h_len[s] = in_lens[s];
    in_extent = std::max(in_extent, in_offs[s] + in_lens[s]);
    if (in_lens[s]) in_lo = std::min(in_lo, in_offs[s]);
    in_total_bits += in_lens[s] * 8;
    tile_first[s] = (uint32_t)tiles;
    // tiles are 64 KiB of memory starting at the stream's first octet rounded down to 16 (see K0)
    tiles += ((((uintptr_t)d_in + in_offs[s]) & 15) + in_lens[s] + SCAN_TILE - 1) / SCAN_TILE;
    if (tiles > 0x7fffffffu) return TBZ_E_ARG;
  }
  tile_first[n] = (uint32_t)tiles;
  if (in_extent && !d_in) return TBZ_E_ARG;

  if ((r = record(ctx, 0))) return r;
  // ---------------------------------------------------------------- K0: markers + items, on the device
  // One pass over the input finds the markers (kept per tile, then compacted in order) and a small kernel
  // builds the K1 items; the host reads 8 bytes + one index per stream.  Host copies of the marker and item
  // arrays are fetched only by the general (host) layout path.
  std::vector<Item> items;
  bool host_tables = false, have_find = false, have_resolve = false;
  uint32_t n_mark = 0, n_fixed_items = 0;  // flush markers; how many of them are followed by a fixed-Huffman block
  uint32_t n_stored_heads = 0;             // streams that begin with a stored block
  std::vector<uint32_t> first_marker(n + 1, 0);
  if (tiles) {
    if ((r = upload(ctx, ctx->d_str_off, h_off))) return r;
    if ((r = upload(ctx, ctx->d_str_len, h_len))) return r;
    if ((r = upload(ctx, ctx->d_tile_first, tile_first))) return r;
    if ((r = ensure(ctx, ctx->d_tile_counts, tiles * 4))) return r;
    if ((r = ensure(ctx, ctx->d_tile_offsets, (tiles + 1) * 4))) return r;
    if ((r = ensure(ctx, ctx->d_k0_slots, tiles * (size_t)K0_SLOTS * 8))) return r;
    if ((r = ensure(ctx, ctx->d_markers, tiles * (size_t)K0_SLOTS * 8 + 16))) return r;
    if ((r = ensure(ctx, ctx->d_k0_fm, (n + 1) * 4 + 16))) return r;  // [0..1] head, [2..] first_marker, [n+3] fixed-block items, [n+4] stored heads
    if ((r = ensure(ctx, ctx->d_items, (tiles * (size_t)K0_SLOTS + n) * sizeof(Item)))) return r;
    if ((r = pinned(ctx, (n + 5) * 4 + 16 + sizeof(K3Global) + (n + 1) * sizeof(K3Stream)))) return r;
    K0Params k0{(const u8*)d_in, (const u64*)ctx->d_str_off.p, (const u64*)ctx->d_str_len.p,
                (const u32*)ctx->d_tile_first.p, (u32)n, (u32)tiles, (u32*)ctx->d_tile_counts.p,
                (u32*)ctx->d_tile_offsets.p, (u64*)ctx->d_markers.p, (u64*)ctx->d_k0_slots.p,
                (u32*)ctx->d_k0_fm.p + 2, (u32*)ctx->d_k0_fm.p, (Item*)ctx->d_items.p, (u32)format, 0, bit_off,
                (u32*)ctx->d_k0_fm.p + (n + 3), resume_abs ? 1u : 0u};
    const size_t max_items = tiles * (size_t)K0_SLOTS + n;
    TBZ_LAUNCH(tbz_k0_scan_tiles, tiles, ctx->stream, k0);
    TBZ_LAUNCH_WG(tbz_k0_scan_offsets, 1, K0_SCAN_THREADS, ctx->stream, k0);
    TBZ_LAUNCH(tbz_k0_compact, tiles, ctx->stream, k0);
    TBZ_LAUNCH(tbz_k0_items, (max_items + 63) / 64, ctx->stream, k0);
    uint32_t* h_head = (uint32_t*)ctx->h_pin;
    TBZ_HIP(hipMemcpyAsync(h_head, ctx->d_k0_fm.p, (n + 5) * 4, hipMemcpyDeviceToHost, ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    n_mark = h_head[0];
    n_fixed_items = h_head[n + 3];
    n_stored_heads = h_head[n + 4];
    for (size_t s = 0; s <= n; s++) first_marker[s] = h_head[2 + s];
    if (h_head[1]) {  // a tile with more markers than slots: the second, emitting pass
      if ((r = ensure(ctx, ctx->d_markers, (size_t)n_mark * 8 + 16))) return r;
      if ((r = ensure(ctx, ctx->d_items, ((size_t)n_mark + n) * sizeof(Item)))) return r;
      k0.markers = (u64*)ctx->d_markers.p;
      k0.items = (Item*)ctx->d_items.p;
      k0.second_pass = 1;
      TBZ_LAUNCH(tbz_k0_scan_emit, tiles, ctx->stream, k0);
      TBZ_LAUNCH(tbz_k0_items, ((size_t)n_mark + n + 63) / 64, ctx->stream, k0);
    }
    TBZ_HIP(hipGetLastError());
    for (size_t s = 0; s < n; s++) {
      StreamPlan& S = sp[s];
      S.first_marker = first_marker[s];
      S.first_item = first_marker[s] + (uint32_t)s;
      S.n_items = 1 + (first_marker[s + 1] - first_marker[s]);
      S.cur_item = S.first_item;
    }
  } else {
    // nothing to scan (every stream empty): one head item per stream, built here
    if ((r = ensure(ctx, ctx->d_markers, 16))) return r;
    if ((r = ensure(ctx, ctx->d_k0_fm, (n + 1) * 4 + 8))) return r;
    std::vector<uint32_t> zero(n + 3, 0);
    if ((r = upload(ctx, ctx->d_k0_fm, zero))) return r;
    if ((r = pinned(ctx, (n + 5) * 4 + 16 + sizeof(K3Global) + (n + 1) * sizeof(K3Stream)))) return r;
    for (size_t s = 0; s < n; s++) {
      StreamPlan& S = sp[s];
      Item it{};
      it.start_bit = S.in_off * 8 + (s == 0 ? bit_off : 0);
      it.limit_bit = ~0ull;
      it.end_byte = S.in_off + S.in_len;
      it.stream = (uint32_t)s;
      it.flags = ((uint32_t)format << ITEM_FMT_SHIFT) | ITEM_HEAD | ((s == 0 && resume_abs) ? ITEM_RESUME : 0u);
      items.push_back(it);
      S.first_item = (uint32_t)s;
      S.first_marker = 0;
      S.n_items = 1;
      S.cur_item = S.first_item;
    }
    if ((r = upload(ctx, ctx->d_items, items))) return r;
    host_tables = true;
  }
  // ---------------------------------------------------------------- K0b: speculative block starts
  // Streams whose items are large (few or no flush markers: ordinary zlib / gzip output) are searched for plausible
  // dynamic-Huffman block headers; the candidates join the markers (one ascending list of bit positions per stream)
  // and the items are built again from the merged list.  Nothing downstream tells a candidate from a marker.
  const u64* d_markers_cur = (const u64*)ctx->d_markers.p;
  const u32* d_first_marker = (const u32*)ctx->d_k0_fm.p + 2;
  if (tiles && ctx->find_mode) {
    constexpr uint64_t FIND_MIN_ITEM_BITS = 8ull * (48u << 10);  // mean compressed octets per item below which it does not pay
    // ... nor when the call already has enough items to fill the chip (a batch of thousands of streams: measured on
    // config 3, 4096 gzip members, splitting them cost more in K2's second plane than it gained in K1)
    // ... nor when every stream begins with a stored block (stored data: nothing to find; config 1)
    const bool enough = (size_t)n_mark + n >= (size_t)ctx->tun.find_enough || (n_stored_heads >= n && ctx->find_mode != 2);
    const uint64_t find_min_len = (uint64_t)ctx->tun.find_min_len;
    std::vector<uint32_t> tfb(n + 1);
    uint64_t tiles_b = 0;
    for (size_t s = 0; s < n; s++) {
      tfb[s] = (uint32_t)tiles_b;
      const uint64_t items_s = 1 + (first_marker[s + 1] - first_marker[s]);
      const bool search = ctx->find_mode == 2 ? sp[s].in_len >= 64 : (!enough && sp[s].in_len >= find_min_len && sp[s].in_len * 8 / items_s >= FIND_MIN_ITEM_BITS);
      if (search) tiles_b += ((((uintptr_t)d_in + in_offs[s]) & 15) + in_lens[s] + K0B_TILE - 1) / K0B_TILE;
      if (tiles_b > 0x7fffffffu) return TBZ_E_ARG;
    }
    tfb[n] = (uint32_t)tiles_b;
    if (tiles_b) {
      TBZ_HIP(hipEventRecord(ctx->ev[8], ctx->stream));
      if ((r = upload(ctx, ctx->d_kb_tf, tfb))) return r;
      if ((r = ensure(ctx, ctx->d_kb_slots, tiles_b * (size_t)K0B_SLOTS * 8))) return r;
      if ((r = ensure(ctx, ctx->d_kb_counts, tiles_b * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_kcounts, tiles_b * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_keep, tiles_b * (size_t)(K0B_SLOTS / 64) * 8))) return r;
      if ((r = ensure(ctx, ctx->d_kb_offsets, (tiles_b + 1) * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_cands, tiles_b * (size_t)K0B_SLOTS * 8 + 16))) return r;
      if ((r = ensure(ctx, ctx->d_kb_fc, (n + 1) * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_head, 16))) return r;
      if ((r = ensure(ctx, ctx->d_kb_fm2, (n + 1) * 4 + 8))) return r;
      if ((r = ensure(ctx, ctx->d_markers2, ((size_t)n_mark + tiles_b * (size_t)K0B_SLOTS) * 8 + 16))) return r;
      K0bParams kb{(const u8*)d_in, (const u64*)ctx->d_str_off.p, (const u64*)ctx->d_str_len.p,
                   (const u32*)ctx->d_kb_tf.p, (u32)n, (u32)tiles_b, (u64*)ctx->d_kb_slots.p, (u32*)ctx->d_kb_counts.p,
                   (u32*)ctx->d_kb_offsets.p, (u64*)ctx->d_kb_cands.p, (u32*)ctx->d_kb_fc.p, (u32*)ctx->d_kb_head.p,
                   (const u64*)ctx->d_markers.p, (const u32*)ctx->d_k0_fm.p + 2, (u64*)ctx->d_markers2.p,
                   (u32*)ctx->d_kb_fm2.p + 2, (u32*)ctx->d_kb_fm2.p, bit_off, K0B_SLOTS,
                   ctx->tun.k0b_pair >= 0 ? (u32)(ctx->tun.k0b_pair != 0) : (tiles_b >= 8192 ? 1u : 0u), nullptr, nullptr,
                   (u64*)ctx->d_kb_keep.p, (u32*)ctx->d_kb_kcounts.p, 0};
      TBZ_LAUNCH(tbz_k0b_scan, tiles_b, ctx->stream, kb);
      // (a launch that fits the chip at once — 64 MiB of input — is as long as its slowest wave: a tile per wave; beyond
      // that it is throughput that counts: two tiles per wave, 32 lanes each)
      TBZ_LAUNCH(tbz_k0b_validate, kb.pair ? (tiles_b + 1) / 2 : tiles_b, ctx->stream, kb);
#ifdef TBZ_WAVE_TRACE
      if (const char* vp = getenv("TBZ_VAL_TRACE")) {
        std::vector<u64> h(8192 * 8);
        hipStreamSynchronize(ctx->stream);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tbz_dbg), h.size() * 8);
        if (FILE* f = fopen(vp, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
#endif
      TBZ_LAUNCH(tbz_k0b_space, tiles_b, ctx->stream, kb);  // (candidates too close to a marker, the head or each other: dropped)
      TBZ_LAUNCH_WG(tbz_k0b_offsets, 1, K0B_SCAN_THREADS, ctx->stream, kb);
      TBZ_LAUNCH(tbz_k0b_compact, tiles_b, ctx->stream, kb);
      const size_t max_merge = (size_t)n_mark + tiles_b * (size_t)K0B_SLOTS;
      TBZ_LAUNCH(tbz_k0b_merge, (max_merge + 63) / 64, ctx->stream, kb);
      uint32_t* h_head = (uint32_t*)ctx->h_pin;
      TBZ_HIP(hipMemcpyAsync(h_head, ctx->d_kb_fm2.p, (n + 3) * 4, hipMemcpyDeviceToHost, ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      const uint32_t n_merged = h_head[0];
      ctx->tim.n_candidates = n_merged - n_mark;
      if (const char* dp = ctx->tun.debug_cands.empty() ? nullptr : ctx->tun.debug_cands.c_str()) {  // the merged list of bit positions, for tools/ that compare it with the true block starts
        std::vector<uint64_t> hm(n_merged);
        TBZ_HIP(hipMemcpy(hm.data(), ctx->d_markers2.p, hm.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(dp, "wb")) {
          fwrite(hm.data(), 8, hm.size(), f);
          fclose(f);
        }
      }
      if (n_merged != n_mark) {
        n_mark = n_merged;
        for (size_t s = 0; s <= n; s++) first_marker[s] = h_head[2 + s];
        if ((r = ensure(ctx, ctx->d_items, ((size_t)n_mark + n) * sizeof(Item)))) return r;
        K0Params k0m{};
        k0m.str_off = (const u64*)ctx->d_str_off.p;
        k0m.str_len = (const u64*)ctx->d_str_len.p;
        k0m.n_streams = (u32)n;
        k0m.markers = (u64*)ctx->d_markers2.p;
        k0m.first_marker = (u32*)ctx->d_kb_fm2.p + 2;
        k0m.head = (u32*)ctx->d_kb_fm2.p;
        k0m.items = (Item*)ctx->d_items.p;
        k0m.format = (u32)format;
        k0m.second_pass = 1;
        k0m.start_bit_off = bit_off;
        k0m.resume = resume_abs ? 1u : 0u;
        TBZ_LAUNCH(tbz_k0_items, ((size_t)n_mark + n + 63) / 64, ctx->stream, k0m);
        d_markers_cur = (const u64*)ctx->d_markers2.p;
        d_first_marker = (const u32*)ctx->d_kb_fm2.p + 2;
        for (size_t s = 0; s < n; s++) {
          StreamPlan& S = sp[s];
          S.first_marker = first_marker[s];
          S.first_item = first_marker[s] + (uint32_t)s;
          S.n_items = 1 + (first_marker[s + 1] - first_marker[s]);
          S.cur_item = S.first_item;
        }
      }
      TBZ_HIP(hipGetLastError());
      TBZ_HIP(hipEventRecord(ctx->ev[9], ctx->stream));
      have_find = true;
    }
  }
  // ---------------------------------------------------------------- K0c: chains of fixed-Huffman blocks
  // Streams whose items are STILL large (K0 and K0b found little in them: fixed-Huffman or stored territory) are
  // searched for "end-of-block + BTYPE 1" patterns; each hit's one block is skimmed and the hits that chain are kept.
  if (tiles && ctx->find_mode) {
    constexpr uint64_t FIXED_MIN_ITEM_BITS = 8ull * (256u << 10);
    // flush-delimited items that begin with fixed-Huffman blocks (K0 counted them) are chains of such blocks, and a
    // periodic bitstream — what fixed-Huffman territory tends to be — never lets K1's lanes fall into step: block starts
    // are what makes such items parallel, so they are searched from 32 Kbit per item on
    const bool fixed_territory = n_fixed_items != 0 && 2 * (uint64_t)n_fixed_items >= n_mark;
    const uint64_t min_item_bits = fixed_territory ? 32u << 10 : FIXED_MIN_ITEM_BITS;
    std::vector<uint32_t> tfc(n + 1);
    uint64_t tiles_c = 0;
    const bool enough = (size_t)n_mark + n >= 2048;
    for (size_t s = 0; s < n; s++) {
      tfc[s] = (uint32_t)tiles_c;
      const uint64_t items_s = 1 + (first_marker[s + 1] - first_marker[s]);
      const bool search = ctx->find_mode == 2 ? sp[s].in_len >= 64
                                              : (!enough && sp[s].in_len * 8 / items_s >= min_item_bits);
      if (search) tiles_c += ((((uintptr_t)d_in + in_offs[s]) & 15) + in_lens[s] + K0C_TILE - 1) / K0C_TILE;
      if (tiles_c > 0x7fffffffu) return TBZ_E_ARG;
    }
    tfc[n] = (uint32_t)tiles_c;
    if (tiles_c) {
      if (!have_find) TBZ_HIP(hipEventRecord(ctx->ev[8], ctx->stream));
      const size_t nslot = tiles_c * (size_t)K0C_SLOTS;
      if ((r = upload(ctx, ctx->d_kc_tf, tfc))) return r;
      if ((r = ensure(ctx, ctx->d_kc_slots, nslot * 8))) return r;
      if ((r = ensure(ctx, ctx->d_kc_ends, nslot * 8))) return r;
      if ((r = ensure(ctx, ctx->d_kc_link, nslot * 2))) return r;
      if ((r = ensure(ctx, ctx->d_kb_counts, tiles_c * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_kcounts, tiles_c * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_keep, tiles_c * (size_t)(K0C_SLOTS / 64) * 8))) return r;
      if ((r = ensure(ctx, ctx->d_kb_offsets, (tiles_c + 1) * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_cands, nslot * 8 + 16))) return r;
      if ((r = ensure(ctx, ctx->d_kb_fc, (n + 1) * 4))) return r;
      if ((r = ensure(ctx, ctx->d_kb_head, 16))) return r;
      if ((r = ensure(ctx, ctx->d_kc_fm2, (n + 1) * 4 + 8))) return r;
      if ((r = ensure(ctx, ctx->d_markers3, ((size_t)n_mark + nslot) * 8 + 16))) return r;
      TBZ_HIP(hipMemsetAsync(ctx->d_kc_link.p, 0, nslot * 2, ctx->stream));
      K0bParams kc{(const u8*)d_in, (const u64*)ctx->d_str_off.p, (const u64*)ctx->d_str_len.p,
                   (const u32*)ctx->d_kc_tf.p, (u32)n, (u32)tiles_c, (u64*)ctx->d_kc_slots.p, (u32*)ctx->d_kb_counts.p,
                   (u32*)ctx->d_kb_offsets.p, (u64*)ctx->d_kb_cands.p, (u32*)ctx->d_kb_fc.p, (u32*)ctx->d_kb_head.p,
                   d_markers_cur, d_first_marker, (u64*)ctx->d_markers3.p,
                   (u32*)ctx->d_kc_fm2.p + 2, (u32*)ctx->d_kc_fm2.p, bit_off, K0C_SLOTS, 0, (u64*)ctx->d_kc_ends.p,
                   (u8*)ctx->d_kc_link.p, (u64*)ctx->d_kb_keep.p, (u32*)ctx->d_kb_kcounts.p,
                   ctx->tun.k0c_max_block ? (u64)ctx->tun.k0c_max_block : K0C_MAX_BLOCK};
      TBZ_LAUNCH(tbz_k0c_scan, tiles_c, ctx->stream, kc);
      TBZ_LAUNCH(tbz_k0c_skim, tiles_c * (size_t)(K0C_SLOTS / 64), ctx->stream, kc);
      TBZ_LAUNCH(tbz_k0c_link, tiles_c, ctx->stream, kc);
      TBZ_LAUNCH(tbz_k0c_filter, tiles_c, ctx->stream, kc);
      TBZ_LAUNCH_WG(tbz_k0b_offsets, 1, K0B_SCAN_THREADS, ctx->stream, kc);
      TBZ_LAUNCH(tbz_k0b_compact, tiles_c, ctx->stream, kc);
      const size_t max_merge = (size_t)n_mark + nslot;
      TBZ_LAUNCH(tbz_k0b_merge, (max_merge + 63) / 64, ctx->stream, kc);
      uint32_t* h_head = (uint32_t*)ctx->h_pin;
      TBZ_HIP(hipMemcpyAsync(h_head, ctx->d_kc_fm2.p, (n + 3) * 4, hipMemcpyDeviceToHost, ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      const uint32_t n_merged = h_head[0];
      ctx->tim.n_candidates += n_merged - n_mark;
      if (n_merged != n_mark) {
        n_mark = n_merged;
        for (size_t s = 0; s <= n; s++) first_marker[s] = h_head[2 + s];
        if ((r = ensure(ctx, ctx->d_items, ((size_t)n_mark + n) * sizeof(Item)))) return r;
        K0Params k0m{};
        k0m.str_off = (const u64*)ctx->d_str_off.p;
        k0m.str_len = (const u64*)ctx->d_str_len.p;
        k0m.n_streams = (u32)n;
        k0m.markers = (u64*)ctx->d_markers3.p;
        k0m.first_marker = (u32*)ctx->d_kc_fm2.p + 2;
        k0m.head = (u32*)ctx->d_kc_fm2.p;
        k0m.items = (Item*)ctx->d_items.p;
        k0m.format = (u32)format;
        k0m.second_pass = 1;
        k0m.start_bit_off = bit_off;
        k0m.resume = resume_abs ? 1u : 0u;
        TBZ_LAUNCH(tbz_k0_items, ((size_t)n_mark + n + 63) / 64, ctx->stream, k0m);
        d_markers_cur = (const u64*)ctx->d_markers3.p;
        d_first_marker = (const u32*)ctx->d_kc_fm2.p + 2;
        for (size_t s = 0; s < n; s++) {
          StreamPlan& S = sp[s];
          S.first_marker = first_marker[s];
          S.first_item = first_marker[s] + (uint32_t)s;
          S.n_items = 1 + (first_marker[s + 1] - first_marker[s]);
          S.cur_item = S.first_item;
        }
      }
      TBZ_HIP(hipGetLastError());
      TBZ_HIP(hipEventRecord(ctx->ev[9], ctx->stream));
      have_find = true;
    }
  }
  const size_t n_items = (size_t)n_mark + n;
  auto fetch_host_tables = [&]() -> int {  // the general layout path walks the items on the host
    if (host_tables) return 0;
    items.resize(n_items);
    TBZ_HIP(hipMemcpyAsync(items.data(), ctx->d_items.p, n_items * sizeof(Item), hipMemcpyDeviceToHost, ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    host_tables = true;
    return 0;
  };
  if ((r = record(ctx, 1))) return r;

  // token pool: one u16 per TWO input bits where gangs decode (a token of ordinary data takes eight bits and more; a
  // lane whose region fills up ends its run early and an item that does not fit is declined — SEG_REDO — and decoded
  // again by one lane into a region of its own: `dense`), one per bit where the one-lane kernel does (K1 never writes
  // more words than bits consumed); run tables: one 16-octet slot per 2^RUN_SHIFT input bits.  Both are addressed by
  // bit position RELATIVE to the first stream octet of the call (pool_base: a batch that is a window into a large
  // buffer pays for its own extent only); the repair launches' pools cover the repaired streams' tails only (pool2_base).
  if (in_lo > in_extent) in_lo = in_extent;
  const uint64_t pool_base = (in_lo * 8) & ~(uint64_t)((1u << RUN_SHIFT) - 1);  // bits
  const uint64_t pool_bits = in_extent * 8 - pool_base;
  uint64_t pool2_base = pool_base, pool2_hi = 0;
  u32 half1 = 0, half2 = 0;  // log2 of the input bits per token word of the call's pool / the repair pool
  if (!ctx->dense.empty()) {  // (the last call's; rare)
    for (DevBuf& b : ctx->dense)
      if (b.p) hipFree(b.p);
    ctx->dense.clear();
  }
  if ((r = ensure(ctx, ctx->d_res, n_items * sizeof(SegResult)))) return r;
  // K1 flavour.  One lane per item is bound by ONE item's serial chain (~1.1 us per token) and decodes
  // without lookup tables; a gang of G lanes shares one item and one set of LDS tables.  G follows the
  // average item size (a lane should get at least ~2 Kibit of bitstream per round); only batches of
  // many tiny items (table set-up would dominate) stay with one lane per item.
  auto k1_gang = [&](size_t n_it) -> int {
    if (ctx->k1_mode) return ctx->k1_mode;
    uint64_t avg_bits = n_it ? in_total_bits / n_it : 0;
    if (avg_bits < 8192 && n_it >= 16384) return 1;
    int G = 8;
    while (G < 64 && avg_bits > (uint64_t)G * 3072) G <<= 1;
    return G;
  };
  half1 = (k1_gang(n_items) == 1 || ctx->tun.tok_full) ? 0u : 1u;
  if ((r = ensure(ctx, ctx->d_tok, ((size_t)pool_bits >> half1) * 2 + 256))) return r;
  if ((r = ensure(ctx, ctx->d_runs, ((size_t)pool_bits >> RUN_SHIFT) * sizeof(RunRec) + 1024))) return r;
  // regions of their own for items that are decoded again: one word per bit of each item, and a run table
  auto make_explicit = [&](std::vector<Item>& its) -> int {
    uint64_t words = 0, slots = 0;
    std::vector<uint64_t> at(its.size()), rat(its.size());
    for (size_t k = 0; k < its.size(); k++) {
      const Item& q = its[k];
      const uint64_t hi = (q.flags & ITEM_FIXUP) ? q.end_byte * 8 : std::min(q.limit_bit, q.end_byte * 8);
      const uint64_t span = hi > q.start_bit ? hi - q.start_bit : 0;
      at[k] = words;
      rat[k] = slots;
      words += (span + 64 + 7) & ~7ull;
      slots += (span >> RUN_SHIFT) + 2;
    }
    ctx->dense.emplace_back();
    int rr = ensure(ctx, ctx->dense.back(), (size_t)words * 2 + (size_t)slots * sizeof(RunRec) + 256);
    if (rr) return rr;
    u16* tk = (u16*)ctx->dense.back().p;
    RunRec* rn = (RunRec*)(tk + words);  // (words is a multiple of 8: 16-octet aligned)
    for (size_t k = 0; k < its.size(); k++) {
      its[k].flags |= ITEM_EXPLICIT;
      its[k].tok = (uint64_t)(tk + at[k]);
      its[k].runs = (uint64_t)(rn + rat[k]);
    }
    return 0;
  };
  auto items_per_wg = [](size_t n_it) {  // lane-per-item flavour: spread few items over all CUs
    u32 ipw = 64;
    while (ipw > 1 && n_it / ipw < 512) ipw >>= 1;
    return ipw;
  };
  // Repair (fix-up) launches decode into pools of their own: a gang that repairs an item runs past the marker
  // it will land on, into the bit range of items whose tokens (position-addressed!) are already in place.
  // (kernels index the pools by absolute bit position: the pointers handed to them are shifted down by the pool's base)
  auto pool_half = [&](bool fix) { return fix ? half2 : half1; };
  auto pool_tok = [&](bool fix) {
    return fix ? (u16*)ctx->d_tok2.p - ((pool2_base >> half2) & ~7ull) : (u16*)ctx->d_tok.p - ((pool_base >> half1) & ~7ull);
  };
  auto pool_runs = [&](bool fix) {
    return fix ? (RunRec*)ctx->d_runs2.p - (pool2_base >> RUN_SHIFT) : (RunRec*)ctx->d_runs.p - (pool_base >> RUN_SHIFT);
  };
  // (the one-lane kernel writes one word per bit at most: its items live in a pool of that density, or bring their own regions)
  auto launch_lane = [&](const Item* d_items, SegResult* d_res, size_t n_it, bool fix, bool explicit_items = false) -> int {
    if (pool_half(fix) && !explicit_items) return TBZ_E_INTERNAL;
    int rr = ensure(ctx, ctx->d_scratch, n_it * (size_t)K1_SCRATCH);
    if (rr) return rr;
    K1Params k1{(const u8*)d_in, pool_tok(fix), d_items, d_res, d_markers_cur,
                (u8*)ctx->d_scratch.p, pool_runs(fix), d_first_marker, (u32)n_mark, (u32)n_it, items_per_wg(n_it), resume_abs};
    TBZ_LAUNCH(tbz_k1_huff_decode, (n_it + k1.items_per_wg - 1) / k1.items_per_wg, ctx->stream, k1);
    return 0;
  };
  auto sub_min_for = [&](int) -> u32 {
    if (ctx->tun.sub_min) return (u32)std::max(64, ctx->tun.sub_min & ~63);
    return KG_SUB_MIN;
  };
  // a gang narrower than 64 lanes declines items it would need more than two full rounds for (the width follows the
  // launch's mean item size: a batch of many small streams and one large one, or K0c's short items and one long block
  // among them); the host decodes those with gangs of 64 (`redo`).  (Eight rounds until late in round 3: config 5's
  // 294 Kbit block of one repeated match stayed on its 8-lane gang for 2.4 ms while the rest of the launch was done
  // after 0.8 — a wider gang had been WORSE there as long as lanes could not fall into step on a periodic bitstream;
  // with kg_periodic they start on a token.)  Forced flavours (tests) keep everything.
  auto wide_for = [&](int G, bool fix) -> u64 {
    if (ctx->tun.wide_bits >= 0) return (u64)ctx->tun.wide_bits;
    if (ctx->k1_mode || fix || G >= 64) return 0;
    return (u64)G * KG_SUB_MAX * 2;
  };
  auto ovl_for = [&](int G) -> u32 {
    if (ctx->tun.ovl) return (u32)std::max(64, ctx->tun.ovl);
    // measured (profiles/README.md): a gang of 64 commits 29 lanes per round at 512 bits of run-up, 59 at 1024 (K1 on
    // the 64 MiB no-flush stream 2.78 -> 1.56 ms, on config 3 8.96 -> 6.05 ms).  Gangs of 32: config 2's text is flat
    // from 512 to 1024 bits (2.56 - 2.68 ms), but a sync-flush stream — more and longer matches, because history
    // reaches across the flush points, so two parses take longer to fall into step — needed 3.2 rounds per item at 512
    // bits (K1 on config 2b 1.28 ms; 0.95 at 768, 0.80 at 1024).  Narrow gangs (K0c's short items, config 5): 768 is best.
    return G >= 32 ? 1024u : 768u;
  };
  // items the gang kernel declined (SEG_REDO: more token words than the pool has for their bits — long runs of one
  // octet code that way —, run table full) are decoded again into regions of their own: by gangs of 64 where the pool
  // holds one word per two bits, and what those decline, or all of them, by the one-lane kernel, which writes one
  // contiguous run
  auto redo = [&](const std::vector<Item>& its, std::vector<SegResult>& rs, bool fix) -> int {
    {  // items a narrow gang handed back (SEG_WIDE): gangs of 64, the leader parsing the first header itself
      std::vector<Item> wide;
      std::vector<size_t> widx;
      for (size_t i = 0; i < rs.size(); i++)
        if (rs[i].status == SEG_WIDE) {
          wide.push_back(its[i]);
          widx.push_back(i);
        }
      if (!wide.empty()) {
        if (ctx->tun.debug) {
          uint64_t mx = 0;
          for (const Item& q : wide) mx = std::max<uint64_t>(mx, std::min(q.limit_bit, q.end_byte * 8) - q.start_bit);
          fprintf(stderr, "tbz: %zu large item(s) handed to gangs of 64 (the largest: %llu bits)\n", wide.size(), (unsigned long long)mx);
        }
        int rr;
        if ((rr = upload(ctx, ctx->d_redo_items, wide))) return rr;
        if ((rr = ensure(ctx, ctx->d_redo_res, wide.size() * sizeof(SegResult)))) return rr;
        if ((rr = ensure(ctx, ctx->d_cold, wide.size() * (size_t)KG_COLD_STRIDE))) return rr;
        K1gParams kg{(const u8*)d_in, pool_tok(fix), pool_runs(fix), pool_half(fix), 0, (const Item*)ctx->d_redo_items.p,
                     (SegResult*)ctx->d_redo_res.p, d_markers_cur, d_first_marker, nullptr, nullptr, (u32)n_mark,
                     (u32)wide.size(), ovl_for(64), sub_min_for(64), 0, resume_abs, (u8*)ctx->d_cold.p};
#ifdef TBZ_WAVE_TRACE
        kg.trace = nullptr;
#endif
        TBZ_LAUNCH(tbz_k1g64_huff_decode, wide.size(), ctx->stream, kg);
        std::vector<SegResult> tmp(wide.size());
        TBZ_HIP(hipMemcpyAsync(tmp.data(), ctx->d_redo_res.p, tmp.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                               ctx->stream));
        TBZ_HIP(hipStreamSynchronize(ctx->stream));
        for (size_t k = 0; k < widx.size(); k++) rs[widx[k]] = tmp[k];
        ctx->tim.huff_launches++;
      }
    }
    std::vector<Item> sub;
    std::vector<size_t> idx;
    for (size_t i = 0; i < rs.size(); i++)
      if (rs[i].status == SEG_REDO) {
        sub.push_back(its[i]);
        idx.push_back(i);
      }
    if (sub.empty()) return 0;
    if (pool_half(fix) && !ctx->k1_mode) {
      if (ctx->tun.debug) fprintf(stderr, "tbz: %zu item(s) decoded again into regions of their own\n", sub.size());
      int rr;
      if ((rr = make_explicit(sub))) return rr;
      if ((rr = upload(ctx, ctx->d_redo_items, sub))) return rr;
      if ((rr = ensure(ctx, ctx->d_redo_res, sub.size() * sizeof(SegResult)))) return rr;
      if ((rr = ensure(ctx, ctx->d_cold, sub.size() * (size_t)KG_COLD_STRIDE))) return rr;
      K1gParams kg{(const u8*)d_in, nullptr, nullptr, 0, 0, (const Item*)ctx->d_redo_items.p,
                   (SegResult*)ctx->d_redo_res.p, d_markers_cur, d_first_marker, nullptr, nullptr, (u32)n_mark,
                   (u32)sub.size(), ovl_for(64), sub_min_for(64), 0, resume_abs, (u8*)ctx->d_cold.p};
#ifdef TBZ_WAVE_TRACE
      kg.trace = nullptr;
#endif
      TBZ_LAUNCH(tbz_k1g64_huff_decode, sub.size(), ctx->stream, kg);
      std::vector<SegResult> tmp(sub.size());
      TBZ_HIP(hipMemcpyAsync(tmp.data(), ctx->d_redo_res.p, tmp.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                             ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      ctx->tim.huff_launches++;
      std::vector<Item> sub2;
      std::vector<size_t> idx2;
      for (size_t k = 0; k < idx.size(); k++) {
        rs[idx[k]] = tmp[k];
        if (tmp[k].status == SEG_REDO) {
          sub2.push_back(its[idx[k]]);
          idx2.push_back(idx[k]);
        }
      }
      sub.swap(sub2);
      idx.swap(idx2);
      if (sub.empty()) return 0;
    }
    if (ctx->tun.debug) fprintf(stderr, "tbz: %zu item(s) redone by the one-lane kernel\n", sub.size());
    int rr;
    if ((rr = make_explicit(sub))) return rr;
    if ((rr = upload(ctx, ctx->d_redo_items, sub))) return rr;
    if ((rr = ensure(ctx, ctx->d_redo_res, sub.size() * sizeof(SegResult)))) return rr;
    if ((rr = launch_lane((const Item*)ctx->d_redo_items.p, (SegResult*)ctx->d_redo_res.p, sub.size(), fix, true))) return rr;
    std::vector<SegResult> tmp(sub.size());
    TBZ_HIP(hipMemcpyAsync(tmp.data(), ctx->d_redo_res.p, tmp.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                           ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    for (size_t k = 0; k < idx.size(); k++) rs[idx[k]] = tmp[k];
    ctx->tim.huff_launches++;
    return 0;
  };
  bool wide_beside = false;  // the main launch's large items were decoded by gangs of 64 on the second stream (d_wide_res)
  auto launch_k1 = [&](const Item* d_items, SegResult* d_res, size_t n_it, bool fix) -> int {
    int G = k1_gang(n_it);
    if (G == 1 && pool_half(fix)) G = 8;  // (a repair launch of many small items into a pool laid out for gangs)
    if (!fix) ctx->tim.k1_gang = (uint32_t)G;
    if (G == 1) return launch_lane(d_items, d_res, n_it, fix);
    size_t per = 64 / G, nwg = (n_it + per - 1) / per;
    // A launch of NARROW gangs (short items) declines the few items that are far larger than its mean (SEG_WIDE).
    // Gangs of 64 take exactly those — a second kernel over the same item list on the second stream, every workgroup of
    // which leaves at once unless its item is such a one — BESIDE the narrow launch instead of after the host has read
    // its results (config 5: one 294 Kbit block among 3 548 items: 0.77 + 0.8 ms one after the other).  Their results
    // go to an array of their own (the narrow gang writes SEG_WIDE into the item's record).
    if (!fix && G <= 16 && wide_for(G, fix) && ctx->stream2) {  // (gangs of 32: the 65 536 workgroups of config 2 that leave at once cost K1 1 - 2 %: measured)
      int rr = ensure(ctx, ctx->d_wide_res, n_it * sizeof(SegResult));
      if (rr) return rr;
      if ((rr = ensure(ctx, ctx->d_cold2, n_it * (size_t)KG_COLD_STRIDE))) return rr;  // (only the large items' slots are touched)
      TBZ_HIP(hipEventRecord(ctx->evw[0], ctx->stream));  // (the items are there)
      TBZ_HIP(hipStreamWaitEvent(ctx->stream2, ctx->evw[0], 0));
      K1gParams kw{(const u8*)d_in, pool_tok(fix), pool_runs(fix), pool_half(fix), 1, d_items, (SegResult*)ctx->d_wide_res.p,
                   d_markers_cur, d_first_marker, nullptr, nullptr, (u32)n_mark, (u32)n_it, ovl_for(64), sub_min_for(64),
                   wide_for(G, fix), resume_abs, (u8*)ctx->d_cold2.p};
#ifdef TBZ_WAVE_TRACE
      kw.trace = nullptr;
#endif
      TBZ_LAUNCH(tbz_k1g64_huff_decode, n_it, ctx->stream2, kw);
      TBZ_HIP(hipEventRecord(ctx->evw[1], ctx->stream2));
      wide_beside = true;
    }
    // K1h: every lane parses the first block header of its own item, so that the gangs need not (their leaders
    // would do it with 2 of 64 lanes busy)
    int rr;
    if ((rr = ensure(ctx, ctx->d_scratch, n_it * (size_t)K1_SCRATCH))) return rr;
    if ((rr = ensure(ctx, ctx->d_hdr, n_it * sizeof(HdrRec)))) return rr;
    if (G >= 32 && (rr = ensure(ctx, ctx->d_cold, n_it * (size_t)KG_COLD_STRIDE))) return rr;  // (narrower gangs keep their lists in LDS)
    K1hParams kh{(const u8*)d_in, d_items, (u8*)ctx->d_scratch.p, (HdrRec*)ctx->d_hdr.p, (u32)n_it};
    if (ctx->k1h) TBZ_LAUNCH(tbz_k1h_headers, (n_it + 63) / 64, ctx->stream, kh);
    K1gParams kg{(const u8*)d_in, pool_tok(fix), pool_runs(fix), pool_half(fix), 0, d_items, d_res,
                 d_markers_cur, d_first_marker, ctx->k1h ? (const HdrRec*)ctx->d_hdr.p : nullptr,
                 (const u8*)ctx->d_scratch.p, (u32)n_mark, (u32)n_it, ovl_for(G), sub_min_for(G), wide_for(G, fix), resume_abs,
                 (u8*)ctx->d_cold.p};
#ifdef TBZ_WAVE_TRACE
    {
      static int n_launch = 0;
      u32 ef = (n_launch++ >= 1 && getenv("TBZ_EXP")) ? (u32)atoi(getenv("TBZ_EXP")) : 0u;
      hipMemcpyToSymbolAsync(HIP_SYMBOL(tbz_exp_flags), &ef, 4, 0, hipMemcpyHostToDevice, ctx->stream);
      hipStreamSynchronize(ctx->stream);
    }
    static u64* d_trace = nullptr;
    static size_t trace_wg = 0;
    if (const char* tp = getenv("TBZ_WAVE_TRACE")) {
      if (d_trace && trace_wg) {  // the previous launch's records
        std::vector<u64> h(trace_wg * 16);
        hipStreamSynchronize(ctx->stream);
        hipMemcpy(h.data(), d_trace, h.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = fopen(tp, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
        hipFree(d_trace);
      }
      hipMalloc((void**)&d_trace, nwg * 128);
      hipMemsetAsync(d_trace, 0, nwg * 128, ctx->stream);
      trace_wg = nwg;
      kg.trace = d_trace;

    } else {
      kg.trace = nullptr;
    }
#endif
    switch (G) {
      case 8: TBZ_LAUNCH(tbz_k1g8_huff_decode, nwg, ctx->stream, kg); break;
      case 16: TBZ_LAUNCH(tbz_k1g16_huff_decode, nwg, ctx->stream, kg); break;
      case 32: TBZ_LAUNCH(tbz_k1g32_huff_decode, nwg, ctx->stream, kg); break;
      default: TBZ_LAUNCH(tbz_k1g64_huff_decode, nwg, ctx->stream, kg); break;
    }
    if (wide_beside && !fix) TBZ_HIP(hipStreamWaitEvent(ctx->stream, ctx->evw[1], 0));  // what follows needs both
    return 0;
  };
  // K3 (layout on the device when every item simply lands on its successor): its stream tables go up before
  // K1 so that nothing but two tiny kernels and a 16-byte read-back stand between K1 and the decision
  const bool try_simple = !ctx->host_layout && n_items != 0;
  const u32 k3_tiles = (u32)((n_items + K3_TILE - 1) / K3_TILE);
  K3Params k3{};
  if (try_simple) {
    std::vector<uint32_t> fi(n), ni(n);
    std::vector<uint64_t> oo(n), oc(n);
    for (size_t s = 0; s < n; s++) {
      fi[s] = sp[s].first_item;
      ni[s] = sp[s].n_items;
      oo[s] = sp[s].out_off;
      oc[s] = size_only ? 0 : sp[s].out_cap;
    }
    if ((r = upload(ctx, ctx->d_k3_fi, fi))) return r;
    if ((r = upload(ctx, ctx->d_k3_ni, ni))) return r;
    if ((r = upload(ctx, ctx->d_k3_oo, oo))) return r;
    if ((r = upload(ctx, ctx->d_k3_oc, oc))) return r;
    if ((r = ensure(ctx, ctx->d_k3_sums, 3 * (size_t)(k3_tiles + 1) * 8))) return r;
    if ((r = ensure(ctx, ctx->d_k3_flags, 4 * (size_t)k3_tiles * 8))) return r;
    if ((r = ensure(ctx, ctx->d_k3_gscan, n_items * 8))) return r;
    if ((r = ensure(ctx, ctx->d_k3_gne, n_items * 4))) return r;
    if ((r = ensure(ctx, ctx->d_segs, n_items * sizeof(Seg)))) return r;
    if ((r = ensure(ctx, ctx->d_groups, n_items * sizeof(Group)))) return r;
    if ((r = ensure(ctx, ctx->d_k3_streams, (n + 1) * sizeof(K3Stream)))) return r;
    if ((r = ensure(ctx, ctx->d_k3_glob, sizeof(K3Global)))) return r;
    k3 = K3Params{(const Item*)ctx->d_items.p, (const SegResult*)ctx->d_res.p, (const u32*)ctx->d_k3_fi.p,
                  (const u32*)ctx->d_k3_ni.p, (const u64*)ctx->d_k3_oo.p, (const u64*)ctx->d_k3_oc.p,
                  (u64*)ctx->d_k3_sums.p, (u64*)ctx->d_k3_flags.p, (u64*)ctx->d_k3_gscan.p, (u32*)ctx->d_k3_gne.p,
                  (Seg*)ctx->d_segs.p, (Group*)ctx->d_groups.p, (K3Stream*)ctx->d_k3_streams.p,
                  (K3Global*)ctx->d_k3_glob.p, (u32)n_items, k3_tiles, (u32)n};
  }
  if ((r = record(ctx, 2))) return r;
  if ((r = launch_k1((const Item*)ctx->d_items.p, (SegResult*)ctx->d_res.p, n_items, false))) return r;
  TBZ_HIP(hipGetLastError());
  if ((r = record(ctx, 3))) return r;
  ctx->tim.huff_launches = 1;
  bool simple = false;
  bool fused_adler = false;  // adler32 partials come from K2 (simple path, zlib, all groups in the two-wave kernel)
  char* pin_k3 = (char*)ctx->h_pin + (((n + 5) * 4 + 15) & ~(size_t)15);
  K3Global* h_glob = (K3Global*)pin_k3;
  K3Stream* h_k3s = (K3Stream*)(pin_k3 + sizeof(K3Global));
  if (try_simple) {
    TBZ_LAUNCH(tbz_k3_tile_sums, k3_tiles, ctx->stream, k3);
    TBZ_LAUNCH(tbz_k3_scan_tiles, 1, ctx->stream, k3);
    TBZ_HIP(hipMemcpyAsync(h_glob, ctx->d_k3_glob.p, sizeof(K3Global), hipMemcpyDeviceToHost, ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    simple = h_glob->not_simple == 0;
    // an item far larger than a fair share of the output is sliced (tbz_k3_slice): the general path lays that out
    if (simple && ctx->sym_hist && !size_only) {
      uint64_t target = std::max<uint64_t>(64u << 10, h_glob->total_out / 4096);
      if (ctx->tun.slice) target = std::max(1024, ctx->tun.slice);
      if (h_glob->max_out >= 2 * target) simple = false;
    }
  }
  // what a stream reports once its status is known (both layout paths)
  auto fill_result = [&](size_t s, int32_t status, uint32_t nseg, bool error_first = false) {
    StreamPlan& S = sp[s];
    tbz_result& R = results[s];
    // 3bz decodes front to back: it reports overflow as soon as a token does not fit, before it
    // could meet a later error / underrun
    uint64_t cap = S.out_cap;
    // … and inside a stored block it asks for output space before it asks for input (copy-byte-or-fail,
    // deflate.lisp:538-573): payload cut off exactly where the buffer is full is output-overflow
    bool overflow = (S.total_out > cap || (S.stored_cut && status == TBZ_INPUT_UNDERRUN && S.total_out == cap)) && !error_first;
    if (overflow) status = TBZ_OUTPUT_OVERFLOW;
    R.status = status;
    R.segments = nseg;
    R.out_total = S.total_out;
    R.out_len = overflow ? cap : S.total_out;
    // not finished: the last flush boundary (octet offset just after its 00 00 FF FF) the block chain landed on
    // — everything before it is decoded and delivered, a decoder may be restarted there (tbz_amd.h)
    R.in_consumed = S.saw_final ? (S.in_end_bit / 8 - S.in_off)
                                : (S.boundary_bit / 8 > S.in_off ? S.boundary_bit / 8 - S.in_off : 0);
    R.boundary_out = S.saw_final ? S.total_out : (S.boundary_bit / 8 > S.in_off ? S.boundary_out : 0);
    R.trailer_check = S.trailer0;
    R.trailer_isize = S.trailer1;
    if (S.saw_final) R.flags |= 2;
    if (S.stored_cut) R.flags |= 4;
    if (status < 0) R.out_len = 0;  // reference signals an error: no partial-result contract
    return status;
  };
  float huff_ms = 0;
  if (simple) {
    // ---------------------------------------------------------------- device layout + K2, host reads n+1 records
    TBZ_LAUNCH(tbz_k3_scan_items, k3_tiles, ctx->stream, k3);
    TBZ_LAUNCH(tbz_k3_emit, k3_tiles, ctx->stream, k3);
    TBZ_HIP(hipMemcpyAsync(h_k3s, ctx->d_k3_streams.p, (n + 1) * sizeof(K3Stream), hipMemcpyDeviceToHost, ctx->stream));
    TBZ_HIP(hipEventRecord(ctx->ev[7], ctx->stream));
    if ((r = record(ctx, 4))) return r;
    if (!size_only && opt && opt->alloc) {  // the buffer comes once the size is known: wait for the stream record
      TBZ_HIP(hipEventSynchronize(ctx->ev[7]));
      d_out = opt->alloc(h_k3s[0].total_out);
      if (!d_out) return TBZ_E_NOMEM;
    }
    if (!size_only) {
      if (!d_out) return TBZ_E_ARG;
      const u32 n_it = (u32)n_items;
      fused_adler = format == TBZ_FORMAT_ZLIB && !ctx->k2_single && h_glob->n_big == 0 && !ctx->tun.no_fused_adler;
      K2Params k2{(const Seg*)ctx->d_segs.p, (const Group*)ctx->d_groups.p, nullptr,
                  (const u8*)d_in, (u8*)d_out, n_it, 0, 0, nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0};
      if (h_glob->n_big < n_it) {
        k2.win_bytes = (u32)((h_glob->max_small + K2_SLACK + 63) & ~63ull);
        k2.cls = h_glob->n_big ? 1 : 0;
        if (fused_adler) {  // every group goes through the two-wave kernel: it leaves the adler32 partials behind
          if ((r = ensure(ctx, ctx->d_gck, (size_t)n_it * sizeof(CkPartial)))) return r;
          if ((r = ensure(ctx, ctx->d_gchunks, (size_t)n_it * sizeof(CkChunk)))) return r;
          k2.gck = (CkPartial*)ctx->d_gck.p;
          k2.gchunks = (CkChunk*)ctx->d_gchunks.p;
        }
        ctx->tim.k2_kinds |= ctx->k2_single ? 2u : 1u;
        if (ctx->k2_single)
          TBZ_LAUNCH_DYN(tbz_k2_lz77_small, n_it, k2.win_bytes + 2 * K2_TOKBUF + 512, ctx->stream, k2);
        else
          TBZ_LAUNCH_DYN_WG(tbz_k2_lz77_dual, n_it, 128, k2.win_bytes + 2 * K2_TOKBUF + 512 + 2 * sizeof(K2Hand),
                            ctx->stream, k2);
      }
#ifdef TBZ_WAVE_TRACE
      if (const char* vp = getenv("TBZ_K2_TRACE")) {
        std::vector<u64> h(8192 * 8);
        hipStreamSynchronize(ctx->stream);
        u32 cn[4] = {0, 0, 0, 0};
        hipMemcpyFromSymbol(cn, HIP_SYMBOL(tbz_dbg_cnt), 16);
        fprintf(stderr, "tbz: K2 resolve since the start: %u batches with matches, %.2f rounds and %.1f matches per batch\n", cn[0],
                cn[0] ? (double)cn[1] / cn[0] : 0.0, cn[0] ? (double)cn[2] / cn[0] : 0.0);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tbz_dbg), h.size() * 8);
        if (FILE* f = fopen(vp, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
#endif
      if (h_glob->n_big) {
        k2.win_bytes = 0;
        k2.cls = h_glob->n_big < n_it ? 2 : 0;
        ctx->tim.k2_kinds |= 4u;
        if (ctx->k2_single) TBZ_LAUNCH(tbz_k2_lz77, n_it, ctx->stream, k2);
        else if (ctx->k2_ring2) TBZ_LAUNCH_WG(tbz_k2_lz77_ring2, n_it, 128, ctx->stream, k2);
        else TBZ_LAUNCH_WG(tbz_k2_lz77_ring3, n_it, 192, ctx->stream, k2);
      }
      TBZ_HIP(hipGetLastError());
    }
    if ((r = record(ctx, 5))) return r;
    TBZ_HIP(hipEventSynchronize(ctx->ev[7]));  // the stream records are here; K2 keeps running
    huff_ms = elapsed(ctx, 2, 3);
    ctx->tim.huff_ms = huff_ms;
    ctx->tim.token_words = h_k3s[n].tok_words;
    ctx->tim.n_segments = ctx->tim.n_groups = h_k3s[n].nonempty;
    for (size_t s = 0; s < n; s++) {
      StreamPlan& S = sp[s];
      const K3Stream& k = h_k3s[s];
      S.total_out = k.total_out;
      S.saw_final = k.last.status == SEG_FINAL;  // else: the last item ran out of input (every other one landed)
      S.trailer0 = k.last.trailer0;
      S.trailer1 = k.last.trailer1;
      S.trailer_have = k.last.trailer_have;
      S.in_end_bit = k.last.end_bit;
      S.boundary_bit = (S.n_items > 1 && !(k.last_start & 7)) ? k.last_start : 0;
      S.boundary_out = k.total_out - k.last.out_bytes;
      if (k.last.status == SEG_UNDERRUN) {
        S.blk_known = true;
        if (k.last.land_marker == 0) {
          S.blk_bit = (uint64_t)k.last.trailer0 | ((uint64_t)k.last.trailer1 << 32);
          S.blk_out = k.total_out - k.last.out_bytes + k.last.reserved;
        } else {
          S.blk_bit = k.last_start;
          S.blk_out = k.total_out - k.last.out_bytes;
        }
        S.hdr_bit = (uint64_t)k.last.trailer0 | ((uint64_t)k.last.trailer1 << 32);
      }
      S.stored_cut = k.last.status == SEG_UNDERRUN && k.last.pad == 2;
      if (k.last.status == SEG_UNDERRUN && k.last.pad == 1 && !(k.last.end_bit & 7)) {
        S.boundary_bit = k.last.end_bit;
        S.boundary_out = k.total_out;
      }
      S.status = S.saw_final ? TBZ_FINISHED : TBZ_INPUT_UNDERRUN;
      S.done = true;
      fill_result(s, S.status, (uint32_t)k.nonempty);
    }
  } else {
  if ((r = fetch_host_tables())) return r;
  std::vector<SegResult> res(n_items);
  TBZ_HIP(hipMemcpyAsync(res.data(), ctx->d_res.p, res.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                         ctx->stream));
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  huff_ms = elapsed(ctx, 2, 3);
  if (wide_beside) {
    bool any = false;
    for (const SegResult& q : res) any = any || q.status == SEG_WIDE;
    if (any) {
      std::vector<SegResult> wr(n_items);
      TBZ_HIP(hipMemcpyAsync(wr.data(), ctx->d_wide_res.p, wr.size() * sizeof(SegResult), hipMemcpyDeviceToHost, ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      for (size_t i = 0; i < n_items; i++)
        if (res[i].status == SEG_WIDE) res[i] = wr[i];
      ctx->tim.huff_launches++;
    }
  }
  if ((r = redo(items, res, false))) return r;

  // ---------------------------------------------------------------- chain walk (+ fix-up rounds)
  std::vector<SegHost> segs;
  segs.reserve(n_items);
  std::vector<std::vector<SegHost>> per_stream(n);
  auto consume = [&](StreamPlan& S, size_t s, const Item& it, const SegResult& q, bool is_fixup) {
    SegHost h;
    h.item = it;
    h.seg.tok = q.tok;
    h.seg.tok_words = q.tok_words;
    h.seg.out_bytes = q.out_bytes;
    h.seg.n_runs = q.n_runs;
    h.seg.run_first = 0;
    h.seg.runs = q.runs;
    h.seg.run0 = q.run0;
    h.stream = (uint32_t)s;
    h.deficit = q.max_deficit;
    h.continues = S.next_continues;
    S.next_continues = false;
    if (ctx->tun.debug2) fprintf(stderr, "consume: start_bit %llu status %d end_bit %llu out %llu tok %llu runs %u deficit %u fixup %d\n", (unsigned long long)it.start_bit, q.status, (unsigned long long)q.end_bit, (unsigned long long)q.out_bytes, (unsigned long long)q.tok_words, q.n_runs, q.max_deficit, (int)is_fixup);
    if (q.tok_words || q.out_bytes) per_stream[s].push_back(h);
    else if (h.continues) S.next_continues = true;  // nothing emitted: carry the flag forward
    S.total_out += q.out_bytes;
    ctx->tim.token_words += q.tok_words;
    if (q.status != SEG_UNDERRUN) {
      ctx->gang_rounds += q.reserved >> 32;
      ctx->gang_valid += q.reserved & 0xffffffffu;
    }
    if (q.status == SEG_LANDED) {
      uint32_t mk = is_fixup ? q.land_marker : 0;
      if (is_fixup) S.cur_item = S.first_item + 1 + (mk - S.first_marker);
      else S.cur_item += 1;
      if (S.cur_item < S.first_item + S.n_items && !(items[S.cur_item].start_bit & 7)) {
        S.boundary_bit = items[S.cur_item].start_bit;  // (octet-aligned boundaries only: that is what in_consumed can say)
        S.boundary_out = S.total_out;
      }
      return;
    }
    if (q.status == SEG_FINAL) {
      S.saw_final = true;
      S.trailer0 = q.trailer0;
      S.trailer1 = q.trailer1;
      S.trailer_have = q.trailer_have;
      S.in_end_bit = q.end_bit;
      S.status = TBZ_FINISHED;
      S.done = true;
      return;
    }
    if (q.status == SEG_OVERSHOOT) {
      S.pending_fixup = true;
      S.fix_start_bit = q.end_bit;
      S.next_continues = true;
      return;
    }
    if (q.status == SEG_UNDERRUN) {
      S.status = TBZ_INPUT_UNDERRUN;
      S.in_end_bit = q.end_bit;
      S.stored_cut = q.pad == 2;
      S.blk_known = true;
      if (q.land_marker == 0) {  // the block in which the input ran out (K1 reports its start and the octets before it)
        S.blk_bit = (uint64_t)q.trailer0 | ((uint64_t)q.trailer1 << 32);
        S.blk_out = S.total_out - q.out_bytes + q.reserved;
      } else {                   // only the item's start is known exactly
        S.blk_bit = it.start_bit;
        S.blk_out = S.total_out - q.out_bytes;
      }
      S.hdr_bit = (uint64_t)q.trailer0 | ((uint64_t)q.trailer1 << 32);
      if (q.pad == 1 && !(q.end_bit & 7)) {  // ran out exactly at an octet-aligned block start
        S.boundary_bit = q.end_bit;
        S.boundary_out = S.total_out;
      }
      S.done = true;
      return;
    }
    S.status = q.status < 0 ? q.status : TBZ_E_INTERNAL;
    S.done = true;
  };
  for (size_t s = 0; s < n; s++) {
    StreamPlan& S = sp[s];
    while (!S.done && !S.pending_fixup) {
      if (S.cur_item >= S.first_item + S.n_items) {  // cannot happen: the last item has no limit
        S.status = TBZ_E_INTERNAL;
        S.done = true;
        break;
      }
      consume(S, s, items[S.cur_item], res[S.cur_item], false);
    }
  }
  for (;;) {
    std::vector<Item> fix;
    std::vector<size_t> fix_stream;
    for (size_t s = 0; s < n; s++)
      if (sp[s].pending_fixup && !sp[s].done) {
        Item it{};
        it.start_bit = sp[s].fix_start_bit;
        it.limit_bit = ~0ull;
        it.end_byte = sp[s].in_off + sp[s].in_len;
        it.stream = (uint32_t)s;
        it.flags = ((uint32_t)format << ITEM_FMT_SHIFT) | ITEM_FIXUP;
        if (resume_abs && s == 0 && it.start_bit == sp[0].in_off * 8 + bit_off) it.flags |= ITEM_RESUME;  // (the block the session resumes in)
        fix.push_back(it);
        fix_stream.push_back(s);
      }
    if (fix.empty()) break;
    ctx->tim.fixup_rounds++;
    if ((r = upload(ctx, ctx->d_items, fix))) return r;
    if ((r = ensure(ctx, ctx->d_res, fix.size() * sizeof(SegResult)))) return r;
    if ((r = record(ctx, 2))) return r;
    {
      // the repair pools span from the earliest repair start to the end of the last repaired stream; they are laid
      // out once (the first round's base stands: later rounds only start further on in the same streams)
      uint64_t lo = ~0ull, hi = 0;
      for (size_t k = 0; k < fix.size(); k++) {
        lo = std::min(lo, fix[k].start_bit);
        hi = std::max(hi, fix[k].end_byte * 8);
      }
      if (ctx->tim.fixup_rounds == 1) {
        pool2_base = lo & ~(uint64_t)((1u << RUN_SHIFT) - 1);
        pool2_hi = 0;
      }
      if (lo < pool2_base) return TBZ_E_INTERNAL;  // (cannot happen: a stream's repairs move forward)
      if (hi > pool2_hi) {
        // growing would move tokens that segments already refer to: size for the whole tail of the call at once
        if (pool2_hi != 0) return TBZ_E_INTERNAL;
        pool2_hi = in_extent * 8;
        const uint64_t bits2 = pool2_hi - pool2_base;
        half2 = (k1_gang(fix.size()) == 1 || ctx->tun.tok_full) ? 0u : 1u;
        if ((r = ensure(ctx, ctx->d_tok2, ((size_t)bits2 >> half2) * 2 + 256))) return r;
        if ((r = ensure(ctx, ctx->d_runs2, ((size_t)bits2 >> RUN_SHIFT) * sizeof(RunRec) + 1024))) return r;
      }
    }
    if ((r = launch_k1((const Item*)ctx->d_items.p, (SegResult*)ctx->d_res.p, fix.size(), true))) return r;
    TBZ_HIP(hipGetLastError());
    if ((r = record(ctx, 3))) return r;
    ctx->tim.huff_launches++;
    std::vector<SegResult> fr(fix.size());
    TBZ_HIP(hipMemcpyAsync(fr.data(), ctx->d_res.p, fr.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                           ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    huff_ms += elapsed(ctx, 2, 3);
    if ((r = redo(fix, fr, true))) return r;
    for (size_t k = 0; k < fix.size(); k++) {
      size_t s = fix_stream[k];
      StreamPlan& S = sp[s];
      S.pending_fixup = false;
      consume(S, s, fix[k], fr[k], true);
      while (!S.done && !S.pending_fixup) {
        if (S.cur_item >= S.first_item + S.n_items) {
          S.status = TBZ_E_INTERNAL;
          S.done = true;
          break;
        }
        consume(S, s, items[S.cur_item], res[S.cur_item], false);
      }
    }
  }
  ctx->tim.huff_ms = huff_ms;
  if (ctx->tun.debug) fprintf(stderr, "tbz: gang rounds %llu, committed lanes %llu (%.2f per round)\n", (unsigned long long)ctx->gang_rounds, (unsigned long long)ctx->gang_valid, ctx->gang_rounds ? (double)ctx->gang_valid / ctx->gang_rounds : 0.0);

  // ---------------------------------------------------------------- distance errors vs output overflow
  // history check (deflate.lisp:343-345): a match may not reach before the stream's first octet.  Every segment
  // here lies before the point where the stream ended / failed, so a front-to-back decoder meets this error
  // before any later one — and before output-overflow iff the offending match STARTS at or before the buffer's
  // end.  K1 reports per item only the largest reach-back; when the buffer ends inside the offending segment
  // that item is decoded once more by the one-lane kernel, told how many octets precede it (ITEM_HIST), which
  // then reports the first offending match's output offset.
  enum { DIST_NONE = 0, DIST_FIRST = 1, DIST_AFTER_OVERFLOW = 2 };
  std::vector<uint8_t> dist_first(n, DIST_NONE);
  uint64_t dist_at = ~0ull;  // (sessions: one stream)
  {
    struct Probe { size_t s, seg; uint64_t before; };
    std::vector<Probe> probes[1];
    for (size_t s = 0; s < n; s++) {
      uint64_t produced = 0;
      const uint64_t hist = s == 0 ? hist_len : 0;  // (a resumed stream: octets of earlier output that are there to copy from)
      for (size_t i = 0; i < per_stream[s].size(); i++) {
        const SegHost& h = per_stream[s][i];
        if (h.deficit && (uint64_t)h.deficit > produced + hist) {
          const uint64_t end = produced + h.seg.out_bytes, cap = sp[s].out_cap;
          if (opt && opt->prefix_on_error) probes[0].push_back({s, i, produced + hist});  // (a session wants the place)
          else if (cap >= end) dist_first[s] = DIST_FIRST;
          else if (cap < produced) dist_first[s] = DIST_AFTER_OVERFLOW;
          else probes[0].push_back({s, i, produced + hist});
          break;
        }
        produced += h.seg.out_bytes;
      }
    }
    if (!probes[0].empty()) {
      auto& pv = probes[0];
      std::vector<Item> its(pv.size());
      for (size_t k = 0; k < pv.size(); k++) {
        its[k] = per_stream[pv[k].s][pv[k].seg].item;
        its[k].flags |= ITEM_PROBE | ((uint32_t)std::min<uint64_t>(pv[k].before, 65535) << ITEM_HIST_SHIFT);  // < 32768: the item's reach-back exceeds it
      }
      if ((r = make_explicit(its))) return r;
      if ((r = upload(ctx, ctx->d_redo_items, its))) return r;
      if ((r = ensure(ctx, ctx->d_redo_res, its.size() * sizeof(SegResult)))) return r;
      if ((r = launch_lane((const Item*)ctx->d_redo_items.p, (SegResult*)ctx->d_redo_res.p, its.size(), false, true))) return r;
      std::vector<SegResult> pr(its.size());
      TBZ_HIP(hipMemcpyAsync(pr.data(), ctx->d_redo_res.p, pr.size() * sizeof(SegResult), hipMemcpyDeviceToHost,
                             ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      ctx->tim.huff_launches++;
      for (size_t k = 0; k < pv.size(); k++) {
        SegHost& h = per_stream[pv[k].s][pv[k].seg];
        // the item's tokens are now the one-lane kernel's (same octets, one run)
        h.seg.tok = pr[k].tok;
        h.seg.tok_words = pr[k].tok_words;
        h.seg.n_runs = pr[k].n_runs;
        h.seg.run0 = pr[k].run0;
        const uint64_t at = pr[k].reserved;  // octets into the item; ~0: none found (cannot happen)
        const uint64_t hist_k = pv[k].s == 0 ? hist_len : 0;
        dist_first[pv[k].s] = (at != ~0ull && pv[k].before - hist_k + at > sp[pv[k].s].out_cap) ? DIST_AFTER_OVERFLOW : DIST_FIRST;
        if (at != ~0ull) dist_at = pv[k].before - hist_k + at;  // octets of the stream before the offending match
      }
    }
  }

  // ---------------------------------------------------------------- per-stream layout, groups, status
  std::vector<Seg> h_segs;
  std::vector<Group> h_groups;
  std::vector<uint8_t> h_hist;      // per group: 1 = H-group (symbolic history)
  std::vector<uint32_t> h_gstream;  // per group: its stream
  std::vector<int32_t> h_grec;      // per group: index of the slice record that describes it (tbz_k3_slice), or -1
  std::vector<BigSeg> bigs;
  uint32_t n_recs = 0;
  // K2's parallelism is its number of groups: segments much larger than a fair share of the call's output are cut
  // into slices at run boundaries on the device (tbz_k3_slice); every slice becomes a group
  uint64_t slice_target = 64u << 10, h_join_below = 48u << 10;
  {
    uint64_t tot = 0;
    for (size_t s = 0; s < n; s++) tot += std::min(sp[s].total_out, sp[s].out_cap);
    slice_target = std::max<uint64_t>(slice_target, tot / 4096);
    if (ctx->tun.slice) slice_target = std::max(1024, ctx->tun.slice);  // (tests force small slices)
    h_join_below = std::min<uint64_t>(std::max<uint64_t>(48u << 10, tot / 2048), 256u << 10);
    if (ctx->tun.h_join) h_join_below = std::max(1024, ctx->tun.h_join);
  }
  for (size_t s = 0; s < n; s++) {
    StreamPlan& S = sp[s];
    tbz_result& R = results[s];
    auto& v = per_stream[s];
    int32_t status = S.status;
    if (dist_first[s] != DIST_NONE) status = TBZ_E_DISTANCE;
    status = fill_result(s, status, (uint32_t)v.size(), dist_first[s] == DIST_FIRST);
    if (status < 0) {
      // reference signals an error: no partial-result contract.  A session hands out what a front-to-back decoder
      // produced before it met the error: everything before the failing token (for a match that reaches before the
      // stream's first octet the one-lane re-decode above found its place)
      if (!(opt && opt->prefix_on_error) || (status == TBZ_E_DISTANCE && dist_at == ~0ull)) continue;
      opt->first_error = status;
      R.out_len = status == TBZ_E_DISTANCE ? dist_at : S.total_out;
      R.out_total = R.out_len;
    }
    if (size_only) continue;
    const uint64_t hist = s == 0 ? hist_len : 0;
    // groups: consecutive segments that share one LZ77 window in one K2 workgroup.  A segment that needs no history
    // opens a group of its own.  One that does (its matches reach before its first octet, or it continues a
    // repaired block) joins the group before it while that group is small; a group whose segments reach before
    // ITS first octet becomes an H-group: decoded against symbolic history, resolved by K6 (so the groups of a
    // stream without flush points run in parallel instead of collapsing into one workgroup).
    uint64_t o = 0;
    S.seg_first = (uint32_t)h_segs.size();
    const size_t g0 = h_groups.size();  // this stream's first group
    // a group keeps taking in history-needing segments below this size: K6's chain grows with the cube root of their number, so
    // groups grow with the call's output while the ring kernel still gets some 2048 of them (x 2 planes)
    const uint64_t H_JOIN_BELOW = h_join_below;
    bool after_big = false;
    for (size_t i = 0; i < v.size(); i++) {
      if (o >= R.out_len) break;
      const bool need = (i > 0 || hist > 0) && (v[i].continues || v[i].deficit > 0);
      // first octet its matches copy from (with earlier output in the buffer it may lie before the stream's own first octet)
      const uint64_t reach = S.out_off + o - std::min<uint64_t>(v[i].deficit, o + hist);
      if (ctx->sym_hist && v[i].seg.out_bytes >= 2 * slice_target && v[i].seg.n_runs >= 2) {
        // a large segment: its slices are laid out by the device, one group each
        BigSeg b{};
        b.seg = v[i].seg;
        b.out_abs = S.out_off + o;
        b.out_end = S.out_off + R.out_len;
        b.target = slice_target;
        b.seg_slot = (uint32_t)h_segs.size();
        b.group_slot = (uint32_t)h_groups.size();
        b.rec_slot = n_recs;
        b.n_slots = (uint32_t)(v[i].seg.out_bytes / slice_target + 1);
        b.first_hist = (need && reach < b.out_abs && (h_groups.size() > g0 || hist > 0)) ? 1u : 0u;
        bigs.push_back(b);
        for (uint32_t k = 0; k < b.n_slots; k++) {
          Group g{};
          g.out_abs = b.out_abs;
          g.out_end = b.out_end;
          g.seg_first = b.seg_slot + k;
          h_groups.push_back(g);
          h_hist.push_back(0);
          h_grec.push_back((int32_t)(n_recs + k));
          h_segs.push_back(Seg{});
        }
        n_recs += b.n_slots;
        o += v[i].seg.out_bytes;
        after_big = true;
        continue;
      }
      bool join = need && !after_big && h_groups.size() > g0;
      if (ctx->sym_hist && join) join = (S.out_off + o) - h_groups.back().out_abs < H_JOIN_BELOW;
      after_big = false;
      if (!join) {
        Group g;
        g.out_abs = S.out_off + o;
        g.out_end = S.out_off + R.out_len;
        g.seg_first = (uint32_t)h_segs.size();
        g.seg_count = 0;
        h_groups.push_back(g);
        h_hist.push_back(0);
        h_grec.push_back(-1);
      }
      if (need && reach < h_groups.back().out_abs) {
        if (ctx->sym_hist && (h_groups.size() > g0 + 1 || hist > 0)) {
          h_hist.back() = 1;
        } else {
          // (round-1 scheme, and always for a stream's first group) take in earlier groups until the history is covered
          while (h_groups.size() > g0 + 1 && h_groups.back().out_abs > reach) {
            const uint32_t cnt = h_groups.back().seg_count;
            const uint8_t hh = h_hist.back();
            h_groups.pop_back();
            h_hist.pop_back();
            h_grec.pop_back();
            h_groups.back().seg_count += cnt;
            h_hist.back() |= hh;
          }
        }
      }
      h_groups.back().seg_count++;
      h_segs.push_back(v[i].seg);
      o += v[i].seg.out_bytes;
    }
    S.seg_count = (uint32_t)h_segs.size() - S.seg_first;
    for (size_t gi = g0; gi < h_groups.size(); gi++) h_gstream.push_back((uint32_t)s);
  }
  ctx->tim.n_segments = h_segs.size();
  ctx->tim.n_groups = h_groups.size();

  // ---------------------------------------------------------------- K2
  if ((r = record(ctx, 4))) return r;
  if (!size_only && opt && opt->alloc) {
    d_out = opt->alloc(sp[0].total_out);
    if (!d_out) return TBZ_E_NOMEM;
  }
  if (!size_only && !h_groups.empty()) {
    if (!d_out) return TBZ_E_ARG;
    if ((r = upload(ctx, ctx->d_segs, h_segs))) return r;
    if ((r = upload(ctx, ctx->d_groups, h_groups))) return r;
    std::vector<SliceRec> recs(n_recs);
    if (!bigs.empty()) {
      if ((r = upload(ctx, ctx->d_bigs, bigs))) return r;
      if ((r = ensure(ctx, ctx->d_recs, (size_t)n_recs * sizeof(SliceRec)))) return r;
      K3sParams ks{(const BigSeg*)ctx->d_bigs.p,
                   (Seg*)ctx->d_segs.p, (Group*)ctx->d_groups.p, (SliceRec*)ctx->d_recs.p, (u32)bigs.size()};
      TBZ_LAUNCH(tbz_k3_slice, bigs.size(), ctx->stream, ks);
      TBZ_HIP(hipMemcpyAsync(recs.data(), ctx->d_recs.p, (size_t)n_recs * sizeof(SliceRec), hipMemcpyDeviceToHost, ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      for (size_t gi = 0; gi < h_groups.size(); gi++)
        if (h_grec[gi] >= 0) {
          const SliceRec& q = recs[h_grec[gi]];
          h_groups[gi].out_abs = q.out_abs;
          h_groups[gi].seg_count = q.used ? 1u : 0u;
          h_hist[gi] = (uint8_t)(q.used && q.hist);
        }
    }
    // groups whose whole output fits a linear LDS window (the common case: flush-delimited segments)
    // run with dynamic LDS sized to the largest of them; the rest take the 32 KiB-history ring kernel;
    // H-groups take the ring kernel twice (octet plane, pointer plane) and K6 afterwards
    std::vector<uint32_t> order_small, order_big, order_h;
    std::vector<HGroupSpan> hgs;
    uint64_t max_small = 0, mark_lo = ~0ull, mark_hi = 0;
    for (size_t gi = 0; gi < h_groups.size(); gi++) {
      uint64_t tot = 0;
      if (h_grec[gi] >= 0) {
        if (!recs[h_grec[gi]].used) continue;  // an empty slot: no workgroup at all
        tot = recs[h_grec[gi]].out_bytes;
      } else {
        for (uint32_t k = 0; k < h_groups[gi].seg_count; k++) tot += h_segs[h_groups[gi].seg_first + k].out_bytes;
      }
      if (h_hist[gi]) {
        const StreamPlan& S = sp[h_gstream[gi]];
        order_h.push_back((uint32_t)gi);
        hgs.push_back(HGroupSpan{h_groups[gi].out_abs, std::max(h_groups[gi].out_abs, std::min(h_groups[gi].out_abs + tot, h_groups[gi].out_end)),
                         S.out_off - (h_gstream[gi] == 0 ? hist_len : 0), h_gstream[gi]});
        mark_lo = std::min(mark_lo, S.out_off);
        mark_hi = std::max(mark_hi, h_groups[gi].out_end);
      } else if (tot + K2_SLACK <= K2_SMALL_MAX) {
        order_small.push_back((uint32_t)gi);
        max_small = std::max(max_small, tot);
      } else {
        order_big.push_back((uint32_t)gi);
      }
    }
    std::vector<uint32_t> order(order_small);
    order.insert(order.end(), order_big.begin(), order_big.end());
    order.insert(order.end(), order_h.begin(), order_h.end());
    ctx->tim.n_groups = order.size();
    if ((r = upload(ctx, ctx->d_order, order))) return r;
    K2Params k2{(const Seg*)ctx->d_segs.p, (const Group*)ctx->d_groups.p, (const u32*)ctx->d_order.p, (const u8*)d_in, (u8*)d_out, 0, 0, 0, nullptr, nullptr,
                0, 0, 0, 0, 0, nullptr, 0};
    if (!order_small.empty()) {
      k2.n_groups = (u32)order_small.size();
      k2.win_bytes = (u32)((max_small + K2_SLACK + 63) & ~63ull);
      ctx->tim.k2_kinds |= ctx->k2_single ? 2u : 1u;
      if (ctx->k2_single)
        TBZ_LAUNCH_DYN(tbz_k2_lz77_small, order_small.size(), k2.win_bytes + 2 * K2_TOKBUF + 512, ctx->stream, k2);
      else
        TBZ_LAUNCH_DYN_WG(tbz_k2_lz77_dual, order_small.size(), 128,
                          k2.win_bytes + 2 * K2_TOKBUF + 512 + 2 * sizeof(K2Hand), ctx->stream, k2);
#ifdef TBZ_WAVE_TRACE
      if (const char* vp = getenv("TBZ_K2_TRACE")) {
        std::vector<u64> h(8192 * 8);
        hipStreamSynchronize(ctx->stream);
        u32 cn[4] = {0, 0, 0, 0};
        hipMemcpyFromSymbol(cn, HIP_SYMBOL(tbz_dbg_cnt), 16);
        fprintf(stderr, "tbz: K2 resolve since the start: %u batches with matches, %.2f rounds and %.1f matches per batch\n", cn[0],
                cn[0] ? (double)cn[1] / cn[0] : 0.0, cn[0] ? (double)cn[2] / cn[0] : 0.0);
        hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tbz_dbg), h.size() * 8);
        if (FILE* f = fopen(vp, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
      }
#endif
    }
    if (!order_big.empty() && order_h.empty()) {
      k2.order = (const u32*)ctx->d_order.p + order_small.size();
      k2.n_groups = (u32)order_big.size();
      k2.win_bytes = 0;
      ctx->tim.k2_kinds |= 4u;
      if (ctx->k2_single) TBZ_LAUNCH(tbz_k2_lz77, order_big.size(), ctx->stream, k2);
      else if (ctx->k2_ring2) TBZ_LAUNCH_WG(tbz_k2_lz77_ring2, order_big.size(), 128, ctx->stream, k2);
        else TBZ_LAUNCH_WG(tbz_k2_lz77_ring3, order_big.size(), 192, ctx->stream, k2);
    }
    if (!order_h.empty()) {
      if ((r = hgroups_launch(ctx, hgs, k2, order_small.size(), order_big.size(), order_h.size(), d_out, mark_lo, mark_hi))) return r;
      have_resolve = true;
    }
    TBZ_HIP(hipGetLastError());
  }
  if ((r = record(ctx, 5))) return r;
  }  // !simple

  // ---------------------------------------------------------------- K4 / K5 + trailer compare
  if (!size_only && format != TBZ_FORMAT_DEFLATE) {
    std::vector<uint64_t> co(n), cl(n);
    std::vector<uint32_t> init(n), sums;
    for (size_t s = 0; s < n; s++) {
      co[s] = sp[s].out_off;
      // update-checksum runs when finished or output-overflow (zlib.lisp:136-137, gzip.lisp:268-269)
      bool want = results[s].status == TBZ_FINISHED || results[s].status == TBZ_OUTPUT_OVERFLOW;
      cl[s] = want ? results[s].out_len : 0;
      init[s] = format == TBZ_FORMAT_ZLIB ? 1u : 0u;  // s1=1,s2=0 (zlib.lisp:11-12) / crc 0 (gzip.lisp:28)
    }
    if (fused_adler) {
      std::vector<uint32_t> fg(n), ng(n);
      for (size_t s = 0; s < n; s++) {
        fg[s] = sp[s].first_item;
        ng[s] = sp[s].n_items;
      }
      if ((r = run_adler_groups(ctx, fg, ng, init, sums))) return r;
    } else if ((r = run_checksums(ctx, format == TBZ_FORMAT_ZLIB ? 1 : 2, d_out, co, cl, init, sums))) {
      return r;
    }
    for (size_t s = 0; s < n; s++) {
      tbz_result& R = results[s];
      if (cl[s] == 0) sums[s] = init[s];  // (the fused partials cover whatever K2 wrote)
      if (format == TBZ_FORMAT_ZLIB) R.adler32 = sums[s];
      else R.crc32 = sums[s];
    }
  } else {
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
  }
  if ((r = record(ctx, 6))) return r;
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  const bool keep_prefix = opt && opt->prefix_on_error;
  if (opt) {
    opt->blk_known = sp[0].blk_known;
    opt->blk_bit = sp[0].blk_bit > sp[0].in_off * 8 ? sp[0].blk_bit - sp[0].in_off * 8 : 0;
    opt->blk_out = sp[0].blk_out;
    opt->hdr_bit = sp[0].hdr_bit > sp[0].in_off * 8 ? sp[0].hdr_bit - sp[0].in_off * 8 : 0;
    opt->tok_bit = sp[0].in_end_bit > sp[0].in_off * 8 ? sp[0].in_end_bit - sp[0].in_off * 8 : 0;
    opt->end_bit = sp[0].in_end_bit > sp[0].in_off * 8 ? sp[0].in_end_bit - sp[0].in_off * 8 : 0;
  }
  for (size_t s = 0; s < n; s++) {
    tbz_result& R = results[s];
    StreamPlan& S = sp[s];
    if (R.status != TBZ_FINISHED || format == TBZ_FORMAT_DEFLATE) continue;
    if (format == TBZ_FORMAT_ZLIB) {
      if (S.trailer_have < 2) R.status = TBZ_INPUT_UNDERRUN;  // zlib.lisp:81-86
      else if (size_only) R.flags |= 0;
      else if (S.trailer0 != R.adler32) { R.status = TBZ_E_ADLER32; R.out_len = keep_prefix ? R.out_len : 0; }
      else R.flags |= 1;
    } else {
      if (S.trailer_have < 1) R.status = TBZ_INPUT_UNDERRUN;  // gzip.lisp:83-86
      else if (!size_only && S.trailer0 != R.crc32) { R.status = TBZ_E_CRC32; R.out_len = keep_prefix ? R.out_len : 0; }
      else if (S.trailer_have < 2) R.status = TBZ_INPUT_UNDERRUN;  // gzip.lisp:96-99
      else if (!size_only) R.flags |= 1;
    }
  }
  ctx->tim.scan_ms = elapsed(ctx, 0, 1);
  ctx->tim.find_ms = have_find ? elapsed(ctx, 8, 9) : 0.f;
  ctx->tim.resolve_ms = have_resolve ? elapsed(ctx, 10, 11) : 0.f;
  ctx->tim.lz_ms = elapsed(ctx, 4, 5) - ctx->tim.resolve_ms;
  ctx->tim.cksum_ms = elapsed(ctx, 5, 6);
  ctx->tim.total_ms = elapsed(ctx, 0, 6);
  ctx->tim.scratch_bytes = scratch_total(ctx);
  if (ctx->tun.debug && !ctx->tim.fixup_rounds)
    fprintf(stderr, "tbz: ms scan %.3f | items+upload %.3f | huff %.3f | results+walk+layout %.3f | lz %.3f | cksum %.3f\n",
            elapsed(ctx, 0, 1), elapsed(ctx, 1, 2), elapsed(ctx, 2, 3), elapsed(ctx, 3, 4), elapsed(ctx, 4, 5),
            elapsed(ctx, 5, 6));
  return 0;
}

}  // namespace tbz

// ====================================================================================================
// C ABI
// ====================================================================================================
extern "C" {

int tbz_abi_version(void) { return TBZ_ABI_VERSION; }

const char* tbz_strerror(int code) {
  switch (code) {
    case TBZ_FINISHED: return "finished";
    case TBZ_INPUT_UNDERRUN: return "input underrun";
    case TBZ_OUTPUT_OVERFLOW: return "output overflow";
    case TBZ_E_BTYPE: return "reserved block type 3";
    case TBZ_E_STORED_LEN: return "stored block LEN/NLEN mismatch";
    case TBZ_E_OVERSUBSCRIBED: return "too many entries in huffman table";
    case TBZ_E_INCOMPLETE: return "incomplete huffman table";
    case TBZ_E_REPEAT_NO_PREV: return "tried to repeat length without previous length";
    case TBZ_E_REPEAT_OVERRUN: return "code-length repeat runs past HLIT+HDIST";
    case TBZ_E_INVALID_CODE: return "invalid huffman code";
    case TBZ_E_DISTANCE: return "distance reaches before start of output (no window)";
    case TBZ_E_ZLIB_HEADER: return "invalid zlib header";
    case TBZ_E_ZLIB_DICT: return "preset dictionary not supported yet";
    case TBZ_E_ADLER32: return "adler32 mismatch";
    case TBZ_E_GZIP_MAGIC: return "bad gzip magic";
    case TBZ_E_GZIP_METHOD: return "unknown gzip compression method";
    case TBZ_E_GZIP_FLAGS: return "reserved gzip flag bits set";
    case TBZ_E_GZIP_HCRC: return "gzip header crc mismatch";
    case TBZ_E_CRC32: return "crc32 mismatch";
    case TBZ_E_ARG: return "bad argument";
    case TBZ_E_HIP: return "HIP runtime error";
    case TBZ_E_NOMEM: return "out of device memory";
    case TBZ_E_NO_DEVICE: return "no HIP device";
    case TBZ_E_UNSUPPORTED: return "unsupported on the device path";
    case TBZ_E_INTERNAL: return "internal error";
  }
  return "unknown";
}

int tbz_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int tbz_ctx_create(int device_id, tbz_ctx** out_ctx) {
  if (!out_ctx) return TBZ_E_ARG;
  *out_ctx = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TBZ_E_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return TBZ_E_ARG;
  tbz_ctx* ctx = new tbz_ctx();
  ctx->device = device_id;
  auto fail = [&](hipError_t e, const char* what) {
    fprintf(stderr, "tbz_ctx_create: %s: %s\n", what, hipGetErrorString(e));
    delete ctx;
    return TBZ_E_HIP;
  };
  hipError_t e;
  if ((e = hipSetDevice(device_id)) != hipSuccess) return fail(e, "hipSetDevice");
  if ((e = hipStreamCreate(&ctx->stream)) != hipSuccess) return fail(e, "hipStreamCreate");
  if ((e = hipStreamCreate(&ctx->stream2)) != hipSuccess) return fail(e, "hipStreamCreate");
  for (auto& ev : ctx->ev)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "hipEventCreate");
  for (auto& ev : ctx->evw)
    if ((e = hipEventCreate(&ev)) != hipSuccess) return fail(e, "hipEventCreate");
  std::vector<uint32_t> t;
  tbz::build_crc_tables(t);
  if (tbz::ensure(ctx, ctx->d_crc_tab, t.size() * 4)) {
    delete ctx;
    return TBZ_E_NOMEM;
  }
  if ((e = hipMemcpy(ctx->d_crc_tab.p, t.data(), t.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
    return fail(e, "hipMemcpy");
  {
    tbz_ctx::Tun& t = ctx->tun;
    if (const char* m = getenv("TBZ_FIND_ENOUGH")) t.find_enough = atol(m);
    if (const char* m = getenv("TBZ_FIND_MIN_LEN")) t.find_min_len = atol(m);
    if (const char* m = getenv("TBZ_K0B_PAIR")) t.k0b_pair = atoi(m) != 0;
    if (const char* m = getenv("TBZ_SUB_MIN")) t.sub_min = atoi(m);
    if (const char* m = getenv("TBZ_OVL")) t.ovl = atoi(m);
    if (const char* m = getenv("TBZ_WIDE_BITS")) t.wide_bits = atol(m);
    if (const char* m = getenv("TBZ_SLICE")) t.slice = atoi(m);
    if (const char* m = getenv("TBZ_H_JOIN")) t.h_join = atoi(m);
    if (const char* m = getenv("TBZ_K6_BLOCK")) t.k6_block = atoi(m);
    if (const char* m = getenv("TBZ_K0C_MAX_BLOCK")) t.k0c_max_block = atol(m);
    t.k6_two_levels = getenv("TBZ_K6_TWO_LEVELS") != nullptr;
    t.no_fused_adler = getenv("TBZ_NO_FUSED_ADLER") != nullptr;
    t.tok_full = getenv("TBZ_TOK_FULL") != nullptr;
    if (const char* m = getenv("TBZ_K6_LDS_MIN")) t.k6_lds_min = atol(m);
    t.debug = getenv("TBZ_DEBUG") != nullptr;
    t.debug2 = getenv("TBZ_DEBUG2") != nullptr;
    if (const char* m = getenv("TBZ_DEBUG_CANDS")) t.debug_cands = m;
  }
  if (const char* m = getenv("TBZ_COPY_THREADS")) ctx->copy_threads = std::max(1, std::min(64, atoi(m)));
  if (const char* m = getenv("TBZ_STAGE_CHUNK_KIB")) ctx->stage_chunk = (size_t)std::max(64, atoi(m)) << 10;
  if (const char* m = getenv("TBZ_SMALL_FUSED")) ctx->small_fused = m[0] != '0';
  if (const char* m = getenv("TBZ_SMALL_MAX_KIB")) ctx->small_max_in = (size_t)std::max(0, atoi(m)) << 10;
  if (const char* m = getenv("TBZ_PIPE_MIN_KIB")) ctx->pipe_min = (size_t)std::max(0, atoi(m)) << 10;
  if (const char* m = getenv("TBZ_PIPE_PART_KIB")) ctx->pipe_part = (size_t)std::max(16, atoi(m)) << 10;
  ctx->copy_threads = std::min<int>(ctx->copy_threads, std::max(1u, std::thread::hardware_concurrency()));
  if (const char* m = getenv("TBZ_HOST_LAYOUT")) ctx->host_layout = m[0] == '1';
  if (const char* m = getenv("TBZ_K2_MODE")) ctx->k2_single = !strcmp(m, "single");
  if (const char* m = getenv("TBZ_K2_RING")) ctx->k2_ring2 = !strcmp(m, "2");
  if (const char* m = getenv("TBZ_K1H")) ctx->k1h = m[0] != '0';
  if (const char* m = getenv("TBZ_HIST")) ctx->sym_hist = strcmp(m, "off") != 0;
  if (const char* m = getenv("TBZ_POOL_CAP_MIB")) ctx->pool_cap = (uint64_t)std::max(1, atoi(m)) << 20;
  if (const char* m = getenv("TBZ_FIND")) ctx->find_mode = !strcmp(m, "off") ? 0 : !strcmp(m, "always") ? 2 : 1;
  if (const char* m = getenv("TBZ_K1_MODE")) {
    if (!strcmp(m, "lane")) ctx->k1_mode = 1;
    else if (!strncmp(m, "gang", 4)) {
      int g = atoi(m + 4);
      if (g == 8 || g == 16 || g == 32 || g == 64) ctx->k1_mode = g;
    }
  }
  *out_ctx = ctx;
  return 0;
}

void tbz_ctx_destroy(tbz_ctx* ctx) {
  if (!ctx) return;
  hipSetDevice(ctx->device);
  if (ctx->stream) hipStreamSynchronize(ctx->stream);
  for (auto* b : tbz::all_pools(ctx))
    if (b->p) hipFree(b->p);
  for (auto& b : ctx->dense)
    if (b.p) hipFree(b.p);
  if (ctx->h_pin) hipHostFree(ctx->h_pin);
  delete ctx->copy_pool;
  delete ctx->copy_pool2;
  if (ctx->stream_in) hipStreamDestroy(ctx->stream_in);
  if (ctx->stream_out) hipStreamDestroy(ctx->stream_out);
  for (int k = 0; k < 4; k++) {
    if (ctx->h_stage[k]) hipHostFree(ctx->h_stage[k]);
    if (ctx->ev_stage[k]) hipEventDestroy(ctx->ev_stage[k]);
  }
  for (auto& ev : ctx->ev)
    if (ev) hipEventDestroy(ev);
  for (auto& ev : ctx->evw)
    if (ev) hipEventDestroy(ev);
  if (ctx->stream2) hipStreamDestroy(ctx->stream2);
  if (ctx->stream) hipStreamDestroy(ctx->stream);
  delete ctx;
}

const char* tbz_last_error(const tbz_ctx* ctx) { return ctx ? ctx->err.c_str() : "no context"; }

int tbz_last_timings(const tbz_ctx* ctx, tbz_timings* out) {
  if (!ctx || !out) return TBZ_E_ARG;
  *out = ctx->tim;
  return 0;
}

// the pipeline over a batch, in as many passes over consecutive streams as the pool cap asks for (scratch is
// 9 octets per input octet of a pass's extent; one stream is never split: its segments share one token pool)
static int inflate_passes(tbz_ctx* ctx, int format, size_t n, const void* d_in, const uint64_t* in_offs,
                          const uint64_t* in_lens, void* d_out, const uint64_t* out_offs, const uint64_t* out_caps,
                          tbz_result* results, bool size_only) {
  if (!ctx || (n && (!in_offs || !in_lens))) return TBZ_E_ARG;
  auto need = [&](uint64_t lo, uint64_t hi) { return (hi - lo) * 9; };
  uint64_t lo = ~0ull, hi = 0;
  for (size_t s = 0; s < n; s++)
    if (in_lens[s]) {
      lo = std::min(lo, in_offs[s]);
      hi = std::max(hi, in_offs[s] + in_lens[s]);
    }
  if (n <= 1 || hi <= lo || need(lo, hi) <= ctx->pool_cap)
    return tbz::inflate_core(ctx, format, n, d_in, in_offs, in_lens, d_out, out_offs, out_caps, results, size_only);
  tbz_timings acc{};
  size_t passes = 0;
  for (size_t a = 0; a < n;) {
    size_t b = a;
    uint64_t plo = ~0ull, phi = 0;
    while (b < n) {
      const uint64_t l2 = in_lens[b] ? std::min(plo, in_offs[b]) : plo, h2 = in_lens[b] ? std::max(phi, in_offs[b] + in_lens[b]) : phi;
      if (b > a && h2 > l2 && need(l2, h2) > ctx->pool_cap) break;
      plo = l2;
      phi = h2;
      b++;
    }
    int r = tbz::inflate_core(ctx, format, b - a, d_in, in_offs + a, in_lens + a, d_out, size_only ? nullptr : out_offs + a,
                              size_only ? nullptr : out_caps + a, results + a, size_only);
    if (r) return r;
    const tbz_timings& t = ctx->tim;
    acc.scan_ms += t.scan_ms; acc.huff_ms += t.huff_ms; acc.lz_ms += t.lz_ms; acc.cksum_ms += t.cksum_ms;
    acc.total_ms += t.total_ms; acc.find_ms += t.find_ms; acc.resolve_ms += t.resolve_ms;
    acc.huff_launches += t.huff_launches; acc.fixup_rounds += t.fixup_rounds; acc.token_words += t.token_words;
    acc.n_segments += t.n_segments; acc.n_groups += t.n_groups; acc.n_candidates += t.n_candidates;
    acc.n_hgroups += t.n_hgroups; acc.k1_gang = t.k1_gang; acc.k2_kinds |= t.k2_kinds;
    acc.scratch_bytes = std::max(acc.scratch_bytes, t.scratch_bytes);
    passes++;
    a = b;
  }
  acc.passes = (uint32_t)passes;
  ctx->tim = acc;
  return 0;
}

int tbz_inflate_batch_device(tbz_ctx* ctx, int format, size_t n, const void* d_in_base, const uint64_t* in_offs,
                             const uint64_t* in_lens, void* d_out_base, const uint64_t* out_offs,
                             const uint64_t* out_caps, tbz_result* results) {
  if (!ctx) return TBZ_E_ARG;
  return inflate_passes(ctx, format, n, d_in_base, in_offs, in_lens, d_out_base, out_offs, out_caps, results, false);
}

}  // extern "C"

// ====================================================================================================
// Sessions: 3bz's chunked protocol (deflate.lisp:114-137, :263-269, :705-716; api.lisp:3-21) with the state on the
// device.  A deflate-state in the reference is resumable at every bit and octet; here the resume point is the TOKEN
// in which the input ran out (the block's header is parsed again — a block's tables are the only state besides the 32
// KiB window, deflate.lisp:518-528 — and the token loop entered there), so a call costs O(new input + one block
// header), and what the caller sees — flags, counts, octets, call by call — is what the reference returns:
//   * input the caller has given and the decoder has not finished with stays in HBM (d_in, from the resume block on);
//   * the 32 KiB of output before the resume point are the window (d_hist); K2 / K6 copy from it as from any history;
//   * octets decoded beyond what the caller's buffer takes wait in HBM (d_dec) for the next buffer (output-overflow);
//   * a stream that turns out to be invalid hands out everything a front-to-back decoder would have produced first.
// ====================================================================================================
struct tbz_session {
  tbz_ctx* ctx = nullptr;
  int format = 0;
  tbz::DevBuf d_in, d_dec, d_hist, d_tmp, d_hdr, d_cat;
  // ---- input.  The stream's octets from the resume point on live at d_in[in_off, in_off + in_len): a grow-only buffer
  // whose front is given up by moving in_off (no allocation, copy or synchronisation per call).
  size_t in_off = 0, in_len = 0;
  uint64_t in_abs = 0;        // how many octets of the stream precede d_in[in_off]
  // ---- resume point.  A block start (tok_bit = 0: the header at bit bit_off of the first octet is where decoding is
  // entered, deflate.lisp:518-528) or a TOKEN inside a block: the header is parsed again and the token loop entered at
  // tok_bit — what the reference gets by pushing the bits of an unfinished symbol back (deflate.lisp:399-427).  Far
  // into a long block the octets between the header and the token are not kept: the header's SESSION_HB octets are set
  // aside in d_hdr (hdr_on) and the engine decodes d_hdr ++ d_in, entering the token loop behind the seam.
  uint32_t bit_off = 0;       // the resume block's header starts at this bit of its first octet (d_in[in_off], or d_hdr[0])
  uint64_t tok_bit = 0;       // != 0: resume at this bit (from d_in[in_off]'s first bit; with hdr_on it is < 8)
  bool hdr_on = false;
  uint64_t hdr_abs = 0;       // hdr_on: how many octets of the stream precede d_hdr[0]
  bool header_done = false;   // the container header is behind the resume point: what is left are raw blocks
  uint64_t hist_len = 0;      // octets in d_hist (the output just before the resume point), at most 32768
  uint64_t out_abs = 0;       // octets of output before the resume point
  uint32_t ck = 0;            // adler32 (s1 | s2 << 16) / crc32 of those octets
  // ---- the decode of the input as it stands (valid until more input arrives)
  bool dec_valid = false;
  bool sliced = false;        // it covered a first part of the input only (decode-ahead is bounded: SESSION_AHEAD)
  uint64_t dec_len = 0;       // octets decoded from the resume point on (at d_dec + 32768)
  uint64_t delivered = 0;     // ... of which the caller has these
  int32_t dec_status = 0;     // finished / input-underrun / the error met after dec_len octets
  uint32_t dec_flags = 0;
  bool blk_known = false;
  uint64_t blk_abs_bit = 0, blk_out = 0;  // block-granular resume point (absolute bit in the stream) and the output before it
  uint64_t hdr_abs_bit = 0, tok_abs_bit = 0, end_abs_bit = 0;  // absolute bits: header of the block the input ran out in,
                                                               // the token it ran out in, the end of the final block
  uint32_t total_ck = 0, trailer_check = 0, trailer_isize = 0;
  uint64_t consumed_abs = 0;  // finished: octets of the stream consumed, trailer included
  bool finished = false;
  int32_t error = 0;          // raised once (the reference signals a condition; the state is dead afterwards)
  // The reference allocates its window (32 KiB of zeros) at the first output-overflow (deflate.lisp:121-137) and from
  // then on copies from it whatever a distance says, without asking whether the stream ever produced that octet
  // (deflate.lisp:343-352: "no window?" is the only check).  A damaged stream whose match reaches before its first
  // octet is therefore an error before the first overflow and reads zeros after it; so does the session:
  bool window = false;        // an output-overflow has been reported
  bool pad_hist = false;      // the history is the full 32 KiB, zeros in front
  // ---- measurement (tbz_session_stats)
  uint64_t n_decodes = 0, in_decoded = 0;  // engine calls made, input octets handed to them (re-decoded ones included)
};

namespace tbz {
constexpr uint64_t SESSION_HB = 320;            // octets that hold any block header (17 + 57 + 316 * 7 bits = 286 octets)
constexpr uint64_t SESSION_AHEAD = 8ull << 20;  // input octets one decode takes at most: what a call materialises is bounded
static uint32_t session_ck_init(int format) { return format == TBZ_FORMAT_ZLIB ? 1u : 0u; }

// checksum of d[0, n) continuing from `init`
static int session_chain_ck(tbz_session* S, const void* d, uint64_t n, uint32_t init, uint32_t* out) {
  *out = init;
  if (S->format == TBZ_FORMAT_DEFLATE || n == 0) return 0;
  std::vector<uint64_t> o{0}, l{n};
  std::vector<uint32_t> in{init}, sums;
  int r = run_checksums(S->ctx, S->format == TBZ_FORMAT_ZLIB ? 1 : 2, d, o, l, in, sums);
  if (r) return r;
  *out = sums[0];
  return 0;
}

static int session_decode(tbz_session* S) {
  tbz_ctx* ctx = S->ctx;
  int r;
  CoreOpts opt;
  opt.start_bit_off = S->bit_off;
  opt.hist_len = S->pad_hist ? 32768 : S->hist_len;
  opt.prefix_on_error = true;
  int alloc_err = 0;
  opt.alloc = [&](uint64_t total) -> void* {
    if ((alloc_err = ensure(ctx, S->d_dec, 32768 + total + 128))) return nullptr;
    if (S->pad_hist && S->hist_len < 32768 &&
        hipMemsetAsync(S->d_dec.p, 0, 32768 - S->hist_len, ctx->stream) != hipSuccess) {
      alloc_err = TBZ_E_HIP;
      return nullptr;
    }
    if (S->hist_len &&
        hipMemcpyAsync((uint8_t*)S->d_dec.p + 32768 - S->hist_len, S->d_hist.p, S->hist_len, hipMemcpyDeviceToDevice,
                       ctx->stream) != hipSuccess) {
      alloc_err = TBZ_E_HIP;
      return nullptr;
    }
    return S->d_dec.p;
  };
  const int fmt = S->header_done ? TBZ_FORMAT_DEFLATE : S->format;
  if ((r = ensure(ctx, S->d_in, 64))) return r;  // (an empty first call: there is a buffer to point at)
  // what the engine sees: the input as it stands (a first part of it), behind the set-aside header when there is one
  const uint64_t take = std::min<uint64_t>(S->in_len, SESSION_AHEAD);
  S->sliced = take < S->in_len;
  const uint8_t* tail = (const uint8_t*)S->d_in.p + S->in_off;
  const void* d_src = tail;
  uint64_t seam = 0;  // octets of header in front of the tail
  if (S->hdr_on) {
    seam = SESSION_HB;
    if ((r = ensure(ctx, S->d_cat, seam + take + 64))) return r;
    TBZ_HIP(hipMemcpyAsync(S->d_cat.p, S->d_hdr.p, seam, hipMemcpyDeviceToDevice, ctx->stream));
    if (take) TBZ_HIP(hipMemcpyAsync((uint8_t*)S->d_cat.p + seam, tail, take, hipMemcpyDeviceToDevice, ctx->stream));
    d_src = S->d_cat.p;
  }
  opt.resume_tok_bit = S->tok_bit ? seam * 8 + S->tok_bit : 0;
  uint64_t io = 0, il = seam + take, oo = 32768, oc = 1ull << 62;
  tbz_result R;
  r = inflate_core(ctx, fmt, 1, d_src, &io, &il, nullptr, &oo, &oc, &R, false, &opt);
  if (r) return alloc_err ? alloc_err : r;
  S->n_decodes++;
  S->in_decoded += il;
  // positions of the engine's input -> absolute bits of the stream
  auto abs_bit = [&](uint64_t p) { return p < seam * 8 ? S->hdr_abs * 8 + p : S->in_abs * 8 + (p - seam * 8); };
  S->dec_valid = true;
  S->dec_len = R.out_total;
  S->dec_status = R.status;
  S->dec_flags = R.flags;
  S->blk_known = opt.blk_known;
  S->blk_abs_bit = abs_bit(opt.blk_bit);
  S->blk_out = opt.blk_out;
  S->hdr_abs_bit = abs_bit(opt.hdr_bit);
  S->tok_abs_bit = abs_bit(opt.tok_bit);
  S->end_abs_bit = abs_bit(opt.end_bit);
  S->trailer_check = R.trailer_check;
  S->trailer_isize = R.trailer_isize;
  if (R.status == TBZ_E_DISTANCE && S->dec_len < S->delivered) S->dec_len = S->delivered;  // (no place found: nothing new)
  if (S->dec_len < S->delivered) return TBZ_E_INTERNAL;       // a longer input cannot decode to less
  const void* d_new = (const uint8_t*)S->d_dec.p + 32768;
  if (R.status == TBZ_FINISHED) {
    S->consumed_abs = abs_bit(R.in_consumed * 8) / 8;
    if (fmt == TBZ_FORMAT_DEFLATE && S->format != TBZ_FORMAT_DEFLATE) {
      // raw blocks were decoded: the container's trailer is read here, as zlib.lisp:80-95 / gzip.lisp:78-106 do
      // (the checksum of ALL output = the resume point's, continued over these octets)
      if ((r = session_chain_ck(S, d_new, S->dec_len, S->ck, &S->total_ck))) return r;
      const uint64_t at = S->end_abs_bit / 8 - S->in_abs;  // (the final block ends in the tail, behind any seam)
      const uint64_t have = S->in_len > at ? S->in_len - at : 0;
      uint8_t tr[8] = {0};
      if (have) TBZ_HIP(hipMemcpy(tr, tail + at, std::min<uint64_t>(have, 8), hipMemcpyDeviceToHost));
      S->dec_flags |= 2;
      if (S->format == TBZ_FORMAT_ZLIB) {
        const uint32_t stored = ((uint32_t)tr[0] << 24) | ((uint32_t)tr[1] << 16) | ((uint32_t)tr[2] << 8) | tr[3];
        if (have < 4) S->dec_status = TBZ_INPUT_UNDERRUN;
        else if (stored != S->total_ck) S->dec_status = TBZ_E_ADLER32;
        else { S->trailer_check = stored; S->dec_flags |= 1; S->consumed_abs = S->in_abs + at + 4; }
      } else {
        const uint32_t stored = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        if (have < 4) S->dec_status = TBZ_INPUT_UNDERRUN;
        else if (stored != S->total_ck) S->dec_status = TBZ_E_CRC32;
        else if (have < 8) S->dec_status = TBZ_INPUT_UNDERRUN;
        else {
          S->trailer_check = stored;
          S->trailer_isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
          S->dec_flags |= 1;
          S->consumed_abs = S->in_abs + at + 8;
        }
      }
      S->blk_known = false;  // (trailer cut off: the next call decodes the last blocks again, which is all there is left)
      S->tok_abs_bit = S->hdr_abs_bit = 0;
    } else {
      S->total_ck = S->format == TBZ_FORMAT_ZLIB ? R.adler32 : R.crc32;
      if (S->out_abs && S->format != TBZ_FORMAT_DEFLATE) return TBZ_E_INTERNAL;  // (a resumed stream is past its header)
    }
  }
  return 0;
}

// give up the front of d_in: the stream's octet `abs_byte` becomes the first
static void session_drop_to(tbz_session* S, uint64_t abs_byte) {
  const uint64_t drop = abs_byte > S->in_abs ? std::min<uint64_t>(abs_byte - S->in_abs, S->in_len) : 0;
  S->in_off += drop;
  S->in_len -= drop;
  S->in_abs += drop;
}

// the caller has everything that was decoded and the input ran out: move the resume point — to the TOKEN the input ran
// out in where that is inside a Huffman block (everything decoded leaves the session's view), else to the start of the
// block it ran out in
static int session_advance(tbz_session* S) {
  tbz_ctx* ctx = S->ctx;
  int r;
  const uint8_t* d_new = (const uint8_t*)S->d_dec.p + 32768;
  const uint64_t B = S->hdr_abs_bit, T = S->tok_abs_bit;
  const bool at_token = S->dec_status == TBZ_INPUT_UNDERRUN && !(S->dec_flags & 4) && S->blk_known && T > B &&
                        T >= S->in_abs * 8 && S->delivered == S->dec_len;
  uint64_t out_gone = 0;  // octets of the decode that lie before the new resume point
  if (at_token) {
    out_gone = S->dec_len;
  } else {
    if (!S->blk_known || S->blk_out > S->dec_len) return 0;
    const uint64_t cur = S->hdr_on ? S->hdr_abs * 8 + S->bit_off : S->in_abs * 8 + S->bit_off;
    if (S->blk_abs_bit <= cur || S->blk_abs_bit < S->in_abs * 8) return 0;  // (the block the session already resumes in)
    if (S->blk_abs_bit / 8 > S->in_abs + S->in_len) return 0;
    out_gone = S->blk_out;
  }
  // checksum and window at the new resume point: octets [0, out_gone) of the decode leave the session's view
  if (out_gone) {
    if ((r = session_chain_ck(S, d_new, out_gone, S->ck, &S->ck))) return r;
    const uint64_t nh = std::min<uint64_t>(32768, (S->pad_hist ? 32768 : S->hist_len) + out_gone);
    if ((r = ensure(ctx, S->d_tmp, 32768 + 64))) return r;
    // (history and decode are contiguous in d_dec: the window is the nh octets that end at the resume point)
    TBZ_HIP(hipMemcpyAsync(S->d_tmp.p, d_new + out_gone - nh, nh, hipMemcpyDeviceToDevice, ctx->stream));
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
    std::swap(S->d_hist, S->d_tmp);
    S->hist_len = nh;
  }
  if (at_token) {
    const uint64_t Bb = B / 8, Tb = T / 8;
    if (S->hdr_on && Bb == S->hdr_abs) {
      // still the block whose header is set aside
    } else if (Tb - Bb >= SESSION_HB) {
      // far into a block: its header's octets are set aside, the octets between header and token given up
      if ((r = ensure(ctx, S->d_hdr, SESSION_HB + 64))) return r;
      TBZ_HIP(hipMemcpyAsync(S->d_hdr.p, (const uint8_t*)S->d_in.p + S->in_off + (Bb - S->in_abs), SESSION_HB,
                             hipMemcpyDeviceToDevice, ctx->stream));
      S->hdr_on = true;
      S->hdr_abs = Bb;
    } else {
      S->hdr_on = false;
    }
    S->bit_off = (uint32_t)(B & 7);
    if (S->hdr_on) {
      session_drop_to(S, Tb);
      S->tok_bit = T & 7;
      if (S->tok_bit == 0) {  // (0 means "none": keep the octet before — it is the token's first bit that counts)
        S->in_off -= 1;
        S->in_len += 1;
        S->in_abs -= 1;
        S->tok_bit = 8;
      }
    } else {
      session_drop_to(S, Bb);
      S->tok_bit = T - Bb * 8;
    }
  } else {
    session_drop_to(S, S->blk_abs_bit / 8);
    S->hdr_on = false;
    S->bit_off = (uint32_t)(S->blk_abs_bit & 7);
    S->tok_bit = 0;
  }
  S->header_done = true;  // (a block start lies behind every container header)
  S->out_abs += out_gone;
  S->dec_len -= out_gone;
  S->delivered -= out_gone;
  S->dec_valid = false;   // d_dec is laid out for the old resume point
  return 0;
}
}  // namespace tbz

extern "C" {

int tbz_session_create(tbz_ctx* ctx, int format, tbz_session** out) {
  if (!ctx || !out || format < 0 || format > 2) return TBZ_E_ARG;
  tbz_session* S = new tbz_session();
  S->ctx = ctx;
  S->format = format;
  S->ck = tbz::session_ck_init(format);
  *out = S;
  return 0;
}

void tbz_session_destroy(tbz_session* S) {
  if (!S) return;
  hipSetDevice(S->ctx->device);
  for (tbz::DevBuf* b : {&S->d_in, &S->d_dec, &S->d_hist, &S->d_tmp, &S->d_hdr, &S->d_cat})
    if (b->p) hipFree(b->p);
  delete S;
}

int tbz_session_feed(tbz_session* S, const void* in, size_t in_len, int in_on_device) {
  using namespace tbz;
  if (!S || (in_len && !in)) return TBZ_E_ARG;
  tbz_ctx* ctx = S->ctx;
  TBZ_HIP(hipSetDevice(ctx->device));
  if (in_len == 0 || S->finished || S->error) return 0;
  const size_t need = S->in_len + in_len + 64;
  if (S->in_off + need > S->d_in.cap) {
    if (need <= S->d_in.cap && S->in_off >= S->in_len) {
      // the octets still wanted fit in front of themselves: move them there (the ranges do not overlap)
      if (S->in_len) TBZ_HIP(hipMemcpy(S->d_in.p, (const uint8_t*)S->d_in.p + S->in_off, S->in_len, hipMemcpyDeviceToDevice));
    } else {  // grow (doubling: amortised), keeping what is there
      DevBuf nb;
      int r = ensure(ctx, nb, need * 2 + 4096);
      if (r) return r;
      if (S->in_len) TBZ_HIP(hipMemcpy(nb.p, (const uint8_t*)S->d_in.p + S->in_off, S->in_len, hipMemcpyDeviceToDevice));
      if (S->d_in.p) TBZ_HIP(hipFree(S->d_in.p));
      S->d_in = nb;
    }
    S->in_off = 0;
  }
  TBZ_HIP(hipMemcpy((uint8_t*)S->d_in.p + S->in_off + S->in_len, in, in_len, in_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
  S->in_len += in_len;
  S->dec_valid = false;
  return 0;
}

int tbz_session_decompress(tbz_session* S, uint8_t* out, size_t out_cap, tbz_result* res) {
  using namespace tbz;
  if (!S || !res || (out_cap && !out)) return TBZ_E_ARG;
  tbz_ctx* ctx = S->ctx;
  TBZ_HIP(hipSetDevice(ctx->device));
  memset(res, 0, sizeof(*res));
  auto report = [&](int32_t status, uint64_t gave) {
    res->status = status;
    res->out_len = gave;
    res->out_total = S->out_abs + S->delivered;   // octets of the stream handed out so far
    res->in_consumed = S->finished ? S->consumed_abs : S->in_abs + S->in_len;
    res->boundary_out = S->out_abs;                // ... of which this many lie before the session's resume point
    res->flags = S->dec_flags;
    res->trailer_check = S->trailer_check;
    res->trailer_isize = S->trailer_isize;
    if (S->format == TBZ_FORMAT_ZLIB) res->adler32 = S->total_ck;
    if (S->format == TBZ_FORMAT_GZIP) res->crc32 = S->total_ck;
    return 0;
  };
  if (S->error) return report(S->error, 0);
  if (S->finished) return report(TBZ_FINISHED, 0);
  int r;
  uint64_t give = 0;
  for (;;) {
    if (!S->dec_valid && (r = session_decode(S))) return r;
    const uint64_t g = std::min<uint64_t>(out_cap - give, S->dec_len - S->delivered);
    if (g) {
      TBZ_HIP(hipMemcpy(out + give, (const uint8_t*)S->d_dec.p + 32768 + S->delivered, g, hipMemcpyDeviceToHost));
      S->delivered += g;
      give += g;
    }
    if (S->delivered == S->dec_len && S->dec_status == TBZ_E_DISTANCE && S->window && !S->pad_hist) {
      // the match that reaches before the stream's first octet: the reference has a window by now and reads zeros
      S->pad_hist = true;
      S->dec_valid = false;
      continue;
    }
    if (S->delivered == S->dec_len && S->dec_status == TBZ_INPUT_UNDERRUN && S->sliced) {
      // the decode covered a first part of the input only: everything it produced is handed out, on with the next part
      const size_t before = S->in_len;
      const uint64_t tok_before = S->tok_bit;
      if ((r = session_advance(S))) return r;
      if (S->dec_valid || (S->in_len == before && S->tok_bit == tok_before)) break;  // (no progress: as a plain underrun)
      continue;
    }
    break;
  }
  const bool stored_cut = S->dec_status == TBZ_INPUT_UNDERRUN && (S->dec_flags & 4) && give == out_cap;
  if (S->delivered < S->dec_len || stored_cut) {
    // (stored_cut: the input ran out inside a stored block just where this buffer is full — the reference asks for
    // output space first there, deflate.lisp:538-573)
    S->window = true;
    return report(TBZ_OUTPUT_OVERFLOW, give);
  }
  if (S->dec_status < 0) {
    S->error = S->dec_status;
    return report(S->error, give);
  }
  if (S->dec_status == TBZ_FINISHED) {
    S->finished = true;
    return report(TBZ_FINISHED, give);
  }
  if (S->dec_valid && (r = session_advance(S))) return r;
  return report(TBZ_INPUT_UNDERRUN, give);
}

int tbz_session_stats(const tbz_session* S, uint64_t* n_decodes, uint64_t* in_decoded) {
  if (!S) return TBZ_E_ARG;
  if (n_decodes) *n_decodes = S->n_decodes;
  if (in_decoded) *in_decoded = S->in_decoded;
  return 0;
}

}  // extern "C"

namespace tbz {
// ====================================================================================================
// Multi-member gzip (K0g in tbz_kernels.hpp).  3bz decodes the first member and stops
// (gzip.lisp:277-286): the caller is expected to call again with :start at the next member.  Here every candidate
// range [c_i, c_i+1) is decoded in ONE batch call, its buffer sized by the ISIZE that ends the range, and the walk
// from offset 0 accepts a range as a member iff it FINISHED having consumed exactly its octets (header, blocks,
// CRC32, ISIZE: all verified by the engine).  Anything else — a later candidate lies inside the member's data (a
// false magic), ISIZE understates, the member is damaged — is settled by the ordinary one-stream call from c_i to the
// end of the input (sized first, then decoded), after which the walk goes on where that member ended.  Octets after
// the last member that do not start a member are ignored (as gzip(1) does).
// `place(k, n)`: device memory for the n octets of member k when it is decoded on its own; `done(k, d_ptr, res)`:
// member k is complete (called in member order).
// ====================================================================================================
struct GzMember {
  tbz_result res;
  uint64_t in_off;
};
static int gzip_members_core(tbz_ctx* ctx, const void* d_in, size_t in_len, void* d_out, size_t out_cap, bool own_out,
                             size_t max_members, const std::function<void*(size_t, uint64_t)>& place,
                             const std::function<int(size_t, const void*, const tbz_result&, uint64_t)>& done,
                             size_t* n_members, uint64_t* out_used) {
  *n_members = 0;
  if (out_used) *out_used = 0;
  if (in_len == 0) return 0;
  TBZ_HIP(hipSetDevice(ctx->device));
  int r;
  // ---- candidates
  std::vector<GzCand> cands;
  {
    size_t cap = std::max<size_t>(4096, ctx->d_gz_cands.cap / sizeof(GzCand));
    for (;;) {
      if ((r = ensure(ctx, ctx->d_gz_cands, cap * sizeof(GzCand)))) return r;
      if ((r = ensure(ctx, ctx->d_gz_count, 16))) return r;
      TBZ_HIP(hipMemsetAsync(ctx->d_gz_count.p, 0, 16, ctx->stream));
      K0gParams kp{(const u8*)d_in, (u64)in_len, (GzCand*)ctx->d_gz_cands.p, (u32*)ctx->d_gz_count.p, (u32)std::min<size_t>(cap, 0x7fffffffu)};
      const size_t rows = ((((uintptr_t)d_in) & 15) + in_len + 1023) / 1024;
      TBZ_LAUNCH(tbz_k0g_scan, rows, ctx->stream, kp);
      TBZ_HIP(hipGetLastError());
      uint32_t cnt = 0;
      TBZ_HIP(hipMemcpyAsync(&cnt, ctx->d_gz_count.p, 4, hipMemcpyDeviceToHost, ctx->stream));
      TBZ_HIP(hipStreamSynchronize(ctx->stream));
      if (cnt <= cap) {
        cands.resize(cnt);
        if (cnt) TBZ_HIP(hipMemcpy(cands.data(), ctx->d_gz_cands.p, cnt * sizeof(GzCand), hipMemcpyDeviceToHost));
        break;
      }
      cap = (size_t)cnt + cnt / 4;
    }
    std::sort(cands.begin(), cands.end(), [](const GzCand& a, const GzCand& b) { return a.pos < b.pos; });
    if (cands.empty() || cands[0].pos != 0) cands.insert(cands.begin(), GzCand{0, 0, 0});  // offset 0 is taken as given
  }
  const size_t nc = cands.size();
  uint32_t tail_isize = 0;
  if (in_len >= 4) TBZ_HIP(hipMemcpy(&tail_isize, (const uint8_t*)d_in + in_len - 4, 4, hipMemcpyDeviceToHost));
  std::vector<uint64_t> io(nc), il(nc), oo(nc), oc(nc);
  uint64_t pos_out = 0;
  for (size_t i = 0; i < nc; i++) {
    io[i] = cands[i].pos;
    il[i] = (i + 1 < nc ? cands[i + 1].pos : (uint64_t)in_len) - cands[i].pos;
    const uint64_t hint = i + 1 < nc ? cands[i + 1].before : tail_isize;
    uint64_t cap = (il[i] >= 18 && hint <= 1032 * il[i] + 64) ? hint : 0;
    if (!own_out) cap = std::min<uint64_t>(cap, out_cap > pos_out ? out_cap - pos_out : 0);
    oo[i] = pos_out;
    oc[i] = cap;
    pos_out += (cap + 15) & ~15ull;
    if (!own_out && pos_out > out_cap) pos_out = out_cap;  // (the last range that got room ends inside the final 16 octets)
  }
  if (own_out) {
    if ((r = ensure(ctx, ctx->d_out_stage, pos_out + 64))) return r;
    d_out = ctx->d_out_stage.p;
  }
  if (out_used) *out_used = pos_out;  // (place() must know what the ranges hold before the walk asks it for room)
  std::vector<tbz_result> res(nc);
  if ((r = inflate_passes(ctx, TBZ_FORMAT_GZIP, nc, d_in, io.data(), il.data(), d_out, oo.data(), oc.data(), res.data(), false)))
    return r;
  tbz_timings acc = ctx->tim;
  // ---- second chances, all at once: the commonest reason for a range not to be a whole member is ONE false magic
  // inside the member's data (one every 2^27 octets of compressed data: five in config 3's 518 MB).  Such a member is
  // the range of c_i and the range of the false candidate c_i+1 together, and the latter's buffer was sized by the
  // member's own ISIZE: every such pair is decoded again as one stream [c_i, c_i+2) into that buffer, in ONE batch call;
  // `merged[i]` = it finished there, consuming exactly the two ranges.  Whatever else is wrong takes the one-stream
  // path of the walk below.
  std::vector<uint8_t> merged(nc, 0), dirty(nc, 0);
  {
    std::vector<size_t> idx;
    auto whole = [&](size_t j) { return res[j].status == TBZ_FINISHED && res[j].in_consumed == il[j]; };
    for (size_t i = 0; i < nc;) {
      if (whole(i)) { i++; continue; }
      // (never into the buffer of a range that decoded as a whole member itself: c_i+1 is a true member then — the
      // pair cannot be one —, its octets are in place, and a failed merge would leave them overwritten under a result
      // that still says "finished, checksum verified": a member whose ISIZE understates, followed by good ones)
      if (i + 1 < nc && whole(i + 1)) { i++; continue; }
      if (i + 1 < nc && oc[i + 1] != 0) idx.push_back(i);
      i += 2;
    }
    if (!idx.empty()) {
      const size_t nr = idx.size();
      std::vector<uint64_t> rio(nr), ril(nr), roo(nr), roc(nr);
      for (size_t q = 0; q < nr; q++) {
        const size_t i = idx[q];
        rio[q] = io[i];
        ril[q] = il[i] + il[i + 1];
        roo[q] = oo[i + 1];
        roc[q] = oc[i + 1];
      }
      std::vector<tbz_result> rres(nr);
      if ((r = inflate_passes(ctx, TBZ_FORMAT_GZIP, nr, d_in, rio.data(), ril.data(), d_out, roo.data(), roc.data(), rres.data(), false)))
        return r;
      acc.total_ms += ctx->tim.total_ms;
      acc.huff_launches += ctx->tim.huff_launches;
      for (size_t q = 0; q < nr; q++)
        if (rres[q].status == TBZ_FINISHED && rres[q].in_consumed == ril[q]) {
          merged[idx[q]] = 1;
          res[idx[q]] = rres[q];
        } else {
          dirty[idx[q] + 1] = 1;  // that range's buffer holds the failed attempt now
        }
    }
  }
  // ---- the walk
  size_t k = 0, i = 0;
  while (i < nc && k < max_members) {
    const uint64_t lo = io[i];
    if (merged[i]) {
      if ((r = done(k, (const uint8_t*)d_out + oo[i + 1], res[i], lo))) return r;
      k++;
      i += 2;
      continue;
    }
    if (res[i].status == TBZ_FINISHED && res[i].in_consumed == il[i] && !dirty[i]) {
      if ((r = done(k, (const uint8_t*)d_out + oo[i], res[i], lo))) return r;
      k++;
      i++;
      continue;
    }
    // not a whole member as it stands: a later candidate may lie inside it (a false magic).  The one-stream call over
    // [lo, c_j) decides, j growing until the member finishes inside the range (one false magic in a member: one try)
    uint64_t o1 = 0, l1 = 0, z = 0, cap1 = 0;
    tbz_result r1;
    for (size_t j = i + 2, step = 1;; j += step, step *= 2) {  // (a stored member full of magics: the range doubles)
      l1 = (j < nc ? io[j] : (uint64_t)in_len) - lo;
      if ((r = inflate_core(ctx, TBZ_FORMAT_GZIP, 1, (const uint8_t*)d_in + lo, &o1, &l1, nullptr, &z, &cap1, &r1, true))) return r;
      acc.total_ms += ctx->tim.total_ms;
      acc.huff_launches += ctx->tim.huff_launches;
      if (r1.status != TBZ_INPUT_UNDERRUN || j >= nc) break;
    }
    if (r1.status != TBZ_FINISHED) {  // damaged or incomplete: what (decompress-vector v :format :gzip :start lo) reports
      if (r1.status == TBZ_OUTPUT_OVERFLOW) r1.status = TBZ_E_INTERNAL;  // (cannot happen: unlimited space)
      r1.out_len = 0;
      if ((r = done(k, nullptr, r1, lo))) return r;
      k++;
      break;
    }
    cap1 = r1.out_total;
    // room for it: the range of a false candidate INSIDE this member, if one is large enough — the last of them ends
    // where the member ends, so it was sized by the member's own ISIZE — else wherever `place` finds some
    void* dst = nullptr;
    for (size_t j = i + 1; j < nc && io[j] < lo + r1.in_consumed; j++)
      if (oc[j] >= cap1) {
        dst = (uint8_t*)d_out + oo[j];
        break;
      }
    if (!dst) dst = place(k, cap1);
    if (!dst && cap1) {  // no room: reported as the reference reports a buffer that is too small
      r1.status = TBZ_OUTPUT_OVERFLOW;
      r1.out_len = 0;
      if ((r = done(k, nullptr, r1, lo))) return r;
      k++;
      break;
    }
    tbz_result r2;
    if ((r = inflate_core(ctx, TBZ_FORMAT_GZIP, 1, (const uint8_t*)d_in + lo, &o1, &l1, dst, &z, &cap1, &r2, false))) return r;
    acc.total_ms += ctx->tim.total_ms;
    acc.huff_launches += ctx->tim.huff_launches;
    if ((r = done(k, dst, r2, lo))) return r;
    k++;
    if (r2.status != TBZ_FINISHED) break;
    const uint64_t nxt = lo + r2.in_consumed;
    while (i < nc && io[i] < nxt) i++;
    if (i < nc && io[i] != nxt) break;  // what follows the member is not a member
  }
  *n_members = k;
  if (out_used) *out_used = pos_out;
  acc.n_candidates = nc;
  ctx->tim = acc;
  return 0;
}
}  // namespace tbz

extern "C" {
int tbz_inflate_gzip_members_device(tbz_ctx* ctx, const void* d_in, size_t in_len, void* d_out, size_t out_cap,
                                    size_t max_members, tbz_result* results, uint64_t* member_in_off,
                                    uint64_t* member_out_off, size_t* n_members) {
  using namespace tbz;
  if (!ctx || !n_members || (max_members && (!results || !member_in_off || !member_out_off)) || (in_len && !d_in)) return TBZ_E_ARG;
  if (out_cap && !d_out) return TBZ_E_ARG;
  // the candidate ranges take [0, out_used) of d_out in candidate order (16-octet aligned, sized by their ISIZE); a member
  // that has to be decoded on its own is placed from the END of the buffer downwards
  uint64_t top = out_cap;
  uint64_t out_used = 0;  // set by the core before its walk: the candidate ranges hold [0, out_used)
  auto place = [&](size_t, uint64_t n) -> void* {
    if (n > top) return nullptr;
    const uint64_t t = (top - n) & ~15ull;
    if (t < out_used) return nullptr;  // would run into a range (whose member may have been delivered already): no room
    top = t;
    return (uint8_t*)d_out + top;
  };
  auto done = [&](size_t k, const void* d_ptr, const tbz_result& r, uint64_t in_off) -> int {
    results[k] = r;
    member_in_off[k] = in_off;
    member_out_off[k] = d_ptr ? (uint64_t)((const uint8_t*)d_ptr - (const uint8_t*)d_out) : 0;
    return 0;
  };
  return gzip_members_core(ctx, d_in, in_len, d_out, out_cap, false, max_members, place, done, n_members, &out_used);
}

int tbz_inflate_gzip_members(tbz_ctx* ctx, const uint8_t* in, size_t in_len, tbz_alloc_fn alloc, void* user,
                             size_t max_members, tbz_result* results, uint64_t* member_in_off, size_t* n_members) {
  using namespace tbz;
  if (!ctx || !alloc || !n_members || (max_members && (!results || !member_in_off)) || (in_len && !in)) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  int r;
  if ((r = ensure(ctx, ctx->d_in_stage, in_len + 64))) return r;
  if ((r = stage_in(ctx, ctx->d_in_stage.p, in, in_len))) return r;
  auto place = [&](size_t, uint64_t n) -> void* {
    if (ensure(ctx, ctx->d_gz_tmp, n + 64)) return nullptr;
    return ctx->d_gz_tmp.p;
  };
  auto done = [&](size_t k, const void* d_ptr, const tbz_result& res, uint64_t in_off) -> int {
    results[k] = res;
    member_in_off[k] = in_off;
    if (res.status >= 0) {
      uint8_t* out = alloc(user, (size_t)res.out_len);
      if (res.out_len) {
        if (!out) return TBZ_E_NOMEM;
        int rr = stage_out(ctx, out, d_ptr, res.out_len);
        if (rr) return rr;
      }
    }
    return 0;
  };
  return gzip_members_core(ctx, ctx->d_in_stage.p, in_len, nullptr, 0, true, max_members, place, done, n_members, nullptr);
}
}  // extern "C"

extern "C" {
int tbz_ctx_trim(tbz_ctx* ctx) {
  using namespace tbz;
  if (!ctx) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  TBZ_HIP(hipStreamSynchronize(ctx->stream));
  for (auto* b : all_pools(ctx))
    if (b != &ctx->d_crc_tab && b->p) {
      TBZ_HIP(hipFree(b->p));
      b->p = nullptr;
      b->cap = 0;
    }
  for (DevBuf& b : ctx->dense)
    if (b.p) TBZ_HIP(hipFree(b.p));
  ctx->dense.clear();
  return 0;
}

int tbz_inflate_device(tbz_ctx* ctx, int format, const void* d_in, size_t in_len, void* d_out, size_t out_cap,
                       tbz_result* res) {
  uint64_t io = 0, il = in_len, oo = 0, oc = out_cap;
  return tbz_inflate_batch_device(ctx, format, 1, d_in, &io, &il, d_out, &oo, &oc, res);
}

// ---- ONE large stream, host to host, part by part: the input of part k+1 on its way to the device, part k being decoded
// and the output of part k-1 on its way back, all at the same time (PCIe is full duplex, and the decode is a quarter of
// either transfer).  The parts are tbz_inflate_sharded_plan's — cut at flush markers, part 0 in the stream's format, the
// rest as raw deflate — and tbz_inflate_sharded_verdict proves the seams from the parts' own results, exactly as for a
// stream sharded across GPUs.  Whatever is not a clean chain (no markers, history across a cut, any error, a buffer that
// is too small) is NOT decided here: *handled = false, the whole input is on the device by then (*uploaded), and the
// caller decodes it by the ordinary path, whose statuses are the answer.
static int sharded_plan(const uint8_t* in, size_t in_len, size_t n_parts, uint64_t* cuts, uint64_t max_scan);
static int inflate_host_pipelined(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                                  tbz_result* res, bool* handled, bool* uploaded) {
  using namespace tbz;
  *handled = *uploaded = false;
  const size_t S = std::min<size_t>(64, std::max<size_t>(2, (in_len + ctx->pipe_part - 1) / ctx->pipe_part));
  std::vector<uint64_t> cuts(S + 1);
  int r;
  if ((r = sharded_plan(in, in_len, S, cuts.data(), 2 * ctx->pipe_part))) return r;  // (a marker within two parts' worth, or none)
  size_t live = 0;
  for (size_t k = 0; k < S; k++) live += cuts[k + 1] > cuts[k];
  if (live < 2) return 0;  // (no flush markers: nothing to cut at)
  if ((r = stage_setup(ctx))) return r;
  // the two transfers are CONTINUOUS chunk pipelines over the whole input / the whole output, not one per part (a
  // pipeline per part pays its fill and its drain eighteen times: measured 38 GB/s where the link does 57): the input
  // thread publishes how many octets have arrived, the output thread follows how many have been decoded
  std::atomic<uint64_t> arrived{0}, avail{0};
  std::atomic<bool> all_decoded{false};
  std::atomic<int> in_err{0}, out_err{0};
  std::atomic<bool> abort_out{false};
  std::vector<uint64_t> off(S + 1, 0);  // where part k's output starts
  std::vector<tbz_result> recs(S);
  std::vector<uint32_t> cks(S, 0);
  for (auto& q : recs) memset(&q, 0, sizeof q);
  const uint64_t ch = ctx->stage_chunk;
  auto hip_ok = [](hipError_t e) { return e == hipSuccess ? 0 : (int)TBZ_E_HIP; };
  std::thread t_in([&]() {
    hipSetDevice(ctx->device);
    const size_t nch = (size_t)((in_len + ch - 1) / ch);
    int e = 0;
    for (size_t j = 0; j < nch && !e; j++) {
      const int b = (int)(j & 1);
      const uint64_t o = j * ch, n = std::min<uint64_t>(ch, in_len - o);
      if (j >= 2) e = hip_ok(hipEventSynchronize(ctx->ev_stage[b]));
      if (!e) {
        if (ctx->copy_pool) ctx->copy_pool->copy(ctx->h_stage[b], in + o, n); else memcpy(ctx->h_stage[b], in + o, n);
        e = hip_ok(hipMemcpyAsync((uint8_t*)ctx->d_in_stage.p + o, ctx->h_stage[b], n, hipMemcpyHostToDevice, ctx->stream_in));
      }
      if (!e) e = hip_ok(hipEventRecord(ctx->ev_stage[b], ctx->stream_in));
      // the chunk before this one moved while this one was being filled: it has (all but) arrived
      if (!e && j >= 1) {
        e = hip_ok(hipEventSynchronize(ctx->ev_stage[b ^ 1]));
        if (!e) arrived.store(j * ch, std::memory_order_release);
      }
    }
    if (!e && nch) e = hip_ok(hipEventSynchronize(ctx->ev_stage[(nch - 1) & 1]));
    if (e) in_err.store(e);
    arrived.store(in_len, std::memory_order_release);
  });
  std::thread t_out([&]() {
    hipSetDevice(ctx->device);
    int e = 0;
    uint64_t issued = 0;        // octets whose transfer has been issued
    uint64_t pend_o[2] = {0, 0}, pend_n[2] = {0, 0};
    size_t j = 0;
    auto drain = [&](int b) {   // the chunk in pinned buffer b: to the caller's buffer
      if (!pend_n[b] || e) return;
      e = hip_ok(hipEventSynchronize(ctx->ev_stage[2 + b]));
      if (!e) {
        if (ctx->copy_pool2) ctx->copy_pool2->copy(out + pend_o[b], ctx->h_stage[2 + b], pend_n[b]);
        else memcpy(out + pend_o[b], ctx->h_stage[2 + b], pend_n[b]);
      }
      pend_n[b] = 0;
    };
    for (;;) {
      // the next chunk: a whole one as soon as that much has been decoded, the rest when everything has
      uint64_t n = 0;
      for (;;) {
        if (abort_out.load()) return;
        const uint64_t av = avail.load(std::memory_order_acquire);
        const bool fin = all_decoded.load(std::memory_order_acquire);
        if (av - issued >= ch) { n = ch; break; }
        if (fin) { n = avail.load(std::memory_order_acquire) - issued; break; }
        std::this_thread::yield();
      }
      const int b = (int)(j & 1);
      drain(b);  // (its buffer is about to be reused)
      if (n && !e) {
        e = hip_ok(hipMemcpyAsync(ctx->h_stage[2 + b], (const uint8_t*)ctx->d_out_stage.p + issued, n, hipMemcpyDeviceToHost, ctx->stream_out));
        if (!e) e = hip_ok(hipEventRecord(ctx->ev_stage[2 + b], ctx->stream_out));
        pend_o[b] = issued;
        pend_n[b] = n;
        issued += n;
        j++;
      }
      drain((int)(j & 1));  // the chunk before the one just issued, while that one moves
      if (e || (all_decoded.load(std::memory_order_acquire) && issued == avail.load(std::memory_order_acquire))) break;
    }
    drain(0);
    drain(1);
    if (e) out_err.store(e);
  });
  size_t last = 0;
  for (size_t k = 0; k < S; k++)
    if (cuts[k + 1] > cuts[k] || k == 0) last = k;
  bool clean = true;
  int rc = 0;
  tbz_timings acc{};
  for (size_t k = 0; k <= last && clean; k++) {
    while (arrived.load(std::memory_order_acquire) < cuts[k + 1]) std::this_thread::yield();
    if (in_err.load()) { clean = false; break; }
    off[k + 1] = off[k];
    const uint64_t n = cuts[k + 1] - cuts[k];
    if (!n && k) continue;
    const int f = k == 0 ? format : TBZ_FORMAT_DEFLATE;
    uint64_t io = cuts[k], il = n, oo = off[k], oc = out_cap - off[k];
    if ((rc = inflate_core(ctx, f, 1, ctx->d_in_stage.p, &io, &il, ctx->d_out_stage.p, &oo, &oc, &recs[k], false))) { clean = false; break; }
    const tbz_timings& t = ctx->tim;
    acc.scan_ms += t.scan_ms; acc.huff_ms += t.huff_ms; acc.lz_ms += t.lz_ms; acc.cksum_ms += t.cksum_ms; acc.total_ms += t.total_ms;
    acc.find_ms += t.find_ms; acc.resolve_ms += t.resolve_ms; acc.huff_launches += t.huff_launches; acc.token_words += t.token_words;
    acc.n_segments += t.n_segments; acc.n_groups += t.n_groups; acc.n_candidates += t.n_candidates; acc.n_hgroups += t.n_hgroups;
    acc.k1_gang = t.k1_gang; acc.k2_kinds |= t.k2_kinds; acc.scratch_bytes = std::max(acc.scratch_bytes, t.scratch_bytes);
    const bool ok = k == last ? recs[k].status == TBZ_FINISHED : (recs[k].status == TBZ_INPUT_UNDERRUN && recs[k].in_consumed == n);
    if (!ok) { clean = false; break; }
    if (format != TBZ_FORMAT_DEFLATE) {
      std::vector<uint64_t> o{off[k]}, l{recs[k].out_len};
      std::vector<uint32_t> init{format == TBZ_FORMAT_ZLIB ? 1u : 0u}, sums;
      if ((rc = run_checksums(ctx, format == TBZ_FORMAT_ZLIB ? 1 : 2, ctx->d_out_stage.p, o, l, init, sums))) { clean = false; break; }
      cks[k] = sums[0];
    }
    off[k + 1] = off[k] + recs[k].out_len;
    for (size_t j = k + 2; j <= S; j++) off[j] = off[k + 1];
    avail.store(off[k + 1], std::memory_order_release);  // (octets that leave before a verdict that turns out not clean are the
                                                         // ones the ordinary path then delivers again, or a prefix of them)
  }
  uint64_t total = 0, consumed = 0;
  uint32_t check = 0;
  if (clean) {
    int why = 0;
    const int v = tbz_inflate_sharded_verdict(format, in, in_len, S, cuts.data(), recs.data(), cks.data(), nullptr, &total, &check,
                                              &consumed, &why);
    if (v != 0) clean = false;
    if (ctx->tun.debug) fprintf(stderr, "tbz: pipelined host decode: %zu parts, verdict %d (why %d)\n", last + 1, v, why);
  } else if (ctx->tun.debug) {
    for (size_t k = 0; k <= last; k++)
      fprintf(stderr, "tbz: pipelined host decode: part %zu status %d consumed %llu of %llu out %llu\n", k, recs[k].status,
              (unsigned long long)recs[k].in_consumed, (unsigned long long)(cuts[k + 1] - cuts[k]), (unsigned long long)recs[k].out_len);
  }
  if (clean) {
    avail.store(off[last + 1], std::memory_order_release);
    all_decoded.store(true, std::memory_order_release);
  } else {
    abort_out.store(true);
  }
  t_in.join();   // (the fallback decodes from the device: the whole input has to be there)
  t_out.join();
  *uploaded = in_err.load() == 0;
  if (rc) return rc;
  if (in_err.load()) return in_err.load();
  if (!clean) return 0;
  if (out_err.load()) return out_err.load();
  memset(res, 0, sizeof *res);
  res->status = TBZ_FINISHED;
  res->out_len = res->out_total = res->boundary_out = total;
  res->in_consumed = consumed;
  res->flags = 2u | (format != TBZ_FORMAT_DEFLATE ? 1u : 0u);
  for (size_t k = 0; k <= last; k++) res->segments += recs[k].segments;
  if (format == TBZ_FORMAT_ZLIB) {
    res->adler32 = res->trailer_check = check;
  } else if (format == TBZ_FORMAT_GZIP) {
    res->crc32 = res->trailer_check = check;
    const uint8_t* t = in + (consumed - 4);
    res->trailer_isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
  }
  acc.passes = (uint32_t)(last + 1);
  ctx->tim = acc;
  *handled = true;
  return 0;
}

// ---- a large BATCH, host to host, sub-batch by sub-batch: the same three movers as above — one thread gathers the
// callers' input buffers into pinned chunks and streams them up, the calling thread decodes a sub-batch of consecutive
// streams (about pipe_part octets of input) as soon as its last stream has arrived, a third thread streams the output
// range down as far as it has been decoded and scatters it into the callers' buffers.  Streams are independent: there is
// no verdict and no fallback, every stream's result is what the one-batch call gives it.
static int batch_host_pipelined(tbz_ctx* ctx, int format, size_t n, const uint8_t* const* ins, uint8_t* const* outs,
                                const std::vector<uint64_t>& io, const std::vector<uint64_t>& il, const std::vector<uint64_t>& oo,
                                const std::vector<uint64_t>& oc, uint64_t in_total, uint64_t out_total, tbz_result* results) {
  using namespace tbz;
  int r;
  if ((r = stage_setup(ctx))) return r;
  std::vector<StagePiece> ps_in, ps_out(n);
  for (size_t i = 0; i < n; i++) {
    if (il[i]) ps_in.push_back(StagePiece{(uint8_t*)ins[i], io[i], il[i]});
    ps_out[i] = StagePiece{outs[i], oo[i], 0};
  }
  std::atomic<uint64_t> arrived{0}, avail{0};
  std::atomic<bool> all_decoded{false}, abort_out{false};
  std::atomic<int> in_err{0}, out_err{0};
  const uint64_t ch = ctx->stage_chunk;
  auto hip_ok = [](hipError_t e) { return e == hipSuccess ? 0 : (int)TBZ_E_HIP; };
  std::thread t_in([&]() {
    hipSetDevice(ctx->device);
    const size_t nch = (size_t)((in_total + ch - 1) / ch);
    std::vector<CopySeg> sg;
    size_t cursor = 0;
    int e = 0;
    for (size_t j = 0; j < nch && !e; j++) {
      const int b = (int)(j & 1);
      const uint64_t o = j * ch, len = std::min<uint64_t>(ch, in_total - o);
      if (j >= 2) e = hip_ok(hipEventSynchronize(ctx->ev_stage[b]));
      if (!e) {
        stage_segments(ps_in, cursor, o, o + len, (uint8_t*)ctx->h_stage[b], true, sg);
        stage_run(ctx->copy_pool, sg);
        e = hip_ok(hipMemcpyAsync((uint8_t*)ctx->d_in_stage.p + o, ctx->h_stage[b], len, hipMemcpyHostToDevice, ctx->stream_in));
      }
      if (!e) e = hip_ok(hipEventRecord(ctx->ev_stage[b], ctx->stream_in));
      if (!e && j >= 1) {
        e = hip_ok(hipEventSynchronize(ctx->ev_stage[b ^ 1]));
        if (!e) arrived.store(j * ch, std::memory_order_release);
      }
    }
    if (!e && nch) e = hip_ok(hipEventSynchronize(ctx->ev_stage[(nch - 1) & 1]));
    if (e) in_err.store(e);
    arrived.store(in_total, std::memory_order_release);
  });
  std::thread t_out([&]() {
    hipSetDevice(ctx->device);
    int e = 0;
    uint64_t issued = 0, pend_o[2] = {0, 0}, pend_n[2] = {0, 0};
    size_t j = 0, cursor = 0;
    std::vector<CopySeg> sg;
    auto drain = [&](int b) {   // the chunk in pinned buffer b: scattered into the callers' buffers
      if (!pend_n[b] || e) return;
      e = hip_ok(hipEventSynchronize(ctx->ev_stage[2 + b]));
      if (!e) {
        stage_segments(ps_out, cursor, pend_o[b], pend_o[b] + pend_n[b], (uint8_t*)ctx->h_stage[2 + b], false, sg);
        stage_run(ctx->copy_pool2, sg);
      }
      pend_n[b] = 0;
    };
    for (;;) {
      uint64_t len = 0;
      for (;;) {
        if (abort_out.load()) return;
        const uint64_t av = avail.load(std::memory_order_acquire);
        const bool fin = all_decoded.load(std::memory_order_acquire);
        if (av - issued >= ch) { len = ch; break; }
        if (fin) { len = avail.load(std::memory_order_acquire) - issued; break; }
        std::this_thread::yield();
      }
      const int b = (int)(j & 1);
      drain(b);
      if (len && !e) {
        e = hip_ok(hipMemcpyAsync(ctx->h_stage[2 + b], (const uint8_t*)ctx->d_out_stage.p + issued, len, hipMemcpyDeviceToHost, ctx->stream_out));
        if (!e) e = hip_ok(hipEventRecord(ctx->ev_stage[2 + b], ctx->stream_out));
        pend_o[b] = issued;
        pend_n[b] = len;
        issued += len;
        j++;
      }
      drain((int)(j & 1));
      if (e || (all_decoded.load(std::memory_order_acquire) && issued == avail.load(std::memory_order_acquire))) break;
    }
    drain(0);
    drain(1);
    if (e) out_err.store(e);
  });
  int rc = 0;
  tbz_timings acc{};
  size_t units = 0;
  for (size_t a = 0; a < n && !rc;) {
    size_t b = a;
    uint64_t got = 0;
    // (a sub-batch must still fill the chip: a gang per stream takes as long for 256 streams as for 4 096 — config 3 in
    // sixteen sub-batches took 52 ms against 45 with its three legs one after the other — so: four units at most)
    const uint64_t unit = std::max<uint64_t>(ctx->pipe_part, in_total / 4);
    while (b < n && (b == a || got < unit)) got += il[b++];
    const uint64_t need = io[b - 1] + il[b - 1];
    while (arrived.load(std::memory_order_acquire) < need) std::this_thread::yield();
    if (in_err.load()) break;
    rc = inflate_passes(ctx, format, b - a, ctx->d_in_stage.p, io.data() + a, il.data() + a, ctx->d_out_stage.p, oo.data() + a,
                        oc.data() + a, results + a, false);
    if (rc) break;
    const tbz_timings& t = ctx->tim;
    acc.scan_ms += t.scan_ms; acc.huff_ms += t.huff_ms; acc.lz_ms += t.lz_ms; acc.cksum_ms += t.cksum_ms; acc.total_ms += t.total_ms;
    acc.find_ms += t.find_ms; acc.resolve_ms += t.resolve_ms; acc.huff_launches += t.huff_launches; acc.token_words += t.token_words;
    acc.n_segments += t.n_segments; acc.n_groups += t.n_groups; acc.n_candidates += t.n_candidates; acc.n_hgroups += t.n_hgroups;
    acc.k1_gang = t.k1_gang; acc.k2_kinds |= t.k2_kinds; acc.scratch_bytes = std::max(acc.scratch_bytes, t.scratch_bytes);
    for (size_t i = a; i < b; i++) ps_out[i].len = (results[i].status >= 0 && outs[i]) ? results[i].out_len : 0;
    avail.store(b < n ? oo[b] : out_total, std::memory_order_release);
    units++;
    a = b;
  }
  if (rc || in_err.load()) abort_out.store(true); else all_decoded.store(true, std::memory_order_release);
  t_in.join();
  t_out.join();
  if (rc) return rc;
  if (in_err.load()) return in_err.load();
  if (out_err.load()) return out_err.load();
  acc.passes = (uint32_t)units;
  acc.h2d_copies = (uint32_t)ps_in.size();
  ctx->tim = acc;
  return 0;
}

static int stage_batch(tbz_ctx* ctx, int format, size_t n, const uint8_t* const* ins, const size_t* in_lens,
                       uint8_t* const* outs, const size_t* out_caps, tbz_result* results, bool size_only) {
  using namespace tbz;
  if (!ctx || !results || (n && (!ins || !in_lens))) return TBZ_E_ARG;
  if (!size_only && n && (!outs || !out_caps)) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> io(n), il(n), oo(n), oc(n);
  uint64_t it = 0, ot = 0;
  for (size_t i = 0; i < n; i++) {
    io[i] = it;
    il[i] = in_lens[i];
    it += (in_lens[i] + 15) & ~15ull;
    oo[i] = ot;
    oc[i] = size_only ? 0 : out_caps[i];
    ot += ((size_only ? 0 : out_caps[i]) + 15) & ~15ull;
  }
  int r;
  if ((r = ensure(ctx, ctx->d_in_stage, it + 64))) return r;
  if (!size_only && (r = ensure(ctx, ctx->d_out_stage, ot + 64))) return r;
  const double t_in = now_ms();
  bool uploaded = false;
  if (n == 1 && !size_only && ctx->pipe_min && in_lens[0] >= ctx->pipe_min && ins[0] && (outs[0] || !out_caps[0])) {
    bool handled = false;
    if ((r = inflate_host_pipelined(ctx, format, ins[0], in_lens[0], outs[0], out_caps[0], &results[0], &handled, &uploaded))) return r;
    if (handled) {
      ctx->tim.h2d_copies = 1;
      ctx->tim.h2d_ms = ctx->tim.d2h_ms = 0.f;  // (the legs overlap: only their sum means anything)
      ctx->tim.host_decode_ms = (float)(now_ms() - t_in);
      return 0;
    }
  }
  if (n > 1 && !size_only && ctx->pipe_min && it >= ctx->pipe_min) {
    for (size_t i = 0; i < n; i++)
      if ((in_lens[i] && !ins[i]) || (out_caps[i] && !outs[i])) return TBZ_E_ARG;
    if ((r = batch_host_pipelined(ctx, format, n, ins, outs, io, il, oo, oc, it, ot, results))) return r;
    ctx->tim.h2d_ms = ctx->tim.d2h_ms = 0.f;
    ctx->tim.host_decode_ms = (float)(now_ms() - t_in);
    return 0;
  }
  if (!uploaded) {
    std::vector<StagePiece> ps;
    for (size_t i = 0; i < n; i++)
      if (in_lens[i]) {
        if (!ins[i]) return TBZ_E_ARG;
        ps.push_back(StagePiece{(uint8_t*)ins[i], io[i], in_lens[i]});
      }
    if ((r = stage_in(ctx, ctx->d_in_stage.p, ps))) return r;
  }
  const double t_dec = now_ms();
  uint32_t n_h2d = 0;
  for (size_t i = 0; i < n; i++) n_h2d += in_lens[i] ? 1u : 0u;
  r = inflate_passes(ctx, format, n, ctx->d_in_stage.p, io.data(), il.data(), size_only ? nullptr : ctx->d_out_stage.p,
                     oo.data(), oc.data(), results, size_only);
  ctx->tim.h2d_copies = n_h2d;
  if (r) return r;
  const double t_out = now_ms();
  if (!size_only) {
    std::vector<StagePiece> ps;
    for (size_t i = 0; i < n; i++)
      if (results[i].status >= 0 && results[i].out_len) {
        if (!outs[i]) return TBZ_E_ARG;
        ps.push_back(StagePiece{outs[i], oo[i], results[i].out_len});
      }
    if ((r = stage_out(ctx, ctx->d_out_stage.p, ps))) return r;
    TBZ_HIP(hipStreamSynchronize(ctx->stream));
  }
  // host wall-clock of the three legs of a host-to-host call (the first leg ends when the last input chunk has been
  // handed to the DMA engine: its tail overlaps the decode)
  ctx->tim.h2d_ms = (float)(t_dec - t_in);
  ctx->tim.host_decode_ms = (float)(t_out - t_dec);
  ctx->tim.d2h_ms = (float)(now_ms() - t_out);
  return 0;
}

int tbz_inflate_batch(tbz_ctx* ctx, int format, size_t n, const uint8_t* const* ins, const size_t* in_lens,
                      uint8_t* const* outs, const size_t* out_caps, tbz_result* results) {
  return stage_batch(ctx, format, n, ins, in_lens, outs, out_caps, results, false);
}

int tbz_inflate(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                tbz_result* res) {
  return stage_batch(ctx, format, 1, &in, &in_len, &out, &out_cap, res, false);
}

int tbz_gzip_header_parse(const uint8_t* in, size_t in_len, tbz_gzip_header* h) {
  if (!h || (in_len && !in)) return TBZ_E_ARG;
  memset(h, 0, sizeof(*h));
  size_t p = 0;
  uint32_t crc = 0xffffffffu;
  auto crc_octet = [&](uint8_t b) {
    crc ^= b;
    for (int k = 0; k < 8; k++) crc = (crc & 1) ? (0xedb88320u ^ (crc >> 1)) : (crc >> 1);
  };
  auto take = [&](size_t n) -> bool {  // the next n octets, all or input-underrun (the reference reads fields whole)
    if (p + n > in_len) {
      h->status = TBZ_INPUT_UNDERRUN;
      return false;
    }
    for (size_t i = 0; i < n; i++) crc_octet(in[p + i]);
    p += n;
    return true;
  };
  // h->stage says how far the header has been read: the reference fills the state's slots as it goes
  if (!take(2)) return 0;                                    // gzip.lisp:113-121
  if (in[0] != 0x1f || in[1] != 0x8b) { h->status = TBZ_E_GZIP_MAGIC; return 0; }
  h->stage = 1;
  if (!take(2)) return 0;                                    // gzip.lisp:123-139
  h->cm = in[2];
  h->flg = in[3];
  if (h->cm != 8) { h->status = TBZ_E_GZIP_METHOD; return 0; }
  if (h->flg >> 5) { h->status = TBZ_E_GZIP_FLAGS; return 0; }
  h->stage = 2;
  if (!take(4)) return 0;                                    // gzip.lisp:144-157
  h->mtime = (uint32_t)in[4] | ((uint32_t)in[5] << 8) | ((uint32_t)in[6] << 16) | ((uint32_t)in[7] << 24);
  h->stage = 3;
  if (!take(2)) return 0;                                    // gzip.lisp:159-176
  h->xfl = in[8];
  h->os = in[9];
  h->stage = 4;
  if (h->flg & 4) {                                          // gzip.lisp:177-196
    if (!take(2)) return 0;
    h->extra_len = (uint32_t)in[p - 2] | ((uint32_t)in[p - 1] << 8);
    h->extra_off = (uint32_t)p;
    if (!take(h->extra_len)) return 0;
  }
  h->stage = 5;
  for (int f = 0; f < 2; f++) {                              // gzip.lisp:197-241: zero-terminated name, comment
    if (h->flg & (f ? 16 : 8)) {
      const size_t s0 = p;
      for (;;) {
        if (!take(1)) return 0;
        if (in[p - 1] == 0) break;
      }
      (f ? h->comment_off : h->name_off) = (uint32_t)s0;
      (f ? h->comment_len : h->name_len) = (uint32_t)(p - 1 - s0);
    }
    h->stage = 6 + (uint32_t)f;
  }
  if (h->flg & 2) {                                          // gzip.lisp:242-255
    const uint32_t want = (crc ^ 0xffffffffu) & 0xffffu;
    if (p + 2 > in_len) { h->status = TBZ_INPUT_UNDERRUN; return 0; }
    h->hcrc_present = 1;
    h->hcrc = (uint32_t)in[p] | ((uint32_t)in[p + 1] << 8);
    p += 2;
    if (h->hcrc != want) { h->status = TBZ_E_GZIP_HCRC; return 0; }
  }
  h->header_len = (uint32_t)p;
  h->stage = 8;
  return 0;
}

int tbz_inflate_alloc(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, tbz_alloc_fn alloc, void* user,
                      tbz_result* res) {
  using namespace tbz;
  if (!ctx || !res || !alloc || (in_len && !in)) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  int r;
  if ((r = ensure(ctx, ctx->d_in_stage, in_len + 64))) return r;
  if ((r = stage_in(ctx, ctx->d_in_stage.p, in, in_len))) return r;
  CoreOpts opt;
  int alloc_err = 0;
  opt.alloc = [&](uint64_t total) -> void* {
    alloc_err = ensure(ctx, ctx->d_out_stage, total + 64);
    return alloc_err ? nullptr : ctx->d_out_stage.p;
  };
  uint64_t io = 0, il = in_len, oo = 0, oc = 1ull << 62;
  r = inflate_core(ctx, format, 1, ctx->d_in_stage.p, &io, &il, nullptr, &oo, &oc, res, false, &opt);
  ctx->tim.h2d_copies = in_len ? 1u : 0u;
  if (r) return alloc_err ? alloc_err : r;
  if (res->status >= 0) {
    uint8_t* out = alloc(user, (size_t)res->out_len);
    if (res->out_len) {
      if (!out) return TBZ_E_NOMEM;
      if ((r = stage_out(ctx, out, ctx->d_out_stage.p, res->out_len))) return r;
    }
  }
  return 0;
}

// one decode of host input into DEVICE memory the caller then owns (tbz_device_free): the buffer is allocated once K1
// has sized it — no sizing pass, no second decode (what a rank of a sharded stream keeps: 3bz_amd/multi.py)
int tbz_inflate_to_device(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, void** d_out, tbz_result* res) {
  using namespace tbz;
  if (!ctx || !res || !d_out || (in_len && !in)) return TBZ_E_ARG;
  *d_out = nullptr;
  TBZ_HIP(hipSetDevice(ctx->device));
  int r;
  if ((r = ensure(ctx, ctx->d_in_stage, in_len + 64))) return r;
  if ((r = stage_in(ctx, ctx->d_in_stage.p, in, in_len))) return r;
  CoreOpts opt;
  void* buf = nullptr;
  bool alloc_failed = false;
  opt.alloc = [&](uint64_t total) -> void* {
    if (hipMalloc(&buf, total + 64) != hipSuccess) {
      (void)hipGetLastError();
      buf = nullptr;
      alloc_failed = true;
    }
    return buf;
  };
  uint64_t io = 0, il = in_len, oo = 0, oc = 1ull << 62;
  r = inflate_core(ctx, format, 1, ctx->d_in_stage.p, &io, &il, nullptr, &oo, &oc, res, false, &opt);
  ctx->tim.h2d_copies = in_len ? 1u : 0u;
  if (r) {
    if (buf) hipFree(buf);
    return alloc_failed ? TBZ_E_NOMEM : r;  // the output allocation refused — anything else is the decode's own error
  }
  *d_out = buf;
  return 0;
}

// ---- several devices from one host process ----------------------------------------------------------------------
// longest-processing-time-first on compressed size, ties by index: the assignment 3bz_amd/multi.py makes for its ranks
int tbz_assign_streams(const size_t* in_lens, size_t n, size_t n_parts, uint32_t* owner) {
  if (!owner || !n_parts || (n && !in_lens)) return TBZ_E_ARG;
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; i++) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return in_lens[a] > in_lens[b]; });
  std::vector<uint64_t> load(n_parts, 0);
  for (size_t i : order) {
    size_t best = 0;
    for (size_t k = 1; k < n_parts; k++)
      if (load[k] < load[best]) best = k;
    owner[i] = (uint32_t)best;
    load[best] += in_lens[i];
  }
  return 0;
}

// n independent streams over n_ctx contexts (one per device, normally): each context decodes its share in ONE batch
// call on a host thread of its own; results come back in stream order.  Returns the first engine error, if any.
int tbz_inflate_batch_multi(tbz_ctx* const* ctxs, size_t n_ctx, int format, size_t n, const uint8_t* const* ins,
                            const size_t* in_lens, uint8_t* const* outs, const size_t* out_caps, tbz_result* results) {
  if (!ctxs || !n_ctx || !results || (n && (!ins || !in_lens || !outs || !out_caps))) return TBZ_E_ARG;
  for (size_t k = 0; k < n_ctx; k++) {
    if (!ctxs[k]) return TBZ_E_ARG;
    for (size_t j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return TBZ_E_ARG;  // (two host threads on one context's pools and streams)
  }
  std::vector<uint32_t> owner(n);
  int r = tbz_assign_streams(in_lens, n, n_ctx, owner.data());
  if (r) return r;
  std::vector<int> rc(n_ctx, 0);
  std::vector<std::thread> th;
  for (size_t k = 0; k < n_ctx; k++)
    th.emplace_back([&, k]() {
      std::vector<size_t> idx;
      for (size_t i = 0; i < n; i++)
        if (owner[i] == k) idx.push_back(i);
      if (idx.empty()) return;
      const size_t m = idx.size();
      std::vector<const uint8_t*> i_(m);
      std::vector<uint8_t*> o_(m);
      std::vector<size_t> il(m), ol(m);
      std::vector<tbz_result> rs(m);
      for (size_t q = 0; q < m; q++) {
        i_[q] = ins[idx[q]];
        il[q] = in_lens[idx[q]];
        o_[q] = outs[idx[q]];
        ol[q] = out_caps[idx[q]];
      }
      rc[k] = tbz_inflate_batch(ctxs[k], format, m, i_.data(), il.data(), o_.data(), ol.data(), rs.data());
      for (size_t q = 0; q < m; q++) results[idx[q]] = rs[q];
    });
  for (auto& t : th) t.join();
  for (size_t k = 0; k < n_ctx; k++)
    if (rc[k]) return rc[k];
  return 0;
}

// the same with the streams ALREADY RESIDENT on the devices (io-mmap.lisp:26-54: foreign / device pointers): context k
// decodes the n_streams[k] streams that live at d_ins[k] + in_offs[k][i] into d_outs[k] + out_offs[k][i] — per context
// the arguments of tbz_inflate_batch_device, all contexts at once, a host thread each.  Nothing crosses PCIe but the
// result records.
int tbz_inflate_batch_multi_device(tbz_ctx* const* ctxs, size_t n_ctx, int format, const size_t* n_streams,
                                   const void* const* d_ins, const uint64_t* const* in_offs, const uint64_t* const* in_lens,
                                   void* const* d_outs, const uint64_t* const* out_offs, const uint64_t* const* out_caps,
                                   tbz_result* const* results) {
  if (!ctxs || !n_ctx || !n_streams || !d_ins || !in_offs || !in_lens || !d_outs || !out_offs || !out_caps || !results)
    return TBZ_E_ARG;
  for (size_t k = 0; k < n_ctx; k++) {
    if (!ctxs[k]) return TBZ_E_ARG;
    for (size_t j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return TBZ_E_ARG;
  }
  std::vector<int> rc(n_ctx, 0);
  std::vector<std::thread> th;
  for (size_t k = 0; k < n_ctx; k++)
    th.emplace_back([&, k]() {
      if (!n_streams[k]) return;
      rc[k] = tbz_inflate_batch_device(ctxs[k], format, n_streams[k], d_ins[k], in_offs[k], in_lens[k], d_outs[k], out_offs[k],
                                       out_caps[k], results[k]);
    });
  for (auto& t : th) t.join();
  for (size_t k = 0; k < n_ctx; k++)
    if (rc[k]) return rc[k];
  return 0;
}

// ---- ONE flush-delimited stream over several decoders (the argument is in 3bz_amd/multi.py's header:
// a deflate stream may be entered at any block boundary, deflate.lisp:518-528, and the octet after 00 00 FF FF is one
// if the marker is real).  The plan and the verdict are host arithmetic over octets the host already holds and over the
// parts' result records, so that any host language can drive the parts — ranks of a torch.distributed job, the contexts
// of one process, or one context after another (tbz_inflate's pipelined path).
//
// plan: cuts[0] = 0 (start), cuts[r] = end of the first flush marker at or after the r-th equal share of the octets
// (in_len when there is none), cuts[n_parts] = in_len.  Part r is in[cuts[r], cuts[r+1]): part 0 in the stream's own
// format, the others as raw deflate.
// max_scan: how far behind a part's nominal start a marker is looked for (the C entry: to the end of the input; the
// pipelined host path: two parts' worth — a stream WITHOUT markers cost a scan of all its octets, 36 ms per GiB of
// no-flush text, before the ordinary path even started)
static int sharded_plan(const uint8_t* in, size_t in_len, size_t n_parts, uint64_t* cuts, uint64_t max_scan) {
  if (!cuts || !n_parts || (in_len && !in)) return TBZ_E_ARG;
  cuts[0] = 0;
  for (size_t r = 1; r < n_parts; r++) {
    const uint64_t share = (uint64_t)((unsigned __int128)in_len * r / n_parts);
    uint64_t p = std::max<uint64_t>(cuts[r - 1], share);
    const uint64_t stop = max_scan < in_len - std::min<uint64_t>(in_len, p) ? p + max_scan : in_len;
    uint64_t cut = in_len;
    while (p + 4 <= stop) {
      const uint8_t* q = (const uint8_t*)memchr(in + p, 0, stop - p - 3);
      if (!q) break;
      p = (uint64_t)(q - in);
      if (in[p + 1] == 0 && in[p + 2] == 0xff && in[p + 3] == 0xff) {
        cut = p + 4;
        break;
      }
      p++;
    }
    cuts[r] = cut;
  }
  cuts[n_parts] = in_len;
  return 0;
}
int tbz_inflate_sharded_plan(const uint8_t* in, size_t in_len, size_t n_parts, uint64_t* cuts) {
  return sharded_plan(in, in_len, n_parts, cuts, ~0ull);
}
}  // extern "C"

namespace tbz {
static uint32_t adler32_combine(uint32_t a1, uint32_t a2, uint64_t len2) {
  const uint32_t B = 65521;
  const uint32_t s1a = a1 & 0xffff, s2a = a1 >> 16, s1b = a2 & 0xffff, s2b = a2 >> 16;
  const uint32_t s1 = (s1a + s1b + B - 1) % B;
  const uint64_t rem = len2 % B;
  const uint32_t s2 = (uint32_t)((s2a + s2b + rem * ((s1a + B - 1) % B)) % B);
  return s1 | (s2 << 16);
}
static uint32_t crc32_combine(uint32_t c1, uint32_t c2, uint64_t len2) {
  // c1 * x^(8 len2) mod P (reflected, #xedb88320: checksums.lisp:177-193), by squaring
  auto mul = [](uint32_t a, uint32_t b) { return h_mulmod(a, b); };
  uint32_t pw = 0x80000000u, sq = 0x00800000u;  // x^0, x^8
  for (uint64_t n = len2; n; n >>= 1) {
    if (n & 1) pw = mul(sq, pw);
    sq = mul(sq, sq);
  }
  return mul(pw, c1) ^ c2;
}
}  // namespace tbz

extern "C" {
// verdict: recs[r] = what part r's decoder reported — its tbz_result and the checksum of its output continued from the
// format's initial value (adler32 from 1 / crc32 from 0; part 0's own checksum field serves).  All seams clean (every part
// but the last live one ran out of input exactly at its range's end, at a block boundary; the last one finished) =>
// returns 0 with *total, *check, *in_consumed filled and out_offs[r] = where part r's octets start in the whole; for a
// container format the trailer that follows the last part is read from `in` and compared.  Anything else => returns 1
// (decode the stream by the ordinary path: ITS statuses are the answer) and *why says which rule failed.
int tbz_inflate_sharded_verdict(int format, const uint8_t* in, size_t in_len, size_t n_parts, const uint64_t* cuts,
                                const tbz_result* recs, const uint32_t* part_check, uint64_t* out_offs, uint64_t* total,
                                uint32_t* check, uint64_t* in_consumed, int* why) {
  using namespace tbz;
  if (!cuts || !recs || !n_parts || format < 0 || format > 2 || (in_len && !in)) return TBZ_E_ARG;
  auto fail = [&](int w) {
    if (why) *why = w;
    return 1;
  };
  size_t last = 0;
  for (size_t r = 0; r < n_parts; r++)
    if (cuts[r + 1] > cuts[r] || r == 0) last = r;
  for (size_t r = 0; r <= last; r++) {
    const bool live = cuts[r + 1] > cuts[r] || r == 0;
    if (!live) continue;
    if (r != last && !(recs[r].status == TBZ_INPUT_UNDERRUN && recs[r].in_consumed == cuts[r + 1] - cuts[r])) return fail(1 + (int)r);
    if (r == last && recs[r].status != TBZ_FINISHED) return fail(-1);
  }
  uint64_t tot = 0;
  for (size_t r = 0; r < n_parts; r++) {
    if (out_offs) out_offs[r] = tot;
    const bool live = r <= last && (cuts[r + 1] > cuts[r] || r == 0);
    tot += live ? recs[r].out_len : 0;
  }
  uint64_t consumed = cuts[last] - cuts[0] + recs[last].in_consumed;
  uint32_t ck = 0;
  if (format != TBZ_FORMAT_DEFLATE && last != 0) {
    if (!part_check) return TBZ_E_ARG;
    bool first = true;
    for (size_t r = 0; r <= last; r++) {
      if (!(cuts[r + 1] > cuts[r] || r == 0)) continue;
      if (first) {
        ck = part_check[r];
        first = false;
      } else {
        ck = format == TBZ_FORMAT_ZLIB ? adler32_combine(ck, part_check[r], recs[r].out_len) : crc32_combine(ck, part_check[r], recs[r].out_len);
      }
    }
    const uint64_t at = cuts[last] + recs[last].in_consumed;  // the raw-deflate part ends at its final block: the trailer follows
    const uint64_t need = format == TBZ_FORMAT_ZLIB ? 4 : 8;
    if (at + need > cuts[n_parts]) return fail(-2);
    const uint8_t* t = in + at;
    const uint32_t stored = format == TBZ_FORMAT_ZLIB ? ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3]
                                                      : (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
    if (stored != ck) return fail(-3);
    consumed += need;
  } else if (format != TBZ_FORMAT_DEFLATE) {
    ck = part_check ? part_check[0] : (format == TBZ_FORMAT_ZLIB ? recs[0].adler32 : recs[0].crc32);
  }
  if (total) *total = tot;
  if (check) *check = ck;
  if (in_consumed) *in_consumed = consumed;
  if (why) *why = 0;
  return 0;
}

// ONE flush-delimited stream over the contexts of several devices, from one host process (the counterpart of
// 3bz_amd/multi.py:inflate_sharded for a host that is not a torch.distributed rank): plan, one host thread per context
// (its part staged to its device, decoded into device memory sized by K1, checksummed there), verdict, and — all seams
// clean — the parts copied into `out` at their offsets.  Anything not clean: context 0 decodes the whole stream by the
// ordinary path (tbz_inflate), so statuses and errors are the single-device ones.  *sharded says which it was.
int tbz_inflate_sharded_multi(tbz_ctx* const* ctxs, size_t n_ctx, int format, const uint8_t* in, size_t in_len, uint8_t* out,
                              size_t out_cap, tbz_result* res, int* sharded) {
  if (!ctxs || !n_ctx || !res || (in_len && !in) || (out_cap && !out) || format < 0 || format > 2) return TBZ_E_ARG;
  for (size_t k = 0; k < n_ctx; k++) {
    if (!ctxs[k]) return TBZ_E_ARG;
    for (size_t j = 0; j < k; j++)
      if (ctxs[j] == ctxs[k]) return TBZ_E_ARG;
  }
  if (sharded) *sharded = 0;
  std::vector<uint64_t> cuts(n_ctx + 1);
  int r = tbz_inflate_sharded_plan(in, in_len, n_ctx, cuts.data());
  if (r) return r;
  std::vector<tbz_result> recs(n_ctx);
  std::vector<uint32_t> cks(n_ctx, 0);
  std::vector<void*> bufs(n_ctx, nullptr);
  std::vector<int> rc(n_ctx, 0);
  for (auto& q : recs) memset(&q, 0, sizeof q);
  size_t live = 0;
  for (size_t k = 0; k < n_ctx; k++) live += cuts[k + 1] > cuts[k];
  if (n_ctx > 1 && live > 1) {
    std::vector<std::thread> th;
    for (size_t k = 0; k < n_ctx; k++)
      th.emplace_back([&, k]() {
        const uint64_t n = cuts[k + 1] - cuts[k];
        if (!n && k) { recs[k].status = TBZ_INPUT_UNDERRUN; return; }
        rc[k] = tbz_inflate_to_device(ctxs[k], k == 0 ? format : TBZ_FORMAT_DEFLATE, in + cuts[k], n, &bufs[k], &recs[k]);
        if (rc[k] || recs[k].status < 0 || format == TBZ_FORMAT_DEFLATE || !recs[k].out_len) return;
        if (format == TBZ_FORMAT_ZLIB) {
          uint32_t s1 = 0, s2 = 0;
          rc[k] = tbz_adler32_device(ctxs[k], bufs[k], recs[k].out_len, 1, 0, &s1, &s2);
          cks[k] = s1 | (s2 << 16);
        } else {
          rc[k] = tbz_crc32_device(ctxs[k], bufs[k], recs[k].out_len, 0, &cks[k]);
        }
      });
    for (auto& t : th) t.join();
    bool ok = true;
    for (size_t k = 0; k < n_ctx; k++) {
      ok = ok && rc[k] == 0;
      if (format == TBZ_FORMAT_ZLIB && recs[k].status >= 0 && !recs[k].out_len) cks[k] = 1;  // (adler32 of nothing)
    }
    std::vector<uint64_t> offs(n_ctx);
    uint64_t total = 0, consumed = 0;
    uint32_t check = 0;
    int why = 0;
    if (ok) ok = tbz_inflate_sharded_verdict(format, in, in_len, n_ctx, cuts.data(), recs.data(), cks.data(), offs.data(), &total,
                                             &check, &consumed, &why) == 0;
    if (ok && total <= out_cap) {
      std::vector<std::thread> cp;
      for (size_t k = 0; k < n_ctx; k++)
        if (bufs[k] && recs[k].out_len)
          cp.emplace_back([&, k]() { rc[k] = tbz_memcpy_d2h(ctxs[k], out + offs[k], bufs[k], recs[k].out_len); });
      for (auto& t : cp) t.join();
      for (size_t k = 0; k < n_ctx; k++) ok = ok && rc[k] == 0;
    } else {
      ok = false;
    }
    for (size_t k = 0; k < n_ctx; k++)
      if (bufs[k]) tbz_device_free(ctxs[k], bufs[k]);
    if (ok) {
      memset(res, 0, sizeof *res);
      res->status = TBZ_FINISHED;
      res->out_len = res->out_total = res->boundary_out = total;
      res->in_consumed = consumed;
      res->flags = 2u | (format != TBZ_FORMAT_DEFLATE ? 1u : 0u);
      for (size_t k = 0; k < n_ctx; k++) res->segments += recs[k].segments;
      if (format == TBZ_FORMAT_ZLIB) res->adler32 = res->trailer_check = check;
      if (format == TBZ_FORMAT_GZIP) {
        res->crc32 = res->trailer_check = check;
        const uint8_t* t = in + (consumed - 4);
        res->trailer_isize = (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
      }
      if (sharded) *sharded = 1;
      return 0;
    }
  }
  return tbz_inflate(ctxs[0], format, in, in_len, out, out_cap, res);
}

int tbz_inflate_size(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, tbz_result* res) {
  return stage_batch(ctx, format, 1, &in, &in_len, nullptr, nullptr, res, true);
}

int tbz_adler32_device(tbz_ctx* ctx, const void* d_buf, size_t len, uint32_t s1, uint32_t s2, uint32_t* out_s1,
                       uint32_t* out_s2) {
  using namespace tbz;
  if (!ctx || !out_s1 || !out_s2 || (len && !d_buf)) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> o{0}, l{len};
  std::vector<uint32_t> init{(s1 & 0xffff) | (s2 << 16)}, sums;
  int r = run_checksums(ctx, 1, d_buf, o, l, init, sums);
  if (r) return r;
  *out_s1 = sums[0] & 0xffff;
  *out_s2 = sums[0] >> 16;
  return 0;
}

int tbz_crc32_device(tbz_ctx* ctx, const void* d_buf, size_t len, uint32_t crc, uint32_t* out_crc) {
  using namespace tbz;
  if (!ctx || !out_crc || (len && !d_buf)) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  std::vector<uint64_t> o{0}, l{len};
  std::vector<uint32_t> init{crc}, sums;
  int r = run_checksums(ctx, 2, d_buf, o, l, init, sums);
  if (r) return r;
  *out_crc = sums[0];
  return 0;
}

int tbz_device_malloc(tbz_ctx* ctx, size_t bytes, void** d_ptr) {
  using namespace tbz;
  if (!ctx || !d_ptr) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  hipError_t e = hipMalloc(d_ptr, bytes ? bytes : 16);
  if (e != hipSuccess) {
    ctx->err = std::string("hipMalloc: ") + hipGetErrorString(e);
    return TBZ_E_NOMEM;
  }
  return 0;
}
int tbz_device_free(tbz_ctx* ctx, void* d_ptr) {
  using namespace tbz;
  if (!ctx) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  TBZ_HIP(hipFree(d_ptr));
  return 0;
}
int tbz_memcpy_h2d(tbz_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  using namespace tbz;
  if (!ctx) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  if (bytes) TBZ_HIP(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int tbz_memcpy_d2h(tbz_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  using namespace tbz;
  if (!ctx) return TBZ_E_ARG;
  TBZ_HIP(hipSetDevice(ctx->device));
  if (bytes) TBZ_HIP(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
  return 0;
}

}  // extern "C"
