// tests/emu/tbz_platform.hpp — TEST INFRASTRUCTURE.  CPU emulation of the primitives in
// 3bz_amd/csrc/tbz_platform.hpp so the UNCHANGED kernel + engine source can run under
// AddressSanitizer/UBSan on the build host (GPU ASan is not available on this pool).
//
// A 64-thread workgroup is emulated by 64 cooperative fibers that hand over to each other at every
// collective (ballot / shuffle / barrier); `__shared__` becomes function-local static storage (workgroups run
// one after another); the HIP runtime calls the engine makes are mapped to malloc/memcpy.
// Nothing here is linked into lib3bz_amd.so; the emulation library is only ever loaded by tests.
#ifndef TBZ_PLATFORM_HPP_INCLUDED
#define TBZ_PLATFORM_HPP_INCLUDED
#include <pthread.h>
#include <ucontext.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <vector>

#define TBZ_EMU 1
#if defined(__SANITIZE_ADDRESS__)
#include <sanitizer/common_interface_defs.h>
#endif
#define TBZ_WAVE 64
#define TBZ_DEV static inline
#define TBZ_DEV_NOINLINE static
#define TBZ_KERNEL static
#define TBZ_KERNEL_OCC(w) static
#define TBZ_SHARED static
#define TBZ_CONSTANT static const
#define TBZ_RESTRICT

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;

struct uint4 {
  u32 x, y, z, w;
};

namespace tbz_emu {
// A workgroup = W wavefronts (W = 1 .. 16) of 64 lanes = 64*W cooperative fibers on ONE OS thread.  Every
// wave-level collective (fence / ballot / shuffle) is "yield to the next lane of my wave"; because convergent
// code makes every lane of a wave execute the same sequence of collectives, one trip round the wave's ring is
// exactly one barrier.  A WORKGROUP barrier is a trip round the own wave followed by switches to the other
// waves until each has reached the same barrier.  Deterministic, and ~100x faster than OS threads on a
// std::barrier.
constexpr size_t kStack = 512 << 10;
constexpr int kMaxWaves = 16;
constexpr int kMaxLanes = 64 * kMaxWaves;
struct State {
  ucontext_t main_ctx;
  ucontext_t fib[kMaxLanes];
  char* stacks[kMaxLanes] = {};
  bool done[kMaxLanes];
  u64 nbar[kMaxLanes];
  u64 nwg[kMaxLanes];  // workgroup barriers passed, per lane
  int cur = 0;         // running fiber: wave = cur >> 6, lane = cur & 63
  int wcur[kMaxWaves] = {};     // fiber to resume when a wave is switched back in
  u64 wgcount[kMaxWaves] = {};  // workgroup barriers the wave has arrived at
  int wdone[kMaxWaves] = {};
  int nwaves = 1;
  u32 block = 0, nblocks = 0;
  const std::function<void()>* fn = nullptr;
  volatile u64 slot[kMaxLanes];
#if defined(__SANITIZE_ADDRESS__)
  void* fake[kMaxLanes + 1] = {};  // [kMaxLanes] = main
  const void* main_bottom = nullptr;
  size_t main_size = 0;
#endif
};
constexpr int kMain = kMaxLanes;
inline State& st() {
  static State s;
  return s;
}
inline bool strict() {
  static int s = -1;
  if (s < 0) s = getenv("TBZ_EMU_STRICT") ? 1 : 0;
  return s == 1;
}
#if defined(__SANITIZE_ADDRESS__)
inline void san_start(int from, int to) {
  State& s = st();
  const void* bottom = to == kMain ? s.main_bottom : (const void*)s.stacks[to];
  size_t size = to == kMain ? s.main_size : kStack;
  __sanitizer_start_switch_fiber(&s.fake[from], bottom, size);
}
inline void san_finish(int me) {
  State& s = st();
  const void* ob;
  size_t os;
  __sanitizer_finish_switch_fiber(s.fake[me], &ob, &os);
}
#else
inline void san_start(int, int) {}
inline void san_finish(int) {}
#endif
inline void switch_to(int me, int nx) {  // me/nx: fiber index, or kMain for the launcher
  State& s = st();
  if (me != kMain) s.wcur[me >> 6] = me;
  if (nx != kMain) {
    s.cur = nx;
    s.wcur[nx >> 6] = nx;
  }
  san_start(me, nx);
  swapcontext(me == kMain ? &s.main_ctx : &s.fib[me], nx == kMain ? &s.main_ctx : &s.fib[nx]);
  san_finish(me);
}
inline int next_alive(int me) {  // next unfinished lane of MY wave
  State& s = st();
  const int base = me & ~63;
  for (int k = 1; k <= 64; k++) {
    int c = base | ((me + k) & 63);
    if (!s.done[c]) return c;
  }
  return -1;
}
inline void barrier() {  // wave-level
  State& s = st();
  int me = s.cur;
  s.nbar[me]++;
  int nx = next_alive(me);
  if (nx < 0 || nx == me) return;
  switch_to(me, nx);
}
inline void wg_barrier() {
  State& s = st();
  barrier();  // every lane of my wave has arrived
  const int me = s.cur, w = me >> 6;
  s.nwg[me]++;
  if (s.wgcount[w] < s.nwg[me]) s.wgcount[w] = s.nwg[me];
  if (s.nwaves < 2) return;
  for (;;) {  // let every other wave catch up
    int o = -1;
    for (int k = 1; k < s.nwaves; k++) {
      const int c = (w + k) % s.nwaves;
      if (s.wgcount[c] < s.nwg[me] && s.wdone[c] < 64) {
        o = c;
        break;
      }
    }
    if (o < 0) break;
    switch_to(me, s.wcur[o]);
  }
}
inline void trampoline() {
  State& s = st();
  san_finish(s.cur);
  int me = s.cur;
  (*s.fn)();
  s.done[me] = true;
  s.wdone[me >> 6]++;
  const int base = me & ~63;
  for (int i = base; i < base + 64; i++)
    if (s.done[i] && s.nbar[i] != s.nbar[me]) {
      fprintf(stderr, "tbz_emu: lanes %d and %d executed a different number of collectives (%llu vs %llu): "
              "divergent barrier / ballot / shuffle\n", i, me, (unsigned long long)s.nbar[i],
              (unsigned long long)s.nbar[me]);
      abort();
    }
  int nx = next_alive(me);
  if (nx < 0) {  // my wave is finished: another wave that still runs, else back to the launcher
    nx = kMain;
    for (int k = 1; k < s.nwaves; k++) {
      const int o = ((me >> 6) + k) % s.nwaves;
      if (s.wdone[o] < 64) {
        nx = s.wcur[o];
        break;
      }
    }
  }
  if (nx != kMain) {
    s.cur = nx;
    s.wcur[nx >> 6] = nx;
  }
#if defined(__SANITIZE_ADDRESS__)
  {
    const void* bottom = nx == kMain ? s.main_bottom : (const void*)s.stacks[nx];
    size_t size = nx == kMain ? s.main_size : kStack;
    __sanitizer_start_switch_fiber(nullptr, bottom, size);  // this fiber is finished
  }
#endif
  setcontext(nx == kMain ? &s.main_ctx : &s.fib[nx]);
}
// run kernel body `fn` for `grid` workgroups of `threads` (a multiple of 64, up to 1024) lanes, one workgroup after another
inline std::recursive_mutex& launch_mutex() {
  static std::recursive_mutex m;
  return m;
}
inline void launch(u32 grid, const std::function<void()>& fn, int threads = 64) {
  if (grid == 0) return;
  // one emulated launch at a time: the lanes of a workgroup are fibers over process-wide state (tbz_inflate_batch_multi
  // drives two contexts from two host threads)
  std::lock_guard<std::recursive_mutex> lock(launch_mutex());
  State& s = st();
  for (int l = 0; l < threads; l++)
    if (!s.stacks[l]) s.stacks[l] = (char*)malloc(kStack);
#if defined(__SANITIZE_ADDRESS__)
  if (!s.main_bottom) {
    pthread_attr_t at;
    pthread_getattr_np(pthread_self(), &at);
    void* addr;
    size_t sz;
    pthread_attr_getstack(&at, &addr, &sz);
    pthread_attr_destroy(&at);
    s.main_bottom = addr;
    s.main_size = sz;
  }
#endif
  s.fn = &fn;
  s.nblocks = grid;
  s.nwaves = threads / 64;
  for (u32 b = 0; b < grid; b++) {
    s.block = b;
    for (int w = 0; w < kMaxWaves; w++) {
      s.wcur[w] = w * 64;
      s.wgcount[w] = 0;
      s.wdone[w] = w < s.nwaves ? 0 : 64;
    }
    for (int l = 0; l < threads; l++) {
      s.done[l] = false;
      s.nbar[l] = 0;
      s.nwg[l] = 0;
      getcontext(&s.fib[l]);
      s.fib[l].uc_stack.ss_sp = s.stacks[l];
      s.fib[l].uc_stack.ss_size = kStack;
      s.fib[l].uc_link = nullptr;
      makecontext(&s.fib[l], (void (*)())trampoline, 0);
    }
    switch_to(kMain, 0);
  }
  s.fn = nullptr;
  s.nwaves = 1;
}
inline u64 xchg(u64 v, u32 src) {  // value of lane `src` of MY wave
  State& s = st();
  s.slot[s.cur] = v;
  barrier();
  u64 r = s.slot[(s.cur & ~63) | (int)(src & 63)];
  barrier();
  return r;
}
}  // namespace tbz_emu

TBZ_DEV u32 tbz_lane() { return (u32)tbz_emu::st().cur & 63; }
TBZ_DEV u32 tbz_wave() { return (u32)tbz_emu::st().cur >> 6; }
TBZ_DEV void tbz_wg_barrier() { tbz_emu::wg_barrier(); }
TBZ_DEV void tbz_device_fence() {}
TBZ_DEV u32 tbz_block() { return tbz_emu::st().block; }
TBZ_DEV u32 tbz_nblocks() { return tbz_emu::st().nblocks; }
TBZ_DEV void tbz_sync() { tbz_emu::barrier(); }
TBZ_DEV u64 tbz_ballot(bool p) {
  tbz_emu::State& s = tbz_emu::st();
  s.slot[s.cur] = p ? 1 : 0;
  tbz_emu::barrier();
  u64 m = 0;
  const int wb = s.cur & ~63;
  for (int i = 0; i < 64; i++) m |= (u64)(s.slot[wb + i] & 1) << i;
  tbz_emu::barrier();
  return m;
}
TBZ_DEV u64 tbz_shfl64(u64 v, int src) { return tbz_emu::xchg(v, (u32)src); }
TBZ_DEV u64 tbz_shfl_up64(u64 v, unsigned d) {
  u32 l = (u32)tbz_emu::st().cur & 63;
  return tbz_emu::xchg(v, l >= d ? l - d : l);
}
TBZ_DEV u64 tbz_shfl_xor64(u64 v, int m) { return tbz_emu::xchg(v, ((u32)tbz_emu::st().cur & 63) ^ (u32)m); }
TBZ_DEV u32 tbz_shfl(u32 v, int src) { return (u32)tbz_emu::xchg(v, (u32)src); }
TBZ_DEV u32 tbz_shfl_up(u32 v, unsigned d) { return (u32)tbz_shfl_up64(v, d); }
TBZ_DEV u32 tbz_shfl_down(u32 v, unsigned d) {
  u32 l = (u32)tbz_emu::st().cur & 63;
  return (u32)tbz_emu::xchg(v, l + d < 64 ? l + d : l);
}
TBZ_DEV u32 tbz_shfl_xor(u32 v, int m) { return (u32)tbz_shfl_xor64(v, m); }
// readfirstlane: on the GPU every lane gets lane 0's value.  In strict mode verify that the value
// really is wave-uniform (costs two barriers per call, so off by default).
TBZ_DEV u32 tbz_uniform(u32 v) {
  if (tbz_emu::strict()) {
    u32 r = (u32)tbz_emu::xchg(v, 0);
    if (r != v) {
      fprintf(stderr, "tbz_emu: non-uniform value passed to tbz_uniform (lane %u: %u vs lane 0: %u)\n",
              (u32)tbz_emu::st().cur, v, r);
      abort();
    }
    return r;
  }
  return v;
}
TBZ_DEV u64 tbz_uniform64(u64 v) { return v; }
TBZ_DEV u32 tbz_popc64(u64 v) { return (u32)__builtin_popcountll(v); }
TBZ_DEV u32 tbz_ffs64(u64 v) { return (u32)__builtin_ffsll((long long)v); }
TBZ_DEV u32 tbz_brev32(u32 v) {
  v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
  v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
  v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
  return __builtin_bswap32(v);
}
TBZ_DEV u32 tbz_clz32(u32 v) { return v ? (u32)__builtin_clz(v) : 32; }
TBZ_DEV u32 tbz_atomic_add_lds(u32* p, u32 v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
TBZ_DEV u32 tbz_atomic_add_global(u32* p, u32 v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }

TBZ_DEV u64 tbz_qsad4(u64 s0, u32 ref) {
  u64 r = 0;
  for (int j = 0; j < 4; j++) {
    u32 sad = 0;
    for (int k = 0; k < 4; k++) {
      int a = (int)((s0 >> (8 * (j + k))) & 0xff), b = (int)((ref >> (8 * k)) & 0xff);
      sad += (u32)(a > b ? a - b : b - a);
    }
    r |= (u64)(sad & 0xffff) << (16 * j);
  }
  return r;
}
TBZ_DEV u32 tbz_pk_min_u16(u32 a, u32 b) {
  u32 lo = (a & 0xffff) < (b & 0xffff) ? (a & 0xffff) : (b & 0xffff);
  u32 hi = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
  return lo | (hi << 16);
}
TBZ_DEV u32 tbz_sum4_u8(u32 x, u32 acc) { return acc + (x & 0xff) + ((x >> 8) & 0xff) + ((x >> 16) & 0xff) + (x >> 24); }
TBZ_DEV u32 tbz_dot4_u8(u32 x, u32 w, u32 acc) {
  for (int k = 0; k < 4; k++) acc += ((x >> (8 * k)) & 0xff) * ((w >> (8 * k)) & 0xff);
  return acc;
}
TBZ_DEV u32 tbz_ld_agent(const u32* p) { return *p; }
TBZ_DEV u32 tbz_ld_agent(const u16* p) { return *p; }
TBZ_DEV u32 tbz_ld_agent(const u8* p) { return *p; }
TBZ_DEV u32 tbz_alignbit(u32 hi, u32 lo, u32 o) { return (u32)(((((u64)hi) << 32) | lo) >> (o & 31)); }
TBZ_DEV u32 tbz_bfe(u32 v, u32 off, u32 n) { return n ? ((v >> (off & 31)) & (n >= 32 ? ~0u : ((1u << n) - 1))) : 0; }
TBZ_DEV u32 tbz_readlane(u32 v, u32 i) { return (u32)tbz_emu::xchg(v, i); }
TBZ_DEV u32 tbz_wave_shr1(u32 v) {
  u32 l = (u32)tbz_emu::st().cur & 63;
  u32 r = (u32)tbz_emu::xchg(v, l ? l - 1 : 0);
  return l ? r : 0;
}
TBZ_DEV u32 tbz_wave_shl1(u32 v) {
  u32 l = (u32)tbz_emu::st().cur & 63;
  u32 r = (u32)tbz_emu::xchg(v, l < 63 ? l + 1 : 63);
  return l < 63 ? r : 0;
}
TBZ_DEV u32 tbz_wave_incl_scan_u32(u32 v) {
  u32 l = (u32)tbz_emu::st().cur & 63;
  for (u32 d = 1; d < 64; d <<= 1) {
    u32 t = (u32)tbz_emu::xchg(v, l >= d ? l - d : l);
    if (l >= d) v += t;
  }
  return v;
}

// (the GPU flavour reads past the CU's L1; here memory is memory)
TBZ_DEV u64 tbz_emu_ld64(const u8* p) { u64 v; memcpy(&v, p, 8); return v; }
struct tbz_u32x4 {
  u32 x, y, z, w;
};
TBZ_DEV tbz_u32x4 tbz_emu_ld128(const u8* p) { tbz_u32x4 v; memcpy(&v, p, 16); return v; }
TBZ_DEV void tbz_gload128x2(const u8* p0, const u8* p1, u64 m1, tbz_u32x4& a, tbz_u32x4& b) {
  a = tbz_emu_ld128(p0);
  if ((m1 >> (tbz_emu::st().cur & 63)) & 1) b = tbz_emu_ld128(p1);
}
TBZ_DEV void tbz_gload128x4(const u8* p0, const u8* p1, const u8* q0, const u8* q1, u64 m1, tbz_u32x4& a, tbz_u32x4& b,
                            tbz_u32x4& c, tbz_u32x4& d) {
  tbz_gload128x2(p0, p1, m1, a, b);
  tbz_gload128x2(q0, q1, m1, c, d);
}
TBZ_DEV void tbz_gload64x2(const u8* p0, const u8* p1, u64& a, u64& b) { a = tbz_emu_ld64(p0); b = tbz_emu_ld64(p1); }
TBZ_DEV void tbz_gload8x2(const u8* p0, const u8* p1, u32& a, u32& b) { a = *p0; b = *p1; }
TBZ_DEV void tbz_vm_drain() {}

#define TBZ_DYN_SHARED(T, name) static __attribute__((aligned(16))) T name[64 * 1024]
#define TBZ_LAUNCH_DYN(kernel, grid, lds_bytes, stream, ...) \
  tbz_emu::launch((u32)(grid), [&] { kernel(__VA_ARGS__); })
#define TBZ_LAUNCH(kernel, grid, stream, ...) \
  tbz_emu::launch((u32)(grid), [&] { kernel(__VA_ARGS__); })
#define TBZ_LAUNCH_DYN_WG(kernel, grid, threads, lds_bytes, stream, ...) \
  tbz_emu::launch((u32)(grid), [&] { kernel(__VA_ARGS__); }, (int)(threads))
#define TBZ_KERNEL_WG(threads, w) static
#define TBZ_LAUNCH_WG(kernel, grid, threads, stream, ...) \
  tbz_emu::launch((u32)(grid), [&] { kernel(__VA_ARGS__); }, (int)(threads))

// ---- the sliver of the HIP runtime the engine uses ----------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
struct tbz_emu_event {
  std::chrono::steady_clock::time_point t;
};
typedef tbz_emu_event* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
static inline const char* hipGetErrorString(hipError_t) { return "emulated HIP error"; }
static inline hipError_t hipGetDeviceCount(int* n) {
  *n = 1;
  return hipSuccess;
}
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) {
  *p = malloc(((n ? n : 1) + 31) & ~(size_t)15);  // device allocations are at least 16-octet granular
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
  memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) {
  memcpy(d, s, n);
  return hipSuccess;
}
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
  memset(d, v, n);
  return hipSuccess;
}
static inline hipError_t hipStreamCreate(hipStream_t* s) {
  *s = nullptr;
  return hipSuccess;
}
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }  // (launches run in issue order here)
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) {
  *e = new tbz_emu_event();
  return hipSuccess;
}
static inline hipError_t hipEventDestroy(hipEvent_t e) {
  delete e;
  return hipSuccess;
}
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) {
  e->t = std::chrono::steady_clock::now();
  return hipSuccess;
}
static inline hipError_t hipHostMalloc(void** p, size_t n) {
  *p = malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipHostFree(void* p) {
  free(p);
  return hipSuccess;
}
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
#endif  // TBZ_PLATFORM_HPP_INCLUDED
