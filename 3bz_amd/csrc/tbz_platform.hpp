// tbz_platform.hpp — the gfx950 flavour of the few primitives the kernels are written against.
//
// Every kernel in this engine is launched with 64-thread workgroups: one workgroup == one
// CDNA4 wavefront.  That makes `__syncthreads()` a single-wave s_barrier (an LDS ordering
// point, essentially free) and lets the cross-lane helpers below be plain wave intrinsics.
//
// tests/emu/ holds a second header of the same name that runs the identical kernel source on
// the CPU (64 host threads per workgroup) under AddressSanitizer — GPU ASan is not available
// on this pool.  That header is test infrastructure; the product is built from THIS file only.
#ifndef TBZ_PLATFORM_HPP_INCLUDED
#define TBZ_PLATFORM_HPP_INCLUDED
#include <hip/hip_runtime.h>

#include <cstdint>

#define TBZ_WAVE 64
#define TBZ_DEV __device__ __forceinline__
#define TBZ_DEV_NOINLINE __device__ __noinline__
#define TBZ_KERNEL extern "C" __global__ __launch_bounds__(64)
#define TBZ_KERNEL_OCC(w) extern "C" __global__ __launch_bounds__(64, w)  // at least w waves per SIMD
#define TBZ_SHARED __shared__
#define TBZ_CONSTANT __constant__
#define TBZ_RESTRICT __restrict__

using u8 = uint8_t;
using u16 = uint16_t;
using u32 = uint32_t;
using u64 = uint64_t;
using i32 = int32_t;
using i64 = int64_t;

TBZ_DEV u32 tbz_lane() { return threadIdx.x & 63; }   // lane within the wavefront
TBZ_DEV u32 tbz_wave() { return threadIdx.x >> 6; }   // wavefront within the workgroup (two-wave kernels)
// the real thing, for the kernels whose workgroup is two wavefronts: LDS drain + s_barrier
TBZ_DEV void tbz_wg_barrier() {
  // __syncthreads() would also drain vmcnt, i.e. wait for the token prefetch in flight; what the waves exchange
  // lives in LDS, so: LDS operations done (lgkmcnt(0); vmcnt / expcnt fields left at "don't wait"), then s_barrier
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// agent-scope release + acquire: this workgroup's stores reach L2 and its L1 lines are dropped, so that octets it
// stored earlier are read back fresh (tbz_k6_window; cost per use: MI355X_MICROARCH.md, __threadfence row)
TBZ_DEV void tbz_device_fence() { __threadfence(); }
TBZ_DEV u32 tbz_block() { return blockIdx.x; }
TBZ_DEV u32 tbz_nblocks() { return gridDim.x; }
// Workgroup == one wavefront, and a wave's LDS (and vector-memory) instructions execute in issue order, so
// "all lanes' earlier accesses are visible to all lanes' later ones" needs no s_barrier and no s_waitcnt:
// a wavefront-scope fence (compiler ordering only) is the whole synchronisation.  In particular it does
// not drain vmcnt, so prefetches and stores stay in flight across it.
TBZ_DEV void tbz_sync() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
TBZ_DEV u64 tbz_ballot(bool p) { return __ballot(p); }
TBZ_DEV u32 tbz_shfl(u32 v, int src) { return (u32)__shfl((int)v, src, 64); }
TBZ_DEV u32 tbz_shfl_up(u32 v, unsigned d) { return (u32)__shfl_up((int)v, d, 64); }
TBZ_DEV u32 tbz_shfl_down(u32 v, unsigned d) { return (u32)__shfl_down((int)v, d, 64); }
TBZ_DEV u32 tbz_shfl_xor(u32 v, int m) { return (u32)__shfl_xor((int)v, m, 64); }
TBZ_DEV u64 tbz_shfl64(u64 v, int src) {
  u32 lo = tbz_shfl((u32)v, src), hi = tbz_shfl((u32)(v >> 32), src);
  return ((u64)hi << 32) | lo;
}
TBZ_DEV u64 tbz_shfl_up64(u64 v, unsigned d) {
  u32 lo = tbz_shfl_up((u32)v, d), hi = tbz_shfl_up((u32)(v >> 32), d);
  return ((u64)hi << 32) | lo;
}
TBZ_DEV u64 tbz_shfl_xor64(u64 v, int m) {
  u32 lo = tbz_shfl_xor((u32)v, m), hi = tbz_shfl_xor((u32)(v >> 32), m);
  return ((u64)hi << 32) | lo;
}
// value of the first active lane, as a wave-uniform (SGPR) value: lets the compiler keep the
// sequential part of the decoders on the scalar unit
TBZ_DEV u32 tbz_uniform(u32 v) { return (u32)__builtin_amdgcn_readfirstlane((int)v); }
TBZ_DEV u64 tbz_uniform64(u64 v) {
  return ((u64)tbz_uniform((u32)(v >> 32)) << 32) | tbz_uniform((u32)v);
}
TBZ_DEV u32 tbz_popc64(u64 v) { return (u32)__popcll(v); }
TBZ_DEV u32 tbz_ffs64(u64 v) { return (u32)__ffsll((long long)v); }  // 1-based, 0 if none
TBZ_DEV u32 tbz_brev32(u32 v) { return __brev(v); }
TBZ_DEV u32 tbz_clz32(u32 v) { return (u32)__clz((int)v); }
TBZ_DEV u32 tbz_atomic_add_lds(u32* p, u32 v) { return atomicAdd(p, v); }
TBZ_DEV u32 tbz_atomic_add_global(u32* p, u32 v) { return atomicAdd(p, v); }

// four sums of absolute differences of the 4 octets `ref` against s0's octets [j, j+4), j = 0..3, as
// 4 x u16 (v_qsad_pk_u16_u8): a zero field is an exact 4-octet match at offset j
TBZ_DEV u64 tbz_qsad4(u64 s0, u32 ref) { return __builtin_amdgcn_qsad_pk_u16_u8(s0, ref, 0ull); }
// per-halfword unsigned minimum (v_pk_min_u16)
typedef unsigned short tbz_us2 __attribute__((ext_vector_type(2)));
TBZ_DEV u32 tbz_pk_min_u16(u32 a, u32 b) {
  tbz_us2 r = __builtin_elementwise_min(__builtin_bit_cast(tbz_us2, a), __builtin_bit_cast(tbz_us2, b));
  return __builtin_bit_cast(u32, r);
}
// acc + sum of the four octets of x (v_sad_u8 against zero); acc + sum of x's octets times w's octets (v_dot4_u32_u8)
TBZ_DEV u32 tbz_sum4_u8(u32 x, u32 acc) { return __builtin_amdgcn_sad_u8(x, 0u, acc); }
TBZ_DEV u32 tbz_dot4_u8(u32 x, u32 w, u32 acc) { return __builtin_amdgcn_udot4(x, w, acc, false); }
// 32 bits of the 64-bit value hi:lo starting at bit o (o < 32): one v_alignbit_b32
TBZ_DEV u32 tbz_alignbit(u32 hi, u32 lo, u32 o) { return __builtin_amdgcn_alignbit(hi, lo, o); }
// n bits of v starting at bit off (n = 0 gives 0): one v_bfe_u32
TBZ_DEV u32 tbz_bfe(u32 v, u32 off, u32 n) { return __builtin_amdgcn_ubfe(v, off, n); }
// value of lane `i` (i wave-uniform): v_readlane, no LDS round trip
TBZ_DEV u32 tbz_readlane(u32 v, u32 i) { return (u32)__builtin_amdgcn_readlane((int)v, (int)i); }
// whole-wave shift by one lane through DPP (no LDS): shr1: lane i <- lane i-1 (lane 0 <- 0);
// shl1: lane i <- lane i+1 (lane 63 <- 0)
TBZ_DEV u32 tbz_wave_shr1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xf, 0xf, false); }
TBZ_DEV u32 tbz_wave_shl1(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xf, 0xf, false); }
// wave64 inclusive prefix sum in 7 DPP adds (row_shr 1/2/3/4/8, row_bcast 15/31)
TBZ_DEV u32 tbz_wave_incl_scan_u32(u32 x) {
  int v0 = (int)x;
  int v = v0 + __builtin_amdgcn_update_dpp(0, v0, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v0, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v0, 0x113, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xe, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xc, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return (u32)v;
}

// ---- coherent read-back of octets this workgroup stored earlier (the ring kernels' far matches).  `sc1` loads are served by
// L2, past the CU's vector L1, which is never refreshed by stores (MI355X_MICROARCH.md, inter-workgroup visibility: a line
// read once while partly written would be served stale).  The loads and their wait are ONE asm statement: the compiler
// knows nothing of a load issued from inline asm, and a register copy it placed between the load and a separate wait
// would copy the register's old contents.  Addresses need no alignment (gfx950 global accesses are octet-addressed).
typedef u32 tbz_u32x4 __attribute__((ext_vector_type(4)));
// 16 octets at p0 for every active lane, 16 more at p1 for the lanes of mask m1 (the others' `b` is left undefined)
TBZ_DEV void tbz_gload128x2(const u8* p0, const u8* p1, u64 m1, tbz_u32x4& a, tbz_u32x4& b) {
  u64 keep;
  asm volatile(
      "global_load_dwordx4 %0, %3, off sc1\n\ts_mov_b64 %2, exec\n\ts_and_b64 exec, exec, %5\n\t"
      "global_load_dwordx4 %1, %4, off sc1\n\ts_mov_b64 exec, %2\n\ts_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&s"(keep)
      : "v"(p0), "v"(p1), "s"(m1)
      : "memory", "scc");  // (s_and_b64 writes SCC; EXEC is saved and restored inside the statement)
}
// the same over two planes (p: octets, q: marks)
TBZ_DEV void tbz_gload128x4(const u8* p0, const u8* p1, const u8* q0, const u8* q1, u64 m1, tbz_u32x4& a, tbz_u32x4& b,
                            tbz_u32x4& c, tbz_u32x4& d) {
  u64 keep;
  asm volatile(
      "global_load_dwordx4 %0, %5, off sc1\n\tglobal_load_dwordx4 %2, %7, off sc1\n\t"
      "s_mov_b64 %4, exec\n\ts_and_b64 exec, exec, %9\n\t"
      "global_load_dwordx4 %1, %6, off sc1\n\tglobal_load_dwordx4 %3, %8, off sc1\n\t"
      "s_mov_b64 exec, %4\n\ts_waitcnt vmcnt(0)"
      : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d), "=&s"(keep)
      : "v"(p0), "v"(p1), "v"(q0), "v"(q1), "s"(m1)
      : "memory", "scc");
}
// eight octets of each of two addresses
TBZ_DEV void tbz_gload64x2(const u8* p0, const u8* p1, u64& a, u64& b) {
  asm volatile("global_load_dwordx2 %0, %2, off sc1\n\tglobal_load_dwordx2 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b)
               : "v"(p0), "v"(p1)
               : "memory");
}
TBZ_DEV void tbz_gload8x2(const u8* p0, const u8* p1, u32& a, u32& b) {
  asm volatile("global_load_ubyte %0, %2, off sc1\n\tglobal_load_ubyte %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(a), "=&v"(b)
               : "v"(p0), "v"(p1)
               : "memory");
}
// loads that see what ANOTHER lane of this workgroup stored to memory earlier in the kernel (the gang kernels' parked
// canonical lists: stored once per block, read back by the exact step): agent-scope loads are served by L2, past the CU's
// vector L1, which a store does not refresh — a line read while it held the previous block's lists would be served stale
TBZ_DEV u32 tbz_ld_agent(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TBZ_DEV u32 tbz_ld_agent(const u16* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
TBZ_DEV u32 tbz_ld_agent(const u8* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// every vector-memory operation this wave has issued (stores too) is complete
TBZ_DEV void tbz_vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#define TBZ_DYN_SHARED(T, name) extern __shared__ __attribute__((aligned(16))) T name[]
#define TBZ_LAUNCH_DYN(kernel, grid, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(64), (lds_bytes), (stream), __VA_ARGS__)
#define TBZ_LAUNCH(kernel, grid, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(64), 0, (stream), __VA_ARGS__)
#define TBZ_LAUNCH_DYN_WG(kernel, grid, threads, lds_bytes, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(threads), (lds_bytes), (stream), __VA_ARGS__)
#define TBZ_KERNEL_WG(threads, w) extern "C" __global__ __launch_bounds__(threads, w)
#define TBZ_LAUNCH_WG(kernel, grid, threads, stream, ...) \
  hipLaunchKernelGGL(kernel, dim3((unsigned)(grid)), dim3(threads), 0, (stream), __VA_ARGS__)
#endif  // TBZ_PLATFORM_HPP_INCLUDED
