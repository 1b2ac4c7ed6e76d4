// issue_roof.hip — what one gfx950 SIMD issues per cycle, measured.
//
// The decode-stage kernels (K1g, K2) are neither HBM- nor MFMA-bound: they are streams of integer VALU, SALU and LDS
// instructions.  `roofline.issue` in bench.py prices them against the rate at which a SIMD can ISSUE such a stream;
// round 3 assumed "one wave-instruction per SIMD per 4 cycles whatever its type" and measured frac > 1, i.e. the
// assumed denominator was not the limit.  This program measures the limit instead:
//
//   mix V   independent v_alignbit_b32 / v_bfe_u32 / v_cndmask_b32 (the bit-reader's instructions), 8 chains per lane
//   mix VS  the same with one SALU (s_and_b64 / s_cselect_b64 on dead SGPRs) after every two VALU (K1g's 2:1 mix)
//   mix VSL VS plus one ds_read_u16 per 40 VALU, not waited for inside the unrolled body (lgkmcnt drained per iteration)
//   mix D   ONE dependent chain of the V instructions (what a wave pays when nothing in it is independent)
//   lat L   a dependent chain of ds_read_u16 -> address -> ds_read_u16 (LDS round trip as the decoder's table walk sees it),
//           conflict-free (all lanes one address, broadcast) and with 64 random u16 entries of a 1 KB table
//
// each at 1, 2, 3, 4, 6, 8 waves per SIMD on every SIMD of the chip (ONE workgroup of 256 x w threads per CU — 150 KB of dynamic
// LDS keep a second one out — so that every SIMD holds exactly w waves; HW_ID is recorded to check the spread).  Time is taken INSIDE the kernel with
// s_memtime (guide: one tick = one shader cycle): cycles per wave = mean over waves of (end - start); the rate printed is
//     wave-instructions per SIMD per cycle = W x instructions per wave / cycles per wave.
//
// build:  hipcc --offload-arch=gfx950 -O2 tools/issue_roof.hip -o tools/issue_roof
// run  :  tools/issue_roof [json-out]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_));               \
      exit(2);                                                                                 \
    }                                                                                          \
  } while (0)

struct Rec {
  uint64_t t0, t1;
  uint32_t hwid, sink;
};

__device__ __forceinline__ uint64_t now() {
  uint64_t t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
__device__ __forceinline__ uint32_t hwid() {
  uint32_t h;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
  return h;
}

// 40 VALU per macro: 5 rounds over 8 independent registers (alignbit, bfe, cndmask, alignbit, bfe)
#define V8_A "v_alignbit_b32 %0, %0, %0, 3\n\tv_bfe_u32 %0, %0, 1, 31\n\tv_cndmask_b32 %0, %0, %0, vcc\n\tv_alignbit_b32 %0, %0, %0, 9\n\t" \
             "v_bfe_u32 %0, %0, 1, 31\n\tv_alignbit_b32 %0, %0, %0, 13\n\tv_cndmask_b32 %0, %0, %0, vcc\n\tv_alignbit_b32 %0, %0, %0, 17\n\t"
// (independent version: each instruction reads and writes ITS OWN register only)
#define V8_I1 "v_alignbit_b32 %0, %0, %0, 3\n\tv_alignbit_b32 %1, %1, %1, 5\n\tv_alignbit_b32 %2, %2, %2, 7\n\tv_alignbit_b32 %3, %3, %3, 9\n\t" \
              "v_alignbit_b32 %4, %4, %4, 11\n\tv_alignbit_b32 %5, %5, %5, 13\n\tv_alignbit_b32 %6, %6, %6, 15\n\tv_alignbit_b32 %7, %7, %7, 17\n\t"
#define V8_I2 "v_bfe_u32 %0, %0, 1, 31\n\tv_bfe_u32 %1, %1, 1, 31\n\tv_bfe_u32 %2, %2, 1, 31\n\tv_bfe_u32 %3, %3, 1, 31\n\t" \
              "v_bfe_u32 %4, %4, 1, 31\n\tv_bfe_u32 %5, %5, 1, 31\n\tv_bfe_u32 %6, %6, 1, 31\n\tv_bfe_u32 %7, %7, 1, 31\n\t"
#define V8_I3 "v_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %1, %1, %1, vcc\n\tv_cndmask_b32 %2, %2, %2, vcc\n\tv_cndmask_b32 %3, %3, %3, vcc\n\t" \
              "v_cndmask_b32 %4, %4, %4, vcc\n\tv_cndmask_b32 %5, %5, %5, vcc\n\tv_cndmask_b32 %6, %6, %6, vcc\n\tv_cndmask_b32 %7, %7, %7, vcc\n\t"
#define V40 V8_I1 V8_I2 V8_I3 V8_I1 V8_I2

// the same 40 VALU with an SALU after every second one (20 SALU)
#define S1 "s_and_b64 s[20:21], s[22:23], s[24:25]\n\t"
#define S2 "s_cselect_b64 s[26:27], s[22:23], s[24:25]\n\t"
#define VS8_1 "v_alignbit_b32 %0, %0, %0, 3\n\tv_alignbit_b32 %1, %1, %1, 5\n\t" S1 "v_alignbit_b32 %2, %2, %2, 7\n\tv_alignbit_b32 %3, %3, %3, 9\n\t" S2 \
              "v_alignbit_b32 %4, %4, %4, 11\n\tv_alignbit_b32 %5, %5, %5, 13\n\t" S1 "v_alignbit_b32 %6, %6, %6, 15\n\tv_alignbit_b32 %7, %7, %7, 17\n\t" S2
#define VS8_2 "v_bfe_u32 %0, %0, 1, 31\n\tv_bfe_u32 %1, %1, 1, 31\n\t" S1 "v_bfe_u32 %2, %2, 1, 31\n\tv_bfe_u32 %3, %3, 1, 31\n\t" S2 \
              "v_bfe_u32 %4, %4, 1, 31\n\tv_bfe_u32 %5, %5, 1, 31\n\t" S1 "v_bfe_u32 %6, %6, 1, 31\n\tv_bfe_u32 %7, %7, 1, 31\n\t" S2
#define VS8_3 "v_cndmask_b32 %0, %0, %0, vcc\n\tv_cndmask_b32 %1, %1, %1, vcc\n\t" S1 "v_cndmask_b32 %2, %2, %2, vcc\n\tv_cndmask_b32 %3, %3, %3, vcc\n\t" S2 \
              "v_cndmask_b32 %4, %4, %4, vcc\n\tv_cndmask_b32 %5, %5, %5, vcc\n\t" S1 "v_cndmask_b32 %6, %6, %6, vcc\n\tv_cndmask_b32 %7, %7, %7, vcc\n\t" S2
#define VS60 VS8_1 VS8_2 VS8_3 VS8_1 VS8_2

// the same 40 VALU with an SALU after EVERY one (40 SALU): K2's mix
#define VT(i) "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 3\n\t" S1 "v_bfe_u32 %" #i ", %" #i ", 1, 31\n\t" S2 \
              "v_cndmask_b32 %" #i ", %" #i ", %" #i ", vcc\n\t" S1 "v_alignbit_b32 %" #i ", %" #i ", %" #i ", 9\n\t" S2 \
              "v_bfe_u32 %" #i ", %" #i ", 1, 31\n\t" S1
#define VS80 VT(0) VT(1) VT(2) VT(3) VT(4) VT(5) VT(6) VT(7)

#define REGS "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define SCLOB "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc"

template <int MIX>
__global__ __launch_bounds__(1024) void k_issue(Rec* rec, int iters, uint32_t seed) {
  extern __shared__ uint16_t tab[];
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = (uint16_t)((i * 2654435761u + seed) >> 7);
  __syncthreads();
  uint32_t a0 = lane + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
  uint32_t l0 = 0, la = (lane * 37u & 511u) * 2;
  (void)l0;
  (void)la;
  const uint64_t t0 = now();
  for (int it = 0; it < iters; it++) {
    if (MIX == 0) {  // V: 80 independent VALU
      asm volatile(V40 V40 : REGS : : "vcc");
    } else if (MIX == 1) {  // VS: 80 VALU + 40 SALU
      asm volatile(VS60 VS60 : REGS : : "vcc", SCLOB);
    } else if (MIX == 2) {  // VSL: 80 VALU + 40 SALU + 2 ds_read_u16 (one per 40 VALU), drained once per iteration
      asm volatile("ds_read_u16 %8, %9\n\t" VS60 "ds_read_u16 %8, %9 offset:128\n\t" VS60 "s_waitcnt lgkmcnt(0)"
                   : REGS, "=&v"(l0)
                   : "v"(la)
                   : "vcc", SCLOB, "memory");
    } else if (MIX == 6) {  // VS11: 80 VALU + 80 SALU
      asm volatile(VS80 VS80 : REGS : : "vcc", SCLOB);
    } else if (MIX == 3) {  // D: one dependent chain, 80 VALU
      asm volatile(V8_A V8_A V8_A V8_A V8_A V8_A V8_A V8_A V8_A V8_A : REGS : : "vcc");
    }
  }
  const uint64_t t1 = now();
  if (lane == 0) {
    Rec r;
    r.t0 = t0;
    r.t1 = t1;
    r.hwid = hwid();
    r.sink = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ l0;
    rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
}

// LDS round trip: 16 dependent ds_read_u16 per iteration; RANDOM = each lane walks its own pseudo-random path through a
// 512-entry u16 table (the decoder's gathers), else all lanes read one address (broadcast, conflict-free)
template <bool RANDOM>
__global__ __launch_bounds__(1024) void k_ldslat(Rec* rec, int iters, uint32_t seed) {
  extern __shared__ uint16_t tab[];
  const uint32_t lane = threadIdx.x & 63;
  // a permutation-ish walk: entry i holds the next index (odd multiplier mod 512)
  for (uint32_t i = threadIdx.x; i < 512; i += blockDim.x) tab[i] = (uint16_t)(RANDOM ? ((i * 205u + 77u + seed) & 511u) : 0u);
  __syncthreads();
  uint32_t idx = RANDOM ? ((lane * 37u + seed) & 511u) : 0u;
  const uint64_t t0 = now();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int k = 0; k < 16; k++) {
      uint32_t v;
      asm volatile("v_lshlrev_b32 %0, 1, %1\n\tds_read_u16 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(idx) : "memory");
      idx = v;
    }
  }
  const uint64_t t1 = now();
  if (lane == 0) {
    Rec r;
    r.t0 = t0;
    r.t1 = t1;
    r.hwid = hwid();
    r.sink = idx + (uint32_t)(uintptr_t)tab;
    rec[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = r;
  }
}

struct Row {
  std::string mix;
  int waves;
  double cyc_per_wave, inst_per_wave, rate, valu_rate, ticks_per_ns, rate_ns, valu_rate_ns;
  int simds_used, max_per_simd;
};

// HW_ID (gfx9 family): wave_id [3:0], simd_id [5:4], pipe_id [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
static uint32_t simd_key(uint32_t hw) { return (hw >> 4) & 0xfffu & ~0xcu; }  // simd, cu, sh, se (pipe dropped)

template <typename K>
static void launch(K kern, int grid, int threads, size_t lds, Rec* d_rec, int iters) {
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, 0, d_rec, iters, 1u);
}

int main(int argc, char** argv) {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount, simds = cus * 4;
  printf("device: %s, %d CUs (%d SIMDs), clockRate %d kHz\n", p.gcnArchName, cus, simds, p.clockRate);
  printf("placement: ONE workgroup of 256*w threads per CU (150 KB of dynamic LDS: a second one does not fit), its 4*w waves\n"
         "           round-robin over the CU's 4 SIMDs -> exactly w waves per SIMD; W = 6, 8: TWO workgroups of 768 / 1024 threads (75 KB each)\n");
  const int iters = 4000;
  std::vector<Row> rows;
  Rec* d_rec;
  const int maxw = simds * 8;
  CK(hipMalloc((void**)&d_rec, sizeof(Rec) * maxw));
  std::vector<Rec> h(maxw);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  struct Shape { int W, threads, per_cu; } shapes[] = {{1, 256, 1}, {2, 512, 1}, {3, 768, 1}, {4, 1024, 1}, {6, 768, 2}, {8, 1024, 2}};
  struct Mix {
    const char* name;
    int id;
    double inst, valu;  // per iteration per wave
  } mixes[] = {{"V   (80 independent VALU)", 0, 80, 80},
               {"VS  (80 VALU + 40 SALU)", 1, 120, 80},
               {"VS11 (80 VALU + 80 SALU)", 6, 160, 80},
               {"VSL (80 VALU + 40 SALU + 2 ds_read_u16)", 2, 122, 80},
               {"D   (80 VALU, one dependent chain)", 3, 80, 80},
               {"L0  (16 x {v_lshlrev, ds_read_u16, wait}, dependent, broadcast)", 4, 32, 16},
               {"L1  (the same, 64 random entries of a 1 KB table)", 5, 32, 16}};
  for (const Mix& m : mixes) {
    for (const Shape& sh : shapes) {
      const int grid = cus * sh.per_cu, nw = grid * (sh.threads / 64);
      const size_t lds = sh.per_cu == 1 ? 150u << 10 : 75u << 10;
      float ms = 0;
      for (int rep = 0; rep < 2; rep++) {  // first repetition warms up
        CK(hipEventRecord(e0, 0));
        switch (m.id) {
          case 0: launch(k_issue<0>, grid, sh.threads, lds, d_rec, iters); break;
          case 1: launch(k_issue<1>, grid, sh.threads, lds, d_rec, iters); break;
          case 2: launch(k_issue<2>, grid, sh.threads, lds, d_rec, iters); break;
          case 3: launch(k_issue<3>, grid, sh.threads, lds, d_rec, iters); break;
          case 6: launch(k_issue<6>, grid, sh.threads, lds, d_rec, iters); break;
          case 4: launch(k_ldslat<false>, grid, sh.threads, lds, d_rec, iters); break;
          default: launch(k_ldslat<true>, grid, sh.threads, lds, d_rec, iters); break;
        }
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&ms, e0, e1));
      }
      CK(hipMemcpy(h.data(), d_rec, sizeof(Rec) * nw, hipMemcpyDeviceToHost));
      double sum = 0;
      std::vector<uint32_t> keys;
      for (int i = 0; i < nw; i++) {
        sum += (double)(h[i].t1 - h[i].t0);
        keys.push_back(simd_key(h[i].hwid));
      }
      // (HW_ID does not carry the XCD: 8 XCDs x 32 CUs give every key 8 x W waves when the spread is even)
      std::sort(keys.begin(), keys.end());
      int used = 0, mx = 0;
      for (size_t a = 0; a < keys.size();) {
        size_t b = a;
        while (b < keys.size() && keys[b] == keys[a]) b++;
        used++;
        mx = std::max(mx, (int)(b - a));
        a = b;
      }
      Row r;
      r.mix = m.name;
      r.waves = sh.W;
      r.cyc_per_wave = sum / nw;
      r.inst_per_wave = m.inst * iters;
      r.rate = sh.W * r.inst_per_wave / r.cyc_per_wave;
      r.valu_rate = sh.W * m.valu * iters / r.cyc_per_wave;
      r.ticks_per_ns = r.cyc_per_wave / (ms * 1e6);  // (the kernel is its waves' lifetime: all start together)
      // the same rates on the WALL clock (HIP events around the launch): the shader clock gives way under load — 2.4 GHz
      // with one wave per SIMD, 1.1 - 1.5 GHz with every SIMD full of vector work — and a kernel's duration is wall time
      r.rate_ns = sh.W * r.inst_per_wave / (ms * 1e6);
      r.valu_rate_ns = sh.W * m.valu * iters / (ms * 1e6);
      r.simds_used = used;
      r.max_per_simd = mx;
      rows.push_back(r);
      printf("%-66s W=%d  ticks/inst/wave %6.2f  wave-inst/SIMD/tick %6.3f  kernel %.3f ms -> %.2f ticks/ns  wave-inst/SIMD/ns %6.3f (first type alone %6.3f)  HW_ID keys %d, most waves on one %d\n",
             m.name, sh.W, r.cyc_per_wave / r.inst_per_wave, r.rate, ms, r.ticks_per_ns, r.rate_ns, r.valu_rate_ns, used, mx);
    }
  }
  if (argc > 1) {
    FILE* f = fopen(argv[1], "w");
    if (f) {
      fprintf(f, "{\"device\": \"%s\", \"cus\": %d, \"tick\": \"s_memtime\", \"rows\": [\n", p.gcnArchName, cus);
      for (size_t i = 0; i < rows.size(); i++)
        fprintf(f, "  {\"mix\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_wave\": %.0f, \"inst_per_wave\": %.0f, \"wave_inst_per_simd_tick\": %.4f, \"first_type_per_simd_tick\": %.4f, \"ticks_per_ns\": %.3f, \"wave_inst_per_simd_ns\": %.4f, \"first_type_per_simd_ns\": %.4f}%s\n",
                rows[i].mix.c_str(), rows[i].waves, rows[i].cyc_per_wave, rows[i].inst_per_wave, rows[i].rate, rows[i].valu_rate, rows[i].ticks_per_ns,
                rows[i].rate_ns, rows[i].valu_rate_ns, i + 1 < rows.size() ? "," : "");
      fprintf(f, "]}\n");
      fclose(f);
    }
  }
  CK(hipFree(d_rec));
  return 0;
}
