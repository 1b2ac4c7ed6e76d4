// tbz_kernels.hpp — the device side of the inflate engine (hand-written for gfx950 / CDNA4).
//
// All kernels run 64-thread workgroups (one wavefront each).  None of this is GEMM-shaped: it is
// integer / byte work bounded by LDS latency (Huffman decode) or HBM bandwidth (scan, LZ77 flush,
// checksums), so there is no MFMA here by design.
//
//   K0  tbz_k0_scan_count / tbz_k0_scan_offsets / tbz_k0_scan_emit
//         find 00 00 FF FF flush markers (candidate independent-segment starts), coalesced reads,
//         wave prefix-sums for an ordered compaction.
//   K1  tbz_k1_huff_decode
//         one wave per item: bit reader + dynamic-header parse + LDS-resident lookup tables
//         (replaces deflate.lisp:518-702 and huffman-tree.lisp:99-218).  Emits u16 tokens, counts
//         output octets, reports where it landed.  Needs no history, so every item is independent.
//   K2  tbz_k2_lz77
//         one wave per group: token stream -> 36 KiB LDS ring (32 KiB history + one batch span) ->
//         16-byte coalesced HBM stores (replaces copy-history / out-byte, deflate.lisp:233-359).
//   K4  tbz_k4_adler_partial / tbz_k4_adler_combine   (checksums.lisp:18-62)
//   K5  tbz_k5_crc_partial / tbz_k5_crc_combine       (checksums.lisp:177-210)
#pragma once
#include "tbz_platform.hpp"
#include "tbz_structs.hpp"

namespace tbz {

// ------------------------------------------------------------------------------------------------
// error / status codes used on the device (mirror include/tbz_amd.h)
// ------------------------------------------------------------------------------------------------
constexpr i32 E_BTYPE = -1, E_STORED_LEN = -2, E_OVERSUB = -3, E_INCOMPLETE = -4, E_REPEAT_NO_PREV = -5,
              E_REPEAT_OVERRUN = -6, E_INVALID_CODE = -7, E_ZLIB_HEADER = -9, E_ZLIB_DICT = -10,
              E_GZIP_MAGIC = -12, E_GZIP_METHOD = -13, E_GZIP_FLAGS = -14, E_GZIP_HCRC = -15;

// ------------------------------------------------------------------------------------------------
// small wave helpers
// ------------------------------------------------------------------------------------------------
TBZ_DEV u32 wave_incl_scan_u32(u32 v) {
  const u32 lane = tbz_lane();
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    u32 t = tbz_shfl_up(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
TBZ_DEV u64 wave_incl_scan_u64(u64 v) {
  const u32 lane = tbz_lane();
#pragma unroll
  for (u32 d = 1; d < 64; d <<= 1) {
    u64 t = tbz_shfl_up64(v, d);
    if (lane >= d) v += t;
  }
  return v;
}
TBZ_DEV u64 wave_sum_u64(u64 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += tbz_shfl_xor64(v, m);
  return v;
}
TBZ_DEV u32 wave_xor_u32(u32 v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v ^= tbz_shfl_xor(v, m);
  return v;
}

// 4 octets at byte index `idx` of a 4-byte aligned word array, never touching a word at or beyond
// `nwords` (words past the end read as 0)
TBZ_DEV u32 load_u32_unaligned(const u32* TBZ_RESTRICT w, u64 idx, u64 nwords) {
  u64 wi = idx >> 2;
  u32 sh = (u32)(idx & 3) * 8;
  u32 a = wi < nwords ? w[wi] : 0;
  if (sh == 0) return a;
  u32 b = (wi + 1) < nwords ? w[wi + 1] : 0;
  return (a >> sh) | (b << (32 - sh));
}

// ================================================================================================
// K0 — marker scan.  A marker is the byte AFTER `00 00 FF FF` (the LEN/NLEN of an empty stored
// block, which zlib emits for Z_SYNC_FLUSH / Z_FULL_FLUSH).  3bz has no counterpart: it is strictly
// sequential (:block-end -> :start-of-block, deflate.lisp:719-722).
// Work split: tile = 16 KiB of one stream = 16 rows of 1 KiB; lane j of row r owns the 16 positions
// starting at tile + r*1024 + j*16, so a row is one fully coalesced 1 KiB read.
// ================================================================================================
struct K0Params {
  const u8* in_base;
  const u64* str_off;     // per stream: byte offset / length in in_base
  const u64* str_len;
  const u32* tile_first;  // n_streams+1 prefix of tiles per stream
  u32 n_streams;
  u32 n_tiles;
  u32* tile_counts;       // [n_tiles]   (count pass out)
  u32* tile_offsets;      // [n_tiles+1] (scan out; [n_tiles] = total)
  u64* markers;           // emit pass out, globally sorted
};

TBZ_DEV u32 k0_find_stream(const K0Params& P, u32 tile) {
  u32 lo = 0, hi = P.n_streams;  // tile_first[lo] <= tile < tile_first[hi]
  while (hi - lo > 1) {
    u32 mid = (lo + hi) >> 1;
    if (P.tile_first[mid] <= tile) lo = mid; else hi = mid;
  }
  return lo;
}

// 16-bit mask of marker-pattern starts among the 16 positions [p0, p0+16) owned by this lane
TBZ_DEV u32 k0_row_mask(const u32* TBZ_RESTRICT w, u64 mis, u64 p0, u64 s_begin, u64 s_end, u64 nwords) {
  // position p is a hit iff bytes p..p+3 = 00 00 FF FF, p >= s_begin and p + 4 < s_end
  if (p0 >= s_end) return 0;
  u64 a = p0 + mis;  // byte index from the aligned base
  u64 wi = a >> 2;
  u32 W[6];
#pragma unroll
  for (int k = 0; k < 6; k++) W[k] = (wi + k) < nwords ? w[wi + k] : 0;
  u32 m = 0;
  u32 b0 = (u32)(a & 3);
#pragma unroll
  for (u32 k = 0; k < 16; k++) {
    u32 b = b0 + k;
    u32 sh = (b & 3) * 8;
    u32 lo = W[b >> 2], hi = W[(b >> 2) + 1];
    u32 v = sh ? ((lo >> sh) | (hi << (32 - sh))) : lo;
    if (v == 0xFFFF0000u && p0 + k + 4 < s_end) m |= 1u << k;
  }
  (void)s_begin;
  return m;
}

struct K0Tile {
  const u32* w;
  u64 mis, nwords, s_begin, s_end, t_begin;
};
TBZ_DEV K0Tile k0_tile(const K0Params& P) {
  K0Tile T;
  u32 tile = tbz_block();
  u32 s = k0_find_stream(P, tile);
  u64 off = P.str_off[s], len = P.str_len[s];
  uintptr_t base = (uintptr_t)P.in_base;
  T.mis = base & 3;
  T.w = (const u32*)(base - T.mis);
  T.s_begin = off;
  T.s_end = off + len;
  T.nwords = (T.mis + T.s_end + 3) >> 2;
  T.t_begin = off + (u64)(tile - P.tile_first[s]) * SCAN_TILE;
  return T;
}

TBZ_KERNEL void tbz_k0_scan_count(K0Params P) {
  K0Tile T = k0_tile(P);
  const u32 lane = tbz_lane();
  u32 cnt = 0;
  for (u32 r = 0; r < SCAN_TILE / 1024; r++) {
    u64 p0 = T.t_begin + r * 1024 + lane * 16;
    cnt += __builtin_popcount(k0_row_mask(T.w, T.mis, p0, T.s_begin, T.s_end, T.nwords));
  }
  u32 tot = (u32)wave_sum_u64(cnt);
  if (lane == 0) P.tile_counts[tbz_block()] = tot;
}

// single-wave exclusive scan over tile counts
TBZ_KERNEL void tbz_k0_scan_offsets(K0Params P) {
  const u32 lane = tbz_lane();
  u32 carry = 0;
  for (u32 i = 0; i < P.n_tiles; i += 64) {
    u32 v = (i + lane) < P.n_tiles ? P.tile_counts[i + lane] : 0;
    u32 inc = wave_incl_scan_u32(v);
    if ((i + lane) < P.n_tiles) P.tile_offsets[i + lane] = carry + inc - v;
    carry += tbz_shfl(inc, 63);
  }
  if (lane == 0) P.tile_offsets[P.n_tiles] = carry;
}

TBZ_KERNEL void tbz_k0_scan_emit(K0Params P) {
  K0Tile T = k0_tile(P);
  const u32 lane = tbz_lane();
  u32 base = P.tile_offsets[tbz_block()];
  for (u32 r = 0; r < SCAN_TILE / 1024; r++) {
    u64 p0 = T.t_begin + r * 1024 + lane * 16;
    u32 m = k0_row_mask(T.w, T.mis, p0, T.s_begin, T.s_end, T.nwords);
    u32 c = __builtin_popcount(m);
    u32 inc = wave_incl_scan_u32(c);
    u32 o = base + inc - c;
    while (m) {
      u32 k = __builtin_ctz(m);
      m &= m - 1;
      P.markers[o++] = p0 + k + 4;
    }
    base += tbz_shfl(inc, 63);
  }
}

// ================================================================================================
// K1 — Huffman decode to tokens
// ================================================================================================

// RFC 1951 tables (restated natively; the reference keeps them merged in constants.lisp:41-61)
TBZ_CONSTANT u16 c_len_base[32] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59,
                                   67, 83, 99, 115, 131, 163, 195, 227, 258, 0, 0, 0};
TBZ_CONSTANT u8 c_len_extra[32] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3,
                                   4, 4, 4, 4, 5, 5, 5, 5, 0, 0, 0, 0};
TBZ_CONSTANT u16 c_dist_base[32] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769,
                                    1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577, 0, 0};
TBZ_CONSTANT u8 c_dist_extra[32] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8,
                                    9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0};
TBZ_CONSTANT u8 c_cl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// lookup-table entry (u32), one flat root table per alphabet + canonical fallback for long codes:
//   [3:0]  code length L          [7:4]  extra-bit count X      [9:8] kind
//   [12]   "long code" flag (kind = SPECIAL): the code is longer than the root index
//   [31:16] value: literal octet / length base / distance base / code-length symbol
constexpr u32 K_LIT = 0, K_BASE = 1, K_EOB = 2, K_SPECIAL = 3;
constexpr u32 ENTRY_INVALID = K_SPECIAL << 8;                 // unassigned bit pattern (hole in an incomplete code)
constexpr u32 ENTRY_LONG = (K_SPECIAL << 8) | (1u << 12);     // code longer than the root index
constexpr u32 ENTRY_BADSYM = (K_SPECIAL << 8) | (1u << 13);   // | L: a coded symbol that may not be used
constexpr int ROOT_LIT = 10, ROOT_DIST = 8, ROOT_CL = 7;
enum { ALPHA_LITLEN = 0, ALPHA_DIST = 1, ALPHA_CL = 2 };

struct CanonMeta {  // per alphabet, for the long-code fallback
  u16 first[16];    // first canonical code of each length
  u16 count[16];
  u16 offs[16];     // index of the first symbol of each length in `sorted`
  u32 min_len;      // shortest code length = width of the reference's root table (huffman-tree.lisp:144)
};

TBZ_DEV u32 make_entry(int alpha, u32 sym, u32 len) {
  if (alpha == ALPHA_LITLEN) {
    if (sym < 256) return (sym << 16) | (K_LIT << 8) | len;
    if (sym == 256) return (K_EOB << 8) | len;
    if (sym <= 285) {
      u32 k = sym - 257;
      return ((u32)c_len_base[k] << 16) | (K_BASE << 8) | ((u32)c_len_extra[k] << 4) | len;
    }
    return ENTRY_BADSYM | len;  // 286/287 take part in the code but may not be used (huffman-tree.lisp:176-177)
  }
  if (alpha == ALPHA_DIST) {
    if (sym <= 29) return ((u32)c_dist_base[sym] << 16) | (K_BASE << 8) | ((u32)c_dist_extra[sym] << 4) | len;
    return ENTRY_BADSYM | len;  // 30/31 (huffman-tree.lisp:172-175)
  }
  return (sym << 16) | (K_BASE << 8) | len;
}

// Wave-parallel canonical-Huffman table build.  `lens` (LDS) holds n code lengths 0..15.
// Same acceptance rules as build-tree-part (huffman-tree.lisp:112-122): over-subscribed -> error;
// incomplete -> error unless at most one symbol is coded; all-zero -> table of INVALID entries.
// All lanes must call this together; the return value is wave-uniform.
template <int ROOT>
TBZ_DEV i32 build_table(const u8* lens, u32 n, int alpha, u32* tbl, u16* sorted, CanonMeta* meta, u32* cnt) {
  const u32 lane = tbz_lane();
  tbz_sync();
  if (lane < 16) cnt[lane] = 0;
  tbz_sync();
  for (u32 i = lane; i < n; i += 64) {
    u32 l = lens[i];
    if (l) tbz_atomic_add_lds(&cnt[l], 1);
  }
  tbz_sync();
  u32 c[16], first[16], offs[16];
  u32 used = 0, code = 0, off = 0, min_len = 0;
  i32 left = 1;
  i32 err = 0;
  c[0] = 0;
  first[0] = 0;
  offs[0] = 0;
#pragma unroll
  for (int L = 1; L < 16; L++) {
    c[L] = tbz_uniform(cnt[L]);
    left <<= 1;
    if ((i32)c[L] > left && !err) err = E_OVERSUB;
    left -= (i32)c[L];
    used += c[L];
    if (c[L] && !min_len) min_len = L;
    code = (code + c[L - 1]) << 1;
    first[L] = code;
    offs[L] = off;
    off += c[L];
  }
  if (!err && left > 0 && used > 1) err = E_INCOMPLETE;
  if (err) return err;
  for (u32 i = lane; i < (1u << ROOT); i += 64) tbl[i] = ENTRY_INVALID;
  if (lane < 16) {
    meta->first[lane] = (u16)first[lane];
    meta->count[lane] = (u16)c[lane];
    meta->offs[lane] = (u16)offs[lane];
  }
  if (lane == 0) meta->min_len = min_len;
  tbz_sync();
  if (used == 0) return 0;
  u32 run[16];
#pragma unroll
  for (int L = 0; L < 16; L++) run[L] = 0;
  const u64 lt = (1ull << lane) - 1;
  for (u32 base = 0; base < n; base += 64) {
    u32 i = base + lane;
    u32 l = i < n ? lens[i] : 0;
    u32 rank = 0, fc = 0, so = 0;
#pragma unroll
    for (int L = 1; L < 16; L++) {
      if (c[L]) {  // wave-uniform
        u64 m = tbz_ballot(l == (u32)L);
        if (l == (u32)L) {
          rank = run[L] + tbz_popc64(m & lt);
          fc = first[L];
          so = offs[L];
        }
        run[L] += tbz_popc64(m);
      }
    }
    if (l) {
      u32 cd = fc + rank;
      sorted[so + rank] = (u16)i;
      u32 rev = tbz_brev32(cd) >> (32 - l);
      if (l <= (u32)ROOT) {
        u32 ent = make_entry(alpha, i, l);
        for (u32 k = rev; k < (1u << ROOT); k += (1u << l)) tbl[k] = ent;
      } else {
        tbl[rev & ((1u << ROOT) - 1)] = ENTRY_LONG;
      }
    }
  }
  tbz_sync();
  return 0;
}

// canonical decode of a code longer than the root index (rare path)
template <int ROOT>
TBZ_DEV u32 decode_long(u32 peek, int alpha, const u16* sorted, const CanonMeta* meta) {
  u32 r = tbz_brev32(peek);
  for (u32 L = ROOT + 1; L <= 15; L++) {
    u32 cd = r >> (32 - L);
    u32 rel = cd - (u32)meta->first[L];
    if (rel < (u32)meta->count[L]) {
      u32 sym = sorted[(u32)meta->offs[L] + rel];
      return tbz_uniform(make_entry(alpha, sym, L));
    }
  }
  return ENTRY_INVALID;
}

// How many bits the reference has to see before it reaches the invalid node an entry stands for:
// a coded-but-forbidden symbol needs its whole code; a hole in an incomplete (single-code) tree is
// found in the root table, i.e. after `min_len` bits (huffman-tree.lisp:144,:212-217).
TBZ_DEV u32 special_bits(u32 ent, const CanonMeta* meta) {
  return (ent & (1u << 13)) ? (ent & 15) : tbz_uniform(meta->min_len);
}

// LSB-first bit reader over global memory (deflate.lisp:140-231 restated for a wave: the state is
// wave-uniform so it can live in SGPRs; 32-bit words are fetched one ahead of use).
struct BitReader {
  const u32* w;
  u64 bias;    // bits between the aligned word base and in_base
  u64 nwords;  // words that contain stream octets
  u32 tail_mask;
  u64 pos;     // bit position relative to in_base
  u64 wi;
  u32 lo, hi, nx, o;
};
TBZ_DEV u32 br_word(const BitReader& b, u64 i) {
  u32 v = 0;
  if (i < b.nwords) {
    v = b.w[i];
    if (i + 1 == b.nwords) v &= b.tail_mask;
  }
  return tbz_uniform(v);
}
TBZ_DEV void br_init(BitReader& b, const u8* in_base, u64 end_byte) {
  uintptr_t base = (uintptr_t)in_base;
  u64 mis = base & 3;
  b.w = (const u32*)(base - mis);
  b.bias = mis * 8;
  u64 endb = mis + end_byte;
  b.nwords = (endb + 3) >> 2;
  u32 tail = (u32)(endb & 3);
  b.tail_mask = tail ? ((1u << (8 * tail)) - 1) : 0xFFFFFFFFu;
}
TBZ_DEV void br_seek(BitReader& b, u64 pos) {
  u64 a = pos + b.bias;
  b.pos = pos;
  b.wi = a >> 5;
  b.o = (u32)(a & 31);
  b.lo = br_word(b, b.wi);
  b.hi = br_word(b, b.wi + 1);
  b.nx = br_word(b, b.wi + 2);
}
TBZ_DEV u32 br_peek(const BitReader& b) { return (u32)(((((u64)b.hi) << 32) | b.lo) >> b.o); }
TBZ_DEV void br_skip(BitReader& b, u32 n) {  // n <= 32
  b.o += n;
  b.pos += n;
  if (b.o >= 32) {
    b.o -= 32;
    b.lo = b.hi;
    b.hi = b.nx;
    b.wi += 1;
    b.nx = br_word(b, b.wi + 2);
  }
}
TBZ_DEV u32 bfe(u32 v, u32 off, u32 n) { return (v >> off) & ((1u << n) - 1); }

struct K1Params {
  const u8* in_base;
  u16* tok;  // token pool: item tokens start at tok[item.start_bit]; words written never exceed bits consumed
  const Item* items;
  SegResult* res;
  const u64* markers;
  u32 n_markers;
  u32 n_items;
};

constexpr u32 K1_STAGE = 1024;  // u16 words staged in LDS between coalesced flushes

struct K1Lds {
  u32 lit[1 << ROOT_LIT];
  u32 dist[1 << ROOT_DIST];
  u32 cl[1 << ROOT_CL];
  u16 stage[K1_STAGE];
  u16 lit_sorted[288];
  u16 dist_sorted[32];
  u16 cl_sorted[32];
  CanonMeta meta[3];
  u32 cnt[16];
  u8 lens[320];
  u8 cl_lens[32];
};

struct K1State {
  BitReader br;
  u64 end_bit, limit_bit;
  u64 produced;   // octets the tokens emitted so far produce
  u64 tok_flushed;  // words already in global memory
  u32 staged;
  u32 deficit;
  u64 tok_base;   // index of this item's first token word
  u64 fail_pos;   // bit position reported on underrun / overshoot
};

TBZ_DEV void k1_flush(K1Lds& S, K1State& st, u16* tok) {
  tbz_sync();
  for (u32 i = tbz_lane(); i < st.staged; i += 64) tok[st.tok_base + st.tok_flushed + i] = S.stage[i];
  tbz_sync();
  st.tok_flushed += st.staged;
  st.staged = 0;
}
TBZ_DEV void k1_emit(K1Lds& S, K1State& st, u16* tok, u32 v) {
  S.stage[st.staged++] = (u16)v;
  if (st.staged >= K1_STAGE - 2) k1_flush(S, st, tok);
}

// availability / landing-limit test after consuming bits for a token or header field that started
// at `p0`.  Underrun wins (deflate.lisp:399-427: the whole symbol is pushed back and re-read).
#define K1_CHECK(p0)                                  \
  do {                                                \
    if (st.br.pos > st.end_bit) {                     \
      st.fail_pos = (p0);                             \
      return SEG_UNDERRUN;                            \
    }                                                 \
    if (st.br.pos > st.limit_bit) {                   \
      st.fail_pos = (p0);                             \
      return SEG_OVERSHOOT;                           \
    }                                                 \
  } while (0)

// fixed (BTYPE=1) code lengths: huffman-tree.lisp:89-97
TBZ_DEV i32 k1_build_fixed(K1Lds& S) {
  const u32 lane = tbz_lane();
  tbz_sync();
  for (u32 i = lane; i < 320; i += 64) {
    u8 l;
    if (i < 144) l = 8;
    else if (i < 256) l = 9;
    else if (i < 280) l = 7;
    else if (i < 288) l = 8;
    else l = 5;
    S.lens[i] = l;
  }
  tbz_sync();
  i32 e = build_table<ROOT_LIT>(S.lens, 288, ALPHA_LITLEN, S.lit, S.lit_sorted, &S.meta[0], S.cnt);
  if (e) return e;
  return build_table<ROOT_DIST>(S.lens + 288, 32, ALPHA_DIST, S.dist, S.dist_sorted, &S.meta[1], S.cnt);
}

// :dynamic-huffman-block … :dht-len-table-data (deflate.lisp:577-669)
TBZ_DEV i32 k1_dynamic_header(K1Lds& S, K1State& st) {
  const u64 p0 = st.br.pos;
  u32 pk = br_peek(st.br);
  u32 hlit = (pk & 31) + 257, hdist = ((pk >> 5) & 31) + 1, hclen = ((pk >> 10) & 15) + 4;
  br_skip(st.br, 14);
  K1_CHECK(p0);
  tbz_sync();
  if (tbz_lane() < 32) S.cl_lens[tbz_lane()] = 0;
  tbz_sync();
  for (u32 i = 0; i < hclen; i++) {
    u32 v = br_peek(st.br) & 7;
    br_skip(st.br, 3);
    S.cl_lens[c_cl_order[i]] = (u8)v;
  }
  K1_CHECK(p0);
  i32 e = build_table<ROOT_CL>(S.cl_lens, 19, ALPHA_CL, S.cl, S.cl_sorted, &S.meta[2], S.cnt);
  if (e) return e;
  const u32 n = hlit + hdist;
  u32 i = 0, last = 0xff;
  while (i < n) {
    const u64 ps = st.br.pos;
    pk = br_peek(st.br);
    u32 ent = tbz_uniform(S.cl[pk & ((1u << ROOT_CL) - 1)]);
    u32 L = ent & 15, sym = ent >> 16;
    bool bad = ((ent >> 8) & 3) == K_SPECIAL;
    u32 x = 0, xb = 0;
    if (!bad) {
      xb = sym == 16 ? 2 : sym == 17 ? 3 : sym == 18 ? 7 : 0;
      x = bfe(pk, L, xb);
      br_skip(st.br, L + xb);
      K1_CHECK(ps);
    } else {
      // an unassigned pattern: error unless the input ends inside it
      if (st.br.pos + special_bits(ent, &S.meta[2]) > st.end_bit) {
        st.fail_pos = ps;
        return SEG_UNDERRUN;
      }
      return E_INVALID_CODE;
    }
    if (sym < 16) {
      S.lens[i++] = (u8)sym;
      last = sym;
    } else {
      u32 rep, val;
      if (sym == 16) {
        if (last >= 16) return E_REPEAT_NO_PREV;
        rep = 3 + x;
        val = last;
      } else {
        rep = (sym == 17 ? 3 : 11) + x;
        val = 0;
        last = 0;
      }
      if (i + rep > n) return E_REPEAT_OVERRUN;
      for (u32 k = 0; k < rep; k++) S.lens[i + k] = (u8)val;
      i += rep;
    }
  }
  tbz_sync();
  e = build_table<ROOT_LIT>(S.lens, hlit, ALPHA_LITLEN, S.lit, S.lit_sorted, &S.meta[0], S.cnt);
  if (e) return e;
  return build_table<ROOT_DIST>(S.lens + hlit, hdist, ALPHA_DIST, S.dist, S.dist_sorted, &S.meta[1], S.cnt);
}

// :decode-compressed-data (deflate.lisp:673-702): returns 0 at end-of-block
TBZ_DEV i32 k1_decode_block(K1Lds& S, K1State& st, u16* tok) {
  for (;;) {
    const u64 p0 = st.br.pos;
    u32 pk = br_peek(st.br);
    u32 ent = tbz_uniform(S.lit[pk & ((1u << ROOT_LIT) - 1)]);
    if (((ent >> 8) & 3) == K_SPECIAL) {
      if (ent & (1u << 12)) ent = decode_long<ROOT_LIT>(pk, ALPHA_LITLEN, S.lit_sorted, &S.meta[0]);
      if (((ent >> 8) & 3) == K_SPECIAL) {
        if (st.br.pos + special_bits(ent, &S.meta[0]) > st.end_bit) {
          st.fail_pos = p0;
          return SEG_UNDERRUN;
        }
        return E_INVALID_CODE;
      }
    }
    u32 L = ent & 15, X = (ent >> 4) & 15, kind = (ent >> 8) & 3, val = ent >> 16;
    if (kind == K_LIT) {
      br_skip(st.br, L);
      K1_CHECK(p0);
      k1_emit(S, st, tok, val);
      st.produced += 1;
    } else if (kind == K_BASE) {
      u32 len = val + bfe(pk, L, X);
      br_skip(st.br, L + X);
      u32 pd = br_peek(st.br);
      u32 de = tbz_uniform(S.dist[pd & ((1u << ROOT_DIST) - 1)]);
      if (((de >> 8) & 3) == K_SPECIAL) {
        if (de & (1u << 12)) de = decode_long<ROOT_DIST>(pd, ALPHA_DIST, S.dist_sorted, &S.meta[1]);
        if (((de >> 8) & 3) == K_SPECIAL) {
          if (st.br.pos + special_bits(de, &S.meta[1]) > st.end_bit) {
            st.fail_pos = p0;
            return SEG_UNDERRUN;
          }
          return E_INVALID_CODE;
        }
      }
      u32 DL = de & 15, DX = (de >> 4) & 15;
      u32 dist = (de >> 16) + bfe(pd, DL, DX);
      br_skip(st.br, DL + DX);
      K1_CHECK(p0);
      if ((u64)dist > st.produced) {
        u32 d = dist - (u32)st.produced;
        if (d > st.deficit) st.deficit = d;
      }
      k1_emit(S, st, tok, 0x8000u | (len - 3));
      k1_emit(S, st, tok, dist - 1);
      st.produced += len;
    } else {  // end of block
      br_skip(st.br, L);
      K1_CHECK(p0);
      return 0;
    }
  }
}

// reflected CRC-32 of one octet, bitwise (only for the optional gzip header crc16, gzip.lisp:244-255)
TBZ_DEV u32 crc_bitwise(u32 crc, u32 byte) {
  crc ^= byte;
  for (int k = 0; k < 8; k++) crc = (crc & 1) ? (0xedb88320u ^ (crc >> 1)) : (crc >> 1);
  return crc;
}
TBZ_DEV i32 k1_byte(K1State& st, u32* out, u64 p0) {
  *out = br_peek(st.br) & 0xff;
  br_skip(st.br, 8);
  if (st.br.pos > st.end_bit) {
    st.fail_pos = p0;
    return SEG_UNDERRUN;
  }
  return 0;
}
// container headers: zlib.lisp:14-37,:110-128  gzip.lisp:113-266
TBZ_DEV i32 k1_container_header(K1State& st, u32 fmt) {
  const u64 p0 = st.br.pos;
  i32 e;
  u32 b0, b1, t;
  if (fmt == 1) {
    if ((e = k1_byte(st, &b0, p0))) return e;
    if ((e = k1_byte(st, &b1, p0))) return e;
    if (((b0 * 256 + b1) % 31) != 0) return E_ZLIB_HEADER;
    if ((b0 & 15) != 8) return E_ZLIB_HEADER;
    if ((b0 >> 4) > 7) return E_ZLIB_HEADER;
    if (b1 & 0x20) return E_ZLIB_DICT;
    return 0;
  }
  if (fmt == 2) {
    u32 crc = 0xffffffffu;
    u32 hdr[10];
    for (int i = 0; i < 2; i++) {
      if ((e = k1_byte(st, &hdr[i], p0))) return e;
      crc = crc_bitwise(crc, hdr[i]);
    }
    if (hdr[0] != 0x1f || hdr[1] != 0x8b) return E_GZIP_MAGIC;
    for (int i = 2; i < 4; i++) {
      if ((e = k1_byte(st, &hdr[i], p0))) return e;
      crc = crc_bitwise(crc, hdr[i]);
    }
    if (hdr[2] != 8) return E_GZIP_METHOD;
    u32 flg = hdr[3];
    if (flg >> 5) return E_GZIP_FLAGS;
    for (int i = 4; i < 10; i++) {
      if ((e = k1_byte(st, &hdr[i], p0))) return e;
      crc = crc_bitwise(crc, hdr[i]);
    }
    if (flg & 4) {
      if ((e = k1_byte(st, &b0, p0))) return e;
      if ((e = k1_byte(st, &b1, p0))) return e;
      crc = crc_bitwise(crc_bitwise(crc, b0), b1);
      u32 xlen = b0 | (b1 << 8);
      for (u32 i = 0; i < xlen; i++) {
        if ((e = k1_byte(st, &t, p0))) return e;
        crc = crc_bitwise(crc, t);
      }
    }
    for (int f = 8; f <= 16; f <<= 1) {  // FNAME, FCOMMENT: zero-terminated
      if (flg & f) {
        for (;;) {
          if ((e = k1_byte(st, &t, p0))) return e;
          crc = crc_bitwise(crc, t);
          if (t == 0) break;
        }
      }
    }
    if (flg & 2) {
      if ((e = k1_byte(st, &b0, p0))) return e;
      if ((e = k1_byte(st, &b1, p0))) return e;
      if ((b0 | (b1 << 8)) != ((crc ^ 0xffffffffu) & 0xffff)) return E_GZIP_HCRC;
    }
    return 0;
  }
  return 0;
}

TBZ_KERNEL void tbz_k1_huff_decode(K1Params P) {
  TBZ_SHARED K1Lds S;
  if (tbz_block() >= P.n_items) return;
  const u32 lane = tbz_lane();
  const Item it = P.items[tbz_block()];
  const u32 fmt = (it.flags >> ITEM_FMT_SHIFT) & 3;
  const bool fixup = (it.flags & ITEM_FIXUP) != 0;
  K1State st;
  br_init(st.br, P.in_base, it.end_byte);
  br_seek(st.br, it.start_bit);
  st.end_bit = it.end_byte * 8;
  st.limit_bit = fixup ? ~0ull : it.limit_bit;
  st.produced = 0;
  st.tok_flushed = 0;
  st.staged = 0;
  st.deficit = 0;
  st.tok_base = it.start_bit;
  st.fail_pos = it.start_bit;

  i32 status = 0;
  u32 land = 0xFFFFFFFFu, tr0 = 0, tr1 = 0, tr_have = 0;
  u64 blk_pos = it.start_bit, blk_prod = 0, blk_tok = 0;
  int tables = 0;  // 0 none, 1 fixed, 2 dynamic

  if (it.flags & ITEM_HEAD) status = k1_container_header(st, fmt);

  while (status == 0) {
    blk_pos = st.br.pos;
    blk_prod = st.produced;
    blk_tok = st.tok_flushed + st.staged;
    u32 pk = br_peek(st.br);
    br_skip(st.br, 3);
    if (st.br.pos > st.end_bit) { st.fail_pos = blk_pos; status = SEG_UNDERRUN; break; }
    if (st.br.pos > st.limit_bit) { status = SEG_OVERSHOOT; break; }
    const u32 bfinal = pk & 1, btype = (pk >> 1) & 3;
    if (btype == 0) {  // :uncompressed-block / :copy-block (deflate.lisp:532-573)
      br_skip(st.br, (u32)((0 - st.br.pos) & 7));
      const u64 ph = st.br.pos;
      u32 ln = br_peek(st.br);
      br_skip(st.br, 32);
      if (st.br.pos > st.end_bit) { st.fail_pos = ph; status = SEG_UNDERRUN; break; }
      u32 LEN = ln & 0xffff, NLEN = ln >> 16;
      if (NLEN != ((~LEN) & 0xffff)) { status = E_STORED_LEN; break; }
      u64 byte0 = st.br.pos >> 3;
      if ((byte0 + LEN) * 8 > st.limit_bit) { status = SEG_OVERSHOOT; break; }
      u64 avail = it.end_byte - byte0;
      u32 ncopy = avail < LEN ? (u32)avail : LEN;
      k1_flush(S, st, P.tok);
      for (u32 i = lane; i < ncopy; i += 64) P.tok[st.tok_base + st.tok_flushed + i] = P.in_base[byte0 + i];
      st.tok_flushed += ncopy;
      st.produced += ncopy;
      if (ncopy < LEN) { st.fail_pos = (byte0 + ncopy) * 8; status = SEG_UNDERRUN; break; }
      br_seek(st.br, (byte0 + LEN) * 8);
    } else if (btype == 3) {
      status = E_BTYPE;  // deflate.lisp:521
      break;
    } else {
      if (btype == 1) {
        if (tables != 1) {
          status = k1_build_fixed(S);
          tables = 1;
        }
      } else {
        status = k1_dynamic_header(S, st);
        tables = 2;
      }
      if (status) break;
      status = k1_decode_block(S, st, P.tok);
      if (status) break;
    }
    // :block-end (deflate.lisp:719-722)
    if (bfinal) {
      status = SEG_FINAL;
      br_skip(st.br, (u32)((0 - st.br.pos) & 7));  // byte-align (zlib.lisp:138, gzip.lisp:271)
      if (fmt == 1) {                               // adler32, big-endian (zlib.lisp:86-90)
        if (st.br.pos + 32 <= st.end_bit) {
          u32 v = br_peek(st.br);
          br_skip(st.br, 32);
          tr0 = (v >> 24) | ((v >> 8) & 0xff00) | ((v << 8) & 0xff0000) | (v << 24);
          tr_have = 2;
        }
      } else if (fmt == 2) {  // crc32 then ISIZE, little-endian (gzip.lisp:82-106)
        if (st.br.pos + 32 <= st.end_bit) {
          tr0 = br_peek(st.br);
          br_skip(st.br, 32);
          tr_have = 1;
          if (st.br.pos + 32 <= st.end_bit) {
            tr1 = br_peek(st.br);
            br_skip(st.br, 32);
            tr_have = 2;
          }
        }
      } else {
        tr_have = 2;
      }
      break;
    }
    if (!fixup) {
      if (st.br.pos == st.limit_bit) { status = SEG_LANDED; break; }
    } else if ((st.br.pos & 7) == 0) {
      u64 b = st.br.pos >> 3;
      u32 lo = 0, hi = P.n_markers;
      while (lo < hi) {
        u32 mid = (lo + hi) >> 1;
        if (P.markers[mid] < b) lo = mid + 1; else hi = mid;
      }
      if (lo < P.n_markers && P.markers[lo] == b) { land = lo; status = SEG_LANDED; break; }
    }
  }

  u64 r_end, r_out, r_tok;
  if (status == SEG_OVERSHOOT) {
    r_end = blk_pos;
    r_out = blk_prod;
    r_tok = blk_tok;
  } else {
    r_end = (status == SEG_UNDERRUN) ? st.fail_pos : st.br.pos;
    r_out = st.produced;
    r_tok = st.tok_flushed + st.staged;
  }
  k1_flush(S, st, P.tok);
  if (lane == 0) {
    SegResult r;
    r.end_bit = r_end;
    r.out_bytes = r_out;
    r.tok_words = r_tok;
    r.status = status;
    r.max_deficit = st.deficit;
    r.trailer0 = tr0;
    r.trailer1 = tr1;
    r.trailer_have = tr_have;
    r.land_marker = land;
    r.reserved = 0;
    P.res[tbz_block()] = r;
  }
}
#undef K1_CHECK

// ================================================================================================
// K2 — LZ77 resolve: tokens -> LDS ring window -> coalesced stores
// ================================================================================================
constexpr u32 K2_WIN = 36864;   // 32 KiB history + one batch span + slack; multiple of 16
constexpr u32 K2_SPAN = 3072;   // max octets one 64-token batch may produce (cut otherwise)
constexpr u32 K2_FLUSH = 8192;  // flush the ring to HBM every this many octets

struct K2Params {
  const u16* tok;
  const Seg* segs;
  const Group* groups;
  u8* out_base;
  u32 n_groups;
};

TBZ_DEV u32 ring(u32 x) { return x >= K2_WIN ? x - K2_WIN : x; }  // x < 2*K2_WIN

// store ring[from..to) (group-relative octet offsets) to out, clipped at `clip`; `a0` = (address of
// the group's first octet) & 15 so that ring index == address (mod 16) and 16-byte chunks are aligned
TBZ_DEV void k2_flush(const u8* win, u8* outp, u64 from, u64 to, u64 clip, u32 a0) {
  if (to > clip) to = clip;
  if (from >= to) return;
  const u32 lane = tbz_lane();
  tbz_sync();
  u64 head_end = ((from + a0 + 15) & ~15ull) - a0;  // first 16-aligned offset >= from
  if (head_end > to) head_end = to;
  u64 body_end = head_end + ((to - head_end) & ~15ull);
  // head octets
  if (from + lane < head_end) outp[from + lane] = win[(u32)((from + lane + a0) % K2_WIN)];
  // 16-byte body
  u64 nchunk = (body_end - head_end) >> 4;
  u32 r0 = (u32)((head_end + a0) % K2_WIN);
  for (u32 c = lane; c < (u32)nchunk; c += 64) {  // nchunk*16 < K2_WIN
    u32 ri = ring(r0 + c * 16);
    uint4 v = *(const uint4*)(win + ri);
    *(uint4*)(outp + head_end + (u64)c * 16) = v;
  }
  // tail octets
  if (body_end + lane < to) outp[body_end + lane] = win[(u32)((body_end + lane + a0) % K2_WIN)];
  tbz_sync();
}

TBZ_KERNEL void tbz_k2_lz77(K2Params P) {
  TBZ_SHARED __attribute__((aligned(16))) u8 win[K2_WIN];
  if (tbz_block() >= P.n_groups) return;
  const u32 lane = tbz_lane();
  const Group g = P.groups[tbz_block()];
  u8* outp = P.out_base + g.out_abs;
  const u32 a0 = (u32)((uintptr_t)outp & 15);
  const u64 clip = g.out_end > g.out_abs ? g.out_end - g.out_abs : 0;
  u64 pos = 0, flushed = 0;
  u32 rpos = a0;  // ring index of `pos`

  for (u32 s = 0; s < g.seg_count && pos < clip; s++) {
    const Seg sg = P.segs[g.seg_first + s];
    u64 p = 0;
    while (p < sg.tok_words && pos < clip) {
      u64 left = sg.tok_words - p;
      u32 n = left < 64 ? (u32)left : 64;
      u32 w = lane < n ? P.tok[sg.tok_index + p + lane] : 0;
      bool head = lane < n && (w & 0x8000u);
      bool prev_head = tbz_shfl_up(head ? 1u : 0u, 1) != 0 && lane > 0;
      bool isdist = lane < n && prev_head;
      head = head && !isdist;
      bool islit = lane < n && !head && !isdist;
      u32 len = head ? (w & 0xff) + 3 : (islit ? 1u : 0u);
      u32 incl = wave_incl_scan_u32(len);
      // a batch may end after a literal or after a distance word, never between head and distance,
      // and may not produce more than K2_SPAN octets (ring sizing)
      u64 ok = tbz_ballot(lane < n && !head && incl <= K2_SPAN);
      if (ok == 0) break;  // malformed token stream (never produced by K1)
      u32 m = 64 - (u32)__builtin_clzll(ok);
      u32 total = tbz_shfl(incl, (int)m - 1);
      bool active = lane < m;
      u32 dofs = incl - len;  // octet offset of this token inside the batch
      u32 dist = (tbz_shfl_down(w, 1) & 0x7fffu) + 1;
      if (active && islit) win[ring(rpos + dofs)] = (u8)w;
      u64 hm = tbz_ballot(active && head);
      tbz_sync();
      while (hm) {
        int i = (int)tbz_ffs64(hm) - 1;
        hm &= hm - 1;
        u32 l = tbz_shfl(len, i), dd = tbz_shfl(dist, i), o = tbz_shfl(dofs, i);
        u32 rd = ring(rpos + o);                       // ring index of the match's first octet
        u32 rs = rd >= dd ? rd - dd : rd + K2_WIN - dd;  // ring index of its source
        float inv = 1.0f / (float)dd;
        for (u32 j = lane; j < l; j += 64) {
          u32 jj = j;
          if (dd < l) {  // overlapping copy: octet j repeats the dd-octet pattern (deflate.lisp:281-334)
            u32 q = (u32)((float)j * inv);
            i32 r = (i32)j - (i32)(q * dd);
            if (r < 0) r += (i32)dd;
            if (r >= (i32)dd) r -= (i32)dd;
            jj = (u32)r;
          }
          u8 b = win[ring(rs + jj)];
          win[ring(rd + j)] = b;
        }
        tbz_sync();
      }
      pos += total;
      rpos = ring(rpos + total);
      p += m;
      if (pos - flushed >= K2_FLUSH) {
        u64 upto = ((pos + a0) & ~15ull) - a0;  // keep the unaligned tail in the ring
        if (upto > flushed) {
          k2_flush(win, outp, flushed, upto, clip, a0);
          flushed = upto;
        }
      }
    }
  }
  k2_flush(win, outp, flushed, pos, clip, a0);
}

// ================================================================================================
// K4 — adler32 (checksums.lisp:18-62).  Partials per 64 KiB chunk, then an ordered combine.
//   A = sum b_i, B = sum (n - i) * b_i  (i = 0..n-1), both mod 65521
// ================================================================================================
struct K4Params {
  const u8* out_base;
  const CkChunk* chunks;
  CkPartial* parts;
  u32 n_chunks;
};
constexpr u32 ADLER_P = 65521;

TBZ_KERNEL void tbz_k4_adler_partial(K4Params P) {
  if (tbz_block() >= P.n_chunks) return;
  const u32 lane = tbz_lane();
  const CkChunk ch = P.chunks[tbz_block()];
  const u8* p = P.out_base + ch.out_abs;
  const u32 n = ch.len;
  u32 head = (u32)((0 - (uintptr_t)p) & 15);
  if (head > n) head = n;
  u32 body = (n - head) & ~15u;
  u64 A = 0, B = 0;
  if (lane < head) {
    u32 b = p[lane];
    A += b;
    B += (u64)(n - lane) * b;
  }
  const uint4* q = (const uint4*)(p + head);
  for (u32 c = lane; c < (body >> 4); c += 64) {
    uint4 v = q[c];
    u32 i0 = head + c * 16;
    u32 wv[4] = {v.x, v.y, v.z, v.w};
    u32 a16 = 0, w16 = 0;
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
      u32 b = (wv[k >> 2] >> ((k & 3) * 8)) & 0xff;
      a16 += b;
      w16 += k * b;
    }
    A += a16;
    B += (u64)(n - i0) * a16 - w16;
  }
  u32 t0 = head + body;
  if (t0 + lane < n) {
    u32 b = p[t0 + lane];
    A += b;
    B += (u64)(n - t0 - lane) * b;
  }
  A = wave_sum_u64(A);
  B = wave_sum_u64(B % ADLER_P);
  if (lane == 0) {
    CkPartial r;
    r.a = (u32)(A % ADLER_P);
    r.b = (u32)(B % ADLER_P);
    P.parts[tbz_block()] = r;
  }
}

struct K4cParams {
  const CkChunk* chunks;
  const CkPartial* parts;
  const CkStream* streams;
  u32* out;  // per stream: s1 | s2 << 16
  u32 n_streams;
};
TBZ_KERNEL void tbz_k4_adler_combine(K4cParams P) {
  if (tbz_block() >= P.n_streams) return;
  const u32 lane = tbz_lane();
  const CkStream cs = P.streams[tbz_block()];
  u64 s1 = cs.init0 & 0xffff, s2 = cs.init0 >> 16;  // running, < P
  for (u32 c0 = 0; c0 < cs.count; c0 += 64) {
    u32 c = c0 + lane;
    bool v = c < cs.count;
    u64 a = v ? P.parts[cs.first + c].a : 0, b = v ? P.parts[cs.first + c].b : 0;
    u64 n = v ? P.chunks[cs.first + c].len : 0;
    u64 inc = wave_incl_scan_u64(a);
    u64 s1_before = (s1 + inc - a) % ADLER_P;
    u64 term = v ? ((n % ADLER_P) * s1_before + b) % ADLER_P : 0;
    u64 tsum = wave_sum_u64(term);
    s2 = (s2 + tsum) % ADLER_P;
    s1 = (s1 + tbz_shfl64(inc, 63)) % ADLER_P;
  }
  if (lane == 0) P.out[tbz_block()] = (u32)s1 | ((u32)s2 << 16);
}

// ================================================================================================
// K5 — crc32 (checksums.lisp:177-210), reflected polynomial 0xedb88320.
// Z = "advance the register by one zero octet" = multiplication by x^8 mod P.  With init 0,
//   r(M) = XOR_i Z^(n-1-i) T[b_i]     and    r(A||B) = Z^|B| r(A) ^ r(B).
// A chunk is read as 256-byte rows, lane j taking word j of each row (fully coalesced); each lane
// runs S <- Z^256(S) ^ word (4 LDS lookups), then lanes are aligned with one constant multiply each.
// ================================================================================================
TBZ_DEV u32 crc_mulmod(u32 a, u32 b) {  // a*b mod P, reflected (bit 31 = x^0)
  u32 p = 0;
  for (int i = 0; i < 32; i++) {
    if (a & (0x80000000u >> i)) p ^= b;
    b = (b & 1) ? ((b >> 1) ^ 0xedb88320u) : (b >> 1);
  }
  return p;
}
TBZ_DEV u32 crc_pow_x8(const u32* x2n, u64 nbytes) {  // x^(8*nbytes) mod P
  u32 p = 0x80000000u;
  u32 k = 3;
  while (nbytes) {
    if (nbytes & 1) p = crc_mulmod(x2n[k & 63], p);
    nbytes >>= 1;
    k++;
  }
  return p;
}
TBZ_DEV u32 crc_bytes_seq(const u32* T, const u8* p, u32 n) {  // r() of n octets, one lookup per octet
  u32 s = 0;
  for (u32 i = 0; i < n; i++) s = (s >> 8) ^ T[(s ^ p[i]) & 0xff];
  return s;
}

struct K5Params {
  const u8* out_base;
  const CkChunk* chunks;
  CkPartial* parts;
  const u32* crc_tab;  // CRC_WORDS words (tbz_structs.hpp)
  u32 n_chunks;
};

TBZ_KERNEL void tbz_k5_crc_partial(K5Params P) {
  TBZ_SHARED u32 tab[CRC_X2N];  // T + K0..K3
  const u32 lane = tbz_lane();
  for (u32 i = lane; i < CRC_X2N; i += 64) tab[i] = P.crc_tab[i];
  tbz_sync();
  if (tbz_block() >= P.n_chunks) return;
  const CkChunk ch = P.chunks[tbz_block()];
  const u8* p = P.out_base + ch.out_abs;
  const u32 n = ch.len;
  u32 head = (u32)((0 - (uintptr_t)p) & 3);
  if (head > n) head = n;
  u32 rows = (n - head) >> 8;
  u32 tail = n - head - rows * 256;
  const u32* q = (const u32*)(p + head);
  const u32* K = tab + CRC_K;
  u32 S = 0;
  for (u32 r = 0; r < rows; r++) {
    u32 w = q[r * 64 + lane];
    S = K[S & 0xff] ^ K[256 + ((S >> 8) & 0xff)] ^ K[512 + ((S >> 16) & 0xff)] ^ K[768 + (S >> 24)] ^ w;
  }
  u32 rr = 0;
  if (rows) {
    rr = crc_mulmod(P.crc_tab[CRC_LANE + lane], S);
    rr = wave_xor_u32(rr);
  }
  // wave-uniform from here
  u32 rtot = rr;
  if (head) {
    u32 rh = crc_bytes_seq(tab, p, head);
    rtot ^= crc_mulmod(crc_pow_x8(P.crc_tab + CRC_X2N, (u64)rows * 256), rh);
  }
  if (tail) {
    u32 rt = crc_bytes_seq(tab, p + head + rows * 256, tail);
    rtot = crc_mulmod(crc_pow_x8(P.crc_tab + CRC_X2N, tail), rtot) ^ rt;
  }
  if (lane == 0) {
    CkPartial o;
    o.a = rtot;
    o.b = 0;
    P.parts[tbz_block()] = o;
  }
}

struct K5cParams {
  const CkChunk* chunks;
  const CkPartial* parts;
  const CkStream* streams;
  const u32* crc_tab;
  u32* out;  // per stream: finalised crc
  u32 n_streams;
};
TBZ_KERNEL void tbz_k5_crc_combine(K5cParams P) {
  if (tbz_block() >= P.n_streams) return;
  const u32 lane = tbz_lane();
  const CkStream cs = P.streams[tbz_block()];
  const u32* x2n = P.crc_tab + CRC_X2N;
  // octets after chunk c = suffix sum of lengths; lanes take chunks round-robin
  u64 total = 0;
  for (u32 c0 = 0; c0 < cs.count; c0 += 64) {
    u32 c = c0 + lane;
    total += wave_sum_u64(c < cs.count ? P.chunks[cs.first + c].len : 0);
  }
  u32 acc = 0;
  u64 before = 0;
  for (u32 c0 = 0; c0 < cs.count; c0 += 64) {
    u32 c = c0 + lane;
    bool v = c < cs.count;
    u64 n = v ? P.chunks[cs.first + c].len : 0;
    u64 inc = wave_incl_scan_u64(n);
    u64 after = total - (before + inc);
    if (v) acc ^= crc_mulmod(crc_pow_x8(x2n, after), P.parts[cs.first + c].a);
    before += tbz_shfl64(inc, 63);
  }
  acc = wave_xor_u32(acc);
  // crc' = ~( Z^n(~crc) ^ r(M) )  — chaining convention of crc32/table (checksums.lisp:201,:210)
  u32 init = cs.init0 ^ 0xffffffffu;
  u32 fin = crc_mulmod(crc_pow_x8(x2n, total), init) ^ acc;
  if (lane == 0) P.out[tbz_block()] = fin ^ 0xffffffffu;
}

}  // namespace tbz
