// tbz_structs.hpp — plain records shared by host orchestration and kernels.
//
// Vocabulary (follows the reference's domain, not ML's):
//   stream   one deflate/zlib/gzip stream = one (decompress-vector …) call of 3bz
//   marker   a candidate block start, as a BIT position: the octet after input[p..p+4) = 00 00 FF FF (an empty
//            stored block: Z_SYNC_FLUSH / Z_FULL_FLUSH), found by K0; or a bit position where a plausible
//            dynamic-Huffman block header begins, found by K0b.  Speculative: the patterns also occur
//            inside compressed data; only a decode that ENDS a block exactly there proves it.
//   item     one unit of K1 work: "decode blocks starting at this bit until you land on the next
//            marker / hit the final block / fail"
//   segment  a proven item: contiguous run of blocks between two proven block boundaries
//   group    consecutive segments of one stream that must share one LZ77 window (the first needs no
//            history; the rest reach back into their predecessors)
//   token    u16 word: literal (0x00bb), or match head (0x8000 | len-3) followed by (dist-1)
//   run      a contiguous piece of an item's token stream (one lane's share of a K1g round), 8-word granular
#pragma once
#include <cstdint>

namespace tbz {

// ---- K1 item flags
enum : uint32_t {
  ITEM_HEAD = 1u,    // starts at the stream's first byte: parse the zlib/gzip container header first
  ITEM_FIXUP = 2u,   // chain repair: ignore `limit_bit`, land on ANY marker (binary search)
  ITEM_PROBE = 4u,   // re-decode by the one-lane kernel to locate a distance error: SegResult.reserved = its place
  ITEM_RESUME = 8u,  // a session's continuation INSIDE a block: the block's header is parsed where the item starts, then the
                     // token loop is entered at the launch's resume_bit (what deflate.lisp:399-427 achieves by pushing the
                     // bits of an unfinished symbol back)
  ITEM_EXPLICIT = 16u, // the item brings its own token region (Item::tok, one word per input bit from its start) and run
                       // table: items decoded AGAIN (SEG_REDO: too dense for a pool of one word per two bits; ITEM_PROBE)
  ITEM_FMT_SHIFT = 8, // format in bits 8..9
  ITEM_HIST_SHIFT = 16 // bits 16..31: octets of the stream before the item (saturated at 65535), for the exact
                       // location of a distance-before-start error (one-lane re-decode of the offending item)
};

// ---- K1 segment status (internal; mapped to tbz_result.status by the host)
enum : int32_t {
  SEG_LANDED = 1,     // ended a block exactly on a marker (index in land_marker)
  SEG_FINAL = 2,      // decoded the BFINAL block (trailer fields valid)
  SEG_OVERSHOOT = 3,  // crossed limit_bit mid-block: resume from end_bit (= start of that block)
  SEG_UNDERRUN = 4,   // input ended (deflate.lisp:114-120); out_bytes/tok_words cover complete tokens
  SEG_REDO = 5,       // the gang kernel declined the item (token stream as dense as the bitstream, run table
                      // full): decode it again with one lane
  SEG_WIDE = 6        // a narrow gang declined an item far larger than the launch's mean (K1gParams::wide_bits): the
                      // host hands it to gangs of 64
  // <0: TBZ_E_* error codes of include/tbz_amd.h
};

struct Item {
  uint64_t start_bit;  // bit position relative to in_base
  uint64_t limit_bit;  // first bit this item must not decode past without landing (next marker*8)
  uint64_t end_byte;   // end of the stream (relative to in_base)
  uint32_t stream;
  uint32_t flags;
  uint64_t tok;        // ITEM_EXPLICIT: device address of the item's token region (16-octet aligned) ...
  uint64_t runs;       // ... and of its run table (gang kernels; (span >> RUN_SHIFT) + 2 entries)
};

// One contiguous piece of an item's token stream: n8 * 8 words at the item's token base + off8 * 8.
// Run 0 of an item lives in its SegResult / Seg; runs 1.. in the item's run table, which starts at slot
// (item.start_bit >> RUN_SHIFT) of the pool's table (position-addressed like the token pool: K1 declines items that
// would need more runs than their span holds slots, and an item that owns slot k+1.. has consumed k+1 slots' worth of
// bits, so that tables never overlap however close two short items start).  `out` and `mdef` let K3 cut a large
// segment into K2 work units at run boundaries without looking at a token (tbz_k3_slice).
struct RunRec {
  uint32_t off8;
  uint32_t n8;
  uint32_t out;   // octets the run's tokens produce (0xffffffff: more than fits — such a segment is not sliced)
  uint32_t mdef;  // max over its matches of (distance - octets the RUN produced before the match), 0 if none reaches
                  // before the run's first octet
};
constexpr uint32_t RUN_SHIFT = 7;  // input bits per run-table slot

struct SegResult {
  uint64_t end_bit;      // LANDED/FINAL: bit after the last block (FINAL: after the trailer);
                         // OVERSHOOT: start of the block that crossed; UNDERRUN: start of the token/field
  uint64_t out_bytes;    // octets the decoded tokens produce
  uint64_t tok_words;    // logical length of the item's token stream in u16 words: the sum of its runs (each
                         // padded to a multiple of 8 words with TOK_NOP)
  int32_t status;
  uint32_t max_deficit;  // max over matches of (distance - octets produced before it in THIS item), 0 if none
  uint32_t trailer0;     // zlib: adler32 (already byte-swapped to host order); gzip: crc32
  uint32_t trailer1;     // gzip: ISIZE
  uint32_t trailer_have; // 0 none, 1 first word only (gzip crc without isize), 2 complete
  uint32_t land_marker;  // LANDED: index of the marker landed on
  uint32_t n_runs;       // entries of the item's run table that make up tok_words
  uint32_t pad;          // UNDERRUN: 1 when the input ended exactly where a block starts (end_bit = that boundary);
                         // 2 when it ended inside a stored block's payload (the reference checks output space first there)
  uint64_t reserved;     // K1g diagnostics: rounds << 32 | committed lanes
  uint64_t tok;          // device address of the item's token base (what the runs' off8 count from)
  uint64_t runs;         // device address of the item's run table (entry 0 unused: run0 below)
  RunRec run0;
};

struct Seg {
  uint64_t tok;        // device address of the item's token base (SegResult::tok)
  uint64_t tok_words;  // logical token words (multiple of 8)
  uint64_t out_bytes;
  uint32_t n_runs;
  uint32_t run_first;  // first run of the item that belongs to this unit (a slice of a large segment)
  uint64_t runs;       // device address of the item's run table (SegResult::runs): runs 1..; run 0 is run0
  RunRec run0;
};

struct Group {
  uint64_t out_abs;    // byte offset in out_base of the group's first octet
  uint64_t out_end;    // byte offset in out_base one past the last octet this group may store (clip)
  uint32_t seg_first;
  uint32_t seg_count;
};

// checksum chunk: `len` octets at out_base[out_abs..)
struct CkChunk {
  uint64_t out_abs;
  uint32_t len;
  uint32_t stream;
};
// per stream: chunks [first, first+count)
struct CkStream {
  uint32_t first;
  uint32_t count;
  uint32_t init0;  // adler: s1 | s2<<16 ; crc: finalised crc to chain from
  uint32_t pad;
};
struct CkPartial {
  uint32_t a;  // adler: sum of octets mod 65521        crc: raw (init 0) crc of the chunk
  uint32_t b;  // adler: sum of (len-i)*octet mod 65521 crc: unused
};

constexpr uint32_t CK_CHUNK = 256u << 10;  // octets per checksum workgroup
constexpr uint32_t SCAN_TILE = 64u << 10;  // octets per marker-scan workgroup

// crc constants table layout (u32 words), filled by the host at context creation
constexpr uint32_t CRC_T = 0;         // 256: classic reflected table (checksums.lisp:177-193)
constexpr uint32_t CRC_K = 256;       // 4*256: advance-by-256-octets operator, split by state byte
constexpr uint32_t CRC_X2N = 1280;    // 64: x^(2^k) mod P, k = 0..63 (reflected)
constexpr uint32_t CRC_LANE = 1344;   // 64: x^(8*(256-4*lane)) mod P
constexpr uint32_t CRC_WORDS = 1408;

}  // namespace tbz
