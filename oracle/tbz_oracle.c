/* tbz_oracle.c — CPU restatement of 3bz's inflate path (plain C).
 *
 * TEST INFRASTRUCTURE ONLY — see tbz_oracle.h.  This file restates the
 * reference's ALGORITHM and observable contract; it is not the product and the
 * product never calls it.  Every function cites the reference file:line it
 * follows (paths relative to the 3bz source tree).
 *
 * What is restated faithfully: the resumable state machine and its tags
 * (deflate.lisp:518-728), the 16-bit node format and nested-table builder
 * (huffman-tree.lisp:10-71, :99-218), the RFC tables (constants.lisp:41-73),
 * the static trees (ht-constants.lisp:9-32), adler32/crc32 (checksums.lisp),
 * the zlib and gzip wrappers (zlib.lisp:39-144, gzip.lisp:30-287) and the easy
 * API (api.lisp).  What is deliberately NOT restated: SBCL's instruction mix —
 * the word64/word32 read-ahead (io.lisp:17-58) and the 8/4/2-byte wide copies
 * (deflate.lisp:275-334) are replaced by byte-granular equivalents whose
 * results on buffer[0,count), status flags and resume behaviour are identical
 * (the wide copies only scribble past `count`, later overwritten).
 */
#include "tbz_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ---- constants.lisp ------------------------------------------------------ */
#define MAX_TREE_SIZE (852 + 592) /* constants.lisp:4-7 */
#define HT_LITERAL 0              /* constants.lisp:14-17 */
#define HT_LINK_END 1
#define HT_LEN_DIST 2
#define HT_INVALID 3
#define END_CODE 256      /* constants.lisp:20 */
#define LENGTHS_START 257 /* constants.lisp:22 */
#define LENGTHS_END 285   /* constants.lisp:24 */
#define LENGTHS_EXTRA_BITS_OFFSET 32
#define ADLER32_PRIME 65521u

/* constants.lisp:41-48: one 61-entry table, distances at 0.., lengths at 32.. */
static const uint8_t EXTRA_BITS[61] = {
    0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 0, 0,
    0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
/* constants.lisp:53-61 */
static const uint16_t LEN_DIST_BASES[61] = {
    1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
    4097, 6145, 8193, 12289, 16385, 24577, 0, 0,
    3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195,
    227, 258};
/* constants.lisp:65-68 */
static const uint8_t LEN_CODE_ORDER[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
/* constants.lisp:70-73 (padded: build-tree-part indexes it with i <= 29) */
static const uint8_t LEN_CODE_EXTRA[61] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 3, 7};

/* ---- huffman-tree.lisp node accessors (:10-71) ---------------------------- */
typedef struct {
  int start_bits; /* ht-start-bits */
  int max_bits;   /* ht-max-bits */
  uint16_t nodes[MAX_TREE_SIZE];
} hufftree;

static void ht_init(hufftree* t) { /* huffman-tree.lisp:78-84: nodes default to invalid */
  t->start_bits = 0;
  t->max_bits = 0;
  for (int i = 0; i < MAX_TREE_SIZE; i++) t->nodes[i] = 0xffff;
}
static inline int ht_node_type(unsigned n) { return n & 3; }             /* :22 */
static inline int ht_endp(unsigned n) { return n == 0x0001; }            /* :20 */
static inline unsigned ht_link_node(unsigned bits, unsigned index) {     /* :26 */
  return HT_LINK_END | (bits << 2) | (index << 6);
}
static inline unsigned ht_link_bits(unsigned n) { return (n >> 2) & 15; } /* :30 */
static inline unsigned ht_link_offset(unsigned n) { return (n >> 6) & 1023; } /* :32 */
static inline unsigned ht_extra_bits(unsigned n) { return (n >> 2) & 15; } /* :43 */
static inline unsigned ht_value(unsigned n) { return (n >> 6) & 1023; }   /* :45 */
static inline unsigned ht_literal_node(unsigned v) { return HT_LITERAL | (v << 6); } /* :50 */
static inline unsigned ht_len_node(unsigned value, const uint8_t* extra) { /* :54-65 */
  unsigned v = LENGTHS_EXTRA_BITS_OFFSET + (value - LENGTHS_START);
  return (HT_LEN_DIST | (v << 6) | ((unsigned)extra[v] << 2)) & 0xffff;
}
static inline unsigned ht_dist_node(unsigned value, const uint8_t* extra) { /* :67-71 */
  return (HT_LEN_DIST | (value << 6) | ((unsigned)extra[value] << 2)) & 0xffff;
}

/* util.lisp:59-69 — bit-rev: reverse the low `bits` bits of x */
static unsigned bit_rev(unsigned x, unsigned bits) {
  unsigned r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
}

enum { T_DIST, T_LITLEN, T_DHTLEN };

typedef struct {
  hufftree* tree;
  const hufftree* terminals;
  int counts[16];
  int offsets[16];
  int next_subtable;
  int err;
} fill_ctx;

/* huffman-tree.lisp:188-212 — labels next-len / subtable */
static int next_len(const fill_ctx* f, int l) {
  for (int i = l; i < 16; i++)
    if (f->counts[i] > 0) return i;
  return -1;
}
static unsigned subtable(fill_ctx* f, int prefix_bits) {
  int entry_bits = (f->counts[prefix_bits] == 0) ? next_len(f, prefix_bits) : prefix_bits;
  if (entry_bits < 0) return 0xffff; /* :212 ht-invalid-node */
  if (prefix_bits == entry_bits) {   /* :197-201 */
    unsigned n = f->terminals->nodes[f->offsets[entry_bits]];
    f->offsets[entry_bits]++;
    f->counts[entry_bits]--;
    return n;
  }
  /* :203-211 — allocate a 2^b sub-table, fill in canonical order at bit-reversed slots */
  int start = f->next_subtable;
  int b = entry_bits - prefix_bits;
  f->next_subtable += 1 << b;
  if (f->next_subtable > MAX_TREE_SIZE || start > 1023) {
    f->err = TBZO_E_TREE_OVERFLOW;
    return 0xffff;
  }
  for (int i = 0; i < (1 << b); i++) {
    unsigned n = subtable(f, entry_bits);
    if (f->err) return 0xffff;
    f->tree->nodes[start + bit_rev((unsigned)i, (unsigned)b)] = (uint16_t)n;
  }
  return ht_link_node((unsigned)b, (unsigned)start);
}

/* Test switch (NOT reference behaviour): with an all-zero alphabet the reference leaves whatever the previous
 * block put in the table (:156-157) — state that crosses block and flush boundaries and that no encoder's
 * stream depends on.  The device path documents that it behaves like a FRESH reference state there (every entry
 * invalid); with this switch on the oracle does the same, so the fuzzers can tell that documented
 * deviation from a real difference. */
static int g_fresh_tables = 0;
void tbzo_set_fresh_tables(int on) { g_fresh_tables = on; }

/* huffman-tree.lisp:99-218 — build-tree-part.  Returns <0 on error, else 0 and
 * (*count,*bits,*max) = (values next-subtable min max-bits). */
static int build_tree_part(hufftree* tree, int tree_offset, const uint8_t* table, int type, int start,
                           int end, hufftree* scratch, const uint8_t* extra_bits, int* count, int* bits,
                           int* max) {
  fill_ctx f;
  f.tree = tree;
  f.terminals = scratch;
  f.err = 0;
  int* a = f.counts;
  for (int i = 0; i < 16; i++) a[i] = 0;
  for (int x = start; x < end; x++) a[table[x] & 15]++; /* :109-111 */
  {                                                     /* :112-122 Kraft check */
    long s = 1;
    for (int i = 0; i < 16; i++) {
      if (i != 0) {
        s <<= 1;
        if (a[i] > s) return TBZO_E_OVERSUBSCRIBED;
        s -= a[i];
      }
    }
    if (s > 0 && 1 < ((end - start) - a[0])) return TBZO_E_INCOMPLETE;
  }
  a[0] = 0; /* :123 */
  int code_offsets[16];
  { /* :126-142 */
    int c = 0, cc = 0;
    for (int i = 0; i < 16; i++) {
      f.offsets[i] = a[i] ? c : 0;
      c += a[i];
      code_offsets[i] = a[i] ? cc : 0;
      cc = (cc + a[i]) << 1;
    }
    (void)code_offsets; /* computed by the reference too, never consumed (:166) */
  }
  int min = -1, last = 0;
  for (int i = 0; i < 16; i++)
    if (a[i]) {
      if (min < 0) min = i;
      last = i;
    }
  int max_bits = last + (type == T_DIST ? 13 : type == T_LITLEN ? 5 : 7); /* :146-150 */
  if (min < 0) { /* :156-157: all-zero lengths — table left untouched */
    if (g_fresh_tables) /* test switch, see tbzo_set_fresh_tables */
      for (int i = tree_offset; i < MAX_TREE_SIZE; i++) tree->nodes[i] = 0xffff;
    *count = 0;
    *bits = 0;
    *max = 0;
    return 0;
  }
  { /* :159-183 sort terminals by (length, symbol) */
    int offset_tmp[16];
    memcpy(offset_tmp, f.offsets, sizeof offset_tmp);
    int i = 0;
    for (int to = start; to < end; to++, i++) {
      int l = table[to] & 15;
      if (l > 0) {
        int o = offset_tmp[l]++;
        unsigned n;
        if (type == T_DIST || type == T_DHTLEN)
          n = (i <= 29) ? ht_dist_node((unsigned)i, extra_bits) : 0xffff;
        else if (i > LENGTHS_END)
          n = 0xffff;
        else if (i >= LENGTHS_START)
          n = ht_len_node((unsigned)i, extra_bits);
        else if (i == END_CODE)
          n = 0x0001;
        else
          n = ht_literal_node((unsigned)i);
        scratch->nodes[o] = (uint16_t)n;
      }
    }
  }
  /* :186-217 fill tree */
  f.next_subtable = tree_offset;
  f.next_subtable += 1 << min;
  if (f.next_subtable > MAX_TREE_SIZE) return TBZO_E_TREE_OVERFLOW;
  for (int i = 0; i < (1 << min); i++) {
    unsigned n = subtable(&f, min);
    if (f.err) return f.err;
    tree->nodes[tree_offset + bit_rev((unsigned)i, (unsigned)min)] = (uint16_t)n;
  }
  *count = f.next_subtable;
  *bits = min;
  *max = max_bits;
  return 0;
}

/* huffman-tree.lisp:272-287 — build-trees* */
static int build_trees_star(hufftree* ltree, hufftree* dtree, const uint8_t* lld, int mid, int end,
                            hufftree* scratch) {
  int count, bits, max, r;
  r = build_tree_part(ltree, 0, lld, T_LITLEN, 0, mid, scratch, EXTRA_BITS, &count, &bits, &max);
  if (r < 0) return r;
  ltree->start_bits = bits;
  ltree->max_bits = max;
  r = build_tree_part(dtree, 0, lld, T_DIST, mid, end, scratch, EXTRA_BITS, &count, &bits, &max);
  if (r < 0) return r;
  dtree->start_bits = bits;
  dtree->max_bits = max;
  return 0;
}

/* ht-constants.lisp:9-32 + huffman-tree.lisp:89-97 — static (BTYPE=1) trees */
static hufftree g_static_len, g_static_dist;
static int g_static_ready = 0;
static void ensure_static_trees(void) {
  if (g_static_ready) return;
  uint8_t lit[288], dist[32];
  for (int i = 0; i <= 143; i++) lit[i] = 8;
  for (int i = 144; i <= 255; i++) lit[i] = 9;
  for (int i = 256; i <= 279; i++) lit[i] = 7;
  for (int i = 280; i <= 287; i++) lit[i] = 8;
  for (int i = 0; i < 32; i++) dist[i] = 5;
  hufftree scratch;
  ht_init(&g_static_len);
  ht_init(&g_static_dist);
  ht_init(&scratch);
  int c, b, m;
  /* huffman-tree.lisp:254-270 build-trees */
  build_tree_part(&g_static_len, 0, lit, T_LITLEN, 0, 288, &scratch, EXTRA_BITS, &c, &b, &m);
  g_static_len.start_bits = b;
  g_static_len.max_bits = m;
  build_tree_part(&g_static_dist, 0, dist, T_DIST, 0, 32, &scratch, EXTRA_BITS, &c, &b, &m);
  g_static_dist.start_bits = b;
  g_static_dist.max_bits = m;
  g_static_ready = 1;
}

/* ---- checksums.lisp -------------------------------------------------------- */
/* checksums.lisp:18-62 adler32/ub64 (selected on 64-bit SBCL, tuning.lisp:19-33):
 * 64-bit accumulators, reduce mod 65521 every <= 380368439 adds and at the end. */
void tbzo_adler32(const uint8_t* buf, size_t end, uint32_t s1_in, uint32_t s2_in, uint32_t* o1,
                  uint32_t* o2) {
  uint64_t s1 = s1_in, s2 = s2_in;
  const size_t chunk = 32u * (380368439u / 32u); /* :23-27 with +adler32-unroll+ = 32 (tuning.lisp:38-41) */
  size_t i = 0;
  while (end - i > 32) { /* :44-54 */
    size_t rem = end - i;
    size_t n = 32 * (rem / 32);
    if (n > chunk) n = chunk;
    size_t c = i + n;
    while (i < c) {
      s1 += buf[i];
      s2 += s1;
      i++;
    }
    s1 %= ADLER32_PRIME;
    s2 %= ADLER32_PRIME;
  }
  for (; i < end; i++) { /* :55-58 tail */
    s1 += buf[i];
    s2 += s1;
  }
  *o1 = (uint32_t)(s1 % ADLER32_PRIME); /* :59-60 */
  *o2 = (uint32_t)(s2 % ADLER32_PRIME);
}

/* checksums.lisp:177-193 — table generation, poly #xedb88320 */
static uint32_t g_crc_table[256];
static int g_crc_ready = 0;
static void ensure_crc_table(void) {
  if (g_crc_ready) return;
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? (0xedb88320u ^ (c >> 1)) : (c >> 1);
    g_crc_table[n] = c;
  }
  g_crc_ready = 1;
}
/* checksums.lisp:196-210 — crc32/table: takes and returns the finalised CRC */
uint32_t tbzo_crc32(const uint8_t* buf, size_t end, uint32_t crc) {
  ensure_crc_table();
  crc ^= 0xffffffffu;
  for (size_t i = 0; i < end; i++) crc = (crc >> 8) ^ g_crc_table[(crc ^ buf[i]) & 0xff];
  return crc ^ 0xffffffffu;
}

/* ---- deflate-state (deflate.lisp:4-62) + zlib-state (zlib.lisp:3-12) +
 *      gzip-state (gzip.lisp:3-28) ------------------------------------------ */
enum {
  S_START_OF_BLOCK,
  S_UNCOMPRESSED_BLOCK,
  S_COPY_BLOCK,
  S_DYNAMIC_HUFFMAN_BLOCK,
  S_DHT_LEN_TABLE,
  S_DHT_LEN_TABLE_DATA,
  S_DECODE_COMPRESSED_DATA,
  S_CONTINUE_COPY_HISTORY,
  S_OUT_BYTE,
  S_BLOCK_END,
  S_DONE
};
enum { Z_HEADER, Z_HEADER2, Z_ADLER, Z_NIL };
enum {
  G_HEADER,
  G_HEADER2,
  G_HEADER_MTIME,
  G_HEADER3,
  G_HEADER_EXTRA,
  G_HEADER_NAME,
  G_HEADER_COMMENT,
  G_HEADER_CRC,
  G_DEFLATE,
  G_FINAL_CRC,
  G_FINAL_LEN,
  G_DONE
};

struct tbzo_state {
  int format;
  /* deflate-state */
  int current_state;
  int last_block_flag;
  hufftree dyn_len, dyn_dist; /* dynamic-huffman-tree (cons) */
  int current_is_static;      /* current-huffman-tree */
  int dht_hlit, dht_hlit_hdist, dht_hclen;
  uint8_t dht_len_codes[19];
  hufftree dht_len_tree;
  uint8_t dht_lld[288 + 32 + 64]; /* dht-lit/len/dist (+slack; reference array is 320) */
  int dht_lld_index;
  int dht_last_len;
  unsigned bytes_to_copy; /* ub16 */
  unsigned copy_offset;   /* ub16 */
  uint64_t partial_bits;
  int bits_remaining;
  int64_t output_offset;
  uint8_t* output_buffer;
  size_t output_len;
  uint8_t* window; /* 32768 + 8, allocated on first overflow (deflate.lisp:124-129) */
  int finished, output_overflow, input_underrun;
  hufftree ht_scratch; /* deflate.lisp:139 */
  /* zlib-state */
  int zlib_state;
  uint32_t s1, s2;
  /* gzip-state */
  int gzip_state;
  int gz_flags;        /* FLG byte */
  int gz_keep_header;  /* header-bytes non-nil */
  uint8_t* gz_header_bytes;
  size_t gz_header_len, gz_header_cap;
  int64_t gz_extra_len, gz_extra_read; /* extra: -1 = not yet allocated */
  int gz_name_started, gz_comment_started;
  /* the metadata slots of gzip-state (gzip.lisp:17-25), as decompress-gzip fills them (gzip.lisp:123-241) */
  int gz_have_cm, gz_have_mtime, gz_have_os, gz_have_extra, gz_have_name, gz_have_comment;
  uint32_t gz_mtime, gz_xfl, gz_os;
  uint8_t *gz_extra, *gz_name, *gz_comment;
  size_t gz_name_len, gz_comment_len, gz_name_cap, gz_comment_cap;
  uint32_t crc32;
  uint32_t gz_isize;
  char errmsg[160];
};

tbzo_state* tbzo_make_state(int format, uint8_t* output_buffer, size_t output_len) {
  ensure_static_trees();
  tbzo_state* st = (tbzo_state*)calloc(1, sizeof *st);
  if (!st) return NULL;
  st->format = format;
  st->current_state = S_START_OF_BLOCK;
  ht_init(&st->dyn_len);
  ht_init(&st->dyn_dist);
  ht_init(&st->dht_len_tree);
  ht_init(&st->ht_scratch);
  st->current_is_static = 1;
  st->dht_last_len = 0xff;
  st->output_buffer = output_buffer;
  st->output_len = output_buffer ? output_len : 0;
  st->zlib_state = Z_HEADER;
  st->s1 = 1; /* zlib.lisp:11-12 */
  st->s2 = 0;
  st->gzip_state = G_HEADER;
  st->gz_keep_header = 1;
  st->gz_extra_len = -1;
  return st;
}
void tbzo_free_state(tbzo_state* st) {
  if (!st) return;
  free(st->window);
  free(st->gz_header_bytes);
  free(st->gz_extra);
  free(st->gz_name);
  free(st->gz_comment);
  free(st);
}
void tbzo_free(void* p) { free(p); }
int tbzo_finished(const tbzo_state* s) { return s->finished; }
int tbzo_input_underrun(const tbzo_state* s) { return s->input_underrun; }
int tbzo_output_overflow(const tbzo_state* s) { return s->output_overflow; }
int64_t tbzo_output_offset(const tbzo_state* s) { return s->output_offset; }
const char* tbzo_errmsg(const tbzo_state* s) { return s->errmsg; }
uint32_t tbzo_checksum(const tbzo_state* s) {
  return s->format == TBZO_GZIP ? s->crc32 : (s->s1 | (s->s2 << 16));
}

static int fail(tbzo_state* st, int code, const char* msg) {
  snprintf(st->errmsg, sizeof st->errmsg, "%s", msg);
  return code;
}

/* ---- bit reader ------------------------------------------------------------
 * deflate.lisp:140-231.  The reference pulls 8 (word64) or 4 (word32) octets at a
 * time (io.lisp:17-58); here octets are pulled one at a time up to the same
 * 64-bit capacity.  Same invariant: partial-bits holds bits-remaining valid bits,
 * LSB first, and anything the context had is in the buffer when we report
 * underrun (deflate.lisp:209-216 stashes what it got before `eoi`). */
static inline void refill(tbzo_context* c, tbzo_state* st) {
  while (st->bits_remaining <= 56 && c->offset < c->end) {
    st->partial_bits |= (uint64_t)c->vec[c->offset++] << st->bits_remaining;
    st->bits_remaining += 8;
  }
}
/* deflate.lisp:185-188 `bits` / :192-231 `bits-full`: 1 = got n bits, 0 = eoi */
static inline int get_bits(tbzo_context* c, tbzo_state* st, int n, uint64_t* out) {
  if (n > st->bits_remaining) {
    refill(c, st);
    if (n > st->bits_remaining) return 0;
  }
  *out = (n == 0) ? 0 : (st->partial_bits & ((n >= 64) ? ~0ull : ((1ull << n) - 1)));
  st->partial_bits = (n >= 64) ? 0 : (st->partial_bits >> n);
  st->bits_remaining -= n;
  return 1;
}
/* deflate.lisp:142-146 byte-align */
static inline void byte_align(tbzo_state* st) {
  int r = st->bits_remaining % 8;
  if (r) {
    st->partial_bits >>= r;
    st->bits_remaining -= r;
  }
}

/* deflate.lisp:121-137 `eoo`: keep the last 32 KiB of output in ds-window */
static int save_window(tbzo_state* st) {
  const int64_t wsz = 32768;
  if (!st->window) {
    st->window = (uint8_t*)calloc((size_t)wsz + 8, 1);
    if (!st->window) return fail(st, TBZO_E_STATE, "out of memory");
  }
  int64_t off = st->output_offset;
  if (off < wsz) memmove(st->window, st->window + off, (size_t)(wsz + 8 - off));
  int64_t start1 = wsz - off > 0 ? wsz - off : 0;
  int64_t start2 = off - wsz > 0 ? off - wsz : 0;
  int64_t n1 = wsz + 8 - start1, n2 = (int64_t)st->output_len - start2;
  int64_t n = n1 < n2 ? n1 : n2;
  if (n > 0) memcpy(st->window + start1, st->output_buffer + start2, (size_t)n);
  return 0;
}

/* decode result of one symbol (values of decode-huffman, deflate.lisp:457-461) */
typedef struct {
  unsigned value, extra, type;
  int consumed; /* offset: bits used by code + extra */
} sym;

/* deflate.lisp:361-461 decode-huffman-full / :465-501 %decode-huffman-fast, merged:
 * walk the nested table using the bits at [skip, avail) of the buffer WITHOUT
 * consuming them.  Returns 1 ok, 0 = not enough bits (the reference's push-back +
 * `eoi`, :399-427), <0 = invalid node.  The caller consumes on success, which
 * gives the same "re-decode the whole symbol (and its length code) on resume"
 * behaviour as the reference's old-bits/old-count rollback. */
static int peek_symbol(const tbzo_state* st, const hufftree* ht, int skip, sym* out) {
  uint64_t bits = (skip >= 64) ? 0 : (st->partial_bits >> skip);
  int avail = st->bits_remaining - skip;
  int ht_bits = ht->start_bits;
  int offset = 0;
  unsigned node = 0, base = 0;
  for (;;) {
    if (ht_bits > avail - offset) return 0;
    unsigned b = ht_bits ? (unsigned)((bits >> offset) & ((1u << ht_bits) - 1)) : 0;
    if (base + b >= MAX_TREE_SIZE) return TBZO_E_TREE_OVERFLOW;
    node = ht->nodes[base + b];
    offset += ht_bits;
    switch (ht_node_type(node)) {
      case HT_LINK_END:
        if (ht_endp(node)) goto done0;
        ht_bits = (int)ht_link_bits(node);
        base = ht_link_offset(node);
        continue;
      case HT_LITERAL:
        goto done0;
      case HT_LEN_DIST: {
        int x = (int)ht_extra_bits(node);
        if (x == 0) goto done0;
        if (x > avail - offset) return 0;
        out->extra = (unsigned)((bits >> offset) & ((1u << x) - 1));
        offset += x;
        goto done;
      }
      default: /* +ht-invalid+: no ecase clause (deflate.lisp:438, :481) */
        return TBZO_E_INVALID_NODE;
    }
  }
done0:
  out->extra = 0;
done:
  out->value = ht_value(node);
  out->type = (unsigned)ht_node_type(node);
  out->consumed = offset;
  return 1;
}
static inline void consume(tbzo_state* st, int n) {
  st->partial_bits = (n >= 64) ? 0 : (st->partial_bits >> n);
  st->bits_remaining -= n;
}

enum { R_CONTINUE = 0, R_EOI = 1, R_EOO = 2 };

/* deflate.lisp:244-335 %copy-history, byte-granular.  The reference takes a slow
 * byte loop when d+count+8 > e (:252-269) and wide copies otherwise (:275-334); on
 * buffer[0,count) both equal this loop.  *d advances; returns bytes left. */
static unsigned copy_bytes(const uint8_t* from, int64_t s, uint8_t* to, int64_t* d, int64_t e,
                           unsigned count) {
  while (*d < e && count > 0) {
    to[*d] = from[s];
    (*d)++;
    s++;
    count--;
  }
  return count;
}

/* deflate.lisp:337-359 copy-history.  Returns R_CONTINUE, R_EOO, or <0. */
static int copy_history(tbzo_state* st, unsigned count, unsigned offset) {
  int64_t d = st->output_offset;
  int64_t s = d - (int64_t)offset;
  int64_t e = (int64_t)st->output_len;
  unsigned n = count;
  unsigned total = count;
  if (s < 0) {
    if (!st->window) return fail(st, TBZO_E_NO_WINDOW, "no window?"); /* :344-345 */
    unsigned c = (unsigned)((int64_t)count < -s ? (int64_t)count : -s);
    unsigned left = copy_bytes(st->window, 32768 + s, st->output_buffer, &d, e, c); /* :347-349 */
    total -= (c - left);
    if (left) goto overflow;
    n -= c;
    s = 0;
  }
  if (n > 0) {
    unsigned left = copy_bytes(st->output_buffer, s, st->output_buffer, &d, e, n); /* :354-355 */
    total -= (n - left);
    if (left) goto overflow;
  }
  st->output_offset += count; /* :358-359 */
  return R_CONTINUE;
overflow: /* :263-269 */
  st->bytes_to_copy = total;
  st->copy_offset = offset;
  st->current_state = S_CONTINUE_COPY_HISTORY;
  st->output_offset = d;
  st->output_overflow = 1;
  {
    int r = save_window(st);
    if (r < 0) return r;
  }
  return R_EOO;
}

/* deflate.lisp:92-730 decompress-deflate.  Returns output-offset or <0. */
static int64_t decompress_deflate(tbzo_context* c, tbzo_state* st) {
  st->output_overflow = 0; /* :102-103 */
  st->input_underrun = 0;
  uint64_t v = 0;
#define EOI()               \
  do {                      \
    st->input_underrun = 1; \
    goto exit_loop;         \
  } while (0)
#define EOO()                  \
  do {                         \
    int r_ = save_window(st);  \
    if (r_ < 0) return r_;     \
    goto exit_loop;            \
  } while (0)
#define NEXT(s)              \
  do {                       \
    st->current_state = (s); \
    goto dispatch;           \
  } while (0)

dispatch:
  switch (st->current_state) { /* :82-84 */
    case S_START_OF_BLOCK: { /* :518-528 */
      if (!get_bits(c, st, 3, &v)) EOI();
      st->last_block_flag = (int)(v & 1);
      switch ((v >> 1) & 3) {
        case 0:
          NEXT(S_UNCOMPRESSED_BLOCK);
        case 1:
          st->current_is_static = 1;
          NEXT(S_DECODE_COMPRESSED_DATA);
        case 2:
          st->current_is_static = 0;
          NEXT(S_DYNAMIC_HUFFMAN_BLOCK);
        default:
          return fail(st, TBZO_E_BTYPE, "reserved block type 3");
      }
    }
    case S_UNCOMPRESSED_BLOCK: { /* :532-537 */
      /* byte-align then bits* 16 16; on underrun the alignment has happened and is idempotent */
      byte_align(st);
      if (!get_bits(c, st, 32, &v)) EOI();
      unsigned s = (unsigned)(v & 0xffff), n = (unsigned)((v >> 16) & 0xffff);
      if (n != ((~s) & 0xffff)) return fail(st, TBZO_E_STORED_LEN, "stored block LEN/NLEN mismatch");
      st->bytes_to_copy = s;
      NEXT(S_COPY_BLOCK);
    }
    case S_COPY_BLOCK: { /* :538-573 — copy-byte-or-fail: overflow is checked before input */
      while (st->bytes_to_copy > 0) {
        if (st->output_offset >= (int64_t)st->output_len) {
          st->output_overflow = 1;
          EOO();
        }
        if (!get_bits(c, st, 8, &v)) EOI();
        st->output_buffer[st->output_offset++] = (uint8_t)v;
        st->bytes_to_copy--;
      }
      NEXT(S_BLOCK_END);
    }
    case S_DYNAMIC_HUFFMAN_BLOCK: { /* :577-595 */
      if (!get_bits(c, st, 26, &v)) EOI();
      unsigned hlit = v & 31, hdist = (v >> 5) & 31, hclen = (v >> 10) & 15;
      memset(st->dht_len_codes, 0, sizeof st->dht_len_codes);
      st->dht_len_codes[16] = (v >> 14) & 7;
      st->dht_len_codes[17] = (v >> 17) & 7;
      st->dht_len_codes[18] = (v >> 20) & 7;
      st->dht_len_codes[0] = (v >> 23) & 7;
      st->dht_hlit = (int)hlit + 257;
      st->dht_hlit_hdist = st->dht_hlit + (int)hdist + 1;
      st->dht_hclen = (int)hclen;
      st->dht_lld_index = 0;
      NEXT(S_DHT_LEN_TABLE);
    }
    case S_DHT_LEN_TABLE: { /* :597-624 */
      int bitcount = st->dht_hclen * 3;
      if (!get_bits(c, st, bitcount, &v)) EOI();
      for (int i = 4, o = 0, k = 0; k < st->dht_hclen; i++, o += 3, k++)
        st->dht_len_codes[LEN_CODE_ORDER[i]] = (v >> o) & 7;
      int count, bits, max;
      int r = build_tree_part(&st->dht_len_tree, 0, st->dht_len_codes, T_DHTLEN, 0, 19, &st->ht_scratch,
                              LEN_CODE_EXTRA, &count, &bits, &max);
      if (r < 0) return fail(st, r, "bad code-length code");
      st->dht_len_tree.start_bits = bits;
      st->dht_len_tree.max_bits = max;
      st->dht_last_len = 0xff;
      NEXT(S_DHT_LEN_TABLE_DATA);
    }
    case S_DHT_LEN_TABLE_DATA: { /* :626-669 */
      int end = st->dht_hlit_hdist;
      while (st->dht_lld_index < end) {
        sym sy;
        refill(c, st);
        int r = peek_symbol(st, &st->dht_len_tree, 0, &sy);
        if (r == 0) EOI();
        if (r < 0) return fail(st, r, "invalid node in code-length tree");
        consume(st, sy.consumed);
        unsigned code = sy.value, extra = sy.extra;
        if (code < 16) {
          st->dht_lld[st->dht_lld_index++] = (uint8_t)code;
          st->dht_last_len = (int)code;
        } else if (code == 16) {
          if (!(st->dht_last_len < 16))
            return fail(st, TBZO_E_REPEAT_NO_PREV, "tried to repeat length without previous length");
          int e = st->dht_lld_index + (int)extra + 3;
          if (e > end) return fail(st, TBZO_E_REPEAT_OVERRUN, "repeat past end of code lengths");
          for (int i = st->dht_lld_index; i < e; i++) st->dht_lld[i] = (uint8_t)st->dht_last_len;
          st->dht_lld_index = e;
        } else {
          int cc = (code == 17) ? 3 : 11;
          int e = st->dht_lld_index + (int)extra + cc;
          if (e > end) return fail(st, TBZO_E_REPEAT_OVERRUN, "repeat past end of code lengths");
          for (int i = st->dht_lld_index; i < e; i++) st->dht_lld[i] = 0;
          st->dht_lld_index = e;
          st->dht_last_len = 0;
        }
      }
      int r = build_trees_star(&st->dyn_len, &st->dyn_dist, st->dht_lld, st->dht_hlit, st->dht_lld_index,
                               &st->ht_scratch); /* :663-668 */
      if (r < 0) return fail(st, r, "bad literal/length or distance code");
      NEXT(S_DECODE_COMPRESSED_DATA);
    }
    case S_DECODE_COMPRESSED_DATA: { /* :673-702 */
      const hufftree* lt = st->current_is_static ? &g_static_len : &st->dyn_len;
      const hufftree* dt = st->current_is_static ? &g_static_dist : &st->dyn_dist;
      for (;;) {
        sym a, b;
        refill(c, st);
        int r = peek_symbol(st, lt, 0, &a);
        if (r == 0) EOI();
        if (r < 0) return fail(st, r, "invalid literal/length code");
        if (a.type == HT_LEN_DIST) {
          unsigned octets = a.extra + LEN_DIST_BASES[a.value]; /* :682 */
          r = peek_symbol(st, dt, a.consumed, &b);              /* :687-689 */
          if (r == 0) EOI(); /* nothing consumed: length code is re-read on resume (:411-426) */
          if (r < 0) return fail(st, r, "invalid distance code");
          consume(st, a.consumed + b.consumed);
          r = copy_history(st, octets, LEN_DIST_BASES[b.value] + b.extra); /* :691 */
          if (r < 0) return r;
          if (r == R_EOO) goto exit_loop;
        } else if (a.type == HT_LITERAL) { /* :692-698 */
          consume(st, a.consumed);
          if (st->output_offset >= (int64_t)st->output_len) {
            st->current_state = S_OUT_BYTE;
            st->bytes_to_copy = a.value;
            st->output_overflow = 1;
            EOO();
          }
          st->output_buffer[st->output_offset++] = (uint8_t)a.value;
        } else { /* +ht-link/end+ :699-702 */
          if (a.value != 0 || a.extra != 0) return fail(st, TBZO_E_END_NODE, "bad end node");
          consume(st, a.consumed);
          NEXT(S_BLOCK_END);
        }
      }
    }
    case S_CONTINUE_COPY_HISTORY: { /* :705-707 */
      int r = copy_history(st, st->bytes_to_copy, st->copy_offset);
      if (r < 0) return r;
      if (r == R_EOO) goto exit_loop;
      NEXT(S_DECODE_COMPRESSED_DATA);
    }
    case S_OUT_BYTE: { /* :709-716 */
      if (st->output_offset >= (int64_t)st->output_len)
        return fail(st, TBZO_E_STATE,
                    "tried to continue from overflow without providing more space in output");
      st->output_buffer[st->output_offset++] = (uint8_t)st->bytes_to_copy;
      NEXT(S_DECODE_COMPRESSED_DATA);
    }
    case S_BLOCK_END: /* :719-722 */
      if (st->last_block_flag) NEXT(S_DONE);
      NEXT(S_START_OF_BLOCK);
    case S_DONE: /* :725-726 */
      st->finished = 1;
      break;
    default:
      return fail(st, TBZO_E_STATE, "bad state");
  }
exit_loop:
  return st->output_offset; /* :729-730 */
#undef EOI
#undef EOO
#undef NEXT
}

/* ---- zlib.lisp ------------------------------------------------------------- */
/* zlib.lisp:14-37 check-zlib-header */
static int check_zlib_header(tbzo_state* st, unsigned cmf, unsigned flg) {
  unsigned cm = cmf & 15, cinfo = cmf >> 4;
  if (((cmf * 256 + flg) % 31) != 0) return fail(st, TBZO_E_ZLIB_HEADER, "invalid zlib header checksum");
  if (cm != 8) return fail(st, TBZO_E_ZLIB_HEADER, "invalid zlib compression type");
  if (cinfo > 7) return fail(st, TBZO_E_ZLIB_HEADER, "invalid window size in zlib header");
  if (flg & 0x20) return fail(st, TBZO_E_ZLIB_DICT, "preset dictionary not supported yet");
  return 0;
}
/* zlib.lisp:80-96 `adler` */
static int zlib_adler(tbzo_context* c, tbzo_state* st, int* underrun) {
  uint64_t v = 0;
  *underrun = 0;
  if (st->bits_remaining < 32) refill(c, st);
  if (st->bits_remaining < 32) {
    st->input_underrun = 1;
    *underrun = 1;
    return 0;
  }
  uint32_t got = 0;
  for (int i = 0; i < 4; i++) {
    get_bits(c, st, 8, &v);
    got = (got << 8) | (uint32_t)v; /* big-endian */
  }
  uint32_t calc = st->s1 | (st->s2 << 16);
  if (got != calc) return fail(st, TBZO_E_ADLER, "adler32 mismatch");
  st->finished = 1;
  return 0;
}
/* zlib.lisp:39-144 decompress-zlib */
static int64_t decompress_zlib(tbzo_context* c, tbzo_state* st) {
  uint64_t v = 0;
  int r, under;
  st->input_underrun = 0; /* :107 */
  if (st->zlib_state != Z_NIL) {
    switch (st->zlib_state) {
      case Z_HEADER: /* :110-128 */
        if (st->bits_remaining < 16) refill(c, st);
        if (st->bits_remaining < 16) {
          st->input_underrun = 1;
          return 0;
        }
        {
          get_bits(c, st, 8, &v);
          unsigned cmf = (unsigned)v;
          get_bits(c, st, 8, &v);
          unsigned flg = (unsigned)v;
          r = check_zlib_header(st, cmf, flg);
          if (r < 0) return r;
        }
        break;
      case Z_HEADER2: /* :129-130 */
        return fail(st, TBZO_E_ZLIB_DICT, "preset dictionary not supported yet");
      case Z_ADLER: /* :131-134 */
        r = zlib_adler(c, st, &under);
        if (r < 0) return r;
        if (under) return st->output_offset;
        st->zlib_state = Z_NIL;
        return st->output_offset;
    }
    st->zlib_state = Z_NIL; /* :135 */
  }
  { /* :136-143 */
    int64_t rr = decompress_deflate(c, st);
    if (rr < 0) return rr;
    if (st->finished || st->output_overflow) /* update-checksum :97-102 */
      tbzo_adler32(st->output_buffer, (size_t)st->output_offset, st->s1, st->s2, &st->s1, &st->s2);
    if (st->finished) {
      byte_align(st);
      st->zlib_state = Z_ADLER;
      st->finished = 0;
    }
  }
  if (st->zlib_state == Z_ADLER) {
    r = zlib_adler(c, st, &under);
    if (r < 0) return r;
  }
  return st->output_offset;
}

/* ---- gzip.lisp ------------------------------------------------------------- */
/* the metadata slots as they stand (test inspection of gzip.lisp:17-25) */
int tbzo_get_gzip_meta(const tbzo_state* st, tbzo_gzip_meta* m) {
  memset(m, 0, sizeof(*m));
  m->have_cm = st->gz_have_cm;
  m->flg = (uint32_t)st->gz_flags;
  m->have_mtime = st->gz_have_mtime;
  m->mtime = st->gz_mtime;
  m->have_os = st->gz_have_os;
  m->xfl = st->gz_xfl;
  m->os = st->gz_os;
  m->have_extra = st->gz_have_extra;
  m->extra = st->gz_extra;
  m->extra_len = st->gz_have_extra ? (size_t)st->gz_extra_len : 0;
  m->have_name = st->gz_have_name;
  m->name = st->gz_name;
  m->name_len = st->gz_name_len;
  m->have_comment = st->gz_have_comment;
  m->comment = st->gz_comment;
  m->comment_len = st->gz_comment_len;
  return 0;
}
static int need_bits(tbzo_context* c, tbzo_state* st, int n) {
  if (st->bits_remaining < n) refill(c, st);
  return st->bits_remaining >= n;
}
/* gzip.lisp:67-72 header-byte */
static unsigned header_byte(tbzo_context* c, tbzo_state* st) {
  uint64_t v = 0;
  get_bits(c, st, 8, &v);
  if (st->gz_keep_header) {
    if (st->gz_header_len == st->gz_header_cap) {
      st->gz_header_cap = st->gz_header_cap ? st->gz_header_cap * 2 : 64;
      st->gz_header_bytes = (uint8_t*)realloc(st->gz_header_bytes, st->gz_header_cap);
    }
    st->gz_header_bytes[st->gz_header_len++] = (uint8_t)v;
  }
  return (unsigned)v;
}
/* gzip.lisp:30-287 decompress-gzip */
static int64_t decompress_gzip(tbzo_context* c, tbzo_state* st) {
  uint64_t v = 0;
  st->input_underrun = 0; /* :109 */
#define GZ_NEED(n)            \
  if (!need_bits(c, st, n)) { \
    st->input_underrun = 1;   \
    return 0;                 \
  }
  while (!(st->finished || st->output_overflow || st->input_underrun)) { /* :110-111 */
    switch (st->gzip_state) {
      case G_HEADER: { /* :113-122 */
        GZ_NEED(16);
        unsigned id1 = header_byte(c, st), id2 = header_byte(c, st);
        if (id1 != 0x1f || id2 != 0x8b) return fail(st, TBZO_E_GZIP_MAGIC, "bad gzip magic");
        st->gzip_state = G_HEADER2;
        break;
      }
      case G_HEADER2: { /* :123-143 */
        GZ_NEED(16);
        unsigned cm = header_byte(c, st), flg = header_byte(c, st);
        if (cm != 8) return fail(st, TBZO_E_GZIP_METHOD, "unknown compression method");
        if ((flg >> 5) & 7) return fail(st, TBZO_E_GZIP_FLAGS, "reserved flag bits set");
        st->gz_flags = (int)flg;
        st->gz_have_cm = 1; /* compression-method :deflate, flags pushed (:129-142) */
        if (!(flg & 2)) st->gz_keep_header = 0; /* no header crc: stop remembering */
        st->gzip_state = G_HEADER_MTIME;
        break;
      }
      case G_HEADER_MTIME: /* :144-157 */
        GZ_NEED(32);
        {
          uint32_t m = 0;
          for (int i = 0; i < 4; i++) m |= (uint32_t)header_byte(c, st) << (8 * i);
          st->gz_mtime = m; /* mtime/unix, mtime/universal only when non-zero (:150-155) */
          st->gz_have_mtime = 1;
        }
        st->gzip_state = G_HEADER3;
        break;
      case G_HEADER3: /* :159-179 */
        GZ_NEED(16);
        st->gz_xfl = header_byte(c, st); /* compression-level (:165-167) */
        st->gz_os = header_byte(c, st);  /* operating-system (:168-175) */
        st->gz_have_os = 1;
        st->gzip_state = G_HEADER_EXTRA;
        break;
      case G_HEADER_EXTRA: /* :180-199 */
        if (st->gz_flags & 4) {
          if (st->gz_extra_len < 0) {
            GZ_NEED(16);
            unsigned lo = header_byte(c, st), hi = header_byte(c, st);
            st->gz_extra_len = lo | (hi << 8);
            st->gz_extra_read = 0;
            st->gz_extra = (uint8_t*)calloc((size_t)st->gz_extra_len + 1, 1);
          }
          while (st->gz_extra_read < st->gz_extra_len) {
            GZ_NEED(8);
            st->gz_extra[st->gz_extra_read] = (uint8_t)header_byte(c, st);
            st->gz_extra_read++;
          }
          st->gz_have_extra = 1; /* coerced to a simple octet vector once complete (:195) */
        }
        st->gzip_state = G_HEADER_NAME;
        break;
      case G_HEADER_NAME: /* :200-221 (string decoding via babel is metadata, not restated) */
        if (st->gz_flags & 8) {
          for (;;) {
            GZ_NEED(8);
            unsigned b = header_byte(c, st);
            if (b == 0) break;
            if (st->gz_name_len == st->gz_name_cap) {
              st->gz_name_cap = st->gz_name_cap ? st->gz_name_cap * 2 : 16;
              st->gz_name = (uint8_t*)realloc(st->gz_name, st->gz_name_cap);
            }
            st->gz_name[st->gz_name_len++] = (uint8_t)b;
          }
          st->gz_have_name = 1; /* the octets; the string (utf-8, else iso-8859-1: :209-217) is made by the binding */
        }
        st->gzip_state = G_HEADER_COMMENT;
        break;
      case G_HEADER_COMMENT: /* :222-243 */
        if (st->gz_flags & 16) {
          for (;;) {
            GZ_NEED(8);
            unsigned b = header_byte(c, st);
            if (b == 0) break;
            if (st->gz_comment_len == st->gz_comment_cap) {
              st->gz_comment_cap = st->gz_comment_cap ? st->gz_comment_cap * 2 : 16;
              st->gz_comment = (uint8_t*)realloc(st->gz_comment, st->gz_comment_cap);
            }
            st->gz_comment[st->gz_comment_len++] = (uint8_t)b;
          }
          st->gz_have_comment = 1;
        }
        st->gzip_state = G_HEADER_CRC;
        break;
      case G_HEADER_CRC: /* :244-266 — crc16 bytes are read with %bits, not header-byte */
        if (st->gz_flags & 2) {
          GZ_NEED(16);
          get_bits(c, st, 8, &v);
          unsigned crc = (unsigned)v;
          get_bits(c, st, 8, &v);
          crc |= (unsigned)v << 8;
          uint32_t full = tbzo_crc32(st->gz_header_bytes, st->gz_header_len, 0);
          if (crc != (full & 0xffff)) return fail(st, TBZO_E_GZIP_HCRC, "gzip header crc mismatch");
        }
        st->gzip_state = G_DEFLATE;
        break;
      case G_DEFLATE: { /* :267-274 */
        int64_t rr = decompress_deflate(c, st);
        if (rr < 0) return rr;
        if (st->finished || st->output_overflow) /* update-checksum :80-81 */
          st->crc32 = tbzo_crc32(st->output_buffer, (size_t)st->output_offset, st->crc32);
        if (st->finished) {
          byte_align(st);
          st->gzip_state = G_FINAL_CRC;
          st->finished = 0;
        }
        break;
      }
      case G_FINAL_CRC: { /* :275-276 → crc :82-94 (header-byte is used; header-bytes may be live) */
        GZ_NEED(32);
        uint32_t crc = 0;
        for (int i = 0; i < 4; i++) crc |= (uint32_t)header_byte(c, st) << (8 * i);
        if (crc != st->crc32) return fail(st, TBZO_E_CRC, "crc32 mismatch");
        st->gzip_state = G_FINAL_LEN;
        break;
      }
      case G_FINAL_LEN: { /* :277-286 → len :95-106; ISIZE is read, never compared */
        GZ_NEED(32);
        uint32_t len = 0;
        for (int i = 0; i < 4; i++) len |= (uint32_t)header_byte(c, st) << (8 * i);
        st->gz_isize = len;
        st->finished = 1;
        st->gzip_state = G_DONE;
        goto out;
      }
      default: /* :done is not an ecase clause: calling again is an error (:280-285) */
        return fail(st, TBZO_E_STATE, "gzip state :done — decompress called after finish");
    }
  }
out:
  return st->output_offset;
#undef GZ_NEED
}

/* ---- api.lisp -------------------------------------------------------------- */
/* api.lisp:3-10 */
int64_t tbzo_decompress(tbzo_context* ctx, tbzo_state* st) {
  switch (st->format) {
    case TBZO_GZIP:
      return decompress_gzip(ctx, st);
    case TBZO_ZLIB:
      return decompress_zlib(ctx, st);
    default:
      return decompress_deflate(ctx, st);
  }
}
/* api.lisp:12-21 */
int tbzo_replace_output_buffer(tbzo_state* st, uint8_t* buf, size_t len) {
  if (!(st->output_offset == 0 || st->output_overflow))
    return fail(st, TBZO_E_REPLACE_BUFFER, "can't switch buffers without filling old one yet.");
  st->output_buffer = buf;
  st->output_len = len;
  st->output_offset = 0;
  st->output_overflow = 0;
  return 0;
}
/* api.lisp:36-47 — :output supplied */
int64_t tbzo_decompress_vector_into(const uint8_t* in, size_t start, size_t end, int format, uint8_t* out,
                                    size_t out_len) {
  tbzo_state* st = tbzo_make_state(format, out, out_len);
  tbzo_context c = {in, start, end, start};
  st->output_offset = 0;
  int64_t r = tbzo_decompress(&c, st);
  if (r >= 0 && !st->finished) {
    if (st->input_underrun)
      r = TBZO_E_INCOMPLETE_STREAM;
    else if (st->output_overflow)
      r = TBZO_E_NO_SPACE;
    else
      r = TBZO_E_STATE;
  }
  tbzo_free_state(st);
  return r;
}
/* api.lisp:48-65 — no :output: min(len,32768) then doubling, collect parts, gather */
int64_t tbzo_decompress_vector(const uint8_t* in, size_t start, size_t end, int format, uint8_t** out) {
  tbzo_state* st = tbzo_make_state(format, NULL, 0);
  tbzo_context c = {in, start, end, start};
  typedef struct {
    uint8_t* p;
    size_t n;
  } part;
  part* parts = NULL;
  size_t nparts = 0, cap = 0;
  size_t sz = (end - start) < 32768 ? (end - start) : 32768;
  int64_t r = 0;
  for (;;) {
    uint8_t* buf = (uint8_t*)malloc(sz ? sz : 1);
    r = tbzo_replace_output_buffer(st, buf, sz);
    if (r < 0) {
      free(buf);
      break;
    }
    r = tbzo_decompress(&c, st);
    if (r < 0) {
      free(buf);
      break;
    }
    if (st->input_underrun) { /* (assert (not (ds-input-underrun state))) :55 */
      free(buf);
      r = TBZO_E_INCOMPLETE_STREAM;
      break;
    }
    if (r == 0) {
      free(buf);
      if (!st->finished) { /* (assert (ds-finished state)) :57 */
        r = TBZO_E_STATE;
        break;
      }
    } else {
      if (nparts == cap) {
        cap = cap ? cap * 2 : 8;
        parts = (part*)realloc(parts, cap * sizeof *parts);
      }
      parts[nparts].p = buf;
      parts[nparts].n = (size_t)r;
      nparts++;
    }
    if (st->finished) break;
    sz *= 2; /* (* 2 (length out)) :52 */
  }
  int64_t total = -1;
  if (r >= 0) {
    total = 0;
    for (size_t i = 0; i < nparts; i++) total += (int64_t)parts[i].n;
    uint8_t* b = (uint8_t*)malloc(total ? (size_t)total : 1);
    size_t o = 0;
    for (size_t i = 0; i < nparts; i++) {
      memcpy(b + o, parts[i].p, parts[i].n);
      o += parts[i].n;
    }
    *out = b;
  } else {
    total = r;
    *out = NULL;
  }
  for (size_t i = 0; i < nparts; i++) free(parts[i].p);
  free(parts);
  tbzo_free_state(st);
  return total;
}

int tbzo_debug_build_trees(const uint8_t* lens, int hlit, int total, uint16_t* lnodes, int* lstart,
                           uint16_t* dnodes, int* dstart) {
  hufftree *l = (hufftree*)malloc(sizeof *l), *d = (hufftree*)malloc(sizeof *d),
           *s = (hufftree*)malloc(sizeof *s);
  ht_init(l);
  ht_init(d);
  ht_init(s);
  int r = build_trees_star(l, d, lens, hlit, total, s);
  if (r >= 0) {
    memcpy(lnodes, l->nodes, sizeof l->nodes);
    memcpy(dnodes, d->nodes, sizeof d->nodes);
    *lstart = l->start_bits;
    *dstart = d->start_bits;
  }
  free(l);
  free(d);
  free(s);
  return r;
}
