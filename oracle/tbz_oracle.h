/* tbz_oracle.h — CPU restatement of 3bz's inflate path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (3bz_amd/, include/) may
 * include, link or call this.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it, and only as the checker / reported
 * baseline.
 *
 * Parity pin: the 37 known-answer vectors of deflate-test.lisp:69-302 and
 * test.deflated (tests/golden/), plus system zlib as an independent cross
 * check on generated corpora.  The reference itself is Common Lisp with
 * un-vendored dependencies and no Lisp exists in the image, so oracle/_ref
 * (a build of the reference) is not possible: see DESIGN.md.
 */
#ifndef TBZ_ORACLE_H
#define TBZ_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* api.lisp:31-34 — format keyword of decompress-vector */
enum { TBZO_DEFLATE = 0, TBZO_ZLIB = 1, TBZO_GZIP = 2 };

/* Lisp conditions become negative return codes; message via tbzo_errmsg. */
enum {
  TBZO_OK = 0,
  TBZO_E_BTYPE = -1,          /* deflate.lisp:521 ecase fall-through (BTYPE 3) */
  TBZO_E_STORED_LEN = -2,     /* deflate.lisp:535 assert LEN = ~NLEN */
  TBZO_E_OVERSUBSCRIBED = -3, /* huffman-tree.lisp:116-117 */
  TBZO_E_INCOMPLETE = -4,     /* huffman-tree.lisp:119-122 */
  TBZO_E_REPEAT_NO_PREV = -5, /* deflate.lisp:642-643 */
  TBZO_E_REPEAT_OVERRUN = -6, /* deflate.lisp:645, :656 assert */
  TBZO_E_INVALID_NODE = -7,   /* deflate.lisp:438/:481 ecase has no +ht-invalid+ clause */
  TBZO_E_NO_WINDOW = -8,      /* deflate.lisp:345 "no window?" */
  TBZO_E_ZLIB_HEADER = -9,    /* zlib.lisp:20-36 */
  TBZO_E_ZLIB_DICT = -10,     /* zlib.lisp:33-35, :77 */
  TBZO_E_ADLER = -11,         /* zlib.lisp:95 */
  TBZO_E_GZIP_MAGIC = -12,    /* gzip.lisp:120-121 */
  TBZO_E_GZIP_METHOD = -13,   /* gzip.lisp:130-132 */
  TBZO_E_GZIP_FLAGS = -14,    /* gzip.lisp:133-134 */
  TBZO_E_GZIP_HCRC = -15,     /* gzip.lisp:255 */
  TBZO_E_CRC = -16,           /* gzip.lisp:93 */
  TBZO_E_STATE = -17,         /* calling again after :done (gzip.lisp:280-285), bad resume */
  TBZO_E_TREE_OVERFLOW = -18, /* node index outside the 1444-entry array / 10-bit link field */
  TBZO_E_REPLACE_BUFFER = -19,/* api.lisp:13-19 */
  TBZO_E_INCOMPLETE_STREAM = -20, /* api.lisp:43-44 */
  TBZO_E_NO_SPACE = -21,      /* api.lisp:45-46 */
  TBZO_E_END_NODE = -22       /* deflate.lisp:700-701 asserts */
};

/* io-common.lisp:8-14, :36-45 — octet-vector-context + context-boxes */
typedef struct {
  const uint8_t* vec;
  size_t start, end, offset;
} tbzo_context;

typedef struct tbzo_state tbzo_state;

/* make-deflate-state / make-zlib-state / make-gzip-state (&key output-buffer) */
tbzo_state* tbzo_make_state(int format, uint8_t* output_buffer, size_t output_len);
void tbzo_free_state(tbzo_state*);

/* api.lisp:3-10 — returns the output offset (or gzip/zlib's early-return 0), <0 on error */
int64_t tbzo_decompress(tbzo_context* ctx, tbzo_state* st);
/* api.lisp:12-21 */
int tbzo_replace_output_buffer(tbzo_state* st, uint8_t* buf, size_t len);
/* api.lisp:67-72 */
int tbzo_finished(const tbzo_state*);
int tbzo_input_underrun(const tbzo_state*);
int tbzo_output_overflow(const tbzo_state*);
/* test switch, not reference behaviour: all-zero alphabets invalidate the table instead of leaving it stale */
void tbzo_set_fresh_tables(int on);
int64_t tbzo_output_offset(const tbzo_state*);
const char* tbzo_errmsg(const tbzo_state*);
/* checksum state of the wrapper (zs-s1 | zs-s2<<16, or gs-crc32) */
uint32_t tbzo_checksum(const tbzo_state*);

/* the gzip-state's metadata slots (gzip.lisp:17-25) as decompress-gzip has filled them so far (gzip.lisp:123-241) */
typedef struct tbzo_gzip_meta {
  int have_cm, have_mtime, have_os, have_extra, have_name, have_comment;
  uint32_t flg, mtime, xfl, os;
  const uint8_t *extra, *name, *comment;
  size_t extra_len, name_len, comment_len;
} tbzo_gzip_meta;
int tbzo_get_gzip_meta(const tbzo_state* st, tbzo_gzip_meta* out);

/* api.lisp:23-65 with :output supplied.  Returns count, or <0. */
int64_t tbzo_decompress_vector_into(const uint8_t* in, size_t start, size_t end, int format,
                                    uint8_t* out, size_t out_len);
/* api.lisp:23-65 without :output: grows 32 KiB, ×2 …, gathers. *out is malloc'd. */
int64_t tbzo_decompress_vector(const uint8_t* in, size_t start, size_t end, int format,
                               uint8_t** out);
void tbzo_free(void*);

/* checksums.lisp:18-62 (via :167-174) and :196-210 */
void tbzo_adler32(const uint8_t* buf, size_t end, uint32_t s1, uint32_t s2, uint32_t* o1, uint32_t* o2);
uint32_t tbzo_crc32(const uint8_t* buf, size_t end, uint32_t crc);

/* debug/inspection: build the two trees from a code-length array the way
 * build-trees* does and copy the node arrays out (huffman-tree.lisp:272-287). */
int tbzo_debug_build_trees(const uint8_t* lens, int hlit, int total, uint16_t* lnodes, int* lstart,
                           uint16_t* dnodes, int* dstart);

#ifdef __cplusplus
}
#endif
#endif
