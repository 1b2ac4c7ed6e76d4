/* tbz_amd.h — C ABI of the MI355X-native inflate engine that drops in behind 3bz's
 * octet-vector path.
 *
 * The reference (3bz, Common Lisp) has NO plugin / FFI interface on this path: decode is
 * plain Lisp functions and the only FFI it uses is cffi:mem-ref for pointer-context input
 * (io-mmap.lisp:71,:84,:95,:108).  The drop-in boundary is therefore 3bz's exported Lisp
 * API (package.lisp:13-27); this header is the C ABI a CFFI shim binds underneath it
 * (lisp/3bz-amd.lisp).  Every entry point cites the reference interface it
 * serves.
 *
 * Conventions
 *   - plain pointers and sizes; no C++/torch types; no exceptions cross the boundary
 *   - all buffers are BORROWED for the duration of the call, never retained
 *     (3bz: caller owns input and output vectors, io-common.lisp:3-4, deflate.lisp:47-48)
 *   - return value: 0 = call executed (look at tbz_result.status per stream),
 *                   <0 = engine failure (HIP error, bad argument); tbz_strerror(code)
 *   - tbz_result.status: TBZ_FINISHED / TBZ_INPUT_UNDERRUN / TBZ_OUTPUT_OVERFLOW are 3bz's
 *     three status flags (api.lisp:67-72; they are flags, not conditions); <0 is a Lisp
 *     `error`/`assert`/`ecase` failure of the reference (SURVEY §8a contract list)
 *   - a context is not thread-safe (one in-flight call); several contexts may coexist
 *     (3bz: one state must not be used concurrently)
 */
#ifndef TBZ_AMD_H
#define TBZ_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TBZ_ABI_VERSION 4

/* decompress-vector's :format keyword (api.lisp:31-34) */
enum { TBZ_FORMAT_DEFLATE = 0, TBZ_FORMAT_ZLIB = 1, TBZ_FORMAT_GZIP = 2 };

/* per-stream status */
enum {
  TBZ_FINISHED = 0,        /* (finished state)        api.lisp:67-68 */
  TBZ_INPUT_UNDERRUN = 1,  /* (input-underrun state)  api.lisp:69-70, deflate.lisp:114-120 */
  TBZ_OUTPUT_OVERFLOW = 2, /* (output-overflow state) api.lisp:71-72, deflate.lisp:121-137 */
  /* reference errors (Lisp conditions) */
  TBZ_E_BTYPE = -1,          /* deflate.lisp:521 */
  TBZ_E_STORED_LEN = -2,     /* deflate.lisp:535 */
  TBZ_E_OVERSUBSCRIBED = -3, /* huffman-tree.lisp:116-117 */
  TBZ_E_INCOMPLETE = -4,     /* huffman-tree.lisp:119-122 */
  TBZ_E_REPEAT_NO_PREV = -5, /* deflate.lisp:642-643 */
  TBZ_E_REPEAT_OVERRUN = -6, /* deflate.lisp:645,:656 */
  TBZ_E_INVALID_CODE = -7,   /* invalid node reached: deflate.lisp:438,:481 */
  TBZ_E_DISTANCE = -8,       /* distance before start of output, no window: deflate.lisp:345 */
  TBZ_E_ZLIB_HEADER = -9,    /* zlib.lisp:20-32 */
  TBZ_E_ZLIB_DICT = -10,     /* zlib.lisp:33-35 */
  TBZ_E_ADLER32 = -11,       /* zlib.lisp:95 */
  TBZ_E_GZIP_MAGIC = -12,    /* gzip.lisp:120-121 */
  TBZ_E_GZIP_METHOD = -13,   /* gzip.lisp:130-132 */
  TBZ_E_GZIP_FLAGS = -14,    /* gzip.lisp:133-134 */
  TBZ_E_GZIP_HCRC = -15,     /* gzip.lisp:255 */
  TBZ_E_CRC32 = -16,         /* gzip.lisp:93 */
  /* engine failures (also used as function return codes) */
  TBZ_E_ARG = -100,
  TBZ_E_HIP = -101,     /* a HIP runtime call failed; tbz_last_error(ctx) has the text */
  TBZ_E_NOMEM = -102,
  TBZ_E_NO_DEVICE = -103,
  TBZ_E_UNSUPPORTED = -104, /* e.g. resuming a stream mid-way on the device path */
  TBZ_E_INTERNAL = -105
};

/* what one call reports per stream: 64 bytes, also the record exchanged between ranks in
 * the batched multi-GPU case */
typedef struct tbz_result {
  int32_t status;        /* TBZ_FINISHED / _INPUT_UNDERRUN / _OUTPUT_OVERFLOW or <0 */
  uint32_t segments;     /* independent segments the stream was split into */
  uint64_t out_len;      /* octets valid in the output = what `decompress` returns
                            (deflate.lisp:730); for gzip/zlib early returns see api notes */
  uint64_t out_total;    /* full decompressed size when known (== out_len when finished) */
  uint64_t in_consumed;  /* finished: input octets consumed including header and trailer.  Otherwise the
                            latest octet-aligned block boundary the decoder is sure of: the offset just
                            after the last flush marker (00 00 FF FF) the block chain landed on, or the
                            end of the input when it ran out exactly where a block starts; 0 if none.
                            Everything before it is decoded and delivered.  (in_consumed == in_len with
                            status input-underrun: the input is a whole number of blocks — this is how a
                            stream sharded across GPUs proves its seams, 3bz_amd/multi.py) */
  uint32_t adler32;      /* computed over the output (zlib.lisp:97-102), s1 | s2<<16 */
  uint32_t crc32;        /* computed over the output (gzip.lisp:80-81) */
  uint32_t trailer_check;/* checksum stored in the stream trailer (0 if not reached) */
  uint32_t trailer_isize;/* gzip ISIZE (read, never compared: gzip.lisp:95-106,:278) */
  uint32_t flags;        /* bit0: checksum verified against trailer; bit1: BFINAL block decoded; bit2: the
                            input ran out inside a stored block's payload — there the reference asks for
                            output space before input (deflate.lisp:538-573), so with the output exactly
                            full the status is output-overflow, and a host that hands the output out in
                            pieces must report the same when a piece ends there */
  uint32_t reserved;
  uint64_t boundary_out; /* not finished: output octets produced by the input before in_consumed (a decoder
                            restarted at that boundary continues the output there); finished: out_len */
} tbz_result;

/* durations of the device stages of the LAST call, from HIP events recorded on the
 * context's stream (ms), and what the engine launched.  bench.py builds roofline.achieved and
 * roofline.kernel from these. */
typedef struct tbz_timings {
  float scan_ms;     /* K0 scan_markers (count + scan + emit) + K0b block-start finder */
  float huff_ms;     /* K1 huff_decode (all rounds) */
  float lz_ms;       /* K2 lz77_resolve (both planes when segments need history they do not hold) */
  float cksum_ms;    /* K4/K5 checksum partials + combine */
  float total_ms;    /* first kernel start .. last kernel end (device time, includes host chain gaps) */
  uint32_t huff_launches;
  uint32_t fixup_rounds;
  uint64_t token_words; /* u16 token words written by K1 (traffic accounting) */
  uint64_t n_segments;
  uint64_t n_groups;
  /* ---- ABI 2 */
  float find_ms;     /* K0b: speculative block-start candidates (part of scan_ms) */
  float resolve_ms;  /* K6: window propagation + marker resolution across LZ77 groups */
  uint32_t k1_gang;  /* K1 flavour of the main launch: 1 = one lane per item, 8/16/32/64 = gang width */
  uint32_t k2_kinds; /* K2 kernels launched: bit0 tbz_k2_lz77_dual, bit1 tbz_k2_lz77_small, bit2 tbz_k2_lz77 [ring],
                        bit3 the ring kernel's second (pointer high octet) plane */
  uint64_t n_candidates; /* block starts proposed by K0b */
  uint64_t n_hgroups;    /* LZ77 groups decoded against a symbolic 32 KiB history (resolved by K6) */
  uint64_t scratch_bytes;/* device scratch held by the context after the call (token pool, run tables, ...) */
  uint32_t h2d_copies;   /* host->device input copies of the call (tbz_inflate / _size: 1 per staging) */
  uint32_t passes;       /* 0/1: one pass; otherwise the batch was decoded in this many passes over consecutive streams
                            because its scratch would have exceeded the pool cap */
  /* ---- ABI 4: the three legs of a host-buffer call (tbz_inflate / _batch / _size), host wall-clock ms; 0 after a
   * device-buffer call.  h2d_ms ends when the last input chunk has been handed to the DMA engine. */
  float h2d_ms;
  float host_decode_ms;
  float d2h_ms;
  uint32_t reserved4;
} tbz_timings;

typedef struct tbz_ctx tbz_ctx;

/* ---- context ------------------------------------------------------------------------
 * An engine handle bound to one HIP device: stream, scratch pools, constant tables.
 * Replaces nothing in 3bz (a deflate-state is self-contained, deflate.lisp:4-62); it is
 * the opaque handle the Lisp shim keeps in a special variable. */
int tbz_ctx_create(int device_id, tbz_ctx** out_ctx);
void tbz_ctx_destroy(tbz_ctx* ctx);
int tbz_abi_version(void);
/* Release the context's device scratch (token pools, run tables, staging buffers: they only grow otherwise and are
 * held until tbz_ctx_destroy).  A long-lived host calls this after an unusually large call. */
int tbz_ctx_trim(tbz_ctx* ctx);
const char* tbz_strerror(int code);
const char* tbz_last_error(const tbz_ctx* ctx);
int tbz_device_count(void);

/* ---- one-shot decode, host buffers ---------------------------------------------------
 * (decompress-vector compressed :format f :start s :end e :output out), api.lisp:23-47.
 * `in`/`out` are host memory (pinned Lisp vectors via cffi:with-pointer-to-vector-data,
 * the pattern of bench.lisp:61).  Status and count land in *res.  With a too-small `out`
 * the buffer is filled with the correct prefix and status is TBZ_OUTPUT_OVERFLOW, exactly
 * what the reference leaves behind (deflate.lisp:254-269,:693-697). */
int tbz_inflate(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, uint8_t* out, size_t out_cap,
                tbz_result* res);

/* Size query so the shim allocates exactly once instead of 3bz's 32 KiB-then-double loop
 * (api.lisp:48-65).  out_total/status as tbz_inflate would report with unlimited space. */
int tbz_inflate_size(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, tbz_result* res);

/* (decompress-vector compressed :format f) WITHOUT :output, in ONE decode: where the reference grows 32 KiB buffers by
 * doubling and gathers them (api.lisp:48-65), the engine decodes once (one host-to-device copy, one Huffman pass) and
 * asks for the result buffer when it knows the size: `alloc(user, n)` returns n writable octets (n may be 0), the
 * octets are copied there, res->out_len = n.  Not called when the stream fails (status < 0). */
typedef uint8_t* (*tbz_alloc_fn)(void* user, size_t n_octets);
int tbz_inflate_alloc(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, tbz_alloc_fn alloc, void* user,
                      tbz_result* res);

/* n independent streams in one call (BASELINE configs 3 and 4): stream i is
 * (decompress-vector ins[i] :format f :output outs[i]). */
int tbz_inflate_batch(tbz_ctx* ctx, int format, size_t n, const uint8_t* const* ins, const size_t* in_lens,
                      uint8_t* const* outs, const size_t* out_caps, tbz_result* results);

/* ---- one-shot decode, device-resident buffers ------------------------------------------
 * Same contract with input and output already in HBM (what bench.py times; also the
 * pointer-context analogue of with-octet-pointer, io-mmap.lisp:26-54).  Stream i reads
 * d_in_base[in_offs[i] .. +in_lens[i]) and writes d_out_base[out_offs[i] .. +out_caps[i]). */
int tbz_inflate_device(tbz_ctx* ctx, int format, const void* d_in, size_t in_len, void* d_out, size_t out_cap,
                       tbz_result* res);
int tbz_inflate_batch_device(tbz_ctx* ctx, int format, size_t n, const void* d_in_base, const uint64_t* in_offs,
                             const uint64_t* in_lens, void* d_out_base, const uint64_t* out_offs,
                             const uint64_t* out_caps, tbz_result* results);

/* One decode of host input into DEVICE memory that the caller then owns (release it with tbz_device_free): *d_out is
 * allocated once K1 has sized the output — no sizing pass, no second decode.  res->out_len octets at *d_out.  *d_out
 * is NULL whenever the call returns non-zero (TBZ_E_NOMEM: the device refused the output allocation) and also for an
 * output of zero octets; a stream-level error (res->status != TBZ_OK) still returns 0 with the octets decoded before
 * it at *d_out. */
int tbz_inflate_to_device(tbz_ctx* ctx, int format, const uint8_t* in, size_t in_len, void** d_out, tbz_result* res);

/* ---- several devices from ONE host process --------------------------------------------------------
 * north_star: "many independent streams shard across the 8 GPUs of one node".  A host that is not a torch.distributed
 * rank (the Lisp shim) creates one context per device (tbz_ctx_create(d, ...)) and hands them all to
 * tbz_inflate_batch_multi: streams are assigned longest-compressed-first to the least loaded context
 * (tbz_assign_streams: the same rule 3bz_amd/multi.py applies to ranks), every context decodes its share in ONE
 * tbz_inflate_batch call on a host thread of its own, results[i] is stream i's.  No data-path collective: a
 * deflate-state is self-contained (deflate.lisp:4-62). */
int tbz_assign_streams(const size_t* in_lens, size_t n, size_t n_parts, uint32_t* owner);
int tbz_inflate_batch_multi(tbz_ctx* const* ctxs, size_t n_ctx, int format, size_t n, const uint8_t* const* ins,
                            const size_t* in_lens, uint8_t* const* outs, const size_t* out_caps, tbz_result* results);
/* The same with the streams ALREADY RESIDENT on the devices (foreign / device pointers: io-mmap.lisp:26-54): context k
 * decodes the n_streams[k] streams at d_ins[k] + in_offs[k][i] into d_outs[k] + out_offs[k][i] — per context the
 * arguments of tbz_inflate_batch_device — all contexts at once, a host thread each; results[k][i] is context k's stream
 * i.  Nothing but the result records crosses PCIe.  The contexts must be distinct (TBZ_E_ARG; also in the host variant:
 * one context is one in-flight call). */
int tbz_inflate_batch_multi_device(tbz_ctx* const* ctxs, size_t n_ctx, int format, const size_t* n_streams,
                                   const void* const* d_ins, const uint64_t* const* in_offs, const uint64_t* const* in_lens,
                                   void* const* d_outs, const uint64_t* const* out_offs, const uint64_t* const* out_caps,
                                   tbz_result* const* results);

/* ---- ONE flush-delimited stream over several decoders ---------------------------
 * 3bz finds a block only by finishing the one before it (deflate.lisp:719-722), but a deflate stream may be ENTERED at
 * any block boundary (deflate.lisp:518-528 reads BFINAL / BTYPE with no other state), and the octet after a flush marker
 * 00 00 FF FF is one — if the marker is real.  tbz_inflate_sharded_plan cuts the stream at such markers near equal shares
 * of its octets: part r = in[cuts[r], cuts[r+1]), part 0 in the stream's own format, the others as raw deflate
 * (TBZ_FORMAT_DEFLATE).  Whoever decodes the parts — the ranks of a distributed job, the contexts of one process, one
 * context after another — hands their result records to tbz_inflate_sharded_verdict: every part but the last must have
 * run out of input (TBZ_INPUT_UNDERRUN) having consumed EXACTLY its range, the last must have finished; by induction
 * from part 0's true start every cut is then a true block boundary and the parts concatenate to what a front-to-back
 * decoder produces.  part_check[r] = checksum of part r's output continued from the format's initial value (adler32 from
 * 1, crc32 from 0; tbz_adler32_device / tbz_crc32_device); the whole's checksum is their ordered combine and is compared
 * with the trailer behind the last part.  Returns 0 (clean: *total, *check, *in_consumed, out_offs[r] filled), 1 (not
 * clean: decode the stream by the ordinary path — its statuses are the answer; *why: r+1 = the seam after part r, -1 the
 * last part did not finish, -2 trailer incomplete, -3 checksum mismatch), or a TBZ_E_* argument error. */
int tbz_inflate_sharded_plan(const uint8_t* in, size_t in_len, size_t n_parts, uint64_t* cuts /* [n_parts + 1] */);
/* The three steps in one call for a host process that holds one context per device (the Lisp shim): plan over n_ctx parts,
 * one host thread per context (stage its part, decode it into device memory, checksum it there), verdict, and — all seams
 * clean — the parts copied into `out` at their offsets (*sharded = 1).  Anything else: ctxs[0] decodes the whole stream by
 * tbz_inflate (*sharded = 0), so that statuses and errors are the single-device ones. */
int tbz_inflate_sharded_multi(tbz_ctx* const* ctxs, size_t n_ctx, int format, const uint8_t* in, size_t in_len, uint8_t* out,
                              size_t out_cap, tbz_result* res, int* sharded);
int tbz_inflate_sharded_verdict(int format, const uint8_t* in, size_t in_len, size_t n_parts, const uint64_t* cuts,
                                const tbz_result* recs, const uint32_t* part_check, uint64_t* out_offs, uint64_t* total,
                                uint32_t* check, uint64_t* in_consumed, int* why);

/* ---- every member of a concatenated gzip file ------------------------------------------------------
 * 3bz decodes ONE member per call and stops after its trailer (gzip.lisp:277-286: "todo: support multiple
 * members"); the caller is expected to call again with :start at the next member, whose offset is the consumed-octet
 * count of the call before.  Here the member starts are found on the device (every `1f 8b 08` with a legal FLG octet
 * is a candidate, gzip.lisp:113-139), all candidate ranges are decoded as one batch, and a candidate counts iff the
 * member before it FINISHED exactly there — so member k is exactly what
 *     (decompress-vector v :format :gzip :start member_in_off[k])
 * returns, with results[k] what that call reports (status, out_len, crc32, trailer, in_consumed).  Walking stops at
 * the first member that is damaged or incomplete (its error / input-underrun is results[n-1]) or where what follows a
 * member is not a member (trailing garbage is ignored, as gzip(1) does); *n_members <= max_members.
 * _device: input and output in HBM.  Member k's octets are d_out[member_out_off[k] .. + results[k].out_len); out_cap
 * should be the sum of the members' sizes plus 16 octets per member (ranges are 16-octet aligned and sized by the
 * ISIZE that ends them); a member whose ISIZE lies, or that holds a false magic, is decoded on its own and placed
 * from the end of the buffer downwards; one that finds no room reports TBZ_OUTPUT_OVERFLOW.
 * host variant: `alloc(user, n)` is called once per delivered member, in order, for its n octets (tbz_inflate_alloc). */
int tbz_inflate_gzip_members_device(tbz_ctx* ctx, const void* d_in, size_t in_len, void* d_out, size_t out_cap,
                                    size_t max_members, tbz_result* results, uint64_t* member_in_off,
                                    uint64_t* member_out_off, size_t* n_members);
int tbz_inflate_gzip_members(tbz_ctx* ctx, const uint8_t* in, size_t in_len, tbz_alloc_fn alloc, void* user,
                             size_t max_members, tbz_result* results, uint64_t* member_in_off, size_t* n_members);

/* ---- resumable decode: 3bz's chunked protocol with the state on the device ----------------------
 * A session is a deflate-state / zlib-state / gzip-state (deflate.lisp:4-62, zlib.lisp:3-12, gzip.lisp:3-28) whose
 * resumable part lives in HBM: the input not yet finished with, the 32 KiB window (deflate.lisp:121-137, :343-352)
 * and the octets decoded beyond what the caller's buffer took.  One (decompress context state) call of the reference
 * (api.lisp:3-10) is
 *     tbz_session_feed(s, <the context's octets offset..end>)   then   tbz_session_decompress(s, buffer + offset, room)
 * and (replace-output-buffer state buffer) (api.lisp:12-21) is simply the next tbz_session_decompress with the new
 * buffer.  Per call: res->status = finished / input-underrun / output-overflow exactly when the reference sets those
 * flags, res->out_len = octets written to `out` by THIS call, res->out_total = octets of the stream handed out so
 * far, res->in_consumed = (finished) octets of the stream consumed including the trailer, flags as in tbz_result.
 * A stream that turns out to be invalid reports its error in the call in which a front-to-back decoder would have
 * met it: after the output before it has been handed out (status < 0 with out_len = the last octets before it).
 * Cost per call: the new input plus one block header: the resume point is the TOKEN the input ran out in (the block's
 * header is parsed again and the token loop entered there: what deflate.lisp:399-427 does by pushing an unfinished
 * symbol's bits back); inside a stored block's payload it is the block's start (at most 64 KiB copied again).  One call
 * decodes at most 8 MiB of input ahead of what the caller's buffers have taken. */
typedef struct tbz_session tbz_session;
int tbz_session_create(tbz_ctx* ctx, int format, tbz_session** out_session);
void tbz_session_destroy(tbz_session* s);
/* `in` is host memory, or device memory when in_on_device != 0 (a pointer context over HBM, io-mmap.lisp:47-54) */
int tbz_session_feed(tbz_session* s, const void* in, size_t in_len, int in_on_device);
int tbz_session_decompress(tbz_session* s, uint8_t* out, size_t out_cap, tbz_result* res);
/* measurement: engine calls the session has made so far and the input octets handed to them (octets decoded again
 * after a resume are counted again: in_decoded / octets fed is the re-decode factor) */
int tbz_session_stats(const tbz_session* s, uint64_t* n_decodes, uint64_t* in_decoded);

/* ---- gzip header metadata (host side; no device involved) --------------------------------------
 * What decompress-gzip leaves in the gzip-state's slots while it reads the header (gzip.lisp:110-266:
 * compression-method, flags, mtime, compression level, operating system, extra / name / comment fields, header CRC).
 * The engine itself only needs to skip the header (K1 does, checking what the reference checks); a host that wants the
 * metadata — the Lisp shim fills the gzip-state's slots from it — parses the octets it holds with this function.
 * status: 0 = complete header; TBZ_INPUT_UNDERRUN = `in` ends inside the header; TBZ_E_GZIP_MAGIC / _METHOD / _FLAGS
 * / _HCRC = the error the reference signals (gzip.lisp:120-134, :255).  Offsets are into `in`; name / comment
 * lengths exclude the terminating zero octet.  Fields up to `stage` are valid whatever the status.
 * One deviation, in metadata only: the extra field is taken all or nothing.  The reference allocates `extra` from XLEN and
 * fills it octet by octet across calls (gzip.lisp:178-196), so a caller that looks at the slot while the input ends
 * INSIDE the extra field sees a partly filled vector there; here `stage` stays 4 (extra_len unset) until the whole
 * field has arrived.  Once the header is complete the slots are identical. */
typedef struct tbz_gzip_header {
  int32_t status;
  uint32_t header_len;   /* octets before the first deflate block */
  uint32_t cm, flg, mtime, xfl, os;
  uint32_t extra_off, extra_len, name_off, name_len, comment_off, comment_len;
  uint32_t hcrc_present, hcrc;
  uint32_t stage;        /* how far the header was read (the reference fills its slots as it goes): 1 magic, 2 CM+FLG,
                            3 MTIME, 4 XFL+OS, 5 extra field, 6 name, 7 comment, 8 header CRC = complete */
} tbz_gzip_header;
int tbz_gzip_header_parse(const uint8_t* in, size_t in_len, tbz_gzip_header* out);

/* ---- checksums over device memory -------------------------------------------------------
 * (adler32 buf end s1 s2) checksums.lisp:167-174 and (crc32/table buf end crc) :196-210,
 * same chaining convention: pass the previous (s1,s2) / finalised crc back in. */
int tbz_adler32_device(tbz_ctx* ctx, const void* d_buf, size_t len, uint32_t s1, uint32_t s2, uint32_t* out_s1,
                       uint32_t* out_s2);
int tbz_crc32_device(tbz_ctx* ctx, const void* d_buf, size_t len, uint32_t crc, uint32_t* out_crc);

/* ---- device memory helpers for hosts without a HIP binding (the Lisp shim) --------------- */
int tbz_device_malloc(tbz_ctx* ctx, size_t bytes, void** d_ptr);
int tbz_device_free(tbz_ctx* ctx, void* d_ptr);
int tbz_memcpy_h2d(tbz_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int tbz_memcpy_d2h(tbz_ctx* ctx, void* h_dst, const void* d_src, size_t bytes);

/* ---- measurement ------------------------------------------------------------------------- */
int tbz_last_timings(const tbz_ctx* ctx, tbz_timings* out);

#ifdef __cplusplus
}
#endif
#endif
