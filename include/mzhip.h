/* mzhip.h -- C ABI of the MI355X codec backend for minizip-ng.
 *
 * Two groups of entry points live in libmzhip.so:
 *
 * (1) The DROP-IN symbols.  They are declared by the reference's own headers
 *     and are re-implemented here with identical names, signatures and error
 *     behaviour, so that mz_zip.c / mz_zip_rw.c / compat/ are compiled
 *     unmodified and linked against this library instead of mz_strm_zlib.o,
 *     mz_strm_lzma.o and the CRC symbol of mz_crypt.o (link-time substitution,
 *     SURVEY 8b):
 *         mz_stream_zlib_{open,is_open,read,write,tell,seek,close,error,
 *                         get_prop_int64,set_prop_int64,create,delete,
 *                         get_interface}          (mz_strm_zlib.h:20-35)
 *         mz_stream_lzma_{...same 13...}          (mz_strm_lzma.h:20-35)
 *         mz_crypt_crc32_update                   (mz_crypt.h:20)
 *     They are declared in include/mz_strm_hip.h.
 *
 * (2) The BATCH entry points below -- the data-parallel path that has no
 *     analogue in the (strictly one-entry-at-a-time) reference: thousands of
 *     independent entries per launch, inputs and outputs resident in HBM.
 *     Plain pointers and sizes only; `stream` is a hipStream_t passed as
 *     void* (NULL = the default stream).  All d_* pointers are device pointers.
 *
 * Status words are numerically the reference's (mz.h:21-26): 0 OK,
 * -3 MZ_DATA_ERROR, -5 MZ_BUF_ERROR (input ended early), plus
 * MZHIP_OUT_FULL (-200) when an entry produces more than its out_cap.
 */
#ifndef MZHIP_H
#define MZHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MZHIP_API __attribute__((visibility("default")))

#define MZHIP_STATUS_OK 0
#define MZHIP_STATUS_DATA_ERROR (-3)
#define MZHIP_STATUS_BUF_ERROR (-5)
#define MZHIP_STATUS_OUT_FULL (-200)
#define MZHIP_STATUS_UNSUPPORTED (-109) /* MZ_SUPPORT_ERROR (mz.h:38): construct outside the backend's scope */

/* Library / device ------------------------------------------------------ */

/* Number of visible HIP devices, or <0 (no HIP runtime / no GPU): callers must
 * treat that as fatal -- there is no CPU fallback in this library. */
MZHIP_API int32_t mzhip_device_count(void);
/* Placement on multi-socket hosts.  mzhip_device_local_cpus: the kernel's cpulist ("64-127,192-255") of the NUMA node
 * `device` is attached to, from /sys/bus/pci/devices/<bdf>/local_cpulist; returns its length, 0 when the platform has
 * one node or does not say.  mzhip_bind_thread_near_device: restrict the CALLING thread (and the threads it creates
 * afterwards) to those CPUs, at most max_cpus of them when max_cpus > 0 (cores before their second hardware threads);
 * returns the number of CPUs bound to, 0 when nothing was changed.  The library binds only its own worker threads; an
 * application (bench.py does) calls this once per process or reader thread before it allocates and reads: page-locked
 * buffers land on the node of the thread that asks for them. */
MZHIP_API int32_t mzhip_device_local_cpus(int32_t device, char *cpulist, int32_t cap);
MZHIP_API int32_t mzhip_bind_thread_near_device(int32_t device, int32_t max_cpus);
/* Bind the calling thread to `device` and create its constant tables.
 * Idempotent.  0 or a negative MZ_* code. */
MZHIP_API int32_t mzhip_init(int32_t device);
MZHIP_API const char *mzhip_last_error(void);
MZHIP_API const char *mzhip_version(void);

/* K1+K2: raw-DEFLATE decode with fused CRC-32 ---------------------------- */

/* Replaces, for n entries at once, the per-entry loop
 *   mz_stream_zlib_read (mz_strm_zlib.c:116-193) + mz_crypt_crc32_update (mz_zip.c:2049).
 * Entry i reads   d_in  + d_in_off[i]  .. + d_in_len[i]   (raw DEFLATE, appnote.txt:2030-2166)
 * and writes      d_out + d_out_off[i] .. at most d_out_cap[i] bytes.
 * Results per entry: d_out_len (== PROP_TOTAL_OUT), d_in_used (== PROP_TOTAL_IN,
 * exact compressed bytes consumed, mz_zip.c:2090,2116), d_crc (CRC-32 of the
 * output, what mz_zip.c:2122 compares with the central directory), d_status.
 * Any stream length the 32-bit d_in_len[] can describe (the reference streams any size).
 * Asynchronous on `stream`. */
/* Decode that can be taken up again: the reference streams an entry of any size through a 32 767-byte buffer
 * (mz_strm_zlib.c:51,116-193); the batch kernel decodes a stream into one buffer.  Between the two: a stream is decoded
 * WINDOW BY WINDOW.  A state names a token boundary of the stream -- the bit position of the header of the block it lies
 * in (the block's Huffman tables are rebuilt from there) and of the next token -- and how many bytes of history sit in
 * front of the output.  When the output buffer is full (MZHIP_STATUS_OUT_FULL: the next token does not fit) or the input
 * ends (MZHIP_STATUS_BUF_ERROR) the kernel reports such a state; the caller keeps the last 32 KiB of what it was given
 * as history, drops the input in front of the block header (and rebases the two bit positions) and calls again. */
typedef struct mzhip_inflate_state {
    uint32_t hdr_bit; /* bit position of the current block's header, from the first input byte of the call */
    uint32_t bit;     /* bit position of the next token (== hdr_bit: at the start of the block) */
    uint32_t out_pos; /* in: bytes of history at the front of the output buffer (<= 32768); out: bytes valid in it */
    uint32_t flags;   /* bit 0 in: take the stream up at (hdr_bit, bit); out: the state is usable.  bit 1 in: stop in front of
                       * the next block header (MZHIP_STATUS_OUT_FULL with bit == hdr_bit) */
} mzhip_inflate_state;
/* mzhip_inflate_batch with one state in / one state out per entry (device pointers, either may be NULL) */
MZHIP_API int32_t mzhip_inflate_resume_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                             const uint64_t *d_out_off, const uint32_t *d_out_cap, uint32_t n,
                                             uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status,
                                             const mzhip_inflate_state *d_resume, mzhip_inflate_state *d_stop, void *stream);
/* ---- ONE raw-DEFLATE entry, or one window of it, through host buffers (H2D + kernel + D2H, synchronous): what the
 * mz_stream_zlib READ shim calls.  One entry point with an args struct: `size` = sizeof(mzhip_inflate_host_args) as the
 * caller was compiled, so fields can be appended without a new symbol.  (Round 5 folded mzhip_inflate_host, _host2,
 * _resume_host, _resume_host_seg and _resume_host_seg2 into this one; since round 6 it is NAMED for what it takes -- the
 * positional mzhip_inflate_host / mzhip_deflate_host of rounds 1 - 4 are exported by no build, so a binary compiled against
 * the old header fails to link instead of handing compressed bytes over as a struct.  A `size` that no build of the header
 * can have produced -- below the first fields, above 4096, not a multiple of 4 -- is MZ_PARAM_ERROR.)
 *   whole entry   state_in = state_out = NULL: in[0 .. in_len) is the stream, buf[0 .. buf_cap) takes its bytes;
 *   one window    state_out != NULL (state_in = NULL or flags bit 0 clear: the start of the stream): buf[0 ..
 *                 state_in->out_pos) is the history the caller kept (the last 32 KiB it was given), the new bytes land
 *                 behind it, buf_cap bytes in all; the verdict is MZHIP_STATUS_OK (stream end), _OUT_FULL (call again with
 *                 state_out and the tail of buf as history), _BUF_ERROR (more input from state_out's block header on) or a
 *                 data error.
 * Results (any pointer may be NULL): *out_len = bytes valid in buf (history included), *in_used, *crc / *adler = CRC-32 /
 * Adler-32 of the NEW bytes; and the CRC-32 of the new bytes in the pieces a caller hands on to a checksum
 * (mz_zip_entry_read -> mz_crypt_crc32_update, 65 535 bytes per call, mz_zip_rw.c:55): the first seg_first new bytes (what
 * completes the piece the previous window left open; 0 = none), then seg_stride at a time, then the rest -- seg_crc[0 ..
 * *nseg) from the device's copy of the window; *nseg = 0 when seg_cap is too small (the window itself is still valid).
 * Returns the device verdict (or an MZ_* error of the runtime). */
typedef struct mzhip_inflate_host_args {
    uint32_t size;
    uint32_t in_len, buf_cap;
    uint32_t seg_first, seg_stride, seg_cap;
    const uint8_t *in;
    uint8_t *buf;
    const mzhip_inflate_state *state_in;
    mzhip_inflate_state *state_out;
    uint32_t *out_len, *in_used, *crc, *adler, *seg_crc, *nseg;
} mzhip_inflate_host_args;
MZHIP_API int32_t mzhip_inflate_host_a(const mzhip_inflate_host_args *a);

/* (diagnostics, tests) The first step of mzhip_inflate_parallel_host alone: every bit offset of [b0, b1) of in[] at which a
 * dynamic-Huffman or stored block header could start (BTYPE, HLIT / HDIST <= 29 and a complete code-length code; LEN = ~NLEN),
 * in no order; *n = how many there are, out[] holds min(*n, cap).  which = 0: the kernel the product runs (32 offsets per lane),
 * 1: one offset per lane, the statement of the test (mz_block_header_plausible), 2: the product's kernel and the second step
 * behind it (k_check_headers: a lane reads each candidate's header to its end and keeps it only if a decoder would get past it).
 * misalign (0 - 15): where the bytes are put relative to a 16-byte boundary on the device. */
MZHIP_API int32_t mzhip_find_blocks_host(const uint8_t *in, uint32_t in_len, uint32_t b0, uint32_t b1, int32_t which, uint32_t misalign,
                                         uint32_t *out, uint32_t cap, uint32_t *n);

/* One window of ONE large entry decoded by as many waves as it holds blocks (replaces the single inflate() state the
 * reference streams an entry of any size through, mz_strm_zlib.c:116-193, where that state is the bottleneck): block
 * headers are searched for at every bit offset, every candidate is parsed by a wave of its own, the chain of blocks that
 * starts at state_in's header is believed, its bytes are produced as a source map and resolved by pointer jumping
 * (csrc/inflate_parallel.inc).  Same buffers as a window of mzhip_inflate_host_a; state_in must stand at a block header
 * (bit == hdr_bit; NULL = the start of the stream).  Returns 0 with *blocks = blocks decoded (0: nothing a wave of its own
 * could take -- go on with mzhip_inflate_host_a, flags bit 1 makes it stop at the next block header), *out_len =
 * bytes valid in buf, *ended = the final block was among them, state_out = the header of the first block not decoded;
 * *crc / *adler (either may be NULL) = CRC-32 / Adler-32 of the new bytes (what the gzip / zlib trailers run over). */
MZHIP_API int32_t mzhip_inflate_parallel_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                              const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out,
                                              uint32_t *out_len, uint32_t *blocks, uint32_t *ended, uint32_t *crc, uint32_t *adler,
                                              uint32_t seg_first, uint32_t seg_stride, uint32_t *seg_crc, uint32_t seg_cap,
                                              uint32_t *nseg);

/* ONE large raw-DEFLATE entry, device-resident, decoded by a wave per DEFLATE block window after window (the batch
 * kernel gives an entry one wave: 0.1 - 0.2 GB/s; this: several GB/s): d_in / d_out are device pointers, the four results
 * host pointers (any may be NULL), status as mzhip_inflate_batch's per-entry status.  Synchronous on `stream`.  What
 * mzhip_prime_* routes entries of 4 MiB and more of compressed bytes through. */
MZHIP_API int32_t mzhip_inflate_large(const void *d_in, uint32_t in_len, void *d_out, uint32_t out_cap, uint32_t *out_len,
                                      uint32_t *in_used, uint32_t *crc, int32_t *status, void *stream);

MZHIP_API int32_t mzhip_inflate_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                      void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap, uint32_t n,
                                      uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status,
                                      void *stream);

/* K2 alone: CRC-32 of n buffers (STORE entries, mz_zip.c:2049 with the raw
 * stream).  d_init may be NULL (all zero) or hold the chaining value of each
 * buffer (mz_crypt_crc32_update's `value`). */
MZHIP_API int32_t mzhip_crc32_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n,
                                    const uint32_t *d_init, uint32_t *d_crc, void *stream);

/* K5: Adler-32 of n buffers -- the zlib-wrapper trailer (RFC 1950), needed when mz_stream_zlib is opened with a
 * positive COMPRESS_WINDOW (mz_strm_zlib.c:80,104,348-350; minigzip.c:80 uses 15+16).  d_adler[i] = adler32(1, buf_i). */
MZHIP_API int32_t mzhip_adler32_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n,
                                      uint32_t *d_adler, void *stream);

/* K3: raw-LZMA1 range decode with fused CRC-32 ---------------------------- */

/* Replaces, for n method-14 entries at once, mz_stream_lzma_read (mz_strm_lzma.c:147-241 ->
 * liblzma lzma_alone_decoder/lzma_code) + mz_crypt_crc32_update (mz_zip.c:2049).
 * Entry i's input starts at the ZIP-LZMA header (2 B version, 2 B props size, 5 B
 * lc/lp/pb + dictionary size; appnote.txt:2232-2275) -- exactly the entry payload as
 * it sits in the archive -- and ends at the end-of-stream marker.  d_max_out (may be
 * NULL) carries PROP_TOTAL_OUT_MAX per entry (mz_zip.c:1845; <0 = none): d_out_len and
 * d_crc are clamped to it like mz_strm_lzma.c:214-215.  d_in_used counts the 9 header
 * bytes (ZIP accounting, mz_strm_lzma.c:124,198).  Status: 0, -3 data error, -5 input
 * ended early (mz_stream_lzma_read reports both as MZ_DATA_ERROR), -200 out_cap hit,
 * lc + lp up to 4, as liblzma (the upper half of an lc + lp = 4 literal model lives in a per-wave HBM scratch). */
MZHIP_API int32_t mzhip_lzma_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                   const uint64_t *d_out_off, const uint32_t *d_out_cap, const int64_t *d_max_out,
                                   uint32_t n, uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc,
                                   int32_t *d_status, void *stream);

/* .xz decode (ZIP method 95) with fused CRC-32 ------------------------------------------ */

/* Replaces, for n method-95 entries at once, mz_stream_lzma_read with lzma_stream_decoder(flags 0)
 * (mz_strm_lzma.c:127-128,147-241) + mz_crypt_crc32_update (mz_zip.c:2049).  Entry i's input is one .xz stream
 * (stream header, blocks of LZMA2 chunks, index, footer); block checks none / CRC32 / CRC64 / SHA-256 are verified
 * on the device.  Same argument and status conventions as mzhip_lzma_batch; d_in_used = bytes through the stream
 * footer.  Filter chains as liblzma 5.2.5 accepts them: LZMA2 last, up to three of Delta / BCJ (x86, PowerPC, IA-64, ARM,
 * ARM-Thumb, SPARC) in front; any other chain is a data error (-3) like LZMA_OPTIONS_ERROR behind mz_stream_lzma_read. */
MZHIP_API int32_t mzhip_xz_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                 const uint64_t *d_out_off, const uint32_t *d_out_cap, const int64_t *d_max_out,
                                 uint32_t n, uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc,
                                 int32_t *d_status, void *stream);

/* SHA-1 / SHA-224 / SHA-256 / SHA-384 / SHA-512 of n buffers ----------------------------- */

/* What the reader's hash verification computes per entry on the CPU: mz_crypt_sha_begin/_update/_end over the
 * decoded bytes (mz_zip_rw.c:409-451,465-466; mz_crypt.h:29-35).  algorithm = MZ_HASH_SHA1 (20), SHA224 (22), SHA256
 * (23), SHA384 (24) or SHA512 (25) (mz.h:127-135).  d_digest receives n x 32 bytes (n x 64 for SHA-384 / SHA-512): the
 * digest in its standard byte order followed by zero bytes.  One lane per buffer. */
MZHIP_API int32_t mzhip_sha_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n,
                                  uint32_t algorithm, void *d_digest, void *stream);

/* K4: raw-DEFLATE encode (dynamic / fixed / stored blocks, whichever is cheapest) with fused CRC-32 of the input -- */

/* Replaces, for n pieces at once, mz_stream_zlib_write/_close (mz_strm_zlib.c:203-264,280-305 -> zlib
 * deflate(), raw, level 1) + mz_crypt_crc32_update (mz_zip.c:2064).  Piece i compresses
 * d_in + d_in_off[i] .. + d_in_len[i] into d_out + d_out_off[i] (d_out_cap[i] >= len + len/8 + 64 always
 * suffices).  d_final (may be NULL = all 1): 1 -> the piece is a complete raw-DEFLATE stream (a ZIP entry);
 * 0 -> a non-final block closed by an empty stored block, so pieces of one stream concatenate on byte
 * boundaries.  The bytes are valid DEFLATE (appnote.txt:2030-2166) but not zlib's bytes: compressor output
 * is not a format property -- parity is "reference inflate(output) == input and CRC equal".
 * d_crc = CRC-32 of the INPUT piece. */
MZHIP_API int32_t mzhip_deflate_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                      void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                      const uint8_t *d_final, uint32_t n, uint32_t *d_out_len, uint32_t *d_crc,
                                      int32_t *d_status, void *stream);
/* the same with the compression level the stream was given (mz_stream_zlib_set_prop_int64 COMPRESS_LEVEL ->
 * deflateInit2(level, ...), mz_strm_zlib.c:87,339-343).  Three classes: 0..3 = fast (one candidate per hash bucket: the
 * ratio of zlib levels 1-2); 4..6 and -1 (the default) = four candidates per bucket, matches handed on to the next
 * positions, and a two-position lazy rule (between zlib levels 3 and 6; 3-4x the work); 7..9 = the same candidates and a
 * cost parse over every 64 KiB block (a backward dynamic programme priced with the block's own code lengths: within 3 %
 * of zlib-9's output, 3x the time of level 6 -- as in zlib, the top levels pay for ratio).  window_log2 = 9..15: matches reach at most 2^window_log2 - 262
 * bytes back (zlib's MAX_DIST), so an inflater with that window decodes the stream.  mzhip_deflate_batch == level 1, 15. */
MZHIP_API int32_t mzhip_deflate_batch_level(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                      void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap,
                                      const uint8_t *d_final, uint32_t n, int32_t level, int32_t window_log2, uint32_t *d_out_len, uint32_t *d_crc,
                                      int32_t *d_status, void *stream);

/* LZMA1 encode (ZIP method 14 payloads, LZMA2 chunk payloads) ---------------------------- */

/* Replaces, for n streams at once, mz_stream_lzma_write/_close (mz_strm_lzma.c:244-332 -> liblzma
 * lzma_alone_encoder) + mz_crypt_crc32_update (mz_zip.c:2064).  Two kernels: the LZ77 parse (one wave per 64 KiB
 * block of any stream) and the adaptive range coder (one wave per stream -- it is strictly serial).  d_mode (may be
 * NULL = all 0): 0 -> a complete ZIP method-14 payload (4-byte magic, lc3/lp0/pb2 + 64 KiB dictionary, data, end
 * marker: what mz_zip.c expects with MZ_ZIP_FLAG_LZMA_EOS_MARKER); 1 -> the raw payload of one LZMA2 chunk.
 * max_in_len = upper bound of d_in_len[] (sizes the token scratch: 4 bytes per input position, stream-ordered
 * allocation).  d_out_cap[i] >= len + len/8 + 1024 always suffices.  The bytes are valid LZMA but not liblzma's:
 * parity is "the reference's mz_stream_lzma_read returns the input".  d_crc = CRC-32 of the INPUT. */
MZHIP_API int32_t mzhip_lzma_encode_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                          uint32_t max_in_len, void *d_out, const uint64_t *d_out_off,
                                          const uint32_t *d_out_cap, const uint8_t *d_mode, uint32_t n,
                                          uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status, void *stream);
/* ... with the preset the reference hands to lzma_lzma_preset (mz_strm_lzma.c:81; COMPRESS_LEVEL, -1 = the default 6):
 * presets 0-3 try one hash candidate per position (what the call above does), 4-9 and the default try four and apply
 * a two-position lazy rule -- smaller output, a slower parse */
MZHIP_API int32_t mzhip_lzma_encode_batch_preset(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len,
                                                 uint32_t max_in_len, void *d_out, const uint64_t *d_out_off,
                                                 const uint32_t *d_out_cap, const uint8_t *d_mode, uint32_t n, int32_t preset,
                                                 uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status, void *stream);

/* Host-buffer conveniences (H2D + kernel + D2H, synchronous); these are what the
 * vtbl shims use for one-entry-at-a-time callers (raw DEFLATE: mzhip_inflate_host_a / mzhip_deflate_host_a above / below). */
MZHIP_API int32_t mzhip_lzma_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                                  uint32_t *out_len, uint32_t *in_used, uint32_t *crc);
MZHIP_API int32_t mzhip_xz_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                                uint32_t *out_len, uint32_t *in_used, uint32_t *crc);
/* a whole entry: ZIP method-14 payload / one .xz stream (a single block of 64 KiB LZMA2 chunks that reset state and
 * properties and keep the dictionary: matches reach back 8 MiB as in the method-14 stream; CRC32 check); *crc = CRC-32 of
 * `in` */
MZHIP_API int32_t mzhip_lzma_encode_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap,
                                         uint32_t *out_len, uint32_t *crc);
MZHIP_API int32_t mzhip_xz_encode_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap,
                                       uint32_t *out_len, uint32_t *crc);
/* Resumable LZMA1 decode (ZIP method 14) -- entries decoded window by window in bounded memory, as the reference streams
 * any entry through 32 767 bytes (mz_strm_lzma.c:147-241).  `state` is sixteen words: flags (in: bit 0 take the stream up
 * from this state, else a fresh stream whose header comes first; bit 1 the input given is the stream's last.  out: 1 = the
 * decoder stopped in front of a packet and can go on from here), the range coder, the four repeat distances, the
 * properties, out_pos (in: bytes of dictionary in front of the room in buf; out: bytes valid in buf) and in_pos (out: bytes
 * of `in` that are done with).  `model` (mzhip_lzma_model_bytes() bytes, the caller's) carries the adaptive model from call
 * to call.  With a state_out the decoder stops when fewer than 274 bytes of room or (bit 1 clear) 64 bytes of input are
 * left: MZHIP_STATUS_OUT_FULL / MZHIP_STATUS_BUF_ERROR with flags = 1.  Between calls the caller keeps at least
 * min(everything produced, dictionary size) bytes in front of the buffer and drops a multiple of 16.  No CRC is computed. */
typedef struct mzhip_lzma_state {
    uint32_t flags, range, code, state, rep0, rep1, rep2, rep3, props, dict, out_pos, in_pos, pad[4];
} mzhip_lzma_state;
MZHIP_API uint32_t mzhip_lzma_model_bytes(void);
/* A ZIP method-14 stream written segment by segment in bounded memory (mz_strm_lzma.c:244-332 stages any entry through
 * 32 767 bytes).  in = [skip_blocks x 64 KiB of the stream's previous bytes | the segment]: the bytes in front are match
 * sources and literal contexts, they are not coded again.  Every segment but the last is a multiple of 64 KiB and leaves
 * the coder in state_out (sixteen words: flags -- in: bit 0 go on from state_in, else a fresh stream whose header comes
 * first --, the range coder with its held-back byte, the packet state, the four repeat distances) and the adaptive model in
 * `model` (mzhip_lzma_model_bytes(), the caller's); last != 0 writes the end marker and flushes the coder.  The bytes of
 * the segments, in order, are the payload.  No CRC is computed.  A caller that keeps
 * min(everything coded so far, mzhip_lzma_encode_history_bytes()) bytes in front of every segment gets the bytes the
 * one-shot coder makes of the whole stream: the encoder's matches reach back that far (8 MiB, liblzma's preset 6:
 * mz_strm_lzma.c:81) and no further. */
MZHIP_API uint32_t mzhip_lzma_encode_history_bytes(void);
/* A ZIP method-95 payload (.xz) written block by block in bounded memory: every call codes one block of independent
 * LZMA2 chunks -- behind the stream header when first != 0 -- and reports the block's unpadded size; when the entry is
 * complete mzhip_xz_encode_finish_host writes the index over all blocks and the stream footer.  liblzma's
 * lzma_stream_decoder (mz_strm_lzma.c:127-128) reads a stream of any number of blocks. */
MZHIP_API int32_t mzhip_xz_encode_block_host(const uint8_t *in, uint32_t in_len, int32_t preset, int32_t first, uint8_t *out,
                                             uint32_t out_cap, uint32_t *out_len, uint32_t *crc, uint64_t *unpadded_size);
MZHIP_API int32_t mzhip_xz_encode_finish_host(const uint64_t *unpadded_size, const uint64_t *uncompressed_size, uint32_t nblocks,
                                              uint8_t *out, uint32_t out_cap, uint32_t *out_len);
typedef struct mzhip_lzma_enc_state {
    uint32_t flags, low_lo, low_hi, range, cache, cache_size, state, rep0, rep1, rep2, rep3, pad[5];
} mzhip_lzma_enc_state;
MZHIP_API int32_t mzhip_lzma_encode_resume_host(const uint8_t *in, uint32_t in_len, uint32_t skip_blocks, uint32_t last,
                                                int32_t preset, const mzhip_lzma_enc_state *state_in,
                                                mzhip_lzma_enc_state *state_out, void *model, uint8_t *out, uint32_t out_cap,
                                                uint32_t *out_len);
MZHIP_API int32_t mzhip_lzma_resume_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                         const mzhip_lzma_state *state_in, mzhip_lzma_state *state_out, void *model,
                                         uint32_t *out_len, uint32_t *in_used);

/* Method 95 (.xz) in windows -- entries of any size in bounded memory, as the reference streams them through 32 767 bytes
 * (mz_strm_lzma.c:127-128,147-241).  The caller walks the container (stream header, block headers, padding, check fields,
 * index, footer); this call decodes ONE block's LZMA2 chunk sequence a window at a time: `in` starts at a chunk header or
 * inside the chunk the call before stopped in, buf[0 .. state_in->out_pos) is the dictionary so far and the room behind it
 * receives the window.  `state` is twenty words: flags (in: 1 the model is in `model`, else a fresh block whose first chunk
 * resets everything; 2 the input given is the entry's last; 4 / 8 the next chunk must bring properties / reset the
 * dictionary -- a fresh block is 4 | 8 --; 32 / 64 inside an uncompressed / LZMA chunk.  out: the same bits, 1 = the
 * decoder stopped where it can go on from, 16 = the block's end byte has been consumed, 128 = a data error was met between
 * chunks rather than inside a packet), the range coder, LZMA state and
 * repeat distances, the properties, the block header's dictionary size, out_pos / in_pos as in mzhip_lzma_state, where the
 * dictionary begins in buf (the caller subtracts what it drops, stopping at 0), what is left of a chunk in progress, and
 * the block's check (id 0 none / 1 CRC-32 / 4 CRC-64; value of the block's bytes so far, 0 at the start of a block).  The
 * decoder stops in front of a chunk, or of a packet, when fewer than 274 bytes of room or (bit 2 clear) fewer than 64 bytes
 * of the chunk's input are left: MZHIP_STATUS_OUT_FULL / MZHIP_STATUS_BUF_ERROR with flags bit 0.  Between calls the caller
 * keeps at least min(everything the stream produced, dictionary size) bytes in front of buf.  Returns the status. */
typedef struct mzhip_lzma2_state {
    uint32_t flags, range, code, state, rep0, rep1, rep2, rep3, props, dict, out_pos, in_pos;
    uint32_t dict_start, chunk_usize_left, chunk_csize_left, check_id, check_lo, check_hi, pad[2];
} mzhip_lzma2_state;
typedef struct mzhip_lzma2_run_args {
    uint32_t size; /* sizeof(mzhip_lzma2_run_args): fields added later are taken as 0 / NULL */
    uint32_t in_len;
    uint32_t buf_cap;
    uint32_t reserved;
    const uint8_t *in;
    uint8_t *buf;
    const mzhip_lzma2_state *state_in;
    mzhip_lzma2_state *state_out;
    void *model;       /* mzhip_lzma_model_bytes(), the caller's */
    uint32_t *out_len; /* bytes valid in buf */
    uint32_t *in_used; /* bytes of `in` that are done with */
} mzhip_lzma2_run_args;
MZHIP_API int32_t mzhip_lzma2_run_host(const mzhip_lzma2_run_args *args);

/* ... at a preset (see mzhip_lzma_encode_batch_preset); what mz_stream_lzma_write / _close use */
MZHIP_API int32_t mzhip_lzma_encode_host_preset(const uint8_t *in, uint32_t in_len, int32_t preset, uint8_t *out,
                                                uint32_t out_cap, uint32_t *out_len, uint32_t *crc);
MZHIP_API int32_t mzhip_xz_encode_host_preset(const uint8_t *in, uint32_t in_len, int32_t preset, uint8_t *out,
                                              uint32_t out_cap, uint32_t *out_len, uint32_t *crc);
/* ONE segment of a raw-DEFLATE stream through host buffers (what mz_stream_zlib_write / _close call): 64 KiB pieces, one
 * wave each, every piece but the last closed on a byte boundary, the last one final iff `final`; `level` / `window_log2` as
 * mzhip_deflate_batch_level (COMPRESS_LEVEL, mz_strm_zlib.c:87,339-343: levels 0 .. 3 are one class; window_log2 0 = 15); *crc / *adler (either
 * may be NULL) = CRC-32 / Adler-32 of `in`.  `size` = sizeof(mzhip_deflate_host_args) as the caller was compiled.  (Round 5
 * folded mzhip_deflate_host_a, _host2 and _host_level into this one.) */
typedef struct mzhip_deflate_host_args {
    uint32_t size;
    uint32_t in_len, final, out_cap;
    int32_t level, window_log2;
    const uint8_t *in;
    uint8_t *out;
    uint32_t *out_len, *crc, *adler;
} mzhip_deflate_host_args;
MZHIP_API int32_t mzhip_deflate_host_a(const mzhip_deflate_host_args *a);
/* mz_crypt_crc32_update on a host buffer (mz_crypt.c:35-92: chaining value in, chaining value out).  Buffers of
 * MZHIP_CRC_HOST_BELOW bytes or more are reduced on the device (K2); smaller ones -- the reference calls the symbol
 * per byte from mz_strm_pkcrypt.c:79,86 -- are folded on the host with the same generated tables.  Never aborts: if
 * the device is unusable the value is still exact and the failure is reported by the next codec-stream call. */
#define MZHIP_CRC_HOST_BELOW 4096u
MZHIP_API uint32_t mzhip_crc32_host(uint32_t value, const uint8_t *buf, size_t size);
/* Device failures met under mz_crypt_crc32_update since the process started.  The symbol has no error channel (mz_crypt.h:20), so
 * the checksum of such a call is folded on the host by this library's own tables; the failure is reported by mzhip_last_error()
 * on the calling thread, by the thread's next codec-stream call (MZ_STREAM_ERROR), once on stderr -- and counted here. */
MZHIP_API uint64_t mzhip_crc_faults(void);

/* Archive index (host, C) --------------------------------------------------------------- */

/* One pass over a memory image of a ZIP archive -> flat entry table, 8 x int64 per entry:
 * method, flag, crc, compressed size, uncompressed size, local-header offset, central-directory
 * position, payload offset (-1 if the local header is unusable).  Same facts the reference yields one
 * entry at a time (mz_zip.c:947-1100, :202-479, :2402-2412); SURVEY 8(f) row 1.  Returns the entry count
 * (which may exceed max_entries: call again with a larger table) or MZ_FORMAT_ERROR (-103). */
MZHIP_API int64_t mzhip_zip_index_mem(const uint8_t *zip, uint64_t zip_len, int64_t *table, int64_t max_entries);
/* The Hash extra field (0x1a51) of every entry of such a table: the first one of its central-directory record (what
 * mz_zip_reader_entry_get_first_hash picks, mz_zip_rw.c:510-540).  algorithm[i] = MZ_HASH_* or 0 (none), digest_size[i],
 * digest + 64 * i.  Returns the number of entries that carry one. */
MZHIP_API int64_t mzhip_zip_index_hash_mem(const uint8_t *zip, uint64_t zip_len, const int64_t *table, int64_t n,
                                           uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest);
/* The same two from the archive's TAIL alone (an archive too large to image: shim_autoprime.c rolls over it window by window):
 * tail[0] = byte tail_off of a file of zip_len bytes; the tail must hold the end records and the whole central directory.
 * MZHIP_INDEX_NEED_MORE = it does not yet: read the tail from *need_from and call again.  Rows come back with payload offset
 * -1 (t[7]): the local headers lie in the body.  mzhip_zip_index_resolve fills in the rows whose local header and payload lie
 * inside a window win[0 .. win_len) = bytes [win_off, win_off + win_len) of the file; returns the number resolved. */
#define MZHIP_INDEX_NEED_MORE (-1000)
MZHIP_API int64_t mzhip_zip_index_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, int64_t *table, int64_t max_entries,
                                       uint64_t *need_from);
MZHIP_API int64_t mzhip_zip_index_hash_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, const int64_t *table, int64_t n,
                                            uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest);
MZHIP_API int64_t mzhip_zip_index_resolve(const uint8_t *win, uint64_t win_off, uint64_t win_len, int64_t *table, int64_t n);

/* Prime (SURVEY 8b "Batching") ------------------------------------------------------------- */

/* Decode every DEFLATE / LZMA / XZ entry (methods 8, 14, 95) of an archive (file or memory image) in one launch
 * per codec and keep the results in a host cache.  Afterwards the drop-in mz_stream_zlib / mz_stream_lzma READ
 * path recognises a primed entry (base-stream position
 * == payload offset, first payload bytes equal) and serves read() calls from the cache; the matching
 * mz_crypt_crc32_update calls are answered from GPU-computed per-65 535-byte-segment CRCs, so the reference's
 * untouched mz_zip_reader loop runs at memcpy speed while mz_zip.c:2116-2128 still verifies every entry against
 * the central directory.  Entries that did not decode cleanly are not cached (they take the ordinary path and
 * its exact error behaviour).  Returns the number of cached entries or a negative MZ_* code. */
MZHIP_API int64_t mzhip_prime_file(const char *path);
MZHIP_API int64_t mzhip_prime_mem(const uint8_t *zip, uint64_t zip_len);
/* The same, in the background: returns as soon as the archive is indexed and its entries are known (their number, or
 * 0 / an MZ error), while a worker thread of the library runs the decode pipeline on the calling thread's device.  A
 * READ stream that opens an entry whose chunk of the pipeline has not arrived yet waits for that chunk only, so reader
 * threads that take the entries front to back run under the decode instead of behind it (integration/extract_threads.c).
 * The image must stay valid and unchanged until mzhip_prime_wait() -- which joins every prime begun so far and returns
 * the number of entries they primed (or the first error) -- or mzhip_prime_clear(), which waits too. */
MZHIP_API int64_t mzhip_prime_mem_begin(const uint8_t *zip, uint64_t zip_len);
MZHIP_API int64_t mzhip_prime_wait(void);
/* archives the READ streams primed on their own (shim_autoprime.c: on by default, MZHIP_AUTOPRIME=0 turns it off) */
MZHIP_API uint64_t mzhip_autoprime_count(void);
/* archives larger than the limit are rolled over window by window: windows primed / evicted so far, page-locked bytes the live
 * windows hold now and held at most */
MZHIP_API void mzhip_autoprime_stats(uint64_t *windows_primed, uint64_t *windows_evicted, uint64_t *live_bytes, uint64_t *peak_bytes);
/* The same over several devices of the node (the host side of the sharded path in C): the entries are
 * independent (mz_zip.c:1682-1863 builds a fresh codec per entry), so the entry table is cut into ndev contiguous
 * slices balanced by compressed + uncompressed bytes (mzhip_shard_bounds) and ONE HOST THREAD PER SLICE decodes it on
 * device devices[i]: only the byte range of the archive that holds the slice's payloads goes to that device, nothing
 * crosses between devices, the per-entry results are merged on the host into one cache generation.  devices == NULL:
 * devices 0 .. ndev-1; ndev <= 0: every visible device.  A device may be listed more than once. */
MZHIP_API int64_t mzhip_prime_file_multi(const char *path, const int32_t *devices, int32_t ndev);
MZHIP_API int64_t mzhip_prime_mem_multi(const uint8_t *zip, uint64_t zip_len, const int32_t *devices, int32_t ndev);
/* bounds[0 .. world]: slice r = entries [bounds[r], bounds[r+1]) of an mzhip_zip_index_mem table (8 x int64 per entry) */
MZHIP_API void mzhip_shard_bounds(const int64_t *table, int64_t n, int32_t world, int64_t *bounds);
/* The per-archive result gather of a sharded decode (north_star: "RCCL over xGMI only for the final per-archive CRC gather").
 * Rank r decoded entries [bounds[r], bounds[r + 1]) of the table (mzhip_shard_bounds); d_crc / d_status = this rank's slice
 * (device memory, bounds[rank + 1] - bounds[rank] words each).  Afterwards d_crc_all / d_status_all (bounds[world] words each,
 * device memory) hold every entry's pair on every rank: ONE ncclAllGather of max-slice-sized {crc, status} blocks + a copy per
 * rank, queued on `stream`.  comm = the job's ncclComm_t (RCCL is opened with dlopen() on first use: libmzhip.so does not link
 * it); world == 1: comm may be NULL, the call is two copies.  Returns 0 or an MZ_* code (MZ_SUPPORT_ERROR: no librccl.so). */
MZHIP_API int32_t mzhip_gather_results(void *comm, int32_t rank, int32_t world, const int64_t *bounds, const uint32_t *d_crc,
                                       const int32_t *d_status, uint32_t *d_crc_all, int32_t *d_status_all, void *stream);
MZHIP_API void mzhip_prime_clear(void);

/* Memory bound of the drop-in READ streams (shim_zlib.c window mode; the reference stages any entry through 32 767 bytes,
 * mz_strm_zlib.c:51,116-193): an entry whose decoded size passes `window_bytes` is decoded window by window with
 * `gulp_bytes` of compressed input pulled ahead of each launch.  Defaults 64 MiB / 16 MiB (or MZHIP_STREAM_WINDOW /
 * MZHIP_STREAM_GULP in the environment); floors 128 KiB / 32 KiB; 0 = back to the default.  Applies to streams opened
 * afterwards. */
MZHIP_API void mzhip_set_stream_window(int64_t window_bytes, int64_t gulp_bytes);
/* bytes of an entry that mz_stream_lzma WRITE codes per launch once the entry is larger than that (whole 64 KiB blocks, at
 * least two, at most the default: 8 MiB; 0 = the default, or MZHIP_WRITE_SEGMENT in the environment).  Independent of the READ
 * window above: what a written stream looks like never depends on a read-side setting. */
MZHIP_API void mzhip_set_write_segment(int64_t segment_bytes);
/* Window mode of mz_stream_zlib READ offers every window that starts at a block header to mzhip_inflate_parallel_host first
 * (a wave per DEFLATE block); 0 turns that off (MZHIP_STREAM_PARALLEL=0 in the environment does the same). */
MZHIP_API void mzhip_set_stream_parallel(int32_t on);
/* ... and while the caller is served one such window, a thread of the stream's own has the device decode the next one (the two
 * buffers swap when the caller has used the first up); 0 turns that off (MZHIP_STREAM_LOOKAHEAD=0 does the same): the device
 * call and the serving then take turns.  What the caller sees -- bytes, return values, totals, errors -- is the same either way.
 * mzhip_stream_lookahead_windows(): windows taken over from such a thread so far, all streams of the process. */
MZHIP_API void mzhip_set_stream_lookahead(int32_t on);
/* mz_stream_zlib WRITE collects 8 MiB per device launch; a full segment is coded by a thread of the stream while the caller
 * fills the next one (its bytes reach the base stream, from the caller's thread, when the segment after it is full or at
 * close).  0 turns that off (MZHIP_WRITE_OVERLAP=0 does the same): the segment is coded before write() returns.  The
 * stream that is written is the same either way. */
MZHIP_API void mzhip_set_write_overlap(int32_t on);
MZHIP_API uint64_t mzhip_stream_lookahead_windows(void);
/* page-locked host memory for a READ stream's window buffer, from the library's pool (next to the current device; NULL when
 * there is none to be had -- the caller then uses plain memory); *cap = what to hand back to mzhip_window_free */
MZHIP_API void *mzhip_window_alloc(size_t bytes, size_t *cap);
MZHIP_API void mzhip_window_free(void *p, size_t cap);
MZHIP_API void mzhip_prime_stats(uint64_t *entries, uint64_t *hits, uint64_t *misses);
/* Entries that carry a SHA-1 / SHA-256 Hash extra field (0x1a51): the prime computes the digest of the decoded bytes on
 * the device in the pass that decodes them (mzhip_sha_batch over the chunk in HBM) and compares it with the field's, as
 * mz_zip_reader_entry_close does on the host (mz_zip_rw.c:439-451).  An entry whose digest differs is not served from the
 * cache -- the ordinary path, and whoever verifies behind it, sees it.  Counters since the process started. */
MZHIP_API void mzhip_prime_hash_stats(uint64_t *checked, uint64_t *mismatched);
/* mz_crypt_sha_end calls that were answered with a device-computed digest (shim_sha.c) */
MZHIP_API uint64_t mzhip_sha_primed_digests(void);

/* Write-side prime (SURVEY 8b "Batching", BASELINE config 5) ------------------------------- */

/* Compress n buffers (blob + off[i], len[i]) in one launch per group and keep the streams in a host cache; method 8
 * (raw DEFLATE) or 14 (ZIP-LZMA payload).  Afterwards the drop-in mz_stream_zlib / mz_stream_lzma WRITE path follows
 * the bytes the reference's untouched writer loop hands it (mz_zip_writer_add_buffer -> mz_zip_entry_write,
 * mz_zip.c:2052-2068) against the primed buffers, chunk by chunk and byte for byte; an entry that is exactly one of
 * them is answered at close() with the cached stream, and the mz_crypt_crc32_update calls on the writer's
 * 65 535-byte chunks (mz_zip_rw.c:55) with device-computed segment CRCs.  An entry that diverges from a primed buffer
 * at any point takes the ordinary path.  Buffers shorter than 16 bytes or longer than 8 MiB are not cached.  The
 * caller keeps the primed buffers valid and unchanged until mzhip_prime_write_clear() (or the next prime of the same
 * method).  Returns the number of cached buffers or a negative MZ_* code. */
MZHIP_API int64_t mzhip_prime_write(int32_t method, const uint8_t *blob, const uint64_t *off, const uint32_t *len, uint32_t n);
MZHIP_API void mzhip_prime_write_clear(void);
MZHIP_API void mzhip_prime_write_stats(uint64_t *entries, uint64_t *hits, uint64_t *misses);

/* Geometry the last launch used (for reports): workgroups, waves per workgroup, LDS bytes per workgroup. */
MZHIP_API void mzhip_inflate_launch_geometry(uint32_t n, uint32_t *grid, uint32_t *waves_per_wg, uint32_t *lds_bytes);

#ifdef __cplusplus
}
#endif
#endif
