/* oracle.h -- CPU restatement of the minizip-ng codec/CRC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is linked, imported or
 * executed by the product path (minizip-ng_amd/); only tests/, bench.py's
 * cpu_baseline leg and __graft_entry__.smoke() may use it, and only as the
 * checker.
 *
 * Parity pinning: every function here is checked (tests/test_oracle.py)
 *   (i)  against the reference itself, compiled from /root/reference into
 *        oracle/_ref/libmzref.so (zlib 1.2.11 + liblzma 5.2.5 behind the
 *        unmodified mz_strm_zlib.c / mz_strm_lzma.c / mz_crypt.c), and
 *   (ii) against the golden (payload, size, crc) triples stored in the ZIP
 *        headers of the reference's own fixture archives
 *        (test/fuzz/unzip_fuzzer_seed_corpus, every .zip -> tests/golden/fixtures.json).
 *
 * The DEFLATE / LZMA arithmetic the reference calls lives in third-party
 * libraries that are NOT under /root/reference (zlib|zlib-ng and liblzma,
 * un-pinned by the reference's CMake; zlib 1.2.11 and liblzma 5.2.5 in this
 * image).  The restatements below follow the published formats:
 *   DEFLATE  : doc/zip/appnote.txt:2030-2166 (== RFC 1951)
 *   ZIP-LZMA : doc/zip/appnote.txt:2182-2275 + the public-domain LZMA spec
 *   CRC-32   : doc/zip/appnote.txt:837-847, mz_crypt.c:51-90
 * and the reference call sites mz_strm_zlib.c:116-193, mz_strm_lzma.c:147-241.
 */
#ifndef MZ_ORACLE_H
#define MZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* status codes: numerically the zlib / MZ_* codes (mz.h:20-26) */
#define ORC_OK          0
#define ORC_DATA_ERROR  (-3)
#define ORC_BUF_ERROR   (-5)   /* input exhausted before end of stream     */
#define ORC_OUT_FULL    (-200) /* output capacity reached (no MZ analogue) */

/* mz_crypt_crc32_update (mz_crypt.c:35-92): chaining CRC-32. */
uint32_t orc_crc32_update(uint32_t value, const uint8_t *buf, size_t size);
/* crc(A||B) from crc(A), crc(B), len(B)  (GF(2) shift by x^(8*len2)). */
uint32_t orc_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2);

/* Raw DEFLATE decode of one complete stream (what inflate() does for
 * mz_stream_zlib_read with window_bits = -15, mz_strm_zlib.c:97,158).
 * in_used = exact compressed bytes consumed (== PROP_TOTAL_IN). */
int32_t orc_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used,
                        size_t *out_len);

/* Raw LZMA1 decode as mz_stream_lzma_read drives it for method 14
 * (mz_strm_lzma.c:118-126,177-201): `in` starts at the ZIP-LZMA header
 * (2 B version, 2 B props size, 5 B props), uncompressed size unknown, so the
 * stream ends at the EOS marker or, if max_out >= 0, is clamped to max_out
 * bytes (PROP_TOTAL_OUT_MAX, :214-215).  in_used counts the 9 header bytes. */
int32_t orc_lzma_zip_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, int64_t max_out,
                            size_t *in_used, size_t *out_len);

/* .xz decode as mz_stream_lzma_read drives it for method 95 (mz_strm_lzma.c:127-128: lzma_stream_decoder,
 * flags 0): one stream, LZMA2 filter only, checks none/CRC32/CRC64/SHA-256 verified.  in_used = bytes through
 * the stream footer.  -109 = filter chain outside the backend's scope. */
int32_t orc_xz_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, int64_t max_out, size_t *in_used,
                      size_t *out_len);
uint64_t orc_crc64(const uint8_t *p, size_t n);                 /* ECMA-182, the .xz check id 4 */
void orc_sha256(const uint8_t *p, size_t n, uint8_t out[32]);  /* FIPS 180-4, the .xz check id 10 */

#ifdef __cplusplus
}
#endif
#endif
