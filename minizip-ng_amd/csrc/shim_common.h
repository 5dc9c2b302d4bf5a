/* shim_common.h -- private glue between the C shims and the prime cache (not part of the ABI). */
#ifndef MZHIP_SHIM_COMMON_H
#define MZHIP_SHIM_COMMON_H
#include <stdint.h>

/* prime cache lookup (mzhip_kernels.hip): 1 = hit */
int32_t mzhip_prime_lookup(int64_t payload_off, const uint8_t *head, int32_t head_len, const uint8_t **data,
                           int64_t *usize, int64_t *csize, uint32_t *crc, const uint32_t **seg_crc);
/* the same for any primed method (8 DEFLATE, 14 LZMA, 95 XZ) */
int32_t mzhip_prime_lookup2(int32_t method, int64_t payload_off, const uint8_t *head, int32_t head_len,
                            const uint8_t **data, int64_t *usize, int64_t *csize, uint32_t *crc, const uint32_t **seg_crc);
/* MZHIP_AUTOPRIME: prime the archive behind a codec stream's base on first use (shim_autoprime.c); no-op otherwise */
struct mzhip_stream_s;
void mzhip_autoprime(struct mzhip_stream_s *codec_base);
/* write-side prime (mzhip_prime_write): follow the bytes a WRITE stream is handed against the primed buffers */
int32_t mzhip_wprime_track(int32_t method, int64_t *id, int64_t pos, const uint8_t *buf, int32_t size, uint32_t *chunk_crc,
                           int32_t *have_crc);
int32_t mzhip_wprime_result(int32_t method, int64_t id, int64_t pos, const uint8_t **src, const uint8_t **out,
                            uint32_t *out_len);
/* crc(A||B) from crc(A), crc(B), |B|: arithmetic on checksums, no data bytes involved */
uint32_t mzhip_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b);

/* adler(A||B) from adler(A), adler(B), |B| */
uint32_t mzhip_adler32_combine(uint32_t ad_a, uint32_t ad_b, uint64_t len_b);

/* The last buffer a primed stream handed to its caller, with the GPU-computed CRC-32 of exactly those bytes:
 * lets mz_crypt_crc32_update (called by mz_zip_entry_read right after the read, mz_zip.c:2047-2049) answer
 * without a second trip to the device. */
typedef struct mzhip_served_s {
    const void *buf;
    int32_t size;
    uint32_t crc;
    int32_t valid;
} mzhip_served;
extern __thread mzhip_served mzhip_last_served;

#define MZHIP_PRIME_SEGMENT 65535 /* the reader's buffer size, mz_zip_rw.c:55 */
#endif
