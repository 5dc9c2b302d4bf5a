/* shim_common.h -- private glue between the C shims and the prime cache (not part of the ABI). */
#ifndef MZHIP_SHIM_COMMON_H
#define MZHIP_SHIM_COMMON_H
#include <stdint.h>

/* prime cache lookup (mzhip_kernels.hip): 1 = hit, for any primed method (8 DEFLATE, 14 LZMA, 95 XZ).  `head` = the
 * payload bytes pulled so far, `max_total_in` = the stream's TOTAL_IN_MAX (<= 0: unknown).  On a hit *pin keeps the
 * cached generation alive; hand it back with mzhip_prime_unpin() when the stream stops reading from `data`. */
int32_t mzhip_prime_lookup3(int32_t method, int64_t payload_off, const uint8_t *head, int32_t head_len, int64_t max_total_in,
                            const uint8_t **data, int64_t *usize, int64_t *csize, uint32_t *crc, const uint32_t **seg_crc,
                            void **pin, uint16_t *hash_alg, const uint8_t **hash_digest);
void mzhip_prime_unpin(void *pin);
/* are these bytes one 65 535-byte reader chunk of a primed STORE entry?  1 = yes (*crc = its device-computed CRC-32;
 * equality was checked byte for byte against the primed payload) */
int32_t mzhip_prime_store_crc(const uint8_t *buf, int32_t size, uint32_t *crc);
/* prime the archive behind a codec stream's base -- whole, or the window around the entry whose payload starts at payload_off --
 * on an entry's first read (shim_autoprime.c; MZHIP_AUTOPRIME=0 turns it off) */
struct mzhip_stream_s;
void mzhip_autoprime(struct mzhip_stream_s *codec_base, int64_t payload_off);
int64_t mzhip_lfh_csize_hint(struct mzhip_stream_s *codec_base, int64_t payload_off); /* (shim_autoprime.c) the entry's compressed size off its local header, or 0 */
/* write-side prime (mzhip_prime_write): follow the bytes a WRITE stream is handed against the primed buffers */
int32_t mzhip_wprime_track(int32_t method, int64_t *id, int64_t pos, const uint8_t *buf, int32_t size, uint32_t *chunk_crc,
                           int32_t *have_crc, const uint8_t **src);
int32_t mzhip_prime_any(void); /* is any archive primed at all? */
int32_t mzhip_wprime_result(int32_t method, int64_t id, int64_t pos, const uint8_t **src, const uint8_t **out,
                            uint32_t *out_len);
/* the device failure a mz_crypt_crc32_update of this thread could not report (the symbol has no error channel); 0 = none */
int32_t mzhip_take_crc_fault(void);
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
    uint32_t slot, epoch; /* the stream that set the hint and the epoch of its slot at that moment: a close / delete /
                        re-open of that stream since -- from any thread -- makes the hint stale, so src is never read after
                        its owner may have given it back.  Slots are shared by streams created 4096 apart: a
                        stranger's close costs the fast path once, never correctness */
    /* ... and for mz_crypt_sha_update (shim_sha.c), which the reader calls on the same buffer right behind the CRC symbol
     * (mz_zip_rw.c:462-467): the served bytes are bytes [ent_off, ent_off + size) of a primed entry of ent_usize bytes whose
     * digest (ent_alg, 32 bytes at ent_digest) the device computed and found equal to the entry's Hash field */
    int32_t valid_sha;
    const uint8_t *ent_base, *ent_digest;
    int64_t ent_off, ent_usize;
    uint16_t ent_alg;
    const void *src; /* the primed bytes that were served (or that a written buffer was found equal to): compared again, byte
                        for byte, when the checksum is asked for -- a caller that changed the buffer in between gets the CRC of
                        what the buffer holds now.  Valid while the hint is: the stream that set it pins the primed generation
                        and every shim call, close and delete included, drops the hint first */
} mzhip_served;
extern __thread mzhip_served mzhip_last_served;
#define MZHIP_STREAM_SLOTS 4096u
extern uint32_t mzhip_stream_epoch[MZHIP_STREAM_SLOTS]; /* bumped (atomically) whenever the slot's stream gives buffers back */
extern uint32_t mzhip_stream_slot_next;
static inline uint32_t mzhip_stream_slot_new(void) { return __atomic_fetch_add(&mzhip_stream_slot_next, 1u, __ATOMIC_RELAXED) % MZHIP_STREAM_SLOTS; }
/* record / drop the hint; every shim read, write and open drops it first, so it only ever describes the bytes the
 * immediately preceding codec call produced */
static inline void mzhip_served_set(const void *buf, int32_t size, uint32_t crc, const void *src, uint32_t slot) {
    mzhip_last_served.buf = buf;
    mzhip_last_served.size = size;
    mzhip_last_served.crc = crc;
    mzhip_last_served.src = src;
    mzhip_last_served.slot = slot % MZHIP_STREAM_SLOTS;
    mzhip_last_served.epoch = __atomic_load_n(&mzhip_stream_epoch[slot % MZHIP_STREAM_SLOTS], __ATOMIC_ACQUIRE);
    mzhip_last_served.valid = 1;
    mzhip_last_served.valid_sha = 0;
}
/* (behind mzhip_served_set) the served buffer is part of a primed entry with a device-verified digest */
static inline void mzhip_served_set_entry(const uint8_t *base, int64_t off, int64_t usize, uint16_t alg, const uint8_t *digest) {
    mzhip_last_served.ent_base = base;
    mzhip_last_served.ent_off = off;
    mzhip_last_served.ent_usize = usize;
    mzhip_last_served.ent_alg = alg;
    mzhip_last_served.ent_digest = digest;
    mzhip_last_served.valid_sha = (alg != 0 && digest != 0) ? 1 : 0;
}
static inline void mzhip_served_drop(void) { mzhip_last_served.valid = mzhip_last_served.valid_sha = 0; }
/* the stream of this slot is about to free (or re-use) buffers a hint of some thread may point into */
static inline void mzhip_buffers_released(uint32_t slot) { (void)__atomic_add_fetch(&mzhip_stream_epoch[slot % MZHIP_STREAM_SLOTS], 1u, __ATOMIC_ACQ_REL); }

/* the memory bound of one READ stream (shim_zlib.c; mzhip_set_stream_window): bytes decoded per window, compressed bytes
 * pulled ahead of a window's launch */
int64_t mzh_stream_window(void);
int64_t mzh_stream_gulp(void);

#define MZHIP_PRIME_SEGMENT 65535 /* the reader's buffer size, mz_zip_rw.c:55 */
#endif
