/* shim_sha.c -- mz_crypt_sha_* (mz_crypt.h:29-35) for entries the device has already hashed.
 *
 * The reference's reader verifies an entry's Hash extra field (0x1a51) by running mz_crypt_sha_update over every
 * buffer mz_zip_entry_read fills and comparing mz_crypt_sha_end's digest with the field in mz_zip_reader_entry_close
 * (mz_zip_rw.c:409-451,462-467).  One message is one serial chain -- a GPU has nothing to offer a single stream of
 * update() calls -- but mzhip_prime_*() decodes a whole archive at once and computes the SHA-1 / SHA-256 of every such
 * entry on the device in the same pass (mzhip_sha_batch over the decoded bytes in HBM).  These seven symbols put that
 * digest where the reference looks for it:
 *
 *   update   a buffer that the READ stream has just served from a primed entry -- in order, from the entry's first byte
 *            -- is not hashed again: the context only remembers how far the entry has been presented (the bytes are
 *            compared with the primed ones, as for the CRC symbol);
 *   end      when exactly the whole primed entry went by and the algorithm is the one its Hash field names, the digest
 *            is the device's; mz_zip_reader_entry_close then compares it with the field as it always does.
 *
 * Anything else -- entries that are not primed, the writer's hash, other algorithms, a caller that hashes something of
 * its own -- is the reference's own implementation (mz_crypt_openssl.c etc.), which the link step keeps under other
 * names exactly as it keeps its CRC (-Dmz_crypt_sha_create=mz_ref_crypt_sha_create ...).  Those seven
 * mz_ref_crypt_sha_* symbols are weak here: without them these functions answer only from primed digests and fail
 * with MZ_SUPPORT_ERROR otherwise -- there is no second SHA implementation in this library.
 * A context that started on primed buffers and then sees anything else catches the reference context up from the primed
 * bytes first, so the result never depends on where the bytes came from. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

#define MZH_HASH_SHA1 20
#define MZH_HASH_SHA224 22
#define MZH_HASH_SHA256 23
#define MZH_HASH_SHA384 24
#define MZH_HASH_SHA512 25


extern void mz_ref_crypt_sha_reset(void *handle) __attribute__((weak));
extern int32_t mz_ref_crypt_sha_begin(void *handle) __attribute__((weak));
extern int32_t mz_ref_crypt_sha_update(void *handle, const void *buf, int32_t size) __attribute__((weak));
extern int32_t mz_ref_crypt_sha_end(void *handle, uint8_t *digest, int32_t digest_size) __attribute__((weak));
extern int32_t mz_ref_crypt_sha_set_algorithm(void *handle, uint16_t algorithm) __attribute__((weak));
extern void *mz_ref_crypt_sha_create(void) __attribute__((weak));
extern void mz_ref_crypt_sha_delete(void **handle) __attribute__((weak));

typedef struct mzhip_sha_s {
    void *ref;           /* the reference's context (null when its symbols are not linked in) */
    uint16_t algorithm;
    int8_t begun;
    int8_t on_ref;       /* the reference context holds everything presented so far */
    /* while every update() so far was the next piece of one primed entry: */
    const uint8_t *ent_base; /* the primed bytes: only read while the slot's epoch is the one recorded below (to_ref) */
    uint8_t ent_digest[32];  /* the device's digest of the entry, COPIED when the first piece went by: the primed
                                generation may be gone by the time end() is called (a stream closed early, mzhip_prime_clear,
                                a caller-owned context), the digest of the bytes that were presented stays what it is */
    int8_t have_digest;
    uint32_t ent_slot, ent_epoch; /* the stream slot whose buffers ent_base points into, and its epoch then */
    int64_t ent_usize, ent_seen;
    uint16_t ent_alg;
} mzhip_sha;

static uint64_t g_sha_primed_digests; /* mz_crypt_sha_end calls answered with a device digest */
MZHIP_API uint64_t mzhip_sha_primed_digests(void) { return __atomic_load_n(&g_sha_primed_digests, __ATOMIC_RELAXED); }

static int32_t have_ref(void) {
    return mz_ref_crypt_sha_create && mz_ref_crypt_sha_delete && mz_ref_crypt_sha_begin && mz_ref_crypt_sha_update &&
           mz_ref_crypt_sha_end && mz_ref_crypt_sha_set_algorithm && mz_ref_crypt_sha_reset;
}

static int32_t digest_bytes(uint16_t a) {
    return a == MZH_HASH_SHA1 ? 20 : a == MZH_HASH_SHA224 ? 28 : a == MZH_HASH_SHA256 ? 32 : a == MZH_HASH_SHA384 ? 48 : 64;
}

/* the reference context takes over: first whatever went by as primed buffers */
static int32_t to_ref(mzhip_sha *s) {
    if (s->on_ref)
        return MZH_OK;
    if (!s->ref)
        return MZH_SUPPORT_ERROR; /* no reference implementation linked in: only primed digests can be answered */
    /* the primed bytes are only there while the stream that served them has not given its buffers back */
    if (s->ent_seen > 0 && (!s->ent_base || s->ent_epoch != __atomic_load_n(&mzhip_stream_epoch[s->ent_slot], __ATOMIC_ACQUIRE)))
        return MZH_HASH_ERROR; /* the bytes that went by cannot be hashed again: the message is lost to this context */
    int64_t pos = 0;
    while (pos < s->ent_seen) {
        const int32_t n = (int32_t)(s->ent_seen - pos < (1 << 20) ? s->ent_seen - pos : (1 << 20));
        const int32_t err = mz_ref_crypt_sha_update(s->ref, s->ent_base + pos, n);
        if (err < 0)
            return err;
        pos += n;
    }
    s->on_ref = 1;
    s->ent_base = NULL;
    s->have_digest = 0;
    s->ent_seen = 0;
    return MZH_OK;
}

void mz_crypt_sha_reset(void *handle) {
    mzhip_sha *s = (mzhip_sha *)handle;
    if (!s)
        return;
    if (s->ref)
        mz_ref_crypt_sha_reset(s->ref);
    s->begun = 0;
    s->on_ref = 0;
    s->ent_base = NULL;
    s->have_digest = 0;
    s->ent_seen = s->ent_usize = 0;
}

int32_t mz_crypt_sha_begin(void *handle) {
    mzhip_sha *s = (mzhip_sha *)handle;
    if (!s)
        return MZH_PARAM_ERROR;
    mz_crypt_sha_reset(handle);
    if (s->ref) {
        const int32_t err = mz_ref_crypt_sha_begin(s->ref);
        if (err != MZH_OK)
            return err;
    }
    s->begun = 1;
    return MZH_OK;
}

int32_t mz_crypt_sha_update(void *handle, const void *buf, int32_t size) {
    mzhip_sha *s = (mzhip_sha *)handle;
    if (!s || !buf || !s->begun)
        return MZH_PARAM_ERROR;
    mzhip_served *h = &mzhip_last_served;
    if (!s->on_ref && h->valid_sha && h->buf == buf && h->size == size && size > 0 &&
        h->epoch == __atomic_load_n(&mzhip_stream_epoch[h->slot], __ATOMIC_ACQUIRE) &&
        (s->ent_seen == 0 ? h->ent_off == 0 : (h->ent_base == s->ent_base && h->ent_off == s->ent_seen)) &&
        memcmp(buf, h->ent_base + h->ent_off, (size_t)size) == 0) {
        /* the next piece of a primed entry, unchanged since it was served: hashed on the device already */
        h->valid_sha = 0;
        s->ent_base = h->ent_base;
        s->ent_slot = h->slot;
        s->ent_epoch = h->epoch;
        if (!s->have_digest) {
            memcpy(s->ent_digest, h->ent_digest, sizeof(s->ent_digest));
            s->have_digest = 1;
        }
        s->ent_usize = h->ent_usize;
        s->ent_alg = h->ent_alg;
        s->ent_seen += size;
        return size;
    }
    h->valid_sha = 0;
    const int32_t err = to_ref(s);
    if (err != MZH_OK)
        return err;
    return mz_ref_crypt_sha_update(s->ref, buf, size);
}

int32_t mz_crypt_sha_end(void *handle, uint8_t *digest, int32_t digest_size) {
    mzhip_sha *s = (mzhip_sha *)handle;
    if (!s || !digest || !s->begun)
        return MZH_PARAM_ERROR;
    if (digest_size < digest_bytes(s->algorithm))
        return MZH_PARAM_ERROR; /* mz_crypt_openssl.c:207-208 */
    if (!s->on_ref && s->ent_seen > 0 && s->ent_seen == s->ent_usize && s->ent_alg == s->algorithm && s->have_digest) {
        memcpy(digest, s->ent_digest, (size_t)digest_bytes(s->algorithm)); /* SHA-1: 20, SHA-256: 32 of the 32 kept */
        (void)__atomic_add_fetch(&g_sha_primed_digests, 1, __ATOMIC_RELAXED);
        return MZH_OK;
    }
    const int32_t err = to_ref(s);
    if (err != MZH_OK)
        return err;
    return mz_ref_crypt_sha_end(s->ref, digest, digest_size);
}

int32_t mz_crypt_sha_set_algorithm(void *handle, uint16_t algorithm) {
    mzhip_sha *s = (mzhip_sha *)handle;
    if (!s || algorithm < MZH_HASH_SHA1 || algorithm > MZH_HASH_SHA512)
        return MZH_PARAM_ERROR; /* mz_crypt_openssl.c:242-243 */
    if (s->ref) {
        const int32_t err = mz_ref_crypt_sha_set_algorithm(s->ref, algorithm);
        if (err != MZH_OK)
            return err;
    }
    s->algorithm = algorithm;
    return MZH_OK;
}

void *mz_crypt_sha_create(void) {
    mzhip_sha *s = (mzhip_sha *)calloc(1, sizeof(mzhip_sha));
    if (!s)
        return NULL;
    s->algorithm = MZH_HASH_SHA256; /* mz_crypt_openssl.c:251 */
    if (have_ref()) {
        s->ref = mz_ref_crypt_sha_create();
        if (!s->ref) {
            free(s);
            return NULL;
        }
    }
    return s;
}

void mz_crypt_sha_delete(void **handle) {
    if (!handle)
        return;
    mzhip_sha *s = (mzhip_sha *)*handle;
    if (s) {
        if (s->ref)
            mz_ref_crypt_sha_delete(&s->ref);
        free(s);
    }
    *handle = NULL;
}
