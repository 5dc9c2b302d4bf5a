/* shim_crc32.c -- mz_crypt_crc32_update re-implemented over the HIP backend.
 *
 * Drop-in for the CRC symbol of the reference's mz_crypt.c:35-92 (declared at
 * mz_crypt.h:20; called from mz_zip.c:2049,2064 and mz_os.c:340).  Same
 * chaining contract: crc(crc(0,A),B) == crc(0,A||B); size <= 0 returns `value`
 * unchanged, as every reference back-end does for an empty buffer.
 * The bytes are reduced on the device (k_crc32_batch, wave-parallel tile
 * folding); there is no table-driven CPU loop in this file.  When the buffer is
 * one that a primed stream just served (see mzhip_prime_*), the device already
 * computed its CRC and only the GF(2) chaining arithmetic happens here; the same for a chunk of a primed STORE entry,
 * recognised by content (fingerprint + memcmp against the primed payload).
 */
#include <string.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

__thread mzhip_served mzhip_last_served;
uint32_t mzhip_stream_epoch[MZHIP_STREAM_SLOTS];
uint32_t mzhip_stream_slot_next;

uint32_t mz_crypt_crc32_update(uint32_t value, const uint8_t *buf, int32_t size) {
    if (size <= 0 || !buf)
        return value;
    if (mzhip_last_served.valid && mzhip_last_served.buf == (const void *)buf && mzhip_last_served.size == size) {
        /* this buffer was just served from the prime cache (or written and found equal to a primed buffer); the CRC of
         * those bytes was computed on the device.  The record is dropped by every other codec call, and the bytes are
         * compared with the primed ones once more, so a buffer that changed in between is never answered from the cache */
        mzhip_last_served.valid = 0;
        if (mzhip_last_served.epoch == __atomic_load_n(&mzhip_stream_epoch[mzhip_last_served.slot], __ATOMIC_ACQUIRE) &&
            memcmp(buf, mzhip_last_served.src, (size_t)size) == 0)
            return mzhip_crc32_combine(value, mzhip_last_served.crc, (uint64_t)size);
    }
    mzhip_last_served.valid = 0;
    if (size >= (int32_t)MZHIP_CRC_HOST_BELOW) {
        /* a chunk of a primed STORE entry (the raw stream handed it over untouched, mz_zip.c:2047-2049): the device
         * computed its CRC when the archive was primed; byte-for-byte equality with the primed payload is checked */
        uint32_t c;
        if (mzhip_prime_store_crc(buf, size, &c))
            return mzhip_crc32_combine(value, c, (uint64_t)size);
    }
    return mzhip_crc32_host(value, buf, (size_t)size);
}
