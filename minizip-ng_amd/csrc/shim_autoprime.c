/* shim_autoprime.c -- the UNMODIFIED reader loop primes itself (on by default since round 5).
 *
 * mz_zip.c:1773 creates one codec stream per entry, so an application that is only re-linked against libmzhip.so pays one
 * launch and one PCIe round trip per entry: 33 MB/s on 64 KiB entries where one reference thread makes 350.
 * mzhip_prime_file() cures that with one line in the application; this file cures it with none: the first read() of an
 * entry walks down its base chain to the archive stream (compress stream -> crypt/raw stream -> zip->stream,
 * mz_zip.c:1765-1850), reads the archive through that stream's own vtbl (seek / tell / read, position restored afterwards),
 * hands the image to mzhip_prime_mem() -- every DEFLATE / LZMA / XZ entry decoded in one launch per codec -- and then looks
 * the entry up in the cache like any primed entry.  Everything else -- including every failure -- takes the ordinary
 * per-entry path with its exact error behaviour.
 *
 * When it happens (all must hold; MZHIP_AUTOPRIME in the environment of the process: "0" = never, "<n>" = archives of up to
 * n MiB, unset = 512):
 *   - the archive stream can seek, tell and read (a pipe cannot; the per-entry path serves it);
 *   - the archive is at most the limit, holds at least MZH_AUTOPRIME_MIN_ENTRIES entries a codec stream would be opened for
 *     (fewer: the per-entry path costs less than imaging the archive), and they decode to at most 4 x the limit (the cache
 *     is page-locked host memory; a bomb, or one entry of a huge archive, is not worth it);
 *   - this image (size + CRC of its last 64 KiB: the central directory) has not been primed already -- readers of several
 *     threads, one mz_zip_reader each over the same file, prime it once -- nor three times before (an application that
 *     alternates between archives entry by entry would otherwise re-image them for ever).
 * A new archive replaces the cache's previous generation (streams still reading from it keep it alive): one archive's
 * worth of decoded bytes at a time.  An application that calls mzhip_prime_* itself is left alone: while the cache holds
 * generations this file did not make, nothing is cleared or added. */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

#ifndef MZH_AUTOPRIME_MIN_ENTRIES
#define MZH_AUTOPRIME_MIN_ENTRIES 8
#endif
#define MZH_AUTOPRIME_DEFAULT_MIB 512
#define MZH_AUTOPRIME_SEEN 8

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static struct {
    int64_t size;
    uint32_t tail_crc;
    int32_t tries;
} g_seen[MZH_AUTOPRIME_SEEN]; /* the images tried so far, newest first */
static int64_t g_cur_size = -1; /* the image the cache holds now */
static uint32_t g_cur_crc;
static struct {
    const void *arch;
    int64_t size;
    uint32_t era;
    int32_t any; /* whether the cache held anything then */
} g_done[16]; /* archive streams that have been dealt with (primed, in the cache already, or not worth it) while the cache
                 was in the state it is in now: the call that every entry's first read() makes returns on these without I/O */
static uint32_t g_done_next, g_era; /* era: moves on whenever the cache changes under this file's feet */
int32_t mzhip_prime_any(void); /* mzhip_runtime.inc: the cache holds at least one generation */
static uint64_t g_autoprimed; /* archives primed this way (tests) */
MZHIP_API uint64_t mzhip_autoprime_count(void) { return __atomic_load_n(&g_autoprimed, __ATOMIC_RELAXED); }

static int32_t read_all(mzhip_stream *s, uint8_t *dst, int64_t n) {
    int64_t got = 0;
    while (got < n) {
        const int64_t left = n - got;
        const int32_t rd = s->vtbl->read(s, dst + got, (int32_t)(left < (1 << 30) ? left : (1 << 30)));
        if (rd <= 0)
            return 0;
        got += rd;
    }
    return 1;
}

void mzhip_autoprime(mzhip_stream *codec_base) {
    const char *env = getenv("MZHIP_AUTOPRIME");
    if (env && env[0] == '0')
        return;
    if (!codec_base || !codec_base->vtbl)
        return;
    mzhip_stream *arch = codec_base->base ? codec_base->base : codec_base;
    if (!arch->vtbl || !arch->vtbl->seek || !arch->vtbl->tell || !arch->vtbl->read || !arch->vtbl->is_open ||
        arch->vtbl->is_open(arch) != MZH_OK)
        return;
    int64_t limit = env && *env ? strtoll(env, NULL, 10) : 0;
    limit = (limit >= 1 ? limit : MZH_AUTOPRIME_DEFAULT_MIB) << 20;
    pthread_mutex_lock(&g_mu);
    if (g_cur_size >= 0 && !mzhip_prime_any()) { /* somebody cleared the cache: it holds nothing of ours any more */
        g_cur_size = -1;
        g_era++;
        for (int i = 0; i < MZH_AUTOPRIME_SEEN; i++)
            g_seen[i].tries = 0; /* (the application manages the cache itself: only images that evict EACH OTHER count as thrashing) */
    }
    const int64_t pos = arch->vtbl->tell(arch);
    if (pos >= 0 && arch->vtbl->seek(arch, 0, MZH_SEEK_END) == MZH_OK) {
        const int64_t size = arch->vtbl->tell(arch);
        uint8_t *buf = NULL;
        int64_t *table = NULL;
        int dealt_with = 0;
        const int32_t any_now = mzhip_prime_any();
        for (int i = 0; i < 16; i++)
            if (g_done[i].arch == (const void *)arch && g_done[i].size == size && g_done[i].era == g_era && g_done[i].any == any_now)
                dealt_with = 1;
        if (!dealt_with && g_cur_size < 0 && any_now) {
            /* the cache holds generations this file did not make: the application primes for itself (mzhip_prime_file ...);
             * nothing is cleared or added under its feet -- entries it did not prime take the per-entry path */
            dealt_with = 2;
        }
        if (!dealt_with && size >= 22 && size <= limit) {
            /* which image is this?  its size and the CRC of its tail (the end record and the central directory's end) */
            const int64_t tail = size < 65536 ? size : 65536;
            uint8_t *tb = (uint8_t *)malloc((size_t)tail);
            uint32_t tcrc = 0;
            int known = -1, ok = 0;
            if (tb && arch->vtbl->seek(arch, size - tail, MZH_SEEK_SET) == MZH_OK && read_all(arch, tb, tail)) {
                tcrc = mzhip_crc32_host(0, tb, (size_t)tail);
                ok = 1;
            }
            free(tb);
            if (ok && !(size == g_cur_size && tcrc == g_cur_crc)) { /* (the cache holds it already: nothing to do) */
                for (int i = 0; i < MZH_AUTOPRIME_SEEN; i++)
                    if (g_seen[i].size == size && g_seen[i].tail_crc == tcrc)
                        known = i;
                const int32_t tries = known >= 0 ? g_seen[known].tries : 0;
                if (tries < 3) {
                    /* newest first; an image seen before keeps its count */
                    const int from = known >= 0 ? known : MZH_AUTOPRIME_SEEN - 1;
                    memmove(&g_seen[1], &g_seen[0], (size_t)from * sizeof(g_seen[0]));
                    g_seen[0].size = size;
                    g_seen[0].tail_crc = tcrc;
                    g_seen[0].tries = tries + 1;
                    buf = (uint8_t *)malloc((size_t)size);
                    if (buf && arch->vtbl->seek(arch, 0, MZH_SEEK_SET) == MZH_OK && read_all(arch, buf, size)) {
                        /* worth it?  entries a codec stream would be opened for, and what they decode to */
                        const int64_t n = mzhip_zip_index_mem(buf, (uint64_t)size, NULL, 0);
                        if (n >= MZH_AUTOPRIME_MIN_ENTRIES && n <= (1 << 26) && (table = (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t))) != NULL &&
                            mzhip_zip_index_mem(buf, (uint64_t)size, table, n) == n) {
                            int64_t cnt = 0, usum = 0;
                            for (int64_t i = 0; i < n; i++) {
                                const int64_t m = table[8 * i];
                                if ((m == 8 || m == 14 || m == 95) && table[8 * i + 7] >= 0 && !(table[8 * i + 1] & 1) /* not encrypted */) {
                                    cnt++;
                                    usum += table[8 * i + 4];
                                }
                            }
                            if (cnt >= MZH_AUTOPRIME_MIN_ENTRIES && usum <= 4 * limit) {
                                mzhip_prime_clear(); /* one archive's worth of cache at a time */
                                g_cur_size = -1;
                                g_era++;
                                if (mzhip_prime_mem(buf, (uint64_t)size) > 0) {
                                    g_cur_size = size;
                                    g_cur_crc = tcrc;
                                    (void)__atomic_add_fetch(&g_autoprimed, 1, __ATOMIC_RELAXED);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (dealt_with != 1) {
            g_done[g_done_next % 16].arch = arch;
            g_done[g_done_next % 16].size = size;
            g_done[g_done_next % 16].era = g_era;
            g_done[g_done_next % 16].any = mzhip_prime_any();
            g_done_next++;
        }
        free(table);
        free(buf);
        arch->vtbl->seek(arch, pos, MZH_SEEK_SET);
    }
    pthread_mutex_unlock(&g_mu);
}
