/* shim_autoprime.c -- opt-in: let the UNMODIFIED reader loop prime itself.
 *
 * mzhip_prime_file() needs one line in the application.  With MZHIP_AUTOPRIME set in the environment the codec
 * streams do it on their own: the first read() of an entry walks down its base chain to the archive stream
 * (compress stream -> crypt/raw stream -> zip->stream, mz_zip.c:1765-1850), reads the whole archive through that
 * stream's own vtbl (seek / tell / read, position restored afterwards), hands the image to mzhip_prime_mem() and
 * then looks the entry up in the cache like any primed entry.  Everything else -- including every failure -- takes
 * the ordinary per-entry path.  MZHIP_AUTOPRIME=<n> bounds the archive size to n MiB (default 1024). */
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static const void *g_last_arch;
static int64_t g_last_size = -1;

void mzhip_autoprime(mzhip_stream *codec_base) {
    const char *env = getenv("MZHIP_AUTOPRIME");
    if (!env || !*env || env[0] == '0')
        return;
    if (!codec_base || !codec_base->vtbl)
        return;
    mzhip_stream *arch = codec_base->base ? codec_base->base : codec_base;
    if (!arch->vtbl || !arch->vtbl->seek || !arch->vtbl->tell || !arch->vtbl->read || !arch->vtbl->is_open ||
        arch->vtbl->is_open(arch) != MZH_OK)
        return;
    int64_t limit = strtoll(env, NULL, 10);
    limit = (limit > 1 ? limit : 1024) << 20;
    pthread_mutex_lock(&g_mu);
    const int64_t pos = arch->vtbl->tell(arch);
    if (pos >= 0 && arch->vtbl->seek(arch, 0, MZH_SEEK_END) == MZH_OK) {
        const int64_t size = arch->vtbl->tell(arch);
        if (size > 0 && size <= limit && !(arch == g_last_arch && size == g_last_size)) {
            uint8_t *buf = (uint8_t *)malloc((size_t)size);
            int64_t got = 0;
            if (buf && arch->vtbl->seek(arch, 0, MZH_SEEK_SET) == MZH_OK) {
                while (got < size) {
                    const int64_t left = size - got;
                    const int32_t rd = arch->vtbl->read(arch, buf + got, (int32_t)(left < (1 << 30) ? left : (1 << 30)));
                    if (rd <= 0)
                        break;
                    got += rd;
                }
            }
            if (got == size)
                mzhip_prime_mem(buf, (uint64_t)size);
            free(buf);
            g_last_arch = arch; /* tried: do not read the same archive image again for every entry */
            g_last_size = size;
        }
        arch->vtbl->seek(arch, pos, MZH_SEEK_SET);
    }
    pthread_mutex_unlock(&g_mu);
}
