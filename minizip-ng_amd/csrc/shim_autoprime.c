/* shim_autoprime.c -- the UNMODIFIED reader loop primes itself: archives of ANY size (round 6).
 *
 * mz_zip.c:1773 creates one codec stream per entry, so an application that is only re-linked against libmzhip.so pays one
 * launch and one PCIe round trip per entry: 33 MB/s on 64 KiB entries where one reference thread makes 350.
 * mzhip_prime_file() cures that with one line in the application; this file cures it with none: the first read() of an
 * entry walks down its base chain to the archive stream (compress stream -> crypt/raw stream -> zip->stream,
 * mz_zip.c:1765-1850) and reads the archive through that stream's own vtbl (seek / tell / read, position restored
 * afterwards).  The reference streams an archive of any size at a constant rate in constant memory (mz_zip.c:1757-1853,
 * mz_strm_zlib.c:116-193); so does this:
 *
 *   small archives (at most the limit: 512 MiB, or MZHIP_AUTOPRIME=<MiB>) are imaged whole and handed to mzhip_prime_mem()
 *     -- every DEFLATE / LZMA / XZ entry decoded in one launch per codec, STORE entries' CRCs too;
 *   larger ones are ROLLED OVER: the tail of the file (end records + central directory) is indexed once
 *     (mzhip_zip_index_tail), the entries a codec stream would be opened for are cut -- in the order their local headers
 *     lie in the file -- into windows of about 256 MiB of compressed + decoded bytes, and a window is imaged through the
 *     reader's stream and decoded AHEAD of the reader: the window the entry at hand lies in (its first use), and the one
 *     behind it right after (the look-ahead), each as a cache generation of its own on a worker thread's pipeline
 *     (mzhip_prime_window_begin: H2D, launches, D2H of chunk i+1 / i / i-1 at once; the reader is served chunk by chunk).
 *     Windows that have been read are evicted, least recently used first, to keep the page-locked bytes of all live windows
 *     under 4 x the limit (2 GiB); readers of several threads -- one mz_zip_reader each over the same file, each in its own
 *     part of it -- share the windows.
 *     When the archive is a regular file of this process (one of /proc/self/fd has its size and its last 64 KiB), the file is
 *     opened once more through that entry and the windows are imaged with pread() by a thread of this file's own -- the reader
 *     never waits for a look-ahead, and never copies the archive through its 32 KiB stream buffer (mz_strm_buf.c:115-180) --;
 *     otherwise (a memory stream, a custom stream) the reader's thread images a window through its stream when it first needs it.
 * Every failure, and every entry no window holds (encrypted, STORE, larger than a window), takes the ordinary per-entry path
 * with its exact error behaviour.
 *
 * When it happens (all must hold; MZHIP_AUTOPRIME in the environment of the process: "0" = never, "<n>" = limit of n MiB
 * ("<n>k": KiB, for tests), unset = 512):
 *   - the archive stream can seek, tell and read (a pipe cannot; the per-entry path serves it);
 *   - the archive holds at least MZH_AUTOPRIME_MIN_ENTRIES entries a codec stream would be opened for (fewer: the per-entry
 *     path costs less than imaging the archive); imaged whole, they decode to at most 4 x the limit (the cache is
 *     page-locked host memory; a bomb is not worth it -- the sum is taken in unsigned arithmetic over exactly the rows
 *     mzhip_prime_mem() would take and cut off at the bound);
 *   - whole image: this image (size + CRC of its last 64 KiB: the central directory) has not been primed already --
 *     readers of several threads prime it once -- nor three times before (an application that alternates between archives
 *     entry by entry would otherwise re-image them for ever); rolled over: a window that was evicted four times in a row
 *     before eight of its entries (a quarter of a small window's) had been read is left to the per-entry path for the next
 *     4096 entries (the same guard, per window).
 * Which archive a stream belongs to is asked on EVERY call, not remembered by stream address (the allocator hands a freed
 * reader's address to the next one): its size and a hash of its last 4 KiB, read through the stream on every call.
 * The cache's other generations are never touched: a new whole image replaces the previous one THIS FILE made (by identity,
 * not mzhip_prime_clear()), windows are dropped one by one.  An application that calls mzhip_prime_* itself is left alone by
 * the whole-image path (while the cache holds generations this file did not make, it adds nothing); rolling over an archive
 * goes on beside them. */
#define _GNU_SOURCE
#include <dirent.h>
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

#ifndef MZH_AUTOPRIME_MIN_ENTRIES
#define MZH_AUTOPRIME_MIN_ENTRIES 8
#endif
#define MZH_AUTOPRIME_DEFAULT_MIB 512
#define MZH_AUTOPRIME_SEEN 8
#define MZH_ROLLS 4                       /* archives rolled over at the same time */
#define MZH_ROLL_WINDOW ((uint64_t)256 << 20) /* compressed + decoded bytes of a window, at most (and at most 1/8 of the budget) */
#define MZH_ROLL_GAP ((uint64_t)8 << 20)  /* bytes between two entries' local headers that a window does not image */
#define MZH_ROLL_FRESH 64                 /* calls (of all readers together): a window used this recently is some reader's; it is not evicted for a look-ahead, and for a needed window only past twice the budget */
#define MZH_ROLL_MAX_WINDOWS 48           /* live at once (the cache holds 64 generations) */
#define MZH_ROLL_DEAD_FOR 4096             /* calls a window that was given up stays with the per-entry path */
#define MZH_TAIL4K 4096

int64_t mzhip_prime_window_begin(uint8_t *img, size_t img_cap, uint64_t win_off, uint64_t win_len, const int64_t *rows, int64_t n,
                                 const uint16_t *alg, const uint16_t *dsz, const uint8_t *dig, uint64_t zip_len, uint64_t ident,
                                 int32_t device, uint64_t *out_bytes);
void mzhip_prime_drop(uint64_t zip_len, uint64_t ident);
int32_t mzhip_prime_has(uint64_t zip_len, uint64_t ident);
void mzhip_prime_counts(int32_t *gens, int32_t *windows, uint64_t *clears);
uint64_t mzhip_prime_clears(void);
int32_t mzhip_prime_current_device(void);

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t g_cv = PTHREAD_COND_INITIALIZER; /* a window left the BUSY state */

/* ---- whole images ---- */
static struct {
    int64_t size;
    uint64_t tail_crc;
    int32_t tries;
} g_seen[MZH_AUTOPRIME_SEEN]; /* the images tried so far, newest first */
static int64_t g_cur_size = -1; /* the whole image the cache holds now: size, CRC of its last 64 KiB, the generation's identity */
static uint64_t g_cur_crc;
static uint64_t g_cur_ident;
static struct {
    int64_t size;
    uint64_t crc4; /* hash of the image's last 4 KiB */
    uint32_t era;
    int32_t foreign; /* whether the cache held generations of the application's then */
} g_done[16]; /* images that have been dealt with (primed, in the cache already, or not worth it) while the cache was in the
                 state it is in now: the call that every entry's first read() makes returns on these after the 4 KiB tail read */
static uint32_t g_done_next, g_era; /* era: moves on whenever the cache changes under this file's feet */
static uint64_t g_clears_seen;
static uint64_t g_autoprimed, g_windows_primed, g_windows_evicted, g_peak_bytes; /* (tests, reports) */
MZHIP_API uint64_t mzhip_autoprime_count(void) { return __atomic_load_n(&g_autoprimed, __ATOMIC_RELAXED); }

#ifndef MZH_ROLL_WINDOW_DIV
#define MZH_ROLL_WINDOW_DIV 16 /* a window is at most the budget / this: 128 MiB of compressed + decoded bytes by default.  (8 until the end of round 6: eight readers, a window and a look-ahead each, are twice the budget in 256 MiB windows -- windows in use were evicted, one in three passes left a window to the per-entry path: 2 GiB/s instead of 6 - 8, tests/diag_roll_T.py) */
#endif
/* ---- archives that are rolled over ---- */
enum { W_NONE = 0, W_BUSY, W_LIVE, W_DEAD };
typedef struct {
    uint64_t lo, hi;  /* the bytes of the file the window images */
    int64_t r0, r1;   /* its rows */
    uint64_t need;    /* bytes of the cache it will take: what its rows decode to, as the cache lays them out */
    uint64_t held;    /* ... and holds while live */
    uint64_t ident;
    uint64_t stamp;   /* the call that used it last */
    uint32_t hits;    /* calls that used it since it went live */
    int32_t state, quick; /* quick: evictions in a row after fewer than 8 hits */
    int32_t ahead;    /* primed as a look-ahead and not asked for yet: nobody's window so far */
} roll_win;
typedef struct {
    int64_t size;
    uint64_t crc4, tail_crc; /* hashes of its last 4 / 64 KiB */
    uint64_t ident, stamp;
    int64_t n;
    int64_t *rows; /* n x 8 (mzhip_zip_index_mem's columns), sorted by local header offset; payload offsets are filled in window by window */
    uint16_t *alg, *dsz; /* Hash extra fields per row, or NULL: no row has one */
    uint8_t *dig;
    int32_t nwin, busy; /* busy: threads inside I/O for one of its windows (and windows queued for the imaging thread) */
    roll_win *win;
    int fd; /* the archive opened once more (>= 0): windows are imaged with pread() by the imaging thread */
    uint64_t wmax; /* what a window of this archive holds at most (compressed + decoded bytes) */
} roll;
static roll *g_rolls[MZH_ROLLS];
static uint64_t g_tick, g_live_bytes; /* page-locked bytes of the live windows of all rolls */
static uint64_t g_busy_bytes;         /* ... and what the windows that are being imaged and decoded right now will take: they count against the
                                       * budget like live ones (with four readers, seven windows on their way were 1.8 GB nobody had counted) */

MZHIP_API void mzhip_autoprime_stats(uint64_t *windows_primed, uint64_t *windows_evicted, uint64_t *live_bytes, uint64_t *peak_bytes) {
    pthread_mutex_lock(&g_mu);
    if (windows_primed)
        *windows_primed = g_windows_primed;
    if (windows_evicted)
        *windows_evicted = g_windows_evicted;
    if (live_bytes)
        *live_bytes = g_live_bytes;
    if (peak_bytes)
        *peak_bytes = g_peak_bytes;
    pthread_mutex_unlock(&g_mu);
}

static uint64_t fnv1a64(const uint8_t *p, uint64_t n, uint64_t h) {
    for (uint64_t i = 0; i < n; i++)
        h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}
#define FNV0 0xCBF29CE484222325ull
/* what names an image: a 64-bit hash of its last bytes, eight at a time, on the host (this runs on every entry's first read:
 * mz_crypt_crc32_update's own arithmetic would send 4 KiB and more to the device) */
static uint64_t tail_hash(const uint8_t *p, size_t n) {
    uint64_t h[4] = {0x9E3779B97F4A7C15ull ^ n, 0xC2B2AE3D27D4EB4Full, 0x165667B19E3779F9ull, 0x27D4EB2F165667C5ull};
    size_t i = 0;
    for (; i + 32 <= n; i += 32) /* four independent chains: the multiplies overlap */
        for (int k = 0; k < 4; k++) {
            uint64_t w;
            memcpy(&w, p + i + 8 * k, 8);
            h[k] = (h[k] ^ w) * 0xFF51AFD7ED558CCDull;
            h[k] ^= h[k] >> 29;
        }
    uint64_t r = h[0];
    for (int k = 1; k < 4; k++)
        r = (r ^ h[k]) * 0xC4CEB9FE1A85EC53ull + k;
    for (; i < n; i++)
        r = (r ^ p[i]) * 0x100000001B3ull;
    return r ^ (r >> 32);
}

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
static int32_t read_at(mzhip_stream *s, int64_t off, uint8_t *dst, int64_t n) {
    return s->vtbl->seek(s, off, MZH_SEEK_SET) == MZH_OK && read_all(s, dst, n);
}

/* is row t one that mzhip_prime_mem() / a window would decode?  (the conditions of prime_publish, mzhip_prime.inc) */
static int row_codec(const int64_t *t) {
    return (t[0] == 8 || t[0] == 14 || t[0] == 95) && !(t[1] & 1) && t[3] >= 0 && t[4] >= 0 && t[3] < ((int64_t)1 << 31) &&
           t[4] < ((int64_t)1 << 31);
}

static int roll_trace(void) {
    static int on = -1;
    if (on < 0)
        on = getenv("MZHIP_PRIME_TRACE") != NULL;
    return on;
}

static void roll_free(roll *r) {
    if (!r)
        return;
    for (int32_t w = 0; w < r->nwin; w++)
        if (r->win[w].state == W_LIVE) {
            mzhip_prime_drop((uint64_t)r->size, r->win[w].ident);
            g_live_bytes -= r->win[w].held;
        }
    free(r->rows);
    free(r->alg);
    free(r->dsz);
    free(r->dig);
    free(r->win);
    if (r->fd >= 0)
        close(r->fd);
    free(r);
}

static int cmp_rows_by_loff(const void *a, const void *b) {
    const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
    return x[5] < y[5] ? -1 : x[5] > y[5] ? 1 : x[6] < y[6] ? -1 : x[6] > y[6];
}

/* Is the archive a regular file this process has open?  The descriptor table is looked through for a file of its size whose last
 * bytes are the ones the stream delivered; that file is opened once more (a description of its own: pread() at any offset from
 * any thread, whatever the application does with its descriptor).  -1: no (a memory stream, another kind of stream, no /proc). */
static int own_descriptor(int64_t size, const uint8_t *tail, int64_t tail_len) {
    DIR *d = opendir("/proc/self/fd");
    if (!d)
        return -1;
    int found = -1;
    uint8_t *probe = (uint8_t *)malloc((size_t)tail_len);
    struct dirent *e;
    while (probe && found < 0 && (e = readdir(d)) != NULL) {
        if (e->d_name[0] < '0' || e->d_name[0] > '9')
            continue;
        const int fd = atoi(e->d_name);
        struct stat sb;
        if (fd == dirfd(d) || fstat(fd, &sb) != 0 || !S_ISREG(sb.st_mode) || (int64_t)sb.st_size != size)
            continue;
        if (pread(fd, probe, (size_t)tail_len, (off_t)(size - tail_len)) != (ssize_t)tail_len || memcmp(probe, tail, (size_t)tail_len) != 0)
            continue;
        char path[64];
        snprintf(path, sizeof(path), "/proc/self/fd/%d", fd);
        found = open(path, O_RDONLY | O_CLOEXEC);
    }
    free(probe);
    closedir(d);
    return found;
}

/* index the archive from its tail and cut it into windows.  NULL: not worth it / not possible (the per-entry path serves it) */
static roll *roll_new(mzhip_stream *arch, int64_t size, uint64_t crc4, uint64_t tail_crc, uint64_t budget) {
    uint64_t from = size > (1 << 18) ? (uint64_t)size - (1 << 18) : 0;
    uint8_t *tail = NULL;
    int64_t *table = NULL;
    roll *r = NULL;
    int64_t n = 0;
    for (int pass = 0; pass < 4; pass++) { /* (end record, ZIP64 end record, central directory: three reasons to go back further) */
        const uint64_t len = (uint64_t)size - from;
        if (len > ((uint64_t)1 << 31))
            goto out; /* a central directory of 2 GiB: tens of millions of entries -- not this file's business */
        free(tail);
        tail = (uint8_t *)malloc((size_t)len);
        if (!tail || !read_at(arch, (int64_t)from, tail, (int64_t)len))
            goto out;
        uint64_t need = 0;
        n = mzhip_zip_index_tail(tail, from, (uint64_t)size, NULL, 0, &need);
        if (n != MZHIP_INDEX_NEED_MORE)
            break;
        if (need >= from)
            goto out;
        from = need;
    }
    if (n < MZH_AUTOPRIME_MIN_ENTRIES || n > (1 << 23)) /* (eight million entries: the table and the Hash fields are ~130 bytes a row) */
        goto out;
    table = (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t));
    if (!table || mzhip_zip_index_tail(tail, from, (uint64_t)size, table, n, NULL) != n)
        goto out;
    {
        /* the rows a window can take: codec entries that fit one, in front of the central directory */
        static int wdiv = 0;
        if (!wdiv) {
            const char *e = getenv("MZHIP_ROLL_WINDOW_DIV"); /* (A/B) windows of budget / this */
            const int v = e ? atoi(e) : 0;
            wdiv = v >= 4 && v <= 256 ? v : MZH_ROLL_WINDOW_DIV;
        }
        const uint64_t wmax = budget / (uint64_t)wdiv < MZH_ROLL_WINDOW ? budget / (uint64_t)wdiv : MZH_ROLL_WINDOW;
        const int64_t cd0 = table[6];
        uint16_t *alg = (uint16_t *)malloc((size_t)n * 2), *dsz = (uint16_t *)malloc((size_t)n * 2);
        uint8_t *dig = (uint8_t *)malloc((size_t)n * 64);
        int64_t nh = -1;
        if (alg && dsz && dig)
            nh = mzhip_zip_index_hash_tail(tail, from, (uint64_t)size, table, n, alg, dsz, dig);
        /* (the Hash fields ride in column 2 -- the CRC, which a window does not need -- as the row's index, through the sort) */
        int64_t k = 0;
        for (int64_t i = 0; i < n; i++) {
            int64_t *t = table + 8 * i;
            if (!row_codec(t) || t[5] < 0 || t[5] >= cd0 || (uint64_t)t[3] + (uint64_t)t[4] > wmax)
                continue;
            if (k != i)
                memcpy(table + 8 * k, t, 8 * sizeof(int64_t));
            table[8 * k + 2] = i;
            table[8 * k + 7] = -1;
            k++;
        }
        if (k < MZH_AUTOPRIME_MIN_ENTRIES) {
            free(alg);
            free(dsz);
            free(dig);
            goto out;
        }
        qsort(table, (size_t)k, 8 * sizeof(int64_t), cmp_rows_by_loff);
        r = (roll *)calloc(1, sizeof(roll));
        if (r)
            r->fd = -1;
        if (r && nh > 0) {
            r->alg = (uint16_t *)malloc((size_t)k * 2);
            r->dsz = (uint16_t *)malloc((size_t)k * 2);
            r->dig = (uint8_t *)malloc((size_t)k * 64);
            if (r->alg && r->dsz && r->dig) {
                for (int64_t j = 0; j < k; j++) {
                    const int64_t i = table[8 * j + 2];
                    r->alg[j] = alg[i];
                    r->dsz[j] = dsz[i];
                    memcpy(r->dig + 64 * j, dig + 64 * i, 64);
                }
            } else {
                free(r->alg);
                free(r->dsz);
                free(r->dig);
                r->alg = r->dsz = NULL;
                r->dig = NULL;
            }
        }
        free(alg);
        free(dsz);
        free(dig);
        if (!r)
            goto out;
        r->size = size;
        r->wmax = wmax;
        r->crc4 = crc4;
        r->tail_crc = tail_crc;
        r->ident = fnv1a64(tail + ((uint64_t)cd0 - from), (uint64_t)size - (uint64_t)cd0, FNV0); /* as a whole image's generation */
        r->n = k;
        r->rows = table;
        table = NULL;
        /* windows: rows in file order until the budget of a window is reached or the file has a gap.  What a row occupies is
         * not known before its local header has been read (30 + name + extra, which may differ from the central record's): a
         * window images 128 KiB + the payload behind its last row's header, cut off at the central directory */
        int32_t cap = 16, nw = 0;
        roll_win *win = (roll_win *)calloc((size_t)cap, sizeof(roll_win));
        uint64_t acc = 0, end = 0;
        for (int64_t j = 0; win && j < k; j++) {
            const int64_t *t = r->rows + 8 * j;
            const uint64_t lo = (uint64_t)t[5], bytes = (uint64_t)t[3] + (uint64_t)t[4];
            uint64_t hi = lo + 30 + 2 * 65535 + (uint64_t)t[3];
            if (hi > (uint64_t)cd0)
                hi = (uint64_t)cd0;
            if (hi < end)
                hi = end; /* (entries that overlap: the window only grows) */
            const int fresh = nw == 0 || acc + bytes > wmax || lo + (uint64_t)t[3] - win[nw - 1].lo > wmax || lo > end + MZH_ROLL_GAP;
            if (fresh) {
                if (nw == cap) {
                    roll_win *nwv = (roll_win *)realloc(win, (size_t)cap * 2 * sizeof(roll_win));
                    if (!nwv) {
                        free(win);
                        win = NULL;
                        break;
                    }
                    memset(nwv + cap, 0, (size_t)cap * sizeof(roll_win));
                    win = nwv;
                    cap *= 2;
                }
                if (nw && lo < win[nw - 1].hi)
                    win[nw - 1].hi = lo > win[nw - 1].lo ? lo : win[nw - 1].lo; /* (windows do not overlap: a look-up finds one) */
                win[nw].lo = lo;
                win[nw].r0 = j;
                win[nw].ident = fnv1a64((const uint8_t *)&lo, 8, r->ident ^ 0x9E3779B97F4A7C15ull);
                nw++;
                acc = 0;
            }
            acc += bytes;
            end = hi;
            win[nw - 1].hi = hi;
            win[nw - 1].r1 = j + 1;
            win[nw - 1].need += ((uint64_t)t[4] + 15) & ~(uint64_t)15;
        }
        if (!win) {
            roll_free(r);
            r = NULL;
            goto out;
        }
        r->win = win;
        r->nwin = nw;
        {
            const char *fenv = getenv("MZHIP_AUTOPRIME_FD"); /* "0": image through the reader's stream only (tests, comparison) */
            const int64_t tl = (uint64_t)size - from < 65536 ? (int64_t)((uint64_t)size - from) : 65536;
            if (!(fenv && fenv[0] == '0'))
                r->fd = own_descriptor(size, tail + (((uint64_t)size - from) - (uint64_t)tl), tl);
        }
    }
out:
    free(tail);
    free(table);
    return r;
}

/* make room for `need` more bytes of cache under `budget`: evict live windows, least recently used first -- never `keep` (of
 * roll rk), and windows some reader is in (used in the last MZH_ROLL_FRESH calls) or is about to be in (a look-ahead nobody
 * has reached) only for a window that is NEEDED and would take the cache past twice the budget.  1 = go ahead. */
static int roll_make_room(uint64_t need, uint64_t budget, const roll *rk, int32_t keep, int lookahead) {
    static int count_busy = -1;
    if (count_busy < 0) {
        /* "1": windows on their way count against the budget like live ones.  Measured with 8 file readers on the config-2 archive
         * (tests/diag_roll_T.py): the look-aheads that then go without make every pass ~2 GiB/s instead of 5 - 8 -- off */
        const char *e = getenv("MZHIP_ROLL_BUSY");
        count_busy = (e && e[0] == '1') ? 1 : 0;
    }
    const uint64_t busy_bytes = count_busy ? g_busy_bytes : 0;
    for (;;) {
        int32_t live = 0;
        roll *vr = NULL, *fr = NULL;
        int32_t vw = -1, fw = -1; /* the least recently used window nobody has used lately / of those in use */
        for (int i = 0; i < MZH_ROLLS; i++) {
            roll *r = g_rolls[i];
            if (!r)
                continue;
            for (int32_t w = 0; w < r->nwin; w++) {
                if (r->win[w].state != W_LIVE)
                    continue;
                live++;
                if (r == rk && (w == keep || (lookahead && w == keep + 1)))
                    continue;
                if (g_tick - r->win[w].stamp <= MZH_ROLL_FRESH || (r->win[w].ahead && r->win[w].hits == 0)) {
                    /* some reader is in it (several threads, a part of the archive each) -- or some reader is about to be: a
                     * look-ahead that has not been reached yet (its stamp is as old as its reader's window is long) */
                    if (fw < 0 || r->win[w].stamp < fr->win[fw].stamp) {
                        fr = r;
                        fw = w;
                    }
                    continue;
                }
                if (vw < 0 || r->win[w].stamp < vr->win[vw].stamp) {
                    vr = r;
                    vw = w;
                }
            }
        }
        if (g_live_bytes + busy_bytes + need <= budget && live < MZH_ROLL_MAX_WINDOWS)
            return 1;
        if (vw < 0) {
            /* nothing but windows in use is left.  A look-ahead does without; a window that is needed goes over the budget -- up
             * to twice: T readers hold T windows -- rather than take the bytes another reader is being served from (it would
             * prime them again a moment later: measured, 62 windows primed for an archive of 32) */
            if (lookahead)
                return 0;
            if (fw < 0 || (g_live_bytes + busy_bytes + need <= 2 * budget && live < MZH_ROLL_MAX_WINDOWS))
                return 1;
            vr = fr;
            vw = fw;
        }
        roll_win *v = &vr->win[vw];
        mzhip_prime_drop((uint64_t)vr->size, v->ident); /* (streams that still read from it keep it alive until they close) */
        g_live_bytes -= v->held;
        v->held = 0;
        {
            /* the thrash guard: priming a window pays after about eight entries have been read from it (a quarter of a small
             * window's: readers of several threads meet in the windows where their shares touch) */
            const int64_t all = v->r1 - v->r0, pays = all >= 32 ? 8 : (all >= 4 ? all / 4 : 1);
            if (!(v->ahead && v->hits == 0)) /* (a look-ahead nobody reached says nothing about the readers' pattern) */
                v->quick = (int64_t)v->hits < pays ? v->quick + 1 : 0;
        }
        v->state = v->quick >= 4 ? W_DEAD : W_NONE;
        v->stamp = g_tick; /* (a dead window comes back after MZH_ROLL_DEAD_FOR calls: the access pattern may have changed) */
        if (v->state == W_DEAD && roll_trace())
            fprintf(stderr, "[mzhip autoprime] window %d evicted four times in a row before it had paid for itself: left to the per-entry path\n", vw);
        g_windows_evicted++;
        if (roll_trace())
            fprintf(stderr, "[mzhip autoprime] window %d evicted after %u hits (%s); %llu bytes live\n", vw, v->hits, lookahead ? "look-ahead" : "needed",
                    (unsigned long long)g_live_bytes);
    }
}

/* image window w of r -- through the reader's stream `arch`, or with pread() on the roll's own descriptor when arch is NULL --
 * and start its decode on `device` (-1: the calling thread's).  The window is BUSY and r->busy counts it; g_mu is NOT held.  Ends with the lock taken, the window LIVE
 * or DEAD and everybody who waits for it woken; returns with the lock HELD. */
static void roll_image(roll *r, int32_t w, mzhip_stream *arch, int lookahead, int32_t device) {
    roll_win *W = &r->win[w];
    const uint64_t lo = W->lo, len = W->hi - W->lo;
    const int64_t r0 = W->r0, nr = W->r1 - W->r0;
    size_t cap = 0;
    uint8_t *img = (uint8_t *)mzhip_window_alloc((size_t)len + 16, &cap); /* page-locked: the H2D copies run at link speed */
    if (!img) {
        cap = 0;
        img = (uint8_t *)malloc((size_t)len + 16);
    }
    int64_t k = -1;
    uint64_t held = 0;
    int64_t *rows = (int64_t *)malloc((size_t)nr * 8 * sizeof(int64_t));
    int got = 0;
    if (img && rows) {
        if (arch) {
            got = read_at(arch, (int64_t)lo, img, (int64_t)len);
        } else {
            uint64_t done = 0;
            while (done < len) {
                const ssize_t rd = pread(r->fd, img + done, (size_t)(len - done < ((uint64_t)1 << 30) ? len - done : ((uint64_t)1 << 30)), (off_t)(lo + done));
                if (rd <= 0)
                    break;
                done += (uint64_t)rd;
            }
            got = done == len;
        }
    }
    if (got) {
        memcpy(rows, r->rows + 8 * r0, (size_t)nr * 8 * sizeof(int64_t)); /* (r->rows itself is immutable: other threads read it) */
        (void)mzhip_zip_index_resolve(img, lo, len, rows, nr);
        k = mzhip_prime_window_begin(img, cap, lo, len, rows, nr, r->alg ? r->alg + r0 : NULL, r->dsz ? r->dsz + r0 : NULL,
                                     r->dig ? r->dig + 64 * r0 : NULL, (uint64_t)r->size, W->ident, device, &held);
        img = NULL; /* (the call took it over) */
    }
    if (img) {
        if (cap)
            mzhip_window_free(img, cap);
        else
            free(img);
    }
    free(rows);
    pthread_mutex_lock(&g_mu);
    r->busy--;
    g_busy_bytes -= W->need < g_busy_bytes ? W->need : g_busy_bytes;
    if (k > 0) {
        W->state = W_LIVE;
        W->held = held;
        W->hits = 0;
        W->ahead = lookahead;
        W->stamp = g_tick;
        g_live_bytes += held;
        if (g_live_bytes > g_peak_bytes)
            g_peak_bytes = g_live_bytes;
        g_windows_primed++;
        if (roll_trace())
            fprintf(stderr, "[mzhip autoprime] window %d [%llu, %llu) %s%s: %lld entries, %llu bytes; %llu bytes live\n", w, (unsigned long long)lo,
                    (unsigned long long)(lo + len), lookahead ? "ahead" : "needed", arch ? "" : " (pread)", (long long)k, (unsigned long long)held,
                    (unsigned long long)g_live_bytes);
    } else {
        W->state = W_DEAD; /* could not be read or decoded: its entries take the per-entry path */
        if (roll_trace())
            fprintf(stderr, "[mzhip autoprime] window %d [%llu, %llu) could not be primed (read %d, result %lld)\n", w, (unsigned long long)lo,
                    (unsigned long long)(lo + len), got, (long long)k);
    }
    pthread_cond_broadcast(&g_cv);
}

/* the imaging threads (up to MZH_IMG_THREADS, started as they are needed): windows of archives this process holds as files, in the
 * order they were asked for */
typedef struct img_job_s {
    roll *r;
    int32_t w, lookahead, device; /* device: the one the reader that asked was on */
    struct img_job_s *next;
} img_job;
static img_job *g_q_head, *g_q_tail;
static pthread_cond_t g_q_cv = PTHREAD_COND_INITIALIZER;
static int g_img_threads, g_img_idle, g_img_failed; /* imaging threads started / waiting for work; one could not be started */
#define MZH_IMG_THREADS 3 /* windows imaged at once: readers of several threads need their first windows at the same moment */

static int g_quiesce, g_img_running; /* the process is exiting: no new window is taken up / windows being imaged right now */
static pthread_cond_t g_quiet_cv = PTHREAD_COND_INITIALIZER;

/* at exit (mzhip_prime.inc prime_at_exit): the imaging threads finish the window they are reading and take no other; windows
 * still queued stay BUSY for ever -- nobody is going to ask */
void mzhip_autoprime_quiesce(void) {
    pthread_mutex_lock(&g_mu);
    g_quiesce = 1;
    while (g_img_running > 0)
        pthread_cond_wait(&g_quiet_cv, &g_mu);
    pthread_mutex_unlock(&g_mu);
}

static void *img_thread(void *arg) {
    (void)arg;
    pthread_mutex_lock(&g_mu);
    for (;;) {
        g_img_idle++;
        while (!g_q_head || g_quiesce)
            pthread_cond_wait(&g_q_cv, &g_mu);
        g_img_idle--;
        g_img_running++;
        img_job *j = g_q_head;
        g_q_head = j->next;
        if (!g_q_head)
            g_q_tail = NULL;
        pthread_mutex_unlock(&g_mu);
        roll_image(j->r, j->w, NULL, j->lookahead, j->device); /* (takes the lock again) */
        free(j);
        g_img_running--;
        pthread_cond_broadcast(&g_quiet_cv);
    }
    return NULL;
}

/* window w of r must be live (or on its way).  g_mu is held on entry and on exit, not while the archive is read.
 * lookahead: the window is not needed yet -- no eviction of windows in use, no waiting. */
static void roll_ensure(roll *r, int32_t w, mzhip_stream *arch, uint64_t budget, int lookahead) {
    roll_win *W = &r->win[w];
    for (;;) {
        if (W->state == W_LIVE && !mzhip_prime_has((uint64_t)r->size, W->ident)) { /* (the cache let it go: 64 generations, a re-prime) */
            g_live_bytes -= W->held;
            W->held = 0;
            W->state = W_NONE;
        }
        if (W->state == W_DEAD && W->quick >= 4 && g_tick - W->stamp > MZH_ROLL_DEAD_FOR) { /* given up for thrashing, long ago: once more */
            W->quick = 0;
            W->state = W_NONE;
        }
        if (W->state == W_LIVE || W->state == W_DEAD)
            return;
        if (W->state == W_BUSY) {
            if (lookahead)
                return;
            pthread_cond_wait(&g_cv, &g_mu);
            continue;
        }
        /* W_NONE */
        if (!roll_make_room(W->need, budget, r, lookahead ? w - 1 : w, lookahead) && lookahead)
            return; /* (a window that is needed goes over the budget rather than without) */
        W->state = W_BUSY;
        r->busy++;
        g_busy_bytes += W->need;
        if (r->fd >= 0 && !(g_img_failed && g_img_threads == 0)) { /* an imaging thread reads it; a needed window is waited for above */
            if (g_img_idle == 0 && g_img_threads < MZH_IMG_THREADS && !g_img_failed) {
                pthread_t t;
                if (pthread_create(&t, NULL, img_thread, NULL) == 0) {
                    pthread_detach(t);
                    g_img_threads++;
                } else {
                    g_img_failed = 1;
                }
            }
            img_job *j = g_img_threads > 0 ? (img_job *)malloc(sizeof(img_job)) : NULL;
            if (j) {
                j->r = r;
                j->w = w;
                j->lookahead = lookahead;
                j->device = mzhip_prime_current_device();
                j->next = NULL;
                if (g_q_tail)
                    g_q_tail->next = j;
                else
                    g_q_head = j;
                g_q_tail = j;
                pthread_cond_signal(&g_q_cv);
                continue;
            }
        }
        pthread_mutex_unlock(&g_mu);
        roll_image(r, w, arch, lookahead, -1);
        return;
    }
}

static void forget_everything(void) {
    g_cur_size = -1;
    g_era++;
    for (int i = 0; i < MZH_AUTOPRIME_SEEN; i++)
        g_seen[i].tries = 0; /* (the application manages the cache itself: only images that evict EACH OTHER count as thrashing) */
    for (int i = 0; i < MZH_ROLLS; i++) {
        roll *r = g_rolls[i];
        if (!r)
            continue;
        for (int32_t w = 0; w < r->nwin; w++)
            if (r->win[w].state == W_LIVE) {
                r->win[w].state = W_NONE;
                r->win[w].held = 0;
            }
    }
    g_live_bytes = 0;
}

/* ... from the central directory, for the entry whose local header stands at loff: the directory of the archive that was asked
 * about last is kept as (local header offset, compressed size) pairs in offset order (16 bytes an entry, archives of up to a
 * million entries; another archive replaces it).  Which archive: its size and a hash of its last 4 KiB, as everywhere in this
 * file.  The stream is the calling thread's own; the lock covers the pairs only. */
static pthread_mutex_t g_hint_mu = PTHREAD_MUTEX_INITIALIZER;
static struct {
    int64_t size, n;
    uint64_t crc4;
    int64_t *pairs;
} g_hint;
static int cmp_pairs(const void *a, const void *b) {
    const int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return x < y ? -1 : x > y;
}
static int64_t cd_csize_hint(mzhip_stream *arch, int64_t loff) {
    const int64_t pos = arch->vtbl->tell(arch);
    if (pos < 0 || arch->vtbl->seek(arch, 0, MZH_SEEK_END) != MZH_OK)
        return 0;
    const int64_t size = arch->vtbl->tell(arch);
    uint8_t t4[MZH_TAIL4K];
    const int64_t n4 = size < MZH_TAIL4K ? size : MZH_TAIL4K;
    int64_t found = 0;
    if (size >= 22 && read_at(arch, size - n4, t4, n4)) {
        const uint64_t crc4 = tail_hash(t4, (size_t)n4);
        int have = 0;
        pthread_mutex_lock(&g_hint_mu);
        have = g_hint.pairs && g_hint.size == size && g_hint.crc4 == crc4;
        pthread_mutex_unlock(&g_hint_mu);
        if (!have) { /* read the directory (outside the lock: this thread's stream), then put it in place */
            uint64_t from = size > (1 << 18) ? (uint64_t)size - (1 << 18) : 0;
            uint8_t *tail = NULL;
            int64_t n = 0, *table = NULL, *pairs = NULL;
            int ok = 1;
            for (int pass = 0; ok && pass < 4; pass++) {
                const uint64_t len = (uint64_t)size - from;
                free(tail);
                tail = len <= ((uint64_t)256 << 20) ? (uint8_t *)malloc((size_t)len) : NULL;
                if (!tail || !read_at(arch, (int64_t)from, tail, (int64_t)len)) {
                    ok = 0;
                    break;
                }
                uint64_t need = 0;
                n = mzhip_zip_index_tail(tail, from, (uint64_t)size, NULL, 0, &need);
                if (n != MZHIP_INDEX_NEED_MORE)
                    break;
                if (need >= from)
                    ok = 0;
                from = need;
            }
            if (ok && n > 0 && n <= (1 << 20) && (table = (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t))) != NULL &&
                mzhip_zip_index_tail(tail, from, (uint64_t)size, table, n, NULL) == n && (pairs = (int64_t *)malloc((size_t)n * 2 * sizeof(int64_t))) != NULL) {
                for (int64_t k = 0; k < n; k++) {
                    pairs[2 * k] = table[8 * k + 5];
                    pairs[2 * k + 1] = table[8 * k + 3];
                }
                qsort(pairs, (size_t)n, 2 * sizeof(int64_t), cmp_pairs);
                pthread_mutex_lock(&g_hint_mu);
                free(g_hint.pairs);
                g_hint.pairs = pairs;
                g_hint.n = n;
                g_hint.size = size;
                g_hint.crc4 = crc4;
                pthread_mutex_unlock(&g_hint_mu);
                pairs = NULL;
                have = 1;
            }
            free(pairs);
            free(table);
            free(tail);
        }
        if (have) {
            pthread_mutex_lock(&g_hint_mu);
            if (g_hint.pairs && g_hint.size == size && g_hint.crc4 == crc4) {
                int64_t lo = 0, hi = g_hint.n - 1;
                while (lo <= hi) {
                    const int64_t mid = (lo + hi) / 2, v = g_hint.pairs[2 * mid];
                    if (v == loff) {
                        found = g_hint.pairs[2 * mid + 1];
                        break;
                    }
                    if (v < loff)
                        lo = mid + 1;
                    else
                        hi = mid - 1;
                }
            }
            pthread_mutex_unlock(&g_hint_mu);
        }
    }
    (void)arch->vtbl->seek(arch, pos, MZH_SEEK_SET);
    return found > 0 ? found : 0;
}

/* How many compressed bytes the entry holds whose payload starts at payload_off of the archive under codec_base, read off the
 * local header in front of it (appnote.txt 4.3.7: signature, flags at +6, compressed size at +18, name and extra lengths at
 * +26 / +28; the ZIP64 extra field when the size says 0xFFFFFFFF) -- or, when the header leaves the sizes to a data descriptor, off
 * the central directory (cd_csize_hint).  0 = not to be had: no header where one should be, a stream that cannot seek.  The codec stream is not told the size (mz_zip.c:1815-1830 sets
 * MZ_STREAM_PROP_TOTAL_IN_MAX for raw, stored and encrypted entries only); the READ shim uses this as a HINT for when to ask the
 * device -- once, with all of the entry, instead of at 32, 64, 128 KiB from the first byte each time -- never for what it reads
 * or returns: a wrong hint costs an attempt, nothing else.  The stream's position is restored. */
int64_t mzhip_lfh_csize_hint(mzhip_stream *codec_base, int64_t payload_off) {
    if (!codec_base || !codec_base->vtbl || payload_off < 30)
        return 0;
    mzhip_stream *arch = codec_base->base ? codec_base->base : codec_base;
    if (!arch->vtbl || !arch->vtbl->seek || !arch->vtbl->tell || !arch->vtbl->read || !arch->vtbl->is_open ||
        arch->vtbl->is_open(arch) != MZH_OK)
        return 0;
    const int64_t pos = arch->vtbl->tell(arch);
    if (pos < 0)
        return 0;
    uint8_t buf[2048];
    const int64_t back = payload_off < (int64_t)sizeof(buf) ? payload_off : (int64_t)sizeof(buf);
    const int got = read_at(arch, payload_off - back, buf, back);
    if (arch->vtbl->seek(arch, pos, MZH_SEEK_SET) != MZH_OK || !got)
        return 0;
    for (int64_t i = back - 30; i >= 0; i--) {
        if (buf[i] != 0x50 || buf[i + 1] != 0x4B || buf[i + 2] != 0x03 || buf[i + 3] != 0x04)
            continue;
        const int64_t n = (int64_t)buf[i + 26] | ((int64_t)buf[i + 27] << 8), m = (int64_t)buf[i + 28] | ((int64_t)buf[i + 29] << 8);
        if (i + 30 + n + m != back)
            continue; /* (these four bytes inside a name or an extra field) */
        const uint32_t flags = (uint32_t)buf[i + 6] | ((uint32_t)buf[i + 7] << 8);
        if (flags & 8u) /* sizes follow the payload (the reference's own writer does that): the central directory has them */
            return cd_csize_hint(arch, payload_off - back + i);
        const uint32_t c32 = (uint32_t)buf[i + 18] | ((uint32_t)buf[i + 19] << 8) | ((uint32_t)buf[i + 20] << 16) | ((uint32_t)buf[i + 21] << 24);
        const uint32_t u32 = (uint32_t)buf[i + 22] | ((uint32_t)buf[i + 23] << 8) | ((uint32_t)buf[i + 24] << 16) | ((uint32_t)buf[i + 25] << 24);
        if (c32 != 0xFFFFFFFFu)
            return (int64_t)c32;
        const uint8_t *x = buf + i + 30 + n, *xe = x + m;
        while (xe - x >= 4) {
            const uint32_t id = (uint32_t)x[0] | ((uint32_t)x[1] << 8), len = (uint32_t)x[2] | ((uint32_t)x[3] << 8);
            if ((int64_t)len > xe - x - 4)
                return 0;
            if (id == 1u) {
                const uint32_t skip = u32 == 0xFFFFFFFFu ? 8u : 0u;
                if (len < skip + 8u)
                    return 0;
                uint64_t c = 0;
                for (int k = 7; k >= 0; k--)
                    c = (c << 8) | x[4 + skip + k];
                return c <= (uint64_t)INT64_MAX ? (int64_t)c : 0;
            }
            x += 4 + len;
        }
        return 0;
    }
    return 0;
}

void mzhip_autoprime(mzhip_stream *codec_base, int64_t payload_off) {
    const char *env = getenv("MZHIP_AUTOPRIME");
    if (env && env[0] == '0')
        return;
    if (!codec_base || !codec_base->vtbl)
        return;
    mzhip_stream *arch = codec_base->base ? codec_base->base : codec_base;
    if (!arch->vtbl || !arch->vtbl->seek || !arch->vtbl->tell || !arch->vtbl->read || !arch->vtbl->is_open ||
        arch->vtbl->is_open(arch) != MZH_OK)
        return;
    /* "<n>": MiB; "<n>k": KiB (tests) */
    char *endp = NULL;
    int64_t limit = env && *env ? strtoll(env, &endp, 10) : 0;
    const int kib = endp && (*endp == 'k' || *endp == 'K');
    limit = (limit >= 1 ? (limit < (1 << 20) ? limit : (1 << 20)) : MZH_AUTOPRIME_DEFAULT_MIB) << (kib ? 10 : 20);
    const uint64_t budget = 4 * (uint64_t)limit;
    /* where the stream stands; how long the archive is (one seek to where the tail starts, or to the end of a shorter one); which
     * image it is: its size and a hash of its last 4 KiB (the end record and the central directory's end).  The stream is this
     * thread's own: none of this is done under the lock the readers of other threads wait for. */
    const int64_t pos = arch->vtbl->tell(arch);
    int64_t size = -1;
    if (pos >= 0) {
        if (arch->vtbl->seek(arch, -(int64_t)MZH_TAIL4K, MZH_SEEK_END) == MZH_OK) {
            const int64_t at = arch->vtbl->tell(arch);
            size = at >= 0 ? at + MZH_TAIL4K : -1;
        } else if (arch->vtbl->seek(arch, 0, MZH_SEEK_END) == MZH_OK) {
            size = arch->vtbl->tell(arch);
            if (size >= MZH_TAIL4K || (size > 0 && arch->vtbl->seek(arch, 0, MZH_SEEK_SET) != MZH_OK))
                size = -1; /* (a stream that cannot do the first seek but is that long: not one to read through) */
        }
    }
    if (pos < 0)
        return;
    uint8_t t4[MZH_TAIL4K];
    const int64_t n4 = size < MZH_TAIL4K ? size : MZH_TAIL4K;
    if (size < 22 || !read_all(arch, t4, n4)) { /* (the stream stands where the tail starts) */
        arch->vtbl->seek(arch, pos, MZH_SEEK_SET);
        return;
    }
    const uint64_t crc4 = tail_hash(t4, (size_t)n4);
    pthread_mutex_lock(&g_mu);
    g_tick++;
    {
        const uint64_t clears = mzhip_prime_clears();
        if (clears != g_clears_seen) { /* somebody cleared the cache: it holds nothing of ours any more */
            g_clears_seen = clears;
            forget_everything();
        }
    }
    {
        uint8_t *buf = NULL;
        int64_t *table = NULL;
        int dealt_with = 0;
        roll *R = NULL;
        for (int i = 0; i < MZH_ROLLS; i++)
            if (g_rolls[i] && g_rolls[i]->size == size && g_rolls[i]->crc4 == crc4)
                R = g_rolls[i];
        if (R)
            goto roll_on;
        {
            int32_t gens = 0, wins = 0;
            mzhip_prime_counts(&gens, &wins, NULL);
            if (g_cur_size >= 0 && !mzhip_prime_has((uint64_t)g_cur_size, g_cur_ident))
                forget_everything(); /* (our whole image left the cache some other way) */
            const int32_t foreign = gens - wins - (g_cur_size >= 0 ? 1 : 0) > 0;
            for (int i = 0; i < 16; i++)
                if (g_done[i].size == size && g_done[i].crc4 == crc4 && g_done[i].era == g_era && g_done[i].foreign == foreign)
                    dealt_with = 1;
            if (dealt_with)
                goto done;
            /* the image's full name: size + hash of its last 64 KiB */
            const int64_t tail = size < 65536 ? size : 65536;
            uint8_t *tb = (uint8_t *)malloc((size_t)tail);
            uint64_t tcrc = 0;
            int ok = 0;
            if (tb && read_at(arch, size - tail, tb, tail)) {
                tcrc = tail_hash(tb, (size_t)tail);
                ok = 1;
            }
            free(tb);
            if (ok && size > limit && mzhip_prime_has((uint64_t)size, 0)) {
                /* the application primed an archive of this length itself (mzhip_prime_file / _mem_begin): nothing to add */
            } else if (ok && size > limit) {
                /* too large to image: roll over it (beside whatever else the cache holds) */
                R = roll_new(arch, size, crc4, tcrc, budget);
                if (R) {
                    int slot = -1;
                    for (int i = 0; i < MZH_ROLLS; i++)
                        if (!g_rolls[i])
                            slot = i;
                    for (int i = 0; slot < 0 && i < MZH_ROLLS; i++) /* the archive nobody has asked about for longest, if nobody is reading it now */
                        if (g_rolls[i]->busy == 0 && (slot < 0 || g_rolls[i]->stamp < g_rolls[slot]->stamp))
                            slot = i;
                    if (slot < 0) {
                        roll_free(R);
                        R = NULL;
                    } else {
                        roll_free(g_rolls[slot]);
                        g_rolls[slot] = R;
                        (void)__atomic_add_fetch(&g_autoprimed, 1, __ATOMIC_RELAXED);
                    }
                }
                if (R)
                    goto roll_on;
            } else if (ok && foreign && g_cur_size < 0) {
                /* the cache holds generations this file did not make: the application primes for itself (mzhip_prime_file ...);
                 * nothing is added under its feet -- entries it did not prime take the per-entry path */
            } else if (ok && !(size == g_cur_size && tcrc == g_cur_crc)) { /* (the cache holds it already: nothing to do) */
                int known = -1;
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
                    if (buf && read_at(arch, 0, buf, size)) {
                        /* worth it?  entries a codec stream would be opened for, and what they decode to */
                        const int64_t n = mzhip_zip_index_mem(buf, (uint64_t)size, NULL, 0);
                        if (n >= MZH_AUTOPRIME_MIN_ENTRIES && n <= (1 << 26) && (table = (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t))) != NULL &&
                            mzhip_zip_index_mem(buf, (uint64_t)size, table, n) == n) {
                            int64_t cnt = 0;
                            uint64_t usum = 0; /* unsigned, over the rows the prime takes, cut off at the bound: a crafted directory cannot wrap it */
                            for (int64_t i = 0; i < n && usum <= budget; i++) {
                                const int64_t *t = table + 8 * i;
                                if (row_codec(t) && t[7] >= 0) {
                                    cnt++;
                                    usum += (uint64_t)t[4];
                                }
                            }
                            if (cnt >= MZH_AUTOPRIME_MIN_ENTRIES && usum <= budget) {
                                if (g_cur_size >= 0)
                                    mzhip_prime_drop((uint64_t)g_cur_size, g_cur_ident); /* one whole image's worth of cache at a time -- ours, by name */
                                g_cur_size = -1;
                                g_era++;
                                if (mzhip_prime_mem(buf, (uint64_t)size) > 0) {
                                    g_cur_size = size;
                                    g_cur_crc = tcrc;
                                    g_cur_ident = fnv1a64(buf + table[6], (uint64_t)size - (uint64_t)table[6], FNV0); /* (prime_prepare's) */
                                    (void)__atomic_add_fetch(&g_autoprimed, 1, __ATOMIC_RELAXED);
                                }
                            }
                        }
                    }
                }
            }
            {
                int32_t g2 = 0, w2 = 0;
                mzhip_prime_counts(&g2, &w2, NULL);
                g_done[g_done_next % 16].size = size;
                g_done[g_done_next % 16].crc4 = crc4;
                g_done[g_done_next % 16].era = g_era;
                g_done[g_done_next % 16].foreign = g2 - w2 - (g_cur_size >= 0 ? 1 : 0) > 0;
                g_done_next++;
            }
            goto done;
        }
    roll_on:
        R->stamp = g_tick;
        /* Every reader that is at work wants the window it is in and the one behind it.  The budget (4 x the limit) holds that
         * for eight readers; more of them -- sixteen threads over one archive -- used to evict each other's windows (3 - 6 GiB/s,
         * thousands of look-ups on the per-entry path).  The budget grows with the readers seen lately: 2 windows each and 2 to
         * spare, never past four times what was asked for. */
        uint64_t budget_now = budget;
        {
            static struct {
                pthread_t t;
                uint64_t tick;
                int used;
            } seen[64];
            const pthread_t me = pthread_self();
            int mine = -1, spare = -1, readers = 0;
            for (int i = 0; i < 64; i++) {
                if (seen[i].used && pthread_equal(seen[i].t, me))
                    mine = i;
                else if (!seen[i].used) {
                    if (spare < 0 || seen[spare].used)
                        spare = i;
                } else if (spare < 0 || (seen[spare].used && seen[i].tick < seen[spare].tick))
                    spare = i; /* (no free slot so far: the one that has not been seen for longest) */
            }
            if (mine < 0)
                mine = spare;
            seen[mine].t = me;
            seen[mine].tick = g_tick;
            seen[mine].used = 1;
            for (int i = 0; i < 64; i++)
                if (seen[i].used && g_tick - seen[i].tick <= 4096)
                    readers++;
            const uint64_t want = (uint64_t)(2 * readers + 2) * R->wmax;
            if (want > budget_now)
                budget_now = want < 4 * budget ? want : 4 * budget;
        }
        if (payload_off >= 0) {
            /* the window the entry at hand lies in; then the one behind it */
            int32_t lo = 0, hi = R->nwin;
            while (lo < hi) {
                const int32_t mid = (lo + hi) / 2;
                if (R->win[mid].lo <= (uint64_t)payload_off)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const int32_t w = lo - 1;
            if (w >= 0 && (uint64_t)payload_off < R->win[w].hi) {
                const int first = R->win[w].state != W_LIVE || R->win[w].hits % 256 == 0; /* (a look-ahead that found no room is tried again) */
                roll_ensure(R, w, arch, budget_now, 0);
                R->win[w].stamp = g_tick;
                R->win[w].hits++;
                if (first && w + 1 < R->nwin)
                    roll_ensure(R, w + 1, arch, budget_now, 1);
            }
        }
    done:
        free(table);
        free(buf);
    }
    pthread_mutex_unlock(&g_mu);
    arch->vtbl->seek(arch, pos, MZH_SEEK_SET);
}
