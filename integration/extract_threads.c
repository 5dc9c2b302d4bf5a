/* extract_threads.c -- the drop-in with a pool of reader threads ("Many readers").
 *
 * What an application that extracts a large archive with T workers does on the reference, unchanged: every thread owns
 * one mz_zip_reader over the same archive (mz_zip_reader_create / _open on a memory stream over one shared read-only
 * mapping of the file, mz_zip_rw.c:160-260), walks its share of
 * the entries (mz_zip_goto_entry) and reads each one through mz_zip_entry_read_open / _read / _close in the reader's
 * 65 535-byte buffer (mz_zip_rw.c:55), so that mz_zip.c:2116-2128 verifies every entry's CRC-32 against the central
 * directory.  The only line that is not the reference's: mzhip_prime_mem() over the same mapping in front (or
 * mzhip_prime_file() / MZHIP_AUTOPRIME in the environment), after which the codec streams and mz_crypt_crc32_update behind those calls answer from the batch
 * decode -- pipelined H2D / kernels / D2H on the device while nothing but memcpy is left for the threads.
 * Linked into integration/_build/libmzhipdrop.so (reference zip layer unmodified + HIP codecs); bench.py's
 * `legs.vtbl_end_to_end_T` and tests/test_gpu_prime.py call it.  The entry table comes from the product's own bulk
 * central-directory indexer (mzhip_zip_index_mem) over a read-only mapping of the file. */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "mz.h"
#include "mz_strm.h"
#include "mz_strm_mem.h"
#include "mz_zip.h"
#include "mz_zip_rw.h"

#include "../include/mzhip.h"

typedef struct {
    const char *path;
    const uint8_t *img; /* the archive, mapped read-only and shared by every thread */
    int64_t img_len;
    const int64_t *table; /* 8 x int64 per entry: method, flag, crc, csize, usize, .., cd position, payload offset */
    int64_t n;               /* entries of the archive */
    volatile int64_t *next;  /* the next entry nobody has taken yet: the threads take blocks of XT_BLOCK from the front, so */
                             /* that all of them read what the decode pipeline delivered first (progressive prime) */
    int64_t ok, bytes;
    int32_t err;
    int32_t started; /* the job runs on a thread of its own (to be joined) */
    double t_goto, t_open, t_read, t_close; /* MZDROP_TRACE: where a thread's time goes */
} xt_job;

static double xt_now(void);
#define XT_BLOCK 16

static void *xt_run(void *arg) {
    xt_job *j = (xt_job *)arg;
    void *reader = mz_zip_reader_create();
    void *zip = NULL;
    uint8_t *buf = (uint8_t *)malloc(UINT16_MAX);
    /* The reader sits on a memory stream over the shared mapping (mz_zip_reader_open, mz_zip_rw.c:160; mz_strm_mem.c),
     * not on mz_zip_reader_open_file: that one puts the split-disk stream under the reader, which closes and re-opens
     * the file twice per entry (mz_zip.c:2358 and :1759 set MZ_STREAM_PROP_DISK_NUMBER to -1 and to the entry's disk,
     * mz_strm_split.c:151-171) -- 2.7 us per call for one thread, and ~1 ms per call when 256 threads do it to the same
     * path (measured: goto + read_open took 150 ms per thread however few entries a thread had). */
    void *mem = mz_stream_mem_create();
    if (mem) mz_stream_mem_set_buffer(mem, (void *)j->img, (int32_t)j->img_len);
    if (!reader || !buf || !mem || mz_stream_open(mem, NULL, MZ_OPEN_MODE_READ) != MZ_OK || mz_zip_reader_open(reader, mem) != MZ_OK) {
        j->err = MZ_OPEN_ERROR;
        free(buf);
        if (reader) mz_zip_reader_delete(&reader);
        if (mem) mz_stream_mem_delete(&mem);
        return NULL;
    }
    mz_zip_reader_get_zip_handle(reader, &zip);
    const int trace = getenv("MZDROP_TRACE") != NULL;
    for (;;) {
      const int64_t i0 = __atomic_fetch_add(j->next, XT_BLOCK, __ATOMIC_RELAXED);
      if (i0 >= j->n) break;
      const int64_t i1 = i0 + XT_BLOCK < j->n ? i0 + XT_BLOCK : j->n;
      for (int64_t i = i0; i < i1; i++) {
        double a = trace ? xt_now() : 0.0, b;
        int32_t err = mz_zip_goto_entry(zip, j->table[i * 8 + 6]);
        int64_t total = 0;
        if (trace) { b = xt_now(); j->t_goto += b - a; a = b; }
        if (err == MZ_OK) err = mz_zip_entry_read_open(zip, 0, NULL);
        if (trace) { b = xt_now(); j->t_open += b - a; a = b; }
        if (err == MZ_OK) {
            for (;;) {
                const int32_t rd = mz_zip_entry_read(zip, buf, UINT16_MAX);
                if (rd < 0) err = rd;
                if (rd <= 0) break;
                total += rd;
            }
            if (trace) { b = xt_now(); j->t_read += b - a; a = b; }
            const int32_t cerr = mz_zip_entry_close(zip); /* MZ_CRC_ERROR when the bytes are not the archive's */
            if (err == MZ_OK) err = cerr;
            if (trace) { b = xt_now(); j->t_close += b - a; a = b; }
        }
        if (err == MZ_OK && total == j->table[i * 8 + 4]) {
            j->ok++;
            j->bytes += total;
        } else if (j->err == MZ_OK) {
            j->err = err != MZ_OK ? err : MZ_DATA_ERROR;
        }
      }
    }
    mz_zip_reader_close(reader);
    mz_zip_reader_delete(&reader);
    mz_stream_mem_delete(&mem);
    free(buf);
    return NULL;
}

typedef struct {
    void *img;
    size_t len;
    int64_t *table;
} xt_release;

static void *xt_release_run(void *arg) {
    xt_release *r = (xt_release *)arg;
    munmap(r->img, r->len);
    free(r->table);
    free(r);
    return NULL;
}

/* one release may be in flight; the next call (or mzdrop_quiesce) joins it before it starts: two address-space operations
 * of this size at once just queue up behind the process's mapping lock */
static pthread_mutex_t xt_rel_mu = PTHREAD_MUTEX_INITIALIZER;
static pthread_t xt_rel_thread;
static int xt_rel_pending;

__attribute__((visibility("default"))) void mzdrop_quiesce(void) {
    pthread_mutex_lock(&xt_rel_mu);
    if (xt_rel_pending) {
        pthread_join(xt_rel_thread, NULL);
        xt_rel_pending = 0;
    }
    pthread_mutex_unlock(&xt_rel_mu);
}

static double xt_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* Extract (decode + verify, bytes discarded) every entry of `path` with `nthreads` reader threads.  prime: 0 = none (every
 * entry through the per-entry path), 1 = mzhip_prime_mem() first, readers afterwards, 2 = mzhip_prime_mem_begin(): the
 * readers start at once and are served as the decode pipeline delivers, front to back.  Returns the seconds of the whole
 * thing (map + prime + index + threads), or a negative MZ_* code; *entries / *bytes = what was read and verified,
 * *prime_s = the time the calling thread spent in the prime calls (prime 2: begin + the wait at the end, the decode itself
 * runs under the readers), *first_err = the first entry error of any thread. */
__attribute__((visibility("default"))) double mzdrop_extract_all(const char *path, int32_t nthreads, int32_t prime,
                                                                  int64_t *entries, int64_t *bytes, double *prime_s,
                                                                  int32_t *first_err) {
    mzdrop_quiesce(); /* (the previous call's mapping is gone before this one's clock starts) */
    const double t0 = xt_now();
    double tp = 0.0;
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return (double)MZ_OPEN_ERROR;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) {
        close(fd);
        return (double)MZ_OPEN_ERROR;
    }
    const uint8_t *img = (const uint8_t *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0); /* (faulted in here, once: 256 threads taking page faults on one mapping queue up behind its lock) */
    close(fd);
    if (img == MAP_FAILED) return (double)MZ_MEM_ERROR;
    if (prime) { /* the same mapping feeds the batch decode and, afterwards, every reader thread: the file is read once */
        const double a = xt_now();
        const int64_t pr = prime == 2 ? mzhip_prime_mem_begin(img, (uint64_t)sb.st_size) : mzhip_prime_mem(img, (uint64_t)sb.st_size);
        if (pr < 0) {
            munmap((void *)img, (size_t)sb.st_size);
            return (double)pr;
        }
        tp = xt_now() - a;
    }
    int64_t n = mzhip_zip_index_mem(img, (uint64_t)sb.st_size, NULL, 0);
    int64_t *table = n > 0 ? (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t)) : NULL;
    if (table) n = mzhip_zip_index_mem(img, (uint64_t)sb.st_size, table, n);
    if (n <= 0 || !table || sb.st_size > INT32_MAX) { /* (mz_stream_mem takes a 32-bit length: archives of 2 GiB and more need the file reader) */
        if (prime == 2) (void)mzhip_prime_wait(); /* (the worker reads the mapping) */
        munmap((void *)img, (size_t)sb.st_size);
        free(table);
        return (double)(n < 0 ? n : MZ_FORMAT_ERROR);
    }
    const double t_idx = xt_now();
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = (int32_t)n;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    xt_job *jobs = (xt_job *)calloc((size_t)nthreads, sizeof(xt_job));
    volatile int64_t next = 0;
    for (int32_t t = 0; t < nthreads; t++) {
        jobs[t].path = path;
        jobs[t].img = img;
        jobs[t].img_len = (int64_t)sb.st_size;
        jobs[t].table = table;
        jobs[t].n = n;
        jobs[t].next = &next;
        jobs[t].started = 0;
        if (nthreads > 1 && pthread_create(&th[t], NULL, xt_run, &jobs[t]) == 0) jobs[t].started = 1;
        else xt_run(&jobs[t]); /* one thread, or a thread that could not be created: the caller's thread does the job */
    }
    int64_t ok = 0, by = 0;
    int32_t err = MZ_OK;
    double tg = 0, to = 0, tr = 0, tc = 0;
    for (int32_t t = 0; t < nthreads; t++) {
        if (jobs[t].started) pthread_join(th[t], NULL);
        ok += jobs[t].ok;
        by += jobs[t].bytes;
        if (err == MZ_OK) err = jobs[t].err;
        tg += jobs[t].t_goto; to += jobs[t].t_open; tr += jobs[t].t_read; tc += jobs[t].t_close;
    }
    const double t_thr = xt_now();
    if (prime == 2) { /* every entry has been served, so the pipeline is through; what is left is the STORE index and the join */
        const int64_t pr = mzhip_prime_wait();
        tp += xt_now() - t_thr;
        if (pr < 0 && err == MZ_OK) err = (int32_t)pr;
    }
    if (getenv("MZDROP_TRACE"))
        fprintf(stderr, "[mzdrop] started %.2f ms, readers started %.2f ms, readers done %.2f ms (CLOCK_MONOTONIC)\n", t0 * 1e3, t_idx * 1e3, t_thr * 1e3);
    if (getenv("MZDROP_TRACE"))
        fprintf(stderr, "[mzdrop] prime %.1f ms, map + index %.1f ms, threads %.1f ms\n", tp * 1e3, (t_idx - t0 - tp) * 1e3, (t_thr - t_idx) * 1e3);
    if (getenv("MZDROP_TRACE"))
        fprintf(stderr, "[mzdrop] %d threads: per thread goto %.1f ms, read_open %.1f ms, read %.1f ms, close %.1f ms\n", nthreads,
                tg / nthreads * 1e3, to / nthreads * 1e3, tr / nthreads * 1e3, tc / nthreads * 1e3);
    /* every entry is read and verified: what is left is giving memory back.  Unmapping 75 000 populated pages of a 300 MB
     * archive costs the kernel ~12 ms (profiles/r3/threads_trace.log) -- a detached thread does it, the caller has its answer */
    /* ... but ONLY when MZDROP_ASYNC_RELEASE is set: the CPU baseline it is compared with (mz_driver.c drv_zip_read_all)
     * unmaps inside its clock, so by default this function does too (the two sides are timed the same way) */
    xt_release *rel = getenv("MZDROP_ASYNC_RELEASE") ? (xt_release *)malloc(sizeof(xt_release)) : NULL;
    if (rel) {
        rel->img = (void *)img;
        rel->len = (size_t)sb.st_size;
        rel->table = table;
    }
    pthread_mutex_lock(&xt_rel_mu);
    const int started = rel && !xt_rel_pending && pthread_create(&xt_rel_thread, NULL, xt_release_run, rel) == 0;
    if (started) xt_rel_pending = 1;
    pthread_mutex_unlock(&xt_rel_mu);
    if (!started) {
        free(rel);
        munmap((void *)img, (size_t)sb.st_size);
        free(table);
    }
    free(th);
    free(jobs);
    if (entries) *entries = ok;
    if (bytes) *bytes = by;
    if (prime_s) *prime_s = tp;
    if (first_err) *first_err = err;
    if (getenv("MZDROP_TRACE")) fprintf(stderr, "[mzdrop] returning %.2f ms after the start\n", (xt_now() - t0) * 1e3);
    return xt_now() - t0;
}

/* ---- the re-linked application on an archive of any size: nothing but the reference's calls ----
 * `minizip -x` without the file writes: every thread owns one mz_zip_reader opened with mz_zip_reader_open_file (the split,
 * buffered and OS streams of mz_zip_rw.c:262-312 under it -- no mapping, no 2 GiB bound), walks its contiguous share of the
 * entries and reads each through mz_zip_entry_read_open / _read (65 535-byte calls, mz_zip_rw.c:55) / _close, so that
 * mz_zip.c:2116-2128 verifies every CRC-32.  No mzhip_prime_* call: whatever batching happens, happens behind the codec
 * streams' first read() (shim_autoprime.c).  One thread walks with mz_zip_goto_first_entry / _next_entry exactly as minizip.c
 * does; with several, each thread jumps to its first entry (mz_zip_goto_entry on a central-directory position from
 * mzhip_zip_index_tail over the file's tail) and walks on from there.  Returns the seconds of the whole call. */
typedef struct {
    const char *path;
    int64_t cd_first; /* < 0: start at the first entry */
    int64_t count;
    int64_t ok, bytes;
    int32_t err, started;
    double t_goto, t_open, t_read, t_close; /* MZDROP_TRACE: where the thread's time goes */
} xf_job;

static void *xf_run(void *arg) {
    xf_job *j = (xf_job *)arg;
    void *reader = mz_zip_reader_create();
    uint8_t *buf = (uint8_t *)malloc(UINT16_MAX);
    void *zip = NULL;
    if (!reader || !buf || mz_zip_reader_open_file(reader, j->path) != MZ_OK) {
        j->err = MZ_OPEN_ERROR;
        free(buf);
        if (reader) mz_zip_reader_delete(&reader);
        return NULL;
    }
    mz_zip_reader_get_zip_handle(reader, &zip);
    const int trace = getenv("MZDROP_TRACE") != NULL;
    double a = trace ? xt_now() : 0.0, b;
    int32_t err = j->cd_first < 0 ? mz_zip_goto_first_entry(zip) : mz_zip_goto_entry(zip, j->cd_first);
    for (int64_t i = 0; err == MZ_OK && i < j->count; i++) {
        int64_t total = 0;
        mz_zip_file *fi = NULL;
        err = mz_zip_entry_get_info(zip, &fi);
        const int64_t want = err == MZ_OK ? fi->uncompressed_size : -1;
        if (trace) { b = xt_now(); j->t_goto += b - a; a = b; }
        if (err == MZ_OK) err = mz_zip_entry_read_open(zip, 0, NULL);
        if (trace) { b = xt_now(); j->t_open += b - a; a = b; }
        if (err == MZ_OK) {
            for (;;) {
                const int32_t rd = mz_zip_entry_read(zip, buf, UINT16_MAX);
                if (rd < 0) err = rd;
                if (rd <= 0) break;
                total += rd;
            }
            if (trace) { b = xt_now(); j->t_read += b - a; a = b; }
            const int32_t cerr = mz_zip_entry_close(zip); /* MZ_CRC_ERROR when the bytes are not the archive's */
            if (err == MZ_OK) err = cerr;
            if (trace) { b = xt_now(); j->t_close += b - a; a = b; }
        }
        if (err == MZ_OK && total == want) {
            j->ok++;
            j->bytes += total;
        } else if (j->err == MZ_OK) {
            j->err = err != MZ_OK ? err : MZ_DATA_ERROR;
        }
        if (i + 1 < j->count) {
            err = mz_zip_goto_next_entry(zip);
            if (err == MZ_END_OF_LIST) {
                err = MZ_OK;
                break;
            }
        }
    }
    if (err != MZ_OK && j->err == MZ_OK) j->err = err;
    mz_zip_reader_close(reader);
    mz_zip_reader_delete(&reader);
    free(buf);
    return NULL;
}

__attribute__((visibility("default"))) double mzdrop_extract_file(const char *path, int32_t nthreads, int64_t *entries, int64_t *bytes,
                                                                   int32_t *first_err) {
    const double t0 = xt_now();
    if (nthreads < 1) nthreads = 1;
    int64_t n = INT64_MAX, *table = NULL;
    if (nthreads > 1) { /* where the threads' shares start: the central directory, indexed from the file's tail */
        const int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd < 0 || fstat(fd, &sb) != 0 || sb.st_size < 22) {
            if (fd >= 0) close(fd);
            return (double)MZ_OPEN_ERROR;
        }
        uint64_t from = sb.st_size > (1 << 18) ? (uint64_t)sb.st_size - (1 << 18) : 0, need = 0;
        uint8_t *tail = NULL;
        n = MZHIP_INDEX_NEED_MORE;
        for (int pass = 0; pass < 4 && n == MZHIP_INDEX_NEED_MORE; pass++) {
            free(tail);
            const size_t len = (size_t)((uint64_t)sb.st_size - from);
            tail = (uint8_t *)malloc(len);
            if (!tail || pread(fd, tail, len, (off_t)from) != (ssize_t)len) {
                n = MZ_READ_ERROR;
                break;
            }
            n = mzhip_zip_index_tail(tail, from, (uint64_t)sb.st_size, NULL, 0, &need);
            if (n == MZHIP_INDEX_NEED_MORE) from = need;
        }
        if (n > 0 && (table = (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t))) != NULL)
            n = mzhip_zip_index_tail(tail, from, (uint64_t)sb.st_size, table, n, NULL);
        free(tail);
        close(fd);
        if (n <= 0 || !table) {
            free(table);
            return (double)(n < 0 ? n : MZ_FORMAT_ERROR);
        }
        if (nthreads > n) nthreads = (int32_t)n;
    }
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    xf_job *jobs = (xf_job *)calloc((size_t)nthreads, sizeof(xf_job));
    for (int32_t t = 0; t < nthreads; t++) {
        jobs[t].path = path;
        if (nthreads == 1) {
            jobs[t].cd_first = -1;
            jobs[t].count = INT64_MAX;
        } else {
            const int64_t lo = n * t / nthreads, hi = n * (t + 1) / nthreads;
            jobs[t].cd_first = table[lo * 8 + 6];
            jobs[t].count = hi - lo;
        }
        if (nthreads > 1 && pthread_create(&th[t], NULL, xf_run, &jobs[t]) == 0) jobs[t].started = 1;
        else xf_run(&jobs[t]);
    }
    int64_t ok = 0, by = 0;
    int32_t err = MZ_OK;
    for (int32_t t = 0; t < nthreads; t++) {
        if (jobs[t].started) pthread_join(th[t], NULL);
        ok += jobs[t].ok;
        by += jobs[t].bytes;
        if (err == MZ_OK) err = jobs[t].err;
        if (getenv("MZDROP_TRACE"))
            fprintf(stderr, "[mzdrop] thread %d: %lld entries: goto + info %.1f ms, read_open %.1f ms, read %.1f ms, close %.1f ms\n", t, (long long)jobs[t].ok,
                    jobs[t].t_goto * 1e3, jobs[t].t_open * 1e3, jobs[t].t_read * 1e3, jobs[t].t_close * 1e3);
    }
    free(table);
    free(th);
    free(jobs);
    if (entries) *entries = ok;
    if (bytes) *bytes = by;
    if (first_err) *first_err = err;
    return xt_now() - t0;
}
