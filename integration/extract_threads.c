/* extract_threads.c -- the drop-in with a pool of reader threads ("Many readers").
 *
 * What an application that extracts a large archive with T workers does on the reference, unchanged: every thread owns
 * one mz_zip_reader over the same file (mz_zip_reader_create / _open_file, mz_zip_rw.c:160-260), walks its share of
 * the entries (mz_zip_goto_entry) and reads each one through mz_zip_entry_read_open / _read / _close in the reader's
 * 65 535-byte buffer (mz_zip_rw.c:55), so that mz_zip.c:2116-2128 verifies every entry's CRC-32 against the central
 * directory.  The only line that is not the reference's: mzhip_prime_file() in front (or MZHIP_AUTOPRIME in the
 * environment), after which the codec streams and mz_crypt_crc32_update behind those calls answer from the batch
 * decode -- pipelined H2D / kernels / D2H on the device while nothing but memcpy is left for the threads.
 * Linked into integration/_build/libmzhipdrop.so (reference zip layer unmodified + HIP codecs); bench.py's
 * `legs.vtbl_end_to_end_T` and tests/test_gpu_prime.py call it.  The entry table comes from the product's own bulk
 * central-directory indexer (mzhip_zip_index_mem) over a read-only mapping of the file. */
#include <fcntl.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "mz.h"
#include "mz_strm.h"
#include "mz_zip.h"
#include "mz_zip_rw.h"

#include "../include/mzhip.h"

typedef struct {
    const char *path;
    const int64_t *table; /* 8 x int64 per entry: method, flag, crc, csize, usize, .., cd position, payload offset */
    int64_t first, count;
    int64_t ok, bytes;
    int32_t err;
} xt_job;

static void *xt_run(void *arg) {
    xt_job *j = (xt_job *)arg;
    void *reader = mz_zip_reader_create();
    void *zip = NULL;
    uint8_t *buf = (uint8_t *)malloc(UINT16_MAX);
    if (!reader || !buf || mz_zip_reader_open_file(reader, j->path) != MZ_OK) {
        j->err = MZ_OPEN_ERROR;
        free(buf);
        if (reader) mz_zip_reader_delete(&reader);
        return NULL;
    }
    mz_zip_reader_get_zip_handle(reader, &zip);
    for (int64_t i = j->first; i < j->first + j->count; i++) {
        int32_t err = mz_zip_goto_entry(zip, j->table[i * 8 + 6]);
        int64_t total = 0;
        if (err == MZ_OK) err = mz_zip_entry_read_open(zip, 0, NULL);
        if (err == MZ_OK) {
            for (;;) {
                const int32_t rd = mz_zip_entry_read(zip, buf, UINT16_MAX);
                if (rd < 0) err = rd;
                if (rd <= 0) break;
                total += rd;
            }
            const int32_t cerr = mz_zip_entry_close(zip); /* MZ_CRC_ERROR when the bytes are not the archive's */
            if (err == MZ_OK) err = cerr;
        }
        if (err == MZ_OK && total == j->table[i * 8 + 4]) {
            j->ok++;
            j->bytes += total;
        } else if (j->err == MZ_OK) {
            j->err = err != MZ_OK ? err : MZ_DATA_ERROR;
        }
    }
    mz_zip_reader_close(reader);
    mz_zip_reader_delete(&reader);
    free(buf);
    return NULL;
}

static double xt_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* Extract (decode + verify, bytes discarded) every entry of `path` with `nthreads` reader threads; prime != 0 runs
 * mzhip_prime_file(path) first.  Returns the seconds of the whole thing (prime + index + threads), or a negative MZ_*
 * code; *entries / *bytes = what was read and verified, *prime_s = the share of mzhip_prime_file, *first_err = the
 * first entry error of any thread. */
__attribute__((visibility("default"))) double mzdrop_extract_all(const char *path, int32_t nthreads, int32_t prime,
                                                                  int64_t *entries, int64_t *bytes, double *prime_s,
                                                                  int32_t *first_err) {
    const double t0 = xt_now();
    double tp = 0.0;
    if (prime) {
        const int64_t pr = mzhip_prime_file(path);
        if (pr < 0) return (double)pr;
        tp = xt_now() - t0;
    }
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return (double)MZ_OPEN_ERROR;
    struct stat sb;
    if (fstat(fd, &sb) != 0 || sb.st_size <= 0) {
        close(fd);
        return (double)MZ_OPEN_ERROR;
    }
    const uint8_t *img = (const uint8_t *)mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (img == MAP_FAILED) return (double)MZ_MEM_ERROR;
    int64_t n = mzhip_zip_index_mem(img, (uint64_t)sb.st_size, NULL, 0);
    int64_t *table = n > 0 ? (int64_t *)malloc((size_t)n * 8 * sizeof(int64_t)) : NULL;
    if (table) n = mzhip_zip_index_mem(img, (uint64_t)sb.st_size, table, n);
    munmap((void *)img, (size_t)sb.st_size);
    if (n <= 0 || !table) {
        free(table);
        return (double)(n < 0 ? n : MZ_FORMAT_ERROR);
    }
    if (nthreads < 1) nthreads = 1;
    if (nthreads > n) nthreads = (int32_t)n;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    xt_job *jobs = (xt_job *)calloc((size_t)nthreads, sizeof(xt_job));
    const int64_t per = n / nthreads, extra = n % nthreads;
    int64_t first = 0;
    for (int32_t t = 0; t < nthreads; t++) {
        jobs[t].path = path;
        jobs[t].table = table;
        jobs[t].first = first;
        jobs[t].count = per + (t < extra ? 1 : 0);
        first += jobs[t].count;
        if (nthreads == 1) xt_run(&jobs[t]);
        else pthread_create(&th[t], NULL, xt_run, &jobs[t]);
    }
    int64_t ok = 0, by = 0;
    int32_t err = MZ_OK;
    for (int32_t t = 0; t < nthreads; t++) {
        if (nthreads > 1) pthread_join(th[t], NULL);
        ok += jobs[t].ok;
        by += jobs[t].bytes;
        if (err == MZ_OK) err = jobs[t].err;
    }
    free(th);
    free(jobs);
    free(table);
    if (entries) *entries = ok;
    if (bytes) *bytes = by;
    if (prime_s) *prime_s = tp;
    if (first_err) *first_err = err;
    return xt_now() - t0;
}
