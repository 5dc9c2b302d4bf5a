/* mz_driver.c -- thin C driver over minizip-ng's PUBLIC API (TEST INFRASTRUCTURE).
 *
 * Own code, no reference source copied: it only calls the reference's exported
 * functions (mz_strm.h, mz_strm_mem.h, mz_strm_zlib.h, mz_strm_lzma.h,
 * mz_crypt.h, mz_zip.h, mz_zip_rw.h).  It is compiled twice:
 *
 *   oracle/_ref/libmzref.so        driver + UNMODIFIED reference sources with
 *                                  zlib 1.2.11 / liblzma 5.2.5  -> the oracle and
 *                                  the CPU baseline ("kind": "reference");
 *   integration/_build/libmzhipdrop.so
 *                                  driver + the same unmodified reference
 *                                  mz_zip.c / mz_zip_rw.c / mz_strm*.c, but with
 *                                  mz_strm_zlib.o / mz_strm_lzma.o /
 *                                  mz_crypt_crc32_update replaced by the HIP
 *                                  backend (link-time substitution).
 *
 * Because both builds expose the same drv_* entry points, the parity tests
 * call the two libraries with identical arguments and compare every return
 * value, property and output byte.
 */
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
#include "mz_crypt.h"
#include "mz_strm.h"
#include "mz_strm_mem.h"
#include "mz_strm_zlib.h"
#include "mz_strm_lzma.h"
#include "mz_zip.h"
#include "mz_zip_rw.h"

#define DRV_EXPORT __attribute__((visibility("default")))

/* ---------------------------------------------------------------- crc32 */

DRV_EXPORT uint32_t drv_crc32_update(uint32_t value, const uint8_t *buf, int32_t size) {
    return mz_crypt_crc32_update(value, buf, size);
}

/* ------------------------------------------------------- codec streams */

static void *codec_create(int32_t method) {
    void *s = NULL;
    if (method == MZ_COMPRESS_METHOD_DEFLATE) {
        s = mz_stream_zlib_create();
    } else if (method == MZ_COMPRESS_METHOD_LZMA || method == MZ_COMPRESS_METHOD_XZ) {
        s = mz_stream_lzma_create();
        if (s)
            mz_stream_set_prop_int64(s, MZ_STREAM_PROP_COMPRESS_METHOD, method);
    }
    return s;
}

/* Decode `in` through the codec stream stacked on a memory stream, reading
 * `chunk` bytes per read() call exactly as mz_zip_entry_read's callers do
 * (mz_zip.c:2047).  Records every read() return value in rets[].
 * info[0]=TOTAL_IN info[1]=TOTAL_OUT info[2]=close() info[3]=error()
 * info[4]=base position after close  info[5]=open() */
DRV_EXPORT int32_t drv_stream_decode(int32_t method, const uint8_t *in, int32_t in_len, int64_t max_in,
                                     int64_t max_out, int32_t window_bits, uint8_t *out, int32_t out_cap,
                                     int32_t chunk, int32_t *rets, int32_t max_rets, int64_t *info) {
    void *mem = mz_stream_mem_create();
    void *codec = codec_create(method);
    int32_t n_rets = 0, produced = 0;
    memset(info, 0, 6 * sizeof(int64_t));
    if (!mem || !codec)
        return MZ_MEM_ERROR;
    mz_stream_mem_set_buffer(mem, (void *)(intptr_t)in, in_len);
    mz_stream_open(mem, NULL, MZ_OPEN_MODE_READ);
    if (max_in > 0)
        mz_stream_set_prop_int64(codec, MZ_STREAM_PROP_TOTAL_IN_MAX, max_in);
    if (max_out >= 0)
        mz_stream_set_prop_int64(codec, MZ_STREAM_PROP_TOTAL_OUT_MAX, max_out);
    if (window_bits != 0)
        mz_stream_set_prop_int64(codec, MZ_STREAM_PROP_COMPRESS_WINDOW, window_bits);
    mz_stream_set_base(codec, mem);
    info[5] = mz_stream_open(codec, NULL, MZ_OPEN_MODE_READ);
    if (info[5] == MZ_OK) {
        for (;;) {
            int32_t want = chunk;
            if (want > out_cap - produced)
                want = out_cap - produced;
            if (want <= 0)
                break;
            int32_t r = mz_stream_read(codec, out + produced, want);
            if (n_rets < max_rets)
                rets[n_rets] = r;
            n_rets++;
            if (r <= 0)
                break;
            produced += r;
        }
        info[2] = mz_stream_close(codec);
        info[3] = mz_stream_error(codec);
        mz_stream_get_prop_int64(codec, MZ_STREAM_PROP_TOTAL_IN, &info[0]);
        mz_stream_get_prop_int64(codec, MZ_STREAM_PROP_TOTAL_OUT, &info[1]);
    }
    info[4] = mz_stream_tell(mem);
    mz_stream_delete(&codec);
    mz_stream_mem_delete(&mem);
    return n_rets;
}

/* A stream that is deleted WITHOUT close() after `nreads` read() calls of `chunk` bytes: legal in the reference
 * (mz_strm_zlib.c:364-371 frees the struct; zlib's state leaks) and must not corrupt the heap in the drop-in
 * (the window-mode piece tables were freed twice).  Returns the bytes read. */
DRV_EXPORT int32_t drv_stream_delete_unclosed(int32_t method, const uint8_t *in, int32_t in_len, uint8_t *out,
                                              int32_t out_cap, int32_t chunk, int32_t nreads) {
    void *mem = mz_stream_mem_create();
    void *codec = codec_create(method);
    int32_t produced = 0;
    if (!mem || !codec)
        return MZ_MEM_ERROR;
    mz_stream_mem_set_buffer(mem, (void *)(intptr_t)in, in_len);
    mz_stream_open(mem, NULL, MZ_OPEN_MODE_READ);
    mz_stream_set_base(codec, mem);
    if (mz_stream_open(codec, NULL, MZ_OPEN_MODE_READ) == MZ_OK) {
        for (int32_t k = 0; k < nreads && produced < out_cap; k++) {
            int32_t want = chunk < out_cap - produced ? chunk : out_cap - produced;
            int32_t r = mz_stream_read(codec, out + produced, want);
            if (r <= 0)
                break;
            produced += r;
        }
    }
    mz_stream_delete(&codec); /* no close() */
    mz_stream_mem_delete(&mem);
    return produced;
}

/* Encode `in` through the codec stream into a growable memory stream using
 * `chunk`-byte write() calls (mz_zip.c:2062).  Returns compressed length or <0.
 * info[0]=TOTAL_IN info[1]=TOTAL_OUT info[2]=close() info[3]=error() info[5]=open() */
DRV_EXPORT int32_t drv_stream_encode(int32_t method, int32_t level, int32_t window_bits, const uint8_t *in,
                                     int32_t in_len, int32_t chunk, uint8_t *out, int32_t out_cap, int64_t *info) {
    void *mem = mz_stream_mem_create();
    void *codec = codec_create(method);
    int32_t pos = 0, ret = 0;
    memset(info, 0, 6 * sizeof(int64_t));
    if (!mem || !codec)
        return MZ_MEM_ERROR;
    mz_stream_mem_set_grow_size(mem, 1 << 20);
    mz_stream_open(mem, NULL, MZ_OPEN_MODE_CREATE);
    mz_stream_set_prop_int64(codec, MZ_STREAM_PROP_COMPRESS_LEVEL, level);
    if (window_bits != 0)
        mz_stream_set_prop_int64(codec, MZ_STREAM_PROP_COMPRESS_WINDOW, window_bits);
    mz_stream_set_base(codec, mem);
    info[5] = mz_stream_open(codec, NULL, MZ_OPEN_MODE_WRITE);
    if (info[5] == MZ_OK) {
        while (pos < in_len) {
            int32_t n = in_len - pos < chunk ? in_len - pos : chunk;
            int32_t w = mz_stream_write(codec, in + pos, n);
            if (w != n) {
                ret = w < 0 ? w : MZ_WRITE_ERROR;
                break;
            }
            pos += n;
        }
        info[2] = mz_stream_close(codec);
        info[3] = mz_stream_error(codec);
        mz_stream_get_prop_int64(codec, MZ_STREAM_PROP_TOTAL_IN, &info[0]);
        mz_stream_get_prop_int64(codec, MZ_STREAM_PROP_TOTAL_OUT, &info[1]);
    } else {
        ret = (int32_t)info[5];
    }
    if (ret == 0) {
        const void *buf = NULL;
        int32_t len = 0;
        mz_stream_mem_get_buffer(mem, &buf);
        mz_stream_mem_get_buffer_length(mem, &len);
        if (len > out_cap)
            ret = MZ_BUF_ERROR;
        else {
            memcpy(out, buf, (size_t)len);
            ret = len;
        }
    }
    mz_stream_delete(&codec);
    mz_stream_mem_delete(&mem);
    return ret;
}

/* ------------------------------------------------------------ archives */

/* Write an archive with the reference writer (mz_zip_writer_add_buffer,
 * mz_zip_rw.c:1546): entry i = blob[offs[i] .. offs[i]+lens[i]), name e/%06d. */
DRV_EXPORT int32_t drv_zip_write(const char *path, int32_t method, int32_t level, const uint8_t *blob,
                                 const int64_t *offs, const int32_t *lens, int32_t n) {
    void *w = mz_zip_writer_create();
    int32_t err;
    if (!w)
        return MZ_MEM_ERROR;
    mz_zip_writer_set_compress_method(w, (uint16_t)method);
    mz_zip_writer_set_compress_level(w, (int16_t)level);
    err = mz_zip_writer_open_file(w, path, 0, 0);
    for (int32_t i = 0; err == MZ_OK && i < n; i++) {
        char name[32];
        mz_zip_file fi;
        memset(&fi, 0, sizeof(fi));
        snprintf(name, sizeof(name), "e/%06d", i);
        fi.filename = name;
        fi.modified_date = 1700000000;
        fi.version_madeby = MZ_HOST_SYSTEM_UNIX << 8 | 63;
        fi.compression_method = (uint16_t)method;
        fi.flag = MZ_ZIP_FLAG_UTF8;
        fi.zip64 = MZ_ZIP64_AUTO;
        err = mz_zip_writer_add_buffer(w, (void *)(intptr_t)(blob + offs[i]), lens[i], &fi);
    }
    if (err == MZ_OK)
        err = mz_zip_writer_close(w);
    else
        mz_zip_writer_close(w);
    mz_zip_writer_delete(&w);
    return err;
}

/* Every entry through the reader OBJECT (mz_zip_reader_goto_first/next_entry, _entry_open, _entry_read in 65 535-byte
 * calls, _entry_close: mz_zip_rw.c:375-467) -- the layer that, in a build with crypto, hashes what it reads and compares the
 * digest with the entry's Hash extra field when the entry is closed (mz_zip_rw.c:409-451).  status[i] = the first error of
 * entry i (open, read or close), ulen[i] = bytes read.  Returns the number of entries walked or < 0. */
DRV_EXPORT int64_t drv_zip_reader_walk(const char *path, int32_t *status, int64_t *ulen, int64_t max_entries) {
    void *r = mz_zip_reader_create();
    uint8_t *buf = (uint8_t *)malloc(UINT16_MAX);
    int64_t n = 0;
    int32_t err;
    if (!r || !buf) {
        free(buf);
        return MZ_MEM_ERROR;
    }
    err = mz_zip_reader_open_file(r, path);
    if (err == MZ_OK)
        err = mz_zip_reader_goto_first_entry(r);
    while (err == MZ_OK && n < max_entries) {
        int32_t st = mz_zip_reader_entry_open(r);
        int64_t got = 0;
        if (st == MZ_OK) {
            for (;;) {
                int32_t rd = mz_zip_reader_entry_read(r, buf, UINT16_MAX);
                if (rd < 0)
                    st = rd;
                if (rd <= 0)
                    break;
                got += rd;
            }
            int32_t cl = mz_zip_reader_entry_close(r);
            if (st == MZ_OK)
                st = cl;
        }
        status[n] = st;
        ulen[n] = got;
        n++;
        err = mz_zip_reader_goto_next_entry(r);
    }
    mz_zip_reader_close(r);
    mz_zip_reader_delete(&r);
    free(buf);
    return (err == MZ_OK || err == MZ_END_OF_LIST) ? n : (int64_t)err;
}

/* ONE entry of `total` bytes -- `piece` over and over -- written through mz_zip_writer_entry_open / _write / _close in
 * 65 535-byte calls (mz_zip_rw.c:1427-1447), so that the caller never holds the entry: the bounded-memory test of the
 * WRITE streams.  A second, small entry follows it. */
DRV_EXPORT int32_t drv_zip_write_repeat(const char *path, int32_t method, int32_t level, const uint8_t *piece,
                                        int32_t piece_len, int64_t total) {
    void *w = mz_zip_writer_create();
    int32_t err;
    if (!w)
        return MZ_MEM_ERROR;
    mz_zip_writer_set_compress_method(w, (uint16_t)method);
    mz_zip_writer_set_compress_level(w, (int16_t)level);
    err = mz_zip_writer_open_file(w, path, 0, 0);
    for (int32_t i = 0; err == MZ_OK && i < 2; i++) {
        mz_zip_file fi;
        int64_t left = i == 0 ? total : (total < piece_len ? total : piece_len);
        int64_t at = 0;
        memset(&fi, 0, sizeof(fi));
        fi.filename = i == 0 ? "huge.bin" : "small.bin";
        fi.modified_date = 1700000000;
        fi.version_madeby = MZ_HOST_SYSTEM_UNIX << 8 | 63;
        fi.compression_method = (uint16_t)method;
        fi.flag = MZ_ZIP_FLAG_UTF8;
        fi.zip64 = MZ_ZIP64_FORCE; /* the size is not known when the local header is written */
        err = mz_zip_writer_entry_open(w, &fi);
        while (err == MZ_OK && left > 0) {
            int32_t off = (int32_t)(at % piece_len);
            int32_t n = piece_len - off;
            if (n > UINT16_MAX)
                n = UINT16_MAX;
            if (n > left)
                n = (int32_t)left;
            int32_t wr = mz_zip_writer_entry_write(w, piece + off, n);
            if (wr != n)
                err = wr < 0 ? wr : MZ_WRITE_ERROR;
            at += n;
            left -= n;
        }
        if (err == MZ_OK)
            err = mz_zip_writer_entry_close(w);
    }
    if (err == MZ_OK)
        err = mz_zip_writer_close(w);
    else
        mz_zip_writer_close(w);
    mz_zip_writer_delete(&w);
    return err;
}

/* Entry table through the reference's own central-directory walk
 * (mz_zip_goto_first/next_entry, mz_zip.c:2349-2412).  Per entry, 8 int64:
 * method, flag, crc, compressed, uncompressed, local header offset, CD
 * position, payload offset (found by opening the entry raw and asking the
 * archive stream where it stands, mz_zip.c:1874-1913). */
DRV_EXPORT int64_t drv_zip_index(const char *path, int64_t *table, int64_t max_entries) {
    void *r = mz_zip_reader_create();
    void *zip = NULL, *strm = NULL;
    int64_t n = 0;
    int32_t err;
    if (!r)
        return MZ_MEM_ERROR;
    err = mz_zip_reader_open_file(r, path);
    if (err != MZ_OK) {
        mz_zip_reader_delete(&r);
        return err;
    }
    mz_zip_reader_get_zip_handle(r, &zip);
    mz_zip_get_stream(zip, &strm);
    err = mz_zip_goto_first_entry(zip);
    while (err == MZ_OK) {
        mz_zip_file *fi = NULL;
        mz_zip_entry_get_info(zip, &fi);
        if (n < max_entries) {
            int64_t *t = table + n * 8;
            t[0] = fi->compression_method;
            t[1] = fi->flag;
            t[2] = fi->crc;
            t[3] = fi->compressed_size;
            t[4] = fi->uncompressed_size;
            t[5] = fi->disk_offset;
            t[6] = mz_zip_get_entry(zip);
            t[7] = -1;
            if (mz_zip_entry_read_open(zip, 1, NULL) == MZ_OK) {
                t[7] = mz_stream_tell(strm);
                mz_zip_entry_close(zip);
            }
        }
        n++;
        err = mz_zip_goto_next_entry(zip);
    }
    mz_zip_reader_close(r);
    mz_zip_reader_delete(&r);
    if (err != MZ_END_OF_LIST)
        return err;
    return n;
}

typedef struct {
    const char *path;
    const int64_t *cd_pos;
    int64_t first, count;
    uint32_t *crc;
    int64_t *ulen;
    int32_t *status;
    uint8_t *out;           /* optional: concatenated output */
    const int64_t *out_off; /* per-entry offsets into out */
    int32_t chunk;
    int32_t own_crc; /* also run drv-side crc over the bytes (parity mode); timing mode leaves it to mz_zip */
    int64_t bytes;
    const uint8_t *img; /* "mapped" mode (own_crc bit 1): the reader sits on mz_stream_mem over one shared read-only mapping */
    int64_t img_len;    /* instead of mz_zip_reader_open_file, whose split stream re-opens the file twice per entry */
} job_t;

/* one thread = one independent reader handle on the same file (distinct
 * handles share no mutable state, SURVEY 8b "Threading"). The per-entry loop
 * is the genuine reference hot path: mz_zip_entry_read -> mz_stream_read on the
 * codec stream -> mz_crypt_crc32_update (mz_zip.c:2031-2054), then the CRC
 * verification in mz_zip_entry_read_close (mz_zip.c:2116-2128). */
static void *job_run(void *arg) {
    job_t *j = (job_t *)arg;
    void *r = mz_zip_reader_create();
    void *zip = NULL;
    void *mem = NULL;
    uint8_t *buf = (uint8_t *)malloc((size_t)j->chunk);
    int32_t oerr = MZ_OPEN_ERROR;
    if (r && buf && j->img) {
        mem = mz_stream_mem_create();
        if (mem) {
            mz_stream_mem_set_buffer(mem, (void *)j->img, (int32_t)j->img_len);
            if (mz_stream_open(mem, NULL, MZ_OPEN_MODE_READ) == MZ_OK)
                oerr = mz_zip_reader_open(r, mem);
        }
    } else if (r && buf) {
        oerr = mz_zip_reader_open_file(r, j->path);
    }
    if (oerr != MZ_OK) {
        for (int64_t i = 0; i < j->count; i++)
            j->status[j->first + i] = MZ_OPEN_ERROR;
        free(buf);
        if (r)
            mz_zip_reader_delete(&r);
        if (mem)
            mz_stream_mem_delete(&mem);
        return NULL;
    }
    mz_zip_reader_get_zip_handle(r, &zip);
    for (int64_t i = j->first; i < j->first + j->count; i++) {
        int32_t err = mz_zip_goto_entry(zip, j->cd_pos[i]);
        int64_t total = 0;
        uint32_t crc = 0;
        if (err == MZ_OK)
            err = mz_zip_entry_read_open(zip, 0, NULL);
        if (err == MZ_OK) {
            for (;;) {
                int32_t rd = mz_zip_entry_read(zip, buf, j->chunk);
                if (rd < 0) {
                    err = rd;
                    break;
                }
                if (rd == 0)
                    break;
                if (j->own_crc & 1)
                    crc = mz_crypt_crc32_update(crc, buf, rd);
                if (j->out)
                    memcpy(j->out + j->out_off[i] + total, buf, (size_t)rd);
                total += rd;
            }
            int32_t cerr = mz_zip_entry_close(zip);
            if (err == MZ_OK)
                err = cerr;
        }
        if (!(j->own_crc & 1) && err == MZ_OK) {
            /* mz_zip verified entry_crc32 == stored crc (mz_zip.c:2122) */
            mz_zip_file *fi = NULL;
            if (mz_zip_goto_entry(zip, j->cd_pos[i]) == MZ_OK && mz_zip_entry_get_info(zip, &fi) == MZ_OK)
                crc = fi->crc;
        }
        j->crc[i] = crc;
        j->ulen[i] = total;
        j->status[i] = err;
        j->bytes += total;
    }
    mz_zip_reader_close(r);
    mz_zip_reader_delete(&r);
    if (mem)
        mz_stream_mem_delete(&mem);
    free(buf);
    return NULL;
}

/* Decode entries [0,n) of the archive with `nthreads` threads (contiguous
 * slices).  Returns elapsed seconds of the decode loop (CLOCK_MONOTONIC). */
DRV_EXPORT double drv_zip_read_all(const char *path, const int64_t *cd_pos, int64_t n, int32_t nthreads,
                                   int32_t chunk, int32_t own_crc, uint32_t *crc, int64_t *ulen, int32_t *status,
                                   uint8_t *out, const int64_t *out_off) {
    if (nthreads < 1)
        nthreads = 1;
    if (nthreads > n && n > 0)
        nthreads = (int32_t)n;
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    job_t *jobs = (job_t *)calloc((size_t)nthreads, sizeof(job_t));
    struct timespec t0, t1;
    int64_t per = n / nthreads, extra = n % nthreads, first = 0;
    const uint8_t *img = NULL;
    int64_t img_len = 0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    if (own_crc & 2) { /* mapped mode: one read-only mapping of the archive for every reader */
        int fd = open(path, O_RDONLY);
        struct stat sb;
        if (fd >= 0 && fstat(fd, &sb) == 0 && sb.st_size > 0 && sb.st_size <= INT32_MAX) {
            void *m = mmap(NULL, (size_t)sb.st_size, PROT_READ, MAP_PRIVATE | MAP_POPULATE, fd, 0);
            if (m != MAP_FAILED) {
                img = (const uint8_t *)m;
                img_len = (int64_t)sb.st_size;
            }
        }
        if (fd >= 0)
            close(fd);
    }
    for (int32_t t = 0; t < nthreads; t++) {
        job_t *j = &jobs[t];
        j->path = path;
        j->cd_pos = cd_pos;
        j->first = first;
        j->count = per + (t < extra ? 1 : 0);
        first += j->count;
        j->crc = crc;
        j->ulen = ulen;
        j->status = status;
        j->out = out;
        j->out_off = out_off;
        j->chunk = chunk;
        j->own_crc = own_crc;
        j->img = img;
        j->img_len = img_len;
        if (nthreads == 1)
            job_run(j);
        else
            pthread_create(&th[t], NULL, job_run, j);
    }
    if (nthreads > 1)
        for (int32_t t = 0; t < nthreads; t++)
            pthread_join(th[t], NULL);
    if (img)
        munmap((void *)img, (size_t)img_len);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    free(jobs);
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
