/* shim_zlib.c -- mz_stream_zlib re-implemented over the HIP backend.
 *
 * Drop-in for the reference's mz_strm_zlib.c (same 13 exported symbols,
 * mz_strm_zlib.h:20-35).  Host side stays C; the DEFLATE arithmetic runs in the
 * gfx950 kernels behind mzhip_inflate_host_a() (include/mzhip.h).  There is no
 * CPU codec in here: if the device is unusable, read() fails with an MZ error.
 *
 * Contract mirrored from the reference (file:line = mz_strm_zlib.c):
 *   create   :357-365  calloc, vtbl first, level -1, window_bits -15
 *   open     :65-107   resets totals; WRITE -> deflate, READ -> inflate
 *   read     :116-193  pulls <=32767-byte chunks from base (:132,146), clamps to
 *                      TOTAL_IN_MAX (:141-144), negative base reads returned
 *                      verbatim (:148-149), zlib-numbered error returned instead
 *                      of a byte count once an error is latched (:186-189)
 *   write    :243-264  returns size; TOTAL_IN += size; compressed bytes go to base in <=32767-byte writes
 *   close    :280-305  WRITE: finish the stream and flush; MZ_CLOSE_ERROR if an error is latched
 *   props    :312-355  TOTAL_IN / TOTAL_IN_MAX / TOTAL_OUT / HEADER_SIZE(0) /
 *                      COMPRESS_WINDOW get; COMPRESS_LEVEL / TOTAL_IN_MAX /
 *                      COMPRESS_WINDOW set; everything else MZ_EXIST_ERROR
 *   tell/seek:266-278  MZ_TELL_ERROR / MZ_SEEK_ERROR
 *
 * How a pull-mode byte stream meets a batch device: the shim pulls staging
 * chunks exactly like the reference, hands everything pulled so far to the
 * device, and asks it to decode the entry; when the device answers "input ended
 * early" (-5) the shim pulls more (doubling the attempt size, so total device
 * work stays within 2x) until the stream ends or base is exhausted.  Decoded
 * bytes are then served to read() calls from the host copy.  TOTAL_IN is the
 * exact number of compressed bytes the stream occupies (what mz_zip.c:2090,2116
 * need); it is only reported once every decoded byte has been served, so a
 * caller that stops early does not trip the CRC comparison at mz_zip.c:2116.
 *
 * COMPRESS_WINDOW (mz_strm_zlib.c:80,104): every value zlib's inflateInit2 / deflateInit2 accept: -8..-15 = raw (-15 is
 * what mz_zip.c uses), 8..15 = zlib wrapper (RFC 1950), 24..31 = gzip wrapper (RFC 1952; minigzip.c:80), 40..47 = detect
 * on READ, 0 = window from the zlib header on READ.  The wrappers are framing around
 * the same device path: header fields are parsed/emitted here, the trailer is checked against / filled from
 * the checksums the device computed (fused CRC-32, K5 Adler-32).  Error numbering follows zlib's inflate():
 * a bad header or trailer is Z_DATA_ERROR (-3), a preset dictionary request is Z_NEED_DICT (2), input that
 * ends inside the header or trailer is Z_BUF_ERROR (-5).  A value outside those ranges fails open() the way zlib's init
 * does (Z_STREAM_ERROR -> the reference's open() returns MZ_OPEN_ERROR); a stream whose header asks for a larger window
 * than the one set is Z_DATA_ERROR, as in inflate().
 */
#ifndef _POSIX_C_SOURCE
#define _POSIX_C_SOURCE 200112L /* clock_gettime */
#endif
#include <stdio.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

typedef struct mzhip_zlib_s {
    mzhip_stream stream; /* must be first (mz_strm.h:69-72) */
    int32_t mode;
    int32_t error;
    int8_t initialized;
    int16_t level;
    int32_t window_bits;
    int64_t total_in;
    int64_t total_out;
    int64_t max_total_in;
    /* read side */
    uint8_t *in;       /* compressed bytes pulled from base so far */
    int64_t in_len, in_cap;
    int8_t base_eof;   /* base returned 0 */
    int32_t base_err;  /* base failed after data had arrived: reported only if the stream turns out to need more */
    uint8_t *out;      /* decoded bytes (host copy) */
    int64_t out_len, out_cap, out_served;
    int8_t decoded;    /* device produced a final verdict */
    int32_t dev_status;
    int64_t dev_in_used;
    int64_t next_attempt; /* try the device again once in_len reaches this */
    int64_t csize_hint;   /* the entry's compressed size as its local header has it (0: unknown): when to ask the device, nothing else */
    /* write side, mzhip_prime_write: the entry so far equals bytes [0, wp_pos) of primed buffer wp_id */
    int64_t wp_id, wp_pos;
    int8_t wp_off; /* this entry is not (or no longer) following a primed buffer */
    int8_t tried_cache, out_borrowed; /* prime cache: looked up once; out points into the cache */
    void *prime_pin;                  /* keeps the cached generation alive while out points into it */
    int64_t base_pos0;                /* base position at the first read = payload offset */
    const uint32_t *seg_crc;          /* GPU CRCs of the 65 535-byte segments of a primed entry */
    int32_t wrap;      /* 0 raw, 1 zlib, 2 gzip (resolved from window_bits; 3 = detect, resolved by the header) */
    int32_t wlog;      /* log2 of the LZ77 window the caller asked for (8..15) */
    int32_t whdr;      /* READ: the window comes from the zlib header (windowBits 0) */
    int64_t hdr_len;   /* wrapper header bytes in front of the DEFLATE payload (0 = not parsed yet) */
    int8_t payload_done; /* device verdict on the payload is in; only the trailer is outstanding */
    uint32_t out_crc, out_adler;
    /* read side, entries too large for one device buffer (MZH_STREAM_WINDOW): decoded window by window.  out[] then holds
     * [history | the window's bytes]; the compressed bytes in front of the current block's header have been dropped */
    int8_t streaming;       /* window mode is on */
    int8_t stream_end;      /* the device has seen the end of the stream */
    int8_t kind_known, kind_no_room; /* refusal_is_not_a_distance(): asked once per stream */
    int8_t trailer_err;     /* the verdict is a trailer check that failed: inflate() gets there without needing room for output, so the call
                             * that returns the payload's last byte reports it even when that byte fills the caller's buffer */
    mzhip_inflate_state sst; /* where the decode goes on (bit positions from in[0]) */
    int64_t in_dropped;     /* compressed bytes dropped from the front of in[] */
    /* ... and the device's CRC-32 of every window in the pieces the caller reads it in (the size of its read() calls:
     * 65 535 from mz_zip_entry_read), so that the mz_crypt_crc32_update behind a read() that was served exactly such
     * pieces is answered from them instead of launching for 64 KiB (3 GiB entry: 22 s -> see DESIGN 4) */
    int64_t g0;             /* offset inside the entry of out[0] */
    int32_t read_stride;    /* the size of the caller's read() calls (0: none seen, or too small to be worth pieces) */
    struct mzh_piece {
        int64_t g;          /* offset inside the entry */
        uint32_t len, crc;
    } *pc;
    int32_t pc_n, pc_cap, pc_head;
    uint32_t *pc_tmp;
    int32_t pc_tmp_cap;
    /* ... under a zlib / gzip wrapper: the trailer's checksum runs over every window (combined from the device's per-window values) */
    uint32_t run_crc, run_adler;
    int64_t run_n;
    /* ... and by many waves when the stream stands at a block header (mzhip_inflate_parallel_host) */
    int8_t in_pinned;       /* ... and in[] (window mode: a gulp goes to the device before every window) */
    size_t in_pin_cap;
    int8_t out_pinned;      /* out[] is page-locked memory of the library's pool (mzhip_window_alloc): the device copies a window
                             * into it at link speed, into pageable memory at a third of that */
    size_t out_pin_cap;
    int64_t par_in_q16;     /* compressed bytes per decoded byte of the last many-wave window (Q16; 0 = none yet): how much of
                             * the input the next one is shown */
    /* where a window-mode stream's time went (MZHIP_STREAM_STATS=1 prints it at close): seconds */
    double t_par, t_serial, t_pull, t_open;
    int32_t n_par, n_serial;
    int64_t par_bytes, serial_bytes;
    int32_t par_miss;       /* windows in a row the many-wave decode got (next to) nothing out of */
    int32_t par_rest;       /* serial windows to go before it is tried again */
    /* ... and AHEAD of the caller (round 6): while read() calls are served from the window in out[], a thread of the stream's own
     * runs the next many-wave window into out2[] = [the last 96 KiB of out[] | room]; stream_next() then swaps the two.  The
     * decode call and its arguments are the ones stream_next() would have made a moment later: what a caller can observe
     * (bytes, return values, verdicts and the call that reports them) does not know the difference; the base stream is pulled
     * one gulp earlier.  The device call and the serving of a window used to take turns: half of a large entry's time each. */
    uint8_t *out2;
    int8_t out2_pinned;
    size_t out2_pin_cap;
    uint32_t *pc_tmp2;
    int32_t pc_tmp2_cap;
    pthread_t la_thread;    /* the stream's look-ahead thread: started with the first window that is decoded ahead, joined at close (a
                             * thread per window paid the HIP runtime's per-thread set-up every 64 MiB: 7 ms of a window's 13) */
    pthread_mutex_t la_mu;
    pthread_cond_t la_cv;
    int8_t la_started;      /* the thread exists */
    int8_t la_job;          /* 0 none, 1 posted / running, 2 done, 3 the thread is to exit */
    int8_t la_active;       /* a window is on its way (or done and not taken over): nothing touches in[], out2[], pc_tmp2 or la but the thread */
    struct {
        mzhip_inflate_state st_in, st_out;
        uint32_t show, hist, pol, pb, pended, wcrc, wadler, nseg, seg_first, stride;
        int32_t pr, device;
        int64_t gnew, bit0;
        double t;
    } la;
    double t_wait;          /* (MZHIP_STREAM_STATS) seconds stream_next() waited for that thread */
    int32_t n_ahead;        /* windows decoded ahead */
    int32_t n_attempts;     /* (stats) whole-entry decodes asked of the device before window mode */
    int64_t gulp_ramp;      /* window mode: compressed bytes pulled ahead of the next window; grows fourfold per window up to mzh_stream_gulp() */
    double t_begin;         /* (stats) when window mode began */
    double t_attempts;      /* (stats) seconds inside those decodes */
    double t_posted, t_overlap, t_book; /* (stats) when the last one was posted; reader time between a post and the next wait; the
                                         * reader's own work between a wait's end and the next post (take-over, input pull, history copy) */
    /* write side */
    uint8_t *wbuf;
    int64_t wlen, wcap;
    /* a full segment is coded by a thread of the stream while the caller fills the next one (wbuf and wjob.in swap) */
    pthread_t w_thread;
    pthread_mutex_t w_mu;
    pthread_cond_t w_cv;
    int8_t w_started, w_state; /* w_state: 0 none, 1 posted / running, 2 done, 3 the thread is to exit */
    int8_t w_pending;          /* a segment is on its way (or done and not pushed to the base stream yet) */
    struct {
        uint8_t *in, *out;     /* in: the other segment buffer */
        int64_t in_len;
        uint32_t out_cap, out_len, crc, adler;
        int32_t st, device;
    } wjob;
    uint32_t w_crc, w_adler;
    int64_t w_total;   /* uncompressed bytes already handed to the device */
    int8_t w_header_done;
    uint32_t slot; /* this stream's cell of mzhip_stream_epoch[] (shim_common.h) */
    uint16_t hash_alg;          /* a primed entry's device-verified digest (shim_sha.c answers mz_crypt_sha_end with it) */
    const uint8_t *hash_digest;
} mzhip_zlib;

static mzhip_stream_vtbl mzhip_zlib_vtbl = {
    mz_stream_zlib_open,   mz_stream_zlib_is_open, mz_stream_zlib_read,           mz_stream_zlib_write,
    mz_stream_zlib_tell,   mz_stream_zlib_seek,    mz_stream_zlib_close,          mz_stream_zlib_error,
    mz_stream_zlib_create, mz_stream_zlib_delete,  mz_stream_zlib_get_prop_int64, mz_stream_zlib_set_prop_int64};

static double mzh_now(void) {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static void out_release(mzhip_zlib *z) {
    if (z->out_pinned)
        mzhip_window_free(z->out, z->out_pin_cap);
    else if (!z->out_borrowed)
        free(z->out);
    z->out = NULL;
    z->out_pinned = 0;
    z->out_borrowed = 0;
}
static void in_release(mzhip_zlib *z) {
    if (z->in_pinned)
        mzhip_window_free(z->in, z->in_pin_cap);
    else
        free(z->in);
    z->in = NULL;
    z->in_pinned = 0;
}
/* in[] grows to ncap bytes; in window mode into page-locked memory when the pool has it */
static int32_t in_grow(mzhip_zlib *z, int64_t ncap) {
    if (z->streaming) {
        size_t cap = 0;
        uint8_t *p = (uint8_t *)mzhip_window_alloc((size_t)ncap, &cap);
        if (p) {
            if (z->in_len > 0)
                memcpy(p, z->in, (size_t)z->in_len);
            in_release(z);
            z->in = p;
            z->in_pinned = 1;
            z->in_pin_cap = cap;
            z->in_cap = ncap;
            return 0;
        }
    }
    if (z->in_pinned) { /* (the pool has run dry since: back to plain memory) */
        uint8_t *p = (uint8_t *)malloc((size_t)ncap);
        if (!p)
            return MZH_MEM_ERROR;
        memcpy(p, z->in, (size_t)z->in_len);
        in_release(z);
        z->in = p;
        z->in_cap = ncap;
        return 0;
    }
    uint8_t *p = (uint8_t *)realloc(z->in, (size_t)ncap);
    if (!p)
        return MZH_MEM_ERROR;
    z->in = p;
    z->in_cap = ncap;
    return 0;
}
/* the window buffer: page-locked when the pool has it, plain memory otherwise */
static uint8_t *out_window_alloc(mzhip_zlib *z, int64_t bytes) {
    size_t cap = 0;
    uint8_t *p = (uint8_t *)mzhip_window_alloc((size_t)bytes, &cap);
    if (p) {
        z->out_pinned = 1;
        z->out_pin_cap = cap;
        return p;
    }
    z->out_pinned = 0;
    return (uint8_t *)malloc((size_t)bytes);
}

/* what mz_stream_read does before dispatching (mz_strm.c:34-41) */
static int32_t base_read(mzhip_stream *base, void *buf, int32_t size) {
    if (!base || !base->vtbl || !base->vtbl->read)
        return MZH_PARAM_ERROR;
    if (!base->vtbl->is_open || base->vtbl->is_open(base) != MZH_OK)
        return MZH_STREAM_ERROR;
    return base->vtbl->read(base, buf, size);
}

static void lookahead_cancel(mzhip_zlib *z);
static void write_ahead_stop(mzhip_zlib *z);
static void free_buffers(mzhip_zlib *z) {
    write_ahead_stop(z); /* (... and the write side's reads wjob.in) */
    lookahead_cancel(z); /* (the look-ahead thread reads in[] and writes out2[]: it is joined before either goes) */
    in_release(z);
    free(z->pc);
    free(z->pc_tmp);
    z->pc = NULL;
    z->pc_tmp = NULL;
    z->pc_n = z->pc_cap = z->pc_head = z->pc_tmp_cap = 0;
    z->g0 = 0;
    out_release(z);
    mzhip_prime_unpin(z->prime_pin);
    z->prime_pin = NULL;
    z->out_borrowed = 0;
    free(z->wbuf);
    z->in = z->out = z->wbuf = NULL;
    z->in_len = z->in_cap = z->out_len = z->out_cap = z->out_served = 0;
    z->wlen = z->wcap = 0;
}

int32_t mz_stream_zlib_open(void *stream, const char *path, int32_t mode) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    mzhip_served_drop();
    mzhip_buffers_released(((mzhip_zlib *)stream)->slot);
    (void)path;
    z->total_in = 0;
    z->total_out = 0;
    z->error = 0;
    free_buffers(z);
    z->base_eof = 0;
    z->base_err = 0;
    z->wp_id = -1;
    z->wp_pos = 0;
    z->wp_off = 0;
    z->decoded = 0;
    z->streaming = z->stream_end = 0;
    z->trailer_err = 0;
    z->kind_known = z->kind_no_room = 0;
    z->t_par = z->t_serial = z->t_pull = z->t_wait = z->t_overlap = z->t_book = z->t_posted = 0.0;
    z->n_ahead = 0;
    z->n_attempts = 0;
    z->t_attempts = 0.0;
    z->t_open = mzh_now();
    z->n_par = z->n_serial = 0;
    z->par_bytes = z->serial_bytes = 0;
    z->par_miss = z->par_rest = 0;
    z->par_in_q16 = 0;
    z->in_dropped = 0;
    z->dev_status = 0;
    z->dev_in_used = 0;
    z->next_attempt = 0;
    z->csize_hint = 0;
    z->tried_cache = 0;
    z->hash_alg = 0;
    z->hash_digest = NULL;
    z->base_pos0 = -1;
    z->seg_crc = NULL;
    z->hdr_len = 0;
    z->payload_done = 0;
    z->w_crc = 0;
    z->w_adler = 1;
    z->w_total = 0;
    z->w_header_done = 0;
    {
        /* windowBits as inflateInit2 / deflateInit2 read it (mz_strm_zlib.c:87,97; any value the caller sets through
         * COMPRESS_WINDOW, :348-350): -8..-15 raw, 8..15 zlib wrapper (0: READ only, window from the header), +16 gzip,
         * +32 (READ only) gzip or zlib decided by the first two bytes.  Anything else fails the init call there
         * (Z_STREAM_ERROR) and therefore the open here. */
        int32_t wb = z->window_bits;
        const int32_t rd = (mode & MZH_OPEN_MODE_WRITE) ? 0 : 1;
        if (rd) { /* inflateReset2 (zlib 1.2.11 inflate.c): wrap from bits 4-5, window from the low four bits; 0 = from the header */
            if (wb < 0) {
                z->wrap = 0;
                wb = -wb;
            } else {
                const int32_t wr = wb >> 4; /* 0 zlib, 1 gzip, 2 either */
                if (wb < 48)
                    wb &= 15;
                z->wrap = wr + 1;
            }
            if (z->wrap > 3 || (wb && (wb < 8 || wb > 15)) || (z->wrap == 0 && wb == 0))
                return MZH_OPEN_ERROR; /* Z_STREAM_ERROR from the init call, mz_strm_zlib.c:101-102 */
            z->wlog = wb ? wb : 15;
            z->whdr = wb == 0; /* the zlib header's own window is the limit */
        } else { /* deflateInit2_ (zlib 1.2.11 deflate.c) */
            if (wb < 0) {
                z->wrap = 0;
                wb = -wb;
            } else if (wb > 15) {
                z->wrap = 2;
                wb -= 16;
            } else {
                z->wrap = 1;
            }
            if (wb < 8 || wb > 15 || (wb == 8 && z->wrap != 1))
                return MZH_OPEN_ERROR;
            z->wlog = wb == 8 ? 9 : wb; /* "until 256-byte window bug fixed" */
            z->whdr = 0;
        }
    }
    if (mode & MZH_OPEN_MODE_WRITE) {
        if (mzhip_device_count() <= 0) {
            z->error = MZH_STREAM_ERROR;
            return MZH_OPEN_ERROR;
        }
    } else if (mode & MZH_OPEN_MODE_READ) {
        if (mzhip_device_count() <= 0) {
            z->error = MZH_STREAM_ERROR;
            return MZH_OPEN_ERROR; /* mz_strm_zlib.c:101-102 */
        }
    }
    z->initialized = 1;
    z->mode = mode;
    return MZH_OK;
}

int32_t mz_stream_zlib_is_open(void *stream) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    return z->initialized == 1 ? MZH_OK : MZH_OPEN_ERROR;
}

/* pull one staging chunk; returns bytes read (0 = base exhausted) or <0 */
static int32_t pull_chunk(mzhip_zlib *z) {
    int32_t want = MZH_STAGING_BYTES;
    /* the first pull of a stream that may be served from a primed archive asks for no more than the bytes the lookup
     * compares (the size of a pull is not observable: the zip layer positions the base stream itself, mz_zip.c:1713);
     * a primed 64 KiB entry costs a 256-byte copy instead of 32 KiB, and the pages behind it are never touched */
    if (z->in_len == 0 && !z->tried_cache && mzhip_prime_any())
        want = 256;
    if (z->max_total_in > 0) {
        int64_t left = z->max_total_in - (z->in_dropped + z->in_len); /* (window mode drops input it is done with) */
        if (left < want)
            want = (int32_t)(left < 0 ? 0 : left);
    }
    if (want == 0) {
        z->base_eof = 1;
        return 0;
    }
    if (z->in_len + want > z->in_cap) {
        int64_t ncap = z->in_cap ? z->in_cap * 2 : (want <= 256 ? 1024 : 65536);
        while (ncap < z->in_len + want)
            ncap *= 2;
        const int32_t gr = in_grow(z, ncap);
        if (gr < 0)
            return gr;
    }
    int32_t rd = base_read(z->stream.base, z->in + z->in_len, want);
    if (rd < 0)
        return rd;
    if (rd == 0)
        z->base_eof = 1;
    z->in_len += rd;
    return rd;
}

static uint32_t le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* final verdict without (further) device work */
static int32_t verdict(mzhip_zlib *z, int32_t status, int64_t in_used) {
    z->dev_status = status;
    z->dev_in_used = in_used;
    z->decoded = 1;
    return 0;
}
/* Does inflate() meet the verdict without needing room for output?  A refused code, block type, header or stored-block length
 * and a failed trailer check: yes -- the call that returns the last byte in front of it reports it even when that byte fills
 * the caller's buffer (the bytes of that call are lost).  A distance too far back: no, MATCH waits for room first
 * (inflate.c) -- and the device's verdict does not say which refusal it was.  Behind 32 KiB of output no distance is too far
 * back any more, so every data error there is of the first kind.  In front of that the device is asked once more, the same
 * payload behind 32 KiB of make-believe history: a distance that reached in front of the entry now lands in it and the decode
 * goes on -- a refusal that stands where it stood was not a distance.  (One more launch for a stream that is refused in its
 * first 32 KiB exactly at the end of a read() call, nothing otherwise.) */
static int32_t refusal_is_not_a_distance(mzhip_zlib *z, const uint8_t *in, int64_t in_len, int64_t in_used, int64_t produced) {
    if (z->kind_known)
        return z->kind_no_room;
    z->kind_known = 1;
    z->kind_no_room = 0;
    if (in_len <= 0 || in_len > 0x7FFFFFFF)
        return 0;
    const int64_t cap = 32768 + produced + 320;
    uint8_t *tmp = (uint8_t *)calloc(1, (size_t)cap);
    if (!tmp)
        return 0;
    mzhip_inflate_state sin;
    memset(&sin, 0, sizeof(sin));
    sin.out_pos = 32768;
    sin.flags = 1;
    uint32_t ol = 0, iu = 0, crc = 0;
    mzhip_inflate_host_args a;
    memset(&a, 0, sizeof(a));
    a.size = (uint32_t)sizeof(a);
    a.in = in;
    a.in_len = (uint32_t)in_len;
    a.buf = tmp;
    a.buf_cap = (uint32_t)cap;
    a.state_in = &sin;
    a.out_len = &ol;
    a.in_used = &iu;
    a.crc = &crc;
    const int32_t st = mzhip_inflate_host_a(&a);
    free(tmp);
    z->kind_no_room = st == MZHIP_STATUS_DATA_ERROR && (int64_t)iu == in_used && (int64_t)ol == 32768 + produced;
    return z->kind_no_room;
}
static int32_t verdict_needs_no_room(mzhip_zlib *z, int64_t produced) {
    if (z->trailer_err)
        return 1;
    if (z->dev_status != MZHIP_STATUS_DATA_ERROR)
        return 0;
    if (produced >= 32768)
        return 1;
    if (z->streaming) /* (the first window of an entry decoded in windows: nothing of its input has been given up yet) */
        return z->g0 == 0 && z->in_dropped == 0 ? refusal_is_not_a_distance(z, z->in, z->in_len, z->dev_in_used, produced) : 0;
    if (z->out_borrowed || z->dev_in_used < z->hdr_len)
        return 0;
    return refusal_is_not_a_distance(z, z->in + z->hdr_len, z->in_len - z->hdr_len, z->dev_in_used - z->hdr_len, produced);
}
/* a trailer check that failed ("incorrect data check" / "incorrect length check") */
static int32_t trailer_verdict(mzhip_zlib *z, int64_t in_used) {
    z->trailer_err = 1;
    return verdict(z, MZHIP_STATUS_DATA_ERROR, in_used);
}

/* Wrapper header in front of the DEFLATE payload, field by field as zlib's inflate() HEAD..HCRC / DICTID
 * states read it.  Returns 1 = header complete (z->hdr_len set), 0 = verdict reached (error), 2 = more input
 * needed. */
static int32_t parse_wrapper_header(mzhip_zlib *z) {
    const uint8_t *p = z->in;
    const int64_t n = z->in_len;
#define NEED(k)                                                    \
    do {                                                           \
        if (n < (k)) {                                             \
            if (!z->base_eof)                                      \
                return 2;                                          \
            verdict(z, MZHIP_STATUS_BUF_ERROR, n);                 \
            return 0;                                              \
        }                                                          \
    } while (0)
    NEED(2);
    int32_t wrap = z->wrap;
    if (wrap == 3)
        wrap = (p[0] == 0x1F && p[1] == 0x8B) ? 2 : 1;
    if (wrap == 2) {
        if (p[0] != 0x1F || p[1] != 0x8B) {
            verdict(z, MZHIP_STATUS_DATA_ERROR, 2); /* "incorrect header check" */
            return 0;
        }
        NEED(4);
        if (p[2] != 8 || (p[3] & 0xE0)) {
            verdict(z, MZHIP_STATUS_DATA_ERROR, 4); /* "unknown compression method" / "unknown header flags set" */
            return 0;
        }
        const int32_t flg = p[3];
        int64_t pos = 10; /* MTIME(4) XFL OS */
        NEED(pos);
        if (flg & 0x04) { /* FEXTRA */
            NEED(pos + 2);
            pos += 2 + ((int64_t)p[pos] | ((int64_t)p[pos + 1] << 8));
            NEED(pos);
        }
        for (int32_t bit = 0x08; bit <= 0x10; bit <<= 1) { /* FNAME, FCOMMENT: zero-terminated */
            if (!(flg & bit))
                continue;
            for (;;) {
                NEED(pos + 1);
                if (p[pos++] == 0)
                    break;
            }
        }
        if (flg & 0x02) { /* FHCRC: low 16 bits of the CRC-32 of the header so far (checksum on the device) */
            NEED(pos + 2);
            const uint32_t want = (uint32_t)p[pos] | ((uint32_t)p[pos + 1] << 8);
            if ((mzhip_crc32_host(0, p, (size_t)pos) & 0xFFFFu) != want) {
                verdict(z, MZHIP_STATUS_DATA_ERROR, pos + 2); /* "header crc mismatch" */
                return 0;
            }
            pos += 2;
        }
        z->wrap = 2;
        z->hdr_len = pos;
        return 1;
    }
    /* zlib: CMF FLG, ((CMF << 8) | FLG) % 31 == 0, CM == 8, CINFO <= 7 */
    if ((((uint32_t)p[0] << 8) | p[1]) % 31u != 0 || (p[0] & 0x0F) != 8 || (p[0] >> 4) > 7 ||
        (!z->whdr && (int32_t)(p[0] >> 4) + 8 > z->wlog)) { /* "invalid window size": the stream needs a larger window than the caller allows */
        verdict(z, MZHIP_STATUS_DATA_ERROR, 2);
        return 0;
    }
    if (p[1] & 0x20) { /* FDICT: 4-byte dictionary id, then inflate() asks for the dictionary */
        NEED(6);
        verdict(z, 2 /* Z_NEED_DICT */, 6);
        return 0;
    }
    z->wrap = 1;
    z->hdr_len = 2;
    return 1;
#undef NEED
}

/* run the device over everything pulled so far; 0 = verdict reached, 1 = wants more input */
/* Entries whose decoded size exceeds one window are decoded window by window (mz_strm_zlib.c:116-193 streams any size
 * through 32 767 bytes; here the unit is what is worth a launch).  Memory is O(window + one DEFLATE block of input). */
#ifndef MZH_STREAM_WINDOW
#define MZH_STREAM_WINDOW (64 << 20)
#endif
#ifndef MZH_STREAM_GULP
#define MZH_STREAM_GULP (16 << 20) /* compressed bytes pulled ahead of a window's launch */
#endif
/* The two sizes are the defaults of a run-time knob (mzhip_set_stream_window(), or MZHIP_STREAM_WINDOW / MZHIP_STREAM_GULP
 * in the environment of the first stream that asks): the memory bound of one READ stream is window + gulp + one block.
 * Small values are what the GPU tests use to cross many windows with streams of a few hundred KiB. */
static int64_t mzh_win_bytes, mzh_gulp_bytes; /* 0 = not decided yet */
static int64_t mzh_env_bytes(const char *name, int64_t dflt, int64_t floor) {
    const char *e = getenv(name);
    int64_t v = e ? strtoll(e, NULL, 0) : 0;
    if (v <= 0)
        v = dflt;
    return v < floor ? floor : v;
}
MZHIP_API void mzhip_set_stream_window(int64_t window_bytes, int64_t gulp_bytes) {
    /* a window holds at least the 32 KiB history plus something to decode into; 0 = back to the default / environment */
    __atomic_store_n(&mzh_win_bytes, window_bytes > 0 ? (window_bytes < (128 << 10) ? (128 << 10) : window_bytes) : 0, __ATOMIC_RELAXED);
    __atomic_store_n(&mzh_gulp_bytes, gulp_bytes > 0 ? (gulp_bytes < (32 << 10) ? (32 << 10) : gulp_bytes) : 0, __ATOMIC_RELAXED);
}
/* (decided lazily by whichever thread asks first; every thread computes the same value, the loads and stores are atomic) */
int64_t mzh_stream_window(void) {
    int64_t v = __atomic_load_n(&mzh_win_bytes, __ATOMIC_RELAXED);
    if (!v) {
        v = mzh_env_bytes("MZHIP_STREAM_WINDOW", MZH_STREAM_WINDOW, 128 << 10);
        __atomic_store_n(&mzh_win_bytes, v, __ATOMIC_RELAXED);
    }
    return v > 0x7FFFFFFF ? 0x7FFFFFFF : v;
}
int64_t mzh_stream_gulp(void) {
    int64_t v = __atomic_load_n(&mzh_gulp_bytes, __ATOMIC_RELAXED);
    if (!v) {
        v = mzh_env_bytes("MZHIP_STREAM_GULP", MZH_STREAM_GULP, 32 << 10);
        __atomic_store_n(&mzh_gulp_bytes, v, __ATOMIC_RELAXED);
    }
    return v;
}

/* One inflate() state per entry is the reference's model and one wave per entry is this backend's; an entry that needs
 * window mode is large, and one wave decodes 0.1 - 0.2 GB/s.  Whenever the stream stands at a block header the window is
 * first offered to mzhip_inflate_parallel_host (a wave per DEFLATE block); what it does not take -- fixed blocks, a block
 * larger than the window, the end of the input -- goes through the serial kernel, which is asked to stop at the next block
 * header as long as the many-wave decode has been worth its search.  MZHIP_STREAM_PARALLEL=0 turns it off. */
#ifndef MZH_PAR_MIN_IN
#define MZH_PAR_MIN_IN (256 << 10) /* compressed bytes below which a window is not worth the search */
#endif
#ifndef MZH_PAR_MIN_ROOM
#define MZH_PAR_MIN_ROOM (1 << 20) /* room in the window below which it is not offered; output below which an offer counts as a miss */
#endif
#ifndef MZH_STREAM_EARLY
#define MZH_STREAM_EARLY (256 << 10) /* compressed bytes pulled, entry not over: window mode from here on */
#endif
#ifndef MZH_STREAM_EARLY_OUT
#define MZH_STREAM_EARLY_OUT (4 << 20) /* ... or decoded bytes */
#endif
static int8_t mzh_par_mode = -1; /* (read and set with atomics: streams of several threads ask, tests/test_autoprime_emul.py under TSan) */
MZHIP_API void mzhip_set_stream_parallel(int32_t on) { __atomic_store_n(&mzh_par_mode, (int8_t)(on ? 1 : 0), __ATOMIC_RELAXED); }
static int32_t mzh_stream_parallel(void) {
    int8_t m = __atomic_load_n(&mzh_par_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        const char *e = getenv("MZHIP_STREAM_PARALLEL");
        m = (e && e[0] == '0') ? 0 : 1;
        __atomic_store_n(&mzh_par_mode, m, __ATOMIC_RELAXED);
    }
    return m;
}

/* How far window mode pulls ahead of a window.  The entry's compressed size is not known here (mz_zip.c:1815-1830 sets
 * MZ_STREAM_PROP_TOTAL_IN_MAX for raw, stored and encrypted entries only), and what is pulled behind the entry's end is
 * read for nothing: 1 MiB in front of the first window, four times as much per window after it, up to the gulp. */
static int64_t gulp_now(const mzhip_zlib *z) {
    const int64_t g = mzh_stream_gulp();
    return z->gulp_ramp > 0 && z->gulp_ramp < g ? z->gulp_ramp : g;
}
static void gulp_grow(mzhip_zlib *z) {
    if (z->gulp_ramp > 0 && z->gulp_ramp < mzh_stream_gulp())
        z->gulp_ramp *= 4;
}

static int32_t stream_drop_input(mzhip_zlib *z) {
    /* everything in front of the current block's header is done with (dword granular: the device addresses dwords) */
    const int64_t drop = (int64_t)((z->sst.hdr_bit >> 3) & ~3u);
    if (drop > 0) {
        memmove(z->in, z->in + drop, (size_t)(z->in_len - drop));
        z->in_len -= drop;
        z->in_dropped += drop;
        z->sst.hdr_bit -= (uint32_t)drop * 8u;
        z->sst.bit -= (uint32_t)drop * 8u;
    }
    return 0;
}

/* the pieces of one decode call: the same cut as a window of mzhip_inflate_host_a makes (first, stride ..., rest) */
static void stream_pieces_add(mzhip_zlib *z, int64_t g, int64_t nbytes, uint32_t first, uint32_t stride, uint32_t nseg) {
    if (nbytes <= 0 || !stride)
        return;
    /* drop what has been served already, make room */
    const int64_t served_to = z->g0 + z->out_served;
    while (z->pc_head < z->pc_n && z->pc[z->pc_head].g + z->pc[z->pc_head].len <= served_to)
        z->pc_head++;
    if (z->pc_head > 0) {
        memmove(z->pc, z->pc + z->pc_head, (size_t)(z->pc_n - z->pc_head) * sizeof(z->pc[0]));
        z->pc_n -= z->pc_head;
        z->pc_head = 0;
    }
    if (z->pc_n + (int32_t)nseg > z->pc_cap) {
        const int32_t ncap = z->pc_n + (int32_t)nseg + 64;
        struct mzh_piece *np = (struct mzh_piece *)realloc(z->pc, (size_t)ncap * sizeof(z->pc[0]));
        if (!np) {
            z->pc_n = z->pc_head = 0; /* no pieces: the checksum calls take the ordinary path */
            return;
        }
        z->pc = np;
        z->pc_cap = ncap;
    }
    int64_t left = nbytes;
    uint32_t i = 0;
    uint32_t n = first < (uint64_t)left ? first : (uint32_t)left;
    if (n) {
        z->pc[z->pc_n].g = g;
        z->pc[z->pc_n].len = n;
        z->pc[z->pc_n++].crc = z->pc_tmp[i++];
        g += n;
        left -= n;
    }
    while (left > 0 && i < nseg) {
        n = (uint64_t)left < stride ? (uint32_t)left : stride;
        z->pc[z->pc_n].g = g;
        z->pc[z->pc_n].len = n;
        z->pc[z->pc_n++].crc = z->pc_tmp[i++];
        g += n;
        left -= n;
    }
}

/* CRC-32 of out[out_served .. out_served + n) from the pieces, when these bytes are exactly a run of whole pieces: 1 */
static int32_t stream_pieces_crc(mzhip_zlib *z, int32_t n, uint32_t *crc) {
    const int64_t a = z->g0 + z->out_served, b = a + n;
    int32_t i = z->pc_head;
    while (i < z->pc_n && z->pc[i].g + z->pc[i].len <= a)
        i++;
    z->pc_head = i;
    if (i >= z->pc_n || z->pc[i].g != a)
        return 0;
    uint32_t c = 0;
    int64_t at = a;
    for (; i < z->pc_n && at < b; i++) {
        if (z->pc[i].g != at || at + z->pc[i].len > b)
            return 0;
        c = (at == a) ? z->pc[i].crc : mzhip_crc32_combine(c, z->pc[i].crc, z->pc[i].len);
        at += z->pc[i].len;
    }
    if (at != b)
        return 0;
    *crc = c;
    return 1;
}

/* n new bytes with these device-computed checksums join the wrapper's running ones */
static void stream_sum(mzhip_zlib *z, int64_t n, uint32_t wcrc, uint32_t wadler) {
    if (z->wrap == 0 || n <= 0)
        return;
    if (z->wrap == 2)
        z->run_crc = z->run_n ? mzhip_crc32_combine(z->run_crc, wcrc, (uint64_t)n) : wcrc;
    else
        z->run_adler = mzhip_adler32_combine(z->run_adler, wadler, (uint64_t)n);
    z->run_n += n;
}

/* the DEFLATE payload ended cleanly `used` bytes into the stream: a raw stream is done; under a wrapper the trailer follows
 * (gzip = CRC-32 then ISIZE, little endian; zlib = Adler-32, big endian) and is checked the way inflate() checks it */
static int32_t stream_finish(mzhip_zlib *z, int64_t used) {
    if (z->wrap == 0)
        return verdict(z, MZHIP_STATUS_OK, used);
    const int64_t tl = z->wrap == 2 ? 8 : 4;
    const int64_t lo = used - z->in_dropped;
    while (!z->base_eof && z->in_len < lo + tl) {
        const int32_t rd = pull_chunk(z);
        if (rd < 0) {
            z->base_err = rd;
            z->base_eof = 1;
        }
    }
    if (z->in_len < lo + tl) {
        /* (a gzip trailer cut inside ISIZE: inflate() has checked the CRC field by then -- a wrong one is "incorrect data
         * check" where it stands, not a request for more input; round 5, a wrapper fuzz on the emulation) */
        if (z->wrap == 2 && z->in_len >= lo + 4 && le32(z->in + lo) != z->run_crc)
            return trailer_verdict(z, used + 4);
        return verdict(z, MZHIP_STATUS_BUF_ERROR, z->in_dropped + z->in_len);
    }
    const uint8_t *t = z->in + lo;
    if (z->wrap == 2) {
        if (le32(t) != z->run_crc) /* "incorrect data check": inflate() stops after the CRC field */
            return trailer_verdict(z, used + 4);
        if (le32(t + 4) != (uint32_t)(z->g0 + z->out_len)) /* "incorrect length check" */
            return trailer_verdict(z, used + 8);
        return verdict(z, MZHIP_STATUS_OK, used + 8);
    }
    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    return want == z->run_adler ? verdict(z, MZHIP_STATUS_OK, used + 4) : trailer_verdict(z, used + 4);
}

/* ---- window mode, one window ahead (see the struct) ---- */
#define MZH_LA_HIST 98304 /* bytes of out[] that go in front of the next window: the 32 KiB of history and what may not have been served yet */
int32_t mzhip_prime_current_device(void);   /* (mzhip_prime.inc) the calling thread's device */
void mzhip_thread_use_device(int32_t dev);  /* (mzhip_runtime.inc) ... and making it another thread's */
static int8_t mzh_la_mode = -1;
static uint64_t mzh_la_windows; /* windows taken over from a look-ahead thread, all streams (tests, reports) */
MZHIP_API void mzhip_set_stream_lookahead(int32_t on) { __atomic_store_n(&mzh_la_mode, (int8_t)(on ? 1 : 0), __ATOMIC_RELAXED); }
MZHIP_API uint64_t mzhip_stream_lookahead_windows(void) { return __atomic_load_n(&mzh_la_windows, __ATOMIC_RELAXED); }
static int32_t mzh_stream_lookahead(void) {
    int8_t m = __atomic_load_n(&mzh_la_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        const char *e = getenv("MZHIP_STREAM_LOOKAHEAD"); /* "0": the device call and the serving take turns, as before round 6 */
        m = (e && e[0] == '0') ? 0 : 1;
        __atomic_store_n(&mzh_la_mode, m, __ATOMIC_RELAXED);
    }
    return m;
}
static void *lookahead_run(void *arg) {
    mzhip_zlib *z = (mzhip_zlib *)arg;
    int32_t dev = -2;
    pthread_mutex_lock(&z->la_mu);
    for (;;) {
        while (z->la_job != 1 && z->la_job != 3)
            pthread_cond_wait(&z->la_cv, &z->la_mu);
        if (z->la_job == 3)
            break;
        pthread_mutex_unlock(&z->la_mu);
        if (z->la.device >= 0 && z->la.device != dev) {
            dev = z->la.device;
            mzhip_thread_use_device(dev);
        }
        const double t0 = mzh_now();
        z->la.pr = mzhip_inflate_parallel_host(z->in, z->la.show, z->out2, (uint32_t)z->out_cap, &z->la.st_in, &z->la.st_out, &z->la.pol, &z->la.pb,
                                               &z->la.pended, z->wrap == 2 ? &z->la.wcrc : NULL, z->wrap == 1 ? &z->la.wadler : NULL, z->la.seg_first,
                                               z->pc_tmp2_cap ? z->la.stride : 0u, z->pc_tmp2, (uint32_t)z->pc_tmp2_cap, &z->la.nseg);
        z->la.t = mzh_now() - t0;
        pthread_mutex_lock(&z->la_mu);
        z->la_job = 2;
        pthread_cond_broadcast(&z->la_cv);
    }
    pthread_mutex_unlock(&z->la_mu);
    return NULL;
}
/* wait for the window that is on its way (if one is) */
static void lookahead_wait(mzhip_zlib *z) {
    if (!z->la_active)
        return;
    pthread_mutex_lock(&z->la_mu);
    while (z->la_job == 1)
        pthread_cond_wait(&z->la_cv, &z->la_mu);
    z->la_job = 0;
    pthread_mutex_unlock(&z->la_mu);
    z->la_active = 0;
}
static void lookahead_cancel(mzhip_zlib *z) {
    lookahead_wait(z);
    if (z->la_started) {
        pthread_mutex_lock(&z->la_mu);
        z->la_job = 3;
        pthread_cond_broadcast(&z->la_cv);
        pthread_mutex_unlock(&z->la_mu);
        pthread_join(z->la_thread, NULL);
        pthread_mutex_destroy(&z->la_mu);
        pthread_cond_destroy(&z->la_cv);
        z->la_started = 0;
        z->la_job = 0;
    }
    if (z->out2) {
        if (z->out2_pinned)
            mzhip_window_free(z->out2, z->out2_pin_cap);
        else
            free(z->out2);
    }
    z->out2 = NULL;
    z->out2_pinned = 0;
    free(z->pc_tmp2);
    z->pc_tmp2 = NULL;
    z->pc_tmp2_cap = 0;
}
/* the many-wave decode has just delivered a window and stands at a block header: start the next one.  Nothing happens when
 * anything is not as the steady state of a large entry has it -- stream_next() then does what it always did. */
static void lookahead_start(mzhip_zlib *z, int32_t par_on) {
    if (!mzh_stream_lookahead() || !par_on || z->la_active || z->stream_end || z->par_rest > 0 || z->par_miss > 0 || z->sst.bit != z->sst.hdr_bit ||
        !(z->sst.flags & 1u) || z->out_len < 32768)
        return;
    const int64_t gulp = gulp_now(z);
    while (!z->base_eof && z->in_len < gulp) { /* the gulp the next window would have pulled */
        const double tp0 = mzh_now();
        const int32_t rd = pull_chunk(z);
        z->t_pull += mzh_now() - tp0;
        if (rd < 0) {
            if (z->in_len == 0)
                return;
            z->base_err = rd;
            z->base_eof = 1;
        }
    }
    gulp_grow(z);
    const int64_t hist = z->out_len < MZH_LA_HIST ? z->out_len : MZH_LA_HIST;
    if (z->in_len < MZH_PAR_MIN_IN || z->out_cap - hist < MZH_PAR_MIN_ROOM)
        return;
    if (!z->out2) {
        size_t cap = 0;
        z->out2 = (uint8_t *)mzhip_window_alloc((size_t)z->out_cap, &cap);
        z->out2_pinned = z->out2 != NULL;
        z->out2_pin_cap = cap;
        if (!z->out2)
            z->out2 = (uint8_t *)malloc((size_t)z->out_cap);
        if (!z->out2)
            return;
    }
    const uint32_t stride = (uint32_t)z->read_stride;
    if (stride) {
        const int32_t want = (int32_t)(z->out_cap / stride) + 4;
        if (want > z->pc_tmp2_cap) {
            free(z->pc_tmp2);
            z->pc_tmp2 = (uint32_t *)malloc((size_t)want * sizeof(uint32_t));
            z->pc_tmp2_cap = z->pc_tmp2 ? want : 0;
        }
    }
    memcpy(z->out2, z->out + (z->out_len - hist), (size_t)hist);
    z->la.st_in = z->sst;
    z->la.st_in.out_pos = (uint32_t)hist;
    z->la.st_in.flags = 1;
    z->la.hist = (uint32_t)hist;
    z->la.gnew = z->g0 + z->out_len;
    z->la.stride = stride;
    z->la.seg_first = stride ? (uint32_t)((stride - z->la.gnew % stride) % stride) : 0u;
    int64_t show = z->in_len;
    if (z->par_in_q16 > 0) {
        const int64_t est = (int64_t)(z->sst.hdr_bit >> 3) + (((z->out_cap - hist) * z->par_in_q16) >> 15) + (256 << 10);
        if (est < show)
            show = est;
    }
    z->la.show = (uint32_t)show;
    z->la.bit0 = z->sst.hdr_bit;
    z->la.pol = z->la.pb = z->la.pended = z->la.nseg = 0;
    z->la.wcrc = 0;
    z->la.wadler = 1;
    z->la.pr = 0;
    z->la.device = mzhip_prime_current_device();
    if (!z->la_started) {
        if (pthread_mutex_init(&z->la_mu, NULL) != 0)
            return;
        if (pthread_cond_init(&z->la_cv, NULL) != 0) {
            pthread_mutex_destroy(&z->la_mu);
            return;
        }
        z->la_job = 0;
        if (pthread_create(&z->la_thread, NULL, lookahead_run, z) != 0) {
            pthread_mutex_destroy(&z->la_mu);
            pthread_cond_destroy(&z->la_cv);
            return;
        }
        z->la_started = 1;
    }
    pthread_mutex_lock(&z->la_mu);
    z->la_job = 1;
    pthread_cond_broadcast(&z->la_cv);
    pthread_mutex_unlock(&z->la_mu);
    z->la_active = 1;
    z->t_posted = mzh_now();
}
/* stream_next(): is a window there that was decoded ahead?  1 = taken over (*ret is stream_next's answer), 0 = no: go on as ever
 * (*skip_par: the many-wave decode has just been asked at this very header and had nothing) */
static int32_t lookahead_adopt(mzhip_zlib *z, int32_t *ret, int32_t *skip_par) {
    if (!z->la_active)
        return 0;
    const double tw0 = mzh_now();
    z->t_overlap += tw0 - z->t_posted;
    lookahead_wait(z);
    z->t_wait += mzh_now() - tw0;
    z->t_par += z->la.t;
    z->n_par++;
    if (z->la.pr < 0) {
        z->stream_end = 1;
        *ret = verdict(z, MZH_STREAM_ERROR, z->in_dropped); /* device / runtime failure */
        return 1;
    }
    if (!z->la.pb) { /* (next to) nothing for a wave of its own at this header: as if stream_next() had asked */
        z->par_miss++;
        *skip_par = 1;
        return 0;
    }
    /* out2[] = [hist bytes of out[]'s end | the new window]: it becomes out[] */
    const int64_t unserved = z->out_len - z->out_served, hist = z->la.hist;
    if (unserved > hist) { /* (cannot be: stream_next() is called with less than 64 KiB unserved) */
        *skip_par = 0;
        return 0;
    }
    mzhip_served_drop();
    {
        uint8_t *t = z->out;
        const int8_t tp = z->out_pinned;
        const size_t tc = z->out_pin_cap;
        z->out = z->out2;
        z->out_pinned = z->out2_pinned;
        z->out_pin_cap = z->out2_pin_cap;
        z->out2 = t;
        z->out2_pinned = tp;
        z->out2_pin_cap = tc;
        uint32_t *q = z->pc_tmp;
        const int32_t qc = z->pc_tmp_cap;
        z->pc_tmp = z->pc_tmp2;
        z->pc_tmp_cap = z->pc_tmp2_cap;
        z->pc_tmp2 = q;
        z->pc_tmp2_cap = qc;
    }
    z->g0 += z->out_len - hist;
    z->out_served = hist - unserved;
    z->out_len = hist;
    const int64_t made = (int64_t)z->la.pol - hist;
    z->par_bytes += made;
    z->n_ahead++;
    __atomic_fetch_add(&mzh_la_windows, 1, __ATOMIC_RELAXED);
    if (z->la.nseg)
        stream_pieces_add(z, z->la.gnew, made, z->la.seg_first, z->la.stride, z->la.nseg);
    if (made > 0)
        z->par_in_q16 = ((((int64_t)z->la.st_out.hdr_bit - z->la.bit0) >> 3) << 16) / made + 1;
    stream_sum(z, made, z->la.wcrc, z->la.wadler);
    z->out_len = z->la.pol;
    z->sst = z->la.st_out;
    if (z->la.pended) {
        z->stream_end = 1;
        *ret = stream_finish(z, z->in_dropped + (((int64_t)z->la.st_out.bit + 7) >> 3));
        return 1;
    }
    stream_drop_input(z);
    z->par_miss = made >= MZH_PAR_MIN_ROOM ? 0 : z->par_miss + 1;
    if (z->par_miss >= 3) {
        z->par_miss = 2;
        z->par_rest = 8;
    }
    if (z->out_len > z->out_served) {
        *ret = 0;
        return 1;
    }
    return 0; /* (nothing new to serve: the loop of stream_next() goes on from the new state) */
}

/* window mode: make more decoded bytes available behind out_served.  Returns 0 (bytes, the stream end or a verdict are
 * there) or a negative MZ error. */
static int32_t stream_next(mzhip_zlib *z) {
    int32_t skip_par = 0;
    {
        int32_t ret = 0;
        const double tb0 = mzh_now(), tw = z->t_wait;
        if (lookahead_adopt(z, &ret, &skip_par)) {
            if (ret == 0 && !z->stream_end) { /* (the same test the synchronous window makes below) */
                const int32_t par_on = mzh_stream_parallel() && z->in_len < ((int64_t)1 << 28) && mzh_stream_gulp() >= MZH_PAR_MIN_IN &&
                                       z->out_cap >= 2 * (int64_t)MZH_PAR_MIN_ROOM;
                lookahead_start(z, par_on);
            }
            z->t_book += (mzh_now() - tb0) - (z->t_wait - tw);
            return ret;
        }
    }
    /* little left to serve: slide.  What stays is what has not been served yet and, in any case, the last 32 KiB that
     * were produced (the history back-references may reach); the rest of the buffer is room for the next window */
    if (z->out_len - z->out_served < 65536) {
        int64_t from = z->out_len > 32768 ? z->out_len - 32768 : 0;
        if (from > z->out_served)
            from = z->out_served;
        if (from > 0) {
            memmove(z->out, z->out + from, (size_t)(z->out_len - from));
            z->out_len -= from;
            z->out_served -= from;
            z->g0 += from;
        }
    }
    for (;;) {
        /* compressed bytes for about a window: pull ahead (each pull <= 32767 bytes like the reference's) */
        const double tp0 = mzh_now();
        const int64_t gulp = gulp_now(z);
        while (!z->base_eof && z->in_len < gulp) {
            const int32_t rd = pull_chunk(z);
            if (rd < 0) {
                if (z->in_len == 0)
                    return rd;
                z->base_err = rd;
                z->base_eof = 1;
            }
        }
        gulp_grow(z);
        z->t_pull += mzh_now() - tp0;
        z->sst.out_pos = (uint32_t)z->out_len;
        z->sst.flags = 1;
        mzhip_inflate_state nst;
        uint32_t out_len = 0, in_used = 0, crc = 0, nseg = 0;
        /* the new bytes start at offset gnew of the entry; pieces end where the caller's read() calls end */
        const int64_t gnew = z->g0 + z->out_len;
        const uint32_t stride = (uint32_t)z->read_stride;
        const uint32_t seg_first = stride ? (uint32_t)((stride - gnew % stride) % stride) : 0u;
        if (stride) {
            const int32_t want = (int32_t)(z->out_cap / stride) + 4;
            if (want > z->pc_tmp_cap) {
                free(z->pc_tmp);
                z->pc_tmp = (uint32_t *)malloc((size_t)want * sizeof(uint32_t));
                z->pc_tmp_cap = z->pc_tmp ? want : 0;
            }
        }
        /* (windows and gulps too small to be offered -- the tests' -- never ask the serial kernel for block boundaries either) */
        const int32_t par_on = mzh_stream_parallel() && z->in_len < ((int64_t)1 << 28) && mzh_stream_gulp() >= MZH_PAR_MIN_IN &&
                               z->out_cap >= 2 * (int64_t)MZH_PAR_MIN_ROOM;
        if (par_on && skip_par) {
            skip_par = 0; /* (asked ahead of time, at this header: nothing for it) */
            if (z->par_miss >= 3) {
                z->par_miss = 2;
                z->par_rest = 8;
            }
        } else if (par_on && z->par_rest > 0)
            z->par_rest--;
        else if (par_on && z->sst.bit == z->sst.hdr_bit && z->in_len >= MZH_PAR_MIN_IN && z->out_cap - z->out_len >= MZH_PAR_MIN_ROOM) {
            uint32_t pb = 0, pended = 0, pol = 0;
            /* how much of the input the search is shown: twice what the room took at the last window's ratio (the whole
             * gulp the first time).  Too little only ends the chain early; too much is searched and parsed for nothing */
            int64_t show = z->in_len;
            if (z->par_in_q16 > 0) {
                const int64_t est = (int64_t)(z->sst.hdr_bit >> 3) + (((z->out_cap - z->out_len) * z->par_in_q16) >> 15) + (256 << 10);
                if (est < show)
                    show = est;
            }
            const int64_t bit0 = z->sst.hdr_bit;
            const double t0 = mzh_now();
            uint32_t wcrc = 0, wadler = 1;
            const int32_t pr = mzhip_inflate_parallel_host(z->in, (uint32_t)show, z->out, (uint32_t)z->out_cap, &z->sst, &nst, &pol, &pb,
                                                           &pended, z->wrap == 2 ? &wcrc : NULL, z->wrap == 1 ? &wadler : NULL, seg_first,
                                                           z->pc_tmp_cap ? stride : 0u, z->pc_tmp, (uint32_t)z->pc_tmp_cap, &nseg);
            z->t_par += mzh_now() - t0;
            z->n_par++;
            if (pr < 0) {
                z->stream_end = 1;
                return verdict(z, MZH_STREAM_ERROR, z->in_dropped); /* device / runtime failure */
            }
            if (pb) {
                const int64_t made = (int64_t)pol - z->out_len;
                z->par_bytes += made;
                if (nseg)
                    stream_pieces_add(z, gnew, made, seg_first, stride, nseg);
                if (made > 0)
                    z->par_in_q16 = ((((int64_t)nst.hdr_bit - bit0) >> 3) << 16) / made + 1;
                stream_sum(z, made, wcrc, wadler);
                z->out_len = pol;
                z->sst = nst;
                if (pended) {
                    z->stream_end = 1;
                    return stream_finish(z, z->in_dropped + (((int64_t)nst.bit + 7) >> 3));
                }
                stream_drop_input(z);
                z->par_miss = made >= MZH_PAR_MIN_ROOM ? 0 : z->par_miss + 1;
            } else
                z->par_miss++;
            /* (next to) nothing for a wave of its own at this header: the serial kernel takes it from here.  Three such
             * windows in a row (fixed blocks throughout, say) and the search rests for a while */
            if (z->par_miss >= 3) {
                z->par_miss = 2;
                z->par_rest = 8;
            }
            if (pb) {
                if (z->out_len > z->out_served) {
                    lookahead_start(z, par_on); /* the next window, while this one is served */
                    return 0;
                }
                continue;
            }
            nseg = 0;
        }
        if (par_on && z->par_rest == 0 && z->par_miss < 3)
            z->sst.flags |= 2u; /* stop at the next block header: the many-wave decode goes on from there */
        const double ts0 = mzh_now();
        uint32_t adler = 1;
        mzhip_inflate_host_args wa;
        memset(&wa, 0, sizeof(wa));
        wa.size = (uint32_t)sizeof(wa);
        wa.in = z->in;
        wa.in_len = (uint32_t)z->in_len;
        wa.buf = z->out;
        wa.buf_cap = (uint32_t)z->out_cap;
        wa.state_in = &z->sst;
        wa.state_out = &nst;
        wa.out_len = &out_len;
        wa.in_used = &in_used;
        wa.crc = &crc;
        wa.adler = z->wrap == 1 ? &adler : NULL;
        wa.seg_first = seg_first;
        wa.seg_stride = z->pc_tmp_cap ? stride : 0u;
        wa.seg_crc = z->pc_tmp;
        wa.seg_cap = (uint32_t)z->pc_tmp_cap;
        wa.nseg = &nseg;
        int32_t st = mzhip_inflate_host_a(&wa);
        z->t_serial += mzh_now() - ts0;
        z->n_serial++;
        if (out_len > (uint32_t)z->out_len) {
            z->serial_bytes += (int64_t)out_len - z->out_len;
            stream_sum(z, (int64_t)out_len - z->out_len, crc, adler);
        }
        if (nseg)
            stream_pieces_add(z, gnew, (int64_t)out_len - z->out_len, seg_first, stride, nseg);
        if (st == MZHIP_STATUS_BUF_ERROR && z->base_eof && (nst.flags & 1u)) {
            /* the stream really ends short.  What this call decoded stays; then once more from where it stopped, without
             * the "all or nothing" rule of a resumable decode, so that what the reference would still have produced (the
             * available part of a stored block) is produced -- when the window has room for it */
            z->out_len = nst.out_pos;
            z->sst = nst;
            stream_drop_input(z);
            if (z->out_cap - z->out_len < 70000 && z->out_len > z->out_served)
                return 0; /* serve first; the next call slides the window and comes back here */
            z->sst.out_pos = (uint32_t)z->out_len;
            z->sst.flags = 1;
            wa.in_len = (uint32_t)z->in_len; /* (stream_drop_input moved the input) */
            wa.state_out = NULL;
            wa.seg_first = wa.seg_stride = wa.seg_cap = 0;
            wa.seg_crc = wa.nseg = NULL;
            st = mzhip_inflate_host_a(&wa);
            if (st == MZHIP_STATUS_OK || st == MZHIP_STATUS_BUF_ERROR || st == MZHIP_STATUS_DATA_ERROR) {
                stream_sum(z, (int64_t)out_len - z->out_len, crc, adler);
                z->out_len = out_len;
            } else
                st = MZH_STREAM_ERROR; /* (a full window cannot be: there was room for a block) */
            z->stream_end = 1;
            if (st == MZHIP_STATUS_OK)
                return stream_finish(z, z->in_dropped + in_used);
            return verdict(z, st, st == MZHIP_STATUS_BUF_ERROR ? z->in_dropped + z->in_len /* inflate() has taken all there was */
                                                               : z->in_dropped + in_used);
        }
        if (st == MZHIP_STATUS_OK || st == MZHIP_STATUS_DATA_ERROR) {
            z->out_len = out_len;
            z->stream_end = 1;
            return st == MZHIP_STATUS_OK ? stream_finish(z, z->in_dropped + in_used) : verdict(z, st, z->in_dropped + in_used);
        }
        if (st != MZHIP_STATUS_OUT_FULL && st != MZHIP_STATUS_BUF_ERROR) {
            z->stream_end = 1;
            return verdict(z, MZH_STREAM_ERROR, z->in_dropped + in_used); /* device / runtime failure */
        }
        if (!(nst.flags & 1u)) {
            z->stream_end = 1;
            return verdict(z, MZH_STREAM_ERROR, z->in_dropped);
        }
        const int64_t had = z->out_len;
        const int32_t moved = nst.hdr_bit != z->sst.hdr_bit || nst.bit != z->sst.bit; /* (an empty block in front of a header it was asked to stop at) */
        z->out_len = nst.out_pos;
        z->sst = nst;
        stream_drop_input(z);
        if (st == MZHIP_STATUS_OUT_FULL) {
            if (z->out_len > z->out_served)
                return 0; /* a window (or what was left of one) is there */
            if (z->out_len == had && !moved) { /* no room for even one token group: cannot happen with a 64 MiB window */
                z->stream_end = 1;
                return verdict(z, MZH_STREAM_ERROR, z->in_dropped);
            }
            continue;
        }
        /* input ended inside the window and the base stream has more: go on with it (what was decoded so far stays) */
        if (z->in_len >= gulp_now(z)) { /* a single block larger than the gulp: let the input buffer grow */
            int tries = 0;
            while (!z->base_eof && tries++ < 512) {
                const int32_t rd = pull_chunk(z);
                if (rd < 0) {
                    z->base_err = rd;
                    z->base_eof = 1;
                }
            }
        }
        if (z->out_len > z->out_served && z->out_len - z->out_served >= 65536)
            return 0; /* enough to serve while more input is fetched */
    }
}

/* from here on the entry is decoded window by window (out[] is a window already) */
static int32_t window_mode_begin(mzhip_zlib *z) {
    z->streaming = 1;
    z->t_begin = mzh_now();
    z->gulp_ramp = 4 * (int64_t)MZH_STREAM_EARLY;
    if (!z->in_pinned) { /* (room for a gulp and the pull that crosses it; grows like any in[] if a block needs more) */
        int64_t want = mzh_stream_gulp() + 2 * MZH_STAGING_BYTES;
        if (want < z->in_cap)
            want = z->in_cap;
        (void)in_grow(z, want);
    }
    z->in_dropped = 0;
    if (z->hdr_len > 0) { /* the wrapper's header is done with: positions stay those of the whole stream */
        memmove(z->in, z->in + z->hdr_len, (size_t)(z->in_len - z->hdr_len));
        z->in_len -= z->hdr_len;
        z->in_dropped = z->hdr_len;
    }
    z->run_crc = 0;
    z->run_adler = 1;
    z->run_n = 0;
    memset(&z->sst, 0, sizeof(z->sst));
    z->out_len = z->out_served = 0;
    const int32_t sr = stream_next(z);
    if (sr < 0)
        return sr;
    z->decoded = 1; /* (bytes are there; the verdict comes with the last window) */
    return 0;
}

static int32_t attempt_decode(mzhip_zlib *z) {
    if (z->wrap != 0 && z->hdr_len == 0) {
        const int32_t h = parse_wrapper_header(z);
        if (h == 0)
            return 0;
        if (h == 2) {
            z->next_attempt = z->in_len + 1;
            return 1;
        }
    }
    if (!z->payload_done && !z->streaming && mzh_stream_parallel() && z->in_len < mzh_stream_window() &&
        (z->max_total_in >= (int64_t)MZH_STREAM_EARLY + z->hdr_len || z->csize_hint >= (int64_t)MZH_STREAM_EARLY ||
         (!z->base_eof && z->in_len - z->hdr_len >= MZH_STREAM_EARLY - (MZH_STREAM_EARLY >> 6)))) { /* (pulls are 32 767 bytes: 8 of them are 262 136) */
        /* the stream is long enough for window mode -- the caller has said so (MZ_STREAM_PROP_TOTAL_IN_MAX: mz_zip.c sets it for
         * raw, stored and encrypted entries), or that much has been pulled and the attempts on the first 32, 64 and 128 KiB all
         * ran out of input: straight there.  (Up to round 6 the attempts went on to 1 MiB, each from the first byte, on one wave:
         * 10 of them, 0.16 of the 0.38 s a 1 GiB entry took, profiles/r6/large_entry_lookahead.log.) */
        if (z->out_cap < mzh_stream_window()) {
            out_release(z);
            z->out = out_window_alloc(z, mzh_stream_window());
            z->out_cap = z->out ? mzh_stream_window() : 0;
            if (!z->out)
                return MZH_MEM_ERROR;
        }
        return window_mode_begin(z);
    }
    while (!z->payload_done) {
        if (z->out_cap == 0) {
            z->out_cap = z->in_len * 8 + 65536; /* (a decode that runs out of room starts over in four times as much) */
            z->out = (uint8_t *)malloc((size_t)z->out_cap);
            if (!z->out)
                return MZH_MEM_ERROR;
        }
        uint32_t out_len = 0, in_used = 0;
        mzhip_inflate_host_args ea; /* the whole entry at once: no states */
        memset(&ea, 0, sizeof(ea));
        ea.size = (uint32_t)sizeof(ea);
        ea.in = z->in + z->hdr_len;
        ea.in_len = (uint32_t)(z->in_len - z->hdr_len);
        ea.buf = z->out;
        ea.buf_cap = (uint32_t)z->out_cap;
        ea.out_len = &out_len;
        ea.in_used = &in_used;
        ea.crc = &z->out_crc;
        ea.adler = z->wrap == 1 ? &z->out_adler : NULL;
        const double ta0 = mzh_now();
        int32_t st = mzhip_inflate_host_a(&ea);
        z->t_attempts += mzh_now() - ta0;
        z->n_attempts++;
        int32_t early = 0;
        const int64_t early_out = MZH_STREAM_EARLY_OUT < mzh_stream_window() ? MZH_STREAM_EARLY_OUT : mzh_stream_window();
        if (mzh_stream_parallel() && z->in_len < mzh_stream_window() &&
            ((st == MZHIP_STATUS_BUF_ERROR && !z->base_eof && (z->in_len >= MZH_STREAM_EARLY || (int64_t)out_len >= early_out)) ||
             (st == MZHIP_STATUS_OUT_FULL && z->out_cap >= early_out))) {
            /* a megabyte of compressed bytes and the entry is not over, or four of decoded ones and counting: it is large
             * enough for the many-wave decode, which lives in window mode.  (Without it the entry would be decoded from its
             * first byte again every time the input has doubled or the buffer has been outgrown, by one wave: 1.3 of the
             * 2.2 s of a 3 GiB entry that compresses 300:1, profiles/r4/large_entry.log.) */
            if (z->out_cap < mzh_stream_window()) {
                out_release(z);
                z->out = out_window_alloc(z, mzh_stream_window());
                z->out_cap = z->out ? mzh_stream_window() : 0;
                if (!z->out)
                    return MZH_MEM_ERROR;
            }
            early = z->out_cap >= mzh_stream_window();
        }
        if (early || (st == MZHIP_STATUS_OUT_FULL && z->out_cap >= mzh_stream_window()))
            return window_mode_begin(z); /* more than a window of output: the first window once more, this time asking where it stops */
        if (st == MZHIP_STATUS_OUT_FULL) {
            if (z->out_cap >= 0x7FFFFFFF)
                return MZH_MEM_ERROR;
            int64_t ncap = z->out_cap * 4;
            if (ncap > mzh_stream_window())
                ncap = mzh_stream_window();
            if (ncap > 0x7FFFFFFF)
                ncap = 0x7FFFFFFF;
            out_release(z);
            z->out = (ncap >= mzh_stream_window()) ? out_window_alloc(z, ncap) : (uint8_t *)malloc((size_t)ncap);
            if (!z->out) {
                z->out_cap = 0;
                return MZH_MEM_ERROR;
            }
            z->out_cap = ncap;
            continue;
        }
        if (st == MZHIP_STATUS_BUF_ERROR && !z->base_eof) {
            z->next_attempt = z->in_len * 2;
            return 1; /* input ended early, but base may have more */
        }
        z->out_len = out_len;
        if (st != MZHIP_STATUS_OK && st != MZHIP_STATUS_BUF_ERROR && st != MZHIP_STATUS_DATA_ERROR)
            return verdict(z, MZH_STREAM_ERROR, z->hdr_len + in_used); /* device/runtime failure: no CPU substitute */
        if (st == MZHIP_STATUS_BUF_ERROR)
            return verdict(z, st, z->in_len); /* input exhausted: inflate() has consumed every byte it was given (total_in) */
        if (st != MZHIP_STATUS_OK || z->wrap == 0)
            return verdict(z, st, z->hdr_len + in_used);
        z->dev_in_used = z->hdr_len + in_used;
        z->payload_done = 1;
    }
    /* wrapper trailer: gzip = CRC-32 then ISIZE, little endian; zlib = Adler-32, big endian */
    const int64_t tl = z->wrap == 2 ? 8 : 4;
    if (z->in_len - z->dev_in_used < tl) {
        if (!z->base_eof) {
            z->next_attempt = z->dev_in_used + tl;
            return 1;
        }
        if (z->wrap == 2 && z->in_len - z->dev_in_used >= 4 && le32(z->in + z->dev_in_used) != z->out_crc)
            return trailer_verdict(z, z->dev_in_used + 4); /* (cut inside ISIZE, the CRC field already wrong: stream_finish()) */
        return verdict(z, MZHIP_STATUS_BUF_ERROR, z->in_len);
    }
    const uint8_t *t = z->in + z->dev_in_used;
    if (z->wrap == 2) {
        if (le32(t) != z->out_crc) /* "incorrect data check": inflate() stops after the CRC field */
            return trailer_verdict(z, z->dev_in_used + 4);
        if (le32(t + 4) != (uint32_t)z->out_len) /* "incorrect length check" */
            return trailer_verdict(z, z->dev_in_used + 8);
        return verdict(z, MZHIP_STATUS_OK, z->dev_in_used + 8);
    }
    const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
    return want == z->out_adler ? verdict(z, MZHIP_STATUS_OK, z->dev_in_used + 4) : trailer_verdict(z, z->dev_in_used + 4);
}

int32_t mz_stream_zlib_read(void *stream, void *buf, int32_t size) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    mzhip_served_drop();
    if (z->error == 0 && mzhip_take_crc_fault() != 0)
        z->error = MZH_STREAM_ERROR; /* a checksum call before this one met a device failure */
    if (z->error != 0)
        return z->error; /* mz_strm_zlib.c:186-189 */
    z->read_stride = size >= 16384 ? size : 0; /* (window mode cuts its CRC pieces where these calls end) */

    if (!z->tried_cache && z->in_len == 0) {
        mzhip_stream *b = z->stream.base;
        if (b && b->vtbl && b->vtbl->tell && b->vtbl->is_open && b->vtbl->is_open(b) == MZH_OK)
            z->base_pos0 = b->vtbl->tell(b);
    }
    while (!z->decoded) {
        int32_t rd = pull_chunk(z);
        if (rd < 0) {
            /* mz_strm_zlib.c:148-149 returns a failing base read -- but the reference reads on demand and so never
             * issues a read its stream does not need, while this side reads ahead of the device's verdict (e.g. past
             * the last disk of a split archive, mz_strm_split.c:237-241).  What arrived is decoded first; the base
             * error is the result only if the stream really ends short. */
            if (z->in_len == 0 || z->base_err != 0)
                return rd;
            z->base_err = rd;
            z->base_eof = 1;
        }
        if (!z->tried_cache) {
            /* was this entry decoded by mzhip_prime_*()?  (payload offset + first payload bytes must agree) */
            z->tried_cache = 1;
            /* the caller has said how long the stream is: the first attempt waits for all of it (or for what sends it to window
             * mode) instead of asking the device at 32 KiB, 64 KiB, 128 KiB ... from the first byte each time */
            /* (... and never for more than half a window: an entry that has pulled a window's worth is not offered window mode) */
            int64_t first_at = (int64_t)MZH_STREAM_EARLY + 1024;
            if (first_at > mzh_stream_window() / 2)
                first_at = mzh_stream_window() / 2;
            if (z->max_total_in > 0 && z->next_attempt == 0)
                z->next_attempt = z->max_total_in < first_at ? z->max_total_in : first_at;
            mzhip_autoprime(z->stream.base, z->base_pos0); /* (shim_autoprime.c: on unless MZHIP_AUTOPRIME=0) */
            const uint8_t *data = NULL;
            int64_t usize = 0, csize = 0;
            uint32_t crc = 0;
            if (z->wrap == 0 && z->base_pos0 >= 0 &&
                mzhip_prime_lookup3(8, z->base_pos0, z->in, (int32_t)(z->in_len < 256 ? z->in_len : 256), z->max_total_in, &data,
                                    &usize, &csize, &crc, &z->seg_crc, &z->prime_pin, &z->hash_alg, &z->hash_digest) == 1) {
                z->out = (uint8_t *)(uintptr_t)data;
                z->out_borrowed = 1;
                z->out_len = usize;
                z->dev_in_used = csize;
                z->dev_status = 0;
                z->decoded = 1;
                break;
            }
            /* not primed, and the caller has not said how long the stream is: the local header in front of the payload may */
            if (z->max_total_in <= 0 && z->wrap == 0 && z->base_pos0 >= 30 && z->next_attempt == 0) {
                z->csize_hint = mzhip_lfh_csize_hint(z->stream.base, z->base_pos0);
                if (z->csize_hint > 0)
                    z->next_attempt = z->csize_hint < first_at ? z->csize_hint : first_at;
            }
        }
        if (!z->base_eof && z->in_len < z->next_attempt)
            continue;
        int32_t r = attempt_decode(z);
        if (r < 0) {
            z->error = r;
            return r;
        }
    }

    if (z->streaming) {
        /* window mode: the call is served across windows -- what the current one still holds, then the next one(s),
         * until `size` bytes are there or the stream is over (inflate() fills the caller's buffer the same way) */
        int32_t got = 0;
        for (;;) {
            const int64_t av = z->out_len - z->out_served;
            const int32_t k = (int32_t)(av < size - got ? av : size - got);
            if (k > 0) {
                uint32_t pcrc;
                if (got == 0 && z->pc_n > z->pc_head && stream_pieces_crc(z, k, &pcrc) && (k == size || av == k))
                    mzhip_served_set(buf, k, pcrc, z->out + z->out_served, z->slot); /* (dropped again below if the call goes on into the next window) */
                else if (got != 0)
                    mzhip_served_drop();
                memcpy((uint8_t *)buf + got, z->out + z->out_served, (size_t)k);
                z->out_served += k;
                z->total_out += k;
                got += k;
            }
            if (got == size || z->stream_end)
                break;
            const int32_t sr = stream_next(z);
            if (sr < 0) {
                z->error = sr;
                return sr;
            }
        }
        if (got == size && !z->stream_end && z->out_served == z->out_len) {
            /* the buffer is full exactly where a window ends: if the stream ends or fails right there, inflate() may have found
             * that in this call (verdict_needs_no_room) -- look.  (The window slides: a CRC hint that points into it is withdrawn) */
            mzhip_served_drop();
            const int32_t sr = stream_next(z);
            if (sr < 0) {
                z->error = sr;
                return sr;
            }
        }
        if (z->dev_status != 0 && (got < size || (z->stream_end && z->out_served == z->out_len && verdict_needs_no_room(z, z->g0 + z->out_len)))) {
            /* the failing call reports the error, not a byte count (mz_strm_zlib.c:186-189) */
            z->error = (z->base_err != 0 && z->dev_status == MZHIP_STATUS_BUF_ERROR) ? z->base_err : z->dev_status;
            z->total_in = z->dev_in_used;
            return z->error;
        }
        /* TOTAL_IN is exact once the stream end has been served; before that it is a lower bound (the blocks that are
         * done with; with the end in sight, all but the last byte -- the rule of the one-buffer path below) */
        z->total_in = !z->stream_end ? z->in_dropped : (z->out_served == z->out_len ? z->dev_in_used : z->dev_in_used - 1);
        return got;
    }
    int64_t avail = z->out_len - z->out_served;
    if (z->dev_status != 0 && (avail < size || (avail == size && verdict_needs_no_room(z, z->out_len)))) {
        /* the failing call reports the error, not a byte count (mz_strm_zlib.c:186-189).  (A failed trailer check is met
         * in the call that returns the payload's last byte even when that byte fills the buffer: behind it inflate() walks
         * through the end-of-block code and the trailer without needing room -- round 5, tests/fuzz_wrappers.py) */
        z->error = (z->base_err != 0 && z->dev_status == MZHIP_STATUS_BUF_ERROR) ? z->base_err : z->dev_status;
        z->total_in = z->dev_in_used;
        z->total_out = z->out_len;
        return z->error;
    }
    int32_t n = (int32_t)(avail < size ? avail : size);
    if (n > 0) {
        memcpy(buf, z->out + z->out_served, (size_t)n);
        if (z->out_borrowed && z->out_served % MZHIP_PRIME_SEGMENT == 0 &&
            (n == MZHIP_PRIME_SEGMENT || z->out_served + n == z->out_len)) {
            /* a whole primed segment: remember its device-computed CRC for the mz_crypt_crc32_update that follows */
            mzhip_served_set(buf, n, z->seg_crc[z->out_served / MZHIP_PRIME_SEGMENT], z->out + z->out_served, z->slot);
            mzhip_served_set_entry(z->out, z->out_served, z->out_len, z->hash_alg, z->hash_digest);
        }
        z->out_served += n;
        z->total_out += n;
    }
    if (z->out_served == z->out_len) {
        z->total_in = z->dev_in_used; /* the stream end has been reached: exact TOTAL_IN */
    } else {
        /* mid-stream: the reference would have consumed only part of the input */
        int64_t est = z->out_len ? (z->dev_in_used * z->out_served) / z->out_len : 0;
        if (est >= z->dev_in_used)
            est = z->dev_in_used - 1;
        z->total_in = est;
    }
    return n;
}

/* what mz_stream_write does before dispatching (mz_strm.c:101-110) */
static int32_t base_write(mzhip_stream *base, const void *buf, int32_t size) {
    if (size == 0)
        return size;
    if (!base || !base->vtbl || !base->vtbl->write)
        return MZH_PARAM_ERROR;
    if (!base->vtbl->is_open || base->vtbl->is_open(base) != MZH_OK)
        return MZH_STREAM_ERROR;
    return base->vtbl->write(base, buf, size);
}

/* Compress what has been collected (device K4) and push it to base in staging-sized writes
 * (the reference flushes its 32767-byte buffer the same way, mz_strm_zlib.c:196-201,211-219). */
static int32_t push_to_base(mzhip_zlib *z, const uint8_t *out, uint32_t out_len) {
    uint32_t pos = 0;
    while (pos < out_len) {
        int32_t n = (int32_t)(out_len - pos < MZH_STAGING_BYTES ? out_len - pos : MZH_STAGING_BYTES);
        if (base_write(z->stream.base, out + pos, n) != n)
            return MZH_WRITE_ERROR; /* mz_strm_zlib.c:198-199 */
        pos += (uint32_t)n;
    }
    z->total_out += out_len;
    return MZH_OK;
}

#ifndef MZH_WRITE_SEGMENT
#define MZH_WRITE_SEGMENT (8 << 20) /* bytes collected per device launch (128 pieces of 64 KiB) */
#endif
static int8_t mzh_wo_mode = -1;
MZHIP_API void mzhip_set_write_overlap(int32_t on) { __atomic_store_n(&mzh_wo_mode, (int8_t)(on ? 1 : 0), __ATOMIC_RELAXED); }
static int32_t mzh_write_overlap(void) {
    int8_t m = __atomic_load_n(&mzh_wo_mode, __ATOMIC_RELAXED);
    if (m < 0) {
        const char *e = getenv("MZHIP_WRITE_OVERLAP"); /* "0": a full segment is coded before write() returns, as before round 6 */
        m = (e && e[0] == '0') ? 0 : 1;
        __atomic_store_n(&mzh_wo_mode, m, __ATOMIC_RELAXED);
    }
    return m;
}

/* one segment through the device: in[0 .. in_len) -> out */
static int32_t segment_encode(const mzhip_zlib *z, const uint8_t *in, int64_t in_len, int32_t final, uint8_t *out, uint32_t cap, uint32_t *out_len,
                              uint32_t *crc, uint32_t *adler) {
    mzhip_deflate_host_args da; /* level and window as mz_strm_zlib.c:87 hands them on */
    memset(&da, 0, sizeof(da));
    da.size = (uint32_t)sizeof(da);
    da.in = in;
    da.in_len = (uint32_t)in_len;
    da.final = (uint32_t)final;
    da.level = z->level;
    da.window_log2 = z->wlog;
    da.out = out;
    da.out_cap = cap;
    da.out_len = out_len;
    da.crc = crc;
    da.adler = z->wrap == 1 ? adler : NULL;
    return mzhip_deflate_host_a(&da);
}
static uint32_t segment_cap(int64_t n) { return (uint32_t)(n + n / 8 + 128 + (n / 65536 + 1) * 80); }
/* ... and what follows it: the running wrapper checksums (arithmetic on the device-computed segment checksums only), the bytes to the base stream */
static int32_t segment_done(mzhip_zlib *z, int32_t st, int64_t in_len, const uint8_t *out, uint32_t out_len, uint32_t crc, uint32_t adler) {
    if (st != 0) {
        z->error = MZH_STREAM_ERROR; /* device failure: never substitute a CPU result */
        return MZH_DATA_ERROR;       /* mz_strm_zlib.c:233-236 */
    }
    z->w_crc = z->w_total == 0 ? crc : mzhip_crc32_combine(z->w_crc, crc, (uint64_t)in_len);
    if (z->wrap == 1)
        z->w_adler = mzhip_adler32_combine(z->w_adler, adler, (uint64_t)in_len);
    z->w_total += in_len;
    return push_to_base(z, out, out_len);
}

static void *write_ahead_run(void *arg) {
    mzhip_zlib *z = (mzhip_zlib *)arg;
    int32_t dev = -2;
    pthread_mutex_lock(&z->w_mu);
    for (;;) {
        while (z->w_state != 1 && z->w_state != 3)
            pthread_cond_wait(&z->w_cv, &z->w_mu);
        if (z->w_state == 3)
            break;
        pthread_mutex_unlock(&z->w_mu);
        if (z->wjob.device >= 0 && z->wjob.device != dev) {
            dev = z->wjob.device;
            mzhip_thread_use_device(dev);
        }
        z->wjob.st = segment_encode(z, z->wjob.in, z->wjob.in_len, 0, z->wjob.out, z->wjob.out_cap, &z->wjob.out_len, &z->wjob.crc, &z->wjob.adler);
        pthread_mutex_lock(&z->w_mu);
        z->w_state = 2;
        pthread_cond_broadcast(&z->w_cv);
    }
    pthread_mutex_unlock(&z->w_mu);
    return NULL;
}
/* the segment that is on its way: wait for it, hand its bytes on */
static int32_t write_ahead_finish(mzhip_zlib *z) {
    if (!z->w_pending)
        return MZH_OK;
    pthread_mutex_lock(&z->w_mu);
    while (z->w_state == 1)
        pthread_cond_wait(&z->w_cv, &z->w_mu);
    z->w_state = 0;
    pthread_mutex_unlock(&z->w_mu);
    z->w_pending = 0;
    return segment_done(z, z->wjob.st, z->wjob.in_len, z->wjob.out, z->wjob.out_len, z->wjob.crc, z->wjob.adler);
}
/* (close, delete, re-open: a segment that is still on its way reads wjob.in and writes wjob.out) */
static void write_ahead_stop(mzhip_zlib *z) {
    if (z->w_pending) {
        pthread_mutex_lock(&z->w_mu);
        while (z->w_state == 1)
            pthread_cond_wait(&z->w_cv, &z->w_mu);
        z->w_state = 0;
        pthread_mutex_unlock(&z->w_mu);
        z->w_pending = 0;
    }
    if (z->w_started) {
        pthread_mutex_lock(&z->w_mu);
        z->w_state = 3;
        pthread_cond_broadcast(&z->w_cv);
        pthread_mutex_unlock(&z->w_mu);
        pthread_join(z->w_thread, NULL);
        pthread_mutex_destroy(&z->w_mu);
        pthread_cond_destroy(&z->w_cv);
        z->w_started = 0;
        z->w_state = 0;
    }
    free(z->wjob.in);
    free(z->wjob.out);
    z->wjob.in = z->wjob.out = NULL;
    z->wjob.out_cap = 0;
}
/* the full segment in wbuf goes to the stream's thread, the caller gets the other buffer: 1 = done, 0 = not possible (the caller codes it itself) */
static int32_t write_ahead_post(mzhip_zlib *z) {
    if (!mzh_write_overlap() || z->wlen < (int64_t)MZH_WRITE_SEGMENT)
        return 0;
    if (!z->wjob.in)
        z->wjob.in = (uint8_t *)malloc(MZH_WRITE_SEGMENT);
    const uint32_t cap = segment_cap(MZH_WRITE_SEGMENT);
    if (!z->wjob.out) {
        z->wjob.out = (uint8_t *)malloc(cap);
        z->wjob.out_cap = z->wjob.out ? cap : 0;
    }
    if (!z->wjob.in || !z->wjob.out)
        return 0;
    if (!z->w_started) {
        if (pthread_mutex_init(&z->w_mu, NULL) != 0)
            return 0;
        if (pthread_cond_init(&z->w_cv, NULL) != 0) {
            pthread_mutex_destroy(&z->w_mu);
            return 0;
        }
        z->w_state = 0;
        if (pthread_create(&z->w_thread, NULL, write_ahead_run, z) != 0) {
            pthread_mutex_destroy(&z->w_mu);
            pthread_cond_destroy(&z->w_cv);
            return 0;
        }
        z->w_started = 1;
    }
    uint8_t *t = z->wbuf;
    z->wbuf = z->wjob.in;
    z->wjob.in = t;
    z->wjob.in_len = z->wlen;
    z->wjob.out_len = 0;
    z->wjob.crc = 0;
    z->wjob.adler = 1;
    z->wjob.st = 0;
    z->wjob.device = mzhip_prime_current_device();
    z->wlen = 0;
    pthread_mutex_lock(&z->w_mu);
    z->w_state = 1;
    pthread_cond_broadcast(&z->w_cv);
    pthread_mutex_unlock(&z->w_mu);
    z->w_pending = 1;
    return 1;
}

static int32_t flush_segment(mzhip_zlib *z, int32_t final) {
    if (z->wlen == 0 && !final && !z->w_pending)
        return MZH_OK;
    if (z->wrap != 0 && !z->w_header_done) {
        /* the header deflate() emits when no gz_header was set: gzip = magic, CM 8, no flags, no mtime, XFL from
         * the level (2 = best, 4 = fastest), OS 3; zlib = CM 8 + CINFO, level class in FLG bits 7:6, FCHECK */
        uint8_t h[10] = {0x1F, 0x8B, 8, 0, 0, 0, 0, 0, 0, 3};
        uint32_t hl = 10;
        if (z->wrap == 2) {
            h[8] = (uint8_t)(z->level == 9 ? 2 : (z->level >= 0 && z->level < 2) ? 4 : 0);
        } else {
            const int32_t lv = z->level < 0 ? 6 : z->level;
            uint32_t w = ((0x08u | ((uint32_t)(z->wlog - 8) << 4)) << 8) | ((uint32_t)(lv < 2 ? 0 : lv < 6 ? 1 : lv == 6 ? 2 : 3) << 6); /* CINFO = log2(window) - 8 */
            w += 31u - w % 31u;
            h[0] = (uint8_t)(w >> 8);
            h[1] = (uint8_t)w;
            hl = 2;
        }
        z->w_header_done = 1;
        if (push_to_base(z, h, hl) != MZH_OK)
            return MZH_WRITE_ERROR;
    }
    /* the segment before this one first: its bytes go out in front of this one's */
    int32_t err = write_ahead_finish(z);
    if (err != MZH_OK)
        return err;
    if (z->wlen == 0 && !final)
        return MZH_OK;
    if (!final && write_ahead_post(z))
        return MZH_OK; /* (coded while the caller fills the next segment; handed on by the next flush) */
    const uint32_t cap = segment_cap(z->wlen);
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out)
        return MZH_MEM_ERROR;
    uint32_t out_len = 0, crc = 0, adler = 1;
    const int32_t st = segment_encode(z, z->wbuf, z->wlen, final, out, cap, &out_len, &crc, &adler);
    err = segment_done(z, st, z->wlen, out, out_len, crc, adler);
    free(out);
    if (err != MZH_OK)
        return err;
    z->wlen = 0;
    if (final && z->wrap == 2) {
        const uint32_t isz = (uint32_t)z->w_total;
        uint8_t t[8] = {(uint8_t)z->w_crc, (uint8_t)(z->w_crc >> 8), (uint8_t)(z->w_crc >> 16), (uint8_t)(z->w_crc >> 24),
                        (uint8_t)isz,      (uint8_t)(isz >> 8),      (uint8_t)(isz >> 16),      (uint8_t)(isz >> 24)};
        return push_to_base(z, t, 8);
    }
    if (final && z->wrap == 1) {
        uint8_t t[4] = {(uint8_t)(z->w_adler >> 24), (uint8_t)(z->w_adler >> 16), (uint8_t)(z->w_adler >> 8),
                        (uint8_t)z->w_adler};
        return push_to_base(z, t, 4);
    }
    return MZH_OK;
}

/* bytes into the segment buffer, launching whenever it is full */
static int32_t collect(mzhip_zlib *z, const uint8_t *p, int64_t left) {
    while (left > 0) {
        if (z->wcap == 0) {
            z->wbuf = (uint8_t *)malloc(MZH_WRITE_SEGMENT);
            if (!z->wbuf)
                return MZH_MEM_ERROR;
            z->wcap = MZH_WRITE_SEGMENT;
        }
        int64_t room = z->wcap - z->wlen;
        int64_t n = left < room ? left : room;
        memcpy(z->wbuf + z->wlen, p, (size_t)n);
        z->wlen += n;
        p += n;
        left -= n;
        if (z->wlen == z->wcap) {
            int32_t err = flush_segment(z, 0);
            if (err != MZH_OK)
                return err;
        }
    }
    return MZH_OK;
}

/* the entry stopped following its primed buffer: what it shared with it goes down the ordinary path */
static int32_t leave_primed(mzhip_zlib *z) {
    int32_t err = MZH_OK;
    if (z->wp_id >= 0) {
        const uint8_t *src = NULL, *out = NULL;
        uint32_t out_len = 0;
        (void)mzhip_wprime_result(8, z->wp_id, -1, &src, &out, &out_len);
        if (!src)
            return MZH_INTERNAL_ERROR; /* the cache was cleared under a stream that was following it */
        err = collect(z, src, z->wp_pos);
    }
    z->wp_id = -1;
    z->wp_off = 1;
    return err;
}

int32_t mz_stream_zlib_write(void *stream, const void *buf, int32_t size) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    mzhip_served_drop();
    if (mzhip_take_crc_fault() != 0) { /* a checksum call before this one met a device failure */
        z->error = MZH_STREAM_ERROR;
        return MZH_STREAM_ERROR;
    }
    if (size > 0 && z->wrap == 0 && !z->wp_off) {
        /* mzhip_prime_write: is this entry, so far, one of the buffers that were compressed ahead of time? */
        uint32_t crc = 0;
        int32_t have_crc = 0;
        const uint8_t *wsrc = NULL;
        if ((z->wp_id >= 0 || z->total_in == 0) &&
            mzhip_wprime_track(8, &z->wp_id, z->wp_pos, (const uint8_t *)buf, size, &crc, &have_crc, &wsrc) == 1) {
            z->wp_pos += size;
            z->total_in += size;
            if (have_crc) { /* the mz_crypt_crc32_update that follows (mz_zip.c:2062-2064) is answered from the cache */
                mzhip_served_set(buf, size, crc, wsrc, z->slot);
            }
            return size;
        }
        int32_t err = leave_primed(z);
        if (err != MZH_OK)
            return err;
    }
    int32_t err = collect(z, (const uint8_t *)buf, size);
    if (err != MZH_OK)
        return err;
    z->total_in += size; /* mz_strm_zlib.c:261 */
    return size;
}

int64_t mz_stream_zlib_tell(void *stream) {
    (void)stream;
    return MZH_TELL_ERROR;
}

int32_t mz_stream_zlib_seek(void *stream, int64_t offset, int32_t origin) {
    (void)stream;
    (void)offset;
    (void)origin;
    return MZH_SEEK_ERROR;
}

int32_t mz_stream_zlib_close(void *stream) {
    mzhip_served_drop(); /* (the hint points into a primed generation this stream pins) */
    mzhip_buffers_released(((mzhip_zlib *)stream)->slot);
    mzhip_zlib *z = (mzhip_zlib *)stream;
    if (z->mode & MZH_OPEN_MODE_WRITE) {
        const uint8_t *src = NULL, *out = NULL;
        uint32_t out_len = 0;
        if (z->wp_id >= 0 && mzhip_wprime_result(8, z->wp_id, z->wp_pos, &src, &out, &out_len) == 1) {
            push_to_base(z, out, out_len); /* the entry is exactly a primed buffer: its stream was coded in the batch */
        } else {
            if (z->wp_id >= 0)
                leave_primed(z); /* a proper prefix of a primed buffer */
            flush_segment(z, 1); /* deflate(Z_FINISH) + flush, return value ignored like mz_strm_zlib.c:287-288 */
        }
        z->wp_id = -1;
    }
    if (z->streaming && (z->mode & MZH_OPEN_MODE_READ)) {
        const char *e = getenv("MZHIP_STREAM_STATS");
        if (e && e[0] == '1')
            fprintf(stderr,
                    "mzhip window mode: %.3f s open to close (window mode from %.3f s on, after %d whole-entry attempts that took %.3f s), %lld bytes out; many-wave windows %d (%.3f s, %lld bytes), serial windows %d "
                    "(%.3f s, %lld bytes), pulling input %.3f s; %d windows decoded ahead of the reader, which waited %.3f s for them (%.3f s of its own "
                    "between a window's start and that wait, %.3f s taking windows over and starting the next)\n",
                    mzh_now() - z->t_open, z->t_begin - z->t_open, z->n_attempts, z->t_attempts, (long long)z->total_out, z->n_par, z->t_par, (long long)z->par_bytes, z->n_serial, z->t_serial,
                    (long long)z->serial_bytes, z->t_pull, z->n_ahead, z->t_wait, z->t_overlap, z->t_book);
    }
    z->initialized = 0;
    lookahead_cancel(z); /* (a window may still be on its way: its thread reads in[]) */
    in_release(z);
    free(z->pc);
    free(z->pc_tmp);
    z->pc = NULL;
    z->pc_tmp = NULL;
    z->pc_n = z->pc_cap = z->pc_head = z->pc_tmp_cap = 0;
    z->g0 = 0;
    out_release(z);
    mzhip_prime_unpin(z->prime_pin);
    z->prime_pin = NULL;
    z->out_borrowed = 0;
    write_ahead_stop(z); /* (flush_segment(final) has waited for the last segment; this ends the thread) */
    free(z->wbuf);
    z->in = z->out = z->wbuf = NULL;
    z->in_cap = z->out_cap = z->wcap = 0;
    if (z->error != 0)
        return MZH_CLOSE_ERROR; /* mz_strm_zlib.c:302-303 */
    return MZH_OK;
}

int32_t mz_stream_zlib_error(void *stream) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    return z->error;
}

int32_t mz_stream_zlib_get_prop_int64(void *stream, int32_t prop, int64_t *value) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    switch (prop) {
    case MZH_PROP_TOTAL_IN:
        *value = z->total_in;
        break;
    case MZH_PROP_TOTAL_IN_MAX:
        *value = z->max_total_in;
        break;
    case MZH_PROP_TOTAL_OUT:
        *value = z->total_out;
        break;
    case MZH_PROP_HEADER_SIZE:
        *value = 0;
        break;
    case MZH_PROP_COMPRESS_WINDOW:
        *value = z->window_bits;
        break;
    default:
        return MZH_EXIST_ERROR;
    }
    return MZH_OK;
}

int32_t mz_stream_zlib_set_prop_int64(void *stream, int32_t prop, int64_t value) {
    mzhip_zlib *z = (mzhip_zlib *)stream;
    switch (prop) {
    case MZH_PROP_COMPRESS_LEVEL:
        z->level = (int16_t)value; /* -1 stays -1 == Z_DEFAULT_COMPRESSION (mz_strm_zlib.c:339-343) */
        break;
    case MZH_PROP_TOTAL_IN_MAX:
        z->max_total_in = value;
        break;
    case MZH_PROP_COMPRESS_WINDOW:
        z->window_bits = (int32_t)value;
        break;
    default:
        return MZH_EXIST_ERROR;
    }
    return MZH_OK;
}

void *mz_stream_zlib_create(void) {
    mzhip_zlib *z = (mzhip_zlib *)calloc(1, sizeof(mzhip_zlib));
    if (z) {
        z->slot = mzhip_stream_slot_new();
        z->stream.vtbl = &mzhip_zlib_vtbl;
        z->level = -1;
        z->window_bits = -15;
    }
    return z;
}

void mz_stream_zlib_delete(void **stream) {
    mzhip_served_drop();
    if (stream && *stream)
        mzhip_buffers_released(((mzhip_zlib *)*stream)->slot);
    mzhip_zlib *z;
    if (!stream)
        return;
    z = (mzhip_zlib *)*stream;
    if (z) {
        free_buffers(z); /* delete without close is legal (the reference only leaks there): every buffer once, pins dropped */
        free(z);
    }
    *stream = NULL;
}

void *mz_stream_zlib_get_interface(void) {
    return (void *)&mzhip_zlib_vtbl;
}
