/* shim_lzma.c -- mz_stream_lzma re-implemented over the HIP backend.
 *
 * Drop-in for the reference's mz_strm_lzma.c (13 exported symbols,
 * mz_strm_lzma.h:20-35).  READ of method 14 (raw LZMA1 behind the ZIP-LZMA
 * header) is decoded by the device range decoder (K3, lzma_core.h) through
 * mzhip_lzma_host(); READ of method 95 (one .xz stream of LZMA2 blocks, what
 * lzma_stream_decoder handles at mz_strm_lzma.c:127-128) by the .xz kernel
 * (xz_core.h) through mzhip_xz_host().  WRITE collects the entry and codes it at
 * close() on the device (lzma_enc_core.h): method 14 as one LZMA1 stream with the
 * ZIP-LZMA header and the end marker (mz_strm_lzma.c:94-104, mz_zip.c:1984), method
 * 95 as one .xz stream whose LZMA2 chunks are coded independently (state and properties reset) over one dictionary.  The bytes are valid but not
 * liblzma's; the reference's reader decodes them back (tests/test_gpu_lzma_enc.py).
 *
 * Contract mirrored from the reference (file:line = mz_strm_lzma.c):
 *   create :429-438  method LZMA, preset default, max_total_out -1
 *   open   :52-138   READ/LZMA consumes the 4-byte magic from base and counts
 *                    it in TOTAL_IN (:118-124)
 *   read   :147-241  staging pulls <=32767 B clamped by TOTAL_IN_MAX (:170-174);
 *                    TOTAL_OUT clamped by TOTAL_OUT_MAX (:214-215); ANY decoder
 *                    error is returned as MZ_DATA_ERROR (:236-237); error()
 *                    keeps the decoder's own code (here: 9 data / 10 buf, the
 *                    lzma_ret values LZMA_DATA_ERROR / LZMA_BUF_ERROR)
 *   props  :380-428  adds COMPRESS_METHOD, TOTAL_OUT_MAX (>= -1), HEADER_SIZE=4
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mz_strm_hip.h"
#include "mzhip.h"
#include "shim_common.h"

#define LZMA_MAGIC_SIZE 4 /* mz_strm_lzma.c:19 */

typedef struct mzhip_lzma_s {
    mzhip_stream stream;
    int32_t mode;
    int32_t error;
    int8_t initialized;
    int16_t method;
    uint32_t preset;
    int64_t total_in, total_out, max_total_in, max_total_out;
    uint8_t *in;
    int64_t in_len, in_cap;
    int8_t base_eof;
    int32_t base_err; /* base failed after data had arrived: reported only if the stream turns out to need more */
    /* write side, mzhip_prime_write: the entry so far equals bytes [0, wp_pos) of primed buffer wp_id */
    int64_t wp_id, wp_pos;
    int8_t wp_off;
    uint8_t *out;
    int64_t out_len, out_cap, out_served;
    int8_t decoded;
    int32_t dev_status;
    int64_t dev_in_used;
    int64_t next_attempt;
    int8_t tried_cache, out_borrowed; /* prime cache: looked up once; out points into the cache */
    void *prime_pin;                  /* keeps the cached generation alive while out points into it */
    int64_t base_pos0;                /* base position at open = payload offset */
    const uint32_t *seg_crc;          /* GPU CRCs of the 65 535-byte segments of a primed entry */
    /* write side: the whole entry is collected, coded at close() */
    uint8_t *wbuf;
    int64_t wlen, wcap;
    uint32_t slot; /* this stream's cell of mzhip_stream_epoch[] (shim_common.h) */
    uint16_t hash_alg;          /* a primed entry's device-verified digest (shim_sha.c answers mz_crypt_sha_end with it) */
    const uint8_t *hash_digest;
    /* read side, method 14, entries larger than one window (mzh_stream_window): decoded window by window by the resumable
     * build of K3.  out[] then holds [the dictionary so far | the window's bytes]; consumed input is dropped */
    int8_t streaming;      /* window mode is on */
    int8_t resumed;        /* lst is a state to go on from */
    int8_t stream_end;     /* the end marker has been decoded */
    int64_t ghost;         /* method 14 in windows: bytes behind TOTAL_OUT_MAX the reference has decoded into its callers' buffers by now */
    int32_t s_err;         /* the device's verdict once the stream cannot go on: served when the bytes in front of it are */
    mzhip_lzma_state lst;
    void *model;           /* mzhip_lzma_model_bytes(): the adaptive model between calls */
    int64_t hist;          /* bytes of dictionary at the front of out[] */
    int64_t out_abs;       /* position in the decoded stream of out[0] (a multiple of 16) */
    int64_t in_dropped;    /* compressed bytes consumed and dropped from the front of in[] */
    int64_t dict_keep;     /* dictionary bytes kept between windows */
    /* read side, method 95 in window mode (streaming == 2): the container is walked here, a block's LZMA2 chunks are decoded
     * window by window by mzhip_lzma2_run_host */
    int8_t xz_phase;       /* XZP_* */
    int8_t xz_no_stream;   /* this stream cannot be read in windows (filters, a SHA-256 check): the one-buffer path has it */
    uint8_t xz_check;      /* the stream flags' check id (0 none, 1 CRC-32, 4 CRC-64) */
    mzhip_lzma2_state xst;
    uint64_t xz_blk_hsize, xz_blk_csize, xz_blk_usize; /* this block: header bytes, chunk bytes consumed, bytes decoded */
    uint64_t xz_want_csize, xz_want_usize;             /* what the block header promises (~0: nothing) */
    uint64_t xz_nblocks, xz_digest;                    /* blocks so far and a running hash of their (unpadded size, size) records */
    uint64_t xz_index_size;
    /* liblzma walks on through block padding, check, the next block header, index and footer in the lzma_code() call that
     * made a block's last bytes -- as far as the 32 767 bytes staged at the time reach (mz_strm_lzma.c:170-210) -- so what it
     * finds wrong there fails the read() that would have returned those bytes, and TOTAL_IN stands where the staging ends */
    int8_t xz_err_eager;   /* s_err was met on that walk */
    int8_t xz_tail_tried;  /* TOTAL_OUT_MAX reached, a later read() has walked the rest of the container */
    int64_t xz_tin_hold;   /* >= 0: TOTAL_IN to report while the walk waits for the next staging buffer */
    /* write side, method 14, entries larger than one segment: coded segment by segment (mzhip_lzma_encode_resume_host), wbuf[]
     * then holds [the previous segment's last 64 KiB | bytes not coded yet] */
    int32_t w_segments;    /* segments coded so far */
    int64_t w_hist;        /* bytes at the front of wbuf[] that are history (whole blocks, at most mzhip_lzma_encode_history_bytes()) */
    mzhip_lzma_enc_state west;
    /* ... and method 95: one .xz block per segment; the index at close() wants every block's sizes */
    uint64_t *xz_unpadded, *xz_usize;
    int32_t xz_cap;
} mzhip_lzma;

static mzhip_stream_vtbl mzhip_lzma_vtbl = {
    mz_stream_lzma_open,   mz_stream_lzma_is_open, mz_stream_lzma_read,           mz_stream_lzma_write,
    mz_stream_lzma_tell,   mz_stream_lzma_seek,    mz_stream_lzma_close,          mz_stream_lzma_error,
    mz_stream_lzma_create, mz_stream_lzma_delete,  mz_stream_lzma_get_prop_int64, mz_stream_lzma_set_prop_int64};

static int32_t base_read(mzhip_stream *base, void *buf, int32_t size) {
    if (!base || !base->vtbl || !base->vtbl->read)
        return MZH_PARAM_ERROR;
    if (!base->vtbl->is_open || base->vtbl->is_open(base) != MZH_OK)
        return MZH_STREAM_ERROR;
    return base->vtbl->read(base, buf, size);
}

static int32_t grow_in(mzhip_lzma *z, int64_t need) {
    if (need <= z->in_cap)
        return MZH_OK;
    int64_t ncap = z->in_cap ? z->in_cap * 2 : (need <= 1024 ? 1024 : 65536);
    while (ncap < need)
        ncap *= 2;
    uint8_t *p = (uint8_t *)realloc(z->in, (size_t)ncap);
    if (!p)
        return MZH_MEM_ERROR;
    z->in = p;
    z->in_cap = ncap;
    return MZH_OK;
}

int32_t mz_stream_lzma_open(void *stream, const char *path, int32_t mode) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    mzhip_served_drop();
    mzhip_buffers_released(((mzhip_lzma *)stream)->slot);
    (void)path;
    z->total_in = z->total_out = 0;
    z->error = 0;
    free(z->in);
    if (!z->out_borrowed)
        free(z->out);
    mzhip_prime_unpin(z->prime_pin);
    z->prime_pin = NULL;
    z->out_borrowed = 0;
    z->tried_cache = 0;
    z->hash_alg = 0;
    z->hash_digest = NULL;
    z->seg_crc = NULL;
    z->base_pos0 = -1;
    z->in = z->out = NULL;
    z->in_len = z->in_cap = z->out_len = z->out_cap = z->out_served = 0;
    z->base_eof = z->decoded = 0;
    z->base_err = 0;
    z->wp_id = -1;
    z->wp_pos = 0;
    z->wp_off = 0;
    z->dev_status = 0;
    z->dev_in_used = 0;
    z->next_attempt = 0;
    z->streaming = z->resumed = z->stream_end = 0;
    z->s_err = 0;
    z->hist = z->out_abs = z->in_dropped = z->dict_keep = 0;
    z->xz_phase = z->xz_no_stream = z->xz_err_eager = z->xz_tail_tried = 0;
    z->xz_tin_hold = -1;
    free(z->model);
    z->model = NULL;
    z->w_segments = 0;
    z->w_hist = 0;
    free(z->xz_unpadded);
    free(z->xz_usize);
    z->xz_unpadded = z->xz_usize = NULL;
    z->xz_cap = 0;
    free(z->wbuf);
    z->wbuf = NULL;
    z->wlen = z->wcap = 0;
    if (mode & MZH_OPEN_MODE_WRITE) {
        if (z->method != MZH_COMPRESS_METHOD_LZMA && z->method != MZH_COMPRESS_METHOD_XZ)
            return MZH_OPEN_ERROR;
        if (mzhip_device_count() <= 0) {
            z->error = 1;
            return MZH_OPEN_ERROR;
        }
    } else if (mode & MZH_OPEN_MODE_READ) {
        if (z->method != MZH_COMPRESS_METHOD_LZMA && z->method != MZH_COMPRESS_METHOD_XZ)
            return MZH_OPEN_ERROR; /* neither decoder initialised: lzma->error stays non-OK, mz_strm_lzma.c:131-132 */
        if (mzhip_device_count() <= 0) {
            z->error = 1;
            return MZH_OPEN_ERROR;
        }
        {
            mzhip_stream *b = z->stream.base;
            if (b && b->vtbl && b->vtbl->tell && b->vtbl->is_open && b->vtbl->is_open(b) == MZH_OK)
                z->base_pos0 = b->vtbl->tell(b); /* the payload offset, for the prime cache */
        }
        /* the 4 magic bytes are consumed at open and kept as the front of the
         * device input, which starts at the ZIP-LZMA header (mz_strm_lzma.c:118-124) */
        if (grow_in(z, 65536) != MZH_OK)
            return MZH_OPEN_ERROR;
        for (int i = 0; z->method == MZH_COMPRESS_METHOD_LZMA && i < LZMA_MAGIC_SIZE; i++) {
            uint8_t b = 0;
            if (base_read(z->stream.base, &b, 1) == 1)
                z->in[z->in_len++] = b;
            else
                z->in[z->in_len++] = 0; /* the reference ignores short reads here too */
        }
        if (z->method == MZH_COMPRESS_METHOD_LZMA)
            z->total_in += LZMA_MAGIC_SIZE;
    }
    z->initialized = 1;
    z->mode = mode;
    return MZH_OK;
}

int32_t mz_stream_lzma_is_open(void *stream) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    return z->initialized == 1 ? MZH_OK : MZH_OPEN_ERROR;
}

static int32_t pull_chunk(mzhip_lzma *z) {
    int32_t want = MZH_STAGING_BYTES;
    /* the first pull of a stream that may be served from a primed archive asks for no more than the bytes the lookup
     * compares (the size of a pull is not observable: the zip layer positions the base stream itself, mz_zip.c:1713);
     * a primed 64 KiB entry costs a 256-byte copy instead of 32 KiB, and the pages behind it are never touched */
    if ((z->in_len == 0 || (z->method == MZH_COMPRESS_METHOD_LZMA && z->in_len == LZMA_MAGIC_SIZE + 5)) && !z->tried_cache && mzhip_prime_any())
        want = 256;
    /* method 14: the reference first asks for exactly what is missing of the 5 header bytes behind the magic
     * (mz_strm_lzma.c:181-183) and hands them to liblzma before it reads on: a properties byte that is refused leaves the
     * base stream at 9 bytes */
    if (z->method == MZH_COMPRESS_METHOD_LZMA && z->in_dropped == 0 && z->in_len < LZMA_MAGIC_SIZE + 5)
        want = (int32_t)(LZMA_MAGIC_SIZE + 5 - z->in_len);
    if (z->max_total_in > 0) {
        int64_t left = z->max_total_in - (z->in_dropped + z->in_len);
        if (left < want)
            want = (int32_t)(left < 0 ? 0 : left);
    }
    if (want == 0) {
        z->base_eof = 1;
        return 0;
    }
    if (grow_in(z, z->in_len + want) != MZH_OK)
        return MZH_MEM_ERROR;
    int32_t rd = base_read(z->stream.base, z->in + z->in_len, want);
    if (rd < 0)
        return rd;
    if (rd == 0)
        z->base_eof = 1;
    z->in_len += rd;
    return rd;
}

#define LZ_START_STREAMING 2 /* attempt_decode(): the stream was (re)started in window mode */
static int32_t lz_stream_start(mzhip_lzma *z);
static int32_t xz_stream_start(mzhip_lzma *z);
static int32_t attempt_decode(mzhip_lzma *z) {
    for (;;) {
        if (z->out_cap == 0) {
            z->out_cap = z->max_total_out >= 0 ? z->max_total_out + 16 : z->in_len * 6 + 65536;
            if (z->method == MZH_COMPRESS_METHOD_LZMA && z->out_cap > mzh_stream_window() && z->in_len >= LZMA_MAGIC_SIZE + 5) {
                z->out_cap = 0;
                return lz_stream_start(z); /* larger than a window: decoded window by window from the start */
            }
            if (z->method == MZH_COMPRESS_METHOD_XZ && z->out_cap > mzh_stream_window() && !z->xz_no_stream) {
                const int64_t cap = z->out_cap;
                z->out_cap = 0;
                const int32_t r = xz_stream_start(z);
                if (r != 0)
                    return r; /* in windows from the start (LZ_START_STREAMING), more input first (1), or out of memory */
                z->out_cap = cap; /* (filters, a SHA-256 check, a malformed header: the one-buffer path and its verdicts) */
            }
            z->out = (uint8_t *)malloc((size_t)z->out_cap);
            if (!z->out)
                return MZH_MEM_ERROR;
        }
        uint32_t out_len = 0, in_used = 0, crc = 0;
        int32_t st = (z->method == MZH_COMPRESS_METHOD_XZ ? mzhip_xz_host : mzhip_lzma_host)(
            z->in, (uint32_t)z->in_len, z->out, (uint32_t)z->out_cap, z->max_total_out, &out_len, &in_used, &crc);
        if (st == MZHIP_STATUS_OUT_FULL) {
            if (z->method == MZH_COMPRESS_METHOD_LZMA && z->out_cap >= mzh_stream_window())
                return lz_stream_start(z); /* more than a window of output: once more, this time window by window */
            if (z->method == MZH_COMPRESS_METHOD_XZ && z->out_cap >= mzh_stream_window() && !z->xz_no_stream) {
                const int32_t r = xz_stream_start(z);
                if (r == LZ_START_STREAMING || r < 0)
                    return r;
                /* (r == 1 cannot be: the one-shot decode has just read past the first block header) */
            }
            if (z->out_cap >= 0x7FFFFFFF)
                return MZH_MEM_ERROR;
            int64_t ncap = z->out_cap * 4;
            if (z->method == MZH_COMPRESS_METHOD_LZMA && ncap > mzh_stream_window())
                ncap = mzh_stream_window();
            if (ncap > 0x7FFFFFFF)
                ncap = 0x7FFFFFFF;
            free(z->out);
            z->out = (uint8_t *)malloc((size_t)ncap);
            if (!z->out)
                return MZH_MEM_ERROR;
            z->out_cap = ncap;
            continue;
        }
        if (st == MZHIP_STATUS_BUF_ERROR && !z->base_eof)
            return 1;
        z->dev_status = st;
        z->out_len = out_len;
        z->dev_in_used = in_used;
        z->decoded = 1;
        return 0;
    }
}

/* ---- window mode (method 14): entries of any size in bounded memory ---------------------------------------------------
 * The reference streams an entry through 32 767 bytes in and whatever the caller's buffer holds out (mz_strm_lzma.c:
 * 147-241); the one-buffer path above holds the whole entry twice.  Past one window (mzh_stream_window, 64 MiB) the stream
 * is decoded by the resumable build of K3 (mzhip_lzma_resume_host): the coder state is a 64-byte record, the adaptive
 * model 28 KB of host memory, out[] = [the dictionary so far, at most what the header asks for | the window].  Memory:
 * dictionary + window + one gulp of input.  One stream is one wave (an LZMA stream is one serial chain): this is about
 * being able to read such an entry at all, not about speed. */
static int32_t lz_stream_start(mzhip_lzma *z) {
    uint64_t dict = (uint64_t)z->in[5] | ((uint64_t)z->in[6] << 8) | ((uint64_t)z->in[7] << 16) | ((uint64_t)z->in[8] << 24);
    if (dict < 4096)
        dict = 4096;
    /* the header's dictionary size is untrusted; no match can reach back further than the entry is long, so an entry of
     * known size (MZ_STREAM_PROP_TOTAL_OUT_MAX, what mz_zip.c hands every LZMA entry) never keeps more than that: a small
     * entry that declares 1 GiB costs its own size, not the gigabyte */
    if (z->max_total_out >= 0 && (uint64_t)z->max_total_out < dict)
        dict = (uint64_t)z->max_total_out < 4096 ? 4096 : (uint64_t)z->max_total_out;
    dict = (dict + 15) & ~(uint64_t)15;
    if (dict > ((uint64_t)1 << 30))
        return MZH_MEM_ERROR; /* (a dictionary beyond 1 GiB: liblzma would allocate it; this backend does not) */
    const int64_t cap = (int64_t)dict + mzh_stream_window() + 16;
    if (cap > 0x7FFFFFFF)
        return MZH_MEM_ERROR;
    free(z->out);
    z->out = (uint8_t *)malloc((size_t)cap);
    if (!z->model)
        z->model = malloc(mzhip_lzma_model_bytes());
    if (!z->out || !z->model)
        return MZH_MEM_ERROR;
    z->out_cap = cap;
    z->dict_keep = (int64_t)dict;
    z->out_len = z->out_served = z->hist = z->out_abs = z->in_dropped = 0;
    z->resumed = z->stream_end = 0;
    z->s_err = 0;
    z->ghost = 0;
    memset(&z->lst, 0, sizeof(z->lst));
    z->streaming = 1;
    z->decoded = 1; /* (the one-buffer loop is done with) */
    return LZ_START_STREAMING;
}

/* the next window: out[hist .. out_len) are new bytes, or the stream has ended / cannot go on (stream_end / s_err) */
static int32_t lz_stream_next(mzhip_lzma *z) {
    if (z->out_len > 0) { /* the dictionary moves to the front; a multiple of 16 goes (position contexts) */
        int64_t keep = z->out_len < z->dict_keep ? z->out_len : z->dict_keep;
        const int64_t drop = (z->out_len - keep) & ~(int64_t)15;
        keep = z->out_len - drop;
        if (drop > 0)
            memmove(z->out, z->out + drop, (size_t)keep);
        z->hist = keep;
        z->out_abs += drop;
        z->out_len = z->out_served = keep;
    }
    int64_t want_in = mzh_stream_gulp();
    for (;;) {
        while (!z->base_eof && z->in_len < want_in) {
            const int32_t rd = pull_chunk(z);
            if (rd < 0) {
                if (rd == MZH_MEM_ERROR)
                    return rd;
                z->base_err = rd; /* a failing base read ends the input; it is the result only if the stream needs more */
                z->base_eof = 1;
            }
        }
        mzhip_lzma_state sin = z->lst, sout;
        sin.flags = (z->resumed ? 1u : 0u) | (z->base_eof ? 2u : 0u);
        sin.out_pos = (uint32_t)z->hist;
        memset(&sout, 0, sizeof(sout));
        uint32_t ol = 0, iu = 0;
        int64_t room = z->hist + mzh_stream_window(); /* a window's worth behind the dictionary, however small that still is */
        if (room > z->out_cap)
            room = z->out_cap;
        const int32_t st = mzhip_lzma_resume_host(z->in, (uint32_t)z->in_len, z->out, (uint32_t)room, &sin, &sout, z->model, &ol, &iu);
        if (st != 0 && st != MZHIP_STATUS_OUT_FULL && st != MZHIP_STATUS_BUF_ERROR && st != MZHIP_STATUS_DATA_ERROR) {
            z->s_err = MZHIP_STATUS_DATA_ERROR; /* device failure: never substitute a CPU result */
            z->error = MZH_STREAM_ERROR;
            return MZH_OK;
        }
        if (iu > (uint32_t)z->in_len)
            iu = (uint32_t)z->in_len;
        if (ol < (uint32_t)z->hist || ol > (uint32_t)z->out_cap)
            ol = (uint32_t)z->hist;
        if (st == 0 || (sout.flags & 1u)) { /* the consumed bytes are done with (a failed call keeps them: TOTAL_IN wants them) */
            memmove(z->in, z->in + iu, (size_t)(z->in_len - iu));
            z->in_len -= iu;
            z->in_dropped += iu;
        } else {
            z->dev_in_used = iu;
        }
        z->out_len = ol;
        z->out_served = z->hist;
        if (st == 0) {
            z->stream_end = 1;
            return MZH_OK;
        }
        if (sout.flags & 1u) {
            z->lst = sout;
            z->resumed = 1;
            if (z->out_len > z->hist)
                return MZH_OK; /* bytes to serve; the next window goes on from the state */
            if (st == MZHIP_STATUS_BUF_ERROR && !z->base_eof) {
                want_in = z->in_len + mzh_stream_gulp(); /* not one packet's worth of input: more, then again */
                continue;
            }
            if (st == MZHIP_STATUS_BUF_ERROR) { /* the input ended inside a packet: truncation, not corruption (LZMA_BUF_ERROR) */
                z->s_err = MZHIP_STATUS_BUF_ERROR;
                return MZH_OK;
            }
            /* no room for one packet in a window: cannot happen (a window is >= 128 KiB) */
            z->s_err = MZHIP_STATUS_DATA_ERROR;
            return MZH_OK;
        }
        z->s_err = st; /* the stream ends short (BUF_ERROR) or is malformed: served once the bytes in front of it are */
        return MZH_OK;
    }
}

/* ---- window mode (method 95): .xz entries of any size in bounded memory -----------------------------------------------
 * lzma_stream_decoder (mz_strm_lzma.c:127-128) streams an .xz entry through the same 32 767-byte staging buffer.  Here the
 * container (.xz 1.0.4: stream header 2.1.1, block header 3.1, block padding and check 3.3 / 3.4, index 4, stream footer
 * 2.1.2) is walked by this file -- a few dozen bytes per block -- and each block's LZMA2 chunk sequence is decoded by the
 * device window by window (mzhip_lzma2_run_host: coder state a 80-byte record, the model 28 KB of host memory, out[] =
 * [dictionary so far | window]), the block's CRC-32 / CRC-64 carried through the windows on the device.  Streams this does
 * not take -- Delta / BCJ filters in front of LZMA2, a SHA-256 check, anything malformed in the first header -- stay with
 * the one-buffer path (the reference reads those in bounded memory too: a limitation). */
enum { XZP_BLOCK = 0, XZP_CHUNKS, XZP_BLOCK_END, XZP_INDEX, XZP_FOOTER };
static uint32_t xz_le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t xz_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xFF51AFD7ED558CCDull;
}
/* variable-length integer (1.2): 1 ok, 0 runs off `lim`, -1 malformed */
static int xz_vli(const uint8_t *b, int64_t *p, int64_t lim, uint64_t *v) {
    uint64_t r = 0;
    for (int i = 0;; i++) {
        if (i == 9)
            return -1;
        if (*p >= lim)
            return 0;
        const uint8_t c = b[(*p)++];
        if (c == 0 && i != 0)
            return -1;
        r |= (uint64_t)(c & 0x7F) << (7 * i);
        if (!(c & 0x80))
            break;
    }
    *v = r;
    return 1;
}
/* a block header of hsize bytes at b (3.1): 0 = LZMA2 alone and well-formed, 1 = anything else */
static int xz_block_header(const uint8_t *b, int64_t hsize, uint64_t *want_c, uint64_t *want_u, uint32_t *dict) {
    if (mzhip_crc32_host(0, b, (size_t)(hsize - 4)) != xz_le32(b + hsize - 4))
        return 1;
    const uint8_t fl = b[1];
    if (fl & 0x3F) /* reserved bits; more filters than LZMA2 */
        return 1;
    int64_t p = 2;
    const int64_t hend = hsize - 4;
    *want_c = *want_u = ~0ull;
    if ((fl & 0x40) && (xz_vli(b, &p, hend, want_c) != 1 || *want_c == 0))
        return 1;
    if ((fl & 0x80) && xz_vli(b, &p, hend, want_u) != 1)
        return 1;
    uint64_t id = 0, psize = 0;
    if (xz_vli(b, &p, hend, &id) != 1 || xz_vli(b, &p, hend, &psize) != 1 || id != 0x21 || psize != 1 || p >= hend)
        return 1;
    const uint32_t db = b[p++];
    if (db > 40)
        return 1;
    *dict = db == 40 ? 0xFFFFFFFFu : (uint32_t)(2u | (db & 1u)) << (db / 2u + 11u);
    for (; p < hend; p++)
        if (b[p] != 0)
            return 1;
    return 0;
}
static void xz_consume(mzhip_lzma *z, int64_t n) {
    memmove(z->in, z->in + n, (size_t)(z->in_len - n));
    z->in_len -= n;
    z->in_dropped += n;
}
static void xz_fresh_block(mzhip_lzma *z, uint32_t dict) {
    memset(&z->xst, 0, sizeof(z->xst));
    z->xst.flags = 4u | 8u;
    z->xst.dict = dict;
    z->xst.check_id = z->xz_check;
    z->xz_blk_csize = z->xz_blk_usize = 0;
}

/* 0: not in windows (the one-buffer path has this stream), 1: more input first, LZ_START_STREAMING, < 0: out of memory */
static int32_t xz_stream_start(mzhip_lzma *z) {
    static const uint8_t magic[6] = {0xFD, '7', 'z', 'X', 'Z', 0x00};
    z->xz_no_stream = 1;
    if (z->in_len < 13 && !z->base_eof) {
        /* a short first read (the 256-byte probe pull, a pipe, a stream that hands back less than it was asked for) says
         * nothing about the header yet: more input first -- only a header that IS malformed or unsupported locks the stream
         * out of window mode (a large .xz entry then ran the one-buffer path into MEM_ERROR past 2 GiB) */
        z->xz_no_stream = 0;
        return 1;
    }
    if (z->in_len < 13 || memcmp(z->in, magic, 6) != 0 || z->in[6] != 0 || (z->in[7] != 0 && z->in[7] != 1 && z->in[7] != 4) ||
        mzhip_crc32_host(0, z->in + 6, 2) != xz_le32(z->in + 8) || z->in[12] == 0)
        return 0;
    const int64_t hsize = ((int64_t)z->in[12] + 1) * 4;
    if (z->in_len < 12 + hsize) {
        if (z->base_eof)
            return 0;
        z->xz_no_stream = 0;
        return 1;
    }
    uint64_t wc, wu;
    uint32_t dict32 = 0;
    if (xz_block_header(z->in + 12, hsize, &wc, &wu, &dict32) != 0)
        return 0;
    uint64_t dict = dict32 < 4096 ? 4096 : dict32;
    /* (no match reaches back further than the entry is long: see lz_stream_start) */
    if (z->max_total_out >= 0 && (uint64_t)z->max_total_out < dict)
        dict = (uint64_t)z->max_total_out < 4096 ? 4096 : (uint64_t)z->max_total_out;
    dict = (dict + 15) & ~(uint64_t)15;
    if (dict > ((uint64_t)1 << 30))
        return MZH_MEM_ERROR; /* (a dictionary beyond 1 GiB: liblzma would allocate it; this backend does not) */
    const int64_t cap = (int64_t)dict + mzh_stream_window() + 16;
    if (cap > 0x7FFFFFFF)
        return MZH_MEM_ERROR;
    free(z->out);
    z->out = (uint8_t *)malloc((size_t)cap);
    if (!z->model)
        z->model = malloc(mzhip_lzma_model_bytes());
    if (!z->out || !z->model)
        return MZH_MEM_ERROR;
    z->out_cap = cap;
    z->dict_keep = (int64_t)dict;
    z->out_len = z->out_served = z->hist = z->out_abs = z->in_dropped = 0;
    z->stream_end = 0;
    z->s_err = 0;
    z->xz_check = z->in[7];
    z->xz_nblocks = z->xz_digest = z->xz_index_size = 0;
    z->xz_blk_hsize = (uint64_t)hsize;
    z->xz_want_csize = wc;
    z->xz_want_usize = wu;
    xz_fresh_block(z, dict32);
    xz_consume(z, 12 + hsize);
    z->xz_phase = XZP_CHUNKS;
    z->xz_no_stream = 0;
    z->streaming = 2;
    z->decoded = 1; /* (the one-buffer loop is done with) */
    return LZ_START_STREAMING;
}

/* the next window of a method-95 entry: new bytes in out[hist .. out_len), or the stream has ended / cannot go on */
static int32_t xz_stream_next(mzhip_lzma *z) {
    if (z->out_len > 0) { /* the dictionary moves to the front; a multiple of 16 goes (position contexts) */
        int64_t keep = z->out_len < z->dict_keep ? z->out_len : z->dict_keep;
        const int64_t drop = (z->out_len - keep) & ~(int64_t)15;
        keep = z->out_len - drop;
        if (drop > 0)
            memmove(z->out, z->out + drop, (size_t)keep);
        z->hist = keep;
        z->out_abs += drop;
        z->out_len = z->out_served = keep;
        z->xst.dict_start = (int64_t)z->xst.dict_start > drop ? (uint32_t)(z->xst.dict_start - drop) : 0u;
    }
    const uint32_t check_size = z->xz_check == 0 ? 0u : z->xz_check == 1 ? 4u : 8u;
    int64_t want_in = mzh_stream_gulp();
    int eager = 0;         /* a block has just ended with bytes still to serve: the walk liblzma makes in the same call */
    int64_t eager_lim = 0; /* ... reaches this far into the stream */
    z->xz_tin_hold = -1;
    z->xz_err_eager = 0;
    for (;;) {
        while (!z->base_eof && z->in_len < want_in) {
            const int32_t rd = pull_chunk(z);
            if (rd < 0) {
                if (rd == MZH_MEM_ERROR)
                    return rd;
                z->base_err = rd; /* a failing base read ends the input; it is the result only if the stream needs more */
                z->base_eof = 1;
            }
        }
/* bytes [0, n) of in[] must be there: more input, or the stream ends short */
#define XZ_WANT(n)                                                                      \
    if (eager && z->in_dropped + (int64_t)(n) > eager_lim) {                            \
        z->xz_tin_hold = eager_lim; /* the staging buffer ends here: the next call's */ \
        return MZH_OK;                                                                  \
    }                                                                                   \
    if (z->in_len < (int64_t)(n)) {                                                     \
        if (!z->base_eof) {                                                             \
            want_in = z->in_len + mzh_stream_gulp();                                    \
            continue;                                                                   \
        }                                                                               \
        if (eager) { /* (liblzma asks for more first: LZMA_BUF_ERROR is the next call's) */ \
            z->xz_tin_hold = z->in_dropped + z->in_len;                                 \
            return MZH_OK;                                                              \
        }                                                                               \
        z->s_err = MZHIP_STATUS_BUF_ERROR;                                              \
        return MZH_OK;                                                                  \
    }
#define XZ_BAD()                                          \
    do {                                                  \
        z->s_err = MZHIP_STATUS_DATA_ERROR;               \
        z->dev_in_used = 0;                               \
        z->xz_err_eager = (int8_t)eager;                  \
        return MZH_OK;                                    \
    } while (0)
        if (z->xz_phase == XZP_BLOCK) {
            XZ_WANT(1);
            if (z->in[0] == 0) {
                z->xz_phase = XZP_INDEX;
                continue;
            }
            const int64_t hsize = ((int64_t)z->in[0] + 1) * 4;
            XZ_WANT(hsize);
            uint32_t dict32 = 0;
            if (xz_block_header(z->in, hsize, &z->xz_want_csize, &z->xz_want_usize, &dict32) != 0)
                XZ_BAD(); /* (a block with filters behind one without: not read in windows) */
            {
                /* a block with a larger dictionary than the buffer was sized for: the buffer grows (bytes still to be
                 * served stay where they are; the block's first chunk starts a new dictionary) */
                uint64_t dict = dict32 < 4096 ? 4096 : dict32;
                if (z->max_total_out >= 0 && (uint64_t)z->max_total_out < dict)
                    dict = (uint64_t)z->max_total_out < 4096 ? 4096 : (uint64_t)z->max_total_out;
                dict = (dict + 15) & ~(uint64_t)15;
                if ((int64_t)dict > z->dict_keep) {
                    const int64_t cap = (int64_t)dict + mzh_stream_window() + 16;
                    uint8_t *nb = (dict > ((uint64_t)1 << 30) || cap > 0x7FFFFFFF) ? NULL : (uint8_t *)realloc(z->out, (size_t)cap);
                    if (!nb)
                        return MZH_MEM_ERROR;
                    z->out = nb;
                    z->out_cap = cap;
                    z->dict_keep = (int64_t)dict;
                }
            }
            z->xz_blk_hsize = (uint64_t)hsize;
            xz_fresh_block(z, dict32);
            z->xst.dict_start = (uint32_t)z->out_len; /* (the first chunk resets the dictionary anyway) */
            xz_consume(z, hsize);
            z->xz_phase = XZP_CHUNKS;
            continue;
        }
        if (z->xz_phase == XZP_CHUNKS) {
            if (eager)
                return MZH_OK; /* the bytes in hand first */
            mzhip_lzma2_state sin = z->xst, sout;
            sin.flags = (sin.flags & ~(2u | 16u)) | (z->base_eof ? 2u : 0u);
            sin.out_pos = (uint32_t)z->out_len;
            memset(&sout, 0, sizeof(sout));
            uint32_t ol = 0, iu = 0;
            int64_t room = z->out_len + mzh_stream_window(); /* a window's worth behind what is there */
            if (room > z->out_cap)
                room = z->out_cap;
            mzhip_lzma2_run_args a;
            memset(&a, 0, sizeof(a));
            a.size = (uint32_t)sizeof(a);
            a.in = z->in;
            a.in_len = (uint32_t)(z->in_len > 0x7FFFFFF0 ? 0x7FFFFFF0 : z->in_len);
            a.buf = z->out;
            a.buf_cap = (uint32_t)room;
            a.state_in = &sin;
            a.state_out = &sout;
            a.model = z->model;
            a.out_len = &ol;
            a.in_used = &iu;
            const int32_t st = mzhip_lzma2_run_host(&a);
            if (st != 0 && st != MZHIP_STATUS_OUT_FULL && st != MZHIP_STATUS_BUF_ERROR && st != MZHIP_STATUS_DATA_ERROR) {
                z->s_err = MZHIP_STATUS_DATA_ERROR; /* device failure: never substitute a CPU result */
                z->error = MZH_STREAM_ERROR;
                return MZH_OK;
            }
            if (iu > a.in_len)
                iu = a.in_len;
            if (ol < (uint32_t)z->out_len || ol > (uint32_t)room)
                ol = (uint32_t)z->out_len;
            const int64_t fresh = (int64_t)ol - z->out_len;
            z->xz_blk_usize += (uint64_t)fresh;
            z->out_len = ol;
            if (st == 0 || (sout.flags & 1u)) { /* the consumed bytes are done with (a failed call keeps them: TOTAL_IN wants them) */
                xz_consume(z, iu);
                z->xz_blk_csize += iu;
                z->xst = sout;
                z->xst.flags = (sout.flags & ~(16u | 2u)) | ((sout.flags & 1u) ? 1u : 0u);
            } else {
                z->dev_in_used = iu;
            }
            if (st == 0) {
                z->xz_phase = XZP_BLOCK_END;
                if (z->out_len > z->out_served) {
                    eager = 1;
                    eager_lim = z->in_dropped > 0 ? ((z->in_dropped - 1) / MZH_STAGING_BYTES + 1) * MZH_STAGING_BYTES : 0;
                    if (z->max_total_in > 0 && eager_lim > z->max_total_in)
                        eager_lim = z->max_total_in;
                }
                continue;
            }
            if (sout.flags & 1u) {
                if (fresh > 0)
                    return MZH_OK; /* bytes to serve; the next window goes on from the state */
                if (st == MZHIP_STATUS_BUF_ERROR && !z->base_eof) {
                    want_in = z->in_len + mzh_stream_gulp(); /* not a chunk header's, or a packet's, worth of input: more, then again */
                    continue;
                }
                if (st == MZHIP_STATUS_BUF_ERROR) { /* the input ended inside the block: truncation, not corruption (LZMA_BUF_ERROR) */
                    z->s_err = MZHIP_STATUS_BUF_ERROR;
                    return MZH_OK;
                }
                z->s_err = MZHIP_STATUS_DATA_ERROR; /* no room for one packet in a window: cannot happen (a window is >= 128 KiB) */
                return MZH_OK;
            }
            z->s_err = st; /* the block ends short (BUF_ERROR) or is malformed: served once the bytes in front of it are */
            if (st == MZHIP_STATUS_DATA_ERROR && (sout.flags & 128u))
                z->xz_err_eager = 1; /* (between chunks: liblzma meets it in the call that made the bytes in front) */
            return MZH_OK;
        }
        if (z->xz_phase == XZP_BLOCK_END) {
            /* sizes, block padding, check (3.3, 3.4) */
            const int64_t pad = (int64_t)((0 - z->xz_blk_csize) & 3u);
            if ((z->xz_want_csize != ~0ull && z->xz_want_csize != z->xz_blk_csize) ||
                (z->xz_want_usize != ~0ull && z->xz_want_usize != z->xz_blk_usize))
                XZ_BAD();
            XZ_WANT(pad + check_size);
            for (int64_t i = 0; i < pad; i++)
                if (z->in[i] != 0)
                    XZ_BAD();
            if (z->xz_check == 1 && xz_le32(z->in + pad) != z->xst.check_lo)
                XZ_BAD();
            if (z->xz_check == 4 && (xz_le32(z->in + pad) != z->xst.check_lo || xz_le32(z->in + pad + 4) != z->xst.check_hi))
                XZ_BAD();
            xz_consume(z, pad + check_size);
            z->xz_nblocks++;
            z->xz_digest = xz_mix(xz_mix(z->xz_digest, z->xz_blk_hsize + z->xz_blk_csize + check_size), z->xz_blk_usize);
            z->xz_phase = XZP_BLOCK;
            continue;
        }
        if (z->xz_phase == XZP_INDEX) {
            /* index (4): indicator, record count, the records, padding, CRC-32 -- all of it in in[] before it is judged */
            int64_t p = 1;
            uint64_t count = 0, dig = 0;
            int r = xz_vli(z->in, &p, z->in_len, &count);
            if (r < 0 || (r == 1 && count != z->xz_nblocks))
                XZ_BAD();
            for (uint64_t i = 0; r == 1 && i < count; i++) {
                uint64_t unpadded = 0, usz = 0;
                r = xz_vli(z->in, &p, z->in_len, &unpadded);
                if (r == 1)
                    r = xz_vli(z->in, &p, z->in_len, &usz);
                if (r < 0)
                    XZ_BAD();
                dig = xz_mix(xz_mix(dig, unpadded), usz);
            }
            if (r == 0) { /* the index runs on behind what there is */
                XZ_WANT(z->in_len + 1);
            }
            if (dig != z->xz_digest)
                XZ_BAD();
            const int64_t isize = ((p + 3) & ~(int64_t)3) + 4;
            XZ_WANT(isize);
            for (int64_t i = p; i < isize - 4; i++)
                if (z->in[i] != 0)
                    XZ_BAD();
            if (mzhip_crc32_host(0, z->in, (size_t)(isize - 4)) != xz_le32(z->in + isize - 4))
                XZ_BAD();
            z->xz_index_size = (uint64_t)isize;
            xz_consume(z, isize);
            z->xz_phase = XZP_FOOTER;
            continue;
        }
        /* stream footer (2.1.2): CRC-32, backward size, stream flags, magic */
        XZ_WANT(12);
        if (z->in[10] != 'Y' || z->in[11] != 'Z' || mzhip_crc32_host(0, z->in + 4, 6) != xz_le32(z->in) || z->in[8] != 0 ||
            z->in[9] != z->xz_check || ((uint64_t)xz_le32(z->in + 4) + 1u) * 4u != z->xz_index_size)
            XZ_BAD();
        xz_consume(z, 12);
        z->stream_end = 1;
        return MZH_OK;
#undef XZ_WANT
#undef XZ_BAD
    }
}

/* liblzma failed inside this call: the reference returns the error, not the bytes of the call, and its totals count
 * everything the decoder took and produced */
static int32_t lz_stream_fail(mzhip_lzma *z) {
    z->error = z->s_err == MZHIP_STATUS_BUF_ERROR ? 10 : 9; /* LZMA_BUF_ERROR : LZMA_DATA_ERROR */
    z->total_out = z->out_abs + z->out_len;
    if (z->max_total_out >= 0 && z->total_out > z->max_total_out)
        z->total_out = z->max_total_out;
    z->total_in = z->s_err == MZHIP_STATUS_BUF_ERROR ? z->in_dropped + z->in_len : z->in_dropped + z->dev_in_used;
    if (z->base_err != 0 && z->s_err == MZHIP_STATUS_BUF_ERROR)
        return z->base_err;
    return MZH_DATA_ERROR;
}

static int32_t lz_stream_read(mzhip_lzma *z, void *buf, int32_t size) {
    int32_t got = 0;
    while (got < size) {
        int64_t avail = z->out_len - z->out_served;
        if (z->max_total_out >= 0 && z->total_out + avail > z->max_total_out)
            avail = z->max_total_out - z->total_out > 0 ? z->max_total_out - z->total_out : 0; /* mz_strm_lzma.c:214-215 */
        if (avail == 0) {
            if (z->stream_end)
                break;
            if (z->max_total_out >= 0 && z->total_out >= z->max_total_out) {
                /* everything the caller may have has been served.  The caller's buffer still has room, so the reference goes
                 * on calling lzma_code (mz_strm_lzma.c:237).  Method 14: liblzma decodes on into that room -- what it writes
                 * there is not counted (mz_strm_lzma.c:214-215) -- until the room is full (the call returns what it had), the
                 * end marker arrives (the same) or the stream fails: a stream cut inside its end marker is LZMA_BUF_ERROR
                 * in the very call that would have returned the entry's last bytes (round 5, tests/fuzz_lzma_windows.py) */
                if (z->streaming != 2) {
                    int64_t room = size - got;
                    int32_t rc14 = MZH_OK;
                    for (;;) {
                        const int64_t have = z->out_abs + z->out_len - z->max_total_out - z->ghost; /* decoded behind the limit, not yet "written" */
                        if (have >= room) {
                            z->ghost += room;
                            room = 0;
                            break;
                        }
                        z->ghost += have;
                        room -= have;
                        if (z->stream_end)
                            break;
                        if (z->s_err != 0)
                            return lz_stream_fail(z);
                        rc14 = lz_stream_next(z);
                        if (rc14 != MZH_OK || z->error != 0)
                            break;
                    }
                    if (rc14 != MZH_OK) {
                        z->error = 5; /* LZMA_MEM_ERROR */
                        return MZH_DATA_ERROR;
                    }
                    if (z->error != 0)
                        return MZH_DATA_ERROR;
                    break;
                }
                if (z->s_err != 0)
                    return lz_stream_fail(z);
                if (z->xz_tail_tried)
                    break;
                z->xz_tail_tried = 1;
            } else if (z->s_err != 0) {
                return lz_stream_fail(z);
            }
            const int32_t rc = z->streaming == 2 ? xz_stream_next(z) : lz_stream_next(z);
            if (rc != MZH_OK) {
                z->error = 5; /* LZMA_MEM_ERROR */
                return MZH_DATA_ERROR;
            }
            if (z->error != 0)
                return MZH_DATA_ERROR;
            continue;
        }
        const int32_t k = (int32_t)(avail < size - got ? avail : size - got);
        memcpy((uint8_t *)buf + got, z->out + z->out_served, (size_t)k);
        z->out_served += k;
        z->total_out += k;
        got += k;
    }
    if (z->streaming == 1 && got == size && z->out_served == z->out_len && !z->stream_end &&
        (z->max_total_out < 0 || z->total_out < z->max_total_out)) {
        /* method 14, the buffer full exactly where the decoded bytes end: liblzma has looked at the next packet in this call
         * (above, the one-buffer path) -- is it one it refuses? */
        if (z->s_err == 0) {
            const int32_t rc = lz_stream_next(z);
            if (rc != MZH_OK) {
                z->error = 5; /* LZMA_MEM_ERROR */
                return MZH_DATA_ERROR;
            }
            if (z->error != 0)
                return MZH_DATA_ERROR;
        }
        if (z->s_err == MZHIP_STATUS_DATA_ERROR && z->out_served == z->out_len)
            return lz_stream_fail(z);
    }
    if (z->streaming == 2 && z->s_err != 0 && z->xz_err_eager && z->out_served == z->out_len)
        return lz_stream_fail(z); /* met on the walk behind the block whose last bytes this call would have returned */
    z->total_in = z->in_dropped; /* exact once the end marker has been decoded (what mz_zip.c:2116 needs) */
    if (z->streaming == 2 && z->xz_tin_hold >= 0 && z->out_served == z->out_len)
        z->total_in = z->xz_tin_hold;
    return got;
}

int32_t mz_stream_lzma_read(void *stream, void *buf, int32_t size) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    mzhip_served_drop();
    if (z->error == 0 && mzhip_take_crc_fault() != 0)
        z->error = MZH_STREAM_ERROR; /* a checksum call before this one met a device failure */
    if (z->error != 0)
        return MZH_DATA_ERROR; /* mz_strm_lzma.c:236-237 */
    if (z->streaming)
        return lz_stream_read(z, buf, size);
    while (!z->decoded) {
        int32_t rd = pull_chunk(z);
        if (rd < 0) {
            /* a failing base read ends the input; it is the result only if the stream needs more (see shim_zlib.c) */
            if (z->in_len <= (z->method == MZH_COMPRESS_METHOD_LZMA ? LZMA_MAGIC_SIZE : 0) || z->base_err != 0)
                return rd;
            z->base_err = rd;
            z->base_eof = 1;
        }
        if (z->method == MZH_COMPRESS_METHOD_LZMA && z->in_len == LZMA_MAGIC_SIZE + 5 && !z->base_eof) {
            const uint32_t d = z->in[4];
            if (d >= 9 * 5 * 5 || (d % 9) + ((d / 9) % 5) > 4) { /* refused by lzma_alone_decoder: nothing more is read */
                z->dev_status = MZHIP_STATUS_DATA_ERROR;
                z->out_len = 0;
                z->dev_in_used = z->in_len;
                z->decoded = 1;
                break;
            }
            continue; /* (the header alone decides nothing else: pull the stream) */
        }
        if (!z->tried_cache) {
            /* was this entry decoded by mzhip_prime_*()?  (payload offset + first payload bytes must agree) */
            z->tried_cache = 1;
            mzhip_autoprime(z->stream.base, z->base_pos0); /* (shim_autoprime.c: on unless MZHIP_AUTOPRIME=0) */
            const uint8_t *data = NULL;
            int64_t usize = 0, csize = 0;
            uint32_t crc = 0;
            if (z->base_pos0 >= 0 &&
                mzhip_prime_lookup3(z->method, z->base_pos0, z->in, (int32_t)(z->in_len < 256 ? z->in_len : 256), z->max_total_in,
                                    &data, &usize, &csize, &crc, &z->seg_crc, &z->prime_pin, &z->hash_alg, &z->hash_digest) == 1 &&
                (z->max_total_out < 0 || z->max_total_out >= usize)) {
                z->out = (uint8_t *)(uintptr_t)data;
                z->out_borrowed = 1;
                z->out_len = usize;
                z->dev_in_used = csize;
                z->dev_status = 0;
                z->decoded = 1;
                break;
            }
        }
        if (!z->base_eof && z->in_len < z->next_attempt)
            continue;
        int32_t r = attempt_decode(z);
        if (r < 0) {
            z->error = 5; /* LZMA_MEM_ERROR */
            return MZH_DATA_ERROR;
        }
        if (r == LZ_START_STREAMING)
            return lz_stream_read(z, buf, size);
        if (r == 1)
            z->next_attempt = z->in_len * 2;
    }
    int64_t avail = z->out_len - z->out_served;
    /* (method 14: liblzma is not told the entry's size, so with the caller's buffer full it still decodes the next packet to
     * see whether it is the end marker (lzma_decoder.c, SEQ_IS_MATCH without no_eopm) -- and a packet it refuses is refused
     * THERE: the call that returns the last bytes in front of a data error fails even when they fill the buffer.  LZMA2
     * chunks know their size: method 95 stops at a full buffer.  Round 5, a fuzz with 7-byte read() calls) */
    if (z->dev_status != 0 && (avail < size || (avail == size && z->method == MZH_COMPRESS_METHOD_LZMA &&
                                                z->dev_status == MZHIP_STATUS_DATA_ERROR &&
                                                (z->max_total_out < 0 || z->total_out + avail <= z->max_total_out)))) {
        z->error = z->dev_status == MZHIP_STATUS_BUF_ERROR ? 10 : 9; /* LZMA_BUF_ERROR : LZMA_DATA_ERROR */
        z->total_in = z->dev_in_used;
        z->total_out = z->out_len;
        if (z->method == MZH_COMPRESS_METHOD_LZMA && z->in_len >= 5 && z->out_len == 0) {
            /* a properties byte lzma_alone_decoder refuses: LZMA_FORMAT_ERROR before a byte is consumed.  The reference's
             * accounting then shows its header trick (mz_strm_lzma.c:197-205): with the 5 header bytes in hand it has
             * already taken the 8 bytes of the size field it appends off TOTAL_IN -> 4 - 8 = -4; with fewer, 4 */
            const uint32_t d = z->in[4];
            if (d >= 9 * 5 * 5 || (d % 9) + ((d / 9) % 5) > 4) {
                z->error = 7; /* LZMA_FORMAT_ERROR */
                z->total_in = z->in_len >= 9 ? -4 : 4;
            }
        }
        if (z->base_err != 0 && z->dev_status == MZHIP_STATUS_BUF_ERROR)
            return z->base_err;
        return MZH_DATA_ERROR;
    }
    int32_t n = (int32_t)(avail < size ? avail : size);
    if (n > 0) {
        memcpy(buf, z->out + z->out_served, (size_t)n);
        if (z->out_borrowed && z->out_served % MZHIP_PRIME_SEGMENT == 0 &&
            (n == MZHIP_PRIME_SEGMENT || z->out_served + n == z->out_len)) {
            /* a whole primed segment: its device-computed CRC answers the mz_crypt_crc32_update that follows */
            mzhip_served_set(buf, n, z->seg_crc[z->out_served / MZHIP_PRIME_SEGMENT], z->out + z->out_served, z->slot);
            mzhip_served_set_entry(z->out, z->out_served, z->out_len, z->hash_alg, z->hash_digest);
        }
        z->out_served += n;
        z->total_out += n;
    }
    if (z->out_served == z->out_len) {
        z->total_in = z->dev_in_used;
    } else {
        int64_t est = z->out_len ? (z->dev_in_used * z->out_served) / z->out_len : 0;
        if (est >= z->dev_in_used)
            est = z->dev_in_used - 1;
        if (est < LZMA_MAGIC_SIZE && z->method == MZH_COMPRESS_METHOD_LZMA)
            est = LZMA_MAGIC_SIZE;
        z->total_in = est;
    }
    return n;
}

#define LZ_WRITE_BLOCK 65536 /* the tokenizer's block: segments are multiples of it */

static int32_t base_write(mzhip_stream *base, const void *buf, int32_t size);

/* bytes coded per launch once an entry is larger than that: 8 MiB by default, whole blocks (mzhip_set_write_segment /
 * MZHIP_WRITE_SEGMENT; its own knob -- round 4 derived it from the READ window, so a read-side setting changed the bytes a
 * method-95 stream was written as) */
static int64_t lz_wseg_bytes; /* 0 = not decided yet; read and written with atomics: streams of several threads ask */
MZHIP_API void mzhip_set_write_segment(int64_t segment_bytes) {
    __atomic_store_n(&lz_wseg_bytes, segment_bytes > 0 ? segment_bytes : 0, __ATOMIC_RELAXED);
}
static int64_t lz_write_segment(void) {
    int64_t sgm = __atomic_load_n(&lz_wseg_bytes, __ATOMIC_RELAXED);
    if (!sgm) {
        const char *e = getenv("MZHIP_WRITE_SEGMENT");
        sgm = e ? strtoll(e, NULL, 0) : 0;
        if (sgm <= 0)
            sgm = 8 << 20;
        __atomic_store_n(&lz_wseg_bytes, sgm, __ATOMIC_RELAXED);
    }
    sgm &= ~(int64_t)(LZ_WRITE_BLOCK - 1);
    if (sgm < 2 * LZ_WRITE_BLOCK)
        sgm = 2 * LZ_WRITE_BLOCK;
    if (sgm > (8 << 20))
        sgm = 8 << 20;
    return sgm;
}

/* Method 14 in bounded memory (mz_strm_lzma.c:244-332 stages any entry through 32 767 bytes): a segment of whole blocks is
 * coded with the range coder's state and the adaptive model carried over (the same tokens and the same bytes as the
 * one-shot coder would make: the LZ77 parse sees the same history either way -- the encoder's matches reach back
 * mzhip_lzma_encode_history_bytes() = 8 MiB), its output goes to base, and the last 8 MiB coded so far stay in front of
 * wbuf[] as match sources and contexts for the next one. */
static int32_t lz_write_segment_out(mzhip_lzma *z, int32_t last) {
    const int64_t fresh = z->wlen - z->w_hist;
    int64_t take = fresh;
    if (!last) {
        take = fresh & ~(int64_t)(LZ_WRITE_BLOCK - 1);
        if (take > lz_write_segment())
            take = lz_write_segment();
        if (take <= 0)
            return MZH_OK;
    }
    if (!z->model)
        z->model = malloc(mzhip_lzma_model_bytes());
    if (!z->model)
        return MZH_MEM_ERROR;
    const int64_t in_len = z->w_hist + take;
    const uint32_t cap = (uint32_t)(take + take / 8 + 4096);
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out)
        return MZH_MEM_ERROR;
    mzhip_lzma_enc_state sin = z->west, sout;
    sin.flags = z->w_segments > 0 ? 1u : 0u;
    memset(&sout, 0, sizeof(sout));
    uint32_t out_len = 0;
    const int32_t st = mzhip_lzma_encode_resume_host(z->wbuf ? z->wbuf : (const uint8_t *)"", (uint32_t)in_len,
                                                     (uint32_t)(z->w_hist / LZ_WRITE_BLOCK), (uint32_t)last, (int32_t)z->preset, &sin,
                                                     &sout, z->model, out, cap, &out_len);
    if (st != 0) {
        free(out);
        return MZH_DATA_ERROR; /* device failure: never substitute a CPU result */
    }
    uint32_t pos = 0;
    while (pos < out_len) {
        const int32_t n = (int32_t)(out_len - pos < MZH_STAGING_BYTES ? out_len - pos : MZH_STAGING_BYTES);
        if (base_write(z->stream.base, out + pos, n) != n) {
            free(out);
            return MZH_WRITE_ERROR;
        }
        pos += (uint32_t)n;
    }
    free(out);
    z->total_out += out_len;
    z->w_segments++;
    if (!last) {
        z->west = sout;
        /* the coded bytes go, but for the history the encoder looks back over (whole blocks); what was not coded moves up
         * behind it */
        int64_t keep = (int64_t)(mzhip_lzma_encode_history_bytes() & ~(uint32_t)(LZ_WRITE_BLOCK - 1));
        if (keep < LZ_WRITE_BLOCK)
            keep = LZ_WRITE_BLOCK;
        if (keep > in_len)
            keep = in_len;
        memmove(z->wbuf, z->wbuf + in_len - keep, (size_t)(keep + (z->wlen - in_len)));
        z->wlen = keep + (z->wlen - in_len);
        z->w_hist = keep;
    }
    return MZH_OK;
}

/* Method 95 in bounded memory: the .xz container holds any number of blocks and a block starts with a fresh dictionary --
 * so a segment (8 MiB: as far as the parse looks back anyway) is simply one block (behind the stream header when it is
 * the first), and close() adds the index over all blocks and the footer.  Nothing is carried from segment to segment but
 * the two sizes per block the index wants. */
static int32_t xz_write_block_out(mzhip_lzma *z, int64_t take) {
    if (z->w_segments >= z->xz_cap) {
        const int32_t ncap = z->xz_cap ? z->xz_cap * 2 : 64;
        uint64_t *a = (uint64_t *)realloc(z->xz_unpadded, (size_t)ncap * sizeof(uint64_t));
        if (a)
            z->xz_unpadded = a;
        uint64_t *b = (uint64_t *)realloc(z->xz_usize, (size_t)ncap * sizeof(uint64_t));
        if (b)
            z->xz_usize = b;
        if (!a || !b)
            return MZH_MEM_ERROR;
        z->xz_cap = ncap;
    }
    const uint32_t cap = (uint32_t)(take + take / 8 + 4096 + (take / 49152 + 1) * 8);
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out)
        return MZH_MEM_ERROR;
    uint32_t out_len = 0, crc = 0;
    uint64_t unp = 0;
    const int32_t st = mzhip_xz_encode_block_host(z->wbuf, (uint32_t)take, (int32_t)z->preset, z->w_segments == 0, out, cap, &out_len, &crc, &unp);
    if (st != 0) {
        free(out);
        return MZH_DATA_ERROR; /* device failure: never substitute a CPU result */
    }
    uint32_t pos = 0;
    while (pos < out_len) {
        const int32_t n = (int32_t)(out_len - pos < MZH_STAGING_BYTES ? out_len - pos : MZH_STAGING_BYTES);
        if (base_write(z->stream.base, out + pos, n) != n) {
            free(out);
            return MZH_WRITE_ERROR;
        }
        pos += (uint32_t)n;
    }
    free(out);
    z->total_out += out_len;
    z->xz_unpadded[z->w_segments] = unp;
    z->xz_usize[z->w_segments] = (uint64_t)take;
    z->w_segments++;
    memmove(z->wbuf, z->wbuf + take, (size_t)(z->wlen - take));
    z->wlen -= take;
    return MZH_OK;
}

static int32_t xz_write_finish(mzhip_lzma *z) {
    if (z->wlen > 0) {
        const int32_t err = xz_write_block_out(z, z->wlen);
        if (err != MZH_OK)
            return err;
    }
    uint8_t tail[64];
    uint8_t *out = tail;
    uint32_t cap = sizeof(tail), out_len = 0;
    if ((size_t)z->w_segments * 18 + 32 > sizeof(tail)) {
        cap = (uint32_t)z->w_segments * 18 + 32;
        out = (uint8_t *)malloc(cap);
        if (!out)
            return MZH_MEM_ERROR;
    }
    int32_t err = mzhip_xz_encode_finish_host(z->xz_unpadded, z->xz_usize, (uint32_t)z->w_segments, out, cap, &out_len) == 0 ? MZH_OK : MZH_DATA_ERROR;
    if (err == MZH_OK && base_write(z->stream.base, out, (int32_t)out_len) != (int32_t)out_len)
        err = MZH_WRITE_ERROR;
    if (out != tail)
        free(out);
    if (err == MZH_OK)
        z->total_out += out_len;
    return err;
}

static int32_t collect(mzhip_lzma *z, const void *buf, int64_t size) {
    if (z->wlen + size > z->wcap) {
        int64_t ncap = z->wcap ? z->wcap * 2 : (1 << 20);
        while (ncap < z->wlen + size)
            ncap *= 2;
        uint8_t *p = (uint8_t *)realloc(z->wbuf, (size_t)ncap);
        if (!p)
            return MZH_MEM_ERROR;
        z->wbuf = p;
        z->wcap = ncap;
    }
    memcpy(z->wbuf + z->wlen, buf, (size_t)size);
    z->wlen += size;
    /* a full segment (and a block more, so that an entry of exactly one segment stays the one-shot case) is coded and leaves */
    while (z->wlen - z->w_hist >= lz_write_segment() + LZ_WRITE_BLOCK) {
        const int32_t err = z->method == MZH_COMPRESS_METHOD_LZMA ? lz_write_segment_out(z, 0) : xz_write_block_out(z, lz_write_segment());
        if (err != MZH_OK) {
            z->error = 11; /* LZMA_PROG_ERROR */
            return err == MZH_WRITE_ERROR ? MZH_WRITE_ERROR : MZH_DATA_ERROR;
        }
    }
    return MZH_OK;
}

/* the entry stopped following its primed buffer (mzhip_prime_write): what it shared with it is collected after all */
static int32_t leave_primed(mzhip_lzma *z) {
    int32_t err = MZH_OK;
    if (z->wp_id >= 0) {
        const uint8_t *src = NULL, *out = NULL;
        uint32_t out_len = 0;
        (void)mzhip_wprime_result(z->method, z->wp_id, -1, &src, &out, &out_len);
        if (!src)
            return MZH_INTERNAL_ERROR; /* the cache was cleared under a stream that was following it */
        err = collect(z, src, z->wp_pos);
    }
    z->wp_id = -1;
    z->wp_off = 1;
    return err;
}

int32_t mz_stream_lzma_write(void *stream, const void *buf, int32_t size) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    mzhip_served_drop();
    if (mzhip_take_crc_fault() != 0) { /* a checksum call before this one met a device failure */
        z->error = MZH_STREAM_ERROR;
        return MZH_STREAM_ERROR;
    }
    if (size <= 0)
        return size;
    if (!z->wp_off) {
        uint32_t crc = 0;
        int32_t have_crc = 0;
        const uint8_t *wsrc = NULL;
        if ((z->wp_id >= 0 || z->total_in == 0) &&
            mzhip_wprime_track(z->method, &z->wp_id, z->wp_pos, (const uint8_t *)buf, size, &crc, &have_crc, &wsrc) == 1) {
            z->wp_pos += size;
            z->total_in += size;
            if (have_crc) { /* answers the mz_crypt_crc32_update that follows (mz_zip.c:2062-2064) */
                mzhip_served_set(buf, size, crc, wsrc, z->slot);
            }
            return size;
        }
        int32_t err = leave_primed(z);
        if (err != MZH_OK)
            return err;
    }
    int32_t err = collect(z, buf, size);
    if (err != MZH_OK)
        return err;
    z->total_in += size; /* mz_strm_lzma.c:322 */
    return size;
}

static int32_t base_write(mzhip_stream *base, const void *buf, int32_t size) {
    if (size == 0)
        return size;
    if (!base || !base->vtbl || !base->vtbl->write)
        return MZH_PARAM_ERROR;
    if (!base->vtbl->is_open || base->vtbl->is_open(base) != MZH_OK)
        return MZH_STREAM_ERROR;
    return base->vtbl->write(base, buf, size);
}

/* code everything collected and hand it to base in staging-sized writes (mz_strm_lzma.c:244-248) */
static int32_t finish_write(mzhip_lzma *z) {
    if (z->wp_id >= 0) {
        const uint8_t *src = NULL, *pout = NULL;
        uint32_t pout_len = 0;
        if (mzhip_wprime_result(z->method, z->wp_id, z->wp_pos, &src, &pout, &pout_len) == 1) {
            /* the entry is exactly a primed buffer: its stream was coded in the batch */
            z->wp_id = -1;
            uint32_t pos = 0;
            while (pos < pout_len) {
                int32_t n = (int32_t)(pout_len - pos < MZH_STAGING_BYTES ? pout_len - pos : MZH_STAGING_BYTES);
                if (base_write(z->stream.base, pout + pos, n) != n)
                    return MZH_WRITE_ERROR;
                pos += (uint32_t)n;
            }
            z->total_out += pout_len;
            return MZH_OK;
        }
        if (leave_primed(z) != MZH_OK) /* a proper prefix of a primed buffer */
            return MZH_MEM_ERROR;
    }
    if (z->w_segments > 0) /* the last segment of an entry that left in segments: end marker and flush / last block, index, footer */
        return z->method == MZH_COMPRESS_METHOD_LZMA ? lz_write_segment_out(z, 1) : xz_write_finish(z);
    uint32_t cap = (uint32_t)(z->wlen + z->wlen / 8 + 4096 + (z->wlen / 49152 + 1) * 8);
    uint8_t *out = (uint8_t *)malloc(cap);
    if (!out)
        return MZH_MEM_ERROR;
    uint32_t out_len = 0, crc = 0;
    int32_t st = (z->method == MZH_COMPRESS_METHOD_XZ ? mzhip_xz_encode_host_preset : mzhip_lzma_encode_host_preset)(
        z->wbuf ? z->wbuf : (const uint8_t *)"", (uint32_t)z->wlen, (int32_t)z->preset, out, cap, &out_len, &crc);
    if (st != 0) {
        free(out);
        return MZH_DATA_ERROR; /* device failure: never substitute a CPU result */
    }
    uint32_t pos = 0;
    while (pos < out_len) {
        int32_t n = (int32_t)(out_len - pos < MZH_STAGING_BYTES ? out_len - pos : MZH_STAGING_BYTES);
        if (base_write(z->stream.base, out + pos, n) != n) {
            free(out);
            return MZH_WRITE_ERROR;
        }
        pos += (uint32_t)n;
    }
    free(out);
    z->total_out += out_len;
    return MZH_OK;
}

int64_t mz_stream_lzma_tell(void *stream) {
    (void)stream;
    return MZH_TELL_ERROR;
}

int32_t mz_stream_lzma_seek(void *stream, int64_t offset, int32_t origin) {
    (void)stream;
    (void)offset;
    (void)origin;
    return MZH_SEEK_ERROR;
}

int32_t mz_stream_lzma_close(void *stream) {
    mzhip_served_drop(); /* (the hint points into a primed generation this stream pins) */
    mzhip_buffers_released(((mzhip_lzma *)stream)->slot);
    mzhip_lzma *z = (mzhip_lzma *)stream;
    if ((z->mode & MZH_OPEN_MODE_WRITE) && z->initialized == 1 && finish_write(z) != MZH_OK)
        z->error = 11; /* LZMA_PROG_ERROR: reported as MZ_CLOSE_ERROR below */
    free(z->wbuf);
    z->wbuf = NULL;
    z->wlen = z->wcap = 0;
    z->initialized = 0;
    free(z->in);
    if (!z->out_borrowed)
        free(z->out);
    free(z->model);
    z->model = NULL;
    z->streaming = 0;
    mzhip_prime_unpin(z->prime_pin);
    z->prime_pin = NULL;
    z->out_borrowed = 0;
    z->in = z->out = NULL;
    z->in_cap = z->out_cap = 0;
    if (z->error != 0)
        return MZH_CLOSE_ERROR;
    return MZH_OK;
}

int32_t mz_stream_lzma_error(void *stream) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    return z->error;
}

int32_t mz_stream_lzma_get_prop_int64(void *stream, int32_t prop, int64_t *value) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
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
    case MZH_PROP_TOTAL_OUT_MAX:
        *value = z->max_total_out;
        break;
    case MZH_PROP_HEADER_SIZE:
        *value = LZMA_MAGIC_SIZE;
        break;
    default:
        return MZH_EXIST_ERROR;
    }
    return MZH_OK;
}

int32_t mz_stream_lzma_set_prop_int64(void *stream, int32_t prop, int64_t value) {
    mzhip_lzma *z = (mzhip_lzma *)stream;
    switch (prop) {
    case MZH_PROP_COMPRESS_LEVEL:
        z->preset = value == -1 ? 6u : (uint32_t)value; /* LZMA_PRESET_DEFAULT == 6 */
        break;
    case MZH_PROP_COMPRESS_METHOD:
        z->method = (int16_t)value;
        break;
    case MZH_PROP_TOTAL_IN_MAX:
        z->max_total_in = value;
        break;
    case MZH_PROP_TOTAL_OUT_MAX:
        if (value < -1)
            return MZH_PARAM_ERROR;
        z->max_total_out = value;
        break;
    default:
        return MZH_EXIST_ERROR;
    }
    return MZH_OK;
}

void *mz_stream_lzma_create(void) {
    mzhip_lzma *z = (mzhip_lzma *)calloc(1, sizeof(mzhip_lzma));
    if (z) {
        z->slot = mzhip_stream_slot_new();
        z->stream.vtbl = &mzhip_lzma_vtbl;
        z->method = MZH_COMPRESS_METHOD_LZMA;
        z->preset = 6;
        z->max_total_out = -1;
    }
    return z;
}

void mz_stream_lzma_delete(void **stream) {
    mzhip_served_drop();
    if (stream && *stream)
        mzhip_buffers_released(((mzhip_lzma *)*stream)->slot);
    mzhip_lzma *z;
    if (!stream)
        return;
    z = (mzhip_lzma *)*stream;
    if (z) {
        free(z->in);
        if (!z->out_borrowed)
            free(z->out);
        free(z->model);
        free(z->xz_unpadded);
        free(z->xz_usize);
        mzhip_prime_unpin(z->prime_pin);
        z->prime_pin = NULL;
        free(z->wbuf);
        free(z);
    }
    *stream = NULL;
}

void *mz_stream_lzma_get_interface(void) {
    return (void *)&mzhip_lzma_vtbl;
}
