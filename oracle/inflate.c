/* inflate.c -- oracle restatement of raw-DEFLATE decoding (TEST INFRASTRUCTURE).
 *
 * What the reference does for a method-8 entry is call zlib's inflate() on a
 * raw stream with a 32 KiB window (mz_strm_zlib.c:97 inflateInit2(-15), :158
 * inflate(Z_SYNC_FLUSH)).  zlib itself is not under /root/reference (system
 * zlib 1.2.11 here), so this file restates the published format the reference
 * ships as doc/zip/appnote.txt:2030-2166 (identical to RFC 1951):
 *   block header / types          appnote.txt:2041-2063
 *   stored blocks                 appnote.txt:2045-2049
 *   fixed Huffman code            appnote.txt:2050-2059
 *   code-length alphabet + order  appnote.txt:2070-2090
 *   length / distance tables      appnote.txt:2107-2133
 *   decode loop                   appnote.txt:2139-2161
 * Deliberately the plain canonical-code decoder (count[]/symbol[] walk, one
 * bit at a time): slow, but with no lookup-table cleverness to get wrong.  The
 * decoder's shape -- count[] / symbol[] tables, the `code - count < first` walk,
 * the MAXBITS / MAXLCODES / MAXDCODES / FIXLCODES names -- is that of Mark Adler's
 * puff.c (zlib contrib/puff, the reference implementation written to specify
 * inflate unambiguously; it is not part of /root/reference), restated here.
 *
 * Error classes follow zlib 1.2.11's (they surface through
 * mz_stream_zlib_read, mz_strm_zlib.c:159-189): every malformed construct is
 * Z_DATA_ERROR (-3), running out of input is Z_BUF_ERROR (-5).
 */
#include "oracle.h"

#include <string.h>

#define MAXBITS 15
#define MAXLCODES 286
#define MAXDCODES 30
#define FIXLCODES 288

typedef struct {
    const uint8_t *in;
    size_t in_len;
    size_t in_pos;
    uint32_t bitbuf;
    int bitcnt;
    uint8_t *out;
    size_t out_cap;
    size_t out_pos;
    int err;
} st_t;

typedef struct {
    uint16_t count[MAXBITS + 1];
    uint16_t symbol[FIXLCODES];
    int maxlen; /* longest code of the set (0: the set is empty) */
} huff_t;

static uint32_t getbits(st_t *s, int need) {
    uint32_t val = s->bitbuf;
    while (s->bitcnt < need) {
        if (s->in_pos == s->in_len) {
            s->err = ORC_BUF_ERROR;
            return 0;
        }
        val |= (uint32_t)s->in[s->in_pos++] << s->bitcnt;
        s->bitcnt += 8;
    }
    s->bitbuf = need < 32 ? (val >> need) : 0;
    s->bitcnt -= need;
    return need < 32 ? (val & ((1u << need) - 1)) : val;
}

/* canonical decode, appnote.txt:2091-2106 (codes of one length are
 * consecutive values, shorter codes precede longer ones) */
static int decode_sym(st_t *s, const huff_t *h) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= MAXBITS; len++) {
        code |= (int)getbits(s, 1);
        if (s->err)
            return -1;
        int count = h->count[len];
        if (code - count < first)
            return h->symbol[index + (code - first)];
        if (len >= h->maxlen)
            return -2; /* an unused code of an incomplete set.  zlib 1.2.11 only lets a set be incomplete when its longest
                        * code has one bit (inflate_table(): "max != 1"), or empty, and marks the unused one-bit entries
                        * invalid: inflate() refuses the stream on that ONE bit -- it does not ask for the fourteen a
                        * bit-by-bit walk would still want, which is Z_BUF_ERROR instead of Z_DATA_ERROR when the input ends
                        * right there (tests/fuzz_gpu.py seed 606: two of 200 000 streams; tests/test_oracle.py) */
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -2;
}

/* returns 0 complete, >0 incomplete (bits left), <0 over-subscribed */
static int build(huff_t *h, const uint8_t *length, int n) {
    uint16_t offs[MAXBITS + 1];
    memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; i++)
        h->count[length[i]]++;
    h->maxlen = 0;
    for (int len = 1; len <= MAXBITS; len++)
        if (h->count[len])
            h->maxlen = len;
    int left = 1;
    for (int len = 1; len <= MAXBITS; len++) {
        left <<= 1;
        left -= h->count[len];
        if (left < 0)
            return left;
    }
    offs[1] = 0;
    for (int len = 1; len < MAXBITS; len++)
        offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++)
        if (length[i])
            h->symbol[offs[length[i]]++] = (uint16_t)i;
    return left;
}

static int max_len(const uint8_t *length, int n) {
    int m = 0;
    for (int i = 0; i < n; i++)
        if (length[i] > m)
            m = length[i];
    return m;
}

static const uint16_t k_lbase[29] = {3,  4,  5,  6,  7,  8,  9,  10, 11,  13,  15,  17,  19,  23, 27,
                                     31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t k_lext[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t k_dbase[30] = {1,   2,   3,   4,   5,   7,    9,    13,   17,   25,   33,   49,   65,    97,    129,
                                     193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t k_dext[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static int codes(st_t *s, const huff_t *lc, const huff_t *dc) {
    for (;;) {
        int sym = decode_sym(s, lc);
        if (s->err)
            return s->err;
        if (sym < 0)
            return ORC_DATA_ERROR; /* invalid literal/length code */
        if (sym < 256) {
            if (s->out_pos == s->out_cap)
                return ORC_OUT_FULL;
            s->out[s->out_pos++] = (uint8_t)sym;
        } else if (sym == 256) {
            return ORC_OK;
        } else {
            sym -= 257;
            if (sym >= 29)
                return ORC_DATA_ERROR; /* 286, 287: invalid literal/length code */
            uint32_t len = k_lbase[sym] + getbits(s, k_lext[sym]);
            if (s->err)
                return s->err;
            int ds = decode_sym(s, dc);
            if (s->err)
                return s->err;
            if (ds < 0 || ds >= 30)
                return ORC_DATA_ERROR; /* invalid distance code */
            uint32_t dist = k_dbase[ds] + getbits(s, k_dext[ds]);
            if (s->err)
                return s->err;
            if (dist > s->out_pos)
                return ORC_DATA_ERROR; /* invalid distance too far back */
            while (len--) {
                if (s->out_pos == s->out_cap)
                    return ORC_OUT_FULL;
                s->out[s->out_pos] = s->out[s->out_pos - dist];
                s->out_pos++;
            }
        }
    }
}

static int stored(st_t *s) {
    /* appnote.txt:2045-2049: skip to a byte boundary, LEN, NLEN, bytes */
    s->bitbuf = 0;
    s->bitcnt = 0;
    if (s->in_len - s->in_pos < 4) {
        s->in_pos = s->in_len;
        return ORC_BUF_ERROR;
    }
    uint32_t len = s->in[s->in_pos] | ((uint32_t)s->in[s->in_pos + 1] << 8);
    uint32_t nlen = s->in[s->in_pos + 2] | ((uint32_t)s->in[s->in_pos + 3] << 8);
    s->in_pos += 4;
    if (len != (~nlen & 0xFFFFu))
        return ORC_DATA_ERROR; /* invalid stored block lengths */
    while (len) {
        if (s->in_pos == s->in_len)
            return ORC_BUF_ERROR;
        if (s->out_pos == s->out_cap)
            return ORC_OUT_FULL;
        s->out[s->out_pos++] = s->in[s->in_pos++];
        len--;
    }
    return ORC_OK;
}

static int fixed(st_t *s) {
    static huff_t lc, dc;
    static int ready;
    if (!ready) {
        uint8_t l[FIXLCODES];
        int i = 0;
        for (; i < 144; i++) l[i] = 8;
        for (; i < 256; i++) l[i] = 9;
        for (; i < 280; i++) l[i] = 7;
        for (; i < FIXLCODES; i++) l[i] = 8;
        build(&lc, l, FIXLCODES);
        for (i = 0; i < 32; i++) l[i] = 5;
        build(&dc, l, 32); /* symbols 30,31 decode but are rejected in codes() */
        ready = 1;
    }
    return codes(s, &lc, &dc);
}

static int dynamic(st_t *s) {
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t lengths[MAXLCODES + 32 + 2];
    huff_t lc, dc;

    int nlen = (int)getbits(s, 5) + 257;
    int ndist = (int)getbits(s, 5) + 1;
    int ncode = (int)getbits(s, 4) + 4;
    if (s->err)
        return s->err;
    if (nlen > MAXLCODES || ndist > MAXDCODES)
        return ORC_DATA_ERROR; /* too many length or distance symbols */

    memset(lengths, 0, sizeof(lengths));
    for (int i = 0; i < ncode; i++) {
        lengths[order[i]] = (uint8_t)getbits(s, 3);
        if (s->err)
            return s->err;
    }
    /* zlib 1.2.11 requires a COMPLETE code-length code (inftrees.c, type
     * CODES) -- except the set with no code at all: inflate_table() hands that
     * back as a table of invalid one-bit entries ("no symbols, but wait for
     * decoding to report error"), inflate()'s CODELENS state reads every one
     * of the nlen + ndist lengths through it as a 0 of one bit, and the block
     * then fails the end-of-block check.  A data error nlen + ndist bits later
     * -- or "input exhausted" when the input ends inside those bits. */
    if (max_len(lengths, 19) == 0) {
        for (int i = 0; i < nlen + ndist; i++) {
            (void)getbits(s, 1);
            if (s->err)
                return s->err;
        }
        return ORC_DATA_ERROR; /* invalid code -- missing end-of-block */
    }
    if (build(&lc, lengths, 19) != 0)
        return ORC_DATA_ERROR; /* invalid code lengths set */

    int idx = 0;
    while (idx < nlen + ndist) {
        int sym = decode_sym(s, &lc);
        if (s->err)
            return s->err;
        if (sym < 0)
            return ORC_DATA_ERROR;
        if (sym < 16) {
            lengths[idx++] = (uint8_t)sym;
        } else {
            int rep, val = 0;
            if (sym == 16) {
                if (idx == 0)
                    return ORC_DATA_ERROR; /* invalid bit length repeat */
                val = lengths[idx - 1];
                rep = 3 + (int)getbits(s, 2);
            } else if (sym == 17) {
                rep = 3 + (int)getbits(s, 3);
            } else {
                rep = 11 + (int)getbits(s, 7);
            }
            if (s->err)
                return s->err;
            if (idx + rep > nlen + ndist)
                return ORC_DATA_ERROR; /* invalid bit length repeat */
            while (rep--)
                lengths[idx++] = (uint8_t)val;
        }
    }
    if (lengths[256] == 0)
        return ORC_DATA_ERROR; /* invalid code -- missing end-of-block */

    /* incomplete sets are accepted only when the longest code is 1 bit
     * (zlib 1.2.11 inftrees.c: "left > 0 && (type == CODES || max != 1)");
     * the distance set may also be empty. */
    int left = build(&lc, lengths, nlen);
    if (left < 0 || (left > 0 && max_len(lengths, nlen) != 1))
        return ORC_DATA_ERROR; /* invalid literal/lengths set */
    left = build(&dc, lengths + nlen, ndist);
    int dmax = max_len(lengths + nlen, ndist);
    if (left < 0 || (left > 0 && dmax > 1))
        return ORC_DATA_ERROR; /* invalid distances set */
    return codes(s, &lc, &dc);
}

int32_t orc_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, size_t *in_used,
                        size_t *out_len) {
    st_t s;
    memset(&s, 0, sizeof(s));
    s.in = in;
    s.in_len = in_len;
    s.out = out;
    s.out_cap = out_cap;
    int err = ORC_OK, last;
    do {
        last = (int)getbits(&s, 1);
        int type = (int)getbits(&s, 2);
        if (s.err) {
            err = s.err;
            break;
        }
        if (type == 0)
            err = stored(&s);
        else if (type == 1)
            err = fixed(&s);
        else if (type == 2)
            err = dynamic(&s);
        else
            err = ORC_DATA_ERROR; /* invalid block type */
    } while (err == ORC_OK && !last);
    /* whole bytes still sitting in the bit buffer were never consumed
     * (zlib hands them back; TOTAL_IN must be exact, mz_zip.c:2090,2116) */
    if (in_used)
        *in_used = s.in_pos - (size_t)(s.bitcnt >> 3);
    if (out_len)
        *out_len = s.out_pos;
    return err;
}
