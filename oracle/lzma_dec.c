/* lzma_dec.c -- oracle restatement of raw-LZMA1 decoding (TEST INFRASTRUCTURE).
 *
 * The reference decodes a method-14 entry by stripping the 4-byte ZIP-LZMA
 * magic (mz_strm_lzma.c:118-124), turning the 5 property bytes plus eight
 * 0xFF bytes into a ".lzma alone" header with unknown size (:177-201) and
 * calling liblzma's lzma_alone_decoder / lzma_code (:126,:210).  liblzma is
 * not under /root/reference (5.2.5 in this image; un-pinned upstream), and the
 * range-coder algorithm is not described anywhere in the reference tree, so
 * this file restates the public-domain LZMA specification (LZMA SDK,
 * "lzma-specification.txt"): 11-bit adaptive probabilities, 12-state machine,
 * rep0..rep3, two length coders, 6-bit position slots, aligned low bits, and
 * the end-of-stream marker (distance 0xFFFFFFFF).  Container framing follows
 * doc/zip/appnote.txt:2232-2275.
 *
 * liblzma behaviours mirrored on purpose (observed on 5.2.5 via oracle/_ref):
 *   - the first range-coder byte must be 0x00 (else LZMA_DATA_ERROR, nothing consumed);
 *   - a distance is valid iff it is < min(bytes produced, dictionary size),
 *     dictionary size rounded up to >= 4096 and to a multiple of 16;
 *   - after the EOS marker the coder normalises once and requires code == 0;
 *   - every failure surfaces as MZ_DATA_ERROR (-3) from mz_stream_lzma_read
 *     (mz_strm_lzma.c:236-237), including truncated input.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define K_TOP (1u << 24)
#define K_BITS 11
#define K_MOVE 5

typedef struct {
    const uint8_t *in;
    size_t in_len;
    size_t in_pos;
    uint32_t range, code;
    int eof; /* tried to read past the end */
} rc_t;

static uint8_t rc_byte(rc_t *rc) {
    if (rc->in_pos >= rc->in_len) {
        rc->eof = 1;
        return 0;
    }
    return rc->in[rc->in_pos++];
}

static void rc_norm(rc_t *rc) {
    if (rc->range < K_TOP) {
        rc->range <<= 8;
        rc->code = (rc->code << 8) | rc_byte(rc);
    }
}

static unsigned rc_bit(rc_t *rc, uint16_t *p) {
    rc_norm(rc); /* liblzma normalises before, not after, each bit */
    uint32_t bound = (rc->range >> K_BITS) * *p;
    if (rc->code < bound) {
        rc->range = bound;
        *p += ((1u << K_BITS) - *p) >> K_MOVE;
        return 0;
    }
    rc->range -= bound;
    rc->code -= bound;
    *p -= *p >> K_MOVE;
    return 1;
}

static uint32_t rc_direct(rc_t *rc, int n) {
    uint32_t r = 0;
    while (n--) {
        rc_norm(rc);
        rc->range >>= 1;
        rc->code -= rc->range;
        uint32_t t = 0u - (rc->code >> 31);
        rc->code += rc->range & t;
        r = (r << 1) + (t + 1);
    }
    return r;
}

static unsigned bittree(rc_t *rc, uint16_t *p, int nbits) {
    unsigned m = 1;
    for (int i = 0; i < nbits; i++)
        m = (m << 1) + rc_bit(rc, &p[m]);
    return m - (1u << nbits);
}

static unsigned bittree_rev(rc_t *rc, uint16_t *p, int nbits) {
    unsigned m = 1, sym = 0;
    for (int i = 0; i < nbits; i++) {
        unsigned b = rc_bit(rc, &p[m]);
        m = (m << 1) + b;
        sym |= b << i;
    }
    return sym;
}

typedef struct {
    uint16_t choice, choice2;
    uint16_t low[16][8];
    uint16_t mid[16][8];
    uint16_t high[256];
} len_t;

static unsigned len_decode(rc_t *rc, len_t *l, unsigned pos_state) {
    if (!rc_bit(rc, &l->choice))
        return bittree(rc, l->low[pos_state], 3);
    if (!rc_bit(rc, &l->choice2))
        return 8 + bittree(rc, l->mid[pos_state], 3);
    return 16 + bittree(rc, l->high, 8);
}

typedef struct {
    uint16_t is_match[12][16];
    uint16_t is_rep[12], is_rep_g0[12], is_rep_g1[12], is_rep_g2[12];
    uint16_t is_rep0_long[12][16];
    uint16_t pos_slot[4][64];
    uint16_t pos_dec[1 + 128 - 14]; /* indexed by base - slot + tree node */
    uint16_t align[16];
    len_t len, rep_len;
} model_t;

static void fill(uint16_t *p, size_t n) {
    for (size_t i = 0; i < n; i++)
        p[i] = 1u << (K_BITS - 1);
}

int32_t orc_lzma_zip_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, int64_t max_out,
                            size_t *in_used, size_t *out_len) {
    size_t opos = 0;
    int32_t ret = ORC_DATA_ERROR;
    uint16_t *lit = NULL;
    model_t *m = NULL;
    rc_t rc;
    memset(&rc, 0, sizeof(rc));

    if (in_used) *in_used = 0;
    if (out_len) *out_len = 0;
    /* 4-byte magic (ignored by the reference, mz_strm_lzma.c:118-121) + 5 props */
    if (in_len < 9) {
        if (in_used) *in_used = in_len;
        return ORC_DATA_ERROR;
    }
    unsigned d = in[4];
    if (d >= 9 * 5 * 5) {
        if (in_used) *in_used = 9;
        return ORC_DATA_ERROR; /* LZMA_FORMAT_ERROR -> MZ_DATA_ERROR */
    }
    unsigned lc = d % 9;
    d /= 9;
    unsigned lp = d % 5, pb = d / 5;
    uint64_t dict = in[5] | ((uint32_t)in[6] << 8) | ((uint32_t)in[7] << 16) | ((uint32_t)in[8] << 24);
    if (dict < 4096)
        dict = 4096;
    dict = (dict + 15) & ~(uint64_t)15;

    size_t nlit = (size_t)0x300 << (lc + lp);
    lit = (uint16_t *)malloc(nlit * sizeof(uint16_t));
    m = (model_t *)malloc(sizeof(model_t));
    if (!lit || !m)
        goto done;
    fill(lit, nlit);
    fill((uint16_t *)m, sizeof(model_t) / sizeof(uint16_t));

    rc.in = in;
    rc.in_len = in_len;
    rc.in_pos = 9;
    rc.range = 0xFFFFFFFFu;
    if (in_len > 9 && in[9] != 0)
        goto done; /* liblzma 5.2.5 rejects a non-zero first byte before consuming it */
    for (int i = 0; i < 5; i++)
        rc.code = (rc.code << 8) | rc_byte(&rc);
    if (rc.eof)
        goto done;

    unsigned state = 0;
    uint32_t rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0;
    const unsigned pb_mask = (1u << pb) - 1, lp_mask = (1u << lp) - 1;

    for (;;) {
        if (rc.eof)
            goto done; /* truncated */
        unsigned ps = (unsigned)opos & pb_mask;
        if (!rc_bit(&rc, &m->is_match[state][ps])) {
            unsigned prev = opos ? out[opos - 1] : 0;
            uint16_t *p = lit + (size_t)0x300 * ((((unsigned)opos & lp_mask) << lc) + (prev >> (8 - lc)));
            unsigned sym = 1;
            if (state >= 7) {
                unsigned mb = out[opos - rep0 - 1];
                do {
                    unsigned mbit = (mb >> 7) & 1;
                    mb <<= 1;
                    unsigned b = rc_bit(&rc, &p[((1 + mbit) << 8) + sym]);
                    sym = (sym << 1) | b;
                    if (mbit != b)
                        break;
                } while (sym < 0x100);
            }
            while (sym < 0x100)
                sym = (sym << 1) | rc_bit(&rc, &p[sym]);
            if (rc.eof)
                goto done;
            if (opos == out_cap) {
                ret = ORC_OUT_FULL;
                goto done;
            }
            out[opos++] = (uint8_t)sym;
            state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);
            continue;
        }
        unsigned len;
        if (rc_bit(&rc, &m->is_rep[state])) {
            if (opos == 0)
                goto done; /* rep with empty dictionary */
            if (!rc_bit(&rc, &m->is_rep_g0[state])) {
                if (!rc_bit(&rc, &m->is_rep0_long[state][ps])) {
                    if (rc.eof)
                        goto done;
                    if (rep0 >= opos || rep0 >= dict)
                        goto done;
                    if (opos == out_cap) {
                        ret = ORC_OUT_FULL;
                        goto done;
                    }
                    out[opos] = out[opos - rep0 - 1];
                    opos++;
                    state = state < 7 ? 9 : 11;
                    continue;
                }
            } else {
                uint32_t dist;
                if (!rc_bit(&rc, &m->is_rep_g1[state])) {
                    dist = rep1;
                } else {
                    if (!rc_bit(&rc, &m->is_rep_g2[state])) {
                        dist = rep2;
                    } else {
                        dist = rep3;
                        rep3 = rep2;
                    }
                    rep2 = rep1;
                }
                rep1 = rep0;
                rep0 = dist;
            }
            len = len_decode(&rc, &m->rep_len, ps);
            state = state < 7 ? 8 : 11;
        } else {
            rep3 = rep2;
            rep2 = rep1;
            rep1 = rep0;
            len = len_decode(&rc, &m->len, ps);
            state = state < 7 ? 7 : 10;
            unsigned slot = bittree(&rc, m->pos_slot[len < 4 ? len : 3], 6);
            if (slot < 4) {
                rep0 = slot;
            } else {
                int nb = (int)(slot >> 1) - 1;
                rep0 = (2 | (slot & 1)) << nb;
                if (slot < 14) {
                    rep0 += bittree_rev(&rc, m->pos_dec + rep0 - slot, nb);
                } else {
                    rep0 += rc_direct(&rc, nb - 4) << 4;
                    rep0 += bittree_rev(&rc, m->align, 4);
                }
            }
            if (rep0 == 0xFFFFFFFFu) {
                /* end-of-stream marker */
                if (rc.eof)
                    goto done;
                rc_norm(&rc);
                if (rc.eof)
                    goto done;
                ret = (rc.code == 0) ? ORC_OK : ORC_DATA_ERROR;
                goto done;
            }
        }
        if (rc.eof)
            goto done;
        len += 2;
        if (rep0 >= opos || rep0 >= dict)
            goto done; /* distance beyond the dictionary */
        while (len--) {
            if (opos == out_cap) {
                ret = ORC_OUT_FULL;
                goto done;
            }
            out[opos] = out[opos - rep0 - 1];
            opos++;
        }
    }

done:
    free(lit);
    free(m);
    if (in_used)
        *in_used = rc.in ? rc.in_pos : 9;
    if (out_len)
        *out_len = (max_out >= 0 && (int64_t)opos > max_out) ? (size_t)max_out : opos;
    return ret;
}
