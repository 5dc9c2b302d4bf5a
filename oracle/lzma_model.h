/* lzma_model.h -- the LZMA1 probability model, range decoder and packet loop shared by the two oracle
 * restatements that need them: lzma_dec.c (method 14, ".lzma alone" framing with an end marker) and xz_dec.c
 * (method 95, LZMA2 chunks inside the .xz container).  TEST INFRASTRUCTURE; see lzma_dec.c for the sources the
 * algorithm is restated from (public-domain LZMA specification; liblzma 5.2.5 observed through oracle/_ref).
 */
#ifndef ORC_LZMA_MODEL_H
#define ORC_LZMA_MODEL_H

#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* study hooks (tests/study/lzma_lockstep.c counts decisions by kind and packet); nothing in an ordinary build */
#ifndef ORC_TRACE_BIT
#define ORC_TRACE_BIT(p_) ((void)0)
#define ORC_TRACE_DIRECT(n_) ((void)0)
#define ORC_TRACE_PACKET() ((void)0)
#endif

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
    ORC_TRACE_BIT(p);
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
    ORC_TRACE_DIRECT(n);
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

/* decoder state that survives from one LZMA2 chunk to the next */
typedef struct {
    rc_t rc;
    model_t m;
    uint16_t *lit; /* 0x300 << (lc + lp) entries */
    unsigned lc, lp, pb;
    unsigned state;
    uint32_t rep0, rep1, rep2, rep3;
    uint8_t *out;
    size_t out_cap, opos;
    size_t dict_start; /* output position of the last dictionary reset (0 for LZMA1) */
    uint64_t dict;     /* dictionary size after liblzma's rounding */
} lz_t;

static void lz_reset_state(lz_t *z) {
    fill(z->lit, (size_t)0x300 << (z->lc + z->lp));
    fill((uint16_t *)&z->m, sizeof(model_t) / sizeof(uint16_t));
    z->state = 0;
    z->rep0 = z->rep1 = z->rep2 = z->rep3 = 0;
}

/* The packet loop.  lzma2 == 0: runs to the end-of-stream marker (ORC_OK iff the coder then holds 0).
 * lzma2 == 1: runs until exactly `limit` bytes of output exist (the chunk's uncompressed size); a match that
 * would cross the limit and an end marker are data errors, as in liblzma's lzma_decoder with a known size.
 * Any other return is the failure (ORC_DATA_ERROR also for input that ends early, like mz_stream_lzma_read
 * reports it; rc.eof tells the two apart). */
static int32_t lz_run(lz_t *z, size_t limit, int lzma2) {
    rc_t *rc = &z->rc;
    model_t *m = &z->m;
    uint8_t *out = z->out;
    const unsigned pb_mask = (1u << z->pb) - 1, lp_mask = (1u << z->lp) - 1, lc = z->lc;
    for (;;) {
        ORC_TRACE_PACKET();
        if (rc->eof)
            return ORC_DATA_ERROR; /* truncated */
        if (lzma2 && z->opos == limit)
            return ORC_OK;
        const size_t opos = z->opos;
        const size_t avail = opos - z->dict_start; /* bytes a distance may reach back over */
        unsigned ps = (unsigned)opos & pb_mask;
        unsigned state = z->state;
        if (!rc_bit(rc, &m->is_match[state][ps])) {
            unsigned prev = avail ? out[opos - 1] : 0;
            uint16_t *p = z->lit + (size_t)0x300 * ((((unsigned)opos & lp_mask) << lc) + (prev >> (8 - lc)));
            unsigned sym = 1;
            if (state >= 7) {
                unsigned mb = out[opos - z->rep0 - 1];
                do {
                    unsigned mbit = (mb >> 7) & 1;
                    mb <<= 1;
                    unsigned b = rc_bit(rc, &p[((1 + mbit) << 8) + sym]);
                    sym = (sym << 1) | b;
                    if (mbit != b)
                        break;
                } while (sym < 0x100);
            }
            while (sym < 0x100)
                sym = (sym << 1) | rc_bit(rc, &p[sym]);
            if (rc->eof)
                return ORC_DATA_ERROR;
            if (opos == z->out_cap)
                return ORC_OUT_FULL;
            out[z->opos++] = (uint8_t)sym;
            z->state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);
            continue;
        }
        unsigned len;
        if (rc_bit(rc, &m->is_rep[state])) {
            if (avail == 0)
                return ORC_DATA_ERROR; /* rep with empty dictionary */
            if (!rc_bit(rc, &m->is_rep_g0[state])) {
                if (!rc_bit(rc, &m->is_rep0_long[state][ps])) {
                    if (rc->eof)
                        return ORC_DATA_ERROR;
                    if (z->rep0 >= avail || z->rep0 >= z->dict)
                        return ORC_DATA_ERROR;
                    if (opos == z->out_cap)
                        return ORC_OUT_FULL;
                    out[opos] = out[opos - z->rep0 - 1];
                    z->opos++;
                    z->state = state < 7 ? 9 : 11;
                    continue;
                }
            } else {
                uint32_t dist;
                if (!rc_bit(rc, &m->is_rep_g1[state])) {
                    dist = z->rep1;
                } else {
                    if (!rc_bit(rc, &m->is_rep_g2[state])) {
                        dist = z->rep2;
                    } else {
                        dist = z->rep3;
                        z->rep3 = z->rep2;
                    }
                    z->rep2 = z->rep1;
                }
                z->rep1 = z->rep0;
                z->rep0 = dist;
            }
            len = len_decode(rc, &m->rep_len, ps);
            z->state = state < 7 ? 8 : 11;
        } else {
            z->rep3 = z->rep2;
            z->rep2 = z->rep1;
            z->rep1 = z->rep0;
            len = len_decode(rc, &m->len, ps);
            z->state = state < 7 ? 7 : 10;
            unsigned slot = bittree(rc, m->pos_slot[len < 4 ? len : 3], 6);
            uint32_t rep0;
            if (slot < 4) {
                rep0 = slot;
            } else {
                int nb = (int)(slot >> 1) - 1;
                rep0 = (2 | (slot & 1)) << nb;
                if (slot < 14) {
                    rep0 += bittree_rev(rc, m->pos_dec + rep0 - slot, nb);
                } else {
                    rep0 += rc_direct(rc, nb - 4) << 4;
                    rep0 += bittree_rev(rc, m->align, 4);
                }
            }
            z->rep0 = rep0;
            if (rep0 == 0xFFFFFFFFu) {
                /* end-of-stream marker */
                if (rc->eof)
                    return ORC_DATA_ERROR;
                if (lzma2)
                    return ORC_DATA_ERROR; /* not allowed when the size is known */
                rc_norm(rc);
                if (rc->eof)
                    return ORC_DATA_ERROR;
                return (rc->code == 0) ? ORC_OK : ORC_DATA_ERROR;
            }
        }
        if (rc->eof)
            return ORC_DATA_ERROR;
        len += 2;
        if (z->rep0 >= avail || z->rep0 >= z->dict)
            return ORC_DATA_ERROR; /* distance beyond the dictionary */
        if (lzma2 && len > limit - opos)
            return ORC_DATA_ERROR; /* match runs past the chunk's uncompressed size */
        while (len--) {
            if (z->opos == z->out_cap)
                return ORC_OUT_FULL;
            out[z->opos] = out[z->opos - z->rep0 - 1];
            z->opos++;
        }
    }
}

#endif
