/* inflate_core.h -- raw-DEFLATE decode of ONE entry by ONE wavefront, with the
 * entry's CRC-32 fused in (kernel K1+K2 of SURVEY 2.1).
 *
 * Replaces what the reference does per entry through mz_stream_zlib_read
 * (mz_strm_zlib.c:116-193 -> zlib inflate(), raw, 32 KiB window) followed by
 * mz_crypt_crc32_update (mz_zip.c:2049).  Format: doc/zip/appnote.txt:2030-2166.
 *
 * MI355X mapping (not a port of zlib's byte-serial state machine):
 *   - Huffman tables are built by the whole wave (histogram -> canonical first
 *     codes -> ranked symbols -> LDS lookup table), see mz_build_huff().
 *   - Symbol decode is SPECULATIVE AND PARALLEL: in every step lane l decodes
 *     the complete token (literal | length+distance | end-of-block) that would
 *     start at bit cursor+l, for all 64 bit offsets at once.  A short
 *     wave-uniform chain walk (v_readlane hops on the scalar unit) then picks
 *     the offsets that really are token starts.  One step therefore retires
 *     >= 64 bits of compressed input (about 6 tokens on text) for roughly the
 *     cost of one serial symbol decode.
 *   - Selected literals are scattered by their lanes in one store; matches
 *     (LZ77 back-references) are copied cooperatively, 64 bytes per
 *     instruction, overlapping (dist < len) runs included.
 *   - The sliding window IS the output buffer: back-references read bytes this
 *     wave wrote earlier (L1/L2-resident; a wave's vector-memory operations
 *     execute in order), so no 32 KiB LDS window is needed and 32 waves/CU fit.
 *   - CRC-32 is folded from the freshly written output one 1 KiB tile at a
 *     time (crc32_core.h), so the output is never re-read from HBM.
 *
 * Error classes mirror zlib's as the reference surfaces them
 * (mz_strm_zlib.c:159-189): malformed data -> -3, input exhausted -> -5.
 */
#ifndef MZHIP_INFLATE_CORE_H
#define MZHIP_INFLATE_CORE_H

#include "crc32_core.h"
#include "wave.h"

#define MZ_LROOT 10 /* literal/length fast-table index bits */
#define MZ_DROOT 9  /* distance fast-table index bits        */
#define MZ_CROOT 7  /* code-length-code table bits (== max)  */

/* per-wave LDS scratch */
typedef struct mz_inflate_lds {
    uint16_t lit_fast[1 << MZ_LROOT]; /* (symbol << 4) | code length, 0 = not a short code */
    uint16_t dist_fast[1 << MZ_DROOT];
    uint16_t clc_fast[1 << MZ_CROOT];
    uint16_t lit_sym[288]; /* symbols sorted by (length, value): canonical order */
    uint16_t dist_sym[32];
    uint16_t clc_sym[20];
    uint16_t lit_first[16], lit_count[16], lit_offs[16];
    uint16_t dist_first[16], dist_count[16], dist_offs[16];
    uint16_t clc_first[16], clc_count[16], clc_offs[16];
    uint16_t rank_base[16];
    uint32_t hist[16];
    uint8_t cl[320]; /* code lengths of the current block (nlen + ndist <= 316; fixed: 288 + 32) */
    uint8_t clc_len[20]; /* lengths of the code-length code */
} mz_inflate_lds;

typedef struct mz_inflate_result {
    int32_t status;
    uint32_t out_len;
    uint32_t in_used;
    uint32_t crc;
} mz_inflate_result;

/* 64 bits of the stream starting at bit `bitpos`, LSB first, zero-padded past
 * the end.  Aligned dword loads; the slow path assembles bytes near the end so
 * no byte outside [in, in+in_len) is ever touched. */
MZ_DEV uint64_t mz_bits_at(const uint8_t *in, uint32_t in_len, uint64_t bitpos) {
    uint32_t byte = (uint32_t)(bitpos >> 3);
    uint32_t sh = (uint32_t)bitpos & 7u;
    const uint8_t *p = in + byte;
    uint32_t mis = (uint32_t)((uintptr_t)p & 3u);
    if (byte >= mis && (uint64_t)byte - mis + 12u <= in_len) {
        const uint32_t *q = (const uint32_t *)(p - mis);
        uint32_t d0 = q[0], d1 = q[1], d2 = q[2];
        uint32_t s = mis * 8u + sh; /* 0..31 */
        uint64_t lo = ((uint64_t)d1 << 32) | d0;
        uint64_t w = lo >> s;
        if (s) w |= (uint64_t)d2 << (64u - s);
        return w;
    }
    uint64_t w = 0;
    uint32_t top = 0;
    for (uint32_t i = 0; i < 9; i++) {
        uint32_t b = (byte + i < in_len) ? in[byte + i] : 0u;
        if (i < 8)
            w |= (uint64_t)b << (8u * i);
        else
            top = b;
    }
    w >>= sh;
    if (sh) w |= (uint64_t)top << (64u - sh);
    return w;
}

/* length symbol 257..285 -> (base, extra bits): appnote.txt:2107-2120, computed */
MZ_DEV void mz_len_base(uint32_t s /* sym-257 */, uint32_t *base, uint32_t *ext) {
    if (s < 8) {
        *base = 3 + s;
        *ext = 0;
    } else if (s == 28) {
        *base = 258;
        *ext = 0;
    } else {
        uint32_t e = (s - 4) >> 2;
        *base = 3 + ((4 + (s & 3)) << e);
        *ext = e;
    }
}
/* distance symbol 0..29 -> (base, extra bits): appnote.txt:2122-2133, computed */
MZ_DEV void mz_dist_base(uint32_t s, uint32_t *base, uint32_t *ext) {
    if (s < 4) {
        *base = 1 + s;
        *ext = 0;
    } else {
        uint32_t e = (s - 2) >> 1;
        *base = 1 + ((2 + (s & 1)) << e);
        *ext = e;
    }
}

/* canonical search for codes longer than the fast table's root
 * (appnote.txt:2091-2106): v15 = next 15 stream bits, MSB-first. */
MZ_DEV uint32_t mz_canon_slow(uint32_t lo, int root, const uint16_t *first, const uint16_t *count,
                              const uint16_t *offs, const uint16_t *symtab, uint32_t *nbits) {
    uint32_t v15 = mz_brev32(lo) >> 17;
    for (int L = root + 1; L <= 15; L++) {
        uint32_t c = v15 >> (15 - L);
        uint32_t d = c - first[L];
        if (d < count[L]) {
            *nbits = (uint32_t)L;
            return symtab[offs[L] + d];
        }
    }
    *nbits = 0;
    return 0;
}

/* A decoded candidate token.
 *   bits : total compressed bits (0 = no valid token starts here)
 *   olen : bytes it produces (1 literal, 3..258 match, 0 end-of-block)
 *   val  : literal byte | match distance | for bits==0: bits the verdict needed */
MZ_DEV void mz_decode_token(uint64_t w, const mz_inflate_lds *t, uint32_t *bits, uint32_t *olen, uint32_t *val) {
    uint32_t lo = (uint32_t)w;
    uint32_t e = t->lit_fast[lo & ((1u << MZ_LROOT) - 1)];
    uint32_t nb = e & 15u, sym = e >> 4;
    if (nb == 0) {
        sym = mz_canon_slow(lo, MZ_LROOT, t->lit_first, t->lit_count, t->lit_offs, t->lit_sym, &nb);
        if (nb == 0) {
            *bits = 0; *olen = 0; *val = 15;
            return;
        }
    }
    if (sym < 256) {
        *bits = nb; *olen = 1; *val = sym;
        return;
    }
    if (sym == 256) {
        *bits = nb; *olen = 0; *val = 0;
        return;
    }
    if (sym > 285) { /* 286, 287: invalid literal/length code */
        *bits = 0; *olen = 0; *val = nb;
        return;
    }
    uint32_t lbase, lext;
    mz_len_base(sym - 257, &lbase, &lext);
    uint32_t len = lbase + ((uint32_t)(w >> nb) & ((1u << lext) - 1));
    nb += lext;
    uint32_t dlo = (uint32_t)(w >> nb);
    uint32_t d = t->dist_fast[dlo & ((1u << MZ_DROOT) - 1)];
    uint32_t dn = d & 15u, dsym = d >> 4;
    if (dn == 0) {
        dsym = mz_canon_slow(dlo, MZ_DROOT, t->dist_first, t->dist_count, t->dist_offs, t->dist_sym, &dn);
        if (dn == 0) {
            *bits = 0; *olen = 0; *val = nb + 15;
            return;
        }
    }
    if (dsym > 29) { /* 30, 31: invalid distance code */
        *bits = 0; *olen = 0; *val = nb + dn;
        return;
    }
    uint32_t dbase, dext;
    mz_dist_base(dsym, &dbase, &dext);
    uint32_t dist = dbase + ((dlo >> dn) & ((1u << dext) - 1));
    *bits = nb + dn + dext; /* <= 15+5+15+13 = 48 */
    *olen = len;
    *val = dist;
}

/* Build one Huffman decoding table from code lengths cl[0..n) with the whole
 * wave.  Returns (wave-uniform) the number of unused codes `left` (>0
 * incomplete, <0 over-subscribed) and the longest length in *maxlen. */
#define MZ_BUILD_HUFF(left_out, maxlen_out, L_, cl_, n_, fast_, root_, symtab_, first_, count_, offs_)         \
    do {                                                                                                       \
        MZ_LANES {                                                                                             \
            if (lane < 16) { (L_)->hist[lane] = 0; (L_)->rank_base[lane] = 0; }                                \
            for (int _k = lane; _k < (1 << (root_)) / 2; _k += 64) ((uint32_t *)(fast_))[_k] = 0;              \
        }                                                                                                      \
        MZ_WAVE_SYNC();                                                                                        \
        MZ_LANES {                                                                                             \
            for (int _s = lane; _s < (int)(n_); _s += 64) {                                                    \
                uint32_t _l = (cl_)[_s];                                                                       \
                if (_l) MZ_LDS_ATOMIC_INC(&(L_)->hist[_l]);                                                    \
            }                                                                                                  \
        }                                                                                                      \
        MZ_WAVE_SYNC();                                                                                        \
        int32_t _left = 1, _over = 0;                                                                          \
        uint32_t _code = 0, _off = 0, _max = 0;                                                                \
        for (int _l = 1; _l <= 15; _l++) {                                                                     \
            uint32_t _c = MZ_UNIFORM((L_)->hist[_l]);                                                          \
            _left = (_left << 1) - (int32_t)_c;                                                                \
            if (_left < 0) _over = 1;                                                                          \
            MZ_LANES { /* uniform store: every lane writes the same value (no lane-0 branch) */              \
                (first_)[_l] = (uint16_t)_code;                                                                \
                (count_)[_l] = (uint16_t)_c;                                                                   \
                (offs_)[_l] = (uint16_t)_off;                                                                  \
            }                                                                                                  \
            _code = (_code + _c) << 1;                                                                         \
            _off += _c;                                                                                        \
            if (_c) _max = (uint32_t)_l;                                                                       \
        }                                                                                                      \
        MZ_WAVE_SYNC();                                                                                        \
        if (!_over) {                                                                                          \
            for (int _base = 0; _base < (int)(n_); _base += 64) {                                              \
                PV(uint32_t, _len);                                                                            \
                PV(uint32_t, _rank);                                                                           \
                MZ_LANES {                                                                                     \
                    int _s = _base + lane;                                                                     \
                    P(_len) = (_s < (int)(n_)) ? (cl_)[_s] : 0u;                                               \
                    P(_rank) = 0;                                                                              \
                }                                                                                              \
                uint64_t _pending;                                                                             \
                MZ_BALLOT(_pending, P(_len) != 0);                                                             \
                while (_pending) {                                                                             \
                    uint32_t _t = mz_ctz64(_pending);                                                          \
                    uint32_t _lt = MZ_READLANE(_len, _t);                                                      \
                    uint64_t _m;                                                                               \
                    MZ_BALLOT(_m, P(_len) == _lt);                                                             \
                    uint32_t _rb = MZ_UNIFORM((L_)->rank_base[_lt]);                                           \
                    MZ_LANES {                                                                                 \
                        if (P(_len) == _lt) P(_rank) = _rb + mz_popc64(_m & ((1ull << lane) - 1));            \
                        (L_)->rank_base[_lt] = (uint16_t)(_rb + mz_popc64(_m)); /* uniform store */          \
                    }                                                                                          \
                    MZ_WAVE_SYNC();                                                                            \
                    _pending &= ~_m;                                                                           \
                }                                                                                              \
                MZ_LANES {                                                                                     \
                    uint32_t _l = P(_len);                                                                     \
                    if (_l) {                                                                                  \
                        uint32_t _s = (uint32_t)(_base + lane);                                                \
                        uint32_t _cd = (uint32_t)(first_)[_l] + P(_rank);                                      \
                        (symtab_)[(offs_)[_l] + P(_rank)] = (uint16_t)_s;                                      \
                        if (_l <= (uint32_t)(root_)) {                                                         \
                            uint32_t _rv = mz_brev32(_cd) >> (32 - _l);                                        \
                            uint16_t _e = (uint16_t)((_s << 4) | _l);                                          \
                            for (uint32_t _k = _rv; _k < (1u << (root_)); _k += (1u << _l)) (fast_)[_k] = _e;  \
                        }                                                                                      \
                    }                                                                                          \
                }                                                                                              \
            }                                                                                                  \
        }                                                                                                      \
        MZ_WAVE_SYNC();                                                                                        \
        (left_out) = _over ? -1 : _left;                                                                       \
        (maxlen_out) = _max;                                                                                   \
    } while (0)

/* uniform n-bit read at the block-header level */
#define MZ_HDR_BITS(dst, n)                                                   \
    do {                                                                      \
        if (bitpos + (uint64_t)(n) > total_bits) {                            \
            status = MZHIP_BUF_ERROR;                                         \
            goto finish;                                                      \
        }                                                                     \
        uint64_t _w = mz_bits_at(in, in_len, bitpos);                         \
        (dst) = MZ_UNIFORM((uint32_t)_w & ((1u << (n)) - 1));                 \
        bitpos += (n);                                                        \
    } while (0)

/* transmission order of the code-length-code lengths, appnote.txt:2083-2090 */
#if defined(MZHIP_HOST_EMUL)
static const uint8_t mz_k_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#else
__device__ static const uint8_t mz_k_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
#endif

/* Decode one raw-DEFLATE entry.  All arguments are wave-uniform. */
MZ_DEV void mz_inflate_entry(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap,
                             mz_inflate_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs,
                             mz_inflate_result *res) {
    MZ_LANE_DECL
    const uint64_t total_bits = (uint64_t)in_len * 8u;
    uint64_t bitpos = 0;
    uint32_t out_pos = 0;
    int32_t status = MZHIP_OK;
    uint32_t last = 0;
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    MZ_LANES { P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u; }

    while (!last) {
        uint32_t hdr;
        MZ_HDR_BITS(hdr, 3);
        last = hdr & 1u;
        uint32_t btype = hdr >> 1;

        if (btype == 0) {
            /* stored block, appnote.txt:2045-2049 */
            uint32_t byte = (uint32_t)((bitpos + 7) >> 3);
            if ((uint64_t)byte + 4 > in_len) {
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            uint32_t len = MZ_UNIFORM((uint32_t)in[byte] | ((uint32_t)in[byte + 1] << 8));
            uint32_t nlen = MZ_UNIFORM((uint32_t)in[byte + 2] | ((uint32_t)in[byte + 3] << 8));
            byte += 4;
            bitpos = (uint64_t)byte * 8u;
            if (len != (~nlen & 0xFFFFu)) {
                status = MZHIP_DATA_ERROR; /* invalid stored block lengths */
                goto finish;
            }
            uint32_t avail = in_len - byte;
            uint32_t n = len < avail ? len : avail;
            if (n > out_cap - out_pos) {
                status = MZHIP_OUT_FULL;
                goto finish;
            }
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[out_pos + i] = in[byte + i];
            }
            MZ_WAVE_SYNC();
            out_pos += n;
            bitpos += (uint64_t)n * 8u;
            if (n < len) {
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, out_pos, crc_tab, tabs->kx);
            continue;
        }
        if (btype == 3) {
            status = MZHIP_DATA_ERROR; /* invalid block type */
            goto finish;
        }

        int32_t left;
        uint32_t maxlen;
        if (btype == 1) {
            /* fixed code, appnote.txt:2050-2059 */
            MZ_LANES {
                for (int s = lane; s < 288 + 32; s += 64)
                    L->cl[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
            }
            MZ_WAVE_SYNC();
            MZ_BUILD_HUFF(left, maxlen, L, L->cl, 288, L->lit_fast, MZ_LROOT, L->lit_sym, L->lit_first, L->lit_count,
                          L->lit_offs);
            MZ_BUILD_HUFF(left, maxlen, L, L->cl + 288, 32, L->dist_fast, MZ_DROOT, L->dist_sym, L->dist_first,
                          L->dist_count, L->dist_offs);
        } else {
            /* dynamic code, appnote.txt:2060-2106 */
            uint32_t h;
            MZ_HDR_BITS(h, 14);
            uint32_t nlen = (h & 31u) + 257, ndist = ((h >> 5) & 31u) + 1, ncode = (h >> 10) + 4;
            if (nlen > 286 || ndist > 30) {
                status = MZHIP_DATA_ERROR; /* too many length or distance symbols */
                goto finish;
            }
            if (bitpos + 3ull * ncode > total_bits) {
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            MZ_LANES {
                if (lane < 19) L->clc_len[lane] = 0;
            }
            MZ_WAVE_SYNC();
            MZ_LANES {
                if ((uint32_t)lane < ncode) {
                    uint64_t w = mz_bits_at(in, in_len, bitpos + 3u * (uint32_t)lane);
                    L->clc_len[mz_k_order[lane]] = (uint8_t)((uint32_t)w & 7u);
                }
            }
            bitpos += 3ull * ncode;
            MZ_WAVE_SYNC();
            MZ_BUILD_HUFF(left, maxlen, L, L->clc_len, 19, L->clc_fast, MZ_CROOT, L->clc_sym, L->clc_first, L->clc_count,
                          L->clc_offs);
            if (left != 0) {
                status = MZHIP_DATA_ERROR; /* invalid code lengths set */
                goto finish;
            }
            /* code lengths: serial by nature (run-length coded), wave-uniform loop
             * consuming a 64-bit window at a time */
            uint32_t idx = 0, prev = 0;
            const uint32_t ntot = nlen + ndist;
            while (idx < ntot) {
                uint64_t w = mz_bits_at(in, in_len, bitpos);
                uint32_t wlo = MZ_UNIFORM((uint32_t)w), whi = MZ_UNIFORM((uint32_t)(w >> 32));
                uint64_t wu = ((uint64_t)whi << 32) | wlo;
                uint32_t used = 0;
                while (idx < ntot && used + 14 <= 64) {
                    uint32_t e = MZ_UNIFORM(L->clc_fast[(uint32_t)(wu >> used) & 127u]);
                    uint32_t nb = e & 15u, sym = e >> 4;
                    if (nb == 0) {
                        status = (bitpos + used + 7 > total_bits) ? MZHIP_BUF_ERROR : MZHIP_DATA_ERROR;
                        goto finish;
                    }
                    uint32_t ext = sym < 16 ? 0u : sym == 16 ? 2u : sym == 17 ? 3u : 7u;
                    if (bitpos + used + nb + ext > total_bits) {
                        status = MZHIP_BUF_ERROR;
                        goto finish;
                    }
                    uint32_t xb = (uint32_t)(wu >> (used + nb)) & ((1u << ext) - 1);
                    used += nb + ext;
                    if (sym < 16) {
                        MZ_LANES { L->cl[idx] = (uint8_t)sym; } /* uniform store */
                        prev = sym;
                        idx++;
                    } else {
                        uint32_t rep, val = 0;
                        if (sym == 16) {
                            if (idx == 0) {
                                status = MZHIP_DATA_ERROR; /* invalid bit length repeat */
                                goto finish;
                            }
                            val = prev;
                            rep = 3 + xb;
                        } else if (sym == 17) {
                            rep = 3 + xb;
                        } else {
                            rep = 11 + xb;
                        }
                        if (idx + rep > ntot) {
                            status = MZHIP_DATA_ERROR; /* invalid bit length repeat */
                            goto finish;
                        }
                        MZ_LANES {
                            for (uint32_t k = (uint32_t)lane; k < rep; k += 64) L->cl[idx + k] = (uint8_t)val;
                        }
                        prev = val;
                        idx += rep;
                    }
                }
                bitpos += used;
            }
            MZ_WAVE_SYNC();
            if (MZ_UNIFORM(L->cl[256]) == 0) {
                status = MZHIP_DATA_ERROR; /* invalid code -- missing end-of-block */
                goto finish;
            }
            MZ_BUILD_HUFF(left, maxlen, L, L->cl, nlen, L->lit_fast, MZ_LROOT, L->lit_sym, L->lit_first,
                          L->lit_count, L->lit_offs);
            if (left < 0 || (left > 0 && maxlen != 1)) {
                status = MZHIP_DATA_ERROR; /* invalid literal/lengths set */
                goto finish;
            }
            MZ_BUILD_HUFF(left, maxlen, L, L->cl + nlen, ndist, L->dist_fast, MZ_DROOT, L->dist_sym, L->dist_first,
                          L->dist_count, L->dist_offs);
            if (left < 0 || (left > 0 && maxlen > 1)) {
                status = MZHIP_DATA_ERROR; /* invalid distances set */
                goto finish;
            }
        }

        /* ---- compressed block body: speculative 64-offset decode ---- */
        for (;;) {
            PV(uint32_t, tbits);
            PV(uint32_t, tolen);
            PV(uint32_t, tval);
            MZ_LANES {
                uint64_t w = mz_bits_at(in, in_len, bitpos + (uint32_t)lane);
                uint32_t b, o, v;
                mz_decode_token(w, L, &b, &o, &v);
                P(tbits) = b;
                P(tolen) = o;
                P(tval) = v;
            }
            /* chain walk from offset 0: which lanes hold real tokens */
            const uint64_t avail = total_bits - bitpos;
            uint32_t pos = 0, eob = 0;
            uint64_t sel = 0;
            int32_t chain_err = MZHIP_OK;
            while (pos < 64) {
                uint32_t nb = MZ_READLANE(tbits, pos);
                if (nb == 0) {
                    uint32_t need = MZ_READLANE(tval, pos);
                    chain_err = ((uint64_t)pos + need > avail) ? MZHIP_BUF_ERROR : MZHIP_DATA_ERROR;
                    break;
                }
                if ((uint64_t)pos + nb > avail) {
                    chain_err = MZHIP_BUF_ERROR;
                    break;
                }
                sel |= 1ull << pos;
                uint32_t ol = MZ_READLANE(tolen, pos);
                pos += nb;
                if (ol == 0) {
                    eob = 1;
                    break;
                }
            }
            bitpos += pos;

            /* output offsets: literals by popcount, matches by a uniform walk */
            uint64_t litm, matm;
            MZ_BALLOT(litm, ((sel >> lane) & 1) && P(tolen) == 1);
            MZ_BALLOT(matm, ((sel >> lane) & 1) && P(tolen) > 1);
            PV(uint32_t, oofs);
            MZ_LANES { P(oofs) = mz_popc64(litm & ((1ull << lane) - 1)); }
            uint32_t total = mz_popc64(litm);
            {
                uint64_t mm = matm;
                while (mm) {
                    uint32_t t = mz_ctz64(mm);
                    mm &= mm - 1;
                    uint32_t ln = MZ_READLANE(tolen, t);
                    MZ_LANES {
                        if ((uint32_t)lane > t) P(oofs) += ln;
                    }
                    total += ln;
                }
            }
            if (total > out_cap - out_pos) {
                status = MZHIP_OUT_FULL;
                goto finish;
            }
            MZ_LANES {
                if ((litm >> lane) & 1) out[out_pos + P(oofs)] = (uint8_t)P(tval);
            }
            MZ_WAVE_SYNC();
            {
                uint64_t mm = matm;
                while (mm) {
                    uint32_t t = mz_ctz64(mm);
                    mm &= mm - 1;
                    uint32_t ln = MZ_READLANE(tolen, t);
                    uint32_t dist = MZ_READLANE(tval, t);
                    uint32_t dst = out_pos + MZ_READLANE(oofs, t);
                    if (dist > dst) {
                        status = MZHIP_DATA_ERROR; /* invalid distance too far back */
                        goto finish;
                    }
                    const uint8_t *src = out + (dst - dist);
                    if (dist >= ln) {
                        MZ_LANES {
                            for (uint32_t i = (uint32_t)lane; i < ln; i += 64) out[dst + i] = src[i];
                        }
                    } else {
                        /* overlapping run: byte i repeats with period dist */
                        MZ_LANES {
                            for (uint32_t i = (uint32_t)lane; i < ln; i += 64) out[dst + i] = src[i % dist];
                        }
                    }
                    MZ_WAVE_SYNC();
                }
            }
            out_pos += total;
            MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, out_pos, crc_tab, tabs->kx);
            if (chain_err != MZHIP_OK) {
                status = chain_err;
                goto finish;
            }
            if (eob) break;
        }
    }

finish:
    res->status = status;
    res->out_len = out_pos;
    res->in_used = (uint32_t)((bitpos + 7) >> 3);
    if (res->in_used > in_len) res->in_used = in_len;
    {
        uint32_t crc;
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, out_pos, crc_tab, tabs->kx);
        MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, out, out_pos, crc_tab, tabs);
        res->crc = crc;
    }
}

#endif
