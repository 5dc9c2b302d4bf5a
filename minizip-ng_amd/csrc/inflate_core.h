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

#ifndef MZ_LROOT
#define MZ_LROOT 11 /* literal/length fast-table index bits */
#endif
#define MZ_DROOT 8  /* distance fast-table index bits        */
#define MZ_CROOT 7  /* code-length-code table bits (== max)  */

/* per-wave LDS scratch */
typedef struct mz_inflate_hdr_scratch { /* live while a block header is parsed */
    uint16_t clc_fast[1 << MZ_CROOT];
    uint16_t clc_sym[20];
    uint16_t clc_first[16], clc_count[16], clc_offs[16];
    uint8_t cl[320];     /* code lengths of the current block (nlen + ndist <= 316; fixed: 288 + 32) */
    uint8_t clc_len[20]; /* lengths of the code-length code */
} mz_inflate_hdr_scratch;

typedef struct mz_inflate_body_scratch { /* live while the block body is decoded */
    uint32_t ring[128]; /* 512 B of compressed stream: aligned dword j of the entry at ring[j & 127] */
    uint8_t mslot[64];  /* lane ids of this step's match tokens, compacted */
} mz_inflate_body_scratch;

typedef struct mz_inflate_lds {
    uint16_t lit_fast[1 << MZ_LROOT]; /* (symbol << 4) | code length, 0 = not a short code */
    uint16_t dist_fast[1 << MZ_DROOT];
    uint16_t lit_sym[288]; /* symbols sorted by (length, value): canonical order */
    uint16_t dist_sym[32];
    uint16_t lit_first[16], lit_count[16], lit_offs[16];
    uint16_t dist_first[16], dist_count[16], dist_offs[16];
    uint16_t rank_base[16];
    uint16_t lit_lim[16];  /* left-justified 15-bit upper bound of the codes of each length */
    int16_t lit_delta[16]; /* lit_offs[L] - lit_first[L] */
    uint16_t dist_lim[16];
    int16_t dist_delta[16];
    uint32_t hist[16];
    union {
        mz_inflate_hdr_scratch h;
        mz_inflate_body_scratch b;
    } u;
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
MZ_DEV uint64_t mz_bits_at(const uint8_t *in, uint32_t in_len, uint32_t bitpos) {
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

/* per-length limits for the long-code (> root bits) search: a 15-bit left-justified stream value v
 * carries a code of length L iff lim[L-1] <= v < lim[L]; the symbol is symtab[delta[L] + (v >> (15-L))]. */
#define MZ_CODE_LIMITS(lim_, delta_, first_, count_, offs_)                                                 \
    do {                                                                                                    \
        MZ_LANES {                                                                                          \
            if (lane >= 1 && lane < 16) {                                                                   \
                (lim_)[lane] = (uint16_t)(((uint32_t)(first_)[lane] + (count_)[lane]) << (15 - lane));      \
                (delta_)[lane] = (int16_t)((int32_t)(offs_)[lane] - (int32_t)(first_)[lane]);               \
            }                                                                                               \
        }                                                                                                   \
        MZ_WAVE_SYNC();                                                                                     \
    } while (0)
#define MZ_LIT_LIMITS(L_) MZ_CODE_LIMITS((L_)->lit_lim, (L_)->lit_delta, (L_)->lit_first, (L_)->lit_count, (L_)->lit_offs)
#define MZ_DIST_LIMITS(L_) \
    MZ_CODE_LIMITS((L_)->dist_lim, (L_)->dist_delta, (L_)->dist_first, (L_)->dist_count, (L_)->dist_offs)

/* branch-free search for a code longer than `root` bits (see MZ_CODE_LIMITS): returns the symbol,
 * *nbits = code length or 0 when no code matches (an unused code of an incomplete set) */
MZ_DEV uint32_t mz_long_code(uint32_t lo, int root, const uint16_t *lim, const int16_t *delta, const uint16_t *symtab,
                             uint32_t nsym, uint32_t *nbits) {
    const uint32_t v15 = mz_brev32(lo) >> 17;
    uint32_t len = (uint32_t)root + 1u;
#pragma unroll
    for (int k = root + 1; k < 15; k++) len += (v15 >= lim[k]) ? 1u : 0u;
    const uint32_t ok = (v15 < lim[15]) ? 1u : 0u;
    const uint32_t idx = (uint32_t)((int32_t)delta[len] + (int32_t)(v15 >> (15u - len)));
    const uint32_t sy = symtab[(ok && idx < nsym) ? idx : 0u];
    *nbits = ok ? len : 0u;
    return sy;
}

/* uniform n-bit read at the block-header level */
#define MZ_HDR_BITS(dst, n)                                                   \
    do {                                                                      \
        if (bitpos + (uint32_t)(n) > total_bits) {                            \
            status = MZHIP_BUF_ERROR;                                         \
            goto finish;                                                      \
        }                                                                     \
        uint64_t _w = mz_bits_at(in, in_len, bitpos);                         \
        (dst) = MZ_UNIFORM((uint32_t)_w & ((1u << (n)) - 1));                 \
        bitpos += (n);                                                        \
    } while (0)

/* Aligned dword j of the compressed stream (dword 0 = the aligned dword holding in[0]); bytes outside
 * [in, in + in_len) read as zero, and no byte outside that range is ever touched except inside an
 * aligned dword that also holds a valid byte. */
MZ_DEV uint32_t mz_load_stream_dword(const uint8_t *in_al, uint32_t in_mis, uint32_t in_len, uint32_t j) {
    const uint64_t lo = (uint64_t)j * 4u, end = (uint64_t)in_mis + in_len;
    if (lo >= end) return 0u;
    uint32_t d = *(const uint32_t *)(in_al + lo);
    if (lo < in_mis) d &= 0xFFFFFFFFu << (8u * (in_mis - (uint32_t)lo)); /* bytes before in[0] */
    if (lo + 4u > end) d &= 0xFFFFFFFFu >> (8u * (uint32_t)(lo + 4u - end)); /* bytes past the end */
    return d;
}

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
    const uint32_t total_bits = in_len * 8u; /* in_len < 2^28, checked below */
    const uint32_t in_mis = (uint32_t)((uintptr_t)in & 3u);
    const uint8_t *in_al = in - in_mis;
    uint32_t bitpos = 0;
    uint32_t out_pos = 0;
    int32_t status = MZHIP_OK;
    uint32_t last = 0;
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    MZ_LANES { P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u; }

    if (in_len >= (1u << 28)) { /* bit cursor is 32-bit: one entry's compressed stream must be < 256 MiB */
        status = MZHIP_UNSUPPORTED;
        goto finish;
    }
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
            bitpos = byte * 8u;
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
            bitpos += n * 8u;
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
                    L->u.h.cl[s] = (uint8_t)(s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : s < 288 ? 8 : 5);
            }
            MZ_WAVE_SYNC();
            MZ_BUILD_HUFF(left, maxlen, L, L->u.h.cl, 288, L->lit_fast, MZ_LROOT, L->lit_sym, L->lit_first, L->lit_count,
                          L->lit_offs);
            MZ_LIT_LIMITS(L);
            MZ_BUILD_HUFF(left, maxlen, L, L->u.h.cl + 288, 32, L->dist_fast, MZ_DROOT, L->dist_sym, L->dist_first,
                          L->dist_count, L->dist_offs);
            MZ_DIST_LIMITS(L);
        } else {
            /* dynamic code, appnote.txt:2060-2106 */
            uint32_t h;
            MZ_HDR_BITS(h, 14);
            uint32_t nlen = (h & 31u) + 257, ndist = ((h >> 5) & 31u) + 1, ncode = (h >> 10) + 4;
            if (nlen > 286 || ndist > 30) {
                status = MZHIP_DATA_ERROR; /* too many length or distance symbols */
                goto finish;
            }
            if (bitpos + 3u * ncode > total_bits) {
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            MZ_LANES {
                if (lane < 19) L->u.h.clc_len[lane] = 0;
            }
            MZ_WAVE_SYNC();
            MZ_LANES {
                if ((uint32_t)lane < ncode) {
                    uint64_t w = mz_bits_at(in, in_len, bitpos + 3u * (uint32_t)lane);
                    L->u.h.clc_len[mz_k_order[lane]] = (uint8_t)((uint32_t)w & 7u);
                }
            }
            bitpos += 3u * ncode;
            MZ_WAVE_SYNC();
            MZ_BUILD_HUFF(left, maxlen, L, L->u.h.clc_len, 19, L->u.h.clc_fast, MZ_CROOT, L->u.h.clc_sym, L->u.h.clc_first, L->u.h.clc_count,
                          L->u.h.clc_offs);
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
                    uint32_t e = MZ_UNIFORM(L->u.h.clc_fast[(uint32_t)(wu >> used) & 127u]);
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
                        MZ_LANES { L->u.h.cl[idx] = (uint8_t)sym; } /* uniform store */
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
                            for (uint32_t k = (uint32_t)lane; k < rep; k += 64) L->u.h.cl[idx + k] = (uint8_t)val;
                        }
                        prev = val;
                        idx += rep;
                    }
                }
                bitpos += used;
            }
            MZ_WAVE_SYNC();
            if (MZ_UNIFORM(L->u.h.cl[256]) == 0) {
                status = MZHIP_DATA_ERROR; /* invalid code -- missing end-of-block */
                goto finish;
            }
            MZ_BUILD_HUFF(left, maxlen, L, L->u.h.cl, nlen, L->lit_fast, MZ_LROOT, L->lit_sym, L->lit_first,
                          L->lit_count, L->lit_offs);
            if (left < 0 || (left > 0 && maxlen != 1)) {
                status = MZHIP_DATA_ERROR; /* invalid literal/lengths set */
                goto finish;
            }
            MZ_LIT_LIMITS(L);
            MZ_BUILD_HUFF(left, maxlen, L, L->u.h.cl + nlen, ndist, L->dist_fast, MZ_DROOT, L->dist_sym, L->dist_first,
                          L->dist_count, L->dist_offs);
            if (left < 0 || (left > 0 && maxlen > 1)) {
                status = MZHIP_DATA_ERROR; /* invalid distances set */
                goto finish;
            }
            MZ_DIST_LIMITS(L);
        }

        /* ---- compressed block body: speculative 64-offset decode ----
         * Compressed bytes are staged through a 512-byte LDS ring (two 256-byte blocks, the next
         * block prefetched into a VGPR one block ahead), so the per-step window fetch is three
         * ds_read_b32 and no global-memory latency sits on the critical path. */
        {
            uint32_t *ring = L->u.b.ring;
            const uint32_t pbase = 8u * in_mis; /* bit offset of `in` inside its aligned dword */
            uint32_t ring_hi;                   /* blocks < ring_hi are in the ring; block ring_hi is in wpre */
            PV(uint32_t, wpre);
            {
                const uint32_t blk = (bitpos + pbase) >> 11; /* 2048 bits per block */
                MZ_LANES {
                    ring[((blk & 1u) << 6) + (uint32_t)lane] = mz_load_stream_dword(in_al, in_mis, in_len, blk * 64u + (uint32_t)lane);
                    ring[(((blk + 1u) & 1u) << 6) + (uint32_t)lane] =
                        mz_load_stream_dword(in_al, in_mis, in_len, (blk + 1u) * 64u + (uint32_t)lane);
                    P(wpre) = mz_load_stream_dword(in_al, in_mis, in_len, (blk + 2u) * 64u + (uint32_t)lane);
                }
                ring_hi = blk + 2u;
                MZ_WAVE_SYNC();
            }

            for (;;) {
                const uint32_t pbit = bitpos + pbase;
                if ((pbit >> 11) + 1u >= ring_hi) {
                    /* the cursor entered the newest block: retire the oldest, start the next fetch */
                    MZ_LANES {
                        ring[((ring_hi & 1u) << 6) + (uint32_t)lane] = P(wpre);
                        P(wpre) = mz_load_stream_dword(in_al, in_mis, in_len, (ring_hi + 1u) * 64u + (uint32_t)lane);
                    }
                    ring_hi++;
                    MZ_WAVE_SYNC();
                }

                /* phase 1: every lane decodes the token that would start at bit cursor + lane */
                PV(uint64_t, win);
                PV(uint32_t, nbl);
                PV(uint32_t, syml);
                MZ_LANES {
                    const uint32_t pl = pbit + (uint32_t)lane;
                    const uint32_t j = pl >> 5, sh = pl & 31u;
                    const uint32_t d0 = ring[j & 127u], d1 = ring[(j + 1u) & 127u], d2 = ring[(j + 2u) & 127u];
                    uint64_t w = ((((uint64_t)d1) << 32) | d0) >> sh;
                    w |= ((uint64_t)d2 << 1) << (63u - sh);
                    const uint32_t e = L->lit_fast[(uint32_t)w & ((1u << MZ_LROOT) - 1)];
                    P(win) = w;
                    P(nbl) = e & 15u;
                    P(syml) = e >> 4;
                }
                uint64_t slow;
                MZ_BALLOT(slow, P(nbl) == 0);
                if (slow) { /* some lane looks at a code longer than the fast table: branch-free limit search */
                    MZ_LANES {
                        uint32_t nb;
                        const uint32_t sy = mz_long_code((uint32_t)P(win), MZ_LROOT, L->lit_lim, L->lit_delta, L->lit_sym, 288u, &nb);
                        if (P(nbl) == 0) {
                            P(syml) = sy;
                            P(nbl) = nb;
                        }
                    }
                }
                PV(uint32_t, lenl);
                PV(uint32_t, nb2l);
                PV(uint32_t, dnl);
                PV(uint32_t, dsyml);
                MZ_LANES {
                    /* length base / extra bits, branch-free (appnote.txt:2107-2120) */
                    const uint32_t s = (P(syml) - 257u) & 31u;
                    const uint32_t ex = (s < 8u || s >= 28u) ? 0u : ((s - 4u) >> 2);
                    uint32_t lbase = (s < 8u) ? (3u + s) : (3u + ((4u + (s & 3u)) << ex));
                    lbase = (s == 28u) ? 258u : lbase;
                    const uint32_t wl = (uint32_t)(P(win) >> P(nbl));
                    P(lenl) = lbase + (wl & ((1u << ex) - 1u));
                    const uint32_t nb2 = P(nbl) + ex;
                    const uint32_t d = L->dist_fast[(uint32_t)(P(win) >> nb2) & ((1u << MZ_DROOT) - 1)];
                    P(nb2l) = nb2;
                    P(dnl) = d & 15u;
                    P(dsyml) = d >> 4;
                }
                MZ_BALLOT(slow, P(syml) > 256u && P(dnl) == 0);
                if (slow) {
                    MZ_LANES {
                        uint32_t dn;
                        const uint32_t sy = mz_long_code((uint32_t)(P(win) >> P(nb2l)), MZ_DROOT, L->dist_lim, L->dist_delta,
                                                         L->dist_sym, 32u, &dn);
                        if (P(dnl) == 0) {
                            P(dsyml) = sy;
                            P(dnl) = dn;
                        }
                    }
                }
                /* packed token: [5:0] bits (0 = invalid), [6] end-of-block, [15:7] bytes produced,
                 * [31:16] literal byte | match distance | (invalid) bits the verdict needed */
                PV(uint32_t, tk);
                MZ_LANES {
                    const uint32_t sym = P(syml), nb = P(nbl);
                    const uint32_t ds = P(dsyml), dn = P(dnl);
                    const uint32_t dex = (ds < 4u) ? 0u : ((ds - 2u) >> 1);
                    const uint32_t dbase = (ds < 4u) ? (1u + ds) : (1u + ((2u + (ds & 1u)) << dex));
                    const uint32_t dlo = (uint32_t)(P(win) >> P(nb2l));
                    const uint32_t dist = dbase + ((dlo >> dn) & ((1u << dex) - 1u));
                    const uint32_t nb2 = P(nb2l);
                    const uint32_t t_match = (nb2 + dn + dex) | (P(lenl) << 7) | (dist << 16);
                    const uint32_t t_badd = ((dn == 0u) ? (nb2 + 15u) : (nb2 + dn)) << 16; /* invalid distance code */
                    const uint32_t t_len = (dn == 0u || ds > 29u) ? t_badd : t_match;
                    const uint32_t t_hi = (sym > 285u) ? (nb << 16) /* 286, 287 */ : t_len;
                    const uint32_t t_lo = (sym < 256u) ? (nb | (1u << 7) | (sym << 16)) : (nb | 64u);
                    uint32_t t = (sym <= 256u) ? t_lo : t_hi;
                    t = (nb == 0u) ? (15u << 16) /* invalid literal/length code: verdict needed 15 bits */ : t;
                    P(tk) = t;
                }

                /* phase 2: which candidates are real tokens?  Token i of this step starts at f^i(0), where
                 * f(l) = l + bits(l).  Instead of hopping along that chain on the scalar unit, f is squared three
                 * times with cross-lane gathers (f^2, f^4, f^8) and lane i (i < 16) composes f^i(0) from the binary
                 * digits of i: seven ds_bpermute rounds, no scalar work, and the step's tokens come out COMPACTED
                 * (lane i holds token i).  A value >= 64 is terminal: plain = bit offset where the next step
                 * starts, |0x100 = end-of-block seen, |0x200 = invalid code at that offset.  At most 15 tokens are
                 * retired per step; lane 15 only supplies the continuation offset. */
                const uint32_t avail = total_bits - bitpos;
                uint32_t pos, eob = 0, ntok;
                int32_t chain_err = MZHIP_OK;
                PV(uint32_t, cpos);
                if (avail >= 64u + 48u) {
                    PV(uint32_t, g1);
                    PV(uint32_t, g2);
                    PV(uint32_t, g4);
                    PV(uint32_t, g8);
                    PV(uint32_t, gt);
                    MZ_LANES {
                        const uint32_t t = P(tk), nb = t & 63u, nx = (uint32_t)lane + nb;
                        P(g1) = (nb == 0u) ? (0x200u | (uint32_t)lane) : ((t & 64u) ? (0x100u | nx) : nx);
                    }
                    /* squaring f and composing f^i(0) are interleaved so that the two gathers of a round are
                     * independent: 5 dependent rounds instead of 7 */
                    PV(uint32_t, ct);
                    MZ_LANES { P(cpos) = 0u; }
                    MZ_GATHER(gt, g1, P(g1));
                    MZ_GATHER(ct, g1, P(cpos));
                    MZ_LANES {
                        P(g2) = (P(g1) < 64u) ? P(gt) : P(g1);
                        P(cpos) = ((uint32_t)lane & 1u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER(gt, g2, P(g2));
                    MZ_GATHER(ct, g2, P(cpos));
                    MZ_LANES {
                        P(g4) = (P(g2) < 64u) ? P(gt) : P(g2);
                        P(cpos) = (((uint32_t)lane & 2u) && P(cpos) < 64u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER(gt, g4, P(g4));
                    MZ_GATHER(ct, g4, P(cpos));
                    MZ_LANES {
                        P(g8) = (P(g4) < 64u) ? P(gt) : P(g4);
                        P(cpos) = (((uint32_t)lane & 4u) && P(cpos) < 64u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER(ct, g8, P(cpos));
                    MZ_LANES { P(cpos) = (((uint32_t)lane & 8u) && P(cpos) < 64u) ? P(ct) : P(cpos); }
                    uint64_t live;
                    MZ_BALLOT(live, lane < 15 && P(cpos) < 64u);
                    ntok = mz_popc64(live); /* tokens are lanes 0..ntok-1 (the chain never resumes once terminal) */
                    const uint32_t term = MZ_READLANE(cpos, ntok);
                    pos = term & 0xFFu;
                    eob = (term >> 8) & 1u;
                    if (term & 0x200u) chain_err = MZHIP_DATA_ERROR;
                } else {
                    /* within 14 bytes of the end of input: serial walk that also polices every token's extent */
                    uint64_t sel = 0;
                    pos = 0;
                    ntok = 0;
                    while (pos < 64u && ntok < 15u) {
                        const uint32_t t = MZ_READLANE(tk, pos);
                        const uint32_t nb = t & 63u;
                        if (nb == 0u) {
                            chain_err = (pos + (t >> 16) > avail) ? MZHIP_BUF_ERROR : MZHIP_DATA_ERROR;
                            break;
                        }
                        if (pos + nb > avail) {
                            chain_err = MZHIP_BUF_ERROR;
                            break;
                        }
                        sel |= 1ull << pos;
                        ntok++;
                        pos += nb;
                        if (t & 64u) {
                            eob = 1;
                            break;
                        }
                    }
                    MZ_LANES {
                        if ((sel >> lane) & 1u) L->u.b.mslot[mz_popc64(sel & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                    }
                    MZ_WAVE_SYNC();
                    MZ_LANES { P(cpos) = ((uint32_t)lane < ntok) ? (uint32_t)L->u.b.mslot[lane & 15] : 0x400u; }
                    MZ_WAVE_SYNC();
                }
                bitpos += pos;

                /* phase 3: compacted tokens -> output offsets (prefix sum inside the first DPP row);
                 * literals scatter in one store */
                PV(uint32_t, tkc);
                PV(uint32_t, olen);
                PV(uint32_t, oend);
                MZ_GATHER(tkc, tk, P(cpos));
                MZ_LANES {
                    if ((uint32_t)lane >= ntok) P(tkc) = 0u;
                    P(olen) = (P(tkc) >> 7) & 511u;
                }
                MZ_INCL_SCAN(oend, olen);
                const uint32_t total = MZ_READLANE(oend, 15);
                if (total > out_cap - out_pos) {
                    status = MZHIP_OUT_FULL;
                    goto finish;
                }
                uint64_t matm;
                MZ_BALLOT(matm, P(olen) > 1u);
                MZ_LANES {
                    if (P(olen) == 1u) out[out_pos + P(oend) - 1u] = (uint8_t)(P(tkc) >> 16);
                }
                MZ_WAVE_SYNC();

                /* phase 4: LZ77 back-references.  Four matches at a time, 16 lanes each: one gather of the
                 * match descriptors, one load, one store, while every source lies before this step's output.
                 * Anything that reads bytes produced in this same step (or overlaps itself) takes the
                 * in-order cooperative path below. */
                if (matm) {
                    const uint32_t nmatch = mz_popc64(matm);
                    uint32_t done_m = 0;
                    MZ_LANES {
                        if (P(olen) > 1u) L->u.b.mslot[mz_popc64(matm & ((1ull << lane) - 1ull))] = (uint8_t)lane;
                    }
                    MZ_WAVE_SYNC();
                    while (done_m < nmatch) {
                        PV(uint32_t, msrc);
                        PV(uint32_t, mtk);
                        PV(uint32_t, mend);
                        MZ_LANES {
                            const uint32_t g = done_m + ((uint32_t)lane >> 4);
                            P(msrc) = (g < nmatch) ? (uint32_t)L->u.b.mslot[g] : 64u;
                        }
                        MZ_GATHER(mtk, tkc, P(msrc));
                        MZ_GATHER(mend, oend, P(msrc));
                        uint64_t dep;
                        MZ_LANES {
                            if (P(msrc) >= 64u) { P(mtk) = 0; P(mend) = 0; }
                        }
                        /* independent iff the source ends at or before this step's first output byte
                         * (out_pos + mend - dist <= out_pos, which also rules out self-overlap) and the distance
                         * stays inside the entry; everything else takes the in-order path below */
                        MZ_BALLOT(dep, P(mend) > (P(mtk) >> 16) ||
                                           (P(mtk) >> 16) > out_pos + P(mend) - ((P(mtk) >> 7) & 511u));
                        if (dep) { break; }
                        MZ_LANES {
                            if (P(msrc) < 64u) {
                                const uint32_t ln = (P(mtk) >> 7) & 511u, dist = P(mtk) >> 16;
                                const uint32_t dst = out_pos + P(mend) - ln;
                                for (uint32_t i = (uint32_t)lane & 15u; i < ln; i += 16u) out[dst + i] = out[dst - dist + i];
                            }
                        }
                        MZ_WAVE_SYNC();
                        done_m += 4u;
                    }
                    /* in-order cooperative path for what is left (64 bytes per instruction) */
                    while (done_m < nmatch) {
                        const uint32_t tl = MZ_UNIFORM(L->u.b.mslot[done_m]);
                        const uint32_t t = MZ_READLANE(tkc, tl);
                        const uint32_t ln = (t >> 7) & 511u, dist = t >> 16;
                        const uint32_t dst = out_pos + MZ_READLANE(oend, tl) - ln;
                        if (dist > dst) {
                            status = MZHIP_DATA_ERROR; /* invalid distance too far back */
                            goto finish;
                        }
                        const uint8_t *src = out + (dst - dist);
                        if (dist >= ln) {
                            MZ_LANES {
                                for (uint32_t i = (uint32_t)lane; i < ln; i += 64u) out[dst + i] = src[i];
                            }
                        } else { /* overlapping run: byte i repeats with period dist */
                            MZ_LANES {
                                for (uint32_t i = (uint32_t)lane; i < ln; i += 64u) out[dst + i] = src[i % dist];
                            }
                        }
                        MZ_WAVE_SYNC();
                        done_m++;
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
    }

finish:
    res->status = status;
    res->out_len = out_pos;
    res->in_used = (bitpos + 7u) >> 3;
    if (res->in_used > in_len) res->in_used = in_len;
    {
        uint32_t crc;
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, out_pos, crc_tab, tabs->kx);
        MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, out, out_pos, crc_tab, tabs);
        res->crc = crc;
    }
}

#endif
