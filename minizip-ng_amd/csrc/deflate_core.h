/* deflate_core.h -- raw-DEFLATE ENCODE of ONE piece by ONE wavefront, fixed-Huffman blocks, CRC-32 of
 * the input fused in (kernel K4 of SURVEY 2.1).
 *
 * Replaces what the reference does per entry through mz_stream_zlib_write / _close
 * (mz_strm_zlib.c:203-264,280-305 -> zlib deflate(), raw, level 1) followed by
 * mz_crypt_crc32_update (mz_zip.c:2064).  Format: doc/zip/appnote.txt:2030-2166, fixed code :2050-2059.
 * The compressed bytes are NOT zlib's (compressor output is not a format property; zlib 1.2.11 and
 * zlib-ng already differ): parity for this kernel is "the reference's inflate of these bytes returns the
 * input, and the CRC matches" (hard parts).
 *
 * MI355X mapping: 64 input positions per step, one per lane.
 *   - match finding: each lane hashes its 4 bytes, looks the hash up in a per-wave LDS table of recent
 *     positions (read-then-update, one table access per lane per step) and measures the match in place
 *     with dword compares (minimum match 4, maximum 258, window 32 KiB);
 *   - greedy token selection is the same successor-chain problem as in the decoder (f(l) = l + len(l) or
 *     l + 1): f is squared five times with cross-lane gathers and lane r composes f^r(start), so the
 *     step's tokens land compacted in lanes 0..n-1 with no serial walk;
 *   - each token's fixed-Huffman bits (<= 31) are computed arithmetically (length / distance symbols by
 *     leading-zero count), bit offsets come from a DPP prefix sum, and lanes OR their bits into a small
 *     LDS staging area that is flushed to HBM as whole bytes.
 */
#ifndef MZHIP_DEFLATE_CORE_H
#define MZHIP_DEFLATE_CORE_H

#include "crc32_core.h"
#include "wave.h"

#define MZ_DEF_HBITS 12
#define MZ_DEF_MINMATCH 4u
#define MZ_DEF_MAXMATCH 258u

typedef struct mz_deflate_lds {
    uint16_t head[1 << MZ_DEF_HBITS]; /* low 16 bits of the most recent position with this hash */
    uint32_t stage[72];               /* this step's bits, OR-ed in by the lanes (<= 64 x 31 bits + carry) */
} mz_deflate_lds;

typedef struct mz_deflate_result {
    int32_t status;
    uint32_t out_len;
    uint32_t crc; /* CRC-32 of the INPUT bytes (what mz_zip_entry_write accumulates, mz_zip.c:2064) */
} mz_deflate_result;

MZ_DEV uint32_t mz_clz32(uint32_t v) {
#if defined(MZHIP_HOST_EMUL)
    return v ? (uint32_t)__builtin_clz(v) : 32u;
#else
    return (uint32_t)__clz((int)v);
#endif
}

/* fixed-Huffman code of a literal/length symbol, already bit-reversed for LSB-first packing
 * (appnote.txt:2050-2059); *nbits = its length */
MZ_DEV uint32_t mz_fixed_litlen(uint32_t sym, uint32_t *nbits) {
    uint32_t code, n;
    if (sym < 144u) {
        code = 0x30u + sym;
        n = 8;
    } else if (sym < 256u) {
        code = 0x190u + (sym - 144u);
        n = 9;
    } else if (sym < 280u) {
        code = sym - 256u;
        n = 7;
    } else {
        code = 0xC0u + (sym - 280u);
        n = 8;
    }
    *nbits = n;
    return mz_brev32(code) >> (32u - n);
}

/* all bits of one token: literal, or length + distance with their extra bits (appnote.txt:2107-2133) */
MZ_DEV uint32_t mz_token_bits(uint32_t mlen, uint32_t val, uint32_t *nbits) {
    uint32_t n;
    if (mlen == 0u) return mz_fixed_litlen(val, nbits);
    /* length symbol */
    uint32_t lsym, lex = 0, lxv = 0;
    const uint32_t l = mlen - 3u;
    if (mlen == 258u) {
        lsym = 285u;
    } else if (l < 8u) {
        lsym = 257u + l;
    } else {
        const uint32_t k = 31u - mz_clz32(l);
        lex = k - 2u;
        lsym = 257u + 4u * lex + 4u + ((l >> lex) & 3u);
        lxv = l & ((1u << lex) - 1u);
    }
    uint32_t bits = mz_fixed_litlen(lsym, &n);
    bits |= lxv << n;
    n += lex;
    /* distance symbol: 5-bit fixed code */
    const uint32_t d = val - 1u;
    uint32_t dsym, dex = 0, dxv = 0;
    if (d < 4u) {
        dsym = d;
    } else {
        const uint32_t k = 31u - mz_clz32(d);
        dex = k - 1u;
        dsym = 2u * k + ((d >> (k - 1u)) & 1u);
        dxv = d & ((1u << dex) - 1u);
    }
    bits |= (mz_brev32(dsym) >> 27) << n;
    n += 5u;
    bits |= dxv << n;
    n += dex;
    *nbits = n; /* <= 8 + 5 + 5 + 13 = 31 */
    return bits;
}

/* Encode in[0..in_len) as one raw-DEFLATE piece.  final != 0: a complete stream (BFINAL block, padded to a
 * byte).  final == 0: a non-final block followed by an empty stored block, so that pieces of one stream can
 * be concatenated on byte boundaries.  All arguments wave-uniform. */
MZ_DEV void mz_deflate_piece(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t final,
                             mz_deflate_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs,
                             mz_deflate_result *res) {
    MZ_LANE_DECL
    int32_t status = MZHIP_OK;
    uint32_t obyte = 0;                            /* whole bytes already written to out */
    uint32_t carry = final ? 3u : 2u, cbits = 3u;  /* pending bits: BFINAL, BTYPE = 01 (fixed) */
    uint32_t skip = 0;                             /* positions at the next step's start covered by a match */
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    MZ_LANES {
        P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u;
        for (uint32_t i = (uint32_t)lane; i < (1u << MZ_DEF_HBITS) / 2u; i += 64u) ((uint32_t *)L->head)[i] = 0u;
    }
    MZ_WAVE_SYNC();

    for (uint32_t p = 0; p < in_len; p += 64u) {
        const uint32_t nv = (in_len - p < 64u) ? (in_len - p) : 64u; /* valid positions in this step */
        /* ---- match candidates ---- */
        PV(uint32_t, hh);
        PV(uint32_t, cand);
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            const uint32_t have4 = (pos + 4u <= in_len) ? 1u : 0u;
            const uint32_t v = have4 ? mz_load_u32(in + pos) : 0u;
            const uint32_t h = (v * 2654435761u) >> (32 - MZ_DEF_HBITS);
            P(hh) = have4 ? h : 0xFFFFFFFFu;
            P(cand) = have4 ? (uint32_t)L->head[h] : 0u;
        }
        MZ_WAVE_SYNC();
        MZ_LANES {
            if (P(hh) != 0xFFFFFFFFu) L->head[P(hh)] = (uint16_t)(p + (uint32_t)lane);
        }
        MZ_WAVE_SYNC();
        PV(uint32_t, pk); /* [8:0] match length (0 = literal), [24:9] distance | literal byte */
        PV(uint32_t, g1); /* 4 * successor lane; bit 12 set: terminal */
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            uint32_t mlen = 0, dist = 0;
            if ((uint32_t)lane < nv) {
                const uint32_t d = (pos - P(cand)) & 0xFFFFu;
                if (P(hh) != 0xFFFFFFFFu && d >= 1u && d <= 32768u && d <= pos) {
                    const uint8_t *a = in + pos, *b = in + (pos - d);
                    const uint32_t maxl = (in_len - pos < MZ_DEF_MAXMATCH) ? (in_len - pos) : MZ_DEF_MAXMATCH;
                    uint32_t l = 0;
                    while (l + 4u <= maxl && mz_load_u32(a + l) == mz_load_u32(b + l)) l += 4u;
                    while (l < maxl && a[l] == b[l]) l++;
                    if (l >= MZ_DEF_MINMATCH) {
                        mlen = l;
                        dist = d;
                    }
                }
            }
            P(pk) = mlen | ((mlen ? dist : (uint32_t)in[pos < in_len ? pos : 0u]) << 9);
            const uint32_t nx = (uint32_t)lane + (mlen ? mlen : 1u);
            P(g1) = ((uint32_t)lane >= nv || nx >= nv) ? (0x1000u | (4u * nx)) : (4u * nx);
        }
        /* ---- greedy selection: lane r <- r-th element of the chain start, f(start), f(f(start)), ... ---- */
        PV(uint32_t, g2);
        PV(uint32_t, g4);
        PV(uint32_t, g8);
        PV(uint32_t, g16);
        PV(uint32_t, g32);
        PV(uint32_t, gt);
        PV(uint32_t, ct);
        PV(uint32_t, cpos);
        MZ_LANES { P(cpos) = (skip >= nv) ? (0x1000u | (4u * skip)) : (4u * skip); }
#define MZ_DEF_ROUND(gin, gout, bit)                                                         \
    MZ_GATHER4(gt, gin, P(gin));                                                             \
    MZ_GATHER4(ct, gin, P(cpos));                                                            \
    MZ_LANES {                                                                               \
        P(gout) = (P(gin) & 0x1000u) ? P(gin) : P(gt);                                       \
        P(cpos) = (((uint32_t)lane & (bit)) && !(P(cpos) & 0x1000u)) ? P(ct) : P(cpos);      \
    }
        MZ_DEF_ROUND(g1, g2, 1u)
        MZ_DEF_ROUND(g2, g4, 2u)
        MZ_DEF_ROUND(g4, g8, 4u)
        MZ_DEF_ROUND(g8, g16, 8u)
        MZ_DEF_ROUND(g16, g32, 16u)
        MZ_GATHER4(ct, g32, P(cpos));
        MZ_LANES { P(cpos) = (((uint32_t)lane & 32u) && !(P(cpos) & 0x1000u)) ? P(ct) : P(cpos); }
#undef MZ_DEF_ROUND
        uint64_t live;
        MZ_BALLOT(live, !(P(cpos) & 0x1000u));
        const uint32_t ntok = mz_popc64(live); /* tokens sit in lanes 0..ntok-1 */
        /* ---- emit ---- */
        PV(uint32_t, tpk);
        PV(uint32_t, tbits);
        PV(uint32_t, tn);
        PV(uint32_t, tend);
        MZ_GATHER4(tpk, pk, P(cpos));
        MZ_LANES {
            uint32_t n = 0, b = 0;
            if ((uint32_t)lane < ntok) b = mz_token_bits(P(tpk) & 511u, P(tpk) >> 9, &n);
            P(tbits) = b;
            P(tn) = n;
        }
        MZ_INCL_SCAN(tend, tn);
        const uint32_t total = MZ_READLANE(tend, 63);
        if (ntok) { /* where the chain left this step: position of the last token + its length */
            const uint32_t lastpos = MZ_READLANE(cpos, ntok - 1u) >> 2;
            const uint32_t lastpk = MZ_READLANE(tpk, ntok - 1u);
            const uint32_t nx = lastpos + ((lastpk & 511u) ? (lastpk & 511u) : 1u);
            skip = nx > 64u ? nx - 64u : 0u;
        } else {
            skip = skip > 64u ? skip - 64u : 0u;
        }
        MZ_LANES {
            for (uint32_t i = (uint32_t)lane; i < 72u; i += 64u) L->stage[i] = (i == 0u) ? carry : 0u;
        }
        MZ_WAVE_SYNC();
        MZ_LANES {
            if (P(tn)) {
                const uint32_t off = cbits + P(tend) - P(tn);
                const uint32_t w = off >> 5, sh = off & 31u;
                MZ_LDS_ATOMIC_OR(&L->stage[w], P(tbits) << sh);
                if (sh + P(tn) > 32u) MZ_LDS_ATOMIC_OR(&L->stage[w + 1u], P(tbits) >> (32u - sh));
            }
        }
        MZ_WAVE_SYNC();
        {
            const uint32_t nbit = cbits + total, nbytes = nbit >> 3;
            if (nbytes > out_cap - obyte) {
                status = MZHIP_OUT_FULL;
                goto finish;
            }
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < nbytes; i += 64u)
                    out[obyte + i] = (uint8_t)(L->stage[i >> 2] >> (8u * (i & 3u)));
            }
            carry = (nbit & 7u) ? ((MZ_UNIFORM(L->stage[nbytes >> 2]) >> (8u * (nbytes & 3u))) & ((1u << (nbit & 7u)) - 1u)) : 0u;
            cbits = nbit & 7u;
            obyte += nbytes;
        }
        MZ_WAVE_SYNC();
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, in, (p + nv), crc_tab, tabs->kx);
    }

    /* end-of-block (7 zero bits), then either pad the final block or append an empty stored block */
    {
        uint64_t tail = carry; /* EOB adds 7 zero bits */
        uint32_t nbit = cbits + 7u;
        if (!final) {
            /* BFINAL = 0, BTYPE = 00, pad to a byte, LEN = 0x0000, NLEN = 0xFFFF (appnote.txt:2045-2049) */
            nbit += 3u;
            nbit = (nbit + 7u) & ~7u;
            tail |= (uint64_t)0xFFFF0000u << nbit;
            nbit += 32u;
        }
        const uint32_t nbytes = (nbit + 7u) >> 3; /* <= 2 (+5) bytes */
        if (nbytes > out_cap - obyte) {
            status = MZHIP_OUT_FULL;
            goto finish;
        }
        MZ_LANES {
            if ((uint32_t)lane < nbytes) out[obyte + (uint32_t)lane] = (uint8_t)(tail >> (8u * (uint32_t)lane));
        }
        MZ_WAVE_SYNC();
        obyte += nbytes;
    }

finish:
    res->status = status;
    res->out_len = obyte;
    {
        uint32_t crc;
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, in, in_len, crc_tab, tabs->kx);
        MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, in, in_len, crc_tab, tabs);
        res->crc = crc;
    }
}

#endif
