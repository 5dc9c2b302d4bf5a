/* lzma_enc_core.h -- LZMA1 ENCODE (ZIP method 14 entries and the LZMA2 chunks of method 95), two kernels.
 *
 * Replaces what the reference does through mz_stream_lzma_write / _close (mz_strm_lzma.c:244-332 -> liblzma
 * lzma_alone_encoder / lzma_stream_encoder, preset 6).  The bytes are not liblzma's (compressor output is not a
 * format property): parity is "liblzma -- the reference's mz_stream_lzma_read -- decodes them back to the input".
 * Format: the LZMA specification (same model as lzma_core.h: 11-bit adaptive probabilities, 12 states, rep0-3, two
 * length coders, 6-bit distance slots, aligned bits) and doc/zip/appnote.txt:2232-2275 for the ZIP framing.
 *
 * Split along what is parallel and what is not:
 *   mz_lz_tokenize   one wave per 64 KiB block, 64 positions per step: the LZ77 parse of deflate_core.h (hash head
 *                    table in LDS, in-place match measurement, lazy rule, greedy selection by pointer doubling);
 *                    the previous 32 KiB of the stream are hashed first so matches may reach back across the block
 *                    boundary; tokens (literal | length 4..258 + distance <= 32 KiB) go to HBM;
 *   mz_lzma_rc_encode one wave per stream: the adaptive range coder is strictly serial, so the token sequence is
 *                    coded by wave-uniform code with the probability model in LDS; 64 tokens at a time are
 *                    prepared in parallel (positions by prefix sum, the literal context byte and the match byte
 *                    fetched from the input by the lanes) so that the serial loop never waits for HBM; output
 *                    bytes are collected in a 256-byte VGPR window and stored coalesced.
 * Properties written: lc 3, lp 0, pb 2, dictionary 64 KiB (every distance is below 32 KiB).
 */
#ifndef MZHIP_LZMA_ENC_CORE_H
#define MZHIP_LZMA_ENC_CORE_H

#include "deflate_core.h"
#include "lzma_core.h"

#define MZ_LZE_LC 3u
#define MZ_LZE_PB 2u
#define MZ_LZE_PROPS 0x5Du /* (pb * 5 + lp) * 9 + lc */
#define MZ_LZE_DICT 0x10000u /* a stream of one block: every distance is below 64 KiB */
/* Streams of more than one block look back over MZ_LZE_FAR_DICT bytes (liblzma's preset 6, mz_strm_lzma.c:81, has 8 MiB):
 * the chain pass below leaves every position its nearest earlier occurrence, the block parse tries it beside the
 * candidates of its own hash table.  A token's distance field has 23 bits. */
#ifndef MZ_LZE_FAR_DICT
#define MZ_LZE_FAR_DICT 0x800000u
#endif
#define MZ_LZE_FAR_MAXDIST (MZ_LZE_FAR_DICT - 1u)
#ifndef MZ_LZE_FAR_HBITS
#define MZ_LZE_FAR_HBITS 18u /* the chain pass's table: 1 MiB per resident wave.  One size for every stream: the links of a
                                stream written in segments are then the links of the same bytes coded in one piece */
#endif
#define MZ_LZE_MAXMATCH 273u /* the longest match LZMA codes (2 + 16 + 255) */
#define MZ_LZE_CHAIN_WAVES 1024u /* resident waves of the chain pass (1 MiB of table each) */
#ifndef MZ_LZE_FAR_MINLEN
#define MZ_LZE_FAR_MINLEN 5u /* a far match shorter than this costs more than the literals it replaces */
#endif

#ifndef MZ_LZE_FAR_GAIN
#define MZ_LZE_FAR_GAIN 0u /* bytes a far match must be longer than a near one that is already there */
#endif
#ifndef MZ_LZE_FAR_NGRAM
#define MZ_LZE_FAR_NGRAM 7u /* bytes the chain pass hashes: the nearest earlier occurrence of that many bytes */
#endif
/* links of the chain the block parse follows from a position, by liblzma preset (mz_strm_lzma.c:81 hands COMPRESS_LEVEL to
 * lzma_lzma_preset, whose presets search deeper as they go up): config-4 entries 0.276 at 4 links, 0.269 at 8, 0.265 at 16
 * (liblzma's own fast mode over a hash chain: 0.338 / 0.301 / 0.276 at depth 4 / 16 / 64; its preset 6 -- a binary tree and
 * a priced parse -- 0.245) */
#define MZ_LZE_DEPTH_FOR_PRESET(p) (((p) >= 0 && (p) <= 3) ? 1u : ((p) == 4 || (p) == 5) ? 4u : ((p) >= 7) ? 16u : 8u)
typedef struct mz_lz_tok_lds {
    uint16_t head[1 << MZ_DEF_HBITS];
} mz_lz_tok_lds;

#if defined(MZHIP_HOST_EMUL)
#define MZ_GLB_ATOMIC_MAX(ptr, v) (*(ptr) = (*(ptr) > (v)) ? *(ptr) : (v))
#define MZ_GLB_LOAD_DEV(ptr) (*(ptr))
#define MZ_VM_DRAIN() ((void)0)
#else
#define MZ_GLB_ATOMIC_MAX(ptr, v) atomicMax((ptr), (v))
/* read where the device-scope atomics of earlier steps landed (L2), not a line the vector L1 may still hold */
#define MZ_GLB_LOAD_DEV(ptr) __hip_atomic_load((ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
/* every vector-memory operation of this wave has completed (loads returned) before the next one is issued */
#define MZ_VM_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

/* The chain pass: ONE wave walks a whole stream, 64 positions per step, and leaves in links[p] the position + 1 of the
 * nearest position of an earlier step whose MZ_LZE_FAR_NGRAM bytes hash like those at p (0 = none) -- a hash chain over the
 * whole stream, which is what lets the block parse (one wave per 64 KiB block, in any order) reach back further than
 * the history it can hash itself: it follows links[p], links[links[p] - 1], ... far_depth deep.  head[]:
 * 1 << MZ_LZE_FAR_HBITS words of global scratch.  Occurrences inside the same step of 64 positions are the block
 * parse's own business (its table sees them).  Deterministic: the table takes positions by atomic maximum, and is read
 * where the atomics land. */
MZ_DEV void mz_lz_chain(const uint8_t *in, uint32_t in_len, uint32_t *links, uint32_t *head) {
    MZ_LANE_DECL
    const uint32_t hb = MZ_LZE_FAR_HBITS;
    MZ_LANES {
        for (uint32_t i = (uint32_t)lane; i < (1u << hb); i += 64u) head[i] = 0u;
    }
    MZ_WAVE_SYNC();
    MZ_VM_DRAIN();
    for (uint32_t p = 0; p < in_len; p += 64u) {
        PV(uint32_t, hh);
        PV(uint32_t, old);
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            /* (the last seven positions of a 64 KiB block stay out: a stream written in segments of whole blocks then has
             * the links of the same stream coded in one piece) */
            const uint32_t bend = (pos | (MZ_DEF_BLOCK - 1u)) + 1u; /* (0 behind the last block of a 4 GiB stream: the difference below is right anyway) */
            const uint32_t ok = (pos < in_len && in_len - pos >= 8u && bend - pos >= 8u) ? 1u : 0u;
            uint32_t h = 0;
            if (ok) {
                const uint64_t v = (uint64_t)mz_load_u32(in + pos) | ((uint64_t)mz_load_u32(in + pos + 4u) << 32);
                h = (uint32_t)(((v << (64u - 8u * MZ_LZE_FAR_NGRAM)) * 0x9E3779B185EBCA87ull) >> (64u - hb));
            }
            P(hh) = ok ? h : 0xFFFFFFFFu;
            P(old) = ok ? MZ_GLB_LOAD_DEV(head + h) : 0u;
        }
        MZ_WAVE_SYNC();
        MZ_VM_DRAIN(); /* the table as the steps before left it: this step's positions go in behind the reads */
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            if (pos < in_len) links[pos] = P(old);
            if (P(hh) != 0xFFFFFFFFu) MZ_GLB_ATOMIC_MAX(head + P(hh), pos + 1u);
        }
        MZ_WAVE_SYNC();
    }
}

/* LZ77 parse of in[blk, blk_end) (blk_end - blk <= MZ_DEF_BLOCK) of the stream in[0..); matches may start up to 32 KiB
 * before blk -- and, with links (the chain pass's array for the whole stream, or null), wherever those point.  Returns the
 * number of tokens written to tok[]: [8:0] match length (0 = literal), [31:9] distance | literal byte.
 * ways / xhead: as in K4 (deflate_core.h) -- the `ways` most recent positions of every hash bucket are tried and the
 * longest match wins; 1 = the fast class (presets 0-3), MZ_DEF_WAYS_BEST = presets 4-9 and the default
 * (mz_strm_lzma.c:81 hands the level to lzma_lzma_preset).  xhead: (ways - 1) more tables behind the wave's LDS.
 * far_depth: links of the chain followed per position (MZ_LZE_DEPTH_FOR_PRESET). */
MZ_DEV uint32_t mz_lz_tokenize(const uint8_t *in, uint32_t blk, uint32_t blk_end, uint32_t *tok, mz_lz_tok_lds *L,
                               uint32_t ways, uint16_t *xhead, const uint32_t *links, uint32_t far_depth) {
    MZ_LANE_DECL
    MZ_LANES {
        for (uint32_t i = (uint32_t)lane; i < (1u << MZ_DEF_HBITS) / 2u; i += 64u) ((uint32_t *)L->head)[i] = 0u;
        for (uint32_t i = (uint32_t)lane; i < (ways - 1u) * ((1u << MZ_DEF_HBITS) / 2u); i += 64u) ((uint32_t *)xhead)[i] = 0u;
    }
    MZ_WAVE_SYNC();
    /* the stream's previous 32 KiB are legal match sources (the dictionary is 64 KiB): enter them into the hash table
     * first, without producing tokens, so that a block does not start with an empty history */
    const uint32_t hist = blk > 32768u ? blk - 32768u : 0u;
    for (uint32_t p = hist; p < blk; p += 64u) {
        PV(uint32_t, hh0);
        PV(uint32_t, c0);
        PV2(uint32_t, cx0, MZ_DEF_WAYS_BEST - 1u);
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            const uint32_t ok = (pos < blk && pos + 4u <= blk_end) ? 1u : 0u;
            const uint32_t h = ok ? (mz_load_u32(in + pos) * 2654435761u) >> (32 - MZ_DEF_HBITS) : 0u;
            P(hh0) = ok ? h : 0xFFFFFFFFu;
            P(c0) = (ok && ways > 1u) ? (uint32_t)L->head[h] : 0u;
            for (uint32_t w = 1; w < MZ_DEF_WAYS_BEST; w++)
                P(cx0)[w - 1u] = (ok && w < ways) ? (uint32_t)xhead[((w - 1u) << MZ_DEF_HBITS) | h] : 0u;
        }
        MZ_WAVE_SYNC();
        MZ_LANES {
            if (P(hh0) != 0xFFFFFFFFu) {
                for (uint32_t w = MZ_DEF_WAYS_BEST - 1u; w >= 1u; w--)
                    if (w < ways) xhead[((w - 1u) << MZ_DEF_HBITS) | P(hh0)] = (uint16_t)(w == 1u ? P(c0) : P(cx0)[w - 2u]);
                L->head[P(hh0)] = (uint16_t)(p + (uint32_t)lane);
            }
        }
        MZ_WAVE_SYNC();
    }
    uint32_t ntokens = 0, skip = 0;
    PV(uint32_t, vnx); /* the four bytes at this lane's position of the NEXT step, fetched one step ahead */
    MZ_LANES {
        const uint32_t pos = blk + (uint32_t)lane;
        P(vnx) = (pos + 4u <= blk_end) ? mz_load_u32(in + pos) : 0u;
    }
    for (uint32_t p = blk; p < blk_end; p += 64u) {
        const uint32_t nv = (blk_end - p < 64u) ? (blk_end - p) : 64u;
        PV(uint32_t, hh);
        PV(uint32_t, cand);
        PV2(uint32_t, candx, MZ_DEF_WAYS_BEST - 1u);
        PV(uint32_t, fcand); /* the chain pass's link for this position (position + 1, 0 = none) */
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            const uint32_t have4 = (pos + 4u <= blk_end) ? 1u : 0u;
            P(fcand) = (links && have4) ? links[pos] : 0u;
            const uint32_t v = P(vnx);
            P(vnx) = (pos + 68u <= blk_end) ? mz_load_u32(in + pos + 64u) : 0u;
            const uint32_t h = (v * 2654435761u) >> (32 - MZ_DEF_HBITS);
            P(hh) = have4 ? h : 0xFFFFFFFFu;
            P(cand) = have4 ? (uint32_t)L->head[h] : 0u;
            for (uint32_t w = 1; w < MZ_DEF_WAYS_BEST; w++)
                P(candx)[w - 1u] = (have4 && w < ways) ? (uint32_t)xhead[((w - 1u) << MZ_DEF_HBITS) | h] : 0u;
        }
        MZ_WAVE_SYNC();
        MZ_LANES {
            if (P(hh) != 0xFFFFFFFFu) {
                for (uint32_t w = MZ_DEF_WAYS_BEST - 1u; w >= 1u; w--)
                    if (w < ways) xhead[((w - 1u) << MZ_DEF_HBITS) | P(hh)] = (uint16_t)(w == 1u ? P(cand) : P(candx)[w - 2u]);
                L->head[P(hh)] = (uint16_t)(p + (uint32_t)lane);
            }
        }
        MZ_WAVE_SYNC();
        PV(uint32_t, pk);
        PV(uint32_t, lit);
        PV(uint32_t, g1);
        MZ_LANES {
            const uint32_t pos = p + (uint32_t)lane;
            uint32_t mlen = 0, dist = 0;
            if ((uint32_t)lane < nv && P(hh) != 0xFFFFFFFFu) {
                const uint32_t maxl = (blk_end - pos < MZ_LZE_MAXMATCH) ? (blk_end - pos) : MZ_LZE_MAXMATCH;
                for (uint32_t w = 0; w < MZ_DEF_WAYS_BEST; w++) { /* most recent first: a tie keeps the shorter distance */
                    if (w >= ways) break;
                    const uint32_t d = (pos - (w ? P(candx)[w - 1u] : P(cand))) & 0xFFFFu;
                    if (d >= 1u && d <= 32768u && d <= pos - hist && d != dist) {
                        const uint32_t l = mz_match_len(in + pos, in + (pos - d), maxl);
                        if (l >= MZ_DEF_MINMATCH && l > mlen) {
                            mlen = l;
                            dist = d;
                        }
                    }
                }
                {
                    uint32_t fc = P(fcand);
                    for (uint32_t dep = 0; fc && dep < far_depth; dep++) {
                        const uint32_t d = pos + 1u - fc;
                        if (d > MZ_LZE_FAR_MAXDIST) break;
                        if (d != dist) {
                            const uint32_t l = mz_match_len(in + pos, in + (pos - d), maxl);
                            if (l >= MZ_DEF_MINMATCH && l > mlen + ((mlen && d > 32768u) ? MZ_LZE_FAR_GAIN : 0u) && (d <= 32768u || l >= MZ_LZE_FAR_MINLEN)) {
                                mlen = l;
                                dist = d;
                            }
                        }
                        fc = (dep + 1u < far_depth) ? links[fc - 1u] : 0u;
                    }
                }
            }
            P(pk) = mlen | ((mlen ? dist : 0u) << 9);
            P(lit) = (uint32_t)in[pos < blk_end ? pos : blk];
        }
        PV(uint32_t, pkn);
        PV(uint32_t, pkn2);
#ifndef MZ_LZE_INHERIT
#define MZ_LZE_INHERIT 1
#endif
        if (MZ_LZE_INHERIT && ways > 1u) {
            /* a match of length L at position q is a match of length L - k at q + k, same distance: a position whose own
             * bucket had lost that candidate inherits it from the lanes 1 and 2 below (deflate_core.h does the same) */
            for (uint32_t sh = 1u; sh <= 2u; sh <<= 1) {
                PV(uint32_t, pin);
                MZ_GATHER4(pin, pk, 4u * (((uint32_t)lane - sh) & 63u));
                MZ_LANES {
                    const uint32_t il = P(pin) & 511u;
                    if ((uint32_t)lane >= sh && (uint32_t)lane < nv && il >= MZ_DEF_MINMATCH + sh && il - sh > (P(pk) & 511u))
                        P(pk) = (il - sh) | (P(pin) & ~511u);
                }
            }
        }
        MZ_GATHER4(pkn, pk, 4u * ((uint32_t)lane + 1u));
        MZ_GATHER4(pkn2, pk, 4u * ((uint32_t)lane + 2u));
        MZ_LANES {
            uint32_t mlen = P(pk) & 511u;
            if (mlen && (uint32_t)lane + 1u < nv && (P(pkn) & 511u) > mlen) mlen = 0u; /* lazy rule */
            if (ways > 1u && mlen && (uint32_t)lane + 2u < nv && (P(pkn2) & 511u) > mlen + 1u) mlen = 0u; /* two ahead */
            P(pk) = mlen ? P(pk) : (P(lit) << 9);
            const uint32_t nx = (uint32_t)lane + (mlen ? mlen : 1u);
            P(g1) = ((uint32_t)lane >= nv || nx >= nv) ? (0x1000u | (4u * nx)) : (4u * nx);
        }
        PV(uint32_t, g2);
        PV(uint32_t, g4);
        PV(uint32_t, g8);
        PV(uint32_t, g16);
        PV(uint32_t, g32);
        PV(uint32_t, gt);
        PV(uint32_t, ct);
        PV(uint32_t, cpos);
        MZ_LANES { P(cpos) = (skip >= nv) ? (0x1000u | (4u * skip)) : (4u * skip); }
#define MZ_LZT_ROUND(gin, gout, bit)                                                         \
    MZ_GATHER4(gt, gin, P(gin));                                                             \
    MZ_GATHER4(ct, gin, P(cpos));                                                            \
    MZ_LANES {                                                                               \
        P(gout) = (P(gin) & 0x1000u) ? P(gin) : P(gt);                                       \
        P(cpos) = (((uint32_t)lane & (bit)) && !(P(cpos) & 0x1000u)) ? P(ct) : P(cpos);      \
    }
        MZ_LZT_ROUND(g1, g2, 1u)
        MZ_LZT_ROUND(g2, g4, 2u)
        MZ_LZT_ROUND(g4, g8, 4u)
        MZ_LZT_ROUND(g8, g16, 8u)
        MZ_LZT_ROUND(g16, g32, 16u)
        MZ_GATHER4(ct, g32, P(cpos));
        MZ_LANES { P(cpos) = (((uint32_t)lane & 32u) && !(P(cpos) & 0x1000u)) ? P(ct) : P(cpos); }
#undef MZ_LZT_ROUND
        uint64_t live;
        MZ_BALLOT(live, !(P(cpos) & 0x1000u));
        const uint32_t ntok = mz_popc64(live);
        PV(uint32_t, tpk);
        MZ_GATHER4(tpk, pk, P(cpos));
        if (ntok) {
            const uint32_t lastpos = MZ_READLANE(cpos, ntok - 1u) >> 2;
            const uint32_t lastpk = MZ_READLANE(tpk, ntok - 1u);
            const uint32_t nx = lastpos + ((lastpk & 511u) ? (lastpk & 511u) : 1u);
            skip = nx > 64u ? nx - 64u : 0u;
        } else {
            skip = skip > 64u ? skip - 64u : 0u;
        }
        MZ_LANES {
            if ((uint32_t)lane < ntok) tok[ntokens + (uint32_t)lane] = P(tpk);
        }
        ntokens += ntok;
    }
    MZ_WAVE_SYNC();
    return ntokens;
}

/* Like K3, the coder arithmetic can issue on the scalar port (LZE_U = readfirstlane) or stay in VGPRs and issue on
 * the vector ports (identity); see DESIGN.md K3 / K6 for the measurement that picks the default. */
#ifndef LZE_U
#if defined(MZHIP_HOST_EMUL) || defined(MZ_LZE_SCALAR_PORT)
#define LZE_U(x) MZ_UNIFORM(x)
#else
#define LZE_U(x) (x)
#endif
#endif

/* ---------------------------------------------------------------------------------------------------------
 * Range ENCODER, wave-uniform.  low is 33 bits wide; a byte leaves only when no later carry can change it. */
#define LZE_OUT_BYTE(b)                                                                  \
    do {                                                                                 \
        ow |= (uint32_t)(b) << (8u * (on & 3u));                                         \
        on++;                                                                            \
        if ((on & 3u) == 0u) {                                                           \
            MZ_WRITELANE(owin, ((on - 1u) >> 2) & 63u, ow);                              \
            ow = 0;                                                                      \
            if ((on & 255u) == 0u) LZE_FLUSH_WIN(256u);                                  \
        }                                                                                \
    } while (0)
/* store the first nb (<= 256) bytes of the window */
#define LZE_FLUSH_WIN(nb)                                                                \
    do {                                                                                 \
        const uint32_t _base = (on - 1u) & ~255u;                                        \
        if (_base + (nb) > out_cap) {                                                    \
            status = MZHIP_OUT_FULL;                                                     \
            goto finish;                                                                 \
        }                                                                                \
        MZ_LANES {                                                                       \
            for (uint32_t _k = 0; _k < 4u; _k++)                                         \
                if (4u * (uint32_t)lane + _k < (nb)) out[_base + 4u * (uint32_t)lane + _k] = (uint8_t)(P(owin) >> (8u * _k)); \
        }                                                                                \
    } while (0)
#define LZE_SHIFT_LOW()                                                                  \
    do {                                                                                 \
        if ((uint32_t)low < 0xFF000000u || (uint32_t)(low >> 32) != 0u) {                \
            const uint32_t _carry = (uint32_t)(low >> 32);                               \
            do {                                                                         \
                LZE_OUT_BYTE((cache + _carry) & 0xFFu);                                  \
                cache = 0xFFu;                                                           \
            } while (--cache_size != 0u);                                                \
            cache = ((uint32_t)low >> 24) & 0xFFu;                                       \
        }                                                                                \
        cache_size++;                                                                    \
        low = (uint64_t)((uint32_t)low & 0x00FFFFFFu) << 8;                              \
    } while (0)
/* The coder knows every bit before it codes it, so the probabilities of a whole symbol's path are read up front
 * (MZ_LZE_PRELOAD: eight independent LDS reads for a literal, then eight codings that wait for nothing) instead of one
 * LDS round trip on the range's dependency chain per bit.  LZE_BIT_P codes one bit with the probability in hand. */
#ifndef MZ_LZE_PRELOAD
#define MZ_LZE_PRELOAD 1
#endif
#define LZE_BIT(idx, bitv) LZE_BIT_P(idx, bitv, LZE_U(pr[(idx)]))
#define LZE_BIT_P(idx, bitv, pv)                                                         \
    do {                                                                                 \
        const uint32_t _pi = (idx);                                                      \
        uint32_t _p = (pv);                                                              \
        const uint32_t _bound = (range >> 11) * _p;                                      \
        if (!(bitv)) {                                                                   \
            range = _bound;                                                              \
            _p += (2048u - _p) >> 5;                                                     \
        } else {                                                                         \
            low += _bound;                                                               \
            range -= _bound;                                                             \
            _p -= _p >> 5;                                                               \
        }                                                                                \
        MZ_LANES { pr[_pi] = (uint16_t)_p; } /* uniform store */                         \
        MZ_WAVE_SYNC();                                                                  \
        while (range < (1u << 24)) {                                                     \
            range <<= 8;                                                                 \
            LZE_SHIFT_LOW();                                                             \
        }                                                                                \
    } while (0)
#define LZE_DIRECT(val, nbits)                                                           \
    do {                                                                                 \
        for (int _i = (int)(nbits) - 1; _i >= 0; _i--) {                                 \
            range >>= 1;                                                                 \
            if (((val) >> _i) & 1u) low += range;                                        \
            while (range < (1u << 24)) {                                                 \
                range <<= 8;                                                             \
                LZE_SHIFT_LOW();                                                         \
            }                                                                            \
        }                                                                                \
    } while (0)
#if MZ_LZE_PRELOAD
#define LZE_BITTREE(base, nbits, sym)                                                    \
    do {                                                                                 \
        uint32_t _pp[8];                                                                 \
        {                                                                                \
            uint32_t _m = 1;                                                             \
            _Pragma("unroll") for (int _k = 0; _k < (int)(nbits); _k++) {                \
                _pp[_k] = LZE_U(pr[(base) + _m]);                                        \
                _m = (_m << 1) | (((sym) >> ((int)(nbits) - 1 - _k)) & 1u);              \
            }                                                                            \
        }                                                                                \
        uint32_t _m = 1;                                                                 \
        _Pragma("unroll") for (int _k = 0; _k < (int)(nbits); _k++) {                    \
            const uint32_t _b = ((sym) >> ((int)(nbits) - 1 - _k)) & 1u;                 \
            LZE_BIT_P((base) + _m, _b, _pp[_k]);                                         \
            _m = (_m << 1) | _b;                                                         \
        }                                                                                \
    } while (0)
/* nbits <= 5 here (the distance's low bits below slot 14, the four aligned bits) */
#define LZE_BITTREE_REV(base, nbits, sym)                                                \
    do {                                                                                 \
        uint32_t _pp[5];                                                                 \
        {                                                                                \
            uint32_t _m = 1;                                                             \
            _Pragma("unroll") for (int _k = 0; _k < 5; _k++) {                           \
                _pp[_k] = (_k < (int)(nbits)) ? LZE_U(pr[(base) + _m]) : 0u;             \
                _m = (_m << 1) | (((sym) >> _k) & 1u);                                   \
            }                                                                            \
        }                                                                                \
        uint32_t _m = 1;                                                                 \
        _Pragma("unroll") for (int _k = 0; _k < 5; _k++) {                               \
            if (_k < (int)(nbits)) {                                                     \
                const uint32_t _b = ((sym) >> _k) & 1u;                                  \
                LZE_BIT_P((base) + _m, _b, _pp[_k]);                                     \
                _m = (_m << 1) | _b;                                                     \
            }                                                                            \
        }                                                                                \
    } while (0)
#else
#define LZE_BITTREE(base, nbits, sym)                                                    \
    do {                                                                                 \
        uint32_t _m = 1;                                                                 \
        for (int _i = (int)(nbits) - 1; _i >= 0; _i--) {                                 \
            const uint32_t _b = ((sym) >> _i) & 1u;                                      \
            LZE_BIT((base) + _m, _b);                                                    \
            _m = (_m << 1) | _b;                                                         \
        }                                                                                \
    } while (0)
#define LZE_BITTREE_REV(base, nbits, sym)                                                \
    do {                                                                                 \
        uint32_t _m = 1;                                                                 \
        for (int _i = 0; _i < (int)(nbits); _i++) {                                      \
            const uint32_t _b = ((sym) >> _i) & 1u;                                      \
            LZE_BIT((base) + _m, _b);                                                    \
            _m = (_m << 1) | _b;                                                         \
        }                                                                                \
    } while (0)
#endif
#define LZE_LEN(lbase, l, ps)                                                            \
    do {                                                                                 \
        if ((l) < 8u) {                                                                  \
            LZE_BIT((lbase), 0u);                                                        \
            LZE_BITTREE((lbase) + 2 + (ps) * 8, 3, (l));                                 \
        } else if ((l) < 16u) {                                                          \
            LZE_BIT((lbase), 1u);                                                        \
            LZE_BIT((lbase) + 1, 0u);                                                    \
            LZE_BITTREE((lbase) + 2 + 128 + (ps) * 8, 3, (l) - 8u);                      \
        } else {                                                                         \
            LZE_BIT((lbase), 1u);                                                        \
            LZE_BIT((lbase) + 1, 1u);                                                    \
            LZE_BITTREE((lbase) + 2 + 256, 8, (l) - 16u);                                \
        }                                                                                \
    } while (0)
/* a new (non-rep) match: length then distance (0-based d) */
#define LZE_MATCH_DIST(d, lenm2)                                                         \
    do {                                                                                 \
        uint32_t _slot, _nb = 0;                                                         \
        if ((d) < 4u) {                                                                  \
            _slot = (d);                                                                 \
        } else {                                                                         \
            _nb = 31u - mz_clz32(d);                                                     \
            _slot = 2u * _nb + (((d) >> (_nb - 1u)) & 1u);                               \
        }                                                                                \
        LZE_BITTREE(LZ_POS_SLOT + ((lenm2) < 4u ? (lenm2) : 3u) * 64, 6, _slot);         \
        if (_slot >= 4u) {                                                               \
            const uint32_t _fb = (_slot >> 1) - 1u;                                      \
            const uint32_t _base = (2u | (_slot & 1u)) << _fb;                           \
            const uint32_t _rem = (d) - _base;                                           \
            if (_slot < 14u) {                                                           \
                LZE_BITTREE_REV(LZ_POS_DEC + _base - _slot, _fb, _rem);                  \
            } else {                                                                     \
                LZE_DIRECT(_rem >> 4, _fb - 4u);                                         \
                LZE_BITTREE_REV(LZ_ALIGN, 4, _rem & 15u);                                \
            }                                                                            \
        }                                                                                \
    } while (0)

typedef struct mz_lzma_enc_result {
    int32_t status;
    uint32_t out_len;
    uint32_t crc; /* CRC-32 of the input (what mz_zip_entry_write accumulates, mz_zip.c:2064) */
} mz_lzma_enc_result;

/* Range-code the token blocks of ONE stream.  in[0..in_len) = the stream's input; its tokens sit in blocks of
 * MZ_DEF_BLOCK positions: block b at tok + b * MZ_DEF_BLOCK, ntok[b] of them.  mode 0: a ZIP method-14 payload
 * (4-byte magic, 5 property bytes, range-coded data, end marker); mode 1: the payload of one LZMA2 chunk (raw
 * range-coded data, no end marker).  All arguments wave-uniform. */
/* Where a method-14 stream that is written segment by segment stands between two launches (the drop-in WRITE stream,
 * shim_lzma.c; mzhip.h declares the same sixteen words as mzhip_lzma_enc_state): the range coder -- low, range, the byte
 * held back for a carry and how many 0xFF follow it --, the packet state and the four repeat distances.  The adaptive model
 * (LZ_NUM_PROBS probabilities) travels beside it in global memory. */
typedef struct mz_lzma_enc_state {
    uint32_t flags; /* in: bit 0 go on from this state (else a fresh stream: header first) */
    uint32_t low_lo, low_hi, range, cache, cache_size;
    uint32_t state, rep0, rep1, rep2, rep3;
    uint32_t pad[5];
} mz_lzma_enc_state;

/* skip_blocks: the first blocks of in[] are the previous segment's last bytes (match sources and contexts), already coded;
 * rs / st / model: take the coder up from / leave it in a state (st: no end marker, no flush -- the stream goes on);
 * without them this is the one-shot coder */
MZ_DEV void mz_lzma_rc_encode_x(const uint8_t *in, uint32_t in_len, const uint32_t *tok, const uint32_t *ntok, uint32_t mode,
                                uint8_t *out, uint32_t out_cap, mz_lzma_lds *L, const uint32_t *crc_tab,
                                const mzhip_crc_tables *tabs, mz_lzma_enc_result *res, uint32_t skip_blocks,
                                const mz_lzma_enc_state *rs, mz_lzma_enc_state *st, uint16_t *model) {
    MZ_LANE_DECL
    uint16_t *pr = L->probs;
    int32_t status = MZHIP_OK;
    uint64_t low = 0;
    uint32_t range = 0xFFFFFFFFu, cache = 0, cache_size = 1;
    uint32_t on = 0, ow = 0; /* bytes produced, dword being assembled */
    PV(uint32_t, owin);
    uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0, pos = skip_blocks * MZ_DEF_BLOCK;
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    const uint32_t resuming = (rs && (MZ_UNIFORM(rs->flags) & 1u)) ? 1u : 0u;
    MZ_LANES {
        P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u;
        P(owin) = 0u;
        for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64)
            ((uint32_t *)pr)[i] = resuming ? ((const uint32_t *)model)[i] : 0x04000400u;
    }
    MZ_WAVE_SYNC();
    if (resuming) {
        low = ((uint64_t)MZ_UNIFORM(rs->low_hi) << 32) | MZ_UNIFORM(rs->low_lo);
        range = MZ_UNIFORM(rs->range);
        cache = MZ_UNIFORM(rs->cache);
        cache_size = MZ_UNIFORM(rs->cache_size);
        state = MZ_UNIFORM(rs->state);
        rep0 = MZ_UNIFORM(rs->rep0);
        rep1 = MZ_UNIFORM(rs->rep1);
        rep2 = MZ_UNIFORM(rs->rep2);
        rep3 = MZ_UNIFORM(rs->rep3);
    }
    if (mode == 0u && !resuming) {
        /* version 9.20, property size 5 (what liblzma-based writers put there; the reader ignores the version,
         * mz_strm_lzma.c:118-121), then lc/lp/pb and the dictionary size (appnote.txt:2232-2275) */
        LZE_OUT_BYTE(9u);
        LZE_OUT_BYTE(20u);
        LZE_OUT_BYTE(5u);
        LZE_OUT_BYTE(0u);
        /* the dictionary the distances may need: a stream of more than one block (and every stream written in segments)
         * was parsed with the chain pass's links */
        const uint32_t dictv = (in_len > MZ_DEF_BLOCK || rs) ? MZ_LZE_FAR_DICT : MZ_LZE_DICT;
        LZE_OUT_BYTE(MZ_LZE_PROPS);
        LZE_OUT_BYTE(dictv & 0xFFu);
        LZE_OUT_BYTE((dictv >> 8) & 0xFFu);
        LZE_OUT_BYTE((dictv >> 16) & 0xFFu);
        LZE_OUT_BYTE((dictv >> 24) & 0xFFu);
    }
    {
        const uint32_t nblocks = (in_len + MZ_DEF_BLOCK - 1u) / MZ_DEF_BLOCK;
        for (uint32_t b = skip_blocks; b < nblocks; b++) {
            const uint32_t *bt = tok + (size_t)b * MZ_DEF_BLOCK;
            const uint32_t nt_blk = MZ_UNIFORM(ntok[b]);
            for (uint32_t t0 = 0; t0 < nt_blk; t0 += 64u) {
                const uint32_t nt = (nt_blk - t0 < 64u) ? (nt_blk - t0) : 64u;
                /* 64 tokens prepared in parallel: start position, previous byte, byte at the previous token's
                 * distance (the match byte if this token is a literal right after a match) */
                PV(uint32_t, tk);
                PV(uint32_t, tlen);
                PV(uint32_t, tend);
                PV(uint32_t, tprev);
                PV(uint32_t, tctx);
                MZ_LANES {
                    const uint32_t t = ((uint32_t)lane < nt) ? bt[t0 + (uint32_t)lane] : 0u;
                    P(tk) = t;
                    P(tlen) = ((uint32_t)lane < nt) ? ((t & 511u) ? (t & 511u) : 1u) : 0u;
                }
                MZ_INCL_SCAN(tend, tlen);
                MZ_GATHER4(tprev, tk, 4u * ((uint32_t)lane - 1u));
                MZ_LANES {
                    const uint32_t p = pos + P(tend) - P(tlen);
                    uint32_t pb = 0, mb = 0;
                    if ((uint32_t)lane < nt) {
                        if (p) pb = in[p - 1u];
                        /* distance of the token before this one; lane 0 takes the coder's rep0 */
                        const uint32_t pd = ((uint32_t)lane == 0u) ? rep0 + 1u : ((P(tprev) & 511u) ? (P(tprev) >> 9) : 0u);
                        if (pd && pd <= p) mb = in[p - pd];
                    }
                    P(tctx) = pb | (mb << 8);
                }
                for (uint32_t i = 0; i < nt; i++) {
                    const uint32_t t = MZ_READLANE(tk, i), ctx = MZ_READLANE(tctx, i);
                    const uint32_t mlen = t & 511u, ps = pos & ((1u << MZ_LZE_PB) - 1u);
                    if (mlen == 0u) {
                        const uint32_t sym = (t >> 9) & 0xFFu;
                        LZE_BIT(LZ_IS_MATCH + state * 16 + ps, 0u);
                        const uint32_t lbase = LZ_LIT + 0x300u * ((ctx & 0xFFu) >> (8u - MZ_LZE_LC));
#if MZ_LZE_PRELOAD
                        {
                            /* the eight nodes of this literal's path -- in the matched trees while its bits follow the match
                             * byte, in the plain tree from the first difference on -- and their probabilities, up front */
                            uint32_t li[8], lp[8];
                            uint32_t m = 1, matching = (state >= 7u) ? 1u : 0u, mbyte = (ctx >> 8) & 0xFFu;
                            _Pragma("unroll") for (int j = 0; j < 8; j++) {
                                const uint32_t bb = (sym >> (7 - j)) & 1u, mbit = (mbyte >> 7) & 1u;
                                mbyte <<= 1;
                                li[j] = lbase + (matching ? ((1u + mbit) << 8) : 0u) + m;
                                lp[j] = LZE_U(pr[li[j]]);
                                if (mbit != bb) matching = 0u;
                                m = (m << 1) | bb;
                            }
                            _Pragma("unroll") for (int j = 0; j < 8; j++) LZE_BIT_P(li[j], (sym >> (7 - j)) & 1u, lp[j]);
                        }
#else
                        uint32_t m = 1;
                        int k = 7;
                        if (state >= 7u) {
                            uint32_t mbyte = (ctx >> 8) & 0xFFu;
                            for (; k >= 0; k--) {
                                const uint32_t mbit = (mbyte >> 7) & 1u, bb = (sym >> k) & 1u;
                                mbyte <<= 1;
                                LZE_BIT(lbase + ((1u + mbit) << 8) + m, bb);
                                m = (m << 1) | bb;
                                if (mbit != bb) {
                                    k--;
                                    break;
                                }
                            }
                        }
                        for (; k >= 0; k--) {
                            const uint32_t bb = (sym >> k) & 1u;
                            LZE_BIT(lbase + m, bb);
                            m = (m << 1) | bb;
                        }
#endif
                        state = state < 4u ? 0u : (state < 10u ? state - 3u : state - 6u);
                        pos += 1u;
                    } else {
                        const uint32_t d = (t >> 9) - 1u; /* 0-based distance */
                        LZE_BIT(LZ_IS_MATCH + state * 16 + ps, 1u);
                        if (d == rep0 || d == rep1 || d == rep2 || d == rep3) {
                            LZE_BIT(LZ_IS_REP + state, 1u);
                            if (d == rep0) {
                                LZE_BIT(LZ_IS_REP_G0 + state, 0u);
                                LZE_BIT(LZ_IS_REP0_LONG + state * 16 + ps, 1u);
                            } else {
                                LZE_BIT(LZ_IS_REP_G0 + state, 1u);
                                if (d == rep1) {
                                    LZE_BIT(LZ_IS_REP_G1 + state, 0u);
                                } else {
                                    LZE_BIT(LZ_IS_REP_G1 + state, 1u);
                                    if (d == rep2) {
                                        LZE_BIT(LZ_IS_REP_G2 + state, 0u);
                                    } else {
                                        LZE_BIT(LZ_IS_REP_G2 + state, 1u);
                                        rep3 = rep2;
                                    }
                                    rep2 = rep1;
                                }
                                rep1 = rep0;
                                rep0 = d;
                            }
                            LZE_LEN(LZ_REP_LEN, mlen - 2u, ps);
                            state = state < 7u ? 8u : 11u;
                        } else {
                            LZE_BIT(LZ_IS_REP + state, 0u);
                            LZE_LEN(LZ_LEN, mlen - 2u, ps);
                            LZE_MATCH_DIST(d, mlen - 2u);
                            rep3 = rep2;
                            rep2 = rep1;
                            rep1 = rep0;
                            rep0 = d;
                            state = state < 7u ? 7u : 10u;
                        }
                        pos += mlen;
                    }
                }
                if (!skip_blocks && !st) MZ_CRC_FOLD_TILES(crc_acc, crc_done, in, pos, crc_tab, tabs->kx);
            }
        }
    }
    if (st) {
        /* the stream goes on in the next launch: the coder as it stands (nothing is flushed: `low` and the held-back byte
         * are state), the model back to global memory; what this segment produced is complete bytes */
        MZ_LANES { /* uniform stores */
            st->flags = 1u;
            st->low_lo = (uint32_t)low;
            st->low_hi = (uint32_t)(low >> 32);
            st->range = range;
            st->cache = cache;
            st->cache_size = cache_size;
            st->state = state;
            st->rep0 = rep0;
            st->rep1 = rep1;
            st->rep2 = rep2;
            st->rep3 = rep3;
            for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64) ((uint32_t *)model)[i] = ((const uint32_t *)pr)[i];
        }
        MZ_WAVE_SYNC();
    } else if (mode == 0u) {
        /* end marker: a match of the minimum length at distance 0xFFFFFFFF (appnote.txt:2262-2275, flag bit 1) */
        const uint32_t ps = pos & ((1u << MZ_LZE_PB) - 1u);
        LZE_BIT(LZ_IS_MATCH + state * 16 + ps, 1u);
        LZE_BIT(LZ_IS_REP + state, 0u);
        LZE_LEN(LZ_LEN, 0u, ps);
        LZE_MATCH_DIST(0xFFFFFFFFu, 0u);
    }
    if (!st)
        for (int i = 0; i < 5; i++) LZE_SHIFT_LOW();
    /* what is left in the window */
    if (on & 3u) MZ_WRITELANE(owin, (on >> 2) & 63u, ow);
    if (on & 255u) {
        const uint32_t nb = on & 255u;
        on += 256u - nb; /* LZE_FLUSH_WIN addresses the window that ends at `on` */
        LZE_FLUSH_WIN(nb);
        on -= 256u - nb;
    }
    MZ_WAVE_SYNC();

finish:
    res->status = status;
    res->out_len = on;
    {
        uint32_t crc = 0;
        if (!skip_blocks && !st && !resuming) { /* (a stream written in segments gets no fused CRC: the zip layer asks for its own, mz_zip.c:2064) */
            MZ_CRC_FOLD_TILES(crc_acc, crc_done, in, in_len, crc_tab, tabs->kx);
            MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, in, in_len, crc_tab, tabs);
        } else {
            (void)crc_tmp;
        }
        res->crc = crc;
    }
}
MZ_DEV void mz_lzma_rc_encode(const uint8_t *in, uint32_t in_len, const uint32_t *tok, const uint32_t *ntok, uint32_t mode,
                              uint8_t *out, uint32_t out_cap, mz_lzma_lds *L, const uint32_t *crc_tab,
                              const mzhip_crc_tables *tabs, mz_lzma_enc_result *res) {
    mz_lzma_rc_encode_x(in, in_len, tok, ntok, mode, out, out_cap, L, crc_tab, tabs, res, 0u, (const mz_lzma_enc_state *)0,
                        (mz_lzma_enc_state *)0, (uint16_t *)0);
}

#endif
