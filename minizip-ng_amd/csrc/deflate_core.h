/* deflate_core.h -- raw-DEFLATE ENCODE of ONE piece by ONE wavefront: LZ77 tokens chosen 64 positions at a
 * time, then per 64 KiB block the cheapest of dynamic-Huffman / fixed-Huffman / stored coding, CRC-32 of the
 * input fused in (kernel K4 of SURVEY 2.1).
 *
 * Replaces what the reference does per entry through mz_stream_zlib_write / _close
 * (mz_strm_zlib.c:203-264,280-305 -> zlib deflate(), raw) followed by mz_crypt_crc32_update (mz_zip.c:2064).
 * Format: doc/zip/appnote.txt:2030-2166 (block types :2041-2063, dynamic header :2064-2106, fixed code
 * :2050-2059).  The compressed bytes are NOT zlib's (compressor output is not a format property; zlib 1.2.11 and
 * zlib-ng already differ): parity for this kernel is "the reference's inflate of these bytes returns the input,
 * and the CRC matches" (hard parts).
 *
 * MI355X mapping, pass 1 (tokens), 64 input positions per step, one per lane:
 *   - match finding: each lane hashes its 4 bytes, looks the hash up in a per-wave LDS table of recent
 *     positions (read-then-update) and measures the match in place with dword compares (min 4, max 258, 32 KiB);
 *   - greedy token selection is the same successor-chain problem as in the decoder (f(l) = l + len(l) or
 *     l + 1): f is squared five times with cross-lane gathers and lane r composes f^r(start), so the step's
 *     tokens land compacted in lanes 0..n-1 with no serial walk;
 *   - tokens go to a per-wave scratch in HBM (<= 65 536 per block) and into LDS symbol histograms (atomics).
 * Between the passes the wave builds the block's codes: rank sort of the used symbols (each lane counts the keys
 * below its own), two-queue Huffman merge (wave-uniform), depths by walking to the root (per lane), halve-and-retry
 * when a code would exceed 15 (7) bits, canonical codes by counting equal-length predecessors, the code-length
 * sequence run-length coded as in appnote.txt:2070-2090, and the three block costs.
 * Pass 2: 64 tokens per step, <= 48 bits each from the LDS code table, bit offsets from a DPP prefix sum, bits
 * OR-ed into an LDS staging area that is flushed to HBM as whole bytes.
 */
#ifndef MZHIP_DEFLATE_CORE_H
#define MZHIP_DEFLATE_CORE_H

#include "crc32_core.h"
#include "wave.h"

/* MZ_PROF (measurement builds): cycles per section of K4 go to mz_prof_buf[16..] (see inflate_core.h) */
#if defined(MZ_PROF) && !defined(MZHIP_HOST_EMUL)
#define MZ_DPROF_DECL uint32_t prof_acc = 0; uint64_t prof_t0 = __builtin_readcyclecounter();
#define MZ_DPROF_MARK(i)                                                          \
    do {                                                                          \
        const uint32_t _pd = (uint32_t)(__builtin_readcyclecounter() - prof_t0);  \
        prof_acc += (lane == (i)) ? _pd : 0u;                                     \
        prof_t0 = __builtin_readcyclecounter();                                   \
    } while (0)
#define MZ_DPROF_FLUSH if (lane >= 16 && lane < 32) atomicAdd(&mz_prof_buf[lane], (unsigned long long)prof_acc);
#else
#define MZ_DPROF_DECL
#define MZ_DPROF_MARK(i) MZ_DPRIO_AT(i)
#define MZ_DPROF_FLUSH
#endif
/* the wave's issue priority by section, as in inflate_core.h (MZ_PRIO_MAP): two bits per mark i - 16 = the priority of what
 * runs BEHIND mark i (28: behind the step loop: code construction).  Everything at 2 except the match measurement (0: its
 * loads are in flight, the wave has little to issue): 20 000 x 64 KiB at level 1 15.25 -> 14.8 ms (+3 %); the measurement
 * alone on top +1 %, codes and pass 2 alone on top +1 % (profiles/r6/ab_setprio_k3_k4.log); as in K1 it is the start of a
 * launch that gains, config 5's 100 000 pieces run as before (69.1 ms).  -DMZ_DPRIO_MAP=0: without. */
#ifndef MZ_DPRIO_MAP
#define MZ_DPRIO_MAP 0x200aa22ull
#endif
#if MZ_DPRIO_MAP && !defined(MZHIP_HOST_EMUL) && !defined(MZ_PROF)
#define MZ_DPRIO_AT(i) __builtin_amdgcn_s_setprio((int)(((MZ_DPRIO_MAP) >> (2 * ((i) - 16))) & 3ull))
#else
#define MZ_DPRIO_AT(i) ((void)0)
#endif
#ifndef MZ_DEF_HBITS
#define MZ_DEF_HBITS 12
#endif
#define MZ_DEF_MINMATCH 4u
#define MZ_DEF_MAXMATCH 258u
#define MZ_DEF_BLOCK 65536u /* input positions per DEFLATE block = token scratch entries per wave */
#define MZ_DEF_NLIT 286u
#define MZ_DEF_NDIST 30u
#define MZ_DEF_NCL 19u
#define MZ_DEF_DIST0 MZ_DEF_NLIT                 /* offsets into freq[] / code[] */
#define MZ_DEF_CL0 (MZ_DEF_NLIT + MZ_DEF_NDIST)
#define MZ_DEF_NSYM (MZ_DEF_NLIT + MZ_DEF_NDIST + MZ_DEF_NCL)

typedef struct mz_deflate_lds {
    union {
        uint16_t head[1 << MZ_DEF_HBITS]; /* pass 1: low 16 bits of the most recent position with this hash */
        struct {                          /* after pass 1 (the hash table is dead): code construction and pass 2 */
            uint32_t wt[2 * MZ_DEF_NLIT];     /* node weights: sorted leaves, then internal nodes */
            uint32_t sf[MZ_DEF_NLIT];         /* frequencies as scaled for the current attempt */
            uint16_t parent[2 * MZ_DEF_NLIT];
            uint16_t sym[MZ_DEF_NLIT];        /* sorted position -> symbol */
            uint8_t cl_seq[MZ_DEF_NLIT + MZ_DEF_NDIST + 4]; /* header: code-length alphabet symbols ... */
            uint8_t cl_ext[MZ_DEF_NLIT + MZ_DEF_NDIST + 4]; /* ... and their extra-bit values */
            uint32_t code[MZ_DEF_NSYM + 1];   /* bit-reversed code | length << 16 */
            uint32_t first[16];               /* canonical first code per length; bl_count while it is being made */
            uint32_t stage[104];              /* pass 2: this step's bits (<= 8 + 64 x 48) */
            uint8_t lens[MZ_DEF_NSYM + 1];
        } hb;
    } u;
    uint32_t freq[MZ_DEF_NSYM + 1]; /* histograms (live during pass 1): literal/length, distance, code-length alphabet */
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

/* Length of the common prefix of a[] and b[], at most maxl.  Sixteen bytes per round: the eight dword loads of a
 * round are issued together, so a typical match (shorter than 16 bytes) costs one memory round trip, not one per
 * four bytes. */
MZ_DEV uint32_t mz_match_len(const uint8_t *a, const uint8_t *b, uint32_t maxl) {
    uint32_t l = 0;
    while (l + 16u <= maxl) {
        const uint32_t x0 = mz_load_u32(a + l) ^ mz_load_u32(b + l);
        const uint32_t x1 = mz_load_u32(a + l + 4u) ^ mz_load_u32(b + l + 4u);
        const uint32_t x2 = mz_load_u32(a + l + 8u) ^ mz_load_u32(b + l + 8u);
        const uint32_t x3 = mz_load_u32(a + l + 12u) ^ mz_load_u32(b + l + 12u);
        if ((x0 | x1 | x2 | x3) == 0u) {
            l += 16u;
            continue;
        }
        /* first differing byte: little-endian dwords, so the lowest set bit of the first non-zero XOR */
        const uint32_t x = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3;
        const uint32_t k = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u;
        return l + k + ((31u - mz_clz32(x & (0u - x))) >> 3);
    }
    while (l + 4u <= maxl && mz_load_u32(a + l) == mz_load_u32(b + l)) l += 4u;
    while (l < maxl && a[l] == b[l]) l++;
    return l;
}

/* length 3..258 -> symbol 257..285, number and value of extra bits (appnote.txt:2107-2120) */
MZ_DEV uint32_t mz_len_sym(uint32_t mlen, uint32_t *ex, uint32_t *xv) {
    const uint32_t l = mlen - 3u;
    *ex = 0;
    *xv = 0;
    if (mlen == 258u) return 285u;
    if (l < 8u) return 257u + l;
    const uint32_t k = 31u - mz_clz32(l);
    *ex = k - 2u;
    *xv = l & ((1u << (k - 2u)) - 1u);
    return 257u + 4u * (k - 2u) + 4u + ((l >> (k - 2u)) & 3u);
}

/* distance 1..32768 -> symbol 0..29 (appnote.txt:2121-2133) */
MZ_DEV uint32_t mz_dist_sym(uint32_t dist, uint32_t *ex, uint32_t *xv) {
    const uint32_t d = dist - 1u;
    *ex = 0;
    *xv = 0;
    if (d < 4u) return d;
    const uint32_t k = 31u - mz_clz32(d);
    *ex = k - 1u;
    *xv = d & ((1u << (k - 1u)) - 1u);
    return 2u * k + ((d >> (k - 1u)) & 1u);
}

/* position i of the code-length code lengths in the header: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
 * (appnote.txt:2083-2084), five bits per entry */
MZ_DEV uint32_t mz_cl_order(uint32_t i) {
    const uint64_t lo = 0x22CAA324E804A30ull, hi = 0x3C2E1346Cull;
    return (uint32_t)((i < 12u ? lo >> (5u * i) : hi >> (5u * (i - 12u))) & 31u);
}

MZ_DEV uint32_t mz_fixed_litlen_bits(uint32_t sym) { return sym < 144u ? 8u : sym < 256u ? 9u : sym < 280u ? 7u : 8u; }

/* Code lengths (<= maxbits) and canonical codes for the n-symbol alphabet at freq[base..]: results in
 * L->u.hb.lens[base + s] and L->u.hb.code[base + s] = reversed code | len << 16.  Uses L->u.hb (the hash table is dead). */
MZ_DEV void mz_huff_build(mz_deflate_lds *L, uint32_t base, uint32_t n, uint32_t maxbits) {
    MZ_LANE_DECL
    uint32_t *sf = L->u.hb.sf, *wt = L->u.hb.wt;
    uint16_t *parent = L->u.hb.parent, *sym = L->u.hb.sym;
    uint8_t *lens = L->u.hb.lens + base;
    PV(uint32_t, cnt);
    MZ_LANES {
        uint32_t c = 0;
        for (uint32_t s = (uint32_t)lane; s < n; s += 64u) {
            const uint32_t f = L->freq[base + s];
            sf[s] = f;
            lens[s] = 0;
            c += f != 0u;
        }
        P(cnt) = c;
    }
    MZ_WAVE_SYNC();
    uint32_t used;
    MZ_WAVE_SUM(used, cnt);
    if (used < 2u) {
        /* a code needs two leaves; an unused symbol is given weight 1 (as zlib's build_tree does) */
        const uint32_t f0 = MZ_UNIFORM(sf[0]);
        MZ_LANES {
            if (used == 0u || f0 == 0u) sf[0] = 1u;
            if (used == 0u || f0 != 0u) sf[1] = 1u;
        }
        MZ_WAVE_SYNC();
        used = 2u;
    }
    const uint32_t m = used, root = 2u * m - 2u;
    for (;;) {
        /* rank sort by (frequency, symbol): every lane counts the keys below its own */
        MZ_LANES {
            for (uint32_t s = (uint32_t)lane; s < n; s += 64u) {
                const uint32_t f = sf[s];
                if (f) {
                    const uint32_t key = (f << 9) | s;
                    uint32_t r = 0;
                    for (uint32_t t = 0; t < n; t++) {
                        const uint32_t ft = sf[t];
                        r += (ft != 0u && ((ft << 9) | t) < key) ? 1u : 0u;
                    }
                    wt[r] = f;
                    sym[r] = (uint16_t)s;
                }
            }
        }
        MZ_WAVE_SYNC();
        /* two-queue merge, wave-uniform: leaves wt[0..m), internal nodes wt[m..2m-1) in creation order */
        {
            uint32_t i = 0, j = m;
            uint32_t wi = MZ_UNIFORM(wt[0]), wj = 0xFFFFFFFFu;
            for (uint32_t k = m; k <= root; k++) {
                uint32_t a, b, wa, wb;
                if (i < m && (j >= k || wi <= wj)) {
                    a = i++;
                    wa = wi;
                    wi = i < m ? MZ_UNIFORM(wt[i]) : 0xFFFFFFFFu;
                } else {
                    a = j++;
                    wa = wj;
                    wj = j < k ? MZ_UNIFORM(wt[j]) : 0xFFFFFFFFu;
                }
                if (i < m && (j >= k || wi <= wj)) {
                    b = i++;
                    wb = wi;
                    wi = i < m ? MZ_UNIFORM(wt[i]) : 0xFFFFFFFFu;
                } else {
                    b = j++;
                    wb = wj;
                    wj = j < k ? MZ_UNIFORM(wt[j]) : 0xFFFFFFFFu;
                }
                MZ_LANES {
                    wt[k] = wa + wb;
                    parent[a] = (uint16_t)k;
                    parent[b] = (uint16_t)k;
                }
                MZ_WAVE_SYNC();
                if (j == k) wj = wa + wb; /* the internal queue was empty: the new node is its head */
            }
        }
        /* depth of every leaf = its code length */
        PV(uint32_t, deep);
        MZ_LANES {
            uint32_t md = 0;
            for (uint32_t r = (uint32_t)lane; r < m; r += 64u) {
                uint32_t d = 0, k = r;
                while (k != root) {
                    k = parent[k];
                    d++;
                }
                lens[sym[r]] = (uint8_t)d;
                md = d > md ? d : md;
            }
            P(deep) = md;
        }
        MZ_WAVE_SYNC();
        uint64_t over;
        MZ_BALLOT(over, P(deep) > maxbits);
        if (!over) break;
        MZ_LANES {
            for (uint32_t s = (uint32_t)lane; s < n; s += 64u) {
                const uint32_t f = sf[s];
                if (f) sf[s] = (f + 1u) >> 1;
            }
        }
        MZ_WAVE_SYNC();
    }
    /* canonical codes (appnote.txt:2091-2106): count per length, first code per length, then rank inside the length */
    MZ_LANES {
        if (lane < 16) L->u.hb.first[lane] = 0u;
    }
    MZ_WAVE_SYNC();
    MZ_LANES {
        for (uint32_t s = (uint32_t)lane; s < n; s += 64u)
            if (lens[s]) MZ_LDS_ATOMIC_INC(&L->u.hb.first[lens[s]]);
    }
    MZ_WAVE_SYNC();
    {
        uint32_t c = 0, prev = 0;
        for (uint32_t b = 1; b <= 15u; b++) {
            const uint32_t nb = MZ_UNIFORM(L->u.hb.first[b]);
            c = (c + prev) << 1;
            prev = nb;
            MZ_LANES { L->u.hb.first[b] = c; }
        }
        MZ_WAVE_SYNC();
    }
    MZ_LANES {
        for (uint32_t s = (uint32_t)lane; s < n; s += 64u) {
            const uint32_t l = lens[s];
            uint32_t v = 0;
            if (l) {
                uint32_t idx = 0;
                for (uint32_t t = 0; t < s; t++) idx += (lens[t] == l) ? 1u : 0u;
                v = (mz_brev32(L->u.hb.first[l] + idx) >> (32u - l)) | (l << 16);
            }
            L->u.hb.code[base + s] = v;
        }
    }
    MZ_WAVE_SYNC();
}

/* append n (<= 32) bits to the output; keeps fewer than 8 bits pending */
#define MZ_PUTBITS(v, n)                                                   \
    do {                                                                   \
        acc |= (uint64_t)(v) << nacc;                                      \
        nacc += (n);                                                       \
        while (nacc >= 8u) {                                               \
            if (obyte == out_cap) {                                        \
                status = MZHIP_OUT_FULL;                                   \
                goto finish;                                               \
            }                                                              \
            MZ_LANES { out[obyte] = (uint8_t)acc; } /* uniform store */    \
            obyte++;                                                       \
            acc >>= 8;                                                     \
            nacc -= 8u;                                                    \
        }                                                                  \
    } while (0)

/* Encode in[warm..in_len) as one raw-DEFLATE piece.  final != 0: a complete stream (last block BFINAL, padded to a
 * byte).  warm (a multiple of 64, 0 for a piece that stands alone): in[0..warm) are the bytes in front of the piece in the
 * SAME stream -- they are hashed into the buckets before the first position is coded and matches may reach back into them
 * (DEFLATE's window spans blocks), so a stream can be cut into pieces for as many waves without paying for it in matches
 * that end at the cut; in_len - warm <= MZ_DEF_BLOCK then (one block: positions are 16 bits wide).  final == 0: non-final blocks followed by an empty stored block, so that pieces of one stream can be
 * concatenated on byte boundaries.  tok = this wave's token scratch (MZ_DEF_BLOCK words).  All arguments
 * wave-uniform. */
/* ways / xhead: the match finder keeps the `ways` most recent positions of every hash bucket (1 = the fast class: zlib
 * levels 1-3; MZ_DEF_WAYS_BEST = the default class: levels 4-9 and -1, mz_strm_zlib.c:87,339-343).  Way 0 is L->u.head;
 * ways 1.. are xhead[(w - 1) << MZ_DEF_HBITS | hash], extra LDS behind the wave's mz_deflate_lds (may be NULL for 1).
 * max_dist: the largest distance a match may use = window - 262 (zlib's MAX_DIST; 32 506 for the 32 KiB window).
 * parse (needs ways > 1): 0 = the lazy rule decides inside every step of 64 positions (levels 4-6 and the default);
 * 1 = COST PARSE (levels 7-9): pass 1 only records every position's best match and counts the lazy choice; the code lengths of that
 * count are the price list of a backward dynamic programme over the block -- the cheapest way from every position to
 * the block's end, a match free to end early (any length from 4 up to the one found is a valid match at the same
 * distance) -- whose choices are then picked up front to back and counted again for the real codes.  What zlib's lazy
 * matching cannot do (shorten a match so that a better one can start, price a far distance against three literals)
 * is worth 3.3 % of the output on the bench corpus with the same match finder (tests/study/enc_parse.c). */
#define MZ_DEF_WAYS_BEST 4u
template <uint32_t parse> /* (a template argument: the cost parse needs 160 registers, the other classes run four waves per SIMD on 128) */
MZ_DEV void mz_deflate_piece(const uint8_t *in, uint32_t in_len, uint32_t warm, uint8_t *out, uint32_t out_cap, uint32_t final,
                             uint32_t *tok, mz_deflate_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs,
                             uint32_t ways, uint16_t *xhead, uint32_t max_dist, mz_deflate_result *res) {
    MZ_LANE_DECL
    int32_t status = MZHIP_OK;
    uint32_t obyte = 0; /* whole bytes already written to out */
    uint64_t acc = 0;   /* pending bits, fewer than 8 between operations */
    uint32_t nacc = 0;
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    MZ_LANES { P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u; }
    MZ_DPROF_DECL

    const uint8_t *const cin = in + warm; /* the piece's own bytes: what the CRC runs over */
    for (uint32_t blk = warm; blk < in_len || blk == warm; blk += MZ_DEF_BLOCK) {
        const uint32_t blk_end = (in_len - blk < MZ_DEF_BLOCK) ? in_len : blk + MZ_DEF_BLOCK;
        const uint32_t lo = (blk == warm) ? 0u : blk; /* the oldest position a match of this block may start at */
        const uint32_t bfinal = (final && blk_end == in_len) ? 1u : 0u;
        /* ================= pass 1: tokens and histograms ================= */
        MZ_LANES {
            for (uint32_t i = (uint32_t)lane; i < (1u << MZ_DEF_HBITS) / 2u; i += 64u) ((uint32_t *)L->u.head)[i] = 0u;
            for (uint32_t i = (uint32_t)lane; i < (ways - 1u) * ((1u << MZ_DEF_HBITS) / 2u); i += 64u) ((uint32_t *)xhead)[i] = 0u;
            for (uint32_t i = (uint32_t)lane; i <= MZ_DEF_NSYM; i += 64u) L->freq[i] = 0u;
        }
        MZ_WAVE_SYNC();
        if (blk == warm && warm) {
            /* the bytes in front of the piece into the buckets, 64 positions a step, nothing else: what the coded steps below do
             * with a position's bucket, without the look at its candidates */
            for (uint32_t p = 0; ways == 1u && p < warm; p += 64u) { /* one way: the bucket is the newest position, nothing moves down */
                MZ_LANES {
                    const uint32_t pos = p + (uint32_t)lane;
                    L->u.head[(mz_load_u32(in + pos) * 2654435761u) >> (32 - MZ_DEF_HBITS)] = (uint16_t)pos;
                }
            }
            if (ways == 1u) MZ_WAVE_SYNC();
            for (uint32_t p = 0; ways > 1u && p < warm; p += 64u) {
                PV(uint32_t, wh);
                PV(uint32_t, wc);
                PV2(uint32_t, wcx, MZ_DEF_WAYS_BEST - 1u);
                MZ_LANES {
                    const uint32_t pos = p + (uint32_t)lane;
                    P(wh) = (mz_load_u32(in + pos) * 2654435761u) >> (32 - MZ_DEF_HBITS); /* (pos + 4 <= warm + 3 < in_len or the piece is empty: below) */
                    P(wc) = (uint32_t)L->u.head[P(wh)];
                    for (uint32_t w = 1; w < MZ_DEF_WAYS_BEST; w++) P(wcx)[w - 1u] = (w < ways) ? (uint32_t)xhead[((w - 1u) << MZ_DEF_HBITS) | P(wh)] : 0u;
                }
                MZ_WAVE_SYNC();
                MZ_LANES {
                    for (uint32_t w = MZ_DEF_WAYS_BEST - 1u; w >= 1u; w--)
                        if (w < ways) xhead[((w - 1u) << MZ_DEF_HBITS) | P(wh)] = (uint16_t)(w == 1u ? P(wc) : P(wcx)[w - 2u]);
                    L->u.head[P(wh)] = (uint16_t)(p + (uint32_t)lane);
                }
                MZ_WAVE_SYNC();
            }
        }
        uint32_t ntokens = 0, skip = 0;
        PV(uint32_t, xbits); /* extra bits of this lane's tokens */
        MZ_LANES { P(xbits) = 0; }
        PV(uint32_t, vnx); /* the four bytes at this lane's position of the NEXT step, fetched one step ahead */
        MZ_LANES {
            const uint32_t pos = blk + (uint32_t)lane;
            P(vnx) = (pos + 4u <= blk_end) ? mz_load_u32(in + pos) : 0u;
        }
        /* The step is software-pipelined: while step s measures its matches, step s + 1 has already looked up its
         * candidates (the bucket update of step s is done by then) and its first sixteen bytes -- its own and the newest
         * candidate's -- are on their way, so the measurement of a typical match (shorter than 16 bytes) never waits for
         * memory.  *_n = what the look-up of the next step found. */
        PV(uint32_t, hh_n);
        PV(uint32_t, cand_n);
        PV(uint32_t, own_n); /* the step's own four bytes (the literal is their low byte) */
        PV2(uint32_t, candx_n, MZ_DEF_WAYS_BEST - 1u);
        PV(uint32_t, pre_n); /* 1: pa_n / pb_n hold the first 16 bytes at the position and at the newest candidate */
        PV2(uint32_t, pa_n, 4);
        PV2(uint32_t, pb_n, 4);
#define MZ_DEF_LOOKUP(pn)                                                                                              \
        MZ_LANES {                                                                                                     \
            const uint32_t pos = (pn) + (uint32_t)lane;                                                                \
            const uint32_t have4 = (pos + 4u <= blk_end) ? 1u : 0u;                                                    \
            const uint32_t v = P(vnx);                                                                                 \
            P(vnx) = (pos + 68u <= blk_end) ? mz_load_u32(in + pos + 64u) : 0u;                                        \
            const uint32_t h = (v * 2654435761u) >> (32 - MZ_DEF_HBITS);                                               \
            P(own_n) = v;                                                                                              \
            P(hh_n) = have4 ? h : 0xFFFFFFFFu;                                                                         \
            P(cand_n) = have4 ? (uint32_t)L->u.head[h] : 0u;                                                           \
            for (uint32_t w = 1; w < MZ_DEF_WAYS_BEST; w++)                                                            \
                P(candx_n)[w - 1u] = (have4 && w < ways) ? (uint32_t)xhead[((w - 1u) << MZ_DEF_HBITS) | h] : 0u;       \
            const uint32_t d = (pos - P(cand_n)) & 0xFFFFu;                                                            \
            uint32_t ok = (have4 && pos + 16u <= blk_end && d >= 1u && d <= max_dist && d <= pos - lo) ? 1u : 0u;      \
            P(pre_n) = ok;                                                                                             \
            if (ok) {                                                                                                  \
                for (uint32_t k = 0; k < 4u; k++) {                                                                    \
                    P(pa_n)[k] = mz_load_u32(in + pos + 4u * k);                                                       \
                    P(pb_n)[k] = mz_load_u32(in + (pos - d) + 4u * k);                                                 \
                }                                                                                                      \
            }                                                                                                          \
        }
        MZ_DEF_LOOKUP(blk)
        MZ_DPROF_MARK(16); /* block set-up: tables cleared */
        for (uint32_t p = blk; p < blk_end; p += 64u) {
            const uint32_t nv = (blk_end - p < 64u) ? (blk_end - p) : 64u; /* valid positions in this step */
            PV(uint32_t, hh);
            PV(uint32_t, cand);
            PV(uint32_t, own);
            PV2(uint32_t, candx, MZ_DEF_WAYS_BEST - 1u); /* the older positions of the bucket (ways > 1) */
            PV(uint32_t, pre);
            PV2(uint32_t, pa, 4);
            PV2(uint32_t, pb, 4);
            MZ_LANES {
                P(hh) = P(hh_n);
                P(cand) = P(cand_n);
                P(own) = P(own_n);
                P(pre) = P(pre_n);
                for (uint32_t w = 1; w < MZ_DEF_WAYS_BEST; w++) P(candx)[w - 1u] = P(candx_n)[w - 1u];
                for (uint32_t k = 0; k < 4u; k++) {
                    P(pa)[k] = P(pa_n)[k];
                    P(pb)[k] = P(pb_n)[k];
                }
                if (P(hh) != 0xFFFFFFFFu) {
                    /* the bucket shifts by one: lanes that share a bucket write the same older entries, one of them
                     * wins way 0 */
                    for (uint32_t w = MZ_DEF_WAYS_BEST - 1u; w >= 1u; w--)
                        if (w < ways) xhead[((w - 1u) << MZ_DEF_HBITS) | P(hh)] = (uint16_t)(w == 1u ? P(cand) : P(candx)[w - 2u]);
                    L->u.head[P(hh)] = (uint16_t)(p + (uint32_t)lane);
                }
            }
            MZ_WAVE_SYNC();
            if (p + 64u < blk_end) {
                MZ_DEF_LOOKUP(p + 64u)
            }
            MZ_DPROF_MARK(17); /* bucket update, look-up of the next step */
            PV(uint32_t, pk); /* [8:0] match length (0 = literal), [24:9] distance | literal byte */
            PV(uint32_t, lit);
            PV(uint32_t, g1); /* 4 * successor lane; bit 12 set: terminal */
            MZ_LANES {
                const uint32_t pos = p + (uint32_t)lane;
                uint32_t mlen = 0, dist = 0;
                if ((uint32_t)lane < nv && P(hh) != 0xFFFFFFFFu) {
                    const uint32_t maxl = (blk_end - pos < MZ_DEF_MAXMATCH) ? (blk_end - pos) : MZ_DEF_MAXMATCH;
                    for (uint32_t w = 0; w < MZ_DEF_WAYS_BEST; w++) { /* most recent first: a tie keeps the shorter distance */
                        if (w >= ways) break;
                        const uint32_t d = (pos - (w ? P(candx)[w - 1u] : P(cand))) & 0xFFFFu;
                        /* the head table is cleared per block, so a candidate never precedes the block */
                        if (d >= 1u && d <= max_dist && d <= pos - lo && d != dist) {
                            uint32_t l;
                            if (w == 0u && P(pre)) { /* the first 16 bytes are here already */
                                const uint32_t x0 = P(pa)[0] ^ P(pb)[0], x1 = P(pa)[1] ^ P(pb)[1], x2 = P(pa)[2] ^ P(pb)[2],
                                               x3 = P(pa)[3] ^ P(pb)[3];
                                if ((x0 | x1 | x2 | x3) == 0u) {
                                    l = 16u + mz_match_len(in + pos + 16u, in + (pos - d) + 16u, maxl - 16u);
                                } else {
                                    const uint32_t x = x0 ? x0 : x1 ? x1 : x2 ? x2 : x3;
                                    const uint32_t k = x0 ? 0u : x1 ? 4u : x2 ? 8u : 12u;
                                    l = k + ((31u - mz_clz32(x & (0u - x))) >> 3);
                                }
                            } else {
                                l = mz_match_len(in + pos, in + (pos - d), maxl);
                            }
                            if (l >= MZ_DEF_MINMATCH && l > mlen) {
                                mlen = l;
                                dist = d;
                            }
                        }
                    }
                }
                P(pk) = mlen | ((mlen ? dist : 0u) << 9);
                P(lit) = (P(hh) != 0xFFFFFFFFu) ? (P(own) & 0xFFu) : (uint32_t)in[pos < blk_end ? pos : blk];
            }
            if (ways > 1u) {
                /* a match of length L at position q is a match of length L - k at q + k, same distance: a position whose own
                 * bucket had lost that candidate inherits it from the lanes 1 and 2 (then up to 3) below -- worth 1.2 % of
                 * the output under the lazy rule (tests/study/enc_parse.c INH=3) */
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
            if (parse) {
                MZ_LANES {
                    if ((uint32_t)lane < nv) tok[(p - blk) + (uint32_t)lane] = P(pk); /* the cost parse looks at every position's match */
                }
            }
            MZ_DPROF_MARK(18); /* match measurement */
            /* lazy evaluation (what zlib does from level 4 up): a match yields to a longer one starting at the
             * next position -- this position then goes out as a literal */
            PV(uint32_t, pkn);
            PV(uint32_t, pkn2);
            MZ_GATHER4(pkn, pk, 4u * ((uint32_t)lane + 1u));
            MZ_GATHER4(pkn2, pk, 4u * ((uint32_t)lane + 2u));
            MZ_LANES {
                uint32_t mlen = P(pk) & 511u;
                if (mlen && (uint32_t)lane + 1u < nv && (P(pkn) & 511u) > mlen) mlen = 0u;
                /* the default class also looks two positions ahead (two literals must buy more than one byte) */
                if (ways > 1u && mlen && (uint32_t)lane + 2u < nv && (P(pkn2) & 511u) > mlen + 1u) mlen = 0u;
                P(pk) = mlen ? P(pk) : (P(lit) << 9);
                const uint32_t nx = (uint32_t)lane + (mlen ? mlen : 1u);
                P(g1) = ((uint32_t)lane >= nv || nx >= nv) ? (0x1000u | (4u * nx)) : (4u * nx);
            }
#define MZ_DEF_STORE_TOKENS (!parse)
#include "deflate_select.inc"
#undef MZ_DEF_STORE_TOKENS
            MZ_DPROF_MARK(20); /* tokens out, histograms */
            MZ_CRC_FOLD_TILES(crc_acc, crc_done, cin, (p + nv) - warm, crc_tab, tabs->kx);
            MZ_DPROF_MARK(21); /* CRC of the input */
        }
        MZ_DPRIO_AT(28);
#undef MZ_DEF_LOOKUP
        if (parse && blk_end > blk) {
            /* ================= cost parse (see the comment above mz_deflate_piece) ================= */
            const uint32_t n = blk_end - blk;
            MZ_LANES { L->freq[256] = 1u; }
            MZ_WAVE_SYNC();
            mz_huff_build(L, 0u, MZ_DEF_NLIT, 15u);
            mz_huff_build(L, MZ_DEF_DIST0, MZ_DEF_NDIST, 15u);
            /* the hash ways are dead: per chain a ring of the cheapest cost from the next 512 positions to the end, and the
             * price of every match length (symbol + extra bits); a symbol the lazy parse never used costs a default */
            MZ_DPROF_MARK(24); /* cost parse: the price list (two code constructions) */
            /* Four chains: the block is cut into four runs of SB positions and the programme runs over all four at once,
             * each as if the block ended behind its run (a match may reach over the end of a run: what it reaches costs
             * nothing -- the choice is a little too fond of such matches for the last dozen positions of three runs, 0.0x %
             * of the output).  One chain is a string of dependent cross-lane and scalar steps with ONE wave per SIMD to
             * hide them (this class runs one workgroup per CU): ~850 cycles per group of four positions; four
             * independent strings in the same instruction stream fill each other's gaps. */
            uint32_t *const cst0 = (uint32_t *)xhead, *const cst1 = cst0 + 512, *const cst2 = cst0 + 1024, *const cst3 = cst0 + 1536;
            uint8_t *const lcost = (uint8_t *)(cst0 + 2048);
            const uint8_t *const lens = L->u.hb.lens;
            const uint32_t nch = (n >= 4096u) ? 4u : 1u;
            const uint32_t SB = (nch == 4u) ? (((n + 255u) >> 8) << 6) : (((n + 63u) >> 6) << 6); /* a multiple of 64; nch * SB >= n */
            MZ_LANES {
                for (uint32_t l = 3u + (uint32_t)lane; l <= MZ_DEF_MAXMATCH; l += 64u) {
                    uint32_t ex, xv;
                    const uint32_t c = lens[mz_len_sym(l, &ex, &xv)];
                    lcost[l] = (uint8_t)((c ? c : 13u) + ex);
                }
                for (uint32_t i = (uint32_t)lane; i < 2048u; i += 64u) cst0[i] = 0u; /* (behind the end of a run everything costs nothing) */
            }
            MZ_WAVE_SYNC();
            PV(uint32_t, lcf);  /* the prices of this lane's first two lengths to try, 4 + lane % 16 and 16 more */
            PV(uint32_t, lcf2);
            MZ_LANES {
                P(lcf) = lcost[4u + ((uint32_t)lane & 15u)];
                P(lcf2) = lcost[20u + ((uint32_t)lane & 15u)];
            }
            uint32_t cn0 = 0, cn1 = 0, cn2 = 0, cn3 = 0; /* the cost from the position behind the group: the previous group's first result */
#define MZ_DP_DECL(c)                                                                        \
            PV(uint32_t, pkb##c); /* this lane's position: its match, */                     \
            PV(uint32_t, lcb##c); /* the price of its literal, */                            \
            PV(uint32_t, dcb##c); /* of its distance, */                                     \
            PV(uint32_t, chb##c); /* and what the programme chooses for it (0 = the literal) */ \
            PV(uint32_t, gp##c);                                                             \
            PV(uint32_t, gd##c);                                                             \
            PV(uint32_t, best##c);
#define MZ_DP_LOAD(c)                                                                        \
            MZ_LANES {                                                                       \
                const uint32_t r = (c) * SB + (uint32_t)R + (uint32_t)lane;                  \
                uint32_t t = 0, lc = 0, dc = 0;                                              \
                if ((c) < nch && r < n) {                                                    \
                    t = tok[r];                                                              \
                    const uint32_t c_ = lens[in[blk + r]];                                   \
                    lc = c_ ? c_ : 13u;                                                      \
                    if (t & 511u) {                                                          \
                        uint32_t ex, xv;                                                     \
                        const uint32_t cd = lens[MZ_DEF_DIST0 + mz_dist_sym(t >> 9, &ex, &xv)]; \
                        dc = (cd ? cd : 10u) + ex;                                           \
                    }                                                                        \
                }                                                                            \
                P(pkb##c) = t;                                                               \
                P(lcb##c) = lc;                                                              \
                P(dcb##c) = dc;                                                              \
                P(chb##c) = 0u;                                                              \
            }
            /* 16 lanes try the lengths of one position each: 4 .. 35 in one go (the ring is read while the position's match
             * is still on its way across the lanes), the rest in a loop that few groups enter */
#define MZ_DP_EVAL(c)                                                                        \
            MZ_GATHER4(gp##c, pkb##c, 4u * (4u * (uint32_t)gl + ((uint32_t)lane >> 4)));    \
            MZ_GATHER4(gd##c, dcb##c, 4u * (4u * (uint32_t)gl + ((uint32_t)lane >> 4)));    \
            MZ_LANES {                                                                       \
                const uint32_t r = (c) * SB + (uint32_t)R + 4u * (uint32_t)gl + ((uint32_t)lane >> 4); \
                const uint32_t l = 4u + ((uint32_t)lane & 15u);                              \
                const uint32_t cfa = cst##c[(r + l) & 511u], cfb = cst##c[(r + l + 16u) & 511u]; \
                const uint32_t mlen = P(gp##c) & 511u;                                       \
                const uint32_t va = (l <= mlen) ? (((P(lcf) + P(gd##c) + cfa) << 9) | l) : 0xFFFFFFFFu; \
                const uint32_t vb = (l + 16u <= mlen) ? (((P(lcf2) + P(gd##c) + cfb) << 9) | (l + 16u)) : 0xFFFFFFFFu; \
                P(best##c) = va < vb ? va : vb;                                              \
                P(lng) |= (mlen > 35u) ? 1u : 0u;                                            \
            }
#define MZ_DP_TAIL(c)                                                                        \
            MZ_LANES {                                                                       \
                const uint32_t r = (c) * SB + (uint32_t)R + 4u * (uint32_t)gl + ((uint32_t)lane >> 4); \
                const uint32_t mlen = P(gp##c) & 511u;                                       \
                uint32_t b = P(best##c);                                                     \
                for (uint32_t l = 36u + ((uint32_t)lane & 15u); l <= mlen; l += 16u) {       \
                    const uint32_t v = (((uint32_t)lcost[l] + P(gd##c) + cst##c[(r + l) & 511u]) << 9) | l; \
                    b = v < b ? v : b;                                                       \
                }                                                                            \
                P(best##c) = b;                                                              \
            }
#define MZ_DP_MIN(c)                                                                         \
            MZ_ROW16_PMIN(best##c, best##c);                                                 \
            m0_##c = MZ_READLANE(best##c, 15);                                               \
            m1_##c = MZ_READLANE(best##c, 31);                                               \
            m2_##c = MZ_READLANE(best##c, 47);                                               \
            m3_##c = MZ_READLANE(best##c, 63);
            /* the literal steps are four scalar additions per chain; a tie goes to the match: fewer tokens */
#define MZ_DP_STEP(c)                                                                        \
            {                                                                                \
                const uint32_t l0 = MZ_READLANE(lcb##c, 4 * gl), l1 = MZ_READLANE(lcb##c, 4 * gl + 1),    \
                               l2 = MZ_READLANE(lcb##c, 4 * gl + 2), l3 = MZ_READLANE(lcb##c, 4 * gl + 3); \
                const uint32_t t3 = cn##c + l3, k3 = (m3_##c >> 9) <= t3 ? 1u : 0u, c3 = k3 ? (m3_##c >> 9) : t3; \
                const uint32_t t2 = c3 + l2, k2 = (m2_##c >> 9) <= t2 ? 1u : 0u, c2 = k2 ? (m2_##c >> 9) : t2;    \
                const uint32_t t1 = c2 + l1, k1 = (m1_##c >> 9) <= t1 ? 1u : 0u, c1 = k1 ? (m1_##c >> 9) : t1;    \
                const uint32_t t0 = c1 + l0, k0 = (m0_##c >> 9) <= t0 ? 1u : 0u, c0 = k0 ? (m0_##c >> 9) : t0;    \
                cn##c = c0;                                                                  \
                MZ_LANES {                                                                   \
                    if (lane < 4) cst##c[((c) * SB + (uint32_t)R + 4u * (uint32_t)gl + (uint32_t)lane) & 511u] = lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : c3; \
                    if (lane == 4 * gl) P(chb##c) = k0 ? (m0_##c & 511u) : 0u;               \
                    if (lane == 4 * gl + 1) P(chb##c) = k1 ? (m1_##c & 511u) : 0u;           \
                    if (lane == 4 * gl + 2) P(chb##c) = k2 ? (m2_##c & 511u) : 0u;           \
                    if (lane == 4 * gl + 3) P(chb##c) = k3 ? (m3_##c & 511u) : 0u;           \
                }                                                                            \
            }
#define MZ_DP_STORE(c)                                                                       \
            MZ_LANES {                                                                       \
                const uint32_t r = (c) * SB + (uint32_t)R + (uint32_t)lane;                  \
                if ((c) < nch && r < n) tok[r] = P(chb##c) ? (P(chb##c) | (P(pkb##c) & ~511u)) : 0u; \
            }
            for (int32_t R = (int32_t)SB - 64; R >= 0; R -= 64) { /* (R counts inside a run: chain c is at c * SB + R) */
                MZ_DP_DECL(0) MZ_DP_DECL(1) MZ_DP_DECL(2) MZ_DP_DECL(3)
                MZ_DP_LOAD(0) MZ_DP_LOAD(1) MZ_DP_LOAD(2) MZ_DP_LOAD(3)
                uint64_t hasm;
                MZ_BALLOT(hasm, ((P(pkb0) | P(pkb1) | P(pkb2) | P(pkb3)) & 511u) != 0u);
                MZ_DPROF_MARK(27); /* cost parse: 4 x 64 positions fetched */
                /* four positions of every chain at a time, last first: a match is at least 4 long, so what the four may
                 * jump to is settled */
                for (int32_t gl = 15; gl >= 0; gl--) {
                    uint32_t m0_0 = 0xFFFFFFFFu, m1_0 = 0xFFFFFFFFu, m2_0 = 0xFFFFFFFFu, m3_0 = 0xFFFFFFFFu;
                    uint32_t m0_1 = 0xFFFFFFFFu, m1_1 = 0xFFFFFFFFu, m2_1 = 0xFFFFFFFFu, m3_1 = 0xFFFFFFFFu;
                    uint32_t m0_2 = 0xFFFFFFFFu, m1_2 = 0xFFFFFFFFu, m2_2 = 0xFFFFFFFFu, m3_2 = 0xFFFFFFFFu;
                    uint32_t m0_3 = 0xFFFFFFFFu, m1_3 = 0xFFFFFFFFu, m2_3 = 0xFFFFFFFFu, m3_3 = 0xFFFFFFFFu;
                    if ((hasm >> (4 * gl)) & 15ull) {
                        PV(uint32_t, lng);
                        MZ_LANES { P(lng) = 0u; }
                        MZ_DP_EVAL(0) MZ_DP_EVAL(1) MZ_DP_EVAL(2) MZ_DP_EVAL(3)
                        uint64_t longm;
                        MZ_BALLOT(longm, P(lng) != 0u);
                        if (longm) {
                            MZ_DP_TAIL(0) MZ_DP_TAIL(1) MZ_DP_TAIL(2) MZ_DP_TAIL(3)
                        }
                        MZ_DP_MIN(0) MZ_DP_MIN(1) MZ_DP_MIN(2) MZ_DP_MIN(3)
                    }
                    MZ_DP_STEP(0) MZ_DP_STEP(1) MZ_DP_STEP(2) MZ_DP_STEP(3)
                    MZ_WAVE_SYNC();
                }
                MZ_DP_STORE(0) MZ_DP_STORE(1) MZ_DP_STORE(2) MZ_DP_STORE(3)
                MZ_DPROF_MARK(25); /* cost parse: the dynamic programme */
            }
#undef MZ_DP_DECL
#undef MZ_DP_LOAD
#undef MZ_DP_EVAL
#undef MZ_DP_TAIL
#undef MZ_DP_MIN
#undef MZ_DP_STEP
#undef MZ_DP_STORE
            /* the choices, front to back: the same selection as in pass 1, this time the tokens are kept (token i never
             * lands behind position i, so the list grows over the per-position words it has already read) */
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < MZ_DEF_NLIT + MZ_DEF_NDIST; i += 64u) L->freq[i] = 0u;
                P(xbits) = 0u;
            }
            MZ_WAVE_SYNC();
            ntokens = 0;
            skip = 0;
            for (uint32_t p = blk; p < blk_end; p += 64u) {
                const uint32_t nv = (blk_end - p < 64u) ? (blk_end - p) : 64u;
                PV(uint32_t, pk);
                PV(uint32_t, g1);
                MZ_LANES {
                    uint32_t t = 0;
                    if ((uint32_t)lane < nv) t = tok[(p - blk) + (uint32_t)lane];
                    const uint32_t mlen = t & 511u;
                    P(pk) = mlen ? t : ((uint32_t)in[(uint32_t)lane < nv ? p + (uint32_t)lane : blk] << 9);
                    const uint32_t nx = (uint32_t)lane + (mlen ? mlen : 1u);
                    P(g1) = ((uint32_t)lane >= nv || nx >= nv) ? (0x1000u | (4u * nx)) : (4u * nx);
                }
                MZ_WAVE_SYNC();
#define MZ_DEF_STORE_TOKENS 1
#include "deflate_select.inc"
#undef MZ_DEF_STORE_TOKENS
                MZ_DPROF_MARK(26); /* cost parse: the choices picked up (the selection of this pass is counted there, not under 19 / 20) */
            }
        }
        MZ_LANES { L->freq[256] = 1u; } /* end of block */
        MZ_WAVE_SYNC();
        uint32_t extra_total;
        MZ_WAVE_SUM(extra_total, xbits);

        /* ================= codes and the three block costs ================= */
        mz_huff_build(L, 0u, MZ_DEF_NLIT, 15u);
        mz_huff_build(L, MZ_DEF_DIST0, MZ_DEF_NDIST, 15u);
        uint32_t nlit = MZ_DEF_NLIT, ndist = MZ_DEF_NDIST;
        while (nlit > 257u && MZ_UNIFORM(L->u.hb.lens[nlit - 1u]) == 0u) nlit--;
        while (ndist > 1u && MZ_UNIFORM(L->u.hb.lens[MZ_DEF_DIST0 + ndist - 1u]) == 0u) ndist--;
        /* code-length sequence with run-length symbols 16 / 17 / 18 (appnote.txt:2070-2090), wave-uniform */
        uint32_t nseq = 0;
        {
            const uint32_t total = nlit + ndist;
#define MZ_DEF_LEN_AT(i) MZ_UNIFORM(L->u.hb.lens[(i) < nlit ? (i) : MZ_DEF_DIST0 + (i) - nlit])
#define MZ_DEF_SEQ(s, x)                                                                   \
    do {                                                                                   \
        const uint32_t _f = MZ_UNIFORM(L->freq[MZ_DEF_CL0 + (s)]) + 1u;                    \
        MZ_LANES {                                                                         \
            L->u.hb.cl_seq[nseq] = (uint8_t)(s);                                           \
            L->u.hb.cl_ext[nseq] = (uint8_t)(x);                                           \
            L->freq[MZ_DEF_CL0 + (s)] = _f;                                                \
        }                                                                                  \
        MZ_WAVE_SYNC();                                                                    \
        nseq++;                                                                            \
    } while (0)
            uint32_t i = 0;
            while (i < total) {
                const uint32_t v = MZ_DEF_LEN_AT(i);
                uint32_t run = 1;
                while (i + run < total && MZ_DEF_LEN_AT(i + run) == v) run++;
                i += run;
                if (v == 0u) {
                    while (run >= 11u) {
                        const uint32_t r = run > 138u ? 138u : run;
                        MZ_DEF_SEQ(18u, r - 11u);
                        run -= r;
                    }
                    if (run >= 3u) {
                        MZ_DEF_SEQ(17u, run - 3u);
                        run = 0;
                    }
                } else {
                    MZ_DEF_SEQ(v, 0u);
                    run--;
                    while (run >= 3u) {
                        const uint32_t r = run > 6u ? 6u : run;
                        MZ_DEF_SEQ(16u, r - 3u);
                        run -= r;
                    }
                }
                while (run) {
                    MZ_DEF_SEQ(v, 0u);
                    run--;
                }
            }
#undef MZ_DEF_SEQ
#undef MZ_DEF_LEN_AT
        }
        /* cl_seq / cl_ext are not among the u.hb members mz_huff_build works in, so they survive this call */
        mz_huff_build(L, MZ_DEF_CL0, MZ_DEF_NCL, 7u);
        uint32_t hclen = 19u;
        PV(uint32_t, cost_d);
        PV(uint32_t, cost_f);
        MZ_LANES {
            uint32_t cd = 0, cf = 0;
            for (uint32_t s = (uint32_t)lane; s < MZ_DEF_NLIT + MZ_DEF_NDIST; s += 64u) {
                const uint32_t f = L->freq[s];
                cd += f * L->u.hb.lens[s];
                cf += f * (s < MZ_DEF_NLIT ? mz_fixed_litlen_bits(s) : 5u);
            }
            for (uint32_t s = (uint32_t)lane; s < MZ_DEF_NCL; s += 64u)
                cd += L->freq[MZ_DEF_CL0 + s] * (L->u.hb.lens[MZ_DEF_CL0 + s] + (s == 16u ? 2u : s == 17u ? 3u : s == 18u ? 7u : 0u));
            P(cost_d) = cd;
            P(cost_f) = cf;
        }
        uint32_t dyn_bits, fix_bits;
        MZ_WAVE_SUM(dyn_bits, cost_d);
        MZ_WAVE_SUM(fix_bits, cost_f);
        {
            /* HCLEN: code-length code lengths are sent in the order 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15 */
            while (hclen > 4u && MZ_UNIFORM(L->u.hb.lens[MZ_DEF_CL0 + mz_cl_order(hclen - 1u)]) == 0u) hclen--;
            dyn_bits += 3u + 5u + 5u + 4u + 3u * hclen + extra_total;
            fix_bits += 3u + extra_total;
            const uint32_t blk_len = blk_end - blk;
            const uint32_t stored_bits = 8u * blk_len + 40u * ((blk_len + 65534u) / 65535u) + 7u;
            if (blk_len != 0u && stored_bits < dyn_bits && stored_bits < fix_bits) {
                /* ---- stored blocks: header, pad to a byte, LEN, NLEN, raw bytes (appnote.txt:2045-2049) ---- */
                for (uint32_t s0 = blk; s0 < blk_end; s0 += 65535u) {
                    const uint32_t n = (blk_end - s0 < 65535u) ? (blk_end - s0) : 65535u;
                    MZ_PUTBITS((bfinal && s0 + n == blk_end) ? 1u : 0u, 3u);
                    if (nacc) MZ_PUTBITS(0u, 8u - nacc);
                    MZ_PUTBITS(n | ((n ^ 0xFFFFu) << 16), 32u);
                    if (n > out_cap - obyte) {
                        status = MZHIP_OUT_FULL;
                        goto finish;
                    }
                    MZ_LANES {
                        for (uint32_t i = (uint32_t)lane; i < n; i += 64u) out[obyte + i] = in[s0 + i];
                    }
                    MZ_WAVE_SYNC();
                    obyte += n;
                }
                continue;
            }
            if (dyn_bits < fix_bits) {
                /* ---- dynamic block header (appnote.txt:2064-2090) ---- */
                MZ_PUTBITS(bfinal | (2u << 1), 3u);
                MZ_PUTBITS(nlit - 257u, 5u);
                MZ_PUTBITS(ndist - 1u, 5u);
                MZ_PUTBITS(hclen - 4u, 4u);
                for (uint32_t i = 0; i < hclen; i++) MZ_PUTBITS(MZ_UNIFORM(L->u.hb.lens[MZ_DEF_CL0 + mz_cl_order(i)]), 3u);
                for (uint32_t i = 0; i < nseq; i++) {
                    const uint32_t s = MZ_UNIFORM(L->u.hb.cl_seq[i]), c = MZ_UNIFORM(L->u.hb.code[MZ_DEF_CL0 + s]);
                    MZ_PUTBITS(c & 0xFFFFu, c >> 16);
                    if (s >= 16u) MZ_PUTBITS(MZ_UNIFORM(L->u.hb.cl_ext[i]), s == 16u ? 2u : s == 17u ? 3u : 7u);
                }
            } else {
                /* ---- fixed block: same pass 2 with the fixed code in the table (appnote.txt:2050-2059) ---- */
                MZ_PUTBITS(bfinal | (1u << 1), 3u);
                MZ_LANES {
                    for (uint32_t s = (uint32_t)lane; s < MZ_DEF_NLIT + MZ_DEF_NDIST; s += 64u) {
                        uint32_t c, n;
                        if (s >= MZ_DEF_NLIT) {
                            c = s - MZ_DEF_NLIT;
                            n = 5;
                        } else if (s < 144u) {
                            c = 0x30u + s;
                            n = 8;
                        } else if (s < 256u) {
                            c = 0x190u + (s - 144u);
                            n = 9;
                        } else if (s < 280u) {
                            c = s - 256u;
                            n = 7;
                        } else {
                            c = 0xC0u + (s - 280u);
                            n = 8;
                        }
                        L->u.hb.code[s] = (mz_brev32(c) >> (32u - n)) | (n << 16);
                    }
                }
                MZ_WAVE_SYNC();
            }
        }
        MZ_DPROF_MARK(22); /* codes, block costs, header */
        /* ================= pass 2: the tokens' bits ================= */
        for (uint32_t t0 = 0; t0 < ntokens; t0 += 64u) {
            const uint32_t nt = (ntokens - t0 < 64u) ? (ntokens - t0) : 64u;
            PV(uint32_t, blo);
            PV(uint32_t, bhi);
            PV(uint32_t, tn);
            PV(uint32_t, tend);
            MZ_LANES {
                uint64_t b = 0;
                uint32_t n = 0;
                if ((uint32_t)lane < nt) {
                    const uint32_t t = tok[t0 + (uint32_t)lane], mlen = t & 511u;
                    if (mlen == 0u) {
                        const uint32_t c = L->u.hb.code[t >> 9];
                        b = c & 0xFFFFu;
                        n = c >> 16;
                    } else {
                        uint32_t ex, xv;
                        const uint32_t c = L->u.hb.code[mz_len_sym(mlen, &ex, &xv)];
                        b = (c & 0xFFFFu) | ((uint64_t)xv << (c >> 16));
                        n = (c >> 16) + ex;
                        const uint32_t cd = L->u.hb.code[MZ_DEF_DIST0 + mz_dist_sym(t >> 9, &ex, &xv)];
                        b |= (uint64_t)((cd & 0xFFFFu) | (xv << (cd >> 16))) << n;
                        n += (cd >> 16) + ex; /* <= 15 + 5 + 15 + 13 = 48 */
                    }
                }
                P(blo) = (uint32_t)b;
                P(bhi) = (uint32_t)(b >> 32);
                P(tn) = n;
            }
            MZ_INCL_SCAN(tend, tn);
            const uint32_t total = MZ_READLANE(tend, 63);
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < 104u; i += 64u) L->u.hb.stage[i] = (i == 0u) ? (uint32_t)acc : 0u;
            }
            MZ_WAVE_SYNC();
            MZ_LANES {
                if (P(tn)) {
                    const uint32_t off = nacc + P(tend) - P(tn);
                    const uint32_t w = off >> 5, sh = off & 31u;
                    const uint64_t b = ((uint64_t)P(bhi) << 32) | P(blo);
                    const uint64_t v0 = b << sh;
                    MZ_LDS_ATOMIC_OR(&L->u.hb.stage[w], (uint32_t)v0);
                    if (sh + P(tn) > 32u) MZ_LDS_ATOMIC_OR(&L->u.hb.stage[w + 1u], (uint32_t)(v0 >> 32));
                    if (sh + P(tn) > 64u) MZ_LDS_ATOMIC_OR(&L->u.hb.stage[w + 2u], (uint32_t)(b >> (64u - sh)));
                }
            }
            MZ_WAVE_SYNC();
            {
                const uint32_t nbit = nacc + total, nbytes = nbit >> 3;
                if (nbytes > out_cap - obyte) {
                    status = MZHIP_OUT_FULL;
                    goto finish;
                }
                MZ_LANES {
                    for (uint32_t i = (uint32_t)lane; i < nbytes; i += 64u)
                        out[obyte + i] = (uint8_t)(L->u.hb.stage[i >> 2] >> (8u * (i & 3u)));
                }
                acc = (nbit & 7u) ? ((MZ_UNIFORM(L->u.hb.stage[nbytes >> 2]) >> (8u * (nbytes & 3u))) & ((1u << (nbit & 7u)) - 1u)) : 0u;
                nacc = nbit & 7u;
                obyte += nbytes;
            }
            MZ_WAVE_SYNC();
        }
        {
            const uint32_t c = MZ_UNIFORM(L->u.hb.code[256]); /* end of block */
            MZ_PUTBITS(c & 0xFFFFu, c >> 16);
        }
    }

    /* pad the final block to a byte, or append an empty stored block so that pieces concatenate on bytes */
    if (final) {
        if (nacc) MZ_PUTBITS(0u, 8u - nacc);
    } else {
        MZ_PUTBITS(0u, 3u); /* BFINAL = 0, BTYPE = 00 */
        if (nacc) MZ_PUTBITS(0u, 8u - nacc);
        MZ_PUTBITS(0xFFFF0000u, 32u); /* LEN = 0x0000, NLEN = 0xFFFF (appnote.txt:2045-2049) */
    }

finish:
    MZ_DPROF_MARK(23); /* pass 2: the tokens' bits */
    res->status = status;
    res->out_len = obyte;
    {
        uint32_t crc;
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, cin, in_len - warm, crc_tab, tabs->kx);
        MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, cin, in_len - warm, crc_tab, tabs);
        res->crc = crc;
    }
    MZ_DPROF_FLUSH
}

#endif
