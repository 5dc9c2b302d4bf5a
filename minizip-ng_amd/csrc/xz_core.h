/* xz_core.h -- .xz decode (ZIP method 95) of ONE entry by ONE wavefront, CRC-32 of the output fused.
 *
 * Replaces what the reference does for a method-95 entry through mz_stream_lzma_read (mz_strm_lzma.c:127-128,
 * 147-241 -> liblzma lzma_stream_decoder(flags 0) / lzma_code): one stream -- stream header, blocks (header,
 * LZMA2 chunks, padding, check), index, stream footer ("The .xz File Format" 1.0.4) -- no concatenation, no
 * stream padding.  The LZMA2 layer re-initialises the range coder for every chunk and carries the dictionary
 * (the output buffer itself), the probability model (LDS) and the state / rep registers across chunks; the
 * packet loop is the one K3 uses (lzma_core.h, LZ_PACKET_LOOP with lzma2 = 1).  Checks verified on the device:
 * none, CRC32, CRC64 (wave-parallel, hash_core.h), SHA-256 (the wave runs the chain redundantly); other check
 * ids are skipped unverified like lzma_stream_decoder does without LZMA_TELL_UNSUPPORTED_CHECK.
 * Framing is parsed by wave-uniform code (it is a few dozen bytes per block).  Filter chains: LZMA2 last, behind it
 * up to three of Delta and the BCJ filters x86 / PowerPC / IA-64 / ARM / ARM-Thumb / SPARC, what liblzma 5.2.5's
 * lzma_stream_decoder accepts (filter_common.c); a block's bytes are unfiltered in place when the block is complete,
 * 8 KiB at a time through the LDS of the probability model (dead between blocks), by wave-uniform byte code -- a BCJ
 * scan is a serial state machine, and filtered entries are rare.  lc + lp = 4: the upper half of the literal model
 * lives in prx[] (HBM).
 * Every framing / check / LZMA2 failure is MZHIP_DATA_ERROR (mz_stream_lzma_read maps all liblzma errors to
 * MZ_DATA_ERROR, mz_strm_lzma.c:236-237); input that ends early is MZHIP_BUF_ERROR.
 */
#ifndef MZHIP_XZ_CORE_H
#define MZHIP_XZ_CORE_H

#include "hash_core.h"
#include "lzma_core.h"

typedef struct mz_xz_lds {
    mz_lzma_lds lz;
    uint64_t crc64_tab[256];
} mz_xz_lds;

/* CRC-32 of a short byte range with the wave-parallel folding of crc32_core.h */
#define XZ_CRC32_RANGE(result, buf, n)                                                   \
    do {                                                                                 \
        PV(uint32_t, _xa);                                                               \
        PV(uint32_t, _xt);                                                               \
        uint32_t _xd = 0;                                                                \
        MZ_LANES { P(_xa) = (lane == 0) ? 0xFFFFFFFFu : 0u; }                            \
        MZ_CRC_FOLD_TILES(_xa, _xd, buf, n, crc_tab, tabs->kx);                          \
        MZ_CRC_FINISH(result, _xa, _xt, _xd, buf, n, crc_tab, tabs);                     \
    } while (0)

#define XZ_BYTE(p) MZ_UNIFORM(in[(p)])
#define XZ_LE32(p) (XZ_BYTE(p) | (XZ_BYTE((p) + 1) << 8) | (XZ_BYTE((p) + 2) << 16) | (XZ_BYTE((p) + 3) << 24))
/* bytes [pos, pos + k) must exist */
#define XZ_NEED(k)                                                                       \
    do {                                                                                 \
        if ((uint64_t)in_len - pos < (uint64_t)(k)) {                                    \
            pos = in_len;                                                                \
            status = MZHIP_BUF_ERROR;                                                    \
            goto finish;                                                                 \
        }                                                                                \
    } while (0)
/* variable-length integer at `p`, limited to `lim` (exclusive); short = what running off `lim` means */
#define XZ_VLI(dst, p, lim, short_status)                                                \
    do {                                                                                 \
        uint64_t _v = 0;                                                                 \
        uint32_t _i = 0;                                                                 \
        for (;;) {                                                                       \
            if (_i == 9) goto finish;                                                    \
            if ((p) >= (lim)) {                                                          \
                status = (short_status);                                                 \
                if ((short_status) == MZHIP_BUF_ERROR) pos = in_len;                     \
                goto finish;                                                             \
            }                                                                            \
            const uint32_t _b = XZ_BYTE(p);                                              \
            (p)++;                                                                       \
            if (_b == 0 && _i != 0) goto finish;                                         \
            _v |= (uint64_t)(_b & 0x7Fu) << (7 * _i);                                    \
            _i++;                                                                        \
            if (!(_b & 0x80u)) break;                                                    \
        }                                                                                \
        (dst) = _v;                                                                      \
    } while (0)

/* ---- Delta / BCJ, decoding direction (liblzma 5.2.5 delta_decoder.c, simple/{x86,powerpc,ia64,arm,armthumb,sparc}.c;
 * oracle/xz_dec.c restates them and is pinned against liblzma).  Streaming form: b[0 .. n) = bytes not yet final, now =
 * their position in the block (+ the filter's start offset); returns how many leading bytes are final now, the rest
 * is presented again with more data behind it.  All values wave-uniform (every lane runs the same bytes). ---- */
#define XZ_FB(i) MZ_UNIFORM(b[(i)])
#define XZ_FS(i, v) do { b[(i)] = (uint8_t)(v); } while (0)
typedef struct mz_xz_filter {
    uint32_t id, arg;             /* 3 = Delta (arg = distance), 4..9 = BCJ (arg = start offset) */
    uint32_t prev_mask, prev_pos; /* x86 */
    uint32_t hpos;                /* Delta: ring position */
} mz_xz_filter;
MZ_DEV uint32_t mz_xz_msb86(uint32_t c) { return (c == 0u || c == 0xFFu) ? 1u : 0u; }
MZ_DEV uint32_t mz_xz_unfilter(mz_xz_filter *f, uint8_t *b, uint32_t n, uint32_t now, uint8_t *hist) {
    if (f->id == 3u) { /* out[i] = in[i] + out[i - distance]; 256-byte ring like liblzma's */
        uint32_t hp = f->hpos;
        for (uint32_t i = 0; i < n; i++) {
            const uint32_t v = (XZ_FB(i) + MZ_UNIFORM(hist[(f->arg + hp) & 0xFFu])) & 0xFFu;
            XZ_FS(i, v);
            hist[hp & 0xFFu] = (uint8_t)v;
            hp--;
        }
        f->hpos = hp;
        return n;
    }
    if (f->id == 4u) {
        uint32_t prev_mask = f->prev_mask, prev_pos = f->prev_pos;
        if (n < 5u) return 0u;
        if (now - prev_pos > 5u) prev_pos = now - 5u;
        const uint32_t limit = n - 5u;
        uint32_t i = 0;
        while (i <= limit) {
            uint32_t c = XZ_FB(i);
            if (c != 0xE8u && c != 0xE9u) {
                i++;
                continue;
            }
            const uint32_t off = now + i - prev_pos;
            prev_pos = now + i;
            if (off > 5u) {
                prev_mask = 0;
            } else {
                for (uint32_t k = 0; k < off; k++) prev_mask = (prev_mask & 0x77u) << 1;
            }
            c = XZ_FB(i + 4u);
            const uint32_t pm = prev_mask >> 1;
            /* MASK_TO_ALLOWED_STATUS = {1,1,1,0,1,0,0,0}, MASK_TO_BIT_NUMBER = {0,1,2,2,3,3,3,3} */
            if (mz_xz_msb86(c) && ((0x17u >> (pm & 7u)) & 1u) && pm < 0x10u) {
                uint32_t src = (c << 24) | (XZ_FB(i + 3u) << 16) | (XZ_FB(i + 2u) << 8) | XZ_FB(i + 1u), dest;
                for (;;) {
                    dest = src - (now + i + 5u);
                    if (prev_mask == 0u) break;
                    const uint32_t k = (0xFFA4u >> (2u * (pm & 7u))) & 3u; /* (pm < 8 here: a set 0x10 has been shifted up at least once) */
                    c = (dest >> (24u - k * 8u)) & 0xFFu;
                    if (!mz_xz_msb86(c)) break;
                    src = dest ^ ((1u << (32u - k * 8u)) - 1u);
                }
                XZ_FS(i + 4u, ~(((dest >> 24) & 1u) - 1u));
                XZ_FS(i + 3u, dest >> 16);
                XZ_FS(i + 2u, dest >> 8);
                XZ_FS(i + 1u, dest);
                i += 5u;
                prev_mask = 0;
            } else {
                i++;
                prev_mask |= 1u;
                if (mz_xz_msb86(c)) prev_mask |= 0x10u;
            }
        }
        f->prev_mask = prev_mask;
        f->prev_pos = prev_pos;
        return i;
    }
    if (f->id == 6u) { /* IA-64: 128-bit bundles, three 41-bit slots, branch slots by template */
        uint32_t i = 0;
        for (; i + 16u <= n; i += 16u) {
            const uint32_t tmpl = XZ_FB(i) & 0x1Fu;
            /* BRANCH_TABLE: templates 16,17 -> 4; 18,19 -> 6; 22,23 -> 7; 24,25 -> 4; 28,29 -> 4; else 0 */
            const uint32_t mask = (tmpl == 16u || tmpl == 17u || tmpl == 24u || tmpl == 25u || tmpl == 28u || tmpl == 29u) ? 4u
                                  : (tmpl == 18u || tmpl == 19u) ? 6u : (tmpl == 22u || tmpl == 23u) ? 7u : 0u;
            uint32_t bit_pos = 5u;
            for (uint32_t slot = 0; slot < 3u; slot++, bit_pos += 41u) {
                if (((mask >> slot) & 1u) == 0u) continue;
                const uint32_t byte_pos = bit_pos >> 3, bit_res = bit_pos & 7u;
                uint64_t ins = 0;
                for (uint32_t j = 0; j < 6u; j++) ins += (uint64_t)XZ_FB(i + j + byte_pos) << (8u * j);
                uint64_t norm = ins >> bit_res;
                if (((norm >> 37) & 0xFu) == 0x5u && ((norm >> 9) & 0x7u) == 0u) {
                    uint32_t src = (uint32_t)((norm >> 13) & 0xFFFFFu);
                    src |= (uint32_t)((norm >> 36) & 1u) << 20;
                    src <<= 4;
                    uint32_t dest = src - (now + i);
                    dest >>= 4;
                    norm &= ~((uint64_t)0x8FFFFF << 13);
                    norm |= (uint64_t)(dest & 0xFFFFFu) << 13;
                    norm |= (uint64_t)(dest & 0x100000u) << (36 - 20);
                    ins &= ((uint64_t)1 << bit_res) - 1u;
                    ins |= norm << bit_res;
                    for (uint32_t j = 0; j < 6u; j++) XZ_FS(i + j + byte_pos, ins >> (8u * j));
                }
            }
        }
        return i;
    }
    if (f->id == 8u) { /* ARM-Thumb: BL pairs on 2-byte boundaries */
        uint32_t i = 0;
        for (; i + 4u <= n; i += 2u) {
            const uint32_t b1 = XZ_FB(i + 1u), b3 = XZ_FB(i + 3u);
            if ((b1 & 0xF8u) == 0xF0u && (b3 & 0xF8u) == 0xF8u) {
                uint32_t src = ((b1 & 7u) << 19) | (XZ_FB(i) << 11) | ((b3 & 7u) << 8) | XZ_FB(i + 2u);
                src <<= 1;
                uint32_t dest = src - (now + i + 4u);
                dest >>= 1;
                XZ_FS(i + 1u, 0xF0u | ((dest >> 19) & 7u));
                XZ_FS(i, dest >> 11);
                XZ_FS(i + 3u, 0xF8u | ((dest >> 8) & 7u));
                XZ_FS(i + 2u, dest);
                i += 2u;
            }
        }
        return i;
    }
    { /* PowerPC (5), ARM (7), SPARC (9): one 32-bit instruction at a time */
        uint32_t i = 0;
        for (; i + 4u <= n; i += 4u) {
            const uint32_t b0 = XZ_FB(i), b1 = XZ_FB(i + 1u), b2 = XZ_FB(i + 2u), b3 = XZ_FB(i + 3u);
            if (f->id == 5u) {
                if ((b0 >> 2) == 0x12u && (b3 & 3u) == 1u) {
                    const uint32_t src = ((b0 & 3u) << 24) | (b1 << 16) | (b2 << 8) | (b3 & ~3u);
                    const uint32_t dest = src - (now + i);
                    XZ_FS(i, 0x48u | ((dest >> 24) & 3u));
                    XZ_FS(i + 1u, dest >> 16);
                    XZ_FS(i + 2u, dest >> 8);
                    XZ_FS(i + 3u, (b3 & 3u) | (dest & 0xFCu));
                }
            } else if (f->id == 7u) {
                if (b3 == 0xEBu) {
                    uint32_t src = ((b2 << 16) | (b1 << 8) | b0) << 2;
                    const uint32_t dest = (src - (now + i + 8u)) >> 2;
                    XZ_FS(i + 2u, dest >> 16);
                    XZ_FS(i + 1u, dest >> 8);
                    XZ_FS(i, dest);
                }
            } else {
                if ((b0 == 0x40u && (b1 & 0xC0u) == 0x00u) || (b0 == 0x7Fu && (b1 & 0xC0u) == 0xC0u)) {
                    uint32_t src = ((b0 << 24) | (b1 << 16) | (b2 << 8) | b3) << 2;
                    uint32_t dest = (src - (now + i)) >> 2;
                    dest = (((0u - ((dest >> 22) & 1u)) << 22) & 0x3FFFFFFFu) | (dest & 0x3FFFFFu) | 0x40000000u;
                    XZ_FS(i, dest >> 24);
                    XZ_FS(i + 1u, dest >> 16);
                    XZ_FS(i + 2u, dest >> 8);
                    XZ_FS(i + 3u, dest);
                }
            }
        }
        return i;
    }
}
#undef XZ_FB
#undef XZ_FS
#define MZ_XZ_FCHUNK 8192u /* bytes of a block unfiltered per round (+ up to 15 carried over) */

/* Undo the filter chain of one finished block in place (out[0 .. n) = the block's bytes, still filtered), last applied
 * first, 8 KiB at a time: the bytes go through `fb` (the LDS of the probability model: every block re-initialises it with
 * its first chunk), where wave-uniform code runs the filter; bytes the filter cannot decide yet (< 16) are carried into
 * the next round.  Filtered entries are rare (minizip never writes them) and a BCJ scan is a serial state machine: this is
 * about being correct, so it is a call, not inline code -- inlined into mz_xz_entry it cost the hot decode loop 24 VGPRs
 * (161 -> 185) and 36 more parked SGPRs. */
#if defined(MZHIP_HOST_EMUL)
static void
#else
__device__ __noinline__ void
#endif
mz_xz_unfilter_block(uint8_t *blk, uint32_t usize_blk, uint32_t npre, uint32_t fids, uint32_t farg0, uint32_t farg1, uint32_t farg2,
                     uint8_t *fb) {
    MZ_LANE_DECL
    uint8_t *hist = fb + MZ_XZ_FCHUNK + 16u; /* MZ_XZ_FCHUNK + 16 bytes of staging, Delta's 256-byte ring behind it */
    for (uint32_t fi = 3u; fi-- > 0u;) {
        if (fi >= npre) continue;
        mz_xz_filter flt;
        flt.id = (fids >> (8u * fi)) & 0xFFu;
        flt.arg = fi == 0u ? farg0 : fi == 1u ? farg1 : farg2;
        flt.prev_mask = 0;
        flt.prev_pos = 0xFFFFFFFBu; /* (uint32_t)-5: x86 */
        flt.hpos = 0;
        uint32_t done = 0, carry = 0; /* bytes final so far; bytes sitting in fb[0 .. carry) */
        if (flt.id == 3u) {
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < 256u; i += 64u) hist[i] = 0;
            }
            MZ_WAVE_SYNC();
        }
        while (done + carry < usize_blk) {
            uint32_t take = usize_blk - done - carry;
            if (take > MZ_XZ_FCHUNK) take = MZ_XZ_FCHUNK;
            const uint8_t *src = blk + done + carry;
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < take; i += 64u) fb[carry + i] = src[i];
            }
            MZ_WAVE_SYNC();
            const uint32_t have = carry + take;
            const uint32_t fin = mz_xz_unfilter(&flt, fb, have, flt.id == 3u ? 0u : flt.arg + done, hist);
            MZ_WAVE_SYNC();
            uint8_t *dst = blk + done;
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < fin; i += 64u) dst[i] = fb[i];
            }
            MZ_WAVE_SYNC();
            carry = have - fin;
            MZ_LANES { /* (carry < 16: one lane each) */
                if ((uint32_t)lane < carry) fb[lane] = fb[fin + (uint32_t)lane];
            }
            MZ_WAVE_SYNC();
            done += fin;
            if (fin == 0u && take == 0u) break;
        }
        /* what is left in the carry never had enough bytes behind it: it stays as it is */
    }
}


MZ_DEV uint64_t mz_xz_mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xFF51AFD7ED558CCDull;
}

#undef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) ((o) < crc_hold ? (o) : crc_hold) /* a filtered block's bytes are not final until it ends */
/* Decode one method-95 entry.  All arguments wave-uniform.  max_out < 0: no clamp. */
MZ_DEV void mz_xz_entry(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                        mz_xz_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs, uint16_t *prx,
                        mz_lzma_result *res) {
    MZ_LANE_DECL
    uint16_t *pr = L->lz.probs;
    const uint64_t *tab64 = L->crc64_tab;
    int32_t status = MZHIP_DATA_ERROR;
    uint32_t pos = 0;  /* container cursor */
    uint32_t opos = 0; /* output cursor = dictionary position */
    /* range coder over the current chunk */
    const uint8_t *rc_in = in;
    uint32_t rc_len = 0, in_pos = 0, in_base = 0, eof = 0, range = 0xFFFFFFFFu, code = 0;
    const uint32_t lzma2 = 1;
    uint32_t chunk_end = 0, dict_start = 0, rc_short = 0;
    uint64_t dict = 4096;
    uint32_t state = 0, rep0 = 0, rep1 = 0, rep2 = 0, rep3 = 0, prev_byte = 0, match_byte = 0;
    uint32_t lc = 0, pb_mask = 0, lp_mask = 0;
    PV(uint32_t, win);
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    uint32_t crc_done = 0;
    uint32_t crc_hold = 0xFFFFFFFFu; /* the fused CRC folds no further than this (start of a filtered block in progress) */
    MZ_LANES {
        P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u;
        P(win) = 0;
    }

    /* stream header (2.1.1): magic, flags {0x00, check id}, CRC32(flags) */
    XZ_NEED(12);
    if (XZ_BYTE(0) != 0xFD || XZ_BYTE(1) != '7' || XZ_BYTE(2) != 'z' || XZ_BYTE(3) != 'X' || XZ_BYTE(4) != 'Z' ||
        XZ_BYTE(5) != 0 || XZ_BYTE(6) != 0 || (XZ_BYTE(7) & 0xF0u))
        goto finish;
    {
        uint32_t k;
        XZ_CRC32_RANGE(k, in + 6, 2u);
        if (k != XZ_LE32(8)) goto finish;
    }
    {
        const uint32_t check = XZ_BYTE(7);
        const uint32_t check_size = check == 0 ? 0u : check < 4 ? 4u : check < 7 ? 8u : check < 10 ? 16u : check < 13 ? 32u : 64u;
        uint64_t nblocks = 0, blocks_digest = 0;
        pos = 12;

        for (;;) {
            XZ_NEED(1);
            if (XZ_BYTE(pos) == 0) break; /* index indicator */
            /* ---- block header (3.1) ---- */
            const uint32_t hpos = pos, hsize = (XZ_BYTE(pos) + 1u) * 4u;
            XZ_NEED(hsize);
            {
                uint32_t k;
                XZ_CRC32_RANGE(k, in + hpos, hsize - 4u);
                if (k != XZ_LE32(hpos + hsize - 4u)) goto finish;
            }
            const uint32_t bflags = XZ_BYTE(hpos + 1u);
            if (bflags & 0x3Cu) goto finish; /* reserved bits */
            uint32_t p = hpos + 2u;
            const uint32_t hend = hpos + hsize - 4u;
            uint64_t want_csize = ~0ull, want_usize = ~0ull;
            if (bflags & 0x40u) {
                XZ_VLI(want_csize, p, hend, MZHIP_DATA_ERROR);
                if (want_csize == 0) goto finish;
            }
            if (bflags & 0x80u) XZ_VLI(want_usize, p, hend, MZHIP_DATA_ERROR);
            /* filter flags (3.1.5, 5.3): LZMA2 last; in front of it Delta and BCJ filters (filter_common.c: only
             * LZMA2 may be last, only these may stand before it).  Anything else is LZMA_OPTIONS_ERROR = a data error. */
            /* what the header says of them is an id and one argument each; the filters' running state starts from
             * constants when the block is unfiltered (mz_xz_unfilter_block) -- four wave-uniform words live across the block's
             * decode loop instead of the fifteen of three filter structs */
            uint32_t fids = 0, farg0 = 0, farg1 = 0, farg2 = 0, npre = 0; /* id of filter k in bits 8k .. 8k + 7 */
            mz_xz_filter fnew;
#define XZ_PUSH_FILTER()                         \
    do {                                         \
        fids |= fnew.id << (8u * npre);          \
        if (npre == 0u) farg0 = fnew.arg;        \
        else if (npre == 1u) farg1 = fnew.arg;   \
        else farg2 = fnew.arg;                   \
        npre++;                                  \
    } while (0)
            const uint32_t nfilt = (bflags & 3u) + 1u;
            for (uint32_t fi = 0; fi < nfilt; fi++) {
                uint64_t id, psize;
                XZ_VLI(id, p, hend, MZHIP_DATA_ERROR);
                XZ_VLI(psize, p, hend, MZHIP_DATA_ERROR);
                if (psize > hend - p) goto finish;
                if (fi + 1u == nfilt) {
                    if (id != 0x21 || psize != 1) goto finish;
                    const uint32_t db = XZ_BYTE(p);
                    p++;
                    if (db > 40) goto finish;
                    dict = db == 40 ? 0xFFFFFFFFull : (uint64_t)(2u | (db & 1u)) << (db / 2u + 11u);
                    if (dict < 4096) dict = 4096;
                    dict = (dict + 15) & ~(uint64_t)15;
                } else if (id == 0x03) {
                    if (psize != 1) goto finish;
                    fnew.id = 3u;
                    fnew.arg = XZ_BYTE(p) + 1u;
                    fnew.prev_mask = fnew.prev_pos = fnew.hpos = 0;
                    XZ_PUSH_FILTER();
                    p++;
                } else if (id >= 0x04 && id <= 0x09) {
                    uint32_t so = 0;
                    if (psize == 4) so = XZ_LE32(p);
                    else if (psize != 0) goto finish;
                    const uint32_t al = id == 4 ? 1u : id == 6 ? 16u : id == 8 ? 2u : 4u;
                    if (so & (al - 1u)) goto finish; /* misaligned start offset */
                    fnew.id = (uint32_t)id;
                    fnew.arg = so;
                    fnew.prev_mask = 0;
                    fnew.prev_pos = 0xFFFFFFFBu; /* (uint32_t)-5 */
                    fnew.hpos = 0;
                    XZ_PUSH_FILTER();
                    p += (uint32_t)psize;
                } else {
                    goto finish;
                }
            }
            for (; p < hend; p++)
                if (XZ_BYTE(p) != 0) goto finish; /* header padding */
            pos = hpos + hsize;

            /* ---- LZMA2 chunks ---- */
            const uint32_t data_pos = pos, block_out = opos;
            if (npre) crc_hold = block_out;
            uint32_t need_props = 1, need_dict_reset = 1;
            for (;;) {
                XZ_NEED(1);
                const uint32_t ctl = XZ_BYTE(pos);
                pos++;
                if (ctl == 0) break;
                if (ctl >= 0xE0 || ctl == 1) {
                    need_props = 1;
                    need_dict_reset = 0;
                    dict_start = opos;
                } else if (need_dict_reset) {
                    goto finish;
                }
                if (ctl >= 0x80) {
                    XZ_NEED(4);
                    const uint32_t usize = ((ctl & 0x1Fu) << 16) + (XZ_BYTE(pos) << 8) + XZ_BYTE(pos + 1) + 1u;
                    const uint32_t csize = (XZ_BYTE(pos + 2) << 8) + XZ_BYTE(pos + 3) + 1u;
                    pos += 4;
                    uint32_t reset_model = ctl >= 0xA0;
                    if (ctl >= 0xC0) {
                        XZ_NEED(1);
                        uint32_t d = XZ_BYTE(pos);
                        pos++;
                        if (d > (4 * 5 + 4) * 9 + 8) goto finish;
                        const uint32_t nlc = d % 9;
                        d /= 9;
                        const uint32_t nlp = d % 5, npb = d / 5;
                        if (nlc + nlp > 4) goto finish;
                        if (nlc + nlp > MZ_LZMA_MAX_LCLP && !prx) {
                            status = MZHIP_UNSUPPORTED; /* the model's upper half needs the overflow scratch */
                            goto finish;
                        }
                        lc = nlc;
                        lp_mask = (1u << nlp) - 1;
                        pb_mask = (1u << npb) - 1;
                        need_props = 0;
                    } else if (need_props) {
                        goto finish;
                    }
                    if (reset_model) {
                        MZ_LANES {
                            for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64)
                                ((uint32_t *)pr)[i] = 0x04000400u;
                            if (prx) /* the upper half of the literal model (lc + lp = 4) */
                                for (uint32_t i = (uint32_t)lane; i < MZ_LZMA_XPROBS / 2; i += 64) ((uint32_t *)prx)[i] = 0x04000400u;
                        }
                        MZ_WAVE_SYNC();
                        state = 0;
                        rep0 = rep1 = rep2 = rep3 = 0;
                    }
                    /* the chunk's compressed bytes are one self-contained range-coder run */
                    const uint32_t short_in = (in_len - pos) < csize;
                    rc_in = in + pos;
                    rc_len = short_in ? in_len - pos : csize;
                    in_pos = 0;
                    in_base = 0;
                    eof = 0;
                    range = 0xFFFFFFFFu;
                    code = 0;
                    if (rc_len > 0 && MZ_UNIFORM(rc_in[0]) != 0) goto finish; /* liblzma: first coder byte must be 0 */
                    LZ_REFILL();
                    for (int i = 0; i < 5; i++) {
                        uint32_t b;
                        LZ_NEXT_BYTE(b);
                        code = (code << 8) | b;
                    }
                    chunk_end = opos + usize; /* running into out_cap first is MZHIP_OUT_FULL from the loop itself */
                    if (opos > dict_start) prev_byte = MZ_UNIFORM(out[opos - 1]);
                    else prev_byte = 0;
                    if (state >= 7 && rep0 < opos - dict_start) match_byte = MZ_UNIFORM(out[opos - rep0 - 1]);
                    rc_short = short_in;
                    if (eof) goto finish;
                    LZ_PACKET_LOOP();
                    LZ_NORM(); /* liblzma normalises once more before it requires the coder to hold 0 */
                    if (eof) goto finish;
                    if (code != 0 || in_pos != csize) goto finish;
                    pos += csize;
                } else {
                    if (ctl > 2) goto finish;
                    XZ_NEED(2);
                    uint32_t n = (XZ_BYTE(pos) << 8) + XZ_BYTE(pos + 1) + 1u;
                    pos += 2;
                    uint32_t short_in = 0, full = 0;
                    if (n > in_len - pos) {
                        n = in_len - pos;
                        short_in = 1;
                    }
                    if (n > out_cap - opos) {
                        n = out_cap - opos;
                        full = 1;
                        short_in = 0;
                    }
                    MZ_LANES {
                        for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[opos + i] = in[pos + i];
                    }
                    MZ_WAVE_SYNC();
                    opos += n;
                    pos += n;
                    MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, LZ_CRC_LIMIT(opos), crc_tab, tabs->kx);
                    if (full) {
                        status = MZHIP_OUT_FULL;
                        goto finish;
                    }
                    if (short_in) {
                        status = MZHIP_BUF_ERROR;
                        goto finish;
                    }
                }
            }
            /* ---- sizes, block padding, check (3.3, 3.4) ---- */
            const uint32_t csize_blk = pos - data_pos, usize_blk = opos - block_out;
            if ((want_csize != ~0ull && want_csize != csize_blk) || (want_usize != ~0ull && want_usize != usize_blk))
                goto finish;
            if (npre) {
                mz_xz_unfilter_block(out + block_out, usize_blk, npre, fids, farg0, farg1, farg2, (uint8_t *)pr);
                crc_hold = 0xFFFFFFFFu;
            }
            while ((pos - data_pos) & 3u) {
                XZ_NEED(1);
                if (XZ_BYTE(pos) != 0) goto finish;
                pos++;
            }
            XZ_NEED(check_size);
            if (check == 1) {
                uint32_t k;
                XZ_CRC32_RANGE(k, out + block_out, usize_blk);
                if (k != XZ_LE32(pos)) goto finish;
            } else if (check == 4) {
                uint64_t k;
                MZ_CRC64(k, out + block_out, (uint64_t)usize_blk, tab64);
                if ((uint32_t)k != XZ_LE32(pos) || (uint32_t)(k >> 32) != XZ_LE32(pos + 4)) goto finish;
            } else if (check == 10) {
                uint32_t h[8], bad = 0;
                mz_sha256_init(h, 0);
                mz_sha256_run(out + block_out, (uint64_t)usize_blk, h);
                for (uint32_t i = 0; i < 8; i++) {
                    const uint32_t w = (XZ_BYTE(pos + 4 * i) << 24) | (XZ_BYTE(pos + 4 * i + 1) << 16) |
                                       (XZ_BYTE(pos + 4 * i + 2) << 8) | XZ_BYTE(pos + 4 * i + 3);
                    bad |= w ^ MZ_UNIFORM(h[i]);
                }
                if (bad) goto finish;
            }
            pos += check_size;
            nblocks++;
            blocks_digest = mz_xz_mix(mz_xz_mix(blocks_digest, (uint64_t)hsize + csize_blk + check_size), usize_blk);
        }

        /* ---- index (4) and stream footer (2.1.2) ---- */
        {
            const uint32_t ipos = pos;
            uint32_t p = ipos + 1u;
            uint64_t count, rec_digest = 0;
            XZ_VLI(count, p, in_len, MZHIP_BUF_ERROR);
            if (count != nblocks) goto finish;
            for (uint64_t i = 0; i < count; i++) {
                uint64_t unpadded, usz;
                XZ_VLI(unpadded, p, in_len, MZHIP_BUF_ERROR);
                XZ_VLI(usz, p, in_len, MZHIP_BUF_ERROR);
                rec_digest = mz_xz_mix(mz_xz_mix(rec_digest, unpadded), usz);
            }
            if (rec_digest != blocks_digest) goto finish;
            pos = p;
            while ((pos - ipos) & 3u) {
                XZ_NEED(1);
                if (XZ_BYTE(pos) != 0) goto finish;
                pos++;
            }
            XZ_NEED(4);
            {
                uint32_t k;
                XZ_CRC32_RANGE(k, in + ipos, pos - ipos);
                if (k != XZ_LE32(pos)) goto finish;
            }
            pos += 4;
            const uint32_t isize = pos - ipos;
            XZ_NEED(12);
            const uint32_t f = pos;
            pos += 12;
            uint32_t k;
            XZ_CRC32_RANGE(k, in + f + 4, 6u);
            if (XZ_BYTE(f + 10) != 'Y' || XZ_BYTE(f + 11) != 'Z' || k != XZ_LE32(f) || XZ_BYTE(f + 8) != 0 ||
                XZ_BYTE(f + 9) != check || ((uint64_t)XZ_LE32(f + 4) + 1u) * 4u != isize)
                goto finish;
            status = MZHIP_OK;
        }
    }

finish:
    {
        uint32_t olen = opos;
        /* an exit inside a block whose filters are not undone yet (output full, a data error, short input): the bytes
         * from the start of that block on are still in the filtered domain -- liblzma streams its filter chain and would
         * never have shown them -- so the result ends where the block began (crc_hold) */
        if (crc_hold != 0xFFFFFFFFu && olen > crc_hold) olen = crc_hold;
        if (max_out >= 0 && (int64_t)olen > max_out) olen = (uint32_t)max_out; /* mz_strm_lzma.c:214-215 */
        if (status == MZHIP_DATA_ERROR && eof && rc_short) {
            /* the coder ran off a chunk that the input does not hold completely: input ended early */
            status = MZHIP_BUF_ERROR;
            pos = in_len;
        }
        res->status = status;
        res->out_len = olen;
        res->in_used = pos > in_len ? in_len : pos;
        uint32_t crc;
        if (crc_done > olen) {
            crc_done = 0;
            MZ_LANES { P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u; }
        }
        MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, olen, crc_tab, tabs->kx);
        MZ_CRC_FINISH(crc, crc_acc, crc_tmp, crc_done, out, olen, crc_tab, tabs);
        res->crc = crc;
    }
}
#undef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) (o)

/* ---- LZMA2 in windows: method-95 entries of any size in bounded memory (shim_lzma.c) ----------------------------------
 * The reference streams an .xz entry through 32 767 bytes (mz_strm_lzma.c:127-128,147-241: lzma_stream_decoder +
 * lzma_code per staging buffer).  Here the container -- stream header, block headers, padding, check fields, index, footer:
 * a few dozen bytes per block -- is walked by the shim, and a block's LZMA2 chunk sequence (.xz 1.0.4 section 5.3.1's
 * filter, liblzma's lzma2_decoder.c) is decoded by this function one window at a time: `in` starts at a chunk header, or
 * inside the chunk the call before stopped in; out[0 .. rs->out_pos) is the dictionary so far.  It stops (st->flags bit 0)
 * in front of a chunk or in front of a packet when fewer than 274 bytes of room or (unless flags bit 1 says the input is
 * the entry's last) fewer than 64 bytes of the chunk's input are left, and after the block's end byte (bit 4).  The block's
 * check -- CRC-32 or CRC-64 of everything the block decodes to -- is carried through the windows in the state. */
typedef struct mz_lzma2_state {
    uint32_t flags;   /* in: 1 model in `model` (else a fresh block: the first chunk resets everything), 2 last input,
                         4 the next LZMA chunk must bring properties, 8 the next chunk must reset the dictionary,
                         32 inside an uncompressed chunk, 64 inside an LZMA chunk; out: the same, 1 = a state to go on
                         from, 16 = the block's end byte has been consumed, 128 = the data error was met between chunks
                         (control byte, chunk header, the coder's first bytes, its state at the chunk's end) and not
                         inside a packet: liblzma meets those without room in the output */
    uint32_t range, code, state;
    uint32_t rep0, rep1, rep2, rep3;
    uint32_t props;   /* lc | lp << 8 | pb << 16 */
    uint32_t dict;    /* the block header's dictionary size */
    uint32_t out_pos; /* in: bytes of dictionary in front of the room; out: bytes valid in the buffer */
    uint32_t in_pos;  /* out: bytes of the given input that are done with */
    uint32_t dict_start; /* where in the buffer the dictionary begins (0: further back than the buffer reaches) */
    uint32_t chunk_usize_left, chunk_csize_left; /* inside a chunk: what is left of it */
    uint32_t check_id; /* 0 none, 1 CRC-32, 4 CRC-64 */
    uint32_t check_lo, check_hi; /* the check of the block's bytes so far */
    uint32_t pad[2];
} mz_lzma2_state;

#undef LZ_RESUME_CHECK
#define LZ_RESUME_CHECK()                                                                                   \
    if (opos + 274u > out_cap || (rc_partial && in_pos + 64u > rc_len)) {                                   \
        status = (opos + 274u > out_cap) ? MZHIP_OUT_FULL : MZHIP_BUF_ERROR;                                \
        goto stop_in_chunk;                                                                                 \
    }
#undef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) 0u /* no fused CRC: the block's check is folded when the window is done */
MZ_DEV void mz_lzma2_run(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, mz_xz_lds *L,
                         const uint32_t *crc_tab, const mzhip_crc_tables *tabs, uint16_t *model, const mz_lzma2_state *rs,
                         mz_lzma2_state *st, mz_lzma_result *res) {
    MZ_LANE_DECL
    uint16_t *pr = L->lz.probs;
    uint16_t *const prx = model + ((LZ_NUM_PROBS + 1u) & ~1u);
    const uint64_t *tab64 = L->crc64_tab;
    int32_t status = MZHIP_DATA_ERROR;
    const uint32_t fin = MZ_UNIFORM(rs->flags);
    const uint32_t last_input = (fin >> 1) & 1u;
    uint32_t need_props = (fin >> 2) & 1u, need_dict_reset = (fin >> 3) & 1u;
    uint32_t in_raw = (fin >> 5) & 1u, in_lz = (fin >> 6) & 1u;
    uint32_t stopped = 0, ended = 0;
    uint32_t front = 1; /* between packets of different chunks: control byte, chunk header, coder start and end */
    uint32_t pos = 0; /* cursor in `in` */
    uint32_t opos = MZ_UNIFORM(rs->out_pos);
    const uint32_t start = opos;
    const uint8_t *rc_in = in;
    uint32_t rc_len = 0, in_pos = 0, in_base = 0, eof = 0, range = MZ_UNIFORM(rs->range), code = MZ_UNIFORM(rs->code);
    const uint32_t lzma2 = 1;
    uint32_t chunk_end = 0, dict_start = MZ_UNIFORM(rs->dict_start), rc_short = 0, rc_partial = 0;
    uint32_t usize_left = MZ_UNIFORM(rs->chunk_usize_left), csize_left = MZ_UNIFORM(rs->chunk_csize_left), csize = 0;
    uint64_t dict = MZ_UNIFORM(rs->dict);
    uint32_t state = MZ_UNIFORM(rs->state), rep0 = MZ_UNIFORM(rs->rep0), rep1 = MZ_UNIFORM(rs->rep1), rep2 = MZ_UNIFORM(rs->rep2),
             rep3 = MZ_UNIFORM(rs->rep3), prev_byte = 0, match_byte = 0;
    const uint32_t pw = MZ_UNIFORM(rs->props);
    uint32_t lc = pw & 0xFFu, lp_mask = (1u << ((pw >> 8) & 7u)) - 1u, pb_mask = (1u << ((pw >> 16) & 7u)) - 1u;
    uint32_t lp_n = (pw >> 8) & 7u, pb_n = (pw >> 16) & 7u;
    PV(uint32_t, win);
    PV(uint32_t, crc_acc);
    uint32_t crc_done = 0;
    MZ_LANES {
        P(crc_acc) = 0u;
        P(win) = 0;
    }
    (void)crc_done;
    if (dict < 4096) dict = 4096;
    dict = (dict + 15) & ~(uint64_t)15;
    if (opos > out_cap || state > 11u || dict_start > opos || lc + lp_n > 4u || pb_n > 4u || (in_raw && in_lz)) goto finish;
    if (fin & 1u) {
        MZ_LANES {
            for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64) ((uint32_t *)pr)[i] = ((const uint32_t *)model)[i];
        }
        MZ_WAVE_SYNC();
    } else if (in_lz || !need_props) {
        goto finish; /* a chunk sequence in progress without its model */
    }

    for (;;) {
        if (in_raw) {
            /* ---- the rest of an uncompressed chunk ---- */
            uint32_t n = usize_left, stop = 0;
            if (n > in_len - pos) {
                n = in_len - pos;
                stop = MZHIP_BUF_ERROR;
            }
            if (n > out_cap - opos) {
                n = out_cap - opos;
                stop = MZHIP_OUT_FULL;
            }
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[opos + i] = in[pos + i];
            }
            MZ_WAVE_SYNC();
            opos += n;
            pos += n;
            usize_left -= n;
            if (usize_left != 0u) {
                status = (int32_t)stop;
                stopped = 1;
                goto finish;
            }
            in_raw = 0;
            continue;
        }
        if (!in_lz) {
            /* ---- in front of a chunk ---- */
            if (opos + 274u > out_cap) {
                status = MZHIP_OUT_FULL;
                stopped = 1;
                goto finish;
            }
            if (pos >= in_len) {
                status = MZHIP_BUF_ERROR;
                stopped = 1;
                goto finish;
            }
            const uint32_t ctl = XZ_BYTE(pos);
            if (ctl == 0) {
                pos++;
                ended = 1;
                status = MZHIP_OK;
                goto finish;
            }
            const uint32_t resets_dict = (ctl >= 0xE0 || ctl == 1) ? 1u : 0u;
            if (!resets_dict && need_dict_reset) goto finish;
            if (ctl < 0x80) {
                if (ctl > 2) goto finish;
                if (in_len - pos < 3u) {
                    status = MZHIP_BUF_ERROR;
                    stopped = 1;
                    goto finish;
                }
                usize_left = (XZ_BYTE(pos + 1) << 8) + XZ_BYTE(pos + 2) + 1u;
                pos += 3;
                if (resets_dict) {
                    need_props = 1;
                    need_dict_reset = 0;
                    dict_start = opos;
                }
                in_raw = 1;
                continue;
            }
            const uint32_t hl = ctl >= 0xC0 ? 6u : 5u;
            if (in_len - pos < hl) {
                status = MZHIP_BUF_ERROR;
                stopped = 1;
                goto finish;
            }
            usize_left = ((ctl & 0x1Fu) << 16) + (XZ_BYTE(pos + 1) << 8) + XZ_BYTE(pos + 2) + 1u;
            csize = (XZ_BYTE(pos + 3) << 8) + XZ_BYTE(pos + 4) + 1u;
            {
                /* all of the chunk, or enough of it for the coder's five bytes and a packet: else more input first */
                const uint32_t avail = in_len - pos - hl;
                if (!last_input && avail < csize && avail < 128u) {
                    status = MZHIP_BUF_ERROR;
                    stopped = 1;
                    goto finish;
                }
            }
            if (resets_dict) {
                need_props = 1;
                need_dict_reset = 0;
                dict_start = opos;
            }
            if (ctl >= 0xC0) {
                uint32_t d = XZ_BYTE(pos + 5);
                if (d > (4 * 5 + 4) * 9 + 8) goto finish;
                const uint32_t nlc = d % 9;
                d /= 9;
                const uint32_t nlp = d % 5, npb = d / 5;
                if (nlc + nlp > 4) goto finish;
                lc = nlc;
                lp_n = nlp;
                pb_n = npb;
                lp_mask = (1u << nlp) - 1;
                pb_mask = (1u << npb) - 1;
                need_props = 0;
            } else if (need_props) {
                goto finish;
            }
            pos += hl;
            if (ctl >= 0xA0) {
                MZ_LANES {
                    for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64) ((uint32_t *)pr)[i] = 0x04000400u;
                    for (uint32_t i = (uint32_t)lane; i < MZ_LZMA_XPROBS / 2; i += 64) ((uint32_t *)prx)[i] = 0x04000400u;
                }
                MZ_WAVE_SYNC();
                state = 0;
                rep0 = rep1 = rep2 = rep3 = 0;
            }
            /* the chunk's compressed bytes are one self-contained range-coder run */
            rc_short = (in_len - pos) < csize;
            rc_partial = rc_short && !last_input;
            rc_in = in + pos;
            rc_len = rc_short ? in_len - pos : csize;
            in_pos = 0;
            in_base = 0;
            eof = 0;
            range = 0xFFFFFFFFu;
            code = 0;
            if (rc_len > 0 && MZ_UNIFORM(rc_in[0]) != 0) goto finish; /* liblzma: first coder byte must be 0 */
            LZ_REFILL();
            for (int i = 0; i < 5; i++) {
                uint32_t b;
                LZ_NEXT_BYTE(b);
                code = (code << 8) | b;
            }
        } else {
            /* ---- inside the LZMA chunk the call before stopped in ---- */
            in_lz = 0;
            csize = csize_left;
            rc_short = (in_len - pos) < csize;
            rc_partial = rc_short && !last_input;
            rc_in = in + pos;
            rc_len = rc_short ? in_len - pos : csize;
            in_pos = 0;
            in_base = 0;
            eof = 0;
            LZ_REFILL();
        }
        chunk_end = opos + usize_left;
        prev_byte = opos > dict_start ? MZ_UNIFORM(out[opos - 1]) : 0u;
        match_byte = (state >= 7 && rep0 < opos - dict_start) ? MZ_UNIFORM(out[opos - rep0 - 1]) : 0u;
        if (eof) goto finish;
        front = 0;
        LZ_PACKET_LOOP();
        front = 1;
        LZ_NORM(); /* liblzma normalises once more before it requires the coder to hold 0 */
        if (eof) goto finish;
        if (code != 0 || in_pos != csize) goto finish;
        pos += csize;
        in_pos = 0;
        usize_left = 0;
    }
stop_in_chunk:
    stopped = 1;
    in_lz = 1;
    usize_left = chunk_end - opos;
    csize_left = csize - in_pos;
    pos += in_pos;
    in_pos = 0;
    eof = 0;

finish:
    if (status == MZHIP_DATA_ERROR && eof && rc_short) {
        status = MZHIP_BUF_ERROR; /* the coder ran off a chunk that the input does not hold completely: input ended early */
        pos = in_len;
    } else if (!stopped && !ended && in_pos != 0u) {
        pos += in_pos > rc_len ? rc_len : in_pos; /* a failed chunk: as far as the coder had read */
    }
    {
        /* the block's check over this window's bytes, taken up from the windows before */
        const uint32_t cid = MZ_UNIFORM(rs->check_id);
        uint32_t klo = MZ_UNIFORM(rs->check_lo), khi = MZ_UNIFORM(rs->check_hi);
        const uint32_t fresh = opos > start ? opos - start : 0u;
        if (cid == 1u && fresh) {
            PV(uint32_t, _xa);
            PV(uint32_t, _xt);
            uint32_t _xd = 0;
            MZ_LANES { P(_xa) = (lane == 0) ? ~klo : 0u; }
            MZ_CRC_FOLD_TILES(_xa, _xd, out + start, fresh, crc_tab, tabs->kx);
            MZ_CRC_FINISH_FROM(klo, _xa, _xt, _xd, out + start, fresh, crc_tab, tabs, ~klo);
        } else if (cid == 4u && fresh) {
            uint64_t k;
            MZ_CRC64_FROM(k, ((uint64_t)khi << 32) | klo, out + start, (uint64_t)fresh, tab64);
            klo = (uint32_t)k;
            khi = (uint32_t)(k >> 32);
        }
        if (stopped) {
            MZ_LANES {
                for (uint32_t i = (uint32_t)lane; i < (LZ_NUM_PROBS + 1) / 2; i += 64) ((uint32_t *)model)[i] = ((const uint32_t *)pr)[i];
            }
        }
        MZ_LANES { /* uniform stores */
            st->flags = stopped | (need_props << 2) | (need_dict_reset << 3) | (ended << 4) | (in_raw << 5) | (in_lz << 6) |
                        ((status == MZHIP_DATA_ERROR && front) ? 128u : 0u);
            st->range = range;
            st->code = code;
            st->state = state;
            st->rep0 = rep0;
            st->rep1 = rep1;
            st->rep2 = rep2;
            st->rep3 = rep3;
            st->props = lc | (lp_n << 8) | (pb_n << 16);
            st->dict = MZ_UNIFORM(rs->dict);
            st->out_pos = opos;
            st->in_pos = pos > in_len ? in_len : pos;
            st->dict_start = dict_start;
            st->chunk_usize_left = usize_left;
            st->chunk_csize_left = csize_left;
            st->check_id = cid;
            st->check_lo = klo;
            st->check_hi = khi;
            st->pad[0] = st->pad[1] = 0u;
        }
        MZ_WAVE_SYNC();
        res->status = status;
        res->out_len = opos;
        res->in_used = pos > in_len ? in_len : pos;
        res->crc = 0;
    }
}
#undef LZ_RESUME_CHECK
#define LZ_RESUME_CHECK() ((void)0)
#undef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) (o)

#endif
