/* lzma_core.h -- raw-LZMA1 range decode of ONE entry by ONE wavefront, CRC-32
 * fused (kernel K3 of SURVEY 2.1).
 *
 * Replaces what the reference does for a method-14 entry through
 * mz_stream_lzma_read (mz_strm_lzma.c:147-241 -> liblzma lzma_alone_decoder /
 * lzma_code): input starts at the ZIP-LZMA header (2 B version, 2 B props size,
 * 5 B props: appnote.txt:2232-2275), uncompressed size unknown, stream ends at
 * the end-of-stream marker, TOTAL_OUT clamped to TOTAL_OUT_MAX
 * (mz_strm_lzma.c:214-215).  Algorithm: the public-domain LZMA specification
 * (11-bit adaptive bit models, 12 states, rep0-3, two length coders, 6-bit
 * position slots, aligned bits).  liblzma 5.2.5 behaviours kept: first coder
 * byte must be 0x00; lazy normalisation (before each bit); distance valid iff
 * < min(produced, dict size rounded up to >=4096, multiple of 16); code == 0
 * required after the EOS marker.
 *
 * MI355X mapping: the adaptive model makes a stream strictly serial, so the
 * parallel axis is entries.  One wave = one entry; the whole probability model
 * (7 990 x u16 = 15.6 KiB for lc+lp <= 3) lives in that wave's LDS slice, ten
 * waves per CU; the range coder state is wave-uniform (scalar unit); compressed
 * bytes are fetched 256 at a time into VGPRs (one coalesced load) and handed to
 * the scalar side by v_readlane; match copies and the CRC tiles use all 64
 * lanes; the dictionary is the output buffer itself.
 */
#ifndef MZHIP_LZMA_CORE_H
#define MZHIP_LZMA_CORE_H

#include "crc32_core.h"
#include "wave.h"

#ifndef MZ_LZMA_MAX_LCLP
#define MZ_LZMA_MAX_LCLP 3 /* literal contexts held in LDS: 0x300 << 3 probabilities (measurement builds: 0, profiles/r3/scripts/ab_k3.sh) */
#endif
#define MZ_LZMA_LIT_PROBS (0x300u << MZ_LZMA_MAX_LCLP)
#define MZ_LZMA_XPROBS MZ_LZMA_LIT_PROBS /* lc + lp = 4: as many again, in a per-wave scratch in HBM */

/* probability model layout inside the per-wave LDS slice (u16 indices) */
#define LZ_IS_MATCH 0                      /* [12][16] */
#define LZ_IS_REP (LZ_IS_MATCH + 192)      /* [12] */
#define LZ_IS_REP_G0 (LZ_IS_REP + 12)      /* [12] */
#define LZ_IS_REP_G1 (LZ_IS_REP_G0 + 12)   /* [12] */
#define LZ_IS_REP_G2 (LZ_IS_REP_G1 + 12)   /* [12] */
#define LZ_IS_REP0_LONG (LZ_IS_REP_G2 + 12) /* [12][16] */
#define LZ_POS_SLOT (LZ_IS_REP0_LONG + 192) /* [4][64] */
#define LZ_POS_DEC (LZ_POS_SLOT + 256)     /* [115] + 1 unused: every bit tree below starts on an even index, so the two
                                              children of a node are one aligned 32-bit word (LZ_BITTREE_PF) */
#define LZ_ALIGN (LZ_POS_DEC + 116)        /* [16] */
#define LZ_LEN (LZ_ALIGN + 16)             /* choice, choice2, low[16][8], mid[16][8], high[256] = 514 */
#define LZ_REP_LEN (LZ_LEN + 514)
#define LZ_LIT (LZ_REP_LEN + 514)
#define LZ_NUM_PROBS (LZ_LIT + MZ_LZMA_LIT_PROBS)

/* how far the fused CRC may fold: everything written -- except while an .xz block with Delta / BCJ filters is being
 * decoded, whose bytes are not final before the filters are undone (xz_core.h redefines this) */
#ifndef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) (o)
#endif

typedef struct mz_lzma_lds {
    uint16_t probs[(LZ_NUM_PROBS + 1) & ~1u];
} mz_lzma_lds;
#ifndef MZ_LZMA_SLOTS
#define MZ_LZMA_SLOTS 4u /* literal contexts the slot build keeps in LDS (see LZ_LITERAL_SITE_SLOT) */
#endif
#define LZ_NUM_PROBS_S (LZ_LIT + 0x300u * MZ_LZMA_SLOTS)
#define MZ_LZMA_SPROBS (0x300u << 4) /* the slot build's whole literal model in HBM, per resident wave */
typedef struct mz_lzma_lds_s {
    uint16_t probs[(LZ_NUM_PROBS_S + 1) & ~1u];
} mz_lzma_lds_s;

/* Where an LZMA1 decode can be taken up again (the resumable build of lzma_entry.inc; mzhip.h declares the same sixteen
 * words as mzhip_lzma_state).  The adaptive model travels beside it: MZ_LZMA_MODEL_U16 probabilities in global memory. */
typedef struct mz_lzma_state {
    uint32_t flags;   /* in: bit 0 take the stream up from this state (else a fresh stream, header first), bit 1 the input
                         given is the stream's last; out: 1 = stopped in front of a packet, the state is one to go on from */
    uint32_t range, code, state;
    uint32_t rep0, rep1, rep2, rep3;
    uint32_t props;   /* lc | lp << 8 | pb << 16 */
    uint32_t dict;    /* the header's dictionary size */
    uint32_t out_pos; /* in: bytes of dictionary in front of the output; out: bytes valid in the buffer */
    uint32_t in_pos;  /* out: bytes of the given input that are done with */
    uint32_t pad[4];
} mz_lzma_state;
#define MZ_LZMA_MODEL_U16 (((LZ_NUM_PROBS + 1u) & ~1u) + MZ_LZMA_XPROBS)

/* the packet loop's stop test: nothing in the one-shot builds */
#define LZ_RESUME_CHECK() ((void)0)
/* the wave's issue priority (s_setprio): the chain of decisions against the work all lanes do between packets (CRC tiles,
 * match copies).  MZ_LZPRIO = chain << 2 | bulk; 0 = no priorities, the default: what pays in K1 and K4 loses here -- one
 * round of 4096 x 1 MiB streams 431 ms without, 440 with the chain on top (3 | 0, 2 | 1), 448 with the bulk work on top: every
 * wave is in its chain nearly all the time, a priority only reorders equals (profiles/r6/ab_setprio_k3_k4.log) */
#ifndef MZ_LZPRIO
#define MZ_LZPRIO 0
#endif
#if MZ_LZPRIO && !defined(MZHIP_HOST_EMUL)
#define LZ_PRIO_CHAIN() __builtin_amdgcn_s_setprio(((MZ_LZPRIO) >> 2) & 3)
#define LZ_PRIO_BULK() __builtin_amdgcn_s_setprio((MZ_LZPRIO) & 3)
#else
#define LZ_PRIO_CHAIN() ((void)0)
#define LZ_PRIO_BULK() ((void)0)
#endif

typedef struct mz_lzma_result {
    int32_t status;
    uint32_t out_len;
    uint32_t in_used;
    uint32_t crc;
} mz_lzma_result;

/* ---- wave-uniform range coder; the 256-byte input window lives one dword per lane ---- */
#define LZ_REFILL()                                                                     \
    do {                                                                                \
        MZ_LANES {                                                                      \
            uint32_t _o = in_base + 4u * (uint32_t)lane;                                \
            uint32_t _d = 0;                                                            \
            for (uint32_t _b = 0; _b < 4; _b++)                                         \
                if (_o + _b < rc_len) _d |= (uint32_t)rc_in[_o + _b] << (8 * _b);          \
            P(win) = _d;                                                                \
        }                                                                               \
    } while (0)

#define LZ_NEXT_BYTE(dst)                                                               \
    do {                                                                                \
        if (in_pos >= rc_len) {                                                         \
            eof = 1;                                                                    \
            (dst) = 0;                                                                  \
        } else {                                                                        \
            if (in_pos - in_base >= 256u) {                                             \
                in_base = in_pos;                                                       \
                LZ_REFILL();                                                            \
            }                                                                           \
            uint32_t _rel = in_pos - in_base;                                           \
            uint32_t _dw = LZ_WIN_DW(_rel >> 2);                                        \
            (dst) = (_dw >> (8u * (_rel & 3u))) & 0xFFu;                                \
            in_pos++;                                                                   \
        }                                                                               \
    } while (0)

#define LZ_NORM()                                                                       \
    do {                                                                                \
        if (LZ_UBR(range < (1u << 24))) {                                               \
            uint32_t _nb;                                                               \
            LZ_NEXT_BYTE(_nb);                                                          \
            range <<= 8;                                                                \
            code = (code << 8) | _nb;                                                   \
        }                                                                               \
    } while (0)

/* `code < bound` and `code - bound` (one subtraction with borrow for both measured slower, round 4: config 4 10.35 -> 10.19
 * GiB/s, profiles/r4/ab_k3_borrow.log) */
#define LZ_BORROW(a, b, d) (*(d) = (a) - (b), (uint32_t)((a) < (b)))

#define LZ_BIT(bit, idx)                                                                \
    do {                                                                                \
        LZ_NORM();                                                                      \
        uint32_t _pi = (idx);                                                           \
        uint32_t _p = LZ_U(pr[_pi]);                                             \
        uint32_t _bound = (range >> 11) * _p;                                           \
        uint32_t _diff;                                                                 \
        const uint32_t _lt = LZ_BORROW(code, _bound, &_diff);                           \
        if (LZ_UBR(_lt != 0u)) {                                                        \
            range = _bound;                                                             \
            _p += (2048u - _p) >> 5;                                                    \
            (bit) = 0;                                                                  \
        } else {                                                                        \
            range -= _bound;                                                            \
            code = _diff;                                                               \
            _p -= _p >> 5;                                                              \
            (bit) = 1;                                                                  \
        }                                                                               \
        MZ_LANES { pr[_pi] = (uint16_t)_p; } /* uniform store, no lane-0 branch */      \
        MZ_WAVE_SYNC();                                                                 \
    } while (0)

/* the same on the overflow part of the literal model (lc + lp = 4: the second half of the 0x300 << 4 literal
 * probabilities does not fit the wave's LDS slice and lives in a per-wave scratch in HBM, prx[]) */
#define LZ_BIT_X(bit, idx)                                                              \
    do {                                                                                \
        LZ_NORM();                                                                      \
        uint32_t _pi = (idx) - LZ_NUM_PROBS;                                            \
        uint32_t _p = LZ_U(prx[_pi]);                                                   \
        uint32_t _bound = (range >> 11) * _p;                                           \
        uint32_t _diff;                                                                 \
        const uint32_t _lt = LZ_BORROW(code, _bound, &_diff);                           \
        if (LZ_UBR(_lt != 0u)) {                                                        \
            range = _bound;                                                             \
            _p += (2048u - _p) >> 5;                                                    \
            (bit) = 0;                                                                  \
        } else {                                                                        \
            range -= _bound;                                                            \
            code = _diff;                                                               \
            _p -= _p >> 5;                                                              \
            (bit) = 1;                                                                  \
        }                                                                               \
        MZ_LANES { prx[_pi] = (uint16_t)_p; } /* uniform store, no lane-0 branch */     \
        MZ_WAVE_SYNC();                                                                 \
    } while (0)

#define LZ_BITTREE(sym, base, nbits)                                                    \
    do {                                                                                \
        uint32_t _m = 1;                                                                \
        for (int _i = 0; _i < (nbits); _i++) {                                          \
            uint32_t _b;                                                                \
            LZ_BIT(_b, (base) + _m);                                                    \
            _m = (_m << 1) + _b;                                                        \
        }                                                                               \
        (sym) = _m - (1u << (nbits));                                                   \
    } while (0)

#define LZ_BITTREE_REV(sym, base, nbits)                                                \
    do {                                                                                \
        uint32_t _m = 1, _s = 0;                                                        \
        for (int _i = 0; _i < (int)(nbits); _i++) {                                     \
            uint32_t _b;                                                                \
            LZ_BIT(_b, (base) + _m);                                                    \
            _m = (_m << 1) + _b;                                                        \
            _s |= _b << _i;                                                             \
        }                                                                               \
        (sym) = _s;                                                                     \
    } while (0)

/* ---- the same trees with the LDS round trip off the decision chain (MZ_LZMA_PAIRS, default on) ----
 * K3 is one chain of dependent instructions per decision; the read of the next node's probability was on it: the bit
 * decides the index, the index the address, the address the read.  The two children of node m are the nodes 2m and
 * 2m + 1 -- neighbours, and with every tree on an even index one aligned word: that word is fetched BEFORE node m is decided
 * and the bit only picks the half.  (Round 2 measured a form of this as slower on the scalar-port build at 10 waves per CU;
 * with the decisions on uniform branches and 16 waves per CU the chain is what is left: profiles/r4/ab_k3_pairs.log.) */
#ifndef MZ_LZMA_PAIRS
#define MZ_LZMA_PAIRS 1
#endif
MZ_DEV uint32_t mz_prob_pair(const uint16_t *pr, uint32_t idx) { /* idx even */
    uint32_t v;
    __builtin_memcpy(&v, pr + idx, 4);
    return v;
}
MZ_DEV uint32_t mz_prob_half(uint32_t pair, uint32_t b) { return (pair >> (b << 4)) & 0xFFFFu; }
/* one bit with the probability already in hand (pv); its update goes to probs[idx] */
#define LZ_BIT_P(bit, idx, pv)                                                          \
    do {                                                                                \
        LZ_NORM();                                                                      \
        uint32_t _pi = (idx);                                                           \
        uint32_t _p = (pv);                                                             \
        uint32_t _bound = (range >> 11) * _p;                                           \
        uint32_t _diff;                                                                 \
        const uint32_t _lt = LZ_BORROW(code, _bound, &_diff); /* code < bound and code - bound from one subtraction */ \
        if (LZ_UBR(_lt != 0u)) {                                                        \
            range = _bound;                                                             \
            _p += (2048u - _p) >> 5;                                                    \
            (bit) = 0;                                                                  \
        } else {                                                                        \
            range -= _bound;                                                            \
            code = _diff;                                                               \
            _p -= _p >> 5;                                                              \
            (bit) = 1;                                                                  \
        }                                                                               \
        MZ_LANES { pr[_pi] = (uint16_t)_p; } /* uniform store, no lane-0 branch */      \
        MZ_WAVE_SYNC();                                                                 \
    } while (0)
#if MZ_LZMA_PAIRS
#define LZ_BITTREE_PF(sym, base, nbits)                                                 \
    do {                                                                                \
        uint32_t _m = 1;                                                                \
        uint32_t _pv = LZ_U(pr[(base) + 1u]);                                           \
        for (int _i = 0; _i < (nbits); _i++) {                                          \
            uint32_t _pair = 0;                                                         \
            if (_i + 1 < (nbits)) _pair = LZ_U(mz_prob_pair(pr, (base) + 2u * _m));     \
            uint32_t _b;                                                                \
            LZ_BIT_P(_b, (base) + _m, _pv);                                             \
            _m = (_m << 1) + _b;                                                        \
            _pv = mz_prob_half(_pair, _b);                                           \
        }                                                                               \
        (sym) = _m - (1u << (nbits));                                                   \
    } while (0)
#define LZ_BITTREE_REV_PF(sym, base, nbits)                                             \
    do {                                                                                \
        uint32_t _m = 1, _s = 0;                                                        \
        uint32_t _pv = LZ_U(pr[(base) + 1u]);                                           \
        for (int _i = 0; _i < (int)(nbits); _i++) {                                     \
            uint32_t _pair = 0;                                                         \
            if (_i + 1 < (int)(nbits)) _pair = LZ_U(mz_prob_pair(pr, (base) + 2u * _m)); \
            uint32_t _b;                                                                \
            LZ_BIT_P(_b, (base) + _m, _pv);                                             \
            _m = (_m << 1) + _b;                                                        \
            _s |= _b << _i;                                                             \
            _pv = mz_prob_half(_pair, _b);                                           \
        }                                                                               \
        (sym) = _s;                                                                     \
    } while (0)
/* a literal out of the LDS model: while the bits follow the match byte the children under the NEXT match bit and the
 * children in the plain tree are both on their way */
#define LZ_LITERAL_PF()                                                                 \
    do {                                                                                \
        uint32_t _pv;                                                                   \
        uint32_t _plain = 1;                                                            \
        if (state >= 7) {                                                               \
            uint32_t mb = match_byte;                                                   \
            uint32_t mbit = (mb >> 7) & 1u;                                             \
            mb <<= 1;                                                                   \
            _pv = LZ_U(pr[lbase + ((1u + mbit) << 8) + sym]);                           \
            _plain = 0;                                                                 \
            for (;;) {                                                                  \
                const uint32_t mbit2 = (mb >> 7) & 1u;                                  \
                uint32_t _pm = 0, _pp = 0;                                              \
                if (sym < 0x80u) {                                                      \
                    _pm = LZ_U(mz_prob_pair(pr, lbase + ((1u + mbit2) << 8) + 2u * sym)); \
                    _pp = LZ_U(mz_prob_pair(pr, lbase + 2u * sym));                     \
                }                                                                       \
                uint32_t b;                                                             \
                LZ_BIT_P(b, lbase + ((1u + mbit) << 8) + sym, _pv);                     \
                sym = (sym << 1) | b;                                                   \
                if (sym >= 0x100u) break;                                               \
                if (mbit != b) {                                                        \
                    _pv = mz_prob_half(_pp, b);                                      \
                    _plain = 2;                                                         \
                    break;                                                              \
                }                                                                       \
                _pv = mz_prob_half(_pm, b);                                          \
                mbit = mbit2;                                                           \
                mb <<= 1;                                                               \
            }                                                                           \
        }                                                                               \
        if (sym < 0x100u) {                                                             \
            if (_plain == 1) _pv = LZ_U(pr[lbase + sym]);                               \
            while (sym < 0x100u) {                                                      \
                uint32_t _pair = 0;                                                     \
                if (sym < 0x80u) _pair = LZ_U(mz_prob_pair(pr, lbase + 2u * sym));      \
                uint32_t b;                                                             \
                LZ_BIT_P(b, lbase + sym, _pv);                                          \
                sym = (sym << 1) | b;                                                   \
                _pv = mz_prob_half(_pair, b);                                        \
            }                                                                           \
        }                                                                               \
    } while (0)
#else
#define LZ_BITTREE_PF(sym, base, nbits) LZ_BITTREE(sym, base, nbits)
#define LZ_BITTREE_REV_PF(sym, base, nbits) LZ_BITTREE_REV(sym, base, nbits)
#define LZ_LITERAL_PF() LZ_LITERAL(LZ_BIT)
#endif

#define LZ_LEN_DECODE(len, lbase, ps)                                                   \
    do {                                                                                \
        uint32_t _c;                                                                    \
        LZ_BIT(_c, (lbase));                                                            \
        if (!_c) {                                                                      \
            LZ_BITTREE_PF(len, (lbase) + 2 + (ps) * 8, 3);                                 \
        } else {                                                                        \
            LZ_BIT(_c, (lbase) + 1);                                                    \
            if (!_c) {                                                                  \
                LZ_BITTREE_PF(len, (lbase) + 2 + 128 + (ps) * 8, 3);                       \
                (len) += 8;                                                             \
            } else {                                                                    \
                LZ_BITTREE_PF(len, (lbase) + 2 + 256, 8);                                  \
                (len) += 16;                                                            \
            }                                                                           \
        }                                                                               \
    } while (0)

/* one literal: with a match byte as context while the coder is in a "after match" state, then plain */
#define LZ_LITERAL(BITM)                                                                \
    do {                                                                                \
        if (state >= 7) {                                                               \
            uint32_t mb = match_byte;                                                   \
            do {                                                                        \
                uint32_t mbit = (mb >> 7) & 1u;                                         \
                mb <<= 1;                                                               \
                uint32_t b;                                                             \
                BITM(b, lbase + ((1u + mbit) << 8) + sym);                              \
                sym = (sym << 1) | b;                                                   \
                if (mbit != b) break;                                                   \
            } while (sym < 0x100);                                                      \
        }                                                                               \
        while (sym < 0x100) {                                                           \
            uint32_t b;                                                                 \
            BITM(b, lbase + sym);                                                       \
            sym = (sym << 1) | b;                                                       \
        }                                                                               \
    } while (0)

/* Where a literal's 0x300 probabilities are.  Full model (K3's fall-back kernel, the .xz kernel): context c at
 * LZ_LIT + 0x300 * c in LDS, the upper half of an lc + lp = 4 model in the HBM scratch. */
#define LZ_LITERAL_SITE_FULL(sym)                                                                                     \
    do {                                                                                                              \
        const uint32_t lbase = LZ_LIT + 0x300u * (((opos & lp_mask) << lc) + (prev_byte >> (8 - lc)));                \
        if (lbase < LZ_NUM_PROBS) {                                                                                   \
            LZ_LITERAL_PF();                                                                                          \
        } else { /* lc + lp = 4, upper half of the literal model */                                                   \
            LZ_LITERAL(LZ_BIT_X);                                                                                     \
        }                                                                                                             \
    } while (0)
/* Slot build (K3's main kernel): LDS holds MZ_LZMA_SLOTS literal contexts, the whole literal model lives in the wave's
 * HBM scratch prx[] (0x300 << 4 probabilities) and a context is swapped in when it is needed (1.5 KiB out, 1.5 KiB in,
 * all lanes).  Text touches three or four contexts (the top lc bits of the previous byte: letters, punctuation, capitals),
 * so it never swaps after the first bytes, and the wave's LDS slice is 9.6 KiB instead of 16.6: 16 streams per CU
 * instead of 10 -- K3 is one serial chain per wave, more waves are the only throughput there is (profiles/r3/
 * ab_k3_residency.log).  A stream that keeps swapping (binary data: all eight contexts live) is given back with
 * MZHIP_RETRY after a bounded number of swaps and decoded by the full-model kernel. */
#define MZHIP_RETRY (-300)
#define LZ_LITERAL_SITE_SLOT(sym)                                                                                     \
    do {                                                                                                              \
        const uint32_t cx_ = MZ_UNIFORM(((opos & lp_mask) << lc) + (prev_byte >> (8 - lc)));                          \
        uint32_t sl_ = 0;                                                                                             \
        uint32_t hit_ = 0;                                                                                            \
        _Pragma("unroll") for (uint32_t k_ = 0; k_ < MZ_LZMA_SLOTS; k_++) {                                           \
            if (stag[k_] == cx_) {                                                                                    \
                sl_ = k_;                                                                                             \
                hit_ = 1;                                                                                             \
            }                                                                                                         \
        }                                                                                                             \
        if (!hit_) {                                                                                                  \
            /* the slot that was used longest ago makes room (round 3 took them in turn: a fifth context that shows up \
             * now and then -- text with a few bytes above 0x7F -- then threw out a busy one every time, and 30 % of    \
             * config 4's entries were given back to the full-model kernel) */                                        \
            uint32_t old_ = 0, best_ = 0xFFFFFFFFu;                                                                   \
            _Pragma("unroll") for (uint32_t k_ = 0; k_ < MZ_LZMA_SLOTS; k_++) {                                       \
                if (sage[k_] < best_) {                                                                               \
                    best_ = sage[k_];                                                                                 \
                    sl_ = k_;                                                                                         \
                }                                                                                                     \
            }                                                                                                         \
            _Pragma("unroll") for (uint32_t k_ = 0; k_ < MZ_LZMA_SLOTS; k_++) {                                       \
                if (k_ == sl_) {                                                                                      \
                    old_ = stag[k_];                                                                                  \
                    stag[k_] = cx_;                                                                                   \
                }                                                                                                     \
            }                                                                                                         \
            uint16_t *const sp_ = pr + LZ_LIT + 0x300u * sl_;                                                         \
            MZ_LANES {                                                                                                \
                for (uint32_t i_ = (uint32_t)lane; i_ < 0x300u / 4u; i_ += 64u) {                                     \
                    uint64_t v_;                                                                                      \
                    if (old_ != 0xFFFFFFFFu) {                                                                        \
                        __builtin_memcpy(&v_, sp_ + 4u * i_, 8);                                                      \
                        __builtin_memcpy(prx + 0x300u * old_ + 4u * i_, &v_, 8);                                      \
                    }                                                                                                 \
                    __builtin_memcpy(&v_, prx + 0x300u * cx_ + 4u * i_, 8);                                           \
                    __builtin_memcpy(sp_ + 4u * i_, &v_, 8);                                                          \
                }                                                                                                     \
            }                                                                                                         \
            MZ_WAVE_SYNC();                                                                                           \
            if (++sswaps > 64u + (opos >> 7)) { /* more than a swap per 128 bytes: not this kernel's data */          \
                status = MZHIP_RETRY;                                                                                 \
                goto finish;                                                                                          \
            }                                                                                                         \
        }                                                                                                             \
        _Pragma("unroll") for (uint32_t k_ = 0; k_ < MZ_LZMA_SLOTS; k_++)                                             \
            if (k_ == sl_) sage[k_] = opos + 1u; /* (0 = never used) */                                               \
        const uint32_t lbase = LZ_LIT + 0x300u * sl_;                                                                 \
        LZ_LITERAL_PF();                                                                                              \
    } while (0)

/* The packet loop, shared by K3 (LZMA1 to the end marker: lzma2 = 0, dict_start = 0) and the .xz kernel
 * (LZMA2 chunk with a known uncompressed size: lzma2 = 1, stops at opos == chunk_end).  Expects the coder,
 * model and output locals of its caller by name; leaves through `goto finish` with `status` on any failure. */
#define LZ_PACKET_LOOP()                                                                                              \
    for (;;) {                                                                                                        \
        LZ_PRIO_CHAIN();                                                                                              \
        if (eof) goto finish; /* truncated input */                                                                   \
        if (lzma2 && opos == chunk_end) break; /* LZMA2: the chunk's uncompressed size has been produced */           \
        LZ_RESUME_CHECK();                                                                                            \
        const uint32_t ps = opos & pb_mask;                                                                           \
        uint32_t bit;                                                                                                 \
        LZ_BIT(bit, LZ_IS_MATCH + state * 16 + ps);                                                                   \
        if (!bit) {                                                                                                   \
            /* literal */                                                                                             \
            uint32_t sym = 1;                                                                                         \
            LZ_LITERAL_SITE(sym);                                                                                     \
            if (eof) goto finish;                                                                                     \
            if (opos == out_cap) {                                                                                    \
                status = MZHIP_OUT_FULL;                                                                              \
                goto finish;                                                                                          \
            }                                                                                                         \
            MZ_LANES { out[opos] = (uint8_t)sym; } /* uniform store */                                                \
            MZ_WAVE_SYNC();                                                                                           \
            prev_byte = sym & 0xFFu;                                                                                  \
            opos++;                                                                                                   \
            state = state < 4 ? 0 : (state < 10 ? state - 3 : state - 6);                                             \
            if ((opos & (MZ_CRC_TILE - 1)) == 0) { LZ_PRIO_BULK(); MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, LZ_CRC_LIMIT(opos), crc_tab, tabs->kx); LZ_PRIO_CHAIN(); }  \
            continue;                                                                                                 \
        }                                                                                                             \
        uint32_t len;                                                                                                 \
        LZ_BIT(bit, LZ_IS_REP + state);                                                                               \
        if (bit) {                                                                                                    \
            if (opos == dict_start) goto finish; /* rep with an empty dictionary */                                   \
            LZ_BIT(bit, LZ_IS_REP_G0 + state);                                                                        \
            if (!bit) {                                                                                               \
                LZ_BIT(bit, LZ_IS_REP0_LONG + state * 16 + ps);                                                       \
                if (!bit) {                                                                                           \
                    /* short rep: one byte from rep0 */                                                               \
                    if (eof) goto finish;                                                                             \
                    if (rep0 >= opos - dict_start || rep0 >= dict) goto finish;                                       \
                    if (opos == out_cap) {                                                                            \
                        status = MZHIP_OUT_FULL;                                                                      \
                        goto finish;                                                                                  \
                    }                                                                                                 \
                    uint32_t b = LZ_U(out[opos - rep0 - 1]);                                                          \
                    MZ_LANES { out[opos] = (uint8_t)b; } /* uniform store */                                          \
                    MZ_WAVE_SYNC();                                                                                   \
                    prev_byte = b;                                                                                    \
                    opos++;                                                                                           \
                    match_byte = LZ_U(out[opos - rep0 - 1]);                                                          \
                    state = state < 7 ? 9 : 11;                                                                       \
                    if ((opos & (MZ_CRC_TILE - 1)) == 0)                                                              \
                        { LZ_PRIO_BULK(); MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, LZ_CRC_LIMIT(opos), crc_tab, tabs->kx); LZ_PRIO_CHAIN(); }                           \
                    continue;                                                                                         \
                }                                                                                                     \
            } else {                                                                                                  \
                uint32_t dist;                                                                                        \
                LZ_BIT(bit, LZ_IS_REP_G1 + state);                                                                    \
                if (!bit) {                                                                                           \
                    dist = rep1;                                                                                      \
                } else {                                                                                              \
                    LZ_BIT(bit, LZ_IS_REP_G2 + state);                                                                \
                    if (!bit) {                                                                                       \
                        dist = rep2;                                                                                  \
                    } else {                                                                                          \
                        dist = rep3;                                                                                  \
                        rep3 = rep2;                                                                                  \
                    }                                                                                                 \
                    rep2 = rep1;                                                                                      \
                }                                                                                                     \
                rep1 = rep0;                                                                                          \
                rep0 = dist;                                                                                          \
            }                                                                                                         \
            LZ_LEN_DECODE(len, LZ_REP_LEN, ps);                                                                       \
            state = state < 7 ? 8 : 11;                                                                               \
        } else {                                                                                                      \
            rep3 = rep2;                                                                                              \
            rep2 = rep1;                                                                                              \
            rep1 = rep0;                                                                                              \
            LZ_LEN_DECODE(len, LZ_LEN, ps);                                                                           \
            state = state < 7 ? 7 : 10;                                                                               \
            uint32_t slot;                                                                                            \
            LZ_BITTREE_PF(slot, LZ_POS_SLOT + (len < 4 ? len : 3u) * 64, 6);                                             \
            if (slot < 4) {                                                                                           \
                rep0 = slot;                                                                                          \
            } else {                                                                                                  \
                const uint32_t nb = (slot >> 1) - 1;                                                                  \
                rep0 = (2u | (slot & 1u)) << nb;                                                                      \
                uint32_t low;                                                                                         \
                if (slot < 14) {                                                                                      \
                    LZ_BITTREE_REV(low, LZ_POS_DEC + rep0 - slot, nb);                                                \
                    rep0 += low;                                                                                      \
                } else {                                                                                              \
                    uint32_t direct = 0;                                                                              \
                    for (uint32_t i = 0; i < nb - 4; i++) {                                                           \
                        LZ_NORM();                                                                                    \
                        range >>= 1;                                                                                  \
                        code -= range;                                                                                \
                        uint32_t t = 0u - (code >> 31);                                                               \
                        code += range & t;                                                                            \
                        direct = (direct << 1) + (t + 1);                                                             \
                    }                                                                                                 \
                    rep0 += direct << 4;                                                                              \
                    LZ_BITTREE_REV_PF(low, LZ_ALIGN, 4);                                                                 \
                    rep0 += low;                                                                                      \
                }                                                                                                     \
            }                                                                                                         \
            if (rep0 == 0xFFFFFFFFu) {                                                                                \
                /* end-of-stream marker */                                                                            \
                if (eof) goto finish;                                                                                 \
                if (lzma2) goto finish; /* not allowed when the size is known */                                      \
                LZ_NORM();                                                                                            \
                if (eof) goto finish;                                                                                 \
                status = (code == 0) ? MZHIP_OK : MZHIP_DATA_ERROR;                                                   \
                goto finish;                                                                                          \
            }                                                                                                         \
        }                                                                                                             \
        if (eof) goto finish;                                                                                         \
        len += 2;                                                                                                     \
        if (rep0 >= opos - dict_start || rep0 >= dict) goto finish; /* distance beyond the dictionary */              \
        if (lzma2 && len > chunk_end - opos) goto finish; /* match runs past the chunk's uncompressed size */         \
        {                                                                                                             \
            uint32_t n = len;                                                                                         \
            int full = 0;                                                                                             \
            if (n > out_cap - opos) {                                                                                 \
                n = out_cap - opos;                                                                                   \
                full = 1;                                                                                             \
            }                                                                                                         \
            LZ_PRIO_BULK();                                                                                           \
            const uint32_t dist = rep0 + 1;                                                                           \
            const uint8_t *src = out + (opos - dist);                                                                 \
            if (dist >= n) {                                                                                          \
                MZ_LANES {                                                                                            \
                    for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[opos + i] = src[i];                         \
                }                                                                                                     \
            } else {                                                                                                  \
                MZ_LANES {                                                                                            \
                    for (uint32_t i = (uint32_t)lane; i < n; i += 64) out[opos + i] = src[i % dist];                  \
                }                                                                                                     \
            }                                                                                                         \
            MZ_WAVE_SYNC();                                                                                           \
            opos += n;                                                                                                \
            if (full) {                                                                                               \
                status = MZHIP_OUT_FULL;                                                                              \
                goto finish;                                                                                          \
            }                                                                                                         \
            prev_byte = LZ_U(out[opos - 1]);                                                                          \
            match_byte = LZ_U(out[opos - dist]);                                                                      \
            { LZ_PRIO_BULK(); MZ_CRC_FOLD_TILES(crc_acc, crc_done, out, LZ_CRC_LIMIT(opos), crc_tab, tabs->kx); LZ_PRIO_CHAIN(); }                                       \
        }                                                                                                             \
    }

/* Two builds of the same decoder, differing only in where the wave-uniform coder state lives:
 *   mz_lzma_entry    LZ_U = readfirstlane: range / code / probabilities in SGPRs, the decision arithmetic issues on
 *                    the CU's scalar port;
 *   mz_lzma_entry_v  LZ_U = identity: the same values stay in VGPRs (all lanes equal), the arithmetic issues on the
 *                    vector ports of the four SIMDs, only the uniform branches remain scalar.
 * Measured for one full round of 2304 resident 1 MiB entries: scalar 520 ms, 5 of 8 workgroups on the vector build 511 ms, all
 * of them 469 ms (and mixes inside a SIMD 9.73 / 9.50 / 8.35 GiB/s against 10.34 on config 4, profiles/r4/ab_k3_ports.log): the
 * kernels run the vector build.  The host emulation builds the first only. */
/* A decision on a lane-invariant value that sits in a VGPR (the vector-port builds): asked through a ballot, the
 * compiler knows the branch is uniform -- one side is executed (s_cbranch) instead of both under exec masks, and what
 * hangs on the decided bit (the symbol, the next probability's index, the state) moves to the scalar unit by itself. */
#ifndef MZ_LZMA_UBR
#define MZ_LZMA_UBR 1
#endif
#if MZ_LZMA_UBR && !defined(MZHIP_HOST_EMUL)
#define MZ_VEC_UBR(c) (__ballot(c) != 0ull)
#else
#define MZ_VEC_UBR(c) (c)
#endif
/* (With the probability on the scalar unit as well -- its update three scalar instructions instead of three to five vector
 * ones -- config 4 ran at 9.26 GiB/s against 10.03: the transfers cost more than they save, profiles/r4/ab_k3_ubr.log.) */
#define LZ_LITERAL_SITE(sym) LZ_LITERAL_SITE_FULL(sym)
#define LZ_LDS_T mz_lzma_lds
#define LZ_ENTRY_PROBS LZ_NUM_PROBS
#define LZ_U(x) MZ_UNIFORM(x)
#define LZ_UBR(c) (c)
#define LZ_WIN_DW(idx) MZ_READLANE(win, idx)
#define LZ_ENTRY_NAME mz_lzma_entry
#include "lzma_entry.inc"
#undef LZ_U
#undef LZ_UBR
#undef LZ_WIN_DW
#undef LZ_ENTRY_NAME

#if !defined(MZHIP_HOST_EMUL)
#define LZ_U(x) (x)
#define LZ_UBR(c) MZ_VEC_UBR(c)
#define LZ_WIN_DW(idx) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)win))
#define LZ_ENTRY_NAME mz_lzma_entry_v
#include "lzma_entry.inc"
#undef LZ_U
#undef LZ_UBR
#undef LZ_WIN_DW
#undef LZ_ENTRY_NAME
#endif

/* the slot build: vector-port form on the device, scalar form in the host emulation */
#undef LZ_LITERAL_SITE
#undef LZ_LDS_T
#undef LZ_ENTRY_PROBS
#define LZ_LITERAL_SITE(sym) LZ_LITERAL_SITE_SLOT(sym)
#define LZ_LDS_T mz_lzma_lds_s
#define LZ_ENTRY_PROBS LZ_NUM_PROBS_S
#define LZ_SLOTS_BUILD 1
#if defined(MZHIP_HOST_EMUL)
#define LZ_U(x) MZ_UNIFORM(x)
#define LZ_UBR(c) (c)
#define LZ_WIN_DW(idx) MZ_READLANE(win, idx)
#else
#define LZ_U(x) (x)
#define LZ_UBR(c) MZ_VEC_UBR(c)
#define LZ_WIN_DW(idx) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)win))
#endif
#define LZ_ENTRY_NAME mz_lzma_entry_s
#include "lzma_entry.inc"
#undef LZ_U
#undef LZ_UBR
#undef LZ_WIN_DW
#undef LZ_ENTRY_NAME
#undef LZ_SLOTS_BUILD
#undef LZ_LITERAL_SITE
#undef LZ_LDS_T
#undef LZ_ENTRY_PROBS
#define LZ_LITERAL_SITE(sym) LZ_LITERAL_SITE_FULL(sym)

/* the resumable build (full model; vector-port form on the device, scalar form in the host emulation): stops in front of a
 * packet when the output buffer or -- unless it is the stream's last -- the input runs low */
#if defined(MZHIP_HOST_EMUL)
#define LZ_U(x) MZ_UNIFORM(x)
#define LZ_UBR(c) (c)
#define LZ_WIN_DW(idx) MZ_READLANE(win, idx)
#else
#define LZ_U(x) (x)
#define LZ_UBR(c) MZ_VEC_UBR(c)
#define LZ_WIN_DW(idx) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)win))
#endif
#define LZ_LDS_T mz_lzma_lds
#define LZ_ENTRY_PROBS LZ_NUM_PROBS
#define LZ_RESUME_BUILD 1
#undef LZ_RESUME_CHECK
#define LZ_RESUME_CHECK()                                                                                   \
    if (st && (opos + 274u > out_cap || (!last_input && in_pos + 64u > rc_len))) {                          \
        status = (opos + 274u > out_cap) ? MZHIP_OUT_FULL : MZHIP_BUF_ERROR;                                \
        goto stop_here;                                                                                     \
    }
#pragma push_macro("LZ_CRC_LIMIT")
#undef LZ_CRC_LIMIT
#define LZ_CRC_LIMIT(o) 0u /* no fused CRC in this build */
#define LZ_ENTRY_NAME mz_lzma_entry_r
#include "lzma_entry.inc"
#undef LZ_ENTRY_NAME
#pragma pop_macro("LZ_CRC_LIMIT")
#undef LZ_RESUME_CHECK
#define LZ_RESUME_CHECK() ((void)0)
#undef LZ_RESUME_BUILD
#undef LZ_U
#undef LZ_UBR
#undef LZ_WIN_DW
#undef LZ_LDS_T
#undef LZ_ENTRY_PROBS

/* for code that expands the coder macros outside the two entry builds (xz_core.h): vector-port forms on the device */
#if defined(MZHIP_HOST_EMUL)
#define LZ_U(x) MZ_UNIFORM(x)
#define LZ_UBR(c) (c)
#define LZ_WIN_DW(idx) MZ_READLANE(win, idx)
#else
#define LZ_U(x) (x)
#define LZ_UBR(c) MZ_VEC_UBR(c)
#define LZ_WIN_DW(idx) ((uint32_t)__builtin_amdgcn_ds_bpermute((int)((idx) << 2), (int)win))
#endif

#endif
