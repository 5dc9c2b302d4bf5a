/* inflate_core.h -- raw-DEFLATE decode of ONE entry by ONE wavefront, with the
 * entry's CRC-32 fused in (kernel K1+K2 of SURVEY 2.1).
 *
 * Replaces what the reference does per entry through mz_stream_zlib_read
 * (mz_strm_zlib.c:116-193 -> zlib inflate(), raw, 32 KiB window) followed by
 * mz_crypt_crc32_update (mz_zip.c:2049).  Format: doc/zip/appnote.txt:2030-2166.
 *
 * MI355X mapping (not a port of zlib's byte-serial state machine):
 *   - Huffman tables are built by the whole wave from code lengths held in registers (a ballot per length gives every
 *     symbol its rank, two wave scans the first codes -> LDS lookup tables whose 32-bit entries are ready-made tokens /
 *     operand descriptors), see inflate_tables.inc.
 *   - Two decode front ends:
 *       CHASE WINDOW (inflate_chase.inc = inflate_walk.inc + the chain + inflate_emit.inc; the default for all but what
 *       needs a verdict): the rest of the stream is cut into <= 64 spans of up to 3072 bits, one per lane.  Pass 1: every
 *       lane walks its span token by token (mz_chase_step: a leading literal + the token behind it per step) and RECORDS
 *       every step -- 4 bytes + 1 byte, in the wave's HBM scratch.  Pass 2: every lane keeps walking into the next span
 *       until it stands on a step start the lane there recorded too (a DEFLATE walk started at an arbitrary bit falls in
 *       step with the true parse after ~8 tokens).  The chain from lane 0 then names the true records; they are taken 512
 *       at a time, turned into staging bytes / back-reference pieces in the LDS pool, far sources fetched from the output
 *       buffer, near ones copied LDS to LDS in dependency order, then one coalesced 16-byte store per lane.  Every token
 *       is Huffman-decoded once (+ the resynchronisation tail of a span).  The two halves are functions of their own
 *       (mz_chase_walk, mz_chase_emit: not inlined, their own register allocations).
 *       STEP LOOP (the last bits of a stream, hand-backs, and every error verdict): lane l decodes the complete token
 *       that would start at bit cursor+l, for all 64 bit offsets at once; which candidates are real is
 *       decided without a serial walk: f(l) = l + bits(l) is squared with cross-lane gathers and lane i
 *       composes f^i(0).  One step retires about 64 bits of input (4.6 tokens on text).  Its tokens are handed to
 *       the flush (inflate_flush.inc) 64 at a time, one per lane: wave prefix sum of the sizes, literals scatter in
 *       one store, matches are copied eight at a time, 8 lanes each, as long as every source of a group ends before
 *       the group's first destination byte; overlapping (dist < len) runs and in-group dependencies take an
 *       in-order cooperative copy.
 *   - The sliding window IS the output buffer: back-references read bytes this
 *     wave wrote earlier (a wave's vector-memory operations execute in order),
 *     so no 32 KiB LDS window is needed (16 waves per CU x 32 KiB would be 3 CUs' worth of LDS, and at 4 waves per CU
 *     the kernel loses a third of its throughput: profiles/r4/ab_k1_residency.log); the step loop stages compressed
 *     input through a 512-byte LDS ring with a register-held prefetch.
 *   - CRC-32 is folded from the freshly written output in 4 KiB super-tiles, 64 bytes per lane (crc32_core.h).
 *   - The per-lane decode is kept to 32-bit funnel shifts (v_alignbit), bit-field extracts and table
 *     entries that need no arithmetic, and LDS is sized for residency: 9.98 KiB per wave (3.8 KiB of tables with an
 *     8-bit literal/length root, 4.75 KiB of per-lane stream rings, a 1.2 KiB pool that runs on through the dead rings
 *     while a window is emitted) = 16 waves per CU, matching the 128-VGPR budget of 4 waves per SIMD.
 *   - Dynamic block headers: the code lengths are decoded 64 bits at a time (MZ_CL_PARALLEL).
 *   - The 32-bit bit cursor looks at the stream through a view that moves forward (MZ_REBASE): any stream length.
 *   - MZ_STATS (host emulation only) counts windows / passes / SIMT steps / pieces; MZ_PROF (device measurement
 *     builds) sums the cycles a wave spends per section; MZ_ABLATE removes stages (wrong output, timing only).
 *
 * Error classes mirror zlib's as the reference surfaces them
 * (mz_strm_zlib.c:159-189): malformed data -> -3, input exhausted -> -5.
 */
#ifndef MZHIP_INFLATE_CORE_H
#define MZHIP_INFLATE_CORE_H
#if defined(MZ_STATS)
extern unsigned long long mz_stats[24], mz_stat_max;
#define MZ_STAT(i, v) (mz_stats[i] += (v))
#define MZ_STAT_LANE(v) (mz_stat_max = (v) > mz_stat_max ? (v) : mz_stat_max) /* the slowest lane ... */
#define MZ_STAT_LANEMAX(i) (mz_stats[i] += mz_stat_max, mz_stat_max = 0)       /* ... is what the wave pays */
#else
#define MZ_STAT(i, v) ((void)0)
#define MZ_STAT_LANE(v) ((void)0)
#define MZ_STAT_LANEMAX(i) ((void)0)
#endif

#include "crc32_core.h"
#include "wave.h"
/* MZ_PROF (measurement builds of the device code only, profiles/gpu.sh build "prof PROF=1"): every wave adds the shader-clock cycles it
 * spent in each section of mz_inflate_entry to mz_prof_buf[]; mzhip_prof_read() hands the sums to tests/perf_probe.py */
#if defined(MZ_PROF) && !defined(MZHIP_HOST_EMUL)
extern __device__ unsigned long long mz_prof_buf[32];
#define MZ_PROF_DECL                     \
    uint32_t prof_acc = 0; /* lane i: cycles of section i */ \
    uint64_t prof_t0 = __builtin_readcyclecounter();
#define MZ_PROF_MARK(i)                                                           \
    do {                                                                          \
        const uint32_t _pd = (uint32_t)(__builtin_readcyclecounter() - prof_t0);  \
        prof_acc += (lane == (i)) ? _pd : 0u;                                     \
        prof_t0 = __builtin_readcyclecounter();                                   \
    } while (0)
#define MZ_PROF_FLUSH                                                             \
    if (lane < 32) atomicAdd(&mz_prof_buf[lane], (unsigned long long)prof_acc);
#else
#define MZ_PROF_DECL
#define MZ_PROF_MARK(i) MZ_PRIO_AT(i)
#define MZ_PROF_FLUSH
#endif
/* The wave's issue priority by section (s_setprio; round 6, profiles/r6/ab_k1_setprio.log).  Two bits per section mark i =
 * the priority of what runs BEHIND mark i (14: the start of pass 1, 15: a block header): the walks 3, near copies 2, far
 * copies / store / CRC 1, the rest 0.  What it buys is the START of a launch: the four waves of a SIMD begin their entries
 * together and stand in the same section at the same time, a chain of dependent LDS look-ups queueing behind three others
 * like it; with priorities the waves fall out of step at once.  20 000 x 64 KiB entries (five rounds of the resident waves)
 * 3.91 -> 3.69 ms (+5.6 %), 200 000 x 8 KiB 7.27 -> 7.13 ms (+2 %); any split with the walks on top is within 1 % of this
 * one.  Over the 24 rounds of config 2 the waves have drifted apart by themselves: 16.04 -> 15.96 ms (+0.5 %), config 3
 * +1.3 %.  It is the short launches that gain: a chunk of a prime, a window of a large entry, a rank's share of a table cut
 * eight ways.  -DMZ_PRIO_MAP=0 is the build without. */
#ifndef MZ_PRIO_MAP
#define MZ_PRIO_MAP 0x34016330ull
#endif
#if MZ_PRIO_MAP && !defined(MZHIP_HOST_EMUL) && !defined(MZ_PROF)
#define MZ_PRIO_AT(i) __builtin_amdgcn_s_setprio((int)(((MZ_PRIO_MAP) >> (2 * (i))) & 3ull))
#else
#define MZ_PRIO_AT(i) ((void)0)
#endif

#ifndef MZ_LROOT
#define MZ_LROOT 8 /* literal/length fast-table index bits */
#endif
#ifndef MZ_DROOT
#define MZ_DROOT 8 /* distance fast-table index bits        */
#endif
#define MZ_CROOT 7 /* code-length-code table bits (== max)  */
/* ---- chase window: spans of up to MZ_CHASE_SMAX bits, one per lane; the stream reaches a lane through
 * its own ring of MZ_CRING_DW dwords in LDS (+ 2 that mirror the first two, so dword a, a + 1, a + 2 are one address),
 * topped up 4 dwords at a time every second step from a prefetch register; a step record is two dwords
 * (mz_chase_step), four records per 16-byte store, row-major [quad of steps][lane] so that the wave's stores of one quad are
 * 1 KiB contiguous. */
#ifndef MZ_CHASE_SMAX
#define MZ_CHASE_SMAX 3072u /* bits per span at most: ~170 steps on text, under the record cap below */
#endif
#ifndef MZ_REC_CAP1
#define MZ_REC_CAP1 256u    /* steps a lane may record on its own span (a multiple of 16) ... */
#endif
#ifndef MZ_REC_CAP2
#define MZ_REC_CAP2 160u    /* ... and while chasing into the next one(s) (a multiple of 4): 0.04 % of the chases on text are longer than 128 */
#endif
#define MZ_REC_AREA(cap_) (((cap_) / 4u) * 1024u) /* records: 4 bytes a step, four steps of a lane = one 16-byte quad, [quad][lane] (inflate_chase.inc MZ_REC_ADDR) */
#define MZ_REC_BYTES (MZ_REC_AREA(MZ_REC_CAP1) + MZ_REC_AREA(MZ_REC_CAP2) + 64u * MZ_REC_CAP1 + 64u * MZ_REC_CAP2 + 1024u) /* HBM scratch per wave: records + a byte per step */
#define MZ_REC_ADDR(quad_, lane_) ((((quad_) * 64u) + (lane_)) << 4) /* where the quad lives inside its area: [quad][lane], a store of the wave is 1 KiB
                                                                     contiguous.  (Round 5 timed a row per lane and [2 / 4 quads][lane]: the L2's memory-side
                                                                     reads fall 7 - 11 %, its writes grow 40 - 60 %, the probe is 0.3 - 7 % slower: profiles/r5/call4_probe.log) */
#define MZ_CRING_DW 16u /* a power of two: the slot of a stream dword is its index & 15, nothing to keep track of */
#define MZ_CRING_RS 19u /* row stride: 16 + 2 mirrored, odd */
#ifndef MZ_PF_GROUPS
#define MZ_PF_GROUPS 2 /* groups of four stream dwords a lane loads at a time (inflate_walk.inc): 1 or 2 */
#endif
#ifndef MZ_EMIT_GROUP
#define MZ_EMIT_GROUP 8u /* records one lane turns into bytes per emit round (4 or 8: records are stored in quads) */
#endif
#ifndef MZ_POOL_BYTES
#define MZ_POOL_BYTES 1264u /* LDS pool of one emit round = staging bytes of its output (from the front, at most 4 KiB), a pending
                               bit per byte, its back-reference list, 4 bytes each (from the back): what the rings leave of the
                               9984 bytes a wave may have at 16 waves per CU (it runs on through the dead rings while a window is
                               emitted; 768 bytes less changed nothing on the probe: profiles/r5/call6_probe.log) */
#endif
#ifndef MZ_LDS_PAD
#define MZ_LDS_PAD 0
#endif
#ifndef MZ_VIEW_MAX
#define MZ_VIEW_MAX 0x0FFFFFFFu   /* bytes of the stream the 32-bit bit cursor can address at a time (see mz_inflate_entry) */
#define MZ_REBASE_BITS (1u << 30) /* the view moves when the cursor is this far into it */
#endif
#ifndef MZ_CL_PARALLEL
#define MZ_CL_PARALLEL 1 /* dynamic block headers: code lengths decoded 64 bits at a time (0: one symbol at a time) */
#endif
#ifndef MZ_ABLATE
#define MZ_ABLATE 0 /* measurement builds only (wrong output): 1 no match copies, 2 no far loads, 4 no CRC, 8 no store */
#endif
#ifndef MZ_MLANES_LOG2
#define MZ_MLANES_LOG2 3 /* step-loop flush: lanes that copy one match together: 2^3 = 8, so 8 matches per round */
#endif

/* ---- table entry formats (32-bit) -------------------------------------------------------------
 * token (what a candidate decodes to):
 *     [5:0] total bits (0 = no valid token starts here)   [6] end-of-block
 *     [15:7] bytes produced (1 literal, 3..258 match, 0 end-of-block)
 *     [31:16] literal byte | match distance | (invalid) number of bits the verdict needed
 * literal/length table entry = descriptor + code length in [5:0]; 0 = no code of <= root bits here:
 *     literal      : 0x80 | byte << 16                    (already the finished token)
 *     end-of-block : 0x40                                 (already the finished token)
 *     length       : MZ_E_LEN | base << 7 | extra_bits << 16
 *     286, 287     : MZ_E_BAD as a descriptor; finished entries hold code length << 16 (an invalid token)
 *     long code    : MZ_E_SUB | sub-table offset << 8 | sub-table index bits   (root table only; the
 *                    sub-table entry, indexed by the next bits of the stream, is one of the above)
 * distance table entry = descriptor + code length in [3:0]; 0 = none:
 *     distance     : extra_bits << 4 | base << 8
 *     30, 31       : MZ_E_LEN (reused as the "invalid" mark)
 * code-length-code entry: symbol << 4 | length. */
#define MZ_E_LEN 0x80000000u
#define MZ_E_BAD 0x40000000u
#define MZ_E_SUB 0x20000000u
/* Second-level entries a complete canonical literal/length code (<= 286 symbols, <= 15 bits) can need: the maximum of
 * sum over root prefixes of 2^(longest code in the prefix - root), found by exhaustive dynamic programming over the
 * code-length counts (the same search as zlib's examples/enough.c; it reproduces zlib's 852 = 512 + 340 for a 9-bit
 * root and 592 = 64 + 528 for the 6-bit / 30-symbol distance case). */
#if MZ_LROOT == 8
#define MZ_LIT_SUB_ENTRIES 404
#elif MZ_LROOT == 9
#define MZ_LIT_SUB_ENTRIES 340
#elif MZ_LROOT == 10
#define MZ_LIT_SUB_ENTRIES 308
#else
#error "MZ_LROOT must be 8, 9 or 10"
#endif

MZ_DEV uint32_t mz_lit_ent(uint32_t s) {
    if (s < 256u) return 0x80u | (s << 16);
    if (s == 256u) return 0x40u;
    if (s > 285u) return MZ_E_BAD;
    s -= 257u; /* length base / extra bits, appnote.txt:2107-2120, computed */
    if (s < 8u) return MZ_E_LEN | ((3u + s) << 7);
    if (s == 28u) return MZ_E_LEN | (258u << 7);
    const uint32_t ex = (s - 4u) >> 2;
    return MZ_E_LEN | ((3u + ((4u + (s & 3u)) << ex)) << 7) | (ex << 16);
}
MZ_DEV uint32_t mz_dist_ent(uint32_t s) {
    if (s > 29u) return MZ_E_LEN;
    if (s < 4u) return (1u + s) << 8; /* distance base / extra bits, appnote.txt:2122-2133, computed */
    const uint32_t ex = (s - 2u) >> 1;
    return (ex << 4) | ((1u + ((2u + (s & 1u)) << ex)) << 8);
}
MZ_DEV uint32_t mz_clc_ent(uint32_t s) { return s << 4; }
/* finished table entry: descriptor + code length; the invalid symbols 286 / 287 become an invalid token right here
 * (0 bits, the code length in [31:16] as "bits the verdict needed"), so the decode loop has no case for them */
MZ_DEV uint32_t mz_fin_ent(uint32_t e, uint32_t len) { return (e & MZ_E_BAD) ? (len << 16) : e + len; }

/* per-wave LDS scratch */
typedef struct mz_inflate_hdr_scratch { /* live while a block header is parsed (the tables themselves are built from registers: inflate_tables.inc) */
    uint16_t clc_fast[1 << MZ_CROOT]; /* the code-length code: symbol << 4 | bits at every 7-bit index */
    uint8_t cl[320];     /* code lengths of the current dynamic block (nlen + ndist <= 316), as the run-length decode leaves them */
} mz_inflate_hdr_scratch;

typedef struct mz_inflate_body_scratch { /* live while the block body is decoded */
    union { /* first: the pool's 16-byte chunks line up with 16-byte stores (the struct sits on a 16-byte boundary) */
        struct {
            uint32_t ring[130]; /* step loop: 512 B of compressed stream: aligned dword j of the entry at ring[j & 127];
                                   entries 128, 129 mirror 0, 1 so that a window read is one address plus constant offsets */
            uint16_t mslot[64]; /* step-loop flush: 4 * lane id of this step's match tokens, compacted */
        } s;
        uint32_t pool[MZ_POOL_BYTES / 4]; /* span path (never live together with the step loop's ring) */
    } x;
    uint32_t win[64 * MZ_CRING_RS]; /* chase window: one ring row per lane while the lanes walk; the chain's per-lane tables
                                       (steps, entry points, group counts) while the records become bytes */
} mz_inflate_body_scratch;
typedef char mz_pool_is_16_byte_granular[(MZ_POOL_BYTES % 16u == 0u && MZ_POOL_BYTES < 65536u && (MZ_LIT_SUB_ENTRIES % 4) == 0 &&
                                          ((1 << MZ_LROOT) % 4) == 0) ? 1 : -1];
#define MZ_L_WIN(L_) ((L_)->u.b.win)
#define MZ_L_RING(L_) ((L_)->u.b.x.s.ring)
#define MZ_L_MSLOT(L_) ((L_)->u.b.x.s.mslot)

typedef struct mz_inflate_lds { /* sits on a 16-byte boundary; every member below starts on one */
    uint32_t lit_fast[1 << MZ_LROOT];
    uint32_t dist_fast[1 << MZ_DROOT];
    uint32_t dist_ent[32]; /* distance descriptors in canonical (length, symbol) order, for codes > root */
    uint16_t dist_lim[16]; /* left-justified 15-bit upper bound of the distance codes of each length */
    int16_t dist_delta[16]; /* rank offset - first code, per length */
    uint32_t lit_sub[MZ_LIT_SUB_ENTRIES]; /* second-level tables for literal/length codes longer than the root
                                             (during the build: descriptors in canonical order).  Sized for the worst
                                             code; the entries a block does not use extend the span path's pool, which
                                             follows directly (u.b.x.pool) */
    union {
        mz_inflate_hdr_scratch h;
        mz_inflate_body_scratch b;
    } u;
#if MZ_LDS_PAD
    uint32_t pad[MZ_LDS_PAD / 4]; /* measurement builds only: occupancy study */
#endif
} mz_inflate_lds;

typedef struct mz_inflate_result {
    int32_t status;
    uint32_t out_len;
    uint32_t in_used;
    uint32_t crc;
} mz_inflate_result;

/* Where a decode may be taken up again (streams decoded window by window, mzhip_inflate_resume_host): a token boundary
 * of the stream, the block it lies in, and how much of the output buffer is history.  The tables of a Huffman block are
 * rebuilt from its header, so a position inside a block is (bit position of the block's header, bit position of the next
 * token); at the start of a block the two are equal.  Bit positions count from the first byte the call is given.
 * mzhip.h declares the same four words as mzhip_inflate_state. */
typedef struct mz_inflate_state {
    uint32_t hdr_bit;  /* the current block's header */
    uint32_t bit;      /* the next token (== hdr_bit: the block has not been entered) */
    uint32_t out_pos;  /* in: bytes of history in front of the output (<= 32768, 0 at the start of a stream); out: bytes valid in the buffer */
    uint32_t flags;    /* bit 0: take the stream up at (hdr_bit, bit) instead of at bit 0; bit 1 (in): stop in front of the next
                        * block header (MZHIP_OUT_FULL with hdr_bit == bit): the caller wants to go on from a block boundary */
} mz_inflate_state;

/* 32 bits of (hi:lo) starting at bit (s & 31): v_alignbit_b32 */
MZ_DEV uint32_t mz_funnel(uint32_t hi, uint32_t lo, uint32_t s) {
#if defined(MZHIP_HOST_EMUL)
    s &= 31u;
    return s ? ((lo >> s) | (hi << (32u - s))) : lo;
#else
    return __builtin_amdgcn_alignbit(hi, lo, s);
#endif
}
/* bits [off, off+width) of x, width 0 -> 0: v_bfe_u32 */
MZ_DEV uint32_t mz_bfe(uint32_t x, uint32_t off, uint32_t width) {
#if defined(MZHIP_HOST_EMUL)
    return width ? ((x >> off) & (0xFFFFFFFFu >> (32u - width))) : 0u;
#else
    return __builtin_amdgcn_ubfe(x, off, width);
#endif
}

/* 64 bits of the stream starting at bit `bitpos`, LSB first, zero-padded past
 * the end (block headers only; the body reads the LDS ring). */
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

/* Branch-free search for a code longer than `root` bits (tables: inflate_tables.inc).  With lim[L] = (first[L] + count[L]) << (15 - L)
 * a 15-bit left-justified stream value v carries a code of length L iff lim[L-1] <= v < lim[L]; its
 * descriptor is ent[delta[L] + (v >> (15 - L))].  Returns descriptor + length, or 0 when no code matches
 * (an unused code of an incomplete set). */
MZ_DEV uint32_t mz_long_code(uint32_t lo, int root, const uint16_t *lim, const int16_t *delta, const uint32_t *ent,
                             uint32_t nent) {
    const uint32_t v15 = mz_brev32(lo) >> 17;
    uint32_t len = (uint32_t)root + 1u;
#pragma unroll
    for (int k = root + 1; k < 15; k++) len += (v15 >= lim[k]) ? 1u : 0u;
    const uint32_t idx = (uint32_t)((int32_t)delta[len] + (int32_t)(v15 >> (15u - len)));
    const uint32_t ok = (v15 < lim[15] && idx < nent) ? 1u : 0u;
    const uint32_t e = ent[ok ? idx : 0u];
    return ok ? e + len : 0u;
}

/* 64 bits of the stream at (uniform) bit position pos_: from the header window hwin (dword hw0 + lane of the stream,
 * bytes outside the input already zero) when the three dwords lie inside it, else from memory -- same value either way */
#define MZ_HDR_WIN_BITS(dst, pos_)                                                            \
    do {                                                                                      \
        const uint32_t _ab = (pos_) + 8u * in_mis, _j = (_ab >> 5) - hw0;                     \
        if (_j + 2u < 64u) {                                                                  \
            const uint32_t _d0 = MZ_READLANE(hwin, _j), _d1 = MZ_READLANE(hwin, _j + 1u), _d2 = MZ_READLANE(hwin, _j + 2u); \
            const uint32_t _s = _ab & 31u;                                                    \
            uint64_t _v = (((uint64_t)_d1 << 32) | _d0) >> _s;                                \
            if (_s) _v |= (uint64_t)_d2 << (64u - _s);                                        \
            (dst) = _v;                                                                       \
        } else {                                                                              \
            (dst) = mz_bits_at(in, in_len, (pos_));                                           \
        }                                                                                     \
    } while (0)

/* uniform n-bit read at the block-header level */
#define MZ_HDR_BITS(dst, n)                                                   \
    do {                                                                      \
        if (bitpos + (uint32_t)(n) > total_bits) {                            \
            status = MZHIP_BUF_ERROR;                                         \
            goto finish;                                                      \
        }                                                                     \
        uint64_t _w;                                                          \
        MZ_HDR_WIN_BITS(_w, bitpos);                                          \
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


/* bits [sh, sh + n) of a 64-bit window as two words, 0 <= sh <= 31, 1 <= n <= 32 (the pending bits of one piece) */
MZ_DEV void mz_bitrange(uint32_t sh, uint32_t n, uint32_t *m0, uint32_t *m1) {
    const uint32_t full = 0xFFFFFFFFu >> (32u - n);
    *m0 = full << sh;
    *m1 = (full >> 1) >> (31u - sh);
}
/* n <= 32 bytes from src to dst; the two ranges do not overlap; every load is issued before the first store, so a
 * source in global memory costs one round trip.  In two halves, so that a lane can have the loads of several pieces
 * in flight before it stores the first. */
typedef struct {
    uint64_t v[4];
    uint32_t t4, t2, t1;
} mz_piece32;
MZ_DEV void mz_copy32_load(mz_piece32 *r, const uint8_t *src, uint32_t n) {
    const uint32_t n8 = n & ~7u;
    r->v[0] = r->v[1] = r->v[2] = r->v[3] = 0;
    r->t4 = r->t2 = r->t1 = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++)
        if (8u * k + 8u <= n) r->v[k] = mz_ld8(src + 8u * k);
    if (n & 4u) r->t4 = mz_ld4(src + n8);
    if (n & 2u) r->t2 = mz_ld2(src + n8 + (n & 4u));
    if (n & 1u) r->t1 = src[n - 1u];
}
MZ_DEV void mz_copy32_store(const mz_piece32 *r, uint8_t *dst, uint32_t n) {
    const uint32_t n8 = n & ~7u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++)
        if (8u * k + 8u <= n) mz_st8(dst + 8u * k, r->v[k]);
    if (n & 4u) mz_st4(dst + n8, r->t4);
    if (n & 2u) mz_st2(dst + n8 + (n & 4u), (uint16_t)r->t2);
    if (n & 1u) dst[n - 1u] = (uint8_t)r->t1;
}
MZ_DEV void mz_copy32(uint8_t *dst, const uint8_t *src, uint32_t n) {
    mz_piece32 r;
    mz_copy32_load(&r, src, n);
    mz_copy32_store(&r, dst, n);
}
/* the same inside the staging area, where the source may end less than n (but at least 8) bytes in front of the
 * destination: 8-byte steps in order, each one reading only what the steps before it (or earlier pieces) wrote */
MZ_DEV void mz_copy32_seq(uint8_t *dst, const uint8_t *src, uint32_t n) {
    const uint32_t n8 = n & ~7u;
#pragma unroll
    for (uint32_t k = 0; k < 4u; k++)
        if (8u * k + 8u <= n) mz_st8(dst + 8u * k, mz_ld8(src + 8u * k));
    if (n & 4u) mz_st4(dst + n8, mz_ld4(src + n8));
    if (n & 2u) mz_st2(dst + n8 + (n & 4u), mz_ld2(src + n8 + (n & 4u)));
    if (n & 1u) dst[n - 1u] = src[n - 1u];
}

/* One step of a chase-window walk (inflate_chase.inc) as its RECORD.  A DEFLATE token walk started at an arbitrary bit falls
 * in step with the true token sequence after 8.5 tokens on average (text at zlib level 6: 88 % of the walks within 256 bits,
 * 98 % within 512, profiles/r1/side_measurements.log), which is what lets 64 lanes walk 64 spans of one stream at once.
 * The table walk of one step: the token that starts at bit `rel`, or, when that is a literal of fewer than `room` bits, the
 * literal and the token behind it (the slowest lanes of a pass are the literal-dense ones: 26.3 -> 16.1 steps for the slowest
 * lane of a window on text, tests/study/span_multi.c); what comes out is what the emit needs and nothing else, in five bytes:
 *   *rec  [7:0] the leading literal and [31] "there is one"; [15:8] the token's literal byte, or match length - 3;
 *         [30:16] match distance - 1
 *   K     (returned) [5:0] bits of the whole step, 1 .. 63 (leading literal <= 15, token <= 48); 0 = no valid step starts
 *         here (an unused or invalid code: the leading literal stays undecoded too); [7:6] the token: 0 match, 1 literal,
 *         2 end of block */
#define MZ_K_LIT 0x40u
#define MZ_K_EOB 0x80u
MZ_DEV uint32_t mz_chase_step(const mz_inflate_lds *L, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t rel, uint32_t room,
                              uint32_t *rec) {
    uint32_t w0 = mz_funnel(d1, d0, rel), w1 = mz_funnel(d2, d1, rel);
    uint32_t e = L->lit_fast[w0 & ((1u << MZ_LROOT) - 1u)];
    if (e & MZ_E_SUB) e = L->lit_sub[((e >> 8) & 0x1FFu) + mz_bfe(w0, MZ_LROOT, e & 7u)];
    uint32_t r = 0, pb = 0;
    if ((((e & (MZ_E_LEN | 0x80u)) == 0x80u) & ((e & 63u) < room)) != 0) { /* a literal that does not end the lane's walk takes the token behind it along (one test, one exec bracket: `&&` made two) */
        pb = e & 63u;
        r = 0x80000000u | mz_bfe(e, 16, 8);
        w0 = mz_funnel(w1, w0, pb);
        w1 >>= pb;
        e = L->lit_fast[w0 & ((1u << MZ_LROOT) - 1u)];
        if (e & MZ_E_SUB) e = L->lit_sub[((e >> 8) & 0x1FFu) + mz_bfe(w0, MZ_LROOT, e & 7u)];
    }
    if (!(e & MZ_E_LEN)) { /* a literal (0x80 | byte << 16 | bits), the end of the block (0x40 | bits), or nothing valid (no bits) */
        const uint32_t nb = e & 63u;
        *rec = r | ((e >> 8) & 0xFF00u);
        return nb ? ((nb + pb) | ((e & 0x80u) ? MZ_K_LIT : MZ_K_EOB)) : 0u;
    }
    const uint32_t nb = e & 63u, ex = mz_bfe(e, 16, 4);
    const uint32_t lenl = mz_bfe(e, 7, 9) + mz_bfe(mz_funnel(w1, w0, nb), 0, ex);
    const uint32_t nb2 = nb + ex;
    const uint32_t dl = mz_funnel(w1, w0, nb2);
    uint32_t dd = L->dist_fast[dl & ((1u << MZ_DROOT) - 1u)];
    MZ_STAT(19, 1);
    if (dd == 0u) {
        MZ_STAT(20, 1);
        dd = mz_long_code(dl, MZ_DROOT, L->dist_lim, L->dist_delta, L->dist_ent, 32u);
    }
    if ((int32_t)dd <= 0) { /* unused / 30 / 31 distance code */
        *rec = 0u;
        return 0u;
    }
    const uint32_t dn = dd & 15u, dex = mz_bfe(dd, 4, 4);
    const uint32_t dist = mz_bfe(dd, 8, 15) + mz_bfe(dl, dn, dex);
    *rec = r | ((lenl - 3u) << 8) | ((dist - 1u) << 16);
    return nb2 + dn + dex + pb;
}
/* dwords d .. d + 3 of the aligned stream (see mz_load_stream_dword), bytes outside the input read as zero */
typedef struct { uint32_t v[4]; } mz_dw4;
MZ_DEV mz_dw4 mz_load_stream_dw4(const uint8_t *in_al, uint32_t in_mis, uint32_t in_len, uint32_t d) {
    mz_dw4 r;
    const uint64_t lo = (uint64_t)d * 4u, end = (uint64_t)in_mis + in_len;
    if (lo >= in_mis && lo + 16u <= end) {
        __builtin_memcpy(&r, in_al + lo, 16); /* one global_load_dwordx4 (4-byte aligned) */
    } else {
#pragma unroll
        for (uint32_t k = 0; k < 4u; k++) r.v[k] = mz_load_stream_dword(in_al, in_mis, in_len, d + k);
    }
    return r;
}


#include "inflate_walk.inc"
#include "inflate_emit.inc"

/* ONE block of a large entry, parsed by a wave of its own (mzhip_inflate_parallel_host: every block of a window at once).
 * The wave starts at the block's header (rs: hdr_bit = bit = the header, out_pos = where the block's bytes go), decodes that
 * one block with the chase window, and does not copy anything: mode 1 counts the bytes the block produces (res->out_len),
 * mode 2 writes literals to out[] and a source index per byte to ptr[] (mz_chase_emit_ptr).  st->bit = the bit behind the
 * block's end-of-block code, res->crc = 1 when the block was the stream's last.  Anything the chase window leaves to the
 * step loop -- an error on the path, the last bits of the stream -- ends the call with MZHIP_PAR_BAIL: that block (and
 * what follows it) is decoded in stream order by the ordinary kernel. */
typedef struct mz_inflate_par {
    uint32_t mode; /* 1 count, 2 emit */
    uint32_t *ptr; /* mode 2: source index of every byte of out[] */
} mz_inflate_par;
#define MZHIP_PAR_BAIL (-301)

/* Stage 1 of the block search of a large entry (mzhip_inflate_parallel_host): could a dynamic-Huffman block header (or a
 * stored block's) start at bit p of in[]?  BTYPE = 2, HLIT <= 29, HDIST <= 29, and the code-length code's lengths form a COMPLETE prefix code
 * (zlib refuses an incomplete one: "invalid code lengths set").  About one bit offset in a few thousand of random data
 * passes; what passes is parsed for real (the whole header, then the block) by a wave of its own, and only blocks that are
 * reached from the known start of the window through a chain of "ends exactly where the next one starts" are believed.
 * Per lane: one bit offset, ~40 instructions. */
MZ_DEV uint32_t mz_block_header_plausible(const uint8_t *in, uint32_t in_len, uint32_t p) {
    if ((uint64_t)(p >> 3) + 24u > in_len) return 0u; /* (the last bytes of the input are the serial decoder's anyway) */
    uint64_t w = mz_bits_at(in, in_len, p);
    const uint32_t h = (uint32_t)w;
    if (((h >> 1) & 3u) == 0u) {
        /* a stored block: LEN and its complement behind the padding (appnote.txt:2045-2049); one offset in 2^18 of random
         * data passes.  Incompressible stretches of an entry are made of these, and the chain must get across them */
        const uint32_t b = (p + 3u + 7u) >> 3;
        const uint32_t len = (uint32_t)in[b] | ((uint32_t)in[b + 1] << 8), nlen = (uint32_t)in[b + 2] | ((uint32_t)in[b + 3] << 8);
        return (len ^ nlen) == 0xFFFFu ? 1u : 0u;
    }
    if (((h >> 1) & 3u) != 2u) return 0u;
    if (((h >> 3) & 31u) > 29u || ((h >> 8) & 31u) > 29u) return 0u;
    const uint32_t ncode = ((h >> 13) & 15u) + 4u;
    uint32_t kraft = 0;
    w >>= 17; /* 47 bits left: 15 lengths of 3 bits */
    for (uint32_t i = 0; i < ncode; i++) {
        if (i == 15u) w = mz_bits_at(in, in_len, p + 17u + 45u);
        const uint32_t l = (uint32_t)w & 7u;
        w >>= 3;
        kraft += l ? (128u >> l) : 0u;
    }
    return kraft == 128u ? 1u : 0u;
}

#include "inflate_header.inc"

/* Decode one raw-DEFLATE entry.  All arguments are wave-uniform. */
MZ_DEV void mz_inflate_entry(const uint8_t *in, uint32_t in_total, uint8_t *out, uint32_t out_cap,
                             mz_inflate_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs,
                             uint32_t use_span, uint8_t *rec /* MZ_REC_BYTES of HBM scratch of this wave (chase window) */,
                             const mz_inflate_state *rs /* take the stream up here (or null) */,
                             mz_inflate_state *st /* where it can be taken up again when the output is full / the input ends (or null) */,
                             mz_inflate_result *res, const mz_inflate_par *par /* one block, no copies (or null: the ordinary decode) */) {
    MZ_LANE_DECL
    uint32_t par_blocks = 0; /* par: blocks completed */
    /* The bit cursor is 32 bits wide, so the decoder looks at the stream through a VIEW of at most MZ_VIEW_MAX bytes
     * ([in, in + in_len), bitpos relative to `in`) and moves the view forward (MZ_REBASE) whenever the cursor is more
     * than MZ_REBASE_BITS into it: at the top of every block and of every window / step, i.e. long before anything
     * could reach the far end of a view that is not the end of the stream (the longest stretch between two checks is
     * one stored block, 64 KiB).  Streams of any length the 32-bit in_len[] of the batch can describe are served
     * (mz_strm_zlib.c:116-193 streams any size). */
    uint32_t in_rem = in_total;  /* bytes from `in` to the real end of the stream */
    uint32_t in_adv = 0;         /* bytes the view has moved */
    uint32_t in_len = in_rem < MZ_VIEW_MAX ? in_rem : MZ_VIEW_MAX;
    uint32_t total_bits = in_len * 8u;
    const uint32_t in_mis = (uint32_t)((uintptr_t)in & 3u); /* the view moves by multiples of 4: the misalignment stays */
    const uint8_t *in_al = in - in_mis;
    uint32_t bitpos = 0;
    /* resumable decode: hdr_bit = the header of the block being decoded, qbit = where the step loop's token queue starts
     * (tokens behind it are decoded but not written yet), in_header = the cursor is inside a block header */
    uint32_t hdr_bit = 0, qbit = 0, in_header = 0, unwritten = 0, resume_at = 0xFFFFFFFFu;
    const uint32_t resumable = st ? 1u : 0u;
    uint32_t stop_at_header = 0;
    if (rs && (MZ_UNIFORM(rs->flags) & 1u)) {
        bitpos = MZ_UNIFORM(rs->hdr_bit);
        resume_at = MZ_UNIFORM(rs->bit);
    }
    if (rs && st) stop_at_header = (MZ_UNIFORM(rs->flags) >> 1) & 1u;
    /* the two groups of four stream dwords that do not lie entirely inside the input, masked (the chase window's refills,
     * inflate_walk.inc): lanes 0 .. 3 the group of dword 0, lanes 4 .. 7 the group with the last dword.  Once per view, so
     * that no window waits for these loads */
    PV(uint32_t, edge_ev);
#define MZ_EDGE_GROUPS()                                                                               \
    MZ_LANES {                                                                                         \
        const uint32_t _gl4 = ((in_mis + in_len) >> 2) & ~3u;                                          \
        P(edge_ev) = (lane < 8) ? mz_load_stream_dword(in_al, in_mis, in_len, (lane < 4) ? (uint32_t)lane : _gl4 + ((uint32_t)lane & 3u)) : 0u; \
    }
    MZ_EDGE_GROUPS();
#define MZ_REBASE(also)                                                \
    if (bitpos >= MZ_REBASE_BITS) {                                    \
        const uint32_t _adv = (bitpos >> 3) & ~3u;                     \
        if (_adv <= in_rem) {                                          \
            in += _adv;                                                \
            in_al += _adv;                                             \
            in_adv += _adv;                                            \
            in_rem -= _adv;                                            \
            in_len = in_rem < MZ_VIEW_MAX ? in_rem : MZ_VIEW_MAX;      \
            total_bits = in_len * 8u;                                  \
            bitpos -= _adv * 8u;                                       \
            MZ_EDGE_GROUPS();                                          \
            also;                                                      \
        }                                                              \
    }
    uint32_t out_pos = rs ? MZ_UNIFORM(rs->out_pos) : 0u; /* bytes [0, out_pos) of out[] are history (back-references may reach them) */
    uint32_t chase_smax = MZ_CHASE_SMAX; /* bits per span at most (inflate_chase.inc halves it when a window runs into a record cap) */
    int32_t status = MZHIP_OK;
    uint32_t last = 0;
    PV(uint32_t, crc_acc);
    PV(uint32_t, crc_tmp);
    const uint32_t crc_base = out_pos; /* the CRC is of the bytes this call produces: the folding counts from here */
    uint32_t crc_done = 0;
    MZ_LANES { P(crc_acc) = (lane == 0) ? 0xFFFFFFFFu : 0u; }
    MZ_PROF_DECL

    while (!last) {
        if (par && par_blocks) goto finish; /* one block per call */
        MZ_REBASE((void)0)
        hdr_bit = bitpos;
        in_header = 1;
        if (stop_at_header && par_blocks) { /* (the state written below is this header, with nothing of the block entered) */
            status = MZHIP_OUT_FULL;
            goto finish;
        }
        /* the block header is read through a 256-byte window of the stream held one dword per lane (one coalesced
         * load instead of a global round trip per 64 bits of header); positions beyond it fall back to memory */
        const uint32_t hw0 = (bitpos + 8u * in_mis) >> 5;
        PV(uint32_t, hwin);
        MZ_LANES { P(hwin) = mz_load_stream_dword(in_al, in_mis, in_len, hw0 + (uint32_t)lane); }
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
            if (resumable && n < len) { /* taken up again at the block's header when the rest of the block has arrived */
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            if (n > out_cap - out_pos) {
                status = MZHIP_OUT_FULL;
                goto finish;
            }
            if (par && n < len) { /* (a stored block that the input does not hold whole: stream order decides) */
                status = MZHIP_PAR_BAIL;
                goto finish;
            }
            if (!par || par->mode == 2u) {
                MZ_LANES {
                    for (uint32_t i = (uint32_t)lane; i < n; i += 64) {
                        out[out_pos + i] = in[byte + i];
                        if (par) par->ptr[out_pos + i] = out_pos + i;
                    }
                }
                MZ_WAVE_SYNC();
            }
            out_pos += n;
            bitpos += n * 8u;
            if (n < len) {
                status = MZHIP_BUF_ERROR;
                goto finish;
            }
            if (!par) MZ_CRC_FOLD_SUPER_BT(crc_acc, crc_done, out + crc_base, out_pos - crc_base, crc_tab, tabs->kx4);
            in_header = 0;
            hdr_bit = bitpos;
            par_blocks++;
            continue;
        }
        if (btype == 3) {
            status = MZHIP_DATA_ERROR; /* invalid block type */
            goto finish;
        }

        uint32_t sub_used; /* second-level entries this block's literal/length code occupies */
        {
            /* the block's code: code lengths -> decode tables in LDS (inflate_header.inc, a function of its own) */
            const mz_code_result cr = mz_block_code(MZ_LDS_HANDLE(L), MZ_GLB_HANDLE(in), in_len, in_mis, total_bits, bitpos, hw0, btype, hwin MZ_PROF_ARGS);
            sub_used = MZ_UNIFORM(cr.sub_used);
            bitpos = MZ_UNIFORM(cr.bitpos);
            if (MZ_UNIFORM(cr.status) != (uint32_t)MZHIP_OK) {
                status = (int32_t)MZ_UNIFORM(cr.status);
                goto finish;
            }
        }
        in_header = 0;
        if (resume_at != 0xFFFFFFFFu) { /* taken up inside this block: the tables are back, on to the next token */
            if (resume_at > bitpos) bitpos = resume_at;
            resume_at = 0xFFFFFFFFu;
        }
        /* ---- compressed block body: speculative 64-offset decode ----
         * Compressed bytes are staged through a 512-byte LDS ring (two 256-byte blocks, the next
         * block prefetched into a VGPR one block ahead), so the per-step window fetch is three
         * ds_read_b32 and no global-memory latency sits on the critical path. */
        {
            uint32_t *ring = MZ_L_RING(L);
            const uint32_t pbase = 8u * in_mis; /* bit offset of `in` inside its aligned dword */
            uint32_t ring_hi = 0;               /* blocks < ring_hi are in the ring; block ring_hi is in wpre */
            uint32_t ring_valid = 0;            /* the ring is (re)loaded on entry to the step loop: the span path moves the cursor */
            PV(uint32_t, wpre);

            PV(uint32_t, tq); /* token queue: lane i = i-th pending token of this flush interval */
            uint32_t qn = 0;
            MZ_LANES { P(tq) = 0u; }

            uint32_t span_skip = 0;     /* the step loop takes the next step (it owns the exact verdicts) */
            uint32_t span_on = (use_span && rec) ? 1u : 0u; /* cleared for the rest of the block when a window cannot be committed here */
            for (;;) {
                MZ_REBASE(ring_valid = 0) /* the ring is indexed by the position inside the view */
                {
                    const uint32_t remain = total_bits - bitpos;
                    /* worth a window: two spans of 128 bits and some */
                    if (span_on && qn == 0u && !span_skip && remain >= 64u + 2u * 128u) {
                        ring_valid = 0; /* the pool covers the step loop's ring */
                        MZ_PROF_MARK(3); /* step loop (if any) */
#include "inflate_chase.inc"
                    }
                    if (par) { /* the step loop is next: not a block for a wave of its own */
                        status = MZHIP_PAR_BAIL;
                        goto finish;
                    }
                    span_skip = 0;
                }
                if (!ring_valid) {
                    const uint32_t blk = (bitpos + pbase) >> 11; /* 2048 bits per block */
                    MZ_LANES {
                        const uint32_t da = mz_load_stream_dword(in_al, in_mis, in_len, blk * 64u + (uint32_t)lane);
                        const uint32_t db = mz_load_stream_dword(in_al, in_mis, in_len, (blk + 1u) * 64u + (uint32_t)lane);
                        ring[((blk & 1u) << 6) + (uint32_t)lane] = da;
                        ring[(((blk + 1u) & 1u) << 6) + (uint32_t)lane] = db;
                        if (lane < 2) ring[128 + lane] = (blk & 1u) ? db : da; /* mirror of ring[0], ring[1] */
                        P(wpre) = mz_load_stream_dword(in_al, in_mis, in_len, (blk + 2u) * 64u + (uint32_t)lane);
                    }
                    ring_hi = blk + 2u;
                    ring_valid = 1;
                    MZ_WAVE_SYNC();
                }
                if (qn == 0u) qbit = bitpos; /* the token queue starts here: a flush that does not fit is taken up again from here */
                const uint32_t pbit = bitpos + pbase;
                if ((pbit >> 11) + 1u >= ring_hi) {
                    /* the cursor entered the newest block: retire the oldest, start the next fetch */
                    MZ_LANES {
                        ring[((ring_hi & 1u) << 6) + (uint32_t)lane] = P(wpre);
                        if (lane < 2 && !(ring_hi & 1u)) ring[128 + lane] = P(wpre); /* mirror of ring[0], ring[1] */
                        P(wpre) = mz_load_stream_dword(in_al, in_mis, in_len, (ring_hi + 1u) * 64u + (uint32_t)lane);
                    }
                    ring_hi++;
                    MZ_WAVE_SYNC();
                }

                /* phase 1: every lane decodes the token that would start at bit cursor + lane.
                 * w0/w1 = stream bits [0,32) / [32,64) from that offset (funnel shifts of ring dwords). */
                PV(uint32_t, w0);
                PV(uint32_t, w1);
                PV(uint32_t, le); /* literal/length table entry */
                MZ_LANES {
                    const uint32_t pl = pbit + (uint32_t)lane;
                    const uint32_t j = pl >> 5;
                    const uint32_t jr = j & 127u;
                    const uint32_t d0 = ring[jr], d1 = ring[jr + 1u], d2 = ring[jr + 2u];
                    P(w0) = mz_funnel(d1, d0, pl);
                    P(w1) = mz_funnel(d2, d1, pl);
                    P(le) = L->lit_fast[P(w0) & ((1u << MZ_LROOT) - 1)];
                }
                MZ_LANES { /* codes longer than the root: one more lookup in the group's sub-table */
                    const uint32_t e = P(le);
                    const uint32_t sx = ((e >> 8) & 0x1FFu) + mz_bfe(P(w0), MZ_LROOT, e & 7u);
                    const uint32_t e2 = L->lit_sub[(e & MZ_E_SUB) ? sx : 0u];
                    P(le) = (e & MZ_E_SUB) ? e2 : e;
                }
                uint64_t slow;
                PV(uint32_t, lenl);
                PV(uint32_t, nb2l);
                PV(uint32_t, dlo);
                PV(uint32_t, de); /* distance table entry */
                MZ_LANES {
                    const uint32_t e = P(le);
                    const uint32_t nb = e & 63u, ex = mz_bfe(e, 16, 4);
                    P(lenl) = mz_bfe(e, 7, 9) + mz_bfe(mz_funnel(P(w1), P(w0), nb), 0, ex);
                    const uint32_t nb2 = nb + ex; /* <= 20 */
                    const uint32_t dl = mz_funnel(P(w1), P(w0), nb2);
                    P(nb2l) = nb2;
                    P(dlo) = dl;
                    P(de) = L->dist_fast[dl & ((1u << MZ_DROOT) - 1)];
                }
                MZ_BALLOT(slow, (P(le) & MZ_E_LEN) && P(de) == 0u);
                if (slow) {
                    MZ_LANES {
                        const uint32_t e = mz_long_code(P(dlo), MZ_DROOT, L->dist_lim, L->dist_delta, L->dist_ent, 32u);
                        if (P(de) == 0u) P(de) = e;
                    }
                }
                PV(uint32_t, tk);
                PV(uint32_t, g1); /* 4 * successor offset; >= 256: terminal (|0x400 end-of-block, |0x800 invalid) */
                /* a candidate that hits an unused literal/length code (entry 0), 286 / 287 (finished as an invalid
                 * token by the table build) or an unused / 30 / 31 distance code ((int)d <= 0) comes out with 0 bits,
                 * which is what stops the chain; only within 14 bytes of the end of the input does the verdict
                 * (data error or input exhausted) need the exact number of bits each invalid candidate looked at */
                const uint32_t avail = total_bits - bitpos;
                MZ_LANES {
                    const uint32_t e = P(le), d = P(de), nb2 = P(nb2l);
                    const uint32_t dn = d & 15u, dex = mz_bfe(d, 4, 4);
                    const uint32_t dist = mz_bfe(d, 8, 15) + mz_bfe(P(dlo), dn, dex);
                    const uint32_t t_match = (nb2 + dn + dex) | (P(lenl) << 7) | (dist << 16);
                    P(tk) = (e & MZ_E_LEN) ? (((int32_t)d > 0) ? t_match : 0u) : e;
                }
                /* an invalid candidate carries the bits inflate() has dropped when it refuses it ([31:16], no bits in [5:0]): the
                 * code's own for 286 / 287 and 30 / 31 (the tables say so), ONE for an unused code -- a set with unused codes
                 * passes only when its longest code is one bit (inflate_table()), and the entry zlib leaves for the other
                 * pattern is one bit wide.  That is what decides "data error" against "input exhausted" at the very end of the
                 * input, and where TOTAL_IN stands after a data error (round 5: it stood at the token's first bit, 1 - 3 bytes
                 * short in fixed-Huffman garbage: tests/fuzz_gpu_windows.py) */
                MZ_LANES {
                    const uint32_t e = P(le), d = P(de), nb2 = P(nb2l);
                    const uint32_t t_badd = ((d == 0u) ? (nb2 + 1u) : (nb2 + (d & 15u))) << 16; /* invalid distance code */
                    uint32_t t = P(tk);
                    t = ((e & MZ_E_LEN) && (int32_t)d <= 0) ? t_badd : t;
                    t = (e == 0u) ? (1u << 16) : t; /* an unused literal/length code */
                    P(tk) = t;
                }
                MZ_LANES {
                    const uint32_t t = P(tk);
                    const uint32_t nb = t & 63u;
                    const uint32_t nx4 = 4u * ((uint32_t)lane + nb) + ((t & 64u) << 4);
                    P(g1) = (nb == 0u) ? (0x800u | (4u * (uint32_t)lane)) : nx4;
                }

                /* phase 2: which candidates are real tokens?  Token i of this step starts at f^i(0), where
                 * f(l) = l + bits(l).  f is squared three times with cross-lane gathers (f^2, f^4, f^8) and
                 * lane i (i < 16) composes f^i(0) from the binary digits of i; squaring and composing are
                 * interleaved, so it is five dependent ds_bpermute rounds, no scalar work, and the step's tokens
                 * come out COMPACTED (lane i holds token i).  Offsets are kept multiplied by 4 (the gather's byte
                 * address).  At most 15 tokens retire per step; lane 15 only supplies the continuation offset. */
                uint32_t pos, eob = 0, ntok, err_bits = 0;
                int32_t chain_err = MZHIP_OK;
                PV(uint32_t, cpos);
                if (avail >= 64u + 48u) {
                    PV(uint32_t, g2);
                    PV(uint32_t, g4);
                    PV(uint32_t, g8);
                    PV(uint32_t, gt);
                    PV(uint32_t, ct);
                    MZ_LANES { P(cpos) = 0u; }
                    MZ_GATHER4(gt, g1, P(g1));
                    MZ_GATHER4(ct, g1, P(cpos));
                    MZ_LANES {
                        P(g2) = (P(g1) < 256u) ? P(gt) : P(g1);
                        P(cpos) = ((uint32_t)lane & 1u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER4(gt, g2, P(g2));
                    MZ_GATHER4(ct, g2, P(cpos));
                    MZ_LANES {
                        P(g4) = (P(g2) < 256u) ? P(gt) : P(g2);
                        P(cpos) = (((uint32_t)lane & 2u) && P(cpos) < 256u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER4(gt, g4, P(g4));
                    MZ_GATHER4(ct, g4, P(cpos));
                    MZ_LANES {
                        P(g8) = (P(g4) < 256u) ? P(gt) : P(g4);
                        P(cpos) = (((uint32_t)lane & 4u) && P(cpos) < 256u) ? P(ct) : P(cpos);
                    }
                    MZ_GATHER4(ct, g8, P(cpos));
                    MZ_LANES { P(cpos) = (((uint32_t)lane & 8u) && P(cpos) < 256u) ? P(ct) : P(cpos); }
                    uint64_t live;
                    MZ_BALLOT(live, lane < 15 && P(cpos) < 256u);
                    ntok = mz_popc64(live); /* tokens are lanes 0..ntok-1 (the chain never resumes once terminal) */
                    const uint32_t term = MZ_READLANE(cpos, ntok);
                    pos = (term >> 2) & 0xFFu;
                    eob = (term >> 10) & 1u;
                    if (term & 0x800u) {
                        chain_err = MZHIP_DATA_ERROR;
                        err_bits = MZ_READLANE(tk, pos) >> 16;
                    }
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
                            err_bits = t >> 16;
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
                        if ((sel >> lane) & 1u) MZ_L_MSLOT(L)[mz_popc64(sel & ((1ull << lane) - 1ull))] = (uint16_t)(4 * lane);
                    }
                    MZ_WAVE_SYNC();
                    MZ_LANES { P(cpos) = ((uint32_t)lane < ntok) ? (uint32_t)MZ_L_MSLOT(L)[lane & 15] : 0x1000u; }
                    MZ_WAVE_SYNC();
                }
                bitpos += pos;

                /* phase 3: the step's compacted tokens join a queue of up to 64 tokens held one per lane
                 * (lane i = i-th pending token); output work below runs once per ~50 tokens instead of once per
                 * step, with every lane busy. */
                {
                    PV(uint32_t, tkc);
                    PV(uint32_t, tsh);
                    MZ_GATHER4(tkc, tk, P(cpos));
                    MZ_GATHER4(tsh, tkc, (4u * ((uint32_t)lane - qn)) & 255u);
                    MZ_LANES {
                        if ((uint32_t)lane >= qn && (uint32_t)lane < qn + ntok) P(tq) = P(tsh);
                    }
                    qn += ntok;
                }
                if (qn + 15u <= 64u && !eob && chain_err == MZHIP_OK) continue;

#include "inflate_flush.inc"
                if (chain_err != MZHIP_OK) {
                    status = chain_err;
                    if (chain_err == MZHIP_DATA_ERROR) bitpos += err_bits; /* (the bits of the code inflate() refuses are consumed: TOTAL_IN) */
                    goto finish;
                }
                if (eob) break;
            }
        }
        par_blocks++;
    }

finish:
    MZ_PROF_MARK(11); /* step loop behind the last window, stored blocks */
    if (st) {
        /* a position from which (with this output, and more room or more input) the decode goes on exactly: the header of
         * the block when the cursor is inside it, else the start of the step loop's unwritten tokens, else the cursor */
        const uint32_t rb = in_header ? hdr_bit : (unwritten ? qbit : bitpos);
        MZ_LANES { /* uniform stores */
            st->hdr_bit = hdr_bit;
            st->bit = rb;
            st->out_pos = out_pos;
            st->flags = (in_adv == 0u) ? 1u : 0u; /* (positions are only meaningful while the view has not moved: the caller keeps calls < 128 MiB) */
        }
    }
    res->status = status;
    res->out_len = out_pos;
    res->in_used = in_adv + ((bitpos + 7u) >> 3);
    if (res->in_used > in_total) res->in_used = in_total;
    if (par) {
        if (status == MZHIP_OK && par_blocks == 0u) res->status = MZHIP_PAR_BAIL;
        res->crc = last; /* (no checksum here: the window's is computed once its bytes exist) */
        (void)crc_tmp;
    } else {
        uint32_t crc;
#if MZ_ABLATE & 4
        crc = 0;
        (void)crc_tmp;
#else
        MZ_CRC_FOLD_SUPER_BT(crc_acc, crc_done, out + crc_base, out_pos - crc_base, crc_tab, tabs->kx4);
        MZ_CRC_FINISH_SUPER_BT(crc, crc_acc, crc_tmp, crc_done, out + crc_base, out_pos - crc_base, crc_tab, tabs);
#endif
        res->crc = crc;
    }
    MZ_PROF_MARK(12); /* CRC tail */
    MZ_PROF_FLUSH
}

/* One candidate block of a window by a wave of its own: res4 = {status, the bit behind the block, the out position behind
 * its last byte (mode 1: the bytes it produces, pos = 0), BFINAL}.  The wave's view of the stream starts at the dword the
 * header lies in, so that bit positions stay small whatever the window's size. */
MZ_DEV void mz_inflate_one_block(const uint8_t *in, uint32_t in_len, uint32_t bit, uint32_t pos, uint32_t mode, uint8_t *out,
                                 uint32_t *ptr, mz_inflate_lds *L, const uint32_t *crc_tab, const mzhip_crc_tables *tabs,
                                 uint8_t *rec, uint32_t *res4) {
    const uint32_t base = (bit >> 5) << 2;
    mz_inflate_state a, b;
    a.hdr_bit = a.bit = bit - base * 8u;
    a.out_pos = pos;
    a.flags = 1u;
    b.hdr_bit = b.bit = b.out_pos = b.flags = 0u;
    mz_inflate_par par;
    par.mode = mode;
    par.ptr = ptr;
    mz_inflate_result r;
    mz_inflate_entry(in + base, in_len - base, out, 0xFFFFFFFFu, L, crc_tab, tabs, 1u, rec, &a, &b, &r, &par);
    int32_t status = r.status;
    if (status == MZHIP_OK && !(b.flags & 1u)) status = MZHIP_PAR_BAIL; /* (a block of more than a view: stream order) */
    MZ_LANES { /* uniform stores */
        res4[0] = (uint32_t)status;
        res4[1] = b.bit + base * 8u;
        res4[2] = r.out_len;
        res4[3] = r.crc;
    }
}

#endif
