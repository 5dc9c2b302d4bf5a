/* crc32_core.h -- wave-parallel CRC-32 (replaces mz_crypt_crc32_update,
 * reference mz_crypt.c:35-92: reflected 0xEDB88320, register inverted on
 * entry and exit).
 *
 * CRC is GF(2)-linear, so a buffer is folded 1 KiB (one "tile") at a time by a
 * whole wavefront: lane l owns the 16 bytes at tile*1024 + 16*l (a coalesced
 * 16 B/lane read) and keeps its own 32-bit accumulator.  Between tiles the
 * accumulator is advanced over the 1008 bytes owned by the other lanes by one
 * multiplication with the constant x^(8*1008) mod P; at the end lane l is
 * advanced by x^(128*(63-l)), the 64 accumulators are XOR-reduced, and the
 * <1 KiB tail is folded the same way with per-lane byte counts.  The initial
 * 0xFFFFFFFF rides in lane 0's accumulator, so no length-dependent fix-up is
 * needed.  Polynomials are held bit-reflected (bit 31 = x^0) like the
 * reference's table (mz_crypt.c:52).
 */
#ifndef MZHIP_CRC32_CORE_H
#define MZHIP_CRC32_CORE_H

#include <stddef.h>
#include "wave.h"

#define MZ_CRC_POLY 0xEDB88320u
#define MZ_CRC_TILE 1024u
#define MZ_CRC_SUPER 4096u /* stand-alone kernel: 64 contiguous bytes per lane per super-tile */

/* constants generated once on the host (mzhip_crc_tables_init) and handed to
 * the kernels; byte_tab is staged into LDS per workgroup. */
typedef struct mzhip_crc_tables {
    uint32_t byte_tab[256]; /* mz_crypt.c:52-80 equivalent, generated */
    uint32_t kx[32];        /* x^(8*1008) * x^j  mod P, j = 0..31       */
    uint32_t x16[64];       /* x^(128*j) mod P                          */
    uint32_t x1[16];        /* x^(8*j) mod P                            */
    uint32_t kx4[32];       /* x^(8*4032) * x^j mod P, j = 0..31 (super-tiles of the stand-alone kernel) */
    uint32_t x64[64];       /* x^(512*j) mod P                          */
    uint32_t mul4[4][256];  /* (v at byte b of the register) * x^(8*4032) mod P: the super-tile advance as four lookups */
    uint32_t slice[4][256]; /* slicing-by-4 tables: slice[0] = byte_tab, slice[k][i] = slice[k-1][i] advanced by one zero byte */
} mzhip_crc_tables;

/* host-side generation (plain arithmetic on 32-bit polynomials) */
static inline uint32_t mzhip_gf2_mul_host(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u) p ^= b;
        a <<= 1;
        b = (b & 1) ? ((b >> 1) ^ MZ_CRC_POLY) : (b >> 1);
    }
    return p;
}
static inline uint32_t mzhip_xpow8_host(uint64_t nbytes) {
    uint32_t r = 0x80000000u, sq = 0x00800000u; /* x^0, x^8 */
    while (nbytes) {
        if (nbytes & 1) r = mzhip_gf2_mul_host(r, sq);
        sq = mzhip_gf2_mul_host(sq, sq);
        nbytes >>= 1;
    }
    return r;
}
static inline void mzhip_crc_tables_init(mzhip_crc_tables *t) {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? (MZ_CRC_POLY ^ (c >> 1)) : (c >> 1);
        t->byte_tab[n] = c;
    }
    uint32_t k = mzhip_xpow8_host(MZ_CRC_TILE - 16);
    for (int j = 0; j < 32; j++) {
        t->kx[j] = k;
        k = (k & 1) ? ((k >> 1) ^ MZ_CRC_POLY) : (k >> 1); /* times x */
    }
    for (int j = 0; j < 64; j++) t->x16[j] = mzhip_xpow8_host(16u * (uint32_t)j);
    for (int j = 0; j < 16; j++) t->x1[j] = mzhip_xpow8_host((uint32_t)j);
    k = mzhip_xpow8_host(MZ_CRC_SUPER - 64);
    for (int j = 0; j < 32; j++) {
        t->kx4[j] = k;
        k = (k & 1) ? ((k >> 1) ^ MZ_CRC_POLY) : (k >> 1);
    }
    for (int j = 0; j < 64; j++) t->x64[j] = mzhip_xpow8_host(64u * (uint32_t)j);
    for (int b = 0; b < 4; b++)
        for (uint32_t n = 0; n < 256; n++) t->mul4[b][n] = mzhip_gf2_mul_host(n << (8 * b), t->kx4[0]);
    for (uint32_t n = 0; n < 256; n++) t->slice[0][n] = t->byte_tab[n];
    for (int k2 = 1; k2 < 4; k2++)
        for (uint32_t n = 0; n < 256; n++) t->slice[k2][n] = (t->slice[k2 - 1][n] >> 8) ^ t->byte_tab[t->slice[k2 - 1][n] & 255];
}
/* crc(A||B) from crc(A), crc(B), |B| -- 32-bit arithmetic on checksums only */
static inline uint32_t mzhip_crc32_combine_host(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return mzhip_gf2_mul_host(mzhip_xpow8_host(len_b), crc_a) ^ crc_b;
}

/* a * b mod P, both per-lane */
MZ_DEV uint32_t mz_gf2_mul(uint32_t a, uint32_t b) {
    uint32_t p = 0;
#pragma unroll 4
    for (int i = 0; i < 32; i++) {
        p ^= b & (0u - (a >> 31));
        a <<= 1;
        b = (b >> 1) ^ (MZ_CRC_POLY & (0u - (b & 1)));
    }
    return p;
}

/* a * K mod P for the compile-time-fixed K whose shifted copies are kx[] */
MZ_DEV uint32_t mz_gf2_mul_kx(uint32_t a, const uint32_t *kx) {
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < 32; j++)
        p ^= kx[j] & (0u - ((a >> (31 - j)) & 1u));
    return p;
}

MZ_DEV uint32_t mz_load_u32(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

/* fold one little-endian dword into a raw (non-inverted) register */
MZ_DEV uint32_t mz_crc_dword(uint32_t r, uint32_t d, const uint32_t *tab) {
    r ^= d;
    r = tab[r & 255] ^ (r >> 8);
    r = tab[r & 255] ^ (r >> 8);
    r = tab[r & 255] ^ (r >> 8);
    r = tab[r & 255] ^ (r >> 8);
    return r;
}

/* Fold every complete tile of buf[*done .. upto) into the per-lane
 * accumulators.  `tab` = byte table in LDS, `kx` = constants (uniform). */
#define MZ_CRC_FOLD_TILES(acc, done, buf, upto, tab, kx)                                   \
    while ((uint64_t)(done) + MZ_CRC_TILE <= (uint64_t)(upto)) {                           \
        MZ_LANES {                                                                         \
            const uint8_t *_p = (buf) + (done) + 16u * (uint32_t)lane;                     \
            uint32_t _r = P(acc);                                                          \
            if ((done) != 0) _r = mz_gf2_mul_kx(_r, (kx));  /* advance over 1008 bytes */    \
            _r = mz_crc_dword(_r, mz_load_u32(_p), (tab));                                 \
            _r = mz_crc_dword(_r, mz_load_u32(_p + 4), (tab));                             \
            _r = mz_crc_dword(_r, mz_load_u32(_p + 8), (tab));                             \
            _r = mz_crc_dword(_r, mz_load_u32(_p + 12), (tab));                            \
            P(acc) = _r;                                                                   \
        }                                                                                  \
        (done) += MZ_CRC_TILE;                                                             \
    }

/* fold one little-endian dword with the four slicing tables (tab4 = slice[0..3] back to back): four independent
 * lookups instead of a chain of four */
MZ_DEV uint32_t mz_crc_dword4(uint32_t r, uint32_t d, const uint32_t *tab4) {
    const uint32_t x = r ^ d;
    return tab4[768u + (x & 255u)] ^ tab4[512u + ((x >> 8) & 255u)] ^ tab4[256u + ((x >> 16) & 255u)] ^ tab4[x >> 24];
}

/* Stand-alone CRC (k_crc32_batch): super-tiles of 4 KiB in which lane l owns the 64 bytes at 64*l, so the advance by
 * one GF(2) multiplication (x^(8*4032)) is paid once per 64 bytes of a lane instead of once per 16.  Folds every
 * complete super-tile of buf[0 .. n) into the per-lane registers; (done) = bytes folded.  tab4 = slicing tables,
 * mul4 = the advance tables, both in LDS. */
#define MZ_CRC_FOLD_SUPER(acc, done, buf, n, tab4, mul4)                                    \
    while ((uint64_t)(done) + MZ_CRC_SUPER <= (uint64_t)(n)) {                             \
        MZ_LANES {                                                                         \
            const uint8_t *_p = (buf) + (done) + 64u * (uint32_t)lane;                     \
            uint32_t _r = P(acc);                                                          \
            if ((done) != 0) /* advance over the 4032 bytes of the other lanes: GF(2)-linear, so bytewise tables */ \
                _r = (mul4)[_r & 255u] ^ (mul4)[256u + ((_r >> 8) & 255u)] ^ (mul4)[512u + ((_r >> 16) & 255u)] ^ (mul4)[768u + (_r >> 24)]; \
            for (int _k = 0; _k < 4; _k++) {                                               \
                uint32_t _q[4];                                                            \
                __builtin_memcpy(_q, _p + 16 * _k, 16); /* one 16-byte load */             \
                _r = mz_crc_dword4(_r, _q[0], (tab4));                                     \
                _r = mz_crc_dword4(_r, _q[1], (tab4));                                     \
                _r = mz_crc_dword4(_r, _q[2], (tab4));                                     \
                _r = mz_crc_dword4(_r, _q[3], (tab4));                                     \
            }                                                                              \
            P(acc) = _r;                                                                   \
        }                                                                                  \
        (done) += MZ_CRC_SUPER;                                                            \
    }
/* the advance of MZ_CRC_FOLD_SUPER_BT over the other lanes' 4032 bytes: four lookups in the bytewise advance tables where
 * they lie in global memory (mzhip_crc_tables.mul4, 4 KiB, hot in the vector L1).  Rounds 3 - 4 multiplied by x^(8*4032)
 * with 32 select-and-xor steps on constants in scalar registers: ~100 instructions per lane and super-tile where this is
 * 15 and a memory round trip -- K1 is short of issue slots, not of latency: +3.5 % on 64 KiB entries, +1.3 % on 8 KiB
 * (profiles/r5/call15_probe.log, call16).  The dword steps stay on the byte table in LDS: the slicing tables through global
 * memory (four independent gathers per dword) measured 5 % slower. */
#define MZ_CRC_ADVANCE_SUPER(r, kxp) mz_crc_advance_gm((r), (const uint32_t *)((const uint8_t *)(kxp) + (offsetof(mzhip_crc_tables, mul4) - offsetof(mzhip_crc_tables, kx4))))
MZ_DEV uint32_t mz_crc_advance_gm(uint32_t r, const uint32_t *mul4) {
    return mul4[r & 255u] ^ mul4[256u + ((r >> 8) & 255u)] ^ mul4[512u + ((r >> 16) & 255u)] ^ mul4[768u + (r >> 24)];
}
/* The same super-tiles with the byte table alone (no slicing / advance tables in LDS): K1's fused epilogue, where LDS
 * decides the occupancy.  The advance over the other lanes' 4032 bytes (MZ_CRC_ADVANCE_SUPER; kx4 = &tabs->kx4[0], the
 * tables are found from there) is paid once per 64 bytes of a lane instead of once per 16 as with the 1 KiB tiles. */
#define MZ_CRC_FOLD_SUPER_BT(acc, done, buf, upto, tab, kx4)                               \
    while ((uint64_t)(done) + MZ_CRC_SUPER <= (uint64_t)(upto)) {                          \
        MZ_LANES {                                                                         \
            const uint8_t *_p = (buf) + (done) + 64u * (uint32_t)lane;                     \
            uint32_t _r = P(acc);                                                          \
            if ((done) != 0) _r = MZ_CRC_ADVANCE_SUPER(_r, (kx4));                         \
            for (int _k = 0; _k < 4; _k++) {                                               \
                uint32_t _q[4];                                                            \
                __builtin_memcpy(_q, _p + 16 * _k, 16); /* one 16-byte load */             \
                _r = mz_crc_dword(_r, _q[0], (tab));                                       \
                _r = mz_crc_dword(_r, _q[1], (tab));                                       \
                _r = mz_crc_dword(_r, _q[2], (tab));                                       \
                _r = mz_crc_dword(_r, _q[3], (tab));                                       \
            }                                                                              \
            P(acc) = _r;                                                                   \
        }                                                                                  \
        (done) += MZ_CRC_SUPER;                                                            \
    }
/* finish a CRC whose complete super-tiles of buf[0 .. n) were folded by MZ_CRC_FOLD_SUPER_BT: collapse the 64 strips,
 * then the remaining < 4 KiB with 1 KiB tiles and the byte-granular tail.  Result (uniform) in `result`. */
#define MZ_CRC_FINISH_SUPER_BT(result, acc, tmp, done, buf, n, tab, tabs)                  \
    do {                                                                                   \
        uint32_t _sreg = 0xFFFFFFFFu;                                                      \
        if ((done) != 0) {                                                                 \
            MZ_CRC_SUPER_REDUCE(_sreg, acc, tmp, tabs);                                    \
        }                                                                                  \
        const uint8_t *_rest = (buf) + (done);                                             \
        const uint32_t _nrest = (uint32_t)((n) - (done));                                  \
        uint32_t _rdone = 0;                                                               \
        MZ_LANES { P(acc) = (lane == 0) ? _sreg : 0u; }                                    \
        MZ_CRC_FOLD_TILES(acc, _rdone, _rest, _nrest, tab, (tabs)->kx);                    \
        MZ_CRC_FINISH_FROM(result, acc, tmp, _rdone, _rest, _nrest, tab, tabs, _sreg);     \
    } while (0)
/* collapse the super-tile registers into the raw register after byte (done) - 1 (uniform) */
#define MZ_CRC_SUPER_REDUCE(reg, acc, tmp, tabs)                                           \
    do {                                                                                   \
        MZ_LANES { P(tmp) = mz_gf2_mul(P(acc), (tabs)->x64[63 - lane]); }                  \
        MZ_WAVE_XOR(reg, tmp);                                                             \
    } while (0)

/* Finish: buf[0..n) has had its floor(n/1024) tiles folded.  Returns the
 * finished CRC-32 in `result` (uniform). `tmp` is a PV(uint32_t) scratch. */
#define MZ_CRC_FINISH_FROM(result, acc, tmp, done, buf, n, tab, tabs, reg0)                \
    do {                                                                                   \
        uint32_t _reg = (reg0);                                                            \
        if ((done) != 0) {                                                                 \
            MZ_LANES { P(tmp) = mz_gf2_mul(P(acc), (tabs)->x16[63 - lane]); }              \
            MZ_WAVE_XOR(_reg, tmp);                                                        \
        }                                                                                  \
        uint32_t _r = (uint32_t)((n) - (done)); /* tail bytes, < 1024 */                   \
        if (_r != 0) {                                                                     \
            MZ_LANES {                                                                     \
                uint32_t _o = 16u * (uint32_t)lane;                                        \
                uint32_t _v = _r > _o ? (_r - _o > 16u ? 16u : _r - _o) : 0u;              \
                uint32_t _c = (lane == 0) ? _reg : 0u;                                     \
                const uint8_t *_p = (buf) + (done) + _o;                                   \
                for (uint32_t _i = 0; _i < _v; _i++)                                       \
                    _c = (tab)[(_c ^ _p[_i]) & 255] ^ (_c >> 8);                           \
                uint32_t _rem = _r - _o - _v; /* bytes after this lane's piece */          \
                if (_v == 0) { _c = 0; _rem = 0; }                                         \
                _c = mz_gf2_mul(_c, (tabs)->x16[_rem >> 4]);                               \
                _c = mz_gf2_mul(_c, (tabs)->x1[_rem & 15]);                                \
                P(tmp) = _c;                                                               \
            }                                                                              \
            MZ_WAVE_XOR(_reg, tmp);                                                        \
        }                                                                                  \
        (result) = ~_reg;                                                                  \
    } while (0)

/* reg0 = the raw register before byte 0 (0xFFFFFFFF for a fresh CRC, ~value when chaining) */
#define MZ_CRC_FINISH(result, acc, tmp, done, buf, n, tab, tabs) \
    MZ_CRC_FINISH_FROM(result, acc, tmp, done, buf, n, tab, tabs, 0xFFFFFFFFu)

#endif
