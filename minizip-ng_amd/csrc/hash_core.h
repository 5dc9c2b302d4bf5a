/* hash_core.h -- the block checks of the .xz container that CRC-32 does not cover, and the hashes of SURVEY 8(f)
 * row 4: SHA-256 / SHA-224 / SHA-1 (FIPS 180-4) as per-LANE functions -- a digest chain is strictly serial inside
 * one message, so the parallel axis is messages: in k_sha_batch every lane hashes its own ZIP entry
 * (mz_zip_rw.c:465-466 feeds mz_crypt_sha_update with the decoded bytes of one entry at a time); in the .xz
 * kernel all 64 lanes of the wave that decoded the block run the same chain redundantly.  CRC-64 (ECMA-182,
 * .xz check id 4) is wave-parallel: 64 contiguous pieces, one table-driven register per lane, then a six-level
 * combine tree of GF(2) multiplications by x^(8 * piece * 2^k).
 */
#ifndef MZHIP_HASH_CORE_H
#define MZHIP_HASH_CORE_H

#include "crc32_core.h"
#include "wave.h"

#if defined(MZHIP_HOST_EMUL)
#define MZ_CONST_TABLE static const
#else
#define MZ_CONST_TABLE __device__ __constant__ static const
#endif

MZ_CONST_TABLE uint32_t mz_k256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

#define MZ_ROR32(x, n) (((x) >> (n)) | ((x) << (32 - (n))))

/* message word i (big endian) of the padded message: data, 0x80, zeros, 64-bit bit count */
MZ_DEV uint32_t mz_sha_word(const uint8_t *p, uint64_t n, uint64_t total_words, uint64_t i) {
    const uint64_t o = 4 * i;
    if (o + 4 <= n) return __builtin_bswap32(mz_load_u32(p + o));
    if (i == total_words - 1) return (uint32_t)(n << 3);
    if (i == total_words - 2) return (uint32_t)(n >> 29);
    uint32_t w = 0;
    for (uint32_t k = 0; k < 4; k++) {
        const uint64_t q = o + k;
        const uint32_t b = q < n ? p[q] : (q == n ? 0x80u : 0u);
        w |= b << (24 - 8 * k);
    }
    return w;
}

/* SHA-256 family: h[] holds the initial value on entry and the digest words on return */
MZ_DEV void mz_sha256_run(const uint8_t *p, uint64_t n, uint32_t h[8]) {
    const uint64_t blocks = (n + 9 + 63) / 64, total_words = blocks * 16;
    const uint64_t full = n / 64; /* blocks made of data only: plain 16-byte loads; the padded tail goes word by word */
    for (uint64_t b = 0; b < blocks; b++) {
        uint32_t w[16];
        if (b < full) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t q[4];
                __builtin_memcpy(q, p + 64 * b + 16 * (uint64_t)i, 16);
                w[4 * i] = __builtin_bswap32(q[0]);
                w[4 * i + 1] = __builtin_bswap32(q[1]);
                w[4 * i + 2] = __builtin_bswap32(q[2]);
                w[4 * i + 3] = __builtin_bswap32(q[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = mz_sha_word(p, n, total_words, b * 16 + (uint64_t)i);
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
        for (int i = 0; i < 64; i++) {
            if (i >= 16) {
                const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
                const uint32_t s0 = MZ_ROR32(w15, 7) ^ MZ_ROR32(w15, 18) ^ (w15 >> 3);
                const uint32_t s1 = MZ_ROR32(w2, 17) ^ MZ_ROR32(w2, 19) ^ (w2 >> 10);
                w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
            }
            const uint32_t t1 = hh + (MZ_ROR32(e, 6) ^ MZ_ROR32(e, 11) ^ MZ_ROR32(e, 25)) + ((e & f) ^ (~e & g)) + mz_k256[i] +
                                w[i & 15];
            const uint32_t t2 = (MZ_ROR32(a, 2) ^ MZ_ROR32(a, 13) ^ MZ_ROR32(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
            hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
}

MZ_DEV void mz_sha256_init(uint32_t h[8], int is224) {
    if (is224) {
        h[0] = 0xc1059ed8; h[1] = 0x367cd507; h[2] = 0x3070dd17; h[3] = 0xf70e5939;
        h[4] = 0xffc00b31; h[5] = 0x68581511; h[6] = 0x64f98fa7; h[7] = 0xbefa4fa4;
    } else {
        h[0] = 0x6a09e667; h[1] = 0xbb67ae85; h[2] = 0x3c6ef372; h[3] = 0xa54ff53a;
        h[4] = 0x510e527f; h[5] = 0x9b05688c; h[6] = 0x1f83d9ab; h[7] = 0x5be0cd19;
    }
}

MZ_DEV void mz_sha1_run(const uint8_t *p, uint64_t n, uint32_t h[5]) {
    const uint64_t blocks = (n + 9 + 63) / 64, total_words = blocks * 16;
    h[0] = 0x67452301; h[1] = 0xEFCDAB89; h[2] = 0x98BADCFE; h[3] = 0x10325476; h[4] = 0xC3D2E1F0;
    const uint64_t full = n / 64; /* blocks made of data only: plain 16-byte loads; the padded tail goes word by word */
    for (uint64_t b = 0; b < blocks; b++) {
        uint32_t w[16];
        if (b < full) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t q[4];
                __builtin_memcpy(q, p + 64 * b + 16 * (uint64_t)i, 16);
                w[4 * i] = __builtin_bswap32(q[0]);
                w[4 * i + 1] = __builtin_bswap32(q[1]);
                w[4 * i + 2] = __builtin_bswap32(q[2]);
                w[4 * i + 3] = __builtin_bswap32(q[3]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 16; i++) w[i] = mz_sha_word(p, n, total_words, b * 16 + (uint64_t)i);
        }
        uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4];
#pragma unroll
        for (int i = 0; i < 80; i++) {
            if (i >= 16) {
                const uint32_t x = w[(i - 3) & 15] ^ w[(i - 8) & 15] ^ w[(i - 14) & 15] ^ w[i & 15];
                w[i & 15] = MZ_ROR32(x, 31);
            }
            uint32_t f, k;
            if (i < 20) { f = (bb & c) | (~bb & d); k = 0x5A827999; }
            else if (i < 40) { f = bb ^ c ^ d; k = 0x6ED9EBA1; }
            else if (i < 60) { f = (bb & c) | (bb & d) | (c & d); k = 0x8F1BBCDC; }
            else { f = bb ^ c ^ d; k = 0xCA62C1D6; }
            const uint32_t t = MZ_ROR32(a, 27) + f + e + k + w[i & 15];
            e = d; d = c; c = MZ_ROR32(bb, 2); bb = a; a = t;
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e;
    }
}

/* ---- SHA-512 / SHA-384 (FIPS 180-4): 128-byte blocks, 80 rounds on 64-bit words ---------------------------- */
MZ_CONST_TABLE uint64_t mz_k512[80] = {
    0x428a2f98d728ae22ull, 0x7137449123ef65cdull, 0xb5c0fbcfec4d3b2full, 0xe9b5dba58189dbbcull,
    0x3956c25bf348b538ull, 0x59f111f1b605d019ull, 0x923f82a4af194f9bull, 0xab1c5ed5da6d8118ull,
    0xd807aa98a3030242ull, 0x12835b0145706fbeull, 0x243185be4ee4b28cull, 0x550c7dc3d5ffb4e2ull,
    0x72be5d74f27b896full, 0x80deb1fe3b1696b1ull, 0x9bdc06a725c71235ull, 0xc19bf174cf692694ull,
    0xe49b69c19ef14ad2ull, 0xefbe4786384f25e3ull, 0x0fc19dc68b8cd5b5ull, 0x240ca1cc77ac9c65ull,
    0x2de92c6f592b0275ull, 0x4a7484aa6ea6e483ull, 0x5cb0a9dcbd41fbd4ull, 0x76f988da831153b5ull,
    0x983e5152ee66dfabull, 0xa831c66d2db43210ull, 0xb00327c898fb213full, 0xbf597fc7beef0ee4ull,
    0xc6e00bf33da88fc2ull, 0xd5a79147930aa725ull, 0x06ca6351e003826full, 0x142929670a0e6e70ull,
    0x27b70a8546d22ffcull, 0x2e1b21385c26c926ull, 0x4d2c6dfc5ac42aedull, 0x53380d139d95b3dfull,
    0x650a73548baf63deull, 0x766a0abb3c77b2a8ull, 0x81c2c92e47edaee6ull, 0x92722c851482353bull,
    0xa2bfe8a14cf10364ull, 0xa81a664bbc423001ull, 0xc24b8b70d0f89791ull, 0xc76c51a30654be30ull,
    0xd192e819d6ef5218ull, 0xd69906245565a910ull, 0xf40e35855771202aull, 0x106aa07032bbd1b8ull,
    0x19a4c116b8d2d0c8ull, 0x1e376c085141ab53ull, 0x2748774cdf8eeb99ull, 0x34b0bcb5e19b48a8ull,
    0x391c0cb3c5c95a63ull, 0x4ed8aa4ae3418acbull, 0x5b9cca4f7763e373ull, 0x682e6ff3d6b2b8a3ull,
    0x748f82ee5defb2fcull, 0x78a5636f43172f60ull, 0x84c87814a1f0ab72ull, 0x8cc702081a6439ecull,
    0x90befffa23631e28ull, 0xa4506cebde82bde9ull, 0xbef9a3f7b2c67915ull, 0xc67178f2e372532bull,
    0xca273eceea26619cull, 0xd186b8c721c0c207ull, 0xeada7dd6cde0eb1eull, 0xf57d4f7fee6ed178ull,
    0x06f067aa72176fbaull, 0x0a637dc5a2c898a6ull, 0x113f9804bef90daeull, 0x1b710b35131c471bull,
    0x28db77f523047d84ull, 0x32caab7b40c72493ull, 0x3c9ebe0a15c9bebcull, 0x431d67c49c100d4cull,
    0x4cc5d4becb3e42b6ull, 0x597f299cfc657e2aull, 0x5fcb6fab3ad6faecull, 0x6c44198c4a475817ull};

#define MZ_ROR64(x, n) (((x) >> (n)) | ((x) << (64 - (n))))

MZ_DEV uint64_t mz_sha512_word(const uint8_t *p, uint64_t n, uint64_t total_words, uint64_t i) {
    const uint64_t o = 8 * i;
    if (o + 8 <= n) {
        uint64_t v;
        __builtin_memcpy(&v, p + o, 8);
        return __builtin_bswap64(v);
    }
    if (i == total_words - 1) return n << 3; /* low half of the 128-bit bit count */
    if (i == total_words - 2) return n >> 61;
    uint64_t w = 0;
    for (uint32_t k = 0; k < 8; k++) {
        const uint64_t q = o + k;
        const uint64_t b = q < n ? p[q] : (q == n ? 0x80u : 0u);
        w |= b << (56 - 8 * k);
    }
    return w;
}

MZ_DEV void mz_sha512_init(uint64_t h[8], int is384) {
    const uint64_t iv512[8] = {0x6a09e667f3bcc908ull, 0xbb67ae8584caa73bull, 0x3c6ef372fe94f82bull, 0xa54ff53a5f1d36f1ull, 0x510e527fade682d1ull, 0x9b05688c2b3e6c1full, 0x1f83d9abfb41bd6bull, 0x5be0cd19137e2179ull};
    const uint64_t iv384[8] = {0xcbbb9d5dc1059ed8ull, 0x629a292a367cd507ull, 0x9159015a3070dd17ull, 0x152fecd8f70e5939ull, 0x67332667ffc00b31ull, 0x8eb44a8768581511ull, 0xdb0c2e0d64f98fa7ull, 0x47b5481dbefa4fa4ull};
    for (int i = 0; i < 8; i++) h[i] = is384 ? iv384[i] : iv512[i];
}

MZ_DEV void mz_sha512_run(const uint8_t *p, uint64_t n, uint64_t h[8]) {
    const uint64_t blocks = (n + 17 + 127) / 128, total_words = blocks * 16;
    for (uint64_t b = 0; b < blocks; b++) {
        /* the 16-word schedule window stays in registers: every index below is a compile-time constant (the rounds
         * run as 5 x 16 with the inner 16 unrolled); a block that lies wholly inside the message is fetched with plain
         * 8-byte loads, only the one or two blocks that carry the padding go through the byte-wise path */
        uint64_t w[16];
        if ((b + 1) * 128 <= n) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                uint64_t v;
                __builtin_memcpy(&v, p + b * 128 + 8 * (uint64_t)i, 8);
                w[i] = __builtin_bswap64(v);
            }
        } else {
            MZ_NOUNROLL
            for (int i = 0; i < 16; i++) {
                const uint64_t v = mz_sha512_word(p, n, total_words, b * 16 + (uint64_t)i);
#pragma unroll
                for (int j = 0; j < 16; j++) w[j] = (j == i) ? v : w[j]; /* selects, not an indexed private array */
            }
        }
        uint64_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
        MZ_NOUNROLL
        for (int r = 0; r < 5; r++) {
#pragma unroll
            for (int j = 0; j < 16; j++) {
                if (r > 0) {
                    const uint64_t w15 = w[(j + 1) & 15], w2 = w[(j + 14) & 15];
                    const uint64_t s0 = MZ_ROR64(w15, 1) ^ MZ_ROR64(w15, 8) ^ (w15 >> 7);
                    const uint64_t s1 = MZ_ROR64(w2, 19) ^ MZ_ROR64(w2, 61) ^ (w2 >> 6);
                    w[j] = w[j] + s0 + w[(j + 9) & 15] + s1;
                }
                const uint64_t t1 = hh + (MZ_ROR64(e, 14) ^ MZ_ROR64(e, 18) ^ MZ_ROR64(e, 41)) + ((e & f) ^ (~e & g)) +
                                    mz_k512[16 * r + j] + w[j];
                const uint64_t t2 = (MZ_ROR64(a, 28) ^ MZ_ROR64(a, 34) ^ MZ_ROR64(a, 39)) + ((a & bb) ^ (a & c) ^ (bb & c));
                hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
            }
        }
        h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
    }
}

/* ---- CRC-64 (ECMA-182 reflected, 0xC96C5795D7870F42), wave-parallel ---------------------------------------- */
#define MZ_CRC64_POLY 0xC96C5795D7870F42ull

static inline void mzhip_crc64_table_init(uint64_t *t) {
    for (uint32_t i = 0; i < 256; i++) {
        uint64_t r = i;
        for (int k = 0; k < 8; k++) r = (r >> 1) ^ ((r & 1) ? MZ_CRC64_POLY : 0);
        t[i] = r;
    }
}

/* a * b mod P64 on reflected polynomials (bit 63 = x^0) */
MZ_DEV uint64_t mz_gf2_mul64(uint64_t a, uint64_t b) {
    uint64_t p = 0;
    for (int i = 0; i < 64; i++) {
        p ^= b & (0ull - (a >> 63));
        a <<= 1;
        b = (b >> 1) ^ (MZ_CRC64_POLY & (0ull - (b & 1)));
    }
    return p;
}

/* result (uniform) = CRC-64/XZ of buf[0..n); tab64 = 256-entry table (LDS) */
#define MZ_CRC64(result, buf, n, tab64) MZ_CRC64_FROM(result, 0ull, buf, n, tab64)
/* ... taken up from `prev`, the CRC-64 of the bytes in front of buf (0 for none): the value of both together */
#define MZ_CRC64_FROM(result, prev, buf, n, tab64)                                                       \
    do {                                                                                                 \
        const uint64_t _n = (n);                                                                         \
        const uint64_t _init = ~(uint64_t)(prev); /* the raw register before byte 0 */                   \
        uint64_t _res = (prev);                                                                          \
        if (_n != 0) {                                                                                   \
            const uint64_t _piece = (_n + 63) / 64;                                                      \
            PV(uint32_t, _lo);                                                                           \
            PV(uint32_t, _hi);                                                                           \
            PV(uint32_t, _olo);                                                                          \
            PV(uint32_t, _ohi);                                                                          \
            /* pieces are aligned to the END of the buffer, so every lane is followed by (63 - lane) whole pieces */ \
            MZ_LANES {                                                                                   \
                const int64_t _s = (int64_t)_n - (int64_t)(64 - lane) * (int64_t)_piece;                 \
                const int64_t _e = _s + (int64_t)_piece;                                                 \
                uint64_t _r = (_s <= 0 && _e > 0) ? _init : 0ull;                                        \
                for (int64_t _i = _s < 0 ? 0 : _s; _i < _e; _i++)                                        \
                    _r = (tab64)[(uint8_t)_r ^ (buf)[_i]] ^ (_r >> 8);                                   \
                P(_lo) = (uint32_t)_r;                                                                   \
                P(_hi) = (uint32_t)(_r >> 32);                                                           \
            }                                                                                            \
            /* X = x^(8 * piece): square-and-multiply on uniform values */                               \
            uint64_t _x = 1ull << 63, _sq = 1ull << 55; /* x^0, x^8 */                                   \
            for (uint64_t _b = _piece; _b; _b >>= 1) {                                                   \
                if (_b & 1) _x = mz_gf2_mul64(_x, _sq);                                                  \
                _sq = mz_gf2_mul64(_sq, _sq);                                                            \
            }                                                                                            \
            for (int _k = 1; _k < 64; _k <<= 1) {                                                        \
                MZ_GATHER(_olo, _lo, lane + _k);                                                         \
                MZ_GATHER(_ohi, _hi, lane + _k);                                                         \
                MZ_LANES {                                                                               \
                    const uint64_t _r = mz_gf2_mul64(((uint64_t)P(_hi) << 32) | P(_lo), _x) ^            \
                                        (((uint64_t)P(_ohi) << 32) | P(_olo));                           \
                    P(_lo) = (uint32_t)_r;                                                               \
                    P(_hi) = (uint32_t)(_r >> 32);                                                       \
                }                                                                                        \
                _x = mz_gf2_mul64(_x, _x);                                                               \
            }                                                                                            \
            _res = ~(((uint64_t)MZ_READLANE(_hi, 0) << 32) | MZ_READLANE(_lo, 0));                       \
        }                                                                                                \
        (result) = _res;                                                                                 \
    } while (0)

#endif
