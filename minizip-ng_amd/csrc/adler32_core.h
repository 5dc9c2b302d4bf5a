/* adler32_core.h -- wave-parallel Adler-32 (the zlib-wrapper checksum, RFC 1950): needed only when the zlib
 * stream is opened with a positive window_bits (mz_strm_zlib.c:348-350 COMPRESS_WINDOW).
 *   a = 1 + sum(d_i) mod 65521,  b = sum of the running a mod 65521,  adler = b << 16 | a.
 * A buffer is folded 1 KiB at a time: lane l owns 16 bytes, contributes a_l = sum(bytes) and
 * b_l = sum((16 - j) * byte_j); a byte's weight in b is the number of bytes from it to the end, so the tile's
 * b is sum_l(b_l + bytes_behind_lane_l * a_l) and the running pair advances as
 *   B += tile_len * A + tile_b,  A += tile_a   (mod 65521, wave-uniform).
 */
#ifndef MZHIP_ADLER32_CORE_H
#define MZHIP_ADLER32_CORE_H

#include "crc32_core.h"
#include "wave.h"

#define MZ_ADLER_MOD 65521u

/* host arithmetic on checksums only: adler(A||B) from adler(A), adler(B), |B| */
static inline uint32_t mzhip_adler32_combine_host(uint32_t ad1, uint32_t ad2, uint64_t len2) {
    const uint32_t M = MZ_ADLER_MOD;
    const uint32_t a1 = ad1 & 0xFFFFu, b1 = ad1 >> 16, a2 = ad2 & 0xFFFFu, b2 = ad2 >> 16;
    const uint32_t rem = (uint32_t)(len2 % M);
    const uint32_t a = (a1 + a2 + M - 1u) % M;
    const uint32_t b = (uint32_t)(((uint64_t)b1 + b2 + (uint64_t)rem * ((a1 + M - 1u) % M)) % M);
    return (b << 16) | a;
}

/* result (uniform) = Adler-32 of buf[0..n), starting from 1 */
#define MZ_ADLER32(result, buf, n)                                                                     \
    do {                                                                                               \
        uint32_t _A = 1u, _B = 0u;                                                                     \
        PV(uint32_t, _pa);                                                                             \
        PV(uint32_t, _pb);                                                                             \
        for (uint64_t _o = 0; _o < (uint64_t)(n); _o += 1024u) {                                       \
            const uint32_t _r = ((uint64_t)(n) - _o < 1024u) ? (uint32_t)((uint64_t)(n) - _o) : 1024u; \
            MZ_LANES {                                                                                 \
                const uint32_t _lo = 16u * (uint32_t)lane;                                             \
                const uint32_t _v = _r > _lo ? (_r - _lo > 16u ? 16u : _r - _lo) : 0u;                 \
                const uint8_t *_p = (buf) + _o + _lo;                                                  \
                uint32_t _a = 0, _b = 0;                                                               \
                for (uint32_t _j = 0; _j < _v; _j++) {                                                 \
                    _a += _p[_j];                                                                      \
                    _b += (_v - _j) * _p[_j];                                                          \
                }                                                                                      \
                const uint32_t _behind = _v ? (_r - _lo - _v) : 0u;                                    \
                P(_pa) = _a;                                                                           \
                P(_pb) = _b + _behind * _a;                                                            \
            }                                                                                          \
            uint32_t _sa, _sb;                                                                         \
            MZ_WAVE_SUM(_sa, _pa);                                                                     \
            MZ_WAVE_SUM(_sb, _pb);                                                                     \
            _B = (uint32_t)(((uint64_t)_B + (uint64_t)_r * _A + _sb) % MZ_ADLER_MOD);                  \
            _A = (_A + _sa) % MZ_ADLER_MOD;                                                            \
        }                                                                                              \
        (result) = (_B << 16) | _A;                                                                    \
    } while (0)

#endif
