/* crc32.c -- oracle restatement of mz_crypt_crc32_update (TEST INFRASTRUCTURE).
 *
 * Follows the in-tree fallback of the reference, mz_crypt.c:51-90: reflected
 * polynomial 0xEDB88320, state inverted on entry and on exit (:81,:90), one
 * 256-entry table step per byte (:84).  The table is generated from the
 * polynomial (appnote.txt:837-847) instead of being spelled out.
 */
#include "oracle.h"

static uint32_t g_tab[256];
static int g_tab_ready;

static void build_table(void) {
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++)
            c = (c & 1) ? (0xEDB88320u ^ (c >> 1)) : (c >> 1);
        g_tab[n] = c;
    }
    g_tab_ready = 1;
}

uint32_t orc_crc32_update(uint32_t value, const uint8_t *buf, size_t size) {
    if (!g_tab_ready)
        build_table();
    value = ~value; /* mz_crypt.c:81 */
    while (size--) /* mz_crypt.c:83-88 */
        value = (value >> 8) ^ g_tab[(value ^ *buf++) & 0xFF];
    return ~value; /* mz_crypt.c:90 */
}

/* multiply two polynomials over GF(2) modulo P, reflected bit order
 * (bit 31 = x^0).  */
static uint32_t gf2_mulmod(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (int i = 0; i < 32; i++) {
        if (a & 0x80000000u)
            p ^= b;
        a <<= 1;
        b = (b & 1) ? ((b >> 1) ^ 0xEDB88320u) : (b >> 1);
    }
    return p;
}

/* x^(8*n) mod P by square-and-multiply */
static uint32_t gf2_xpow8n(uint64_t n) {
    uint32_t r = 0x80000000u;   /* x^0 */
    uint32_t sq = 0x00800000u;  /* x^8 */
    while (n) {
        if (n & 1)
            r = gf2_mulmod(r, sq);
        sq = gf2_mulmod(sq, sq);
        n >>= 1;
    }
    return r;
}

uint32_t orc_crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    /* crc(A||B) = crc(A)*x^(8|B|) + crc(B)  (the pre/post inversions cancel) */
    return gf2_mulmod(gf2_xpow8n(len2), crc1) ^ crc2;
}
