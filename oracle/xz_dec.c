/* xz_dec.c -- oracle restatement of .xz decoding as the reference drives it for method 95 (TEST INFRASTRUCTURE).
 *
 * mz_stream_lzma_open (mz_strm_lzma.c:127-128) hands a method-95 entry to liblzma's
 * lzma_stream_decoder(memlimit = UINT64_MAX, flags = 0) and mz_stream_lzma_read (:147-241) maps every failure
 * to MZ_DATA_ERROR (:236-237).  liblzma is not under /root/reference (5.2.5 in this image), so this file restates
 * the published container -- "The .xz File Format" 1.0.4 (tukaani.org/xz/xz-file-format.txt): stream header and
 * footer (2.1.1, 2.1.2), block header (3.1), block padding and check (3.3, 3.4), index (4), variable-length
 * integers (1.2) -- and the LZMA2 chunk layer (control byte, sizes, property byte, reset classes) around the
 * LZMA1 packet loop of lzma_model.h.  Integrity checks: none, CRC32, CRC64 (ECMA-182), SHA-256 (FIPS 180-4);
 * other check IDs are skipped unverified, as lzma_stream_decoder does without LZMA_TELL_UNSUPPORTED_CHECK.
 * flags = 0 means: exactly one stream, no stream padding, no concatenation.
 *
 * Filter chain: LZMA2 last (what mz_stream_lzma_open writes alone, mz_strm_lzma.c:86-89,106), behind it up to three
 * of the filters liblzma 5.2.5 knows -- Delta (0x03) and the BCJ filters x86 (0x04), PowerPC (0x05), IA-64 (0x06), ARM
 * (0x07), ARM-Thumb (0x08), SPARC (0x09) -- as lzma_stream_decoder(flags 0) accepts them behind mz_strm_lzma.c:127-128.
 * The filters are third-party code that is not under /root/reference (liblzma 5.2.5: src/liblzma/delta/delta_decoder.c,
 * src/liblzma/simple/{x86,powerpc,ia64,arm,armthumb,sparc}.c, the chain rules of common/filter_common.c); restated here
 * from the format (".xz File Format" 1.0.4, 5.3) and pinned against liblzma itself in tests/test_oracle.py (Python's
 * lzma module is that library).  Anything else in the chain is LZMA_OPTIONS_ERROR there, a data error here.
 */
#include "lzma_model.h"

#define ORC_UNSUPPORTED (-109)

/* ---- checks -------------------------------------------------------------------------------------------- */
static uint64_t crc64_tab[256];
static void crc64_init(void) {
    if (crc64_tab[1])
        return;
    for (unsigned i = 0; i < 256; i++) {
        uint64_t r = i;
        for (int k = 0; k < 8; k++)
            r = (r >> 1) ^ ((r & 1) ? 0xC96C5795D7870F42ull : 0);
        crc64_tab[i] = r;
    }
}
uint64_t orc_crc64(const uint8_t *p, size_t n) {
    crc64_init();
    uint64_t c = ~0ull;
    while (n--)
        c = crc64_tab[(uint8_t)c ^ *p++] ^ (c >> 8);
    return ~c;
}

static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha256_block(uint32_t h[8], const uint8_t *b) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++)
        w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = h[0], bb = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = hh + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        hh = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    h[0] += a; h[1] += bb; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}
void orc_sha256(const uint8_t *p, size_t n, uint8_t out[32]) {
    uint32_t h[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    size_t full = n / 64;
    for (size_t i = 0; i < full; i++)
        sha256_block(h, p + 64 * i);
    uint8_t tail[128];
    size_t r = n - 64 * full;
    memset(tail, 0, sizeof(tail));
    memcpy(tail, p + 64 * full, r);
    tail[r] = 0x80;
    size_t tl = r + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)n * 8;
    for (int i = 0; i < 8; i++)
        tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    sha256_block(h, tail);
    if (tl == 128)
        sha256_block(h, tail + 64);
    for (int i = 0; i < 8; i++)
        for (int k = 0; k < 4; k++)
            out[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
}

/* ---- byte cursor --------------------------------------------------------------------------------------- */
typedef struct {
    const uint8_t *in;
    size_t len, pos;
    int eof;
} cur_t;

static int need(cur_t *c, size_t n) {
    if (c->len - c->pos < n) {
        c->eof = 1;
        c->pos = c->len;
        return 0;
    }
    return 1;
}
static uint32_t le32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
/* file format 1.2: 1-9 bytes, 7 bits each, little endian, minimal encoding; <0 = malformed, -2 = out of input */
static int vli(const uint8_t *p, size_t avail, size_t *used, uint64_t *v) {
    uint64_t r = 0;
    for (size_t i = 0; i < 9; i++) {
        if (i == avail)
            return -2;
        uint8_t b = p[i];
        if (b == 0 && i != 0)
            return -1;
        r |= (uint64_t)(b & 0x7F) << (7 * i);
        if (!(b & 0x80)) {
            *used = i + 1;
            *v = r;
            return 0;
        }
    }
    return -1;
}
static int check_size(unsigned id) {
    static const uint8_t sz[16] = {0, 4, 4, 4, 8, 8, 8, 16, 16, 16, 32, 32, 32, 64, 64, 64};
    return sz[id];
}
/* running digest of the (unpadded size, uncompressed size) sequence: blocks on one side, index records on the other */
static uint64_t mix(uint64_t h, uint64_t v) {
    h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    return h * 0xFF51AFD7ED558CCDull;
}

/* ---- LZMA2 (one block's compressed data) --------------------------------------------------------------- */
static int32_t lzma2_block(lz_t *z, cur_t *c) {
    int need_props = 1, need_dict_reset = 1;
    for (;;) {
        if (!need(c, 1))
            return ORC_BUF_ERROR;
        unsigned ctl = c->in[c->pos++];
        if (ctl == 0)
            return ORC_OK;
        if (ctl >= 0xE0 || ctl == 1) {
            need_props = 1;
            need_dict_reset = 0;
            z->dict_start = z->opos;
        } else if (need_dict_reset) {
            return ORC_DATA_ERROR;
        }
        if (ctl >= 0x80) {
            if (!need(c, 4))
                return ORC_BUF_ERROR;
            size_t usize = (((size_t)ctl & 0x1F) << 16) + ((size_t)c->in[c->pos] << 8) + c->in[c->pos + 1] + 1;
            size_t csize = ((size_t)c->in[c->pos + 2] << 8) + c->in[c->pos + 3] + 1;
            c->pos += 4;
            if (ctl >= 0xC0) {
                if (!need(c, 1))
                    return ORC_BUF_ERROR;
                unsigned d = c->in[c->pos++];
                if (d > (4 * 5 + 4) * 9 + 8)
                    return ORC_DATA_ERROR;
                unsigned lc = d % 9;
                d /= 9;
                unsigned lp = d % 5, pb = d / 5;
                if (lc + lp > 4)
                    return ORC_DATA_ERROR;
                z->lc = lc;
                z->lp = lp;
                z->pb = pb;
                need_props = 0;
                lz_reset_state(z);
            } else if (need_props) {
                return ORC_DATA_ERROR;
            } else if (ctl >= 0xA0) {
                lz_reset_state(z);
            }
            /* the chunk's compressed bytes are a self-contained range-coder run */
            int short_in = c->len - c->pos < csize;
            z->rc.in = c->in + c->pos;
            z->rc.in_len = short_in ? c->len - c->pos : csize;
            z->rc.in_pos = 0;
            z->rc.eof = 0;
            z->rc.range = 0xFFFFFFFFu;
            z->rc.code = 0;
            if (z->rc.in_len > 0 && z->rc.in[0] != 0)
                return ORC_DATA_ERROR; /* liblzma 5.2.5: the first range-coder byte must be 0x00 (observed via oracle/_ref) */
            for (int i = 0; i < 5; i++)
                z->rc.code = (z->rc.code << 8) | rc_byte(&z->rc);
            int32_t r = z->rc.eof ? ORC_DATA_ERROR : lz_run(z, z->opos + usize, 1);
            if (r == ORC_OK) {
                rc_norm(&z->rc);
                if (!z->rc.eof && (z->rc.code != 0 || z->rc.in_pos != csize))
                    return ORC_DATA_ERROR;
            }
            if (z->rc.eof) {
                /* ran off the chunk: off the input as well, or the chunk's compressed size was a lie */
                c->pos = c->len;
                c->eof = short_in;
                return short_in ? ORC_BUF_ERROR : ORC_DATA_ERROR;
            }
            if (r != ORC_OK)
                return r;
            c->pos += csize;
        } else {
            if (ctl > 2)
                return ORC_DATA_ERROR;
            if (!need(c, 2))
                return ORC_BUF_ERROR;
            size_t n = ((size_t)c->in[c->pos] << 8) + c->in[c->pos + 1] + 1;
            c->pos += 2;
            while (n--) {
                if (!need(c, 1))
                    return ORC_BUF_ERROR;
                if (z->opos == z->out_cap)
                    return ORC_OUT_FULL;
                z->out[z->opos++] = c->in[c->pos++];
            }
        }
    }
}

/* ---- the filters in front of LZMA2, decoding direction, over one whole block (a filter only looks at the bytes and at
 * their position in the block, so one call over the block equals liblzma's buffered calls) ---- */
static void unfilter_delta(uint8_t *b, size_t n, unsigned dist) {
    for (size_t i = dist; i < n; i++) b[i] = (uint8_t)(b[i] + b[i - dist]);
}
static int msbyte86(uint8_t b) { return b == 0 || b == 0xFF; }
static void unfilter_x86(uint8_t *b, size_t n, uint32_t now) {
    static const uint8_t allowed[8] = {1, 1, 1, 0, 1, 0, 0, 0}, bitnum[8] = {0, 1, 2, 2, 3, 3, 3, 3};
    uint32_t prev_mask = 0, prev_pos = (uint32_t)-5;
    if (n < 5) return;
    if (now - prev_pos > 5) prev_pos = now - 5;
    const size_t limit = n - 5;
    size_t i = 0;
    while (i <= limit) {
        uint8_t c = b[i];
        if (c != 0xE8 && c != 0xE9) {
            i++;
            continue;
        }
        const uint32_t off = now + (uint32_t)i - prev_pos;
        prev_pos = now + (uint32_t)i;
        if (off > 5) {
            prev_mask = 0;
        } else {
            for (uint32_t k = 0; k < off; k++) {
                prev_mask &= 0x77;
                prev_mask <<= 1;
            }
        }
        c = b[i + 4];
        if (msbyte86(c) && allowed[(prev_mask >> 1) & 7] && (prev_mask >> 1) < 0x10) {
            uint32_t src = ((uint32_t)c << 24) | ((uint32_t)b[i + 3] << 16) | ((uint32_t)b[i + 2] << 8) | b[i + 1], dest;
            for (;;) {
                dest = src - (now + (uint32_t)i + 5);
                if (prev_mask == 0) break;
                const uint32_t k = bitnum[prev_mask >> 1];
                c = (uint8_t)(dest >> (24 - k * 8));
                if (!msbyte86(c)) break;
                src = dest ^ ((1u << (32 - k * 8)) - 1);
            }
            b[i + 4] = (uint8_t)~(((dest >> 24) & 1) - 1);
            b[i + 3] = (uint8_t)(dest >> 16);
            b[i + 2] = (uint8_t)(dest >> 8);
            b[i + 1] = (uint8_t)dest;
            i += 5;
            prev_mask = 0;
        } else {
            i++;
            prev_mask |= 1;
            if (msbyte86(c)) prev_mask |= 0x10;
        }
    }
}
static void unfilter_powerpc(uint8_t *b, size_t n, uint32_t now) {
    for (size_t i = 0; i + 4 <= n; i += 4)
        if ((b[i] >> 2) == 0x12 && (b[i + 3] & 3) == 1) {
            const uint32_t src = ((uint32_t)(b[i] & 3) << 24) | ((uint32_t)b[i + 1] << 16) | ((uint32_t)b[i + 2] << 8) | (b[i + 3] & ~3u);
            const uint32_t dest = src - (now + (uint32_t)i);
            b[i] = (uint8_t)(0x48 | ((dest >> 24) & 3));
            b[i + 1] = (uint8_t)(dest >> 16);
            b[i + 2] = (uint8_t)(dest >> 8);
            b[i + 3] = (uint8_t)((b[i + 3] & 3) | (dest & ~3u));
        }
}
static void unfilter_ia64(uint8_t *b, size_t n, uint32_t now) {
    static const uint8_t branch[32] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 6, 6, 0, 0, 7, 7, 4, 4, 0, 0, 4, 4, 0, 0};
    for (size_t i = 0; i + 16 <= n; i += 16) {
        const uint32_t mask = branch[b[i] & 0x1F];
        uint32_t bit_pos = 5;
        for (unsigned slot = 0; slot < 3; slot++, bit_pos += 41) {
            if (((mask >> slot) & 1) == 0) continue;
            const size_t byte_pos = bit_pos >> 3;
            const uint32_t bit_res = bit_pos & 7;
            uint64_t ins = 0;
            for (unsigned j = 0; j < 6; j++) ins += (uint64_t)b[i + j + byte_pos] << (8 * j);
            uint64_t norm = ins >> bit_res;
            if (((norm >> 37) & 0xF) == 0x5 && ((norm >> 9) & 0x7) == 0) {
                uint32_t src = (uint32_t)((norm >> 13) & 0xFFFFF);
                src |= (uint32_t)((norm >> 36) & 1) << 20;
                src <<= 4;
                uint32_t dest = src - (now + (uint32_t)i);
                dest >>= 4;
                norm &= ~((uint64_t)0x8FFFFF << 13);
                norm |= (uint64_t)(dest & 0xFFFFF) << 13;
                norm |= (uint64_t)(dest & 0x100000) << (36 - 20);
                ins &= ((uint64_t)1 << bit_res) - 1;
                ins |= norm << bit_res;
                for (unsigned j = 0; j < 6; j++) b[i + j + byte_pos] = (uint8_t)(ins >> (8 * j));
            }
        }
    }
}
static void unfilter_arm(uint8_t *b, size_t n, uint32_t now) {
    for (size_t i = 0; i + 4 <= n; i += 4)
        if (b[i + 3] == 0xEB) {
            uint32_t src = ((uint32_t)b[i + 2] << 16) | ((uint32_t)b[i + 1] << 8) | b[i];
            src <<= 2;
            uint32_t dest = src - (now + (uint32_t)i + 8);
            dest >>= 2;
            b[i + 2] = (uint8_t)(dest >> 16);
            b[i + 1] = (uint8_t)(dest >> 8);
            b[i] = (uint8_t)dest;
        }
}
static void unfilter_armthumb(uint8_t *b, size_t n, uint32_t now) {
    for (size_t i = 0; i + 4 <= n; i += 2)
        if ((b[i + 1] & 0xF8) == 0xF0 && (b[i + 3] & 0xF8) == 0xF8) {
            uint32_t src = ((uint32_t)(b[i + 1] & 7) << 19) | ((uint32_t)b[i] << 11) | ((uint32_t)(b[i + 3] & 7) << 8) | b[i + 2];
            src <<= 1;
            uint32_t dest = src - (now + (uint32_t)i + 4);
            dest >>= 1;
            b[i + 1] = (uint8_t)(0xF0 | ((dest >> 19) & 7));
            b[i] = (uint8_t)(dest >> 11);
            b[i + 3] = (uint8_t)(0xF8 | ((dest >> 8) & 7));
            b[i + 2] = (uint8_t)dest;
            i += 2;
        }
}
static void unfilter_sparc(uint8_t *b, size_t n, uint32_t now) {
    for (size_t i = 0; i + 4 <= n; i += 4)
        if ((b[i] == 0x40 && (b[i + 1] & 0xC0) == 0x00) || (b[i] == 0x7F && (b[i + 1] & 0xC0) == 0xC0)) {
            uint32_t src = ((uint32_t)b[i] << 24) | ((uint32_t)b[i + 1] << 16) | ((uint32_t)b[i + 2] << 8) | b[i + 3];
            src <<= 2;
            uint32_t dest = src - (now + (uint32_t)i);
            dest >>= 2;
            dest = (((0 - ((dest >> 22) & 1)) << 22) & 0x3FFFFFFF) | (dest & 0x3FFFFF) | 0x40000000;
            b[i] = (uint8_t)(dest >> 24);
            b[i + 1] = (uint8_t)(dest >> 16);
            b[i + 2] = (uint8_t)(dest >> 8);
            b[i + 3] = (uint8_t)dest;
        }
}

int32_t orc_xz_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, int64_t max_out, size_t *in_used,
                      size_t *out_len) {
    static const uint8_t magic[6] = {0xFD, '7', 'z', 'X', 'Z', 0x00};
    int32_t ret = ORC_DATA_ERROR;
    cur_t c = {in, in_len, 0, 0};
    lz_t *z = (lz_t *)calloc(1, sizeof(lz_t));
    if (!z)
        return ORC_DATA_ERROR;
    z->out = out;
    z->out_cap = out_cap;
    z->lit = (uint16_t *)malloc(((size_t)0x300 << 4) * sizeof(uint16_t));
    if (!z->lit)
        goto done;

    /* stream header: magic, flags {0x00, check id}, CRC32(flags) */
    if (!need(&c, 12)) {
        ret = ORC_BUF_ERROR;
        goto done;
    }
    if (memcmp(in, magic, 6) != 0 || le32(in + 8) != orc_crc32_update(0, in + 6, 2) || in[6] != 0 || (in[7] & 0xF0))
        goto done;
    const unsigned check = in[7];
    c.pos = 12;

    uint64_t nblocks = 0, blocks_digest = 0;
    for (;;) {
        if (!need(&c, 1)) {
            ret = ORC_BUF_ERROR;
            goto done;
        }
        if (in[c.pos] == 0)
            break; /* index indicator */
        /* block header (3.1) */
        const size_t hpos = c.pos, hsize = ((size_t)in[c.pos] + 1) * 4;
        if (!need(&c, hsize)) {
            ret = ORC_BUF_ERROR;
            goto done;
        }
        if (le32(in + hpos + hsize - 4) != orc_crc32_update(0, in + hpos, hsize - 4))
            goto done;
        const unsigned bflags = in[hpos + 1];
        if (bflags & 0x3C)
            goto done; /* reserved bits: LZMA_OPTIONS_ERROR */
        size_t p = hpos + 2, used;
        const size_t hend = hpos + hsize - 4;
        uint64_t want_csize = UINT64_MAX, want_usize = UINT64_MAX, v;
        if (bflags & 0x40) {
            if (vli(in + p, hend - p, &used, &want_csize) != 0 || want_csize == 0)
                goto done;
            p += used;
        }
        if (bflags & 0x80) {
            if (vli(in + p, hend - p, &used, &want_usize) != 0)
                goto done;
            p += used;
        }
        uint64_t dict = 0;
        unsigned pre_id[3], npre = 0; /* the filters in front of LZMA2, in header order */
        uint32_t pre_arg[3];          /* delta: distance; BCJ: start offset */
        const unsigned nfilt = (bflags & 3u) + 1u;
        for (unsigned f = 0; f < nfilt; f++) {
            uint64_t id, psize;
            if (vli(in + p, hend - p, &used, &id) != 0)
                goto done;
            p += used;
            if (vli(in + p, hend - p, &used, &psize) != 0 || psize > hend - p - used)
                goto done;
            p += used;
            if (f + 1 == nfilt) { /* the last filter must be LZMA2 (filter_common.c: last_ok) */
                if (id != 0x21 || psize != 1 || in[p] > 40)
                    goto done; /* LZMA_OPTIONS_ERROR */
                dict = in[p] == 40 ? 0xFFFFFFFFull : (uint64_t)(2u | (in[p] & 1u)) << (in[p] / 2 + 11);
            } else if (id == 0x03) { /* Delta: one byte, distance - 1 */
                if (psize != 1)
                    goto done;
                pre_id[npre] = 3;
                pre_arg[npre++] = (uint32_t)in[p] + 1u;
            } else if (id >= 0x04 && id <= 0x09) { /* BCJ: no properties, or a 32-bit start offset */
                static const uint8_t align[6] = {1, 4, 16, 4, 2, 4};
                uint32_t so = 0;
                if (psize == 4)
                    so = le32(in + p);
                else if (psize != 0)
                    goto done;
                if (so & (align[id - 4] - 1u))
                    goto done; /* LZMA_OPTIONS_ERROR: misaligned start offset */
                pre_id[npre] = (unsigned)id;
                pre_arg[npre++] = so;
            } else {
                goto done; /* unknown filter, or LZMA2 where it cannot stand: LZMA_OPTIONS_ERROR */
            }
            p += (size_t)psize;
        }
        for (; p < hend; p++)
            if (in[p] != 0)
                goto done; /* header padding must be zero */
        (void)v;
        if (dict < 4096)
            dict = 4096;
        z->dict = (dict + 15) & ~(uint64_t)15;
        c.pos = hpos + hsize;

        /* compressed data, padding to a multiple of four, check */
        const size_t data_pos = c.pos, out_pos0 = z->opos;
        int32_t r = lzma2_block(z, &c);
        if (r != ORC_OK) {
            ret = r;
            goto done;
        }
        const uint64_t csize = c.pos - data_pos, usize = z->opos - out_pos0;
        if ((want_csize != UINT64_MAX && want_csize != csize) || (want_usize != UINT64_MAX && want_usize != usize))
            goto done;
        for (unsigned f = npre; f-- > 0;) { /* undo the filters, last applied first */
            uint8_t *b = out + out_pos0;
            const size_t n = (size_t)usize;
            switch (pre_id[f]) {
            case 3: unfilter_delta(b, n, pre_arg[f]); break;
            case 4: unfilter_x86(b, n, pre_arg[f]); break;
            case 5: unfilter_powerpc(b, n, pre_arg[f]); break;
            case 6: unfilter_ia64(b, n, pre_arg[f]); break;
            case 7: unfilter_arm(b, n, pre_arg[f]); break;
            case 8: unfilter_armthumb(b, n, pre_arg[f]); break;
            default: unfilter_sparc(b, n, pre_arg[f]); break;
            }
        }
        while ((c.pos - data_pos) & 3) {
            if (!need(&c, 1)) {
                ret = ORC_BUF_ERROR;
                goto done;
            }
            if (in[c.pos++] != 0)
                goto done;
        }
        const size_t cs = (size_t)check_size(check);
        if (!need(&c, cs)) {
            ret = ORC_BUF_ERROR;
            goto done;
        }
        if (check == 1 && le32(in + c.pos) != orc_crc32_update(0, out + out_pos0, (size_t)usize))
            goto done;
        if (check == 4) {
            uint64_t k = orc_crc64(out + out_pos0, (size_t)usize);
            if (le32(in + c.pos) != (uint32_t)k || le32(in + c.pos + 4) != (uint32_t)(k >> 32))
                goto done;
        }
        if (check == 10) {
            uint8_t dg[32];
            orc_sha256(out + out_pos0, (size_t)usize, dg);
            if (memcmp(dg, in + c.pos, 32) != 0)
                goto done;
        }
        c.pos += cs;
        nblocks++;
        blocks_digest = mix(mix(blocks_digest, hsize + csize + cs), usize);
    }

    /* index (4): indicator 0x00, record count, records, padding, CRC32 */
    {
        const size_t ipos = c.pos;
        size_t p = ipos + 1, used;
        uint64_t count, rec_digest = 0;
        int e = vli(in + p, in_len - p, &used, &count);
        if (e != 0) {
            ret = e == -2 ? ORC_BUF_ERROR : ORC_DATA_ERROR;
            c.pos = e == -2 ? in_len : c.pos;
            goto done;
        }
        p += used;
        if (count != nblocks)
            goto done;
        for (uint64_t i = 0; i < count; i++) {
            uint64_t unpadded, usize;
            if ((e = vli(in + p, in_len - p, &used, &unpadded)) == 0) {
                p += used;
                e = vli(in + p, in_len - p, &used, &usize);
            }
            if (e != 0) {
                ret = e == -2 ? ORC_BUF_ERROR : ORC_DATA_ERROR;
                c.pos = e == -2 ? in_len : c.pos;
                goto done;
            }
            p += used;
            rec_digest = mix(mix(rec_digest, unpadded), usize);
        }
        if (rec_digest != blocks_digest)
            goto done;
        c.pos = p;
        while ((c.pos - ipos) & 3) {
            if (!need(&c, 1)) {
                ret = ORC_BUF_ERROR;
                goto done;
            }
            if (in[c.pos++] != 0)
                goto done;
        }
        if (!need(&c, 4)) {
            ret = ORC_BUF_ERROR;
            goto done;
        }
        if (le32(in + c.pos) != orc_crc32_update(0, in + ipos, c.pos - ipos))
            goto done;
        c.pos += 4;
        /* stream footer (2.1.2): CRC32, backward size, flags, "YZ" */
        const size_t isize = c.pos - ipos;
        if (!need(&c, 12)) {
            ret = ORC_BUF_ERROR;
            goto done;
        }
        const uint8_t *f = in + c.pos;
        c.pos += 12;
        if (f[10] != 'Y' || f[11] != 'Z' || le32(f) != orc_crc32_update(0, f + 4, 6) || f[8] != 0 || f[9] != check ||
            ((uint64_t)le32(f + 4) + 1) * 4 != isize)
            goto done;
        ret = ORC_OK;
    }

done:
    if (in_used)
        *in_used = c.pos;
    if (out_len)
        *out_len = (max_out >= 0 && (int64_t)z->opos > max_out) ? (size_t)max_out : z->opos;
    free(z->lit);
    free(z);
    return ret;
}
