// emul.cpp -- 64-lane HOST EMULATION of the device cores (TEST INFRASTRUCTURE).
//
// Compiles minizip-ng_amd/csrc/*_core.h with -DMZHIP_HOST_EMUL (wave.h turns per-lane regions
// into loops over lane = 0..63) so the kernels' control flow can be checked against the oracle
// in a container without a GPU.  Never linked into libmzhip.so; the product path is the HIP
// build of the same headers.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* emul_inflate() runs the product configuration (span path on); emul_inflate_steps() the step loop alone */
#include "inflate_core.h"
#if defined(MZ_STATS)
unsigned long long mz_stats[24], mz_stat_max;
extern "C" unsigned long long *emul_stats() { return mz_stats; }
#endif

static mzhip_crc_tables g_tabs;

static void ready() { /* (several host threads decode at once behind the prime cache: initialised exactly once) */
    static const int once = (mzhip_crc_tables_init(&g_tabs), 1);
    (void)once;
}

extern "C" int32_t emul_inflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                                uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_inflate_lds *L = (mz_inflate_lds *)malloc(sizeof(mz_inflate_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_inflate_result r;
    uint8_t *rec = (uint8_t *)malloc(MZ_REC_BYTES + 1024); /* the wave's HBM record scratch (chase window) */
    memset(rec, 0x5A, MZ_REC_BYTES + 1024);
    mz_inflate_entry(in, in_len, out, out_cap, L, g_tabs.byte_tab, &g_tabs, 1u, rec, (const mz_inflate_state *)0, (mz_inflate_state *)0, &r, (const mz_inflate_par *)0);
    free(rec);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}

/* resumable decode (mzhip_inflate_resume_host's device side): out[0 .. st_in->out_pos) is history */
extern "C" int32_t emul_inflate_resume(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, const uint32_t *st_in,
                                       uint32_t *st_out, uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_inflate_lds *L = (mz_inflate_lds *)malloc(sizeof(mz_inflate_lds));
    memset(L, 0xA5, sizeof(*L));
    uint8_t *rec = (uint8_t *)malloc(MZ_REC_BYTES + 1024);
    memset(rec, 0x5A, MZ_REC_BYTES + 1024);
    mz_inflate_state a, b;
    memset(&a, 0, sizeof(a));
    if (st_in) memcpy(&a, st_in, sizeof(a));
    memset(&b, 0, sizeof(b));
    mz_inflate_result r;
    mz_inflate_entry(in, in_len, out, out_cap, L, g_tabs.byte_tab, &g_tabs, 1u, rec, &a, st_out ? &b : (mz_inflate_state *)0, &r, (const mz_inflate_par *)0);
    if (st_out) memcpy(st_out, &b, sizeof(b));
    free(rec);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}

/* ---- one large entry on many waves (mzhip_inflate_parallel_host): the device functions behind its steps ---- */
extern "C" uint32_t emul_find_blocks(const uint8_t *in, uint32_t in_len, uint32_t b0, uint32_t b1, uint32_t *out, uint32_t cap) {
    uint32_t n = 0;
    for (uint32_t p = b0; p < b1; p++)
        if (mz_block_header_plausible(in, in_len, p)) {
            if (n < cap) out[n] = p;
            n++;
        }
    return n;
}
/* one candidate block: res = {status, end bit, out position behind its last byte, BFINAL} */
extern "C" void emul_inflate_block(const uint8_t *in, uint32_t in_len, uint32_t bit, uint32_t pos, uint32_t mode, uint8_t *out,
                                   uint32_t *ptr, uint32_t *res4) {
    ready();
    mz_inflate_lds *L = (mz_inflate_lds *)malloc(sizeof(mz_inflate_lds));
    memset(L, 0xA5, sizeof(*L));
    uint8_t *rec = (uint8_t *)malloc(MZ_REC_BYTES + 1024);
    memset(rec, 0x5A, MZ_REC_BYTES + 1024);
    mz_inflate_one_block(in, in_len, bit, pos, mode, out, ptr, L, g_tabs.byte_tab, &g_tabs, rec, res4);
    free(rec);
    free(L);
}
/* the serial decode with "stop in front of the next block header" is emul_inflate_resume with flags bit 1 */

/* the step loop alone (the span path is off) */
extern "C" int32_t emul_inflate_steps(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                                      uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_inflate_lds *L = (mz_inflate_lds *)malloc(sizeof(mz_inflate_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_inflate_result r;
    mz_inflate_entry(in, in_len, out, out_cap, L, g_tabs.byte_tab, &g_tabs, 0u, (uint8_t *)0, (const mz_inflate_state *)0, (mz_inflate_state *)0, &r, (const mz_inflate_par *)0);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}

extern "C" uint32_t emul_crc32(const uint8_t *buf, uint32_t n) {
    ready();
    uint32_t acc[64], tmp[64], done = 0, result;
    for (int l = 0; l < 64; l++) acc[l] = l == 0 ? 0xFFFFFFFFu : 0u;
    const uint32_t *tab = g_tabs.byte_tab;
    const mzhip_crc_tables *tabs = &g_tabs;
    MZ_CRC_FOLD_TILES(acc, done, buf, n, tab, tabs->kx);
    MZ_CRC_FINISH(result, acc, tmp, done, buf, n, tab, tabs);
    return result;
}

#include "adler32_core.h"

extern "C" uint32_t emul_adler32(const uint8_t *buf, uint32_t n) {
    uint32_t result;
    MZ_ADLER32(result, buf, n);
    return result;
}

extern "C" uint32_t emul_adler32_combine(uint32_t a, uint32_t b, uint64_t len_b) {
    return mzhip_adler32_combine_host(a, b, len_b);
}

#include "hash_core.h"

extern "C" uint64_t emul_crc64(const uint8_t *buf, uint64_t n) {
    static uint64_t tab[256];
    if (!tab[1]) mzhip_crc64_table_init(tab);
    uint64_t r;
    MZ_CRC64(r, buf, n, tab);
    return r;
}

extern "C" void emul_sha(const uint8_t *buf, uint64_t n, int alg, uint8_t *digest) {
    if (alg == 24 || alg == 25) {
        uint64_t g[8];
        mz_sha512_init(g, alg == 24);
        mz_sha512_run(buf, n, g);
        for (int i = 0; i < (alg == 24 ? 6 : 8); i++)
            for (int k = 0; k < 8; k++) digest[8 * i + k] = (uint8_t)(g[i] >> (56 - 8 * k));
        return;
    }
    uint32_t h[8];
    int words = 8;
    if (alg == 20) {
        mz_sha1_run(buf, n, h);
        words = 5;
    } else {
        mz_sha256_init(h, alg == 22);
        mz_sha256_run(buf, n, h);
        words = alg == 22 ? 7 : 8;
    }
    for (int i = 0; i < words; i++)
        for (int k = 0; k < 4; k++) digest[4 * i + k] = (uint8_t)(h[i] >> (24 - 8 * k));
}

#include "xz_core.h"

extern "C" int32_t emul_xz(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                           uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_xz_lds *L = (mz_xz_lds *)malloc(sizeof(mz_xz_lds));
    memset(L, 0xA5, sizeof(*L));
    mzhip_crc64_table_init(L->crc64_tab);
    mz_lzma_result r;
    uint16_t *prx = (uint16_t *)malloc(MZ_LZMA_XPROBS * sizeof(uint16_t));
    memset(prx, 0x5A, MZ_LZMA_XPROBS * sizeof(uint16_t));
    mz_xz_entry(in, in_len, out, out_cap, max_out, L, g_tabs.byte_tab, &g_tabs, prx, &r);
    free(prx);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}

extern "C" uint32_t emul_xz_lds_bytes(void) { return (uint32_t)sizeof(mz_xz_lds); }
/* one window of one block's LZMA2 chunk sequence: state = 20 words (mz_lzma2_state), model as emul_lzma_resume */
extern "C" int32_t emul_lzma2_run(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap, const uint32_t *st_in,
                                  uint32_t *st_out, uint16_t *model, uint32_t *out_len, uint32_t *in_used) {
    ready();
    mz_xz_lds *L = (mz_xz_lds *)malloc(sizeof(mz_xz_lds));
    memset(L, 0xA5, sizeof(*L));
    mzhip_crc64_table_init(L->crc64_tab);
    mz_lzma2_state a, b;
    memcpy(&a, st_in, sizeof(a));
    memset(&b, 0, sizeof(b));
    mz_lzma_result r;
    mz_lzma2_run(in, in_len, buf, buf_cap, L, g_tabs.byte_tab, &g_tabs, model, &a, &b, &r);
    memcpy(st_out, &b, sizeof(b));
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    return r.status;
}

#include "lzma_enc_core.h"

/* links of the chain the block parse follows (MZ_LZE_DEPTH_FOR_PRESET of the preset the caller means; 0 = by the class:
 * the default preset's for four ways, one link for the fast class) */
static uint32_t g_far_depth = 0;
extern "C" void emul_set_far_depth(uint32_t d) { g_far_depth = d; }
static uint32_t far_depth_for(uint32_t ways) { return g_far_depth ? g_far_depth : (ways > 1u ? MZ_LZE_DEPTH_FOR_PRESET(6) : 1u); }

/* mode 0: ZIP method-14 payload; mode 1: raw LZMA2 chunk payload */
extern "C" int32_t emul_lzma_encode_ways(const uint8_t *in, uint32_t in_len, uint32_t mode, uint32_t ways, uint8_t *out,
                                         uint32_t out_cap, uint32_t *out_len, uint32_t *crc);
extern "C" int32_t emul_lzma_encode(const uint8_t *in, uint32_t in_len, uint32_t mode, uint8_t *out, uint32_t out_cap,
                                    uint32_t *out_len, uint32_t *crc) {
    return emul_lzma_encode_ways(in, in_len, mode, 1u, out, out_cap, out_len, crc);
}
/* ways: 1 = presets 0-3, MZ_DEF_WAYS_BEST = presets 4-9 and the default */
extern "C" int32_t emul_lzma_encode_ways(const uint8_t *in, uint32_t in_len, uint32_t mode, uint32_t ways, uint8_t *out,
                                         uint32_t out_cap, uint32_t *out_len, uint32_t *crc) {
    ready();
    const uint32_t nblocks = (in_len + MZ_DEF_BLOCK - 1) / MZ_DEF_BLOCK;
    uint32_t *tok = (uint32_t *)malloc((size_t)(nblocks ? nblocks : 1) * MZ_DEF_BLOCK * sizeof(uint32_t));
    uint32_t *ntok = (uint32_t *)calloc(nblocks ? nblocks : 1, sizeof(uint32_t));
    mz_lz_tok_lds *T = (mz_lz_tok_lds *)malloc(sizeof(mz_lz_tok_lds));
    const size_t xbytes = (MZ_DEF_WAYS_BEST - 1u) * (sizeof(uint16_t) << MZ_DEF_HBITS);
    uint16_t *xhead = (uint16_t *)malloc(xbytes);
    uint32_t *links = (uint32_t *)0;
    if ((mode == 0u && in_len > MZ_DEF_BLOCK)) { /* the chain pass (k_lz_chain_batch): one wave over the whole stream */
        uint32_t *head = (uint32_t *)malloc(sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        links = (uint32_t *)malloc((size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        memset(head, 0xA5, sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        memset(links, 0xA5, (size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        mz_lz_chain(in, in_len, links, head);
        free(head);
    }
    for (uint32_t b = 0; b < nblocks; b++) {
        memset(T, 0xA5, sizeof(*T));
        memset(xhead, 0xA5, xbytes);
        const uint32_t lo = b * MZ_DEF_BLOCK, hi = (in_len - lo < MZ_DEF_BLOCK) ? in_len : lo + MZ_DEF_BLOCK;
        ntok[b] = mz_lz_tokenize(in, lo, hi, tok + (size_t)b * MZ_DEF_BLOCK, T, ways, ways > 1u ? xhead : (uint16_t *)0, links, far_depth_for(ways));
    }
    free(xhead);
    free(links);
    mz_lzma_lds *L = (mz_lzma_lds *)malloc(sizeof(mz_lzma_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_lzma_enc_result r;
    mz_lzma_rc_encode(in, in_len, tok, ntok, mode, out, out_cap, L, g_tabs.byte_tab, &g_tabs, &r);
    free(L);
    free(T);
    free(tok);
    free(ntok);
    *out_len = r.out_len;
    *crc = r.crc;
    return r.status;
}

/* The LZMA2 chunks of one .xz block (mzhip_xz_encode_*'s device side): the block is parsed as ONE stream -- chain pass and
 * all, so a chunk's matches reach back over the chunks before it -- and every 64 KiB block of it is range-coded by itself
 * with a fresh state (LZMA2 control 0xE0 for the first chunk, 0xC0 = state and properties reset, dictionary kept, for the
 * others).  out[b * stride ...] / out_lens[b] receive chunk b's payload. */
extern "C" int32_t emul_lzma2_chunks_encode(const uint8_t *in, uint32_t in_len, uint32_t ways, uint8_t *out, uint32_t stride,
                                            uint32_t *out_lens) {
    ready();
    const uint32_t nblocks = (in_len + MZ_DEF_BLOCK - 1) / MZ_DEF_BLOCK;
    uint32_t *tok = (uint32_t *)malloc((size_t)(nblocks ? nblocks : 1) * MZ_DEF_BLOCK * sizeof(uint32_t));
    uint32_t *ntok = (uint32_t *)calloc(nblocks ? nblocks : 1, sizeof(uint32_t));
    mz_lz_tok_lds *T = (mz_lz_tok_lds *)malloc(sizeof(mz_lz_tok_lds));
    const size_t xbytes = (MZ_DEF_WAYS_BEST - 1u) * (sizeof(uint16_t) << MZ_DEF_HBITS);
    uint16_t *xhead = (uint16_t *)malloc(xbytes);
    uint32_t *links = (uint32_t *)0;
    if (in_len > MZ_DEF_BLOCK) {
        uint32_t *head = (uint32_t *)malloc(sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        links = (uint32_t *)malloc((size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        memset(head, 0xA5, sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        memset(links, 0xA5, (size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        mz_lz_chain(in, in_len, links, head);
        free(head);
    }
    for (uint32_t b = 0; b < nblocks; b++) {
        memset(T, 0xA5, sizeof(*T));
        memset(xhead, 0xA5, xbytes);
        const uint32_t lo = b * MZ_DEF_BLOCK, hi = (in_len - lo < MZ_DEF_BLOCK) ? in_len : lo + MZ_DEF_BLOCK;
        ntok[b] = mz_lz_tokenize(in, lo, hi, tok + (size_t)b * MZ_DEF_BLOCK, T, ways, ways > 1u ? xhead : (uint16_t *)0, links, far_depth_for(ways));
    }
    free(xhead);
    free(links);
    mz_lzma_lds *L = (mz_lzma_lds *)malloc(sizeof(mz_lzma_lds));
    int32_t status = 0;
    for (uint32_t b = 0; b < nblocks && status == 0; b++) {
        memset(L, 0xA5, sizeof(*L));
        const uint32_t hi = (in_len - b * MZ_DEF_BLOCK < MZ_DEF_BLOCK) ? in_len : (b + 1u) * MZ_DEF_BLOCK;
        mz_lzma_enc_result r;
        mz_lzma_rc_encode_x(in, hi, tok, ntok, 1u, out + (size_t)b * stride, stride, L, g_tabs.byte_tab, &g_tabs, &r, b,
                            (const mz_lzma_enc_state *)0, (mz_lzma_enc_state *)0, (uint16_t *)0);
        status = r.status;
        out_lens[b] = r.out_len;
    }
    free(L);
    free(T);
    free(tok);
    free(ntok);
    return status;
}

/* one segment of a method-14 stream written in segments (mzhip_lzma_encode_resume_host's device side): state = 16 words
 * (mz_lzma_enc_state), model = LZ_NUM_PROBS probabilities, both the caller's */
extern "C" int32_t emul_lzma_encode_resume(const uint8_t *in, uint32_t in_len, uint32_t skip_blocks, uint32_t last, uint32_t ways,
                                           const uint32_t *st_in, uint32_t *st_out, uint16_t *model, uint8_t *out, uint32_t out_cap,
                                           uint32_t *out_len) {
    ready();
    const uint32_t nblocks = (in_len + MZ_DEF_BLOCK - 1) / MZ_DEF_BLOCK;
    uint32_t *tok = (uint32_t *)malloc((size_t)(nblocks ? nblocks : 1) * MZ_DEF_BLOCK * sizeof(uint32_t));
    uint32_t *ntok = (uint32_t *)calloc(nblocks ? nblocks : 1, sizeof(uint32_t));
    mz_lz_tok_lds *T = (mz_lz_tok_lds *)malloc(sizeof(mz_lz_tok_lds));
    const size_t xbytes = (MZ_DEF_WAYS_BEST - 1u) * (sizeof(uint16_t) << MZ_DEF_HBITS);
    uint16_t *xhead = (uint16_t *)malloc(xbytes);
    uint32_t *links = (uint32_t *)0;
    if ((in_len > MZ_DEF_BLOCK)) { /* the chain pass (k_lz_chain_batch): one wave over the whole stream */
        uint32_t *head = (uint32_t *)malloc(sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        links = (uint32_t *)malloc((size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        memset(head, 0xA5, sizeof(uint32_t) << MZ_LZE_FAR_HBITS);
        memset(links, 0xA5, (size_t)nblocks * MZ_DEF_BLOCK * sizeof(uint32_t));
        mz_lz_chain(in, in_len, links, head);
        free(head);
    }
    for (uint32_t b = skip_blocks; b < nblocks; b++) { /* (the history blocks are linked, not parsed) */
        memset(T, 0xA5, sizeof(*T));
        memset(xhead, 0xA5, xbytes);
        const uint32_t lo = b * MZ_DEF_BLOCK, hi = (in_len - lo < MZ_DEF_BLOCK) ? in_len : lo + MZ_DEF_BLOCK;
        ntok[b] = mz_lz_tokenize(in, lo, hi, tok + (size_t)b * MZ_DEF_BLOCK, T, ways, ways > 1u ? xhead : (uint16_t *)0, links, far_depth_for(ways));
    }
    free(xhead);
    free(links);
    mz_lzma_lds *L = (mz_lzma_lds *)malloc(sizeof(mz_lzma_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_lzma_enc_state a, b;
    memset(&a, 0, sizeof(a));
    memset(&b, 0, sizeof(b));
    if (st_in) memcpy(&a, st_in, sizeof(a));
    mz_lzma_enc_result r;
    mz_lzma_rc_encode_x(in, in_len, tok, ntok, 0u, out, out_cap, L, g_tabs.byte_tab, &g_tabs, &r, skip_blocks, &a,
                        last ? (mz_lzma_enc_state *)0 : &b, model);
    if (!last && st_out) memcpy(st_out, &b, sizeof(b));
    free(L);
    free(T);
    free(tok);
    free(ntok);
    *out_len = r.out_len;
    return r.status;
}

/* the stand-alone kernel's path: 4 KiB super-tiles, then 1 KiB tiles, then the tail */
extern "C" uint32_t emul_crc32_super(const uint8_t *buf, uint32_t n, uint32_t init) {
    ready();
    uint32_t acc[64], tmp[64], done = 0, result, reg = ~init;
    for (int l = 0; l < 64; l++) acc[l] = l == 0 ? ~init : 0u;
    const uint32_t *tab = g_tabs.byte_tab;
    const mzhip_crc_tables *tabs = &g_tabs;
    MZ_CRC_FOLD_SUPER(acc, done, buf, n, &g_tabs.slice[0][0], &g_tabs.mul4[0][0]);
    if (done) {
        MZ_CRC_SUPER_REDUCE(reg, acc, tmp, tabs);
        for (int l = 0; l < 64; l++) acc[l] = l == 0 ? reg : 0u;
    }
    const uint8_t *rest = buf + done;
    const uint32_t nrest = n - done;
    uint32_t rdone = 0;
    MZ_CRC_FOLD_TILES(acc, rdone, rest, nrest, tab, tabs->kx);
    MZ_CRC_FINISH_FROM(result, acc, tmp, rdone, rest, nrest, tab, tabs, reg);
    return result;
}

extern "C" uint32_t emul_lds_bytes(void) { return (uint32_t)sizeof(mz_inflate_lds); }

#include "lzma_core.h"

extern "C" int32_t emul_lzma(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                             uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_lzma_lds *L = (mz_lzma_lds *)malloc(sizeof(mz_lzma_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_lzma_result r;
    uint16_t *prx = (uint16_t *)malloc(MZ_LZMA_XPROBS * sizeof(uint16_t));
    memset(prx, 0x5A, MZ_LZMA_XPROBS * sizeof(uint16_t));
    mz_lzma_entry(in, in_len, out, out_cap, max_out, L, g_tabs.byte_tab, &g_tabs, prx, &r);
    free(prx);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}
/* K3's slot build (MZ_LZMA_SLOTS literal contexts in LDS, the model in the scratch); -300 = MZHIP_RETRY: the stream swaps
 * too much and belongs to the full-model kernel.  *swapped: whether it was given back. */
extern "C" int32_t emul_lzma_slots(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                                   uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    ready();
    mz_lzma_lds_s *L = (mz_lzma_lds_s *)malloc(sizeof(mz_lzma_lds_s));
    memset(L, 0xA5, sizeof(*L));
    mz_lzma_result r;
    uint16_t *prx = (uint16_t *)malloc(MZ_LZMA_SPROBS * sizeof(uint16_t));
    memset(prx, 0x5A, MZ_LZMA_SPROBS * sizeof(uint16_t));
    mz_lzma_entry_s(in, in_len, out, out_cap, max_out, L, g_tabs.byte_tab, &g_tabs, prx, &r);
    free(prx);
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    *crc = r.crc;
    return r.status;
}
/* the resumable build: state = 16 words (mz_lzma_state), model = MZ_LZMA_MODEL_U16 probabilities, both owned by the caller */
extern "C" uint32_t emul_lzma_model_u16(void) { return MZ_LZMA_MODEL_U16; }
extern "C" uint32_t emul_lzma_encode_history_bytes(void) { return MZ_LZE_FAR_DICT; }
extern "C" int32_t emul_lzma_resume(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap, const uint32_t *st_in,
                                    uint32_t *st_out, uint16_t *model, uint32_t *out_len, uint32_t *in_used) {
    ready();
    mz_lzma_lds *L = (mz_lzma_lds *)malloc(sizeof(mz_lzma_lds));
    memset(L, 0xA5, sizeof(*L));
    mz_lzma_state a, b;
    memset(&a, 0, sizeof(a));
    memset(&b, 0, sizeof(b));
    if (st_in) memcpy(&a, st_in, sizeof(a));
    mz_lzma_result r;
    mz_lzma_entry_r(in, in_len, buf, buf_cap, (int64_t)-1, L, g_tabs.byte_tab, &g_tabs, model, &a, st_out ? &b : (mz_lzma_state *)0, &r);
    if (st_out) memcpy(st_out, &b, sizeof(b));
    free(L);
    *out_len = r.out_len;
    *in_used = r.in_used;
    return r.status;
}
extern "C" uint32_t emul_lzma_slots_lds_bytes(void) { return (uint32_t)sizeof(mz_lzma_lds_s); }
extern "C" uint32_t emul_lzma_lds_bytes(void) { return (uint32_t)sizeof(mz_lzma_lds); }

#include "deflate_core.h"

static uint32_t g_def_max_dist = 32768u - 262u;
extern "C" void emul_deflate_window(uint32_t window_log2) { g_def_max_dist = (1u << window_log2) - 262u; }
static uint32_t g_def_warm = 0; /* the next calls' in[] starts this many bytes in front of the piece (emul_deflate_warm) */
extern "C" void emul_deflate_warm(uint32_t warm) { g_def_warm = warm; }
static int32_t emul_deflate_ways(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t final,
                                 uint32_t ways, uint32_t parse, uint32_t *out_len, uint32_t *crc) {
    ready();
    mz_deflate_lds *L = (mz_deflate_lds *)malloc(sizeof(mz_deflate_lds));
    memset(L, 0xA5, sizeof(*L));
    uint16_t *xh = (uint16_t *)malloc((MZ_DEF_WAYS_BEST - 1u) * sizeof(uint16_t) << MZ_DEF_HBITS);
    memset(xh, 0x5A, (MZ_DEF_WAYS_BEST - 1u) * sizeof(uint16_t) << MZ_DEF_HBITS);
    mz_deflate_result r;
    uint32_t *tok = (uint32_t *)malloc(MZ_DEF_BLOCK * sizeof(uint32_t));
    if (parse) mz_deflate_piece<1u>(in, in_len, g_def_warm, out, out_cap, final, tok, L, g_tabs.byte_tab, &g_tabs, ways, xh, g_def_max_dist, &r);
    else mz_deflate_piece<0u>(in, in_len, g_def_warm, out, out_cap, final, tok, L, g_tabs.byte_tab, &g_tabs, ways, xh, g_def_max_dist, &r);
    free(tok);
    free(xh);
    free(L);
    *out_len = r.out_len;
    *crc = r.crc;
    return r.status;
}
extern "C" int32_t emul_deflate(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t final,
                                uint32_t *out_len, uint32_t *crc) {
    return emul_deflate_ways(in, in_len, out, out_cap, final, 1u, 0u, out_len, crc);
}
/* levels 7-9: four candidates per hash bucket, cost parse */
extern "C" int32_t emul_deflate_best(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t final,
                                     uint32_t *out_len, uint32_t *crc) {
    return emul_deflate_ways(in, in_len, out, out_cap, final, MZ_DEF_WAYS_BEST, 1u, out_len, crc);
}
/* the default compression class (levels 4-6 and -1): four candidates, the lazy rule decides inside every step */
extern "C" int32_t emul_deflate_lazy(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t final,
                                     uint32_t *out_len, uint32_t *crc) {
    return emul_deflate_ways(in, in_len, out, out_cap, final, MZ_DEF_WAYS_BEST, 0u, out_len, crc);
}
extern "C" uint32_t emul_deflate_lds_bytes(void) { return (uint32_t)sizeof(mz_deflate_lds); }
