/* lzma_dec.c -- oracle restatement of raw-LZMA1 decoding (TEST INFRASTRUCTURE).
 *
 * The reference decodes a method-14 entry by stripping the 4-byte ZIP-LZMA
 * magic (mz_strm_lzma.c:118-124), turning the 5 property bytes plus eight
 * 0xFF bytes into a ".lzma alone" header with unknown size (:177-201) and
 * calling liblzma's lzma_alone_decoder / lzma_code (:126,:210).  liblzma is
 * not under /root/reference (5.2.5 in this image; un-pinned upstream), and the
 * range-coder algorithm is not described anywhere in the reference tree, so
 * this file restates the public-domain LZMA specification (LZMA SDK,
 * "lzma-specification.txt"): 11-bit adaptive probabilities, 12-state machine,
 * rep0..rep3, two length coders, 6-bit position slots, aligned low bits, and
 * the end-of-stream marker (distance 0xFFFFFFFF).  Container framing follows
 * doc/zip/appnote.txt:2232-2275.
 *
 * liblzma behaviours mirrored on purpose (observed on 5.2.5 via oracle/_ref):
 *   - the first range-coder byte must be 0x00 (else LZMA_DATA_ERROR, nothing consumed);
 *   - a distance is valid iff it is < min(bytes produced, dictionary size),
 *     dictionary size rounded up to >= 4096 and to a multiple of 16;
 *   - after the EOS marker the coder normalises once and requires code == 0;
 *   - every failure surfaces as MZ_DATA_ERROR (-3) from mz_stream_lzma_read
 *     (mz_strm_lzma.c:236-237), including truncated input.
 */
#include "lzma_model.h"

int32_t orc_lzma_zip_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap, int64_t max_out,
                            size_t *in_used, size_t *out_len) {
    int32_t ret = ORC_DATA_ERROR;
    lz_t *z = NULL;

    if (in_used) *in_used = 0;
    if (out_len) *out_len = 0;
    /* 4-byte magic (ignored by the reference, mz_strm_lzma.c:118-121) + 5 props */
    if (in_len < 9) {
        if (in_used) *in_used = in_len;
        return ORC_DATA_ERROR;
    }
    unsigned d = in[4];
    if (d >= 9 * 5 * 5) {
        if (in_used) *in_used = 9;
        return ORC_DATA_ERROR; /* LZMA_FORMAT_ERROR -> MZ_DATA_ERROR */
    }
    z = (lz_t *)calloc(1, sizeof(lz_t));
    if (!z)
        return ORC_DATA_ERROR;
    z->lc = d % 9;
    d /= 9;
    z->lp = d % 5;
    z->pb = d / 5;
    if (z->lc + z->lp > 4) {
        /* liblzma 5.2.5 lzma_lzma_lclppb_decode(): lc + lp > LZMA_LCLP_MAX is refused with the header (LZMA_FORMAT_ERROR from
         * the alone decoder -> MZ_DATA_ERROR, mz_strm_lzma.c:236), although the LZMA specification allows lc up to 8.  The
         * device refused it all along (lzma_entry.inc); the restatement did not until tests/fuzz_oracle_lzma.py, round 6. */
        free(z);
        if (in_used) *in_used = 9;
        return ORC_DATA_ERROR;
    }
    uint64_t dict = in[5] | ((uint32_t)in[6] << 8) | ((uint32_t)in[7] << 16) | ((uint32_t)in[8] << 24);
    if (dict < 4096)
        dict = 4096;
    z->dict = (dict + 15) & ~(uint64_t)15;
    z->out = out;
    z->out_cap = out_cap;
    z->lit = (uint16_t *)malloc(((size_t)0x300 << (z->lc + z->lp)) * sizeof(uint16_t));
    if (!z->lit)
        goto done;
    lz_reset_state(z);

    z->rc.in = in;
    z->rc.in_len = in_len;
    z->rc.in_pos = 9;
    z->rc.range = 0xFFFFFFFFu;
    if (in_len > 9 && in[9] != 0)
        goto done; /* liblzma 5.2.5 rejects a non-zero first byte before consuming it */
    for (int i = 0; i < 5; i++)
        z->rc.code = (z->rc.code << 8) | rc_byte(&z->rc);
    if (z->rc.eof)
        goto done;
    ret = lz_run(z, (size_t)-1, 0);

done:
    if (in_used)
        *in_used = z->rc.in ? z->rc.in_pos : 9;
    if (out_len)
        *out_len = (max_out >= 0 && (int64_t)z->opos > max_out) ? (size_t)max_out : z->opos;
    free(z->lit);
    free(z);
    return ret;
}
