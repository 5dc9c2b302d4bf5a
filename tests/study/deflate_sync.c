// Design study behind K1's span path: how fast does a DEFLATE token walk started at an arbitrary bit
// re-synchronise with the true token sequence?  Not part of the product or of the test suite.
//   python -c "import sys,zlib,struct; sys.path.insert(0,'.'); from tests import synth; sl=synth.slices(400,65536,1234); \
//     f=open('/tmp/streams.bin','wb'); f.write(struct.pack('<I',len(sl))); \
//     [(lambda z: (f.write(struct.pack('<I',len(z))), f.write(z)))((lambda c,d: c.compress(d)+c.flush())(zlib.compressobj(6,zlib.DEFLATED,-15),d)) for d in sl]"
//   gcc -O2 -o /tmp/deflate_sync tests/study/deflate_sync.c && for S in 128 256 512 1024; do /tmp/deflate_sync /tmp/streams.bin $S; done
// Input: u32 count, then per stream u32 length + raw DEFLATE bytes.  Output: for spans of S bits, the share of walks that
// have met the true parse within a given distance (results in profiles/r1/side_measurements.log).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef struct { uint16_t count[16], symbol[288]; } huff_t;
static const uint8_t *in; static size_t in_len;
static inline uint32_t bits_at(uint64_t pos, int n) { // n<=16
    uint64_t by = pos >> 3; uint32_t v = 0;
    for (int k = 0; k < 4; k++) if (by + k < in_len) v |= (uint32_t)in[by + k] << (8 * k);
    return (v >> (pos & 7)) & ((1u << n) - 1);
}
static int build(huff_t *h, const uint8_t *length, int n) {
    uint16_t offs[16]; memset(h->count, 0, sizeof(h->count));
    for (int i = 0; i < n; i++) h->count[length[i]]++;
    int left = 1;
    for (int len = 1; len <= 15; len++) { left <<= 1; left -= h->count[len]; if (left < 0) return left; }
    offs[1] = 0; for (int len = 1; len < 15; len++) offs[len + 1] = offs[len] + h->count[len];
    for (int i = 0; i < n; i++) if (length[i]) h->symbol[offs[length[i]]++] = (uint16_t)i;
    return left;
}
static int decode(const huff_t *h, uint64_t *pos) {
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        code |= (int)bits_at(*pos, 1); (*pos)++;
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
    }
    return -2;
}
static const uint8_t k_lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t k_dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
// one token from pos; returns 0 ok, 1 EOB, -1 invalid
static int token(const huff_t *lc, const huff_t *dc, uint64_t *pos) {
    int sym = decode(lc, pos);
    if (sym < 0) return -1;
    if (sym < 256) return 0;
    if (sym == 256) return 1;
    sym -= 257; if (sym >= 29) return -1;
    *pos += k_lext[sym];
    int ds = decode(dc, pos);
    if (ds < 0 || ds >= 30) return -1;
    *pos += k_dext[ds];
    return 0;
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int S = atoi(argv[2]);
    uint32_t n; fread(&n, 4, 1, f);
    uint64_t hist[64] = {0}, lanes = 0, never = 0, invalid = 0, tot_tokens = 0, tot_bits = 0, wasted_tok = 0, blocks = 0;
    uint64_t sync_tok_sum = 0;
    for (uint32_t e = 0; e < n; e++) {
        uint32_t len; fread(&len, 4, 1, f);
        uint8_t *buf = malloc(len + 8); fread(buf, 1, len, f); memset(buf + len, 0, 8);
        in = buf; in_len = len;
        uint64_t pos = 0; int last = 0;
        while (!last) {
            last = bits_at(pos, 1); int type = bits_at(pos + 1, 2); pos += 3;
            huff_t lc, dc;
            if (type == 0) { pos = (pos + 7) & ~7ull; uint32_t l = bits_at(pos, 16); pos += 32 + 8ull * l; continue; }
            if (type == 1) { uint8_t L[288]; int i = 0; for (; i < 144; i++) L[i] = 8; for (; i < 256; i++) L[i] = 9; for (; i < 280; i++) L[i] = 7; for (; i < 288; i++) L[i] = 8; build(&lc, L, 288); for (i = 0; i < 30; i++) L[i] = 5; build(&dc, L, 30); }
            else {
                int nlen = bits_at(pos, 5) + 257, ndist = bits_at(pos + 5, 5) + 1, ncode = bits_at(pos + 10, 4) + 4; pos += 14;
                static const uint8_t order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};
                uint8_t L[320]; memset(L, 0, sizeof(L)); uint8_t cl[19]; memset(cl, 0, 19);
                for (int i = 0; i < ncode; i++) { cl[order[i]] = bits_at(pos, 3); pos += 3; }
                huff_t ch; build(&ch, cl, 19);
                int idx = 0;
                while (idx < nlen + ndist) {
                    int sym = decode(&ch, &pos);
                    if (sym < 16) L[idx++] = sym;
                    else { int prev = 0, rep; if (sym == 16) { prev = L[idx - 1]; rep = 3 + bits_at(pos, 2); pos += 2; } else if (sym == 17) { rep = 3 + bits_at(pos, 3); pos += 3; } else { rep = 11 + bits_at(pos, 7); pos += 7; } while (rep--) L[idx++] = prev; }
                }
                build(&lc, L, nlen); build(&dc, L + nlen, ndist);
            }
            blocks++;
            // true boundaries
            uint64_t start = pos; size_t cap = 1 << 16, nb = 0; uint64_t *tb = malloc(cap * 8);
            for (;;) { if (nb == cap) { cap *= 2; tb = realloc(tb, cap * 8); } tb[nb++] = pos; int r = token(&lc, &dc, &pos); if (r) break; }
            uint64_t end = pos; // after EOB
            tot_tokens += nb; tot_bits += end - start;
            // mark boundaries in a bitmap
            uint64_t span = end - start + 64; uint8_t *isb = calloc(span, 1);
            for (size_t i = 0; i < nb; i++) isb[tb[i] - start] = 1;
            // lanes: windows of 64*S bits starting at true boundaries
            size_t ti = 0;
            while (ti < nb) {
                uint64_t W = tb[ti];
                for (int lane = 1; lane < 64; lane++) {
                    uint64_t p = W + (uint64_t)lane * S; if (p >= end) break;
                    lanes++;
                    uint64_t q = p; int toks = 0, ok = 0;
                    while (q < end && q < p + 4ull * S) {
                        if (isb[q - start]) { ok = 1; break; }
                        int r = token(&lc, &dc, &q); toks++;
                        if (r < 0) { ok = -1; break; }
                        if (r == 1) { ok = -2; break; }
                    }
                    if (ok == 1) { uint64_t d = q - p; int b = (int)(d * 16 / S); if (b > 63) b = 63; hist[b]++; sync_tok_sum += toks; }
                    else if (ok < 0) invalid++; else never++;
                }
                // next window
                uint64_t nextW = W + 64ull * S; while (ti < nb && tb[ti] < nextW) ti++;
            }
            free(tb); free(isb);
        }
        free(buf);
    }
    printf("S=%d bits: entries=%u blocks=%llu tokens=%llu bits/token=%.2f\n", S, n, (unsigned long long)blocks, (unsigned long long)tot_tokens, (double)tot_bits / tot_tokens);
    printf("lanes=%llu invalid-or-eob=%llu (%.2f%%) never(4S)=%llu  mean tokens to sync=%.1f\n", (unsigned long long)lanes, (unsigned long long)invalid, 100.0 * invalid / lanes, (unsigned long long)never, (double)sync_tok_sum / (lanes - invalid - never + 1));
    uint64_t cum = 0; printf("sync distance / S (cumulative %% of lanes): ");
    for (int b = 0; b < 64; b++) { cum += hist[b]; if (b == 1 || b == 3 || b == 7 || b == 11 || b == 15 || b == 23 || b == 31 || b == 47 || b == 63) printf(" <=%.2fS:%.1f%%", (b + 1) / 16.0, 100.0 * cum / lanes); }
    printf("\n");
    return 0;
}
