// Design study behind K1's span walks: a pass of the lock-step walks is as long as its slowest lane,
// and the slowest lanes are the literal-dense ones.  How many SIMT steps per window are left if one step may take
// up to K literals and then one more token of any kind (K = 0: one token per step, the r2 kernel)?
// Not part of the product or of the test suite.  Input: see deflate_sync.c.
//   gcc -O2 -o /tmp/span_multi tests/study/span_multi.c && /tmp/span_multi /tmp/streams.bin 256
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
typedef struct { uint16_t count[16], symbol[288]; } huff_t;
static const uint8_t *in; static size_t in_len;
static inline uint32_t bits_at(uint64_t pos, int n) {
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
static int token(const huff_t *lc, const huff_t *dc, uint64_t *pos) { // 0 ok, 1 EOB, -1 invalid
    int sym = decode(lc, pos);
    if (sym < 0) return -1;
    if (sym < 256) return 2;
    if (sym == 256) return 1;
    sym -= 257; if (sym >= 29) return -1;
    *pos += k_lext[sym];
    int ds = decode(dc, pos);
    if (ds < 0 || ds >= 30) return -1;
    *pos += k_dext[ds];
    return 0;
}
#define NK 4
static uint64_t n_win, steps_k[NK], maxlane_k[NK], meanlane_k[NK], passes;
// walk one span; returns the crossing; steps[k] = SIMT steps of this lane when a step takes up to k literals + one token
static uint64_t walk(const huff_t *lc, const huff_t *dc, uint64_t q, uint64_t lim, int *steps) {
    int lits[NK] = {0}, st[NK] = {0};
    while (q < lim) {
        int r = token(lc, dc, &q);
        if (r < 0) break;
        for (int k = 0; k < NK; k++) {
            if (r == 2 && lits[k] < k) lits[k]++;          /* rides along with the step's final token */
            else { st[k]++; lits[k] = 0; }
        }
        if (r == 1) break;
    }
    for (int k = 0; k < NK; k++) steps[k] = st[k] + (lits[k] ? 1 : 0);
    return q;
}
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb"); int S = atoi(argv[2]);
    uint32_t n; if (!f || fread(&n, 4, 1, f) != 1) return 1;
    for (uint32_t e = 0; e < n; e++) {
        uint32_t len; if (fread(&len, 4, 1, f) != 1) return 1;
        uint8_t *buf = malloc(len + 8); if (fread(buf, 1, len, f) != len) return 1; memset(buf + len, 0, 8);
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
            uint64_t p = pos; for (;;) { int r = token(&lc, &dc, &p); if (r == 1 || r < 0) break; }
            uint64_t end = p;
            uint64_t W = pos;
            while (W + 2ull * S + 64 < end) {
                int nact = (int)((end - 64 - W) / S); if (nact > 64) nact = 64;
                uint64_t st[64], cr[64]; int ns[64][NK];
                for (int i = 0; i < nact; i++) st[i] = W + (uint64_t)i * S;
                int pass = 0, moved = 1;
                while (moved && pass < 6) {
                    int mx[NK] = {0};
                    for (int i = 0; i < nact; i++) {
                        cr[i] = walk(&lc, &dc, st[i], W + (uint64_t)(i + 1) * S, ns[i]);
                        for (int k = 0; k < NK; k++) if (ns[i][k] > mx[k]) mx[k] = ns[i][k];
                    }
                    for (int k = 0; k < NK; k++) steps_k[k] += mx[k];
                    pass++; moved = 0;
                    for (int i = 1; i < nact; i++) if (cr[i - 1] != st[i]) { st[i] = cr[i - 1]; moved = 1; }
                }
                for (int k = 0; k < NK; k++) { int mx = 0, sum = 0; for (int i = 0; i < nact; i++) { if (ns[i][k] > mx) mx = ns[i][k]; sum += ns[i][k]; } maxlane_k[k] += mx; meanlane_k[k] += sum * 100 / nact; }
                passes += pass; n_win++;
                W = cr[nact - 1];
            }
            pos = end;
        }
        free(buf);
    }
    printf("S=%d: %llu windows, %.2f passes\n", S, (unsigned long long)n_win, (double)passes / n_win);
    for (int k = 0; k < NK; k++)
        printf("  up to %d literal(s) + 1 token per step: final walk mean %.1f / slowest lane %.1f steps; all passes %.1f steps per window\n",
               k, meanlane_k[k] / 100.0 / n_win, (double)maxlane_k[k] / n_win, (double)steps_k[k] / n_win);
    return 0;
}
