// Design study behind K1's round-3 window: "decode once, then chase".
//   pass 1   lane i walks its span of S bits from the span's first bit (lane 0: the true cursor) until it crosses into
//            span i + 1, and RECORDS its steps (a step = a leading literal + the token behind it);
//   pass 2   lane i keeps walking from its crossing, inside span i + 1 (and further), until it stands on a token start
//            that lane i + 1's recorded walk also visited: from there on lane i + 1's records are the true parse.
// No pass ever re-decodes what pass 1 decoded; the price is the chase, whose length is the self-synchronisation distance
// of DEFLATE (tests/study/deflate_sync.c), and both passes last as long as their slowest lane.  How many SIMT steps per
// 16 128 compressed bits (one window of the r2 kernel: 63 spans of 256 bits, 61.6 counting + 16.8 emitting steps)?
// Not part of the product or of the test suite.
//   gcc -O2 -o /tmp/span_chase tests/study/span_chase.c && for S in 256 512 1024 2048; do /tmp/span_chase /tmp/study/s64k.bin $S; done
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
static int token(const huff_t *lc, const huff_t *dc, uint64_t *pos) { // 0 match, 2 literal, 1 EOB, -1 invalid
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
// one step from q: a leading literal (when it ends before lim) and the token behind it.  Returns 0 went on, 1 EOB, -1 invalid;
// *mid = the token start inside the step (0 when the step had no leading literal)
static int step(const huff_t *lc, const huff_t *dc, uint64_t *q, uint64_t lim, uint64_t *mid) {
    *mid = 0;
    uint64_t p = *q;
    int r = token(lc, dc, &p);
    if (r < 0) return -1;
    if (r == 2 && p < lim) { *mid = p; r = token(lc, dc, &p); if (r < 0) { *q = *mid; return -1; } }
    *q = p;
    return r == 1 ? 1 : 0;
}
#define MAXB (1 << 18)
#define MAXN 256
static uint8_t bmap[MAXN][MAXB / 8]; // token starts of lane i's own walk, relative to W
int main(int argc, char **argv) {
    if (argc < 3) return 2;
    FILE *f = fopen(argv[1], "rb"); int S0 = atoi(argv[2]); int NS = argc > 3 ? atoi(argv[3]) : 64;
    uint32_t n; if (!f || fread(&n, 4, 1, f) != 1) return 1;
    uint64_t n_win = 0, s1 = 0, s2 = 0, s12 = 0, bits = 0, through = 0, lanes = 0, sum1 = 0, sum2 = 0, hist2[8] = {0}, over128 = 0;
    for (uint32_t e = 0; e < n; e++) {
        uint32_t len; if (fread(&len, 4, 1, f) != 1) return 1;
        uint8_t *buf = malloc(len + 8); if (fread(buf, 1, len, f) != len) return 1; memset(buf + len, 0, 8);
        in = buf; in_len = len;
        uint64_t pos = 0, total_bits = 8ull * len; int last = 0;
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
            uint64_t end = p; // the block's end (the kernel does not know it: it sizes its spans by the rest of the STREAM)
            uint64_t W = pos;
            while (W + 320 < end) {
                // span size: what the rest of the stream gives 64 lanes, at most S0, a multiple of 32, at least 128 bits
                uint64_t rest = total_bits - W;
                uint64_t S = (rest / NS + 31) / 32 * 32; if (S > (uint64_t)S0) S = S0; if (S < 128) S = 128;
                int nact = NS;
                if ((uint64_t)NS * S > MAXB) return 3;
                uint64_t ex[MAXN]; int fl[MAXN], ns[MAXN], nc[MAXN];
                memset(bmap, 0, sizeof(bmap));
                for (int i = 0; i < nact; i++) { // pass 1
                    uint64_t q = W + (uint64_t)i * S, lim = W + (uint64_t)(i + 1) * S, mid; int st = 0, r = 0;
                    if (q >= total_bits) { fl[i] = 2; ns[i] = 0; ex[i] = q; continue; }
                    while (q < lim) {
                        uint64_t q0 = q;
                        bmap[i][(q0 - W) >> 3] |= 1 << ((q0 - W) & 7);
                        r = step(&lc, &dc, &q, lim, &mid); st++;
                        if (mid && mid - W < MAXB) bmap[i][(mid - W) >> 3] |= 1 << ((mid - W) & 7);
                        if (r) break;
                    }
                    ex[i] = q; fl[i] = r > 0 ? 1 : (r < 0 ? 2 : 0); ns[i] = st;
                }
                // pass 2: lane i walks on from ex[i] until it stands on a token start of the lane whose span it is in
                int mx1 = 0, mx2 = 0, mx12 = 0;
                int on = 0, stop = 0; uint64_t wend = 0; // the true chain: lane 0, then whoever it synchronises into ...
                for (int i = 0; i < nact; i++) nc[i] = 0;
                int nxt[MAXN];
                for (int i = 0; i < nact; i++) {
                    nxt[i] = -1;
                    if (fl[i]) continue;
                    uint64_t q = ex[i], mid; int st = 0, r = 0;
                    for (;;) {
                        int tl = (int)((q - W) / S);
                        if (tl >= nact) { nxt[i] = MAXN; break; } // walked out of the window: the next window starts here
                        if (bmap[tl][(q - W) >> 3] & (1 << ((q - W) & 7))) { nxt[i] = tl; break; }
                        r = step(&lc, &dc, &q, ~0ull, &mid); st++;
                        if (r) { nxt[i] = r > 0 ? MAXN + 1 : MAXN + 2; break; }
                    }
                    nc[i] = st; ex[i] = q; /* where the chain goes on */
                }
                // follow the chain from lane 0
                int used[MAXN] = {0};
                while (!stop) {
                    used[on] = 1;
                    if (fl[on]) { wend = ex[on]; stop = 1; break; }
                    if (nxt[on] >= MAXN) { wend = ex[on]; stop = 1; break; }
                    if (nxt[on] != on + 1) through++;
                    on = nxt[on];
                }
                for (int i = 0; i < nact; i++) {
                    if (!used[i] && 0) continue; // every lane walks in lock step, used or not
                    if (ns[i] > mx1) mx1 = ns[i];
                    if (nc[i] > mx2) mx2 = nc[i];
                    if (ns[i] + nc[i] > mx12) mx12 = ns[i] + nc[i];
                    sum1 += ns[i]; sum2 += nc[i]; lanes++;
                    int b = nc[i] < 4 ? 0 : nc[i] < 8 ? 1 : nc[i] < 16 ? 2 : nc[i] < 24 ? 3 : nc[i] < 32 ? 4 : nc[i] < 48 ? 5 : nc[i] < 64 ? 6 : 7; if (nc[i] >= 128) over128++;
                    hist2[b]++;
                }
                if (NS != 64) { /* 64 lanes take the spans in order as they get free: the window lasts until the last lane is done */
                    int busy[64] = {0}; int nxtspan = 0, t = 0, left = nact;
                    while (left > 0) { for (int l = 0; l < 64; l++) { if (busy[l] == 0 && nxtspan < nact) { busy[l] = ns[nxtspan] + nc[nxtspan]; nxtspan++; if (busy[l] == 0) left--; } }
                        int any = 0; for (int l = 0; l < 64; l++) if (busy[l] > 0) { any = 1; if (--busy[l] == 0) left--; } if (!any) break; t++; }
                    mx12 = t; }
                s1 += mx1; s2 += mx2; s12 += mx12; n_win++;
                if (wend <= W) { fprintf(stderr, "no progress\n"); return 4; }
                bits += (wend > end ? end : wend) - W;
                W = wend;
                if (wend >= end) break;
            }
            pos = end;
        }
        free(buf);
    }
    double k = 16128.0 / (double)bits;
    printf("S<=%d: %llu windows, %.0f bits each; per 16128 bits: pass 1 %.1f steps, chase %.1f, together %.1f (one merged loop: %.1f); "
           "mean lane: walk %.1f chase %.1f steps; chains that skipped a lane: %llu\n", S0, (unsigned long long)n_win, (double)bits / n_win,
           s1 * k, s2 * k, (s1 + s2) * k, s12 * k, (double)sum1 / lanes, (double)sum2 / lanes, (unsigned long long)through);
    printf("   chase length histogram (<4 <8 <16 <24 <32 <48 <64 more): ");
    for (int b = 0; b < 8; b++) printf("%.1f%% ", 100.0 * hist2[b] / lanes);
    printf(" >=128: %llu of %llu lanes\n", (unsigned long long)over128, (unsigned long long)lanes);
    return 0;
}
