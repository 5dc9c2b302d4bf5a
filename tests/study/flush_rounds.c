// Design study behind K1's window flush: with the literals and the far back-references of a whole span
// window (64 lanes x S bits) already in the LDS staging area, how many dependency rounds do the remaining (near)
// back-references of a window need when they are copied 64 at a time, one per lane, under the frontier rule
// "a match is ready when its source ends at or before the destination of the first unfinished match"?
// Not part of the product or of the test suite.  Input: see deflate_sync.c (u32 count, then u32 length + raw DEFLATE).
//   gcc -O2 -o /tmp/flush_rounds tests/study/flush_rounds.c && /tmp/flush_rounds /tmp/streams.bin 256
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
static const uint16_t k_lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
static const uint8_t k_lext[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint16_t k_dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
static const uint8_t k_dext[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
typedef struct { uint64_t bit; uint32_t len, dist; } tok_t; // len 1 = literal
static int token(const huff_t *lc, const huff_t *dc, uint64_t *pos, tok_t *t) {
    t->bit = *pos;
    int sym = decode(lc, pos);
    if (sym < 0) return -1;
    if (sym < 256) { t->len = 1; t->dist = 0; return 0; }
    if (sym == 256) return 1;
    sym -= 257; if (sym >= 29) return -1;
    t->len = k_lbase[sym] + bits_at(*pos, k_lext[sym]); *pos += k_lext[sym];
    int ds = decode(dc, pos);
    if (ds < 0 || ds >= 30) return -1;
    t->dist = k_dbase[ds] + bits_at(*pos, k_dext[ds]); *pos += k_dext[ds];
    return 0;
}
static uint64_t n_win, n_tok, n_lit, n_match, n_far, n_pfar, n_near, n_batch, n_rounds, n_roundlanes, sum_len, n_len_gt32, n_len_gt64, n_dist_lt8, n_selfov;
static uint64_t it8_sum, it8_far_sum, win_bytes, win_bytes_max, near_batches_compact, rounds_compact, it8_compact;
static uint64_t pool_rounds, pool_it8, pool_done;
static uint64_t rounds_hist[65], ex_rounds[2], ex_batches[2], ex_it8[2], ex_hist[2][65];
static int cmpu(const void *a, const void *b) { return 0; }
// simulate one window: toks[0..n), window output starts at absolute output offset `base`
static void window(const tok_t *tk, size_t n, uint64_t base) {
    static uint32_t dst[8192], len[8192], dist[8192];
    size_t M = 0; uint32_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        if (tk[i].len == 1) { n_lit++; pos++; continue; }
        dst[M] = pos; len[M] = tk[i].len; dist[M] = tk[i].dist; M++;
        pos += tk[i].len; sum_len += tk[i].len; n_match++;
        if (tk[i].len > 32) n_len_gt32++;
        if (tk[i].len > 64) n_len_gt64++;
        if (tk[i].dist < 8) n_dist_lt8++;
        if (tk[i].dist < tk[i].len) n_selfov++;
    }
    n_win++; n_tok += n; win_bytes += pos; if (pos > win_bytes_max) win_bytes_max = pos;
    (void)base;
    // all matches of the window, 64 per batch
    for (int compact = 0; compact < 2; compact++) {
        static uint32_t cd[8192], cl[8192], cs[8192]; size_t K = 0;
        // list: compact=0 -> every match (far ones have rem 0), compact=1 -> only matches with a near remainder
        for (size_t j = 0; j < M; j++) {
            uint32_t nf = dist[j] > dst[j] ? (dist[j] - dst[j] < len[j] ? dist[j] - dst[j] : len[j]) : 0;
            if (!compact) { if (nf == len[j]) n_far++; else if (nf) n_pfar++; else n_near++; }
            uint32_t rem = len[j] - nf;
            if (compact && rem == 0) continue;
            cd[K] = dst[j] + nf; cl[K] = rem; cs[K] = dist[j]; K++;
        }
        for (size_t b = 0; b < K; b += 64) {
            size_t e = b + 64 < K ? b + 64 : K;
            uint64_t U = 0;
            for (size_t j = b; j < e; j++) if (cl[j]) U |= 1ull << (j - b);
            if (!U) continue;
            int rounds = 0;
            if (compact) near_batches_compact++; else n_batch++;
            while (U) {
                int first = __builtin_ctzll(U);
                uint32_t F = cd[b + first];
                uint64_t ready = 0; uint32_t maxit = 0;
                for (size_t j = b; j < e; j++) {
                    if (!((U >> (j - b)) & 1)) continue;
                    uint32_t se = cd[j] - cs[j] + cl[j]; if (se > cd[j]) se = cd[j];
                    if (se <= F) { ready |= 1ull << (j - b); uint32_t it = (cl[j] + 7) / 8; if (it > maxit) maxit = it; }
                }
                U &= ~ready; rounds++;
                if (compact) { it8_compact += maxit; } else { it8_sum += maxit; n_roundlanes += __builtin_popcountll(ready); }
            }
            if (compact) rounds_compact += rounds; else { n_rounds += rounds; rounds_hist[rounds > 64 ? 64 : rounds]++; }
        }
    }
    // exact rule: a match is ready when no byte of its source (the part before its own destination) is still pending
    for (int compact = 0; compact < 2; compact++) {
        static uint8_t pend[1 << 16];
        static uint32_t cd[8192], cl[8192], cs[8192]; size_t K = 0;
        memset(pend, 0, pos + 8);
        for (size_t j = 0; j < M; j++) {
            uint32_t nf = dist[j] > dst[j] ? (dist[j] - dst[j] < len[j] ? dist[j] - dst[j] : len[j]) : 0;
            uint32_t rem = len[j] - nf;
            if (compact && rem == 0) continue;
            cd[K] = dst[j] + nf; cl[K] = rem; cs[K] = dist[j]; K++;
            for (uint32_t k = 0; k < rem; k++) pend[dst[j] + nf + k] = 1;
        }
        for (size_t b = 0; b < K; b += 64) {
            size_t e = b + 64 < K ? b + 64 : K;
            uint64_t U = 0;
            for (size_t j = b; j < e; j++) if (cl[j]) U |= 1ull << (j - b);
            if (!U) continue;
            int rounds = 0; ex_batches[compact]++;
            while (U) {
                uint64_t ready = 0; uint32_t maxit = 0;
                for (size_t j = b; j < e; j++) {
                    if (!((U >> (j - b)) & 1)) continue;
                    uint32_t s0 = cd[j] - cs[j], n = cl[j] < cs[j] ? cl[j] : cs[j]; int ok = 1;
                    for (uint32_t k = 0; k < n; k++) if (pend[s0 + k]) { ok = 0; break; }
                    if (ok) { ready |= 1ull << (j - b); uint32_t it = (cl[j] + 7) / 8; if (it > maxit) maxit = it; }
                }
                if (!ready) { fprintf(stderr, "stuck\n"); exit(1); }
                for (size_t j = b; j < e; j++) if ((ready >> (j - b)) & 1) for (uint32_t k = 0; k < cl[j]; k++) pend[cd[j] + k] = 0;
                U &= ~ready; rounds++; ex_it8[compact] += maxit;
            }
            ex_rounds[compact] += rounds; ex_hist[compact][rounds > 64 ? 64 : rounds]++;
        }
    }
    // rolling pool of 64 near matches: every round the lanes whose match finished take the next ones of the list
    {
        static uint8_t pend[1 << 16];
        static uint32_t cd[8192], cl[8192], cs[8192]; size_t K = 0;
        memset(pend, 0, pos + 8);
        for (size_t j = 0; j < M; j++) {
            uint32_t nf = dist[j] > dst[j] ? (dist[j] - dst[j] < len[j] ? dist[j] - dst[j] : len[j]) : 0;
            uint32_t rem = len[j] - nf;
            if (rem == 0) continue;
            cd[K] = dst[j] + nf; cl[K] = rem; cs[K] = dist[j]; K++;
        }
        int slot[64]; for (int i = 0; i < 64; i++) slot[i] = -1;
        size_t next = 0; int live = 0;
        for (;;) {
            for (int i = 0; i < 64 && next < K; i++) if (slot[i] < 0) { slot[i] = (int)next; for (uint32_t k = 0; k < cl[next]; k++) pend[cd[next] + k] = 1; next++; live++; }
            if (!live) break;
            int rdy[64]; uint32_t maxit = 0;
            for (int i = 0; i < 64; i++) {
                rdy[i] = 0; int j = slot[i]; if (j < 0) continue;
                uint32_t s0 = cd[j] - cs[j], n = cl[j] < cs[j] ? cl[j] : cs[j]; int ok = 1;
                for (uint32_t k = 0; k < n; k++) if (pend[s0 + k]) { ok = 0; break; }
                if (ok) { rdy[i] = 1; uint32_t it = (cl[j] + 7) / 8; if (it > maxit) maxit = it; }
            }
            for (int i = 0; i < 64; i++) if (rdy[i]) { int j = slot[i]; for (uint32_t k = 0; k < cl[j]; k++) pend[cd[j] + k] = 0; slot[i] = -1; live--; pool_done++; }
            pool_rounds++; pool_it8 += maxit;
        }
    }
    (void)cmpu;
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int S = atoi(argv[2]);
    uint32_t n; if (fread(&n, 4, 1, f) != 1) return 1;
    for (uint32_t e = 0; e < n; e++) {
        uint32_t len; if (fread(&len, 4, 1, f) != 1) return 1;
        uint8_t *buf = malloc(len + 8); if (fread(buf, 1, len, f) != len) return 1; memset(buf + len, 0, 8);
        in = buf; in_len = len;
        uint64_t pos = 0, outpos = 0; int last = 0;
        while (!last) {
            last = bits_at(pos, 1); int type = bits_at(pos + 1, 2); pos += 3;
            huff_t lc, dc;
            if (type == 0) { pos = (pos + 7) & ~7ull; uint32_t l = bits_at(pos, 16); pos += 32 + 8ull * l; outpos += l; continue; }
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
            size_t cap = 1 << 16, nt = 0; tok_t *tk = malloc(cap * sizeof(tok_t));
            for (;;) { if (nt == cap) { cap *= 2; tk = realloc(tk, cap * sizeof(tok_t)); } int r = token(&lc, &dc, &pos, &tk[nt]); if (r) break; nt++; }
            // windows of 64 * S bits
            size_t ti = 0;
            while (ti < nt) {
                uint64_t W = tk[ti].bit, nextW = W + 64ull * S; size_t tj = ti;
                uint64_t bytes = 0;
                while (tj < nt && tk[tj].bit < nextW) { bytes += tk[tj].len; tj++; }
                window(tk + ti, tj - ti, outpos);
                outpos += bytes; ti = tj;
            }
            free(tk);
        }
        free(buf);
    }
    printf("S=%d: windows=%llu tokens/window=%.0f bytes/window=%.0f (max %llu) literals=%.1f%% of tokens, match len avg=%.1f, len>32: %.2f%%, len>64: %.2f%%, dist<8: %.2f%%, dist<len: %.2f%%\n",
           S, (unsigned long long)n_win, (double)n_tok / n_win, (double)win_bytes / n_win, (unsigned long long)win_bytes_max, 100.0 * n_lit / n_tok, (double)sum_len / n_match,
           100.0 * n_len_gt32 / n_match, 100.0 * n_len_gt64 / n_match, 100.0 * n_dist_lt8 / n_match, 100.0 * n_selfov / n_match);
    printf("matches/window=%.0f: far %.1f%%, partly far %.1f%%, near %.1f%%\n", (double)n_match / n_win, 100.0 * n_far / n_match, 100.0 * n_pfar / n_match, 100.0 * n_near / n_match);
    printf("all-match batches: %.2f per window, rounds/batch=%.2f, lanes/round=%.1f, 8-byte iterations (sum of per-round max)/batch=%.1f\n",
           (double)n_batch / n_win, (double)n_rounds / n_batch, (double)n_roundlanes / n_rounds, (double)it8_sum / n_batch);
    printf("near-only batches: %.2f per window, rounds/batch=%.2f, 8-byte iterations/batch=%.1f\n", (double)near_batches_compact / n_win,
           (double)rounds_compact / near_batches_compact, (double)it8_compact / near_batches_compact);
    for (int c = 0; c < 2; c++) {
        printf("exact rule, %s batches: %.2f per window, rounds/batch=%.2f, 8-byte iterations/batch=%.1f, hist:", c ? "near-only" : "all-match", (double)ex_batches[c] / n_win,
               (double)ex_rounds[c] / ex_batches[c], (double)ex_it8[c] / ex_batches[c]);
        for (int r = 1; r <= 64; r++) if (ex_hist[c][r]) printf(" %d:%llu", r, (unsigned long long)ex_hist[c][r]);
        printf("\n");
    }
    printf("rolling pool (near matches): rounds/window=%.1f, matches/round=%.1f, 8-byte iterations/round=%.2f\n", (double)pool_rounds / n_win, (double)pool_done / pool_rounds, (double)pool_it8 / pool_rounds);
    printf("rounds histogram (all-match batches):"); for (int r = 1; r <= 64; r++) if (rounds_hist[r]) printf(" %d:%llu", r, (unsigned long long)rounds_hist[r]); printf("\n");
    return 0;
}
