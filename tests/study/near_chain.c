// Design study for K1's commit (inflate_commit.inc), round 3: the near copies are 24 % of a wave's time -- 15 dependency
// rounds per chunk of ~3.4 KB with ~11 of 64 lanes busy.  What would fewer rounds cost?  Chunks as the chase window cuts
// them (bytes <= CB, pieces of <= 32 bytes <= CP); for each chunk the rounds of
//   A  the shipped rule: near pieces 64 at a time, a piece is copied when no byte of its source is pending;
//   B  all near pieces of the chunk at once (as many slots per lane as it takes): the true dependency depth;
//   C  k passes of SOURCE REWRITING first: a piece whose source lies inside the destination of ONE pending piece takes
//      that piece's source instead (shifted) -- pointer jumping on pieces, no byte moves; then A or B.
// Not part of the product or of the test suite.  Input: u32 count, then per stream u32 length + raw DEFLATE
// (tests/study/deflate_sync.c says how to make one).   gcc -O2 -o /tmp/near_chain tests/study/near_chain.c && /tmp/near_chain /tmp/streams.bin
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

#define CB 3400u
#define CP 350u
typedef struct { uint32_t d, l; int64_t s; } piece_t; /* destination, length, source (all chunk-relative; s < 0: before the chunk) */
static uint64_t n_chunk, n_piece, n_near, n_far, r_A, r_B, r_CA[4], r_CB[4], far_after[4], rewr[4], lanes_A, self_ov;
static int depth_rounds(piece_t *p, int K, int per, uint32_t nbytes, uint64_t *lanes) {
    static uint8_t pend[8192]; static uint8_t done[4096];
    memset(pend, 0, nbytes + 64); memset(done, 0, K);
    int rounds = 0;
    for (int b = 0; b < K; b += per) {
        int e = b + per < K ? b + per : K;
        for (int j = b; j < e; j++) for (uint32_t k = 0; k < p[j].l; k++) pend[p[j].d + k] = 1;
        int left = e - b;
        while (left) {
            int rdy[4096], nr = 0;
            for (int j = b; j < e; j++) {
                if (done[j]) continue;
                int64_t s0 = p[j].s; uint32_t n = p[j].l; int ok = 1;
                if ((int64_t)p[j].d - s0 < (int64_t)n) n = (uint32_t)(p[j].d - s0); /* the part that is not the piece's own output */
                for (uint32_t k = 0; k < n && ok; k++) if (s0 + k >= 0 && pend[s0 + k]) ok = 0;
                if (ok) rdy[nr++] = j;
            }
            if (!nr) { fprintf(stderr, "stuck\n"); exit(1); }
            for (int i = 0; i < nr; i++) { int j = rdy[i]; done[j] = 1; for (uint32_t k = 0; k < p[j].l; k++) pend[p[j].d + k] = 0; }
            left -= nr; rounds++; if (lanes) *lanes += nr;
        }
    }
    return rounds;
}
static void chunk(const tok_t *tk, size_t n) {
    static piece_t all[4096], near[4096], cur[4096], nxt[4096];
    int M = 0; uint32_t pos = 0;
    for (size_t i = 0; i < n; i++) {
        if (tk[i].len == 1) { pos++; continue; }
        for (uint32_t o = 0; o < tk[i].len; o += 32) { all[M].d = pos + o; all[M].l = tk[i].len - o < 32 ? tk[i].len - o : 32; all[M].s = (int64_t)pos + o - tk[i].dist; M++; }
        pos += tk[i].len;
    }
    n_chunk++; n_piece += M;
    /* far part first (as the kernel does): what of a piece's source lies before the chunk is fetched; the rest is near */
    int K = 0;
    for (int j = 0; j < M; j++) {
        uint32_t nf = all[j].s < 0 ? (uint32_t)(-all[j].s < (int64_t)all[j].l ? -all[j].s : all[j].l) : 0;
        if (nf) n_far++;
        if (nf == all[j].l) continue;
        near[K].d = all[j].d + nf; near[K].l = all[j].l - nf; near[K].s = all[j].s + nf; K++;
        if (all[j].d - all[j].s < all[j].l) self_ov++;
    }
    n_near += K;
    r_A += depth_rounds(near, K, 64, pos, &lanes_A);
    r_B += depth_rounds(near, K, 4096, pos, NULL);
    /* rewriting passes over ALL pieces of the chunk (before the far phase): Jacobi -- every piece looks at the previous pass */
    memcpy(cur, all, sizeof(piece_t) * M);
    for (int pass = 0; pass < 4; pass++) {
        for (int j = 0; j < M; j++) {
            nxt[j] = cur[j];
            if (cur[j].s < 0 || cur[j].d - cur[j].s < cur[j].l) continue; /* far already, or feeds on itself */
            /* the pending piece that holds the first source byte (pieces are in stream order: binary search by destination) */
            int lo = 0, hi = j;
            while (lo < hi) { int mid = (lo + hi) / 2; if ((int64_t)cur[mid].d + cur[mid].l <= cur[j].s) lo = mid + 1; else hi = mid; }
            if (lo < j && (int64_t)cur[lo].d <= cur[j].s && cur[j].s + cur[j].l <= (int64_t)cur[lo].d + cur[lo].l && !(cur[lo].d - cur[lo].s < cur[lo].l)) {
                nxt[j].s = cur[lo].s + (cur[j].s - cur[lo].d); rewr[pass]++;
            }
        }
        memcpy(cur, nxt, sizeof(piece_t) * M);
        int K2 = 0;
        for (int j = 0; j < M; j++) {
            uint32_t nf = cur[j].s < 0 ? (uint32_t)(-cur[j].s < (int64_t)cur[j].l ? -cur[j].s : cur[j].l) : 0;
            if (nf) far_after[pass]++;
            if (nf == cur[j].l) continue;
            near[K2].d = cur[j].d + nf; near[K2].l = cur[j].l - nf; near[K2].s = cur[j].s + nf; K2++;
        }
        r_CA[pass] += depth_rounds(near, K2, 64, pos, NULL);
        r_CB[pass] += depth_rounds(near, K2, 4096, pos, NULL);
    }
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); int S = 0;
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
            for (;;) { if (nt == cap) { cap *= 2; tk = realloc(tk, cap * sizeof(tok_t)); } int r = token(&lc, &dc, &pos, &tk[nt]); if (r) { if (r < 0) fprintf(stderr, "bad token in stream %u after %zu tokens\n", e, nt); break; } nt++; }
            size_t ti = 0;
            while (ti < nt) { /* chunks of <= CB bytes and <= CP pieces; only chunks that have 32 KiB of history behind them count as typical */
                size_t tj = ti; uint64_t bytes = 0, pcs = 0;
                while (tj < nt && bytes + tk[tj].len <= CB && pcs + (tk[tj].len > 1 ? (tk[tj].len + 31) / 32 : 0) <= CP) { bytes += tk[tj].len; pcs += tk[tj].len > 1 ? (tk[tj].len + 31) / 32 : 0; tj++; }
                if (tj == ti) tj = ti + 1, bytes = tk[ti].len;
                chunk(tk + ti, tj - ti);
                outpos += bytes; ti = tj;
            }
            free(tk);
        }
        free(buf);
    }
    (void)S;
    printf("chunks %llu: pieces/chunk %.0f, far (whole or part) %.0f, near %.0f (feeding on themselves: %.1f)\n", (unsigned long long)n_chunk, (double)n_piece / n_chunk, (double)n_far / n_chunk, (double)n_near / n_chunk, (double)self_ov / n_chunk);
    printf("A  64 at a time, exact rule:       %.1f rounds per chunk, %.1f pieces per round\n", (double)r_A / n_chunk, (double)lanes_A / r_A);
    printf("B  all near pieces at once:        %.1f rounds per chunk\n", (double)r_B / n_chunk);
    for (int p = 0; p < 4; p++)
        printf("C  %d rewriting pass(es): %.1f pieces rewritten per chunk in this pass, far pieces then %.0f; rounds 64 at a time %.1f, all at once %.1f\n", p + 1,
               (double)rewr[p] / n_chunk, (double)far_after[p] / n_chunk, (double)r_CA[p] / n_chunk, (double)r_CB[p] / n_chunk);
    return 0;
}
