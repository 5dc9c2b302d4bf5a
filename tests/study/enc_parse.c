/* Not part of the product or of the test suite.
 *   python -c "import sys; sys.path.insert(0, '.'); from tests import synth; open('/tmp/corpus.bin','wb').write(bytes(synth.bench_corpus()[0]))"
 *   gcc -O2 -o /tmp/enc_parse tests/study/enc_parse.c -lm && /tmp/enc_parse /tmp/corpus.bin [HB=12] [W=4] [MINM=4] [STEP=64]
 *       [PARSE=0 greedy | 1 lazy (K4 levels 4-6) | 2 cost parse over the block (levels 7-9) | 3 cost-aware lazy heuristics]
 *       [INTCOST=1 integer prices] [ITER=n] [SEQ=1 candidates of the own step visible] [INH=3 matches handed on] [HASH3=1]
 * Round-3 results on 24 x 64 KiB of the bench corpus (entropy-priced, so ~1.5 % below the real streams): shipped r2 design
 * 0.2961; +inheritance 0.2927; cost parse 0.2866 (integer prices, one iteration); 8 ways at 11 hash bits 0.2908 (with cost
 * parse 0.2831); candidates 16 positions at a time 0.2932; zlib's own matcher depth (32 ways, 15 bits) 0.2797; zlib-6: 0.2823.
 *
 * study: where K4's ratio gap to zlib-6 comes from.  Models the token choice of deflate_core.h (hash of 4 bytes, HB bits,
 * W ways, candidates looked up for 64 positions at once BEFORE those positions are inserted, lazy rules) and variants; the
 * cost of a parse = entropy of its symbols (dynamic Huffman, one block per 64 KiB) + extra bits + a 60-byte header. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define MAXN 65536
static int INTCOST = 0, ITER = 2, INH = 0;
static int HB = 12, W = 4, MINM = 4, STEP = 64, PARSE = 1 /* 0 greedy 1 lazy2 2 optimal-in-step */, HASH3 = 0, SEQ = 0;
static uint32_t ld32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static int mlen_of(const uint8_t *a, const uint8_t *b, int maxl) { int l = 0; while (l < maxl && a[l] == b[l]) l++; return l; }
static int lsym(int l) { l -= 3; if (l == 255) return 285; if (l < 8) return 257 + l; int k = 31 - __builtin_clz(l); return 257 + 4 * (k - 2) + 4 + ((l >> (k - 2)) & 3); }
static int lext(int l) { l -= 3; if (l == 255 || l < 8) return 0; return 31 - __builtin_clz(l) - 2; }
static int dsym(int d) { d -= 1; if (d < 4) return d; int k = 31 - __builtin_clz(d); return 2 * k + ((d >> (k - 1)) & 1); }
static int dext(int d) { d -= 1; if (d < 4) return 0; return 31 - __builtin_clz(d) - 1; }
static int bl[MAXN], bd[MAXN]; /* best match per position */
static double clit[286], cdist[30]; /* current cost model (bits) */
static double mcost(int l, int d) { return clit[lsym(l)] + lext(l) + cdist[dsym(d)] + dext(d); }
static int tokl[MAXN], tokd[MAXN], ntok;
static void find_matches(const uint8_t *in, int n) {
    int nb = 1 << HB;
    int32_t *tab = malloc(sizeof(int32_t) * nb * W);
    for (int i = 0; i < nb * W; i++) tab[i] = -1;
    for (int p0 = 0; p0 < n; p0 += STEP) {
        int e = p0 + STEP < n ? p0 + STEP : n;
        for (int pass = 0; pass < 2; pass++) {
            for (int p = p0; p < e; p++) {
                if (p + 4 > n) { if (!pass) { bl[p] = 0; bd[p] = 0; } continue; }
                uint32_t v = ld32(in + p); if (HASH3) v &= 0xFFFFFF;
                uint32_t h = (v * 2654435761u) >> (32 - HB);
                int32_t *b = tab + (size_t)h * W;
                if ((pass == 0) != (SEQ != 0) || (SEQ && pass == 0)) { /* lookup: in pass 0 (batched), or per position when SEQ */
                    if (pass == 0) {
                        int best = 0, bdist = 0, maxl = n - p < 258 ? n - p : 258;
                        for (int w = 0; w < W; w++) { if (b[w] < 0) continue; int d = p - b[w]; if (d < 1 || d > 32506) continue; int l = mlen_of(in + p, in + b[w], maxl); if (l >= MINM && l > best && !(l == 3 && d > 4096)) { best = l; bdist = d; } }
                        bl[p] = best; bd[p] = bdist;
                    }
                }
                if ((SEQ && pass == 0) || (!SEQ && pass == 1)) { for (int w = W - 1; w >= 1; w--) b[w] = b[w - 1]; b[0] = p; }
            }
        }
    }
    free(tab);
    if (INH == 1) for (int p = 1; p < n; p++) if (bl[p - 1] - 1 > bl[p] && bl[p - 1] - 1 >= MINM) { bl[p] = bl[p - 1] - 1; bd[p] = bd[p - 1]; }
    if (INH > 1) { /* within a 64-step, from the lanes 1, 2, 4 below (INH = number of doubling rounds + 1) */
        static int nl[MAXN], nd[MAXN];
        for (int sh = 1, r = 1; r < INH; r++, sh *= 2) {
            for (int p = 0; p < n; p++) { nl[p] = bl[p]; nd[p] = bd[p]; if ((p & 63) >= sh && bl[p - sh] - sh > bl[p] && bl[p - sh] - sh >= MINM) { nl[p] = bl[p - sh] - sh; nd[p] = bd[p - sh]; } }
            memcpy(bl, nl, sizeof(int) * n); memcpy(bd, nd, sizeof(int) * n);
        }
    }
}
static void parse(int n) {
    ntok = 0;
    if (PARSE == 3) { /* lazy + "a match must be cheaper than its bytes as literals" + lazy decided by cost per byte */
        extern const uint8_t *g_in;
        for (int p = 0; p < n;) {
            int l = bl[p];
            if (l) { double lc = 0; for (int k = 0; k < l; k++) lc += clit[g_in[p + k]]; if (mcost(l, bd[p]) >= lc) l = 0; }
            if (l) {
                if (p + 1 < n && bl[p + 1] > l) l = 0; else if (p + 2 < n && bl[p + 2] > l + 1) l = 0;
                else if (p + 1 < n && bl[p + 1] == l && mcost(l, bd[p + 1]) + clit[g_in[p]] + 1e-9 < mcost(l, bd[p]) - 0.0 - clit[g_in[p + l]]) l = 0;
            }
            tokl[ntok] = l; tokd[ntok] = l ? bd[p] : 0; ntok++; p += l ? l : 1;
        }
    } else if (PARSE < 2) {
        for (int p = 0; p < n;) {
            int l = bl[p];
            if (PARSE >= 1 && l) { if (p + 1 < n && bl[p + 1] > l) l = 0; else if (W > 1 && p + 2 < n && bl[p + 2] > l + 1) l = 0; }
            tokl[ntok] = l; tokd[ntok] = l ? bd[p] : 0; ntok++; p += l ? l : 1;
        }
    } else { /* optimal under the cost model, whole block (an upper bound for "optimal in a 64-step") */
        static double cost[MAXN + 1]; static int choice[MAXN + 1];
        cost[n] = 0;
        for (int p = n - 1; p >= 0; p--) {
            double c = 0; /* set below */
            extern const uint8_t *g_in; c = clit[g_in[p]] + cost[p + 1]; int ch = 0;
            for (int l = MINM; l <= bl[p]; l++) { double m = mcost(l, bd[p]) + cost[p + l]; if (m < c) { c = m; ch = l; } }
            cost[p] = c; choice[p] = ch;
        }
        for (int p = 0; p < n;) { int l = choice[p]; tokl[ntok] = l; tokd[ntok] = l ? bd[p] : 0; ntok++; p += l ? l : 1; }
    }
}
const uint8_t *g_in;
static double block_bits(const uint8_t *in, int n, int update_model) {
    double fl[286] = {0}, fd[30] = {0}; double extra = 0; int p = 0;
    for (int i = 0; i < ntok; i++) { if (tokl[i]) { fl[lsym(tokl[i])]++; fd[dsym(tokd[i])]++; extra += lext(tokl[i]) + dext(tokd[i]); p += tokl[i]; } else { fl[in[p]]++; p++; } }
    fl[256] = 1; double tl = 0, td = 0; for (int i = 0; i < 286; i++) tl += fl[i]; for (int i = 0; i < 30; i++) td += fd[i];
    double bits = extra + 60 * 8;
    for (int i = 0; i < 286; i++) if (fl[i]) bits += fl[i] * -log2(fl[i] / tl);
    for (int i = 0; i < 30; i++) if (fd[i]) bits += fd[i] * -log2(fd[i] / td);
    if (update_model) { for (int i = 0; i < 286; i++) clit[i] = fl[i] ? -log2(fl[i] / tl) : 14; for (int i = 0; i < 30; i++) cdist[i] = fd[i] ? -log2(fd[i] / td) : 12;
        if (INTCOST) { for (int i = 0; i < 286; i++) { clit[i] = fl[i] ? floor(clit[i] + 0.5) : 12; if (clit[i] < 1) clit[i] = 1; if (clit[i] > 15) clit[i] = 15; } for (int i = 0; i < 30; i++) { cdist[i] = fd[i] ? floor(cdist[i] + 0.5) : 10; if (cdist[i] < 1) cdist[i] = 1; } } }
    return bits;
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET); uint8_t *c = malloc(sz); fread(c, 1, sz, f);
    for (int i = 2; i < argc; i++) { char *a = argv[i]; if (!strncmp(a, "HB=", 3)) HB = atoi(a + 3); if (!strncmp(a, "W=", 2)) W = atoi(a + 2); if (!strncmp(a, "MINM=", 5)) MINM = atoi(a + 5); if (!strncmp(a, "STEP=", 5)) STEP = atoi(a + 5); if (!strncmp(a, "PARSE=", 6)) PARSE = atoi(a + 6); if (!strncmp(a, "HASH3=", 6)) HASH3 = atoi(a + 6); if (!strncmp(a, "SEQ=", 4)) SEQ = atoi(a + 4); if (!strncmp(a, "INTCOST=", 8)) INTCOST = atoi(a + 8); if (!strncmp(a, "ITER=", 5)) ITER = atoi(a + 5); if (!strncmp(a, "INH=", 4)) INH = atoi(a + 4); }
    double tot = 0; long nin = 0;
    for (long off = 50000; off + 65536 <= sz && nin < 24 * 65536; off += 65536 * 11) {
        const uint8_t *in = c + off; g_in = in; int n = 65536;
        find_matches(in, n);
        int pm = PARSE; PARSE = pm >= 2 ? 1 : pm; parse(n); double b = block_bits(in, n, 1); PARSE = pm;
        if (pm >= 2) for (int it = 0; it < ITER; it++) { parse(n); b = block_bits(in, n, 1); }
        tot += b / 8; nin += n;
    }
    printf("HB=%d W=%d MINM=%d STEP=%d PARSE=%d HASH3=%d SEQ=%d: ratio %.4f\n", HB, W, MINM, STEP, PARSE, HASH3, SEQ, tot / nin);
}
