/* tests/fuzz_zip_index.c -- TEST INFRASTRUCTURE (not pytest).  Memory safety of the central-directory walk the prime runs over archives
 * it is handed (minizip-ng_amd/csrc/zip_index.c): mutated and truncated images of a seed archive, each in a heap block of exactly
 * its size, through mzhip_zip_index_mem / mzhip_zip_index_hash_mem under AddressSanitizer + UBSan.
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude tests/fuzz_zip_index.c minizip-ng_amd/csrc/zip_index.c -o /tmp/zipidx_fuzz
 *   /tmp/zipidx_fuzz some.zip [images=100000]
 * Round 5: 900 000 images of three seed archives (40 / 3 / 300 entries, with and without Hash extra fields and a comment): no fault.
 * Round 6: the same images also through the walk from the archive's TAIL alone (mzhip_zip_index_tail / _hash_tail: the tail in a heap
 * block of exactly its size, re-read from where the call says it needs it) and mzhip_zip_index_resolve over random windows of the body
 * (each in a block of exactly its size): no fault, and wherever both walks accept the image they agree row for row -- the payload offsets
 * after the windows have been resolved included. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int64_t mzhip_zip_index_mem(const uint8_t *zip, uint64_t zip_len, int64_t *table, int64_t max_entries);
int64_t mzhip_zip_index_hash_mem(const uint8_t *zip, uint64_t zip_len, const int64_t *table, int64_t n, uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest);
int64_t mzhip_zip_index_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, int64_t *table, int64_t max_entries, uint64_t *need_from);
int64_t mzhip_zip_index_hash_tail(const uint8_t *tail, uint64_t tail_off, uint64_t zip_len, const int64_t *table, int64_t n, uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest);
int64_t mzhip_zip_index_resolve(const uint8_t *win, uint64_t win_off, uint64_t win_len, int64_t *table, int64_t n);
#define NEED_MORE (-1000)
static uint64_t rng = 88172645463325252ull;
static uint32_t r32(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }
int main(int argc, char **argv) {
    long iters = argc > 2 ? atol(argv[2]) : 100000;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *orig = malloc(n); if (fread(orig, 1, n, f) != (size_t)n) return 2; fclose(f);
    long ok = 0, neg = 0, tails = 0;
    for (long it = 0; it < iters; it++) {
        long len = n; if (r32() % 4 == 0) len = r32() % (n + 1);
        /* exact-size heap copy: ASAN sees any read past the end */
        uint8_t *z = malloc(len ? len : 1); memcpy(z, orig + (n - len) * (r32() & 1 && len < n ? 0 : 0), len);
        if (len < n && (r32() & 1)) memcpy(z, orig + (n - len), len); /* keep the tail (EOCD) half of the time */
        int flips = r32() % 6;
        for (int k = 0; k < flips && len; k++) {
            long at = (r32() % 3) ? len - 1 - (long)(r32() % (len < 4096 ? len : 4096)) : (long)(r32() % len);
            if (r32() & 1) z[at] ^= (uint8_t)(1u << (r32() % 8)); else z[at] = (uint8_t)r32();
        }
        int64_t maxe = (r32() % 5 == 0) ? (int64_t)(r32() % 4) : 4096;
        int64_t *table = malloc((size_t)(maxe ? maxe : 1) * 8 * sizeof(int64_t));
        int64_t cnt = mzhip_zip_index_mem(z, (uint64_t)len, table, maxe);
        if (cnt >= 0) {
            ok++;
            int64_t m = cnt < maxe ? cnt : maxe;
            uint16_t *alg = malloc((size_t)(m ? m : 1) * sizeof(uint16_t)); uint8_t *dg = malloc((size_t)(m ? m : 1) * 64);
            { uint16_t *ds = malloc((size_t)(m ? m : 1) * 2); (void)mzhip_zip_index_hash_mem(z, (uint64_t)len, table, m, alg, ds, dg); free(ds); }
            free(alg); free(dg);
        } else neg++;
        /* the same image from its tail alone */
        {
            uint64_t from = len > 300 ? (uint64_t)len - (uint64_t)(r32() % 300) : 0, need = 0;
            int64_t tc = NEED_MORE;
            uint8_t *tail = NULL;
            int64_t *tt = malloc((size_t)(maxe ? maxe : 1) * 8 * sizeof(int64_t));
            for (int pass = 0; pass < 6 && tc == NEED_MORE; pass++) {
                free(tail);
                const uint64_t tl = (uint64_t)len - from;
                tail = malloc(tl ? tl : 1); memcpy(tail, z + from, tl);
                tc = mzhip_zip_index_tail(tail, from, (uint64_t)len, tt, maxe, &need);
                if (tc == NEED_MORE) { if (need >= from) { fprintf(stderr, "tail walk asks for %llu from %llu\n", (unsigned long long)need, (unsigned long long)from); return 1; } from = need; }
            }
            if (tc == NEED_MORE) { fprintf(stderr, "tail walk never settles\n"); return 1; }
            if ((tc >= 0) != (cnt >= 0) || (tc >= 0 && tc != cnt)) { fprintf(stderr, "image %ld: whole walk %lld, tail walk %lld\n", it, (long long)cnt, (long long)tc); return 1; }
            if (tc >= 0) {
                tails++;
                const int64_t m = tc < maxe ? tc : maxe;
                { uint16_t *alg = malloc((size_t)(m ? m : 1) * 2), *ds = malloc((size_t)(m ? m : 1) * 2); uint8_t *dg = malloc((size_t)(m ? m : 1) * 64);
                  (void)mzhip_zip_index_hash_tail(tail, from, (uint64_t)len, tt, m, alg, ds, dg); free(alg); free(ds); free(dg); }
                /* windows of the body, each in a block of its own size; afterwards every row the whole walk resolved is resolved the same */
                for (int w = 0; w < 4 && len; w++) {
                    const uint64_t wo = r32() % (uint64_t)len, wl = 1 + r32() % ((uint64_t)len - wo);
                    uint8_t *win = malloc(wl); memcpy(win, z + wo, wl);
                    (void)mzhip_zip_index_resolve(win, wo, wl, tt, m);
                    free(win);
                }
                { uint8_t *win = malloc(len ? len : 1); memcpy(win, z, len); (void)mzhip_zip_index_resolve(win, 0, (uint64_t)len, tt, m); free(win); }
                for (int64_t i = 0; i < m; i++)
                    for (int c = 0; c < 8; c++)
                        if (tt[8 * i + c] != table[8 * i + c]) { fprintf(stderr, "image %ld row %lld column %d: %lld (whole) / %lld (tail + windows)\n", it, (long long)i, c, (long long)table[8 * i + c], (long long)tt[8 * i + c]); return 1; }
            }
            free(tail); free(tt);
        }
        free(table); free(z);
    }
    printf("zip index fuzz: %ld images, %ld indexed, %ld refused, %ld of them also from the tail + windows with the same rows, no fault\n", iters, ok, neg, tails);
    return 0;
}
