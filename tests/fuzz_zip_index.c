/* tests/fuzz_zip_index.c -- TEST INFRASTRUCTURE (not pytest).  Memory safety of the central-directory walk the prime runs over archives
 * it is handed (minizip-ng_amd/csrc/zip_index.c): mutated and truncated images of a seed archive, each in a heap block of exactly
 * its size, through mzhip_zip_index_mem / mzhip_zip_index_hash_mem under AddressSanitizer + UBSan.
 *   gcc -O1 -g -fsanitize=address,undefined -Iinclude tests/fuzz_zip_index.c minizip-ng_amd/csrc/zip_index.c -o /tmp/zipidx_fuzz
 *   /tmp/zipidx_fuzz some.zip [images=100000]
 * Round 5: 900 000 images of three seed archives (40 / 3 / 300 entries, with and without Hash extra fields and a comment): no fault. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
int64_t mzhip_zip_index_mem(const uint8_t *zip, uint64_t zip_len, int64_t *table, int64_t max_entries);
int64_t mzhip_zip_index_hash_mem(const uint8_t *zip, uint64_t zip_len, const int64_t *table, int64_t n, uint16_t *algorithm, uint16_t *digest_size, uint8_t *digest);
static uint64_t rng = 88172645463325252ull;
static uint32_t r32(void) { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return (uint32_t)(rng >> 16); }
int main(int argc, char **argv) {
    long iters = argc > 2 ? atol(argv[2]) : 100000;
    FILE *f = fopen(argv[1], "rb"); if (!f) return 2;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    uint8_t *orig = malloc(n); if (fread(orig, 1, n, f) != (size_t)n) return 2; fclose(f);
    long ok = 0, neg = 0;
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
        free(table); free(z);
    }
    printf("zip index fuzz: %ld images, %ld indexed, %ld refused, no fault\n", iters, ok, neg);
    return 0;
}
