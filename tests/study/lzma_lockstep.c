/* lzma_lockstep.c -- STUDY (not product, not a test): what would several LZMA streams per wave cost in divergence?
 *
 * K3 gives one stream one wave and computes lane-invariant values in 64 lanes (52 vector instructions per output byte,
 * the vector ports 89 % busy: DESIGN 3 K3).  VERDICT r5 asks for "several streams per instruction": a quarter-wave per
 * stream, four streams per wave.  Four streams in one wave execute in lock step: where they stand in different parts of
 * the decoder, the wave runs every part that any of them needs.  This program decodes real streams with the oracle's
 * decoder (oracle/lzma_model.h, traced), records every binary decision with its KIND (which part of the model the
 * probability belongs to = which piece of decoder code asks for it) and every packet boundary, and replays G streams in lock
 * step under two models of how a compiler / a hand-written kernel would reconverge them:
 *   A  "one decision per step": a generic decode-one-bit loop, every stream takes one decision per step whatever its kind;
 *      the step costs the shared bit arithmetic once plus the transition code of every DISTINCT kind present;
 *   B  "one packet per round" (the shape of the existing kernel: a literal path, a match path, a rep path ...): every stream
 *      decodes one packet per round; the round costs, for every packet type present, the LONGEST decision chain of that type.
 * Output: decisions per byte, and for G = 1, 2, 4 the wave-steps per stream-decision -- 1/G would be perfect sharing, 1 is no
 * gain over one stream per wave.
 *
 *   gcc -O2 -I oracle tests/study/lzma_lockstep.c -o /tmp/lzma_lockstep && /tmp/lzma_lockstep stream0.bin stream1.bin ...
 * (streams = ZIP method-14 payloads: tests/study/lzma_lockstep.py writes them from the bench's config-4 text) */
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

static uint8_t *g_kinds;   /* kind of every decision of the stream being decoded */
static uint32_t *g_pk;     /* index of the first decision of every packet */
static size_t g_nk, g_capk, g_np, g_capp;
static const void *g_model_lo, *g_model_hi; /* set per stream: address range of the model_t */
static void trace_bit(const void *p);
static void trace_direct(int n);
static void trace_packet(void);
#define ORC_TRACE_BIT(p_) trace_bit(p_)
#define ORC_TRACE_DIRECT(n_) trace_direct(n_)
#define ORC_TRACE_PACKET() trace_packet()
#include "lzma_dec.c"

/* kinds: 0 is_match, 1 literal bit, 2 is_rep / g0 / g1 / g2 / rep0_long, 3 length choice, 4 length tree bit, 5 pos slot bit,
 * 6 pos_dec (reverse tree) bit, 7 align bit, 8 direct bit */
enum { K_ISMATCH, K_LIT, K_REP, K_LENCH, K_LENBIT, K_SLOT, K_POSDEC, K_ALIGN, K_DIRECT, K_N };
static const char *kname[K_N] = {"is_match", "literal bit", "rep flags", "len choice", "len tree", "pos slot", "pos_dec", "align", "direct"};
static lz_t *g_z;
static void push_kind(uint8_t k) {
    if (g_nk == g_capk) {
        g_capk = g_capk ? g_capk * 2 : 1 << 20;
        g_kinds = (uint8_t *)realloc(g_kinds, g_capk);
    }
    g_kinds[g_nk++] = k;
}
static void trace_bit(const void *pv) {
    const model_t *m = (const model_t *)g_model_lo;
    const uint16_t *p = (const uint16_t *)pv;
    uint8_t k;
    if (pv < g_model_lo || pv >= g_model_hi) k = K_LIT;
    else if (p < &m->is_rep[0]) k = K_ISMATCH;
    else if (p < &m->pos_slot[0][0]) k = K_REP;
    else if (p < &m->pos_dec[0]) k = K_SLOT;
    else if (p < &m->align[0]) k = K_POSDEC;
    else if (p < (const uint16_t *)&m->len) k = K_ALIGN;
    else {
        const len_t *l = p < (const uint16_t *)&m->rep_len ? &m->len : &m->rep_len;
        k = (p == &l->choice || p == &l->choice2) ? K_LENCH : K_LENBIT;
    }
    push_kind(k);
}
static void trace_direct(int n) {
    for (int i = 0; i < n; i++) push_kind(K_DIRECT);
}
static void trace_packet(void) {
    if (g_np == g_capp) {
        g_capp = g_capp ? g_capp * 2 : 1 << 18;
        g_pk = (uint32_t *)realloc(g_pk, g_capp * sizeof(uint32_t));
    }
    g_pk[g_np++] = (uint32_t)g_nk;
}

typedef struct {
    uint8_t *kinds;
    uint32_t *pk;
    size_t nk, np, out_len;
} stream_t;

/* a private copy of orc_lzma_zip_decode's set-up is not needed: the model lives inside the lz_t that function allocates; its
 * address range is found through a second hook -- the first traced pointer inside a calloc'ed lz_t is is_match, and model_t is
 * the struct's second member.  Simpler: decode with a local re-implementation of the 20 lines of set-up. */
static int decode_traced(const uint8_t *in, size_t n, uint8_t *out, size_t cap, size_t *out_len) {
    if (n < 14 || in[4] >= 225) return -1;
    lz_t *z = (lz_t *)calloc(1, sizeof(lz_t));
    unsigned d = in[4];
    z->lc = d % 9; d /= 9; z->lp = d % 5; z->pb = d / 5;
    uint64_t dict = in[5] | ((uint32_t)in[6] << 8) | ((uint32_t)in[7] << 16) | ((uint32_t)in[8] << 24);
    if (dict < 4096) dict = 4096;
    z->dict = (dict + 15) & ~(uint64_t)15;
    z->out = out; z->out_cap = cap;
    z->lit = (uint16_t *)malloc(((size_t)0x300 << (z->lc + z->lp)) * sizeof(uint16_t));
    lz_reset_state(z);
    z->rc.in = in; z->rc.in_len = n; z->rc.in_pos = 9; z->rc.range = 0xFFFFFFFFu;
    for (int i = 0; i < 5; i++) z->rc.code = (z->rc.code << 8) | rc_byte(&z->rc);
    g_model_lo = &z->m; g_model_hi = (const uint8_t *)&z->m + sizeof(model_t);
    g_z = z;
    const int32_t r = lz_run(z, (size_t)-1, 0);
    *out_len = z->opos;
    free(z->lit); free(z);
    return r;
}

int main(int argc, char **argv) {
    int ns = argc - 1;
    if (ns < 4) { fprintf(stderr, "usage: %s stream.bin x (4 or more)\n", argv[0]); return 2; }
    stream_t *S = (stream_t *)calloc((size_t)ns, sizeof(stream_t));
    uint8_t *out = (uint8_t *)malloc(64u << 20);
    double tot_dec = 0, tot_bytes = 0, tot_pk = 0;
    double hist[K_N] = {0};
    for (int s = 0; s < ns; s++) {
        FILE *f = fopen(argv[1 + s], "rb");
        if (!f) { perror(argv[1 + s]); return 1; }
        fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
        uint8_t *in = (uint8_t *)malloc((size_t)n);
        if (fread(in, 1, (size_t)n, f) != (size_t)n) return 1;
        fclose(f);
        g_kinds = NULL; g_pk = NULL; g_nk = g_capk = g_np = g_capp = 0;
        size_t ol = 0;
        const int r = decode_traced(in, (size_t)n, out, 64u << 20, &ol);
        if (r != 0) { fprintf(stderr, "%s: decode returned %d\n", argv[1 + s], r); return 1; }
        trace_packet(); /* sentinel: the end of the last packet */
        S[s].kinds = g_kinds; S[s].pk = g_pk; S[s].nk = g_nk; S[s].np = g_np - 1; S[s].out_len = ol;
        tot_dec += (double)g_nk; tot_bytes += (double)ol; tot_pk += (double)(g_np - 1);
        for (size_t i = 0; i < g_nk; i++) hist[g_kinds[i]] += 1;
        free(in);
    }
    printf("%d streams, %.0f bytes out: %.3f decisions per byte, %.2f decisions per packet, %.2f bytes per packet\n", ns, tot_bytes,
           tot_dec / tot_bytes, tot_dec / tot_pk, tot_bytes / tot_pk);
    for (int k = 0; k < K_N; k++) printf("  %-12s %5.1f %% of the decisions\n", kname[k], 100.0 * hist[k] / tot_dec);
    /* transition cost of a kind relative to the shared bit arithmetic (= 1.0): rough instruction counts of the device code's
     * paths between two decisions (~11 vector instructions a decision, of which ~7 the bit arithmetic) */
    const double trans[K_N] = {0.6, 0.4, 0.5, 0.4, 0.3, 0.3, 0.4, 0.4, 0.3};
    for (int G = 1; G <= 4; G *= 2) {
        double stepsA = 0, costA = 0, decA = 0, roundsB = 0, costB = 0, decB = 0;
        for (int g0 = 0; g0 + G <= ns; g0 += G) {
            /* model A */
            size_t maxn = 0;
            for (int j = 0; j < G; j++) if (S[g0 + j].nk > maxn) maxn = S[g0 + j].nk;
            size_t minn = maxn;
            for (int j = 0; j < G; j++) if (S[g0 + j].nk < minn) minn = S[g0 + j].nk;
            for (size_t t = 0; t < minn; t++) { /* (while all G streams are alive) */
                unsigned present = 0;
                for (int j = 0; j < G; j++) present |= 1u << S[g0 + j].kinds[t];
                double c = 1.0;
                for (int k = 0; k < K_N; k++) if (present >> k & 1u) c += trans[k];
                costA += c; stepsA += 1; decA += G;
            }
            /* model B: one packet per stream and round; a packet's type = literal / match / rep (by its second decision) */
            size_t minp = S[g0].np;
            for (int j = 1; j < G; j++) if (S[g0 + j].np < minp) minp = S[g0 + j].np;
            for (size_t r = 0; r < minp; r++) {
                uint32_t longest[3] = {0, 0, 0};
                for (int j = 0; j < G; j++) {
                    const stream_t *s = &S[g0 + j];
                    const uint32_t a = s->pk[r], b = s->pk[r + 1], n = b - a;
                    const int type = (n >= 2 && s->kinds[a + 1] == K_LIT) ? 0 : (n >= 2 && s->kinds[a + 1] == K_REP && n >= 3 && s->kinds[a + 2] == K_REP) ? 2 : 1;
                    /* (decision 1 = is_match; a match packet's decision 2 is is_rep (kind REP) = 0 and then a len choice;
                     *  a rep packet's is is_rep = 1 and then is_rep_g0 (kind REP again)) */
                    if (n > longest[type]) longest[type] = n;
                    decB += n;
                }
                costB += longest[0] + longest[1] + longest[2];
                roundsB += 1;
            }
        }
        const double perA1 = 1.0 + (trans[0] * hist[0] + trans[1] * hist[1] + trans[2] * hist[2] + trans[3] * hist[3] + trans[4] * hist[4] + trans[5] * hist[5] +
                                    trans[6] * hist[6] + trans[7] * hist[7] + trans[8] * hist[8]) / tot_dec;
        printf("G = %d streams per wave:  model A  %.3f wave-steps per stream-decision (cost %.3f of a one-stream decision; perfect sharing %.3f)"
               "   model B  %.3f wave-decisions per stream-decision (perfect %.3f)\n",
               G, stepsA / decA, costA / decA / perA1, 1.0 / G, costB / decB, 1.0 / G);
    }
    return 0;
}
