// mzhip_kernels.hip -- gfx950 kernels + host runtime of libmzhip.so (the batch C ABI of
// include/mzhip.h).  The per-entry algorithms live in inflate_core.h / crc32_core.h; this file
// owns launch geometry, work distribution and the device context.
//
// Launch shape of K1 (MI355X: 256 CUs x 4 SIMDs, 160 KiB LDS/CU, 8 XCDs):
//   - one wavefront per ZIP entry, 4 wavefronts per workgroup, 9.8 KiB LDS per wave (3.8 KiB of Huffman tables, the
//     2.8 KiB span window, a 3.2 KiB pool in which a chunk of a window is resolved: staging bytes, pending bits,
//     back-reference list; the LZ77 window beyond the chunk is the output buffer itself), so 4 workgroups = 16 waves
//     fit per CU by LDS; the kernel is compiled for 4 waves per SIMD (<= 128 VGPRs);
//   - persistent waves: the grid is sized to the chip (CUs x resident workgroups) and every wave pulls
//     its next entry index from one device-scope counter, so short and long entries balance and
//     a 100k-entry batch is a single launch with no host involvement.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>
#include <sched.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/mzhip.h"
#include "inflate_core.h"
#include "lzma_core.h"
#include "deflate_core.h"
#include "adler32_core.h"
#include "xz_core.h"
#include "lzma_enc_core.h"

#ifndef MZ_WAVES_PER_WG
#define MZ_WAVES_PER_WG 4
#endif
#if defined(MZ_PROF)
/* measurement builds (make PROF=1): cycles per section of K1, summed over all waves since the last read */
__device__ unsigned long long mz_prof_buf[32];
extern "C" __attribute__((visibility("default"))) int mzhip_prof_read(unsigned long long *out32, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(mz_prof_buf), sizeof(mz_prof_buf)) != hipSuccess) return -1;
    if (reset) {
        static const unsigned long long zero[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(mz_prof_buf), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
#define MZ_CRC_TAB_BYTES 1024
#define MZ_LDS_STRIDE ((sizeof(mz_inflate_lds) + 15) & ~(size_t)15)
#define MZ_NUM_COUNTERS 64
#ifndef MZ_MIN_WAVES_PER_SIMD
#define MZ_MIN_WAVES_PER_SIMD 4 /* register budget: 128 VGPRs -> 4 waves per SIMD, 16 per CU (matches the LDS budget: 4 workgroups) */
#endif

struct InflateArgs {
    const uint8_t *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    uint32_t n;
    uint32_t *out_len;
    uint32_t *in_used;
    uint32_t *crc;
    int32_t *status;
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
    uint8_t *rec; // MZ_REC_BYTES of step-record scratch per wave of the grid (chase window), or null
    const mz_inflate_state *resume; // per entry: take the stream up at this state (window-by-window decode), or null
    mz_inflate_state *stop;         // per entry: where it can be taken up again, or null
};

__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64, MZ_MIN_WAVES_PER_SIMD) void k_inflate_batch(InflateArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crc_tab = (uint32_t *)smem;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const int wave = threadIdx.x >> 6;
    mz_inflate_lds *L = (mz_inflate_lds *)(smem + MZ_CRC_TAB_BYTES + wave * MZ_LDS_STRIDE);
    uint8_t *const rec = a.rec ? a.rec + ((size_t)blockIdx.x * MZ_WAVES_PER_WG + (size_t)wave) * MZ_REC_BYTES : nullptr;
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        mz_inflate_result r;
        // entry descriptors are wave-uniform: pin them to SGPRs so addresses use the scalar base
        const uint64_t io = a.in_off[e], oo = a.out_off[e];
        const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
        uint8_t *out = a.out + (((uint64_t)MZ_UNIFORM((uint32_t)(oo >> 32)) << 32) | MZ_UNIFORM((uint32_t)oo));
        mz_inflate_entry(in, MZ_UNIFORM(a.in_len[e]), out, MZ_UNIFORM(a.out_cap[e]), L, crc_tab, a.tabs, 1u, rec,
                         a.resume ? a.resume + e : nullptr, a.stop ? a.stop + e : nullptr, &r);
        // wave-uniform results: stored by all lanes (same address, same value), see MZ_WAVE_FETCH_ADD
        a.out_len[e] = r.out_len;
        a.in_used[e] = r.in_used;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}

struct CrcArgs {
    const uint8_t *buf;
    const uint64_t *off;
    const uint32_t *len;
    uint32_t n;
    const uint32_t *init;
    uint32_t *crc;
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
};

// K2 stand-alone: one wave per buffer, same tile folding as the fused epilogue.
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_crc32_batch(CrcArgs a) {
    __shared__ uint32_t crc_tab4[1024]; // slicing-by-4 tables; the first 256 entries are the byte table
    __shared__ uint32_t crc_mul4[1024]; // the super-tile advance as four lookups
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) {
        crc_tab4[i] = a.tabs->slice[i >> 8][i & 255];
        crc_mul4[i] = a.tabs->mul4[i >> 8][i & 255];
    }
    __syncthreads();
    const uint32_t *crc_tab = crc_tab4;
    MZ_LANE_DECL
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        const uint8_t *buf = a.buf + a.off[e];
        const uint32_t n = a.len[e];
        const uint32_t init = a.init ? a.init[e] : 0u;
        uint32_t acc = (lane == 0) ? ~init : 0u; // register = ~value on entry (mz_crypt.c:81)
        uint32_t tmp, done = 0, result, reg = ~init;
        const mzhip_crc_tables *tabs = a.tabs;
        // 4 KiB super-tiles first (64 contiguous bytes per lane), the remainder with the 1 KiB tiles of the fused epilogues
        MZ_CRC_FOLD_SUPER(acc, done, buf, n, crc_tab4, crc_mul4);
        if (done) {
            MZ_CRC_SUPER_REDUCE(reg, acc, tmp, tabs);
            acc = (lane == 0) ? reg : 0u;
        }
        {
            const uint8_t *rest = buf + done;
            const uint32_t nrest = n - done;
            uint32_t rdone = 0;
            MZ_CRC_FOLD_TILES(acc, rdone, rest, nrest, crc_tab, tabs->kx);
            MZ_CRC_FINISH_FROM(result, acc, tmp, rdone, rest, nrest, crc_tab, tabs, reg);
        }
        a.crc[e] = result; // uniform store
    }
}

// Adler-32 of n buffers (zlib-wrapper trailer): one wave per buffer, persistent waves.
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_adler32_batch(CrcArgs a) {
    MZ_LANE_DECL
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        const uint8_t *buf = a.buf + a.off[e];
        const uint32_t n = MZ_UNIFORM(a.len[e]);
        uint32_t r;
        MZ_ADLER32(r, buf, n);
        a.crc[e] = r; // uniform store
    }
}

struct LzmaArgs {
    const uint8_t *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    const int64_t *max_out; // may be null: no TOTAL_OUT_MAX clamp
    uint32_t n;
    uint32_t *out_len;
    uint32_t *in_used;
    uint32_t *crc;
    int32_t *status;
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
    const uint64_t *tab64; // CRC-64 byte table (.xz kernel only)
    uint16_t *xprobs;      // MZ_LZMA_XPROBS u16 per resident wave: upper half of the literal model when lc + lp = 4
    uint16_t *sprobs;      // slot kernel: MZ_LZMA_SPROBS u16 per resident wave, the whole literal model
    uint32_t *retry_list;  // slot kernel: entries given back (MZHIP_RETRY) are appended here, ...
    uint32_t *retry_n;     // ... counted here; the full-model kernel then decodes exactly those (list != null: n = *retry_n)
};

#ifndef MZ_LZMA_VPORT_OF_8
#define MZ_LZMA_VPORT_OF_8 8u
#endif

// K3: one wave per workgroup, the wave's whole probability model (15.6 KiB) in LDS -> 10 waves per CU.
#ifdef MZ_LZMA_WAVES /* measurement builds (profiles/ab_k3.sh): waves per SIMD the register allocation is held to */
__global__ __launch_bounds__(64, MZ_LZMA_WAVES) void k_lzma_batch(LzmaArgs a) {
#else
__global__ __launch_bounds__(64) void k_lzma_batch(LzmaArgs a) {
#endif
    __shared__ __attribute__((aligned(16))) mz_lzma_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const uint32_t n_here = a.retry_list ? MZ_UNIFORM(*a.retry_n) : a.n; // second launch: the entries the slot kernel gave back
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= n_here) break;
        if (a.retry_list) e = MZ_UNIFORM(a.retry_list[e]);
        mz_lzma_result r;
        // how many of every 8 workgroups run the vector-port build of the decoder (measured: 0/8 520 ms, 5/8 511 ms,
        // 8/8 469 ms for one full round of 2304 resident 1 MiB entries)
        if ((blockIdx.x & 7u) < MZ_LZMA_VPORT_OF_8)
            mz_lzma_entry_v(a.in + a.in_off[e], a.in_len[e], a.out + a.out_off[e], a.out_cap[e],
                            a.max_out ? a.max_out[e] : (int64_t)-1, &lds, crc_tab, a.tabs,
                            a.xprobs + (size_t)blockIdx.x * MZ_LZMA_XPROBS, &r);
        else
            mz_lzma_entry(a.in + a.in_off[e], a.in_len[e], a.out + a.out_off[e], a.out_cap[e],
                          a.max_out ? a.max_out[e] : (int64_t)-1, &lds, crc_tab, a.tabs,
                          a.xprobs + (size_t)blockIdx.x * MZ_LZMA_XPROBS, &r);
        // wave-uniform results: stored by all lanes (same address, same value), see MZ_WAVE_FETCH_ADD
        a.out_len[e] = r.out_len;
        a.in_used[e] = r.in_used;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}

// K3, resumable: ONE stream taken up where the call before left it (the window mode of the drop-in READ stream,
// shim_lzma.c).  The coder state travels in a 64-byte record, the adaptive model in global memory beside it.
struct LzmaResumeArgs {
    const uint8_t *in;
    uint32_t in_len;
    uint8_t *buf; // [dictionary so far | room for this window]
    uint32_t buf_cap;
    const mz_lzma_state *rs;
    mz_lzma_state *st; // null: decode to the end of what there is, no state kept
    uint16_t *model;   // MZ_LZMA_MODEL_U16 probabilities, in and out
    uint32_t *out_len;
    uint32_t *in_used;
    int32_t *status;
    const mzhip_crc_tables *tabs;
};
__global__ __launch_bounds__(64) void k_lzma_resume(LzmaResumeArgs a) {
    __shared__ __attribute__((aligned(16))) mz_lzma_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    mz_lzma_result r;
    mz_lzma_entry_r(a.in, a.in_len, a.buf, a.buf_cap, (int64_t)-1, &lds, crc_tab, a.tabs, a.model, a.rs, a.st, &r);
    *a.out_len = r.out_len; // wave-uniform results: stored by all lanes
    *a.in_used = r.in_used;
    *a.status = r.status;
}

// K3, main kernel: the slot build of the decoder (lzma_core.h LZ_LITERAL_SITE_SLOT) -- 9.6 KiB of LDS per wave, 16 waves
// per CU (registers held to 128 by the launch bounds).  Streams whose literal contexts do not fit the slots are given
// back through retry_list and decoded by k_lzma_batch right behind this launch.
__global__ __launch_bounds__(256, 4) void k_lzma_slot_batch(LzmaArgs a) {
    // four waves per workgroup, each with its own model slice, one CRC table between them: 4 x 9840 + 1024 bytes = four
    // workgroups per CU (single-wave workgroups with a table each come to 15 waves)
    __shared__ __attribute__((aligned(16))) mz_lzma_lds_s lds4[4];
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const uint32_t wave = threadIdx.x >> 6;
    mz_lzma_lds_s &lds = lds4[wave];
    const size_t wave_id = (size_t)blockIdx.x * 4u + wave;
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        mz_lzma_result r;
        mz_lzma_entry_s(a.in + a.in_off[e], a.in_len[e], a.out + a.out_off[e], a.out_cap[e],
                        a.max_out ? a.max_out[e] : (int64_t)-1, &lds, crc_tab, a.tabs,
                        a.sprobs + wave_id * MZ_LZMA_SPROBS, &r);
        if (r.status == MZHIP_RETRY) {
            uint32_t slot;
            MZ_WAVE_FETCH_ADD(slot, a.retry_n);
            a.retry_list[slot] = e; // (uniform store)
        }
        // wave-uniform results: stored by all lanes (same address, same value), see MZ_WAVE_FETCH_ADD
        a.out_len[e] = r.out_len;
        a.in_used[e] = r.in_used;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}

// .xz (method 95): one wave per workgroup like K3; LDS = probability model + CRC-64 table (17.6 KiB) -> 8 per CU.
__global__ __launch_bounds__(64, 2) void k_xz_batch(LzmaArgs a) {
    __shared__ __attribute__((aligned(16))) mz_xz_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        crc_tab[i] = a.tabs->byte_tab[i];
        lds.crc64_tab[i] = a.tab64[i];
    }
    __syncthreads();
    MZ_LANE_DECL
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        mz_lzma_result r;
        mz_xz_entry(a.in + a.in_off[e], a.in_len[e], a.out + a.out_off[e], a.out_cap[e],
                    a.max_out ? a.max_out[e] : (int64_t)-1, &lds, crc_tab, a.tabs,
                    a.xprobs + (size_t)blockIdx.x * MZ_LZMA_XPROBS, &r);
        a.out_len[e] = r.out_len; // wave-uniform results: stored by all lanes
        a.in_used[e] = r.in_used;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}

struct ShaArgs {
    const uint8_t *buf;
    const uint64_t *off;
    const uint32_t *len;
    uint32_t n;
    uint32_t algorithm; // MZ_HASH_SHA1 20, SHA224 22, SHA256 23, SHA384 24, SHA512 25 (mz.h:127-135)
    uint8_t *digest;    // n x 32 bytes (n x 64 for SHA-384 / SHA-512), standard byte order, unused tail bytes zero
};

// SHA-1 / SHA-224 / SHA-256 of n buffers: a digest chain is serial, so ONE LANE hashes one buffer (64 per wave).
// One instantiation per algorithm: the unrolled rounds of both families in one kernel cost 177 VGPRs.
template <int ALG>
__global__ __launch_bounds__(256, 4) void k_sha_batch(ShaArgs a) {
    const uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= a.n) return;
    const uint8_t *p = a.buf + a.off[e];
    const uint64_t n = a.len[e];
    if (ALG == 24 || ALG == 25) {
        uint64_t g[8];
        mz_sha512_init(g, ALG == 24);
        mz_sha512_run(p, n, g);
        uint64_t *d8 = (uint64_t *)(a.digest + (size_t)e * 64);
        for (uint32_t i = 0; i < 8; i++) d8[i] = i < (ALG == 24 ? 6u : 8u) ? __builtin_bswap64(g[i]) : 0ull;
        return;
    }
    uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t words;
    if (ALG == 20) {
        mz_sha1_run(p, n, h);
        words = 5;
    } else {
        mz_sha256_init(h, ALG == 22);
        mz_sha256_run(p, n, h);
        words = ALG == 22 ? 7 : 8;
    }
    uint32_t *d = (uint32_t *)(a.digest + (size_t)e * 32);
    for (uint32_t i = 0; i < 8; i++) d[i] = i < words ? __builtin_bswap32(h[i]) : 0u;
}

struct DeflateArgs {
    const uint8_t *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    const uint8_t *final_flag; // may be null: every piece is a complete stream
    uint32_t n;
    uint32_t *out_len;
    uint32_t *crc;
    int32_t *status;
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
    uint32_t *tok; // token scratch: MZ_DEF_BLOCK words per resident wave
    uint32_t ways; // hash-bucket depth of the match finder: 1 (levels 1-3) or MZ_DEF_WAYS_BEST (levels 4-9, -1)
    uint32_t parse; // 1: cost parse over every block (levels 7-9), 0: the lazy rule per step (levels 4-6 and -1; always with ways == 1)
    uint32_t max_dist; // largest match distance: window - 262
};

#define MZ_DEF_LDS_STRIDE ((sizeof(mz_deflate_lds) + 15) & ~(size_t)15)

// K4: one wave per piece, 4 waves per workgroup, 9.3 KiB LDS per wave (hash heads, reused for code construction, code
// table and bit staging once pass 1 is over; histograms) and 256 KiB of token scratch in HBM per resident wave.  The
// default compression class adds (MZ_DEF_WAYS_BEST - 1) x 8 KiB of older bucket entries per wave behind those.
#define MZ_DEF_XHEAD_BYTES ((MZ_DEF_WAYS_BEST - 1u) * (sizeof(uint16_t) << MZ_DEF_HBITS))
template <uint32_t kParse>
__device__ __forceinline__ void deflate_batch_body(const DeflateArgs &a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crc_tab = (uint32_t *)smem;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const int wave = threadIdx.x >> 6;
    mz_deflate_lds *L = (mz_deflate_lds *)(smem + MZ_CRC_TAB_BYTES + wave * MZ_DEF_LDS_STRIDE);
    uint16_t *xhead = a.ways > 1u ? (uint16_t *)(smem + MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * MZ_DEF_LDS_STRIDE + wave * MZ_DEF_XHEAD_BYTES)
                                  : nullptr;
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        const uint64_t io = a.in_off[e], oo = a.out_off[e];
        const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
        uint8_t *out = a.out + (((uint64_t)MZ_UNIFORM((uint32_t)(oo >> 32)) << 32) | MZ_UNIFORM((uint32_t)oo));
        const uint32_t fin = a.final_flag ? MZ_UNIFORM((uint32_t)a.final_flag[e]) : 1u;
        mz_deflate_result r;
        mz_deflate_piece<kParse>(in, MZ_UNIFORM(a.in_len[e]), out, MZ_UNIFORM(a.out_cap[e]), fin,
                         a.tok + (size_t)(blockIdx.x * MZ_WAVES_PER_WG + wave) * MZ_DEF_BLOCK, L, crc_tab, a.tabs,
                         MZ_UNIFORM(a.ways), xhead, MZ_UNIFORM(a.max_dist), &r);
        a.out_len[e] = r.out_len;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_deflate_batch(DeflateArgs a) { deflate_batch_body<0u>(a); }
// levels 7-9: the same piece loop with the cost parse compiled in (160 VGPRs; this class runs one workgroup per CU anyway)
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_deflate_cost_batch(DeflateArgs a) { deflate_batch_body<1u>(a); }

struct LzmaEncArgs {
    const uint8_t *in;
    const uint64_t *in_off;
    const uint32_t *in_len;
    uint8_t *out;
    const uint64_t *out_off;
    const uint32_t *out_cap;
    const uint8_t *mode; // may be null (all 0): 0 = ZIP method-14 payload, 1 = raw LZMA2 chunk payload
    uint32_t n;
    uint32_t maxb; // 64 KiB blocks reserved per entry in tok / ntok
    uint32_t ways; // hash candidates per position in the LZ77 parse (1 or MZ_DEF_WAYS_BEST), by preset
    uint32_t *tok;
    uint32_t *ntok;
    uint32_t *out_len;
    uint32_t *crc;
    int32_t *status;
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
};

// LZMA encode, pass 1: the LZ77 parse, one wave per 64 KiB block of any entry (work item = entry * maxb + block).
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_lz_tokenize_batch(LzmaEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; /* one head table per wave, then (ways - 1) more each */
    MZ_LANE_DECL
    const uint32_t wave = threadIdx.x >> 6;
    mz_lz_tok_lds *L = (mz_lz_tok_lds *)smem + wave;
    uint16_t *xhead = a.ways > 1u ? (uint16_t *)(smem + MZ_WAVES_PER_WG * sizeof(mz_lz_tok_lds) + wave * MZ_DEF_XHEAD_BYTES) : (uint16_t *)nullptr;
    const uint32_t items = a.n * a.maxb;
    for (;;) {
        uint32_t w;
        MZ_WAVE_FETCH_ADD(w, a.counter);
        if (w >= items) break;
        const uint32_t e = w / a.maxb, b = w - e * a.maxb;
        const uint32_t len = MZ_UNIFORM(a.in_len[e]);
        const uint32_t lo = b * MZ_DEF_BLOCK;
        uint32_t nt = 0;
        if (lo < len) {
            const uint64_t io = a.in_off[e];
            const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
            nt = mz_lz_tokenize(in, lo, (len - lo < MZ_DEF_BLOCK) ? len : lo + MZ_DEF_BLOCK, a.tok + (size_t)w * MZ_DEF_BLOCK, L,
                                MZ_UNIFORM(a.ways), xhead);
        }
        a.ntok[w] = nt; // uniform store
    }
}

// LZMA encode, pass 2: the adaptive range coder, one wave per stream (model in LDS like K3).
__global__ __launch_bounds__(64) void k_lzma_rc_encode_batch(LzmaEncArgs a) {
    __shared__ __attribute__((aligned(16))) mz_lzma_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter + 1);
        if (e >= a.n) break;
        const uint64_t io = a.in_off[e], oo = a.out_off[e];
        const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
        uint8_t *out = a.out + (((uint64_t)MZ_UNIFORM((uint32_t)(oo >> 32)) << 32) | MZ_UNIFORM((uint32_t)oo));
        mz_lzma_enc_result r;
        mz_lzma_rc_encode(in, MZ_UNIFORM(a.in_len[e]), a.tok + (size_t)e * a.maxb * MZ_DEF_BLOCK, a.ntok + (size_t)e * a.maxb,
                          a.mode ? MZ_UNIFORM((uint32_t)a.mode[e]) : 0u, out, MZ_UNIFORM(a.out_cap[e]), &lds, crc_tab, a.tabs, &r);
        a.out_len[e] = r.out_len;
        a.crc[e] = r.crc;
        a.status[e] = r.status;
    }
}

// LZMA encode, pass 2 of ONE stream that is written segment by segment (the drop-in WRITE stream, shim_lzma.c): the
// coder goes on from / is left in a 64-byte state, the adaptive model travels in global memory beside it.
struct LzmaEncResumeArgs {
    LzmaEncArgs a;       // entry 0 of a: [history blocks | the segment], its token blocks
    uint32_t skip_blocks;
    const mz_lzma_enc_state *rs;
    mz_lzma_enc_state *st; // null: the last segment -- end marker and flush
    uint16_t *model;
};
__global__ __launch_bounds__(64) void k_lzma_rc_encode_resume(LzmaEncResumeArgs r) {
    __shared__ __attribute__((aligned(16))) mz_lzma_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = r.a.tabs->byte_tab[i];
    __syncthreads();
    const LzmaEncArgs &a = r.a;
    mz_lzma_enc_result res;
    mz_lzma_rc_encode_x(a.in + a.in_off[0], a.in_len[0], a.tok, a.ntok, 0u, a.out + a.out_off[0], a.out_cap[0], &lds, crc_tab, a.tabs,
                        &res, r.skip_blocks, r.rs, r.st, r.model);
    a.out_len[0] = res.out_len; // wave-uniform results: stored by all lanes
    a.crc[0] = res.crc;
    a.status[0] = res.status;
}

// ---------------------------------------------------------------------------------- host

namespace {

struct DeviceCtx {
    bool ready = false;
    mzhip_crc_tables *d_tabs = nullptr;
    uint64_t *d_tab64 = nullptr;
    struct ScratchEnt {
        void *p = nullptr;
        size_t cap = 0;
        hipEvent_t ev = nullptr;    // recorded behind the last launch that used the buffer
        hipStream_t last = nullptr; // ... on this stream
        bool used = false, held = false;
    } scratch[32]; // per-launch scratch (K4 / K6 tokens) and the host-buffer calls' staging, see scratch_acquire()
    uint32_t *d_counters = nullptr; // MZ_NUM_COUNTERS work-queue heads, see CounterLease
    struct CounterSlot {
        hipEvent_t ev = nullptr;    // recorded behind the last launch that used the counter
        hipStream_t last = nullptr; // ... on this stream
        bool used = false, held = false;
    } cslots[MZ_NUM_COUNTERS];
    uint32_t next_counter = 0;
    int cu_count = 0;
    int inflate_wgs_per_cu = 1;
    // the four-candidate classes of K4 / the LZ tokenizer need > 64 KiB of dynamic LDS per workgroup: asked for once per
    // device; 0 = not asked yet, 1 = granted, -1 = refused (the one-candidate class is used instead)
    std::atomic<int> big_lds_deflate{0}, big_lds_deflate_cost{0}, big_lds_tok{0};
};

// may this device run `kernel` with `bytes` of dynamic LDS per workgroup?  (asked once per device and kernel)
bool big_lds_ok(std::atomic<int> &state, const void *kernel, size_t bytes) {
    int st = state.load(std::memory_order_acquire);
    if (st == 0) {
        const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
        if (!ok) (void)hipGetLastError(); /* (a launch that is refused all the same turns the state to -1, too) */
        st = ok ? 1 : -1;
        state.store(st, std::memory_order_release);
    }
    return st > 0;
}

constexpr int kMaxDevices = 16;
DeviceCtx g_ctx[kMaxDevices];
std::mutex g_mu;
thread_local char g_err[256] = "";

int32_t fail(const char *what, hipError_t e) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -104; /* MZ_INTERNAL_ERROR */
}

#define HIP_TRY(expr)                          \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) return fail(#expr, _e); \
    } while (0)

int32_t ctx_for_current(DeviceCtx **out) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) {
        snprintf(g_err, sizeof(g_err), "device index %d out of range", dev);
        return -102;
    }
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx &c = g_ctx[dev];
    if (!c.ready) {
        mzhip_crc_tables h;
        mzhip_crc_tables_init(&h);
        HIP_TRY(hipMalloc((void **)&c.d_tabs, sizeof(h)));
        HIP_TRY(hipMemcpy(c.d_tabs, &h, sizeof(h), hipMemcpyHostToDevice));
        uint64_t t64[256];
        mzhip_crc64_table_init(t64);
        HIP_TRY(hipMalloc((void **)&c.d_tab64, sizeof(t64)));
        HIP_TRY(hipMemcpy(c.d_tab64, t64, sizeof(t64), hipMemcpyHostToDevice));
        HIP_TRY(hipMalloc((void **)&c.d_counters, 2 * MZ_NUM_COUNTERS * sizeof(uint32_t)));
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        c.cu_count = prop.multiProcessorCount;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_inflate_batch, MZ_WAVES_PER_WG * 64,
                                                         MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * MZ_LDS_STRIDE) == hipSuccess &&
            nb > 0)
            c.inflate_wgs_per_cu = nb;
        c.ready = true;
    }
    *out = &c;
    return 0;
}

// Per-launch device scratch, stream-ordered without the runtime's memory pools.  hipMallocAsync/hipFreeAsync was the
// first implementation; on this stack (ROCm 7.2, gfx950) the second allocation of a process intermittently came back
// with the kernels' and copies' early writes wiped (the whole block read as zero afterwards: 16 of 100 fresh processes,
// profiles/r1/side_measurements.log), so buffers are plain hipMalloc memory cached here.  A buffer is handed out again
// when the next launch is on the stream that used it last (stream order protects it) or when the event recorded behind
// its last use has completed; otherwise another buffer is allocated, so concurrent streams never share scratch.
// Identity of a stream for the "same stream => stream order protects the buffer" shortcut of the two caches below.
// hipStreamPerThread is ONE handle value that names a different stream in every host thread: two threads must not
// take each other for the same stream (they did for a day: a second thread's launch reused the work-queue head and the
// record scratch of a kernel that was still running -- CRC errors in the two-thread drop-in test).
hipStream_t stream_key(hipStream_t s) {
    static thread_local char tls_marker;
    return s == hipStreamPerThread ? (hipStream_t)(void *)&tls_marker : s;
}

int32_t scratch_acquire(DeviceCtx *c, size_t bytes, hipStream_t s, int *slot, void **p) {
    std::lock_guard<std::mutex> lk(g_mu);
    constexpr int kSlots = (int)(sizeof(c->scratch) / sizeof(c->scratch[0]));
    int best = -1, empty = -1, victim = -1;
    for (int i = 0; i < kSlots; i++) {
        DeviceCtx::ScratchEnt &e = c->scratch[i];
        if (!e.p) {
            if (empty < 0) empty = i;
            continue;
        }
        if (e.held) continue;
        const bool done = !e.used || hipEventQuery(e.ev) == hipSuccess;
        if (e.cap >= bytes && (done || e.last == stream_key(s))) {
            if (best < 0 || e.cap < c->scratch[best].cap) best = i;
        } else if (done && (victim < 0 || e.cap < c->scratch[victim].cap)) {
            victim = i; // idle and too small
        }
    }
    if (best < 0) {
        int i = empty >= 0 ? empty : victim;
        if (i < 0) { /* every buffer is busy on another stream: wait for one */
            for (int k = 0; k < kSlots && i < 0; k++)
                if (!c->scratch[k].held) i = k;
            if (i < 0) {
                snprintf(g_err, sizeof(g_err), "scratch: more than %d concurrent launches", kSlots);
                return -104;
            }
            HIP_TRY(hipEventSynchronize(c->scratch[i].ev));
        }
        DeviceCtx::ScratchEnt &e = c->scratch[i];
        if (e.p) {
            HIP_TRY(hipFree(e.p));
            e.p = nullptr;
            e.cap = 0;
        }
        const size_t cap = (bytes + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        HIP_TRY(hipMalloc(&e.p, cap));
        e.cap = cap;
        e.used = false;
        if (!e.ev) HIP_TRY(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));
        best = i;
    }
    c->scratch[best].held = true;
    *slot = best;
    *p = c->scratch[best].p;
    return 0;
}

int32_t scratch_release(DeviceCtx *c, int slot, hipStream_t s) {
    std::lock_guard<std::mutex> lk(g_mu);
    DeviceCtx::ScratchEnt &e = c->scratch[slot];
    e.held = false;
    e.used = true;
    e.last = stream_key(s);
    HIP_TRY(hipEventRecord(e.ev, s));
    return 0;
}

// The work-queue head of one launch.  The batch entry points are asynchronous on caller-supplied streams, so a counter
// may only be handed out again when the launch that used it last is known to be over: the next launch is on the same
// stream (stream order puts its memset behind that kernel) or the event recorded behind it has completed.  Otherwise
// another slot is taken; with every slot busy on other streams the oldest one is waited for.
struct CounterLease {
    DeviceCtx *c = nullptr;
    hipStream_t s = nullptr;
    int idx = -1;
    uint32_t *p = nullptr;
    int32_t get(DeviceCtx *ctx, hipStream_t st) {
        c = ctx;
        s = st;
        hipEvent_t wait_for = nullptr;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            for (uint32_t k = 0; k < MZ_NUM_COUNTERS && idx < 0; k++) {
                const uint32_t i = (c->next_counter + k) % MZ_NUM_COUNTERS;
                DeviceCtx::CounterSlot &e = c->cslots[i];
                if (e.held) continue;
                if (!e.used || e.last == stream_key(s) || hipEventQuery(e.ev) == hipSuccess) idx = (int)i;
            }
            if (idx < 0) { /* every counter is busy on another stream: take one that is not being set up right now and wait for its launch */
                for (uint32_t k = 0; k < MZ_NUM_COUNTERS && idx < 0; k++) {
                    const uint32_t i = (c->next_counter + k) % MZ_NUM_COUNTERS;
                    if (!c->cslots[i].held) idx = (int)i;
                }
                if (idx < 0) {
                    snprintf(g_err, sizeof(g_err), "more than %d launches being set up at once", MZ_NUM_COUNTERS);
                    return -104;
                }
                wait_for = c->cslots[idx].ev;
            }
            c->cslots[idx].held = true; /* ours from here on: the wait below happens outside the lock */
            c->next_counter = (uint32_t)idx + 1u;
        }
        if (wait_for) HIP_TRY(hipEventSynchronize(wait_for));
        p = c->d_counters + 2 * idx; /* two words per slot: the LZMA encoder's two kernels each have a head */
        HIP_TRY(hipMemsetAsync(p, 0, 2 * sizeof(uint32_t), s));
        return 0;
    }
    ~CounterLease() {
        if (idx < 0) return;
        std::lock_guard<std::mutex> lk(g_mu);
        DeviceCtx::CounterSlot &e = c->cslots[idx];
        if (!e.ev && hipEventCreateWithFlags(&e.ev, hipEventDisableTiming) != hipSuccess) e.ev = nullptr;
        if (e.ev) (void)hipEventRecord(e.ev, s);
        e.used = e.ev != nullptr;
        e.last = stream_key(s);
        e.held = false;
    }
};

uint32_t grid_for(const DeviceCtx *c, uint32_t n) {
    uint32_t wgs_needed = (n + MZ_WAVES_PER_WG - 1) / MZ_WAVES_PER_WG;
    uint32_t resident = (uint32_t)(c->cu_count * c->inflate_wgs_per_cu); /* persistent waves: fill the chip once */
    if (wgs_needed < 1) wgs_needed = 1;
    return wgs_needed < resident ? wgs_needed : resident;
}

} // namespace

extern "C" {

const char *mzhip_last_error(void) { return g_err; }
const char *mzhip_version(void) { return "mzhip 0.1 (gfx950)"; }

int32_t mzhip_device_count(void) {
    static std::atomic<int> known{0}; /* every open() of a codec stream asks: one runtime call per process is enough */
    if (known.load(std::memory_order_relaxed) > 0) return known.load(std::memory_order_relaxed);
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e == hipSuccess && n > 0) known.store(n);
    if (e != hipSuccess) {
        fail("hipGetDeviceCount", e);
        return -1;
    }
    return n;
}

/* The CPUs next to a device (the NUMA node its PCIe root hangs off), from sysfs: page-locked memory that the device
 * copies from and to, and the host threads that read it, belong there -- on a two-socket box the far socket costs a
 * third of the link rate and half of the readers' memcpy rate (profiles/r3/threads_numa.log). */
int32_t mzhip_device_local_cpus(int32_t device, char *cpulist, int32_t cap) {
    if (!cpulist || cap < 2) return -102;
    cpulist[0] = 0;
    char bdf[64] = "";
    HIP_TRY(hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), device));
    for (char *p = bdf; *p; p++)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    char path[160];
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf);
    FILE *f = fopen(path, "r");
    int node = -1;
    if (f) {
        if (fscanf(f, "%d", &node) != 1) node = -1;
        fclose(f);
    }
    if (node < 0) return 0; /* one node, or the platform does not say */
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/local_cpulist", bdf);
    f = fopen(path, "r");
    if (!f) return 0;
    if (!fgets(cpulist, cap, f)) cpulist[0] = 0;
    fclose(f);
    size_t n = strlen(cpulist);
    while (n && (cpulist[n - 1] == '\n' || cpulist[n - 1] == ' ')) cpulist[--n] = 0;
    return (int32_t)n;
}

int32_t mzhip_bind_thread_near_device(int32_t device, int32_t max_cpus) {
    char list[1024];
    const int32_t n = mzhip_device_local_cpus(device, list, (int32_t)sizeof(list));
    if (n <= 0) return n;
    cpu_set_t have, want;
    CPU_ZERO(&want);
    if (sched_getaffinity(0, sizeof(have), &have) != 0) return 0;
    int taken = 0;
    for (const char *p = list; *p;) { /* "64-127,192-255": the cores first, their second hardware threads behind them */
        char *e = nullptr;
        long a = strtol(p, &e, 10), b = a;
        if (e == p) break;
        if (*e == '-') b = strtol(e + 1, &e, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET((int)c, &have) && (max_cpus <= 0 || taken < max_cpus)) {
                CPU_SET((int)c, &want);
                taken++;
            }
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',' ) break;
    }
    if (!taken) return 0; /* the thread may not run on any of them: left where it is */
    if (sched_setaffinity(0, sizeof(want), &want) != 0) return 0;
    return taken;
}

int32_t mzhip_init(int32_t device) {
    HIP_TRY(hipSetDevice(device));
    DeviceCtx *c = nullptr;
    return ctx_for_current(&c);
}

void mzhip_inflate_launch_geometry(uint32_t n, uint32_t *grid, uint32_t *waves_per_wg, uint32_t *lds_bytes) {
    DeviceCtx *c = nullptr;
    uint32_t g = 0;
    if (ctx_for_current(&c) == 0) g = grid_for(c, n);
    if (grid) *grid = g;
    if (waves_per_wg) *waves_per_wg = MZ_WAVES_PER_WG;
    if (lds_bytes) *lds_bytes = (uint32_t)(MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * MZ_LDS_STRIDE);
}

int32_t mzhip_inflate_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                            const uint64_t *d_out_off, const uint32_t *d_out_cap, uint32_t n, uint32_t *d_out_len,
                            uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status, void *stream) {
    return mzhip_inflate_resume_batch(d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, n, d_out_len, d_in_used, d_crc,
                                      d_status, nullptr, nullptr, stream);
}

int32_t mzhip_inflate_resume_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                   const uint64_t *d_out_off, const uint32_t *d_out_cap, uint32_t n, uint32_t *d_out_len,
                                   uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status, const mzhip_inflate_state *d_resume,
                                   mzhip_inflate_state *d_stop, void *stream) {
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    InflateArgs a;
    a.in = (const uint8_t *)d_in;
    a.in_off = d_in_off;
    a.in_len = d_in_len;
    a.out = (uint8_t *)d_out;
    a.out_off = d_out_off;
    a.out_cap = d_out_cap;
    a.n = n;
    a.out_len = d_out_len;
    a.in_used = d_in_used;
    a.crc = d_crc;
    a.status = d_status;
    CounterLease lease;
    rc = lease.get(c, s);
    if (rc) return rc;
    a.counter = lease.p;
    a.tabs = c->d_tabs;
    const size_t lds = MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * MZ_LDS_STRIDE;
    const uint32_t grid = grid_for(c, n);
    a.rec = nullptr;
    a.resume = (const mz_inflate_state *)d_resume;
    a.stop = (mz_inflate_state *)d_stop;
    int slot = -1;
#if MZ_SPAN_DW && MZ_WINDOW_CHASE
    { /* the step records of the chase window: MZ_REC_BYTES per wave of the launch (176 KiB x 4096 resident waves at most) */
        void *scratch = nullptr;
        rc = scratch_acquire(c, (size_t)grid * MZ_WAVES_PER_WG * MZ_REC_BYTES + 1024, s, &slot, &scratch);
        if (rc) return rc;
        a.rec = (uint8_t *)scratch;
    }
#endif
    hipLaunchKernelGGL(k_inflate_batch, dim3(grid), dim3(MZ_WAVES_PER_WG * 64), lds, s, a);
    const hipError_t le = hipGetLastError();
    if (slot >= 0) {
        const int32_t rr = scratch_release(c, slot, s);
        if (rr && le == hipSuccess) return rr;
    }
    if (le != hipSuccess) return fail("k_inflate_batch", le);
    return 0;
}

int32_t mzhip_crc32_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n,
                          const uint32_t *d_init, uint32_t *d_crc, void *stream) {
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CrcArgs a;
    a.buf = (const uint8_t *)d_buf;
    a.off = d_off;
    a.len = d_len;
    a.n = n;
    a.init = d_init;
    a.crc = d_crc;
    CounterLease lease;
    rc = lease.get(c, s);
    if (rc) return rc;
    a.counter = lease.p;
    a.tabs = c->d_tabs;
    hipLaunchKernelGGL(k_crc32_batch, dim3(grid_for(c, n)), dim3(MZ_WAVES_PER_WG * 64), 0, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t mzhip_adler32_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n,
                            uint32_t *d_adler, void *stream) {
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    CrcArgs a;
    a.buf = (const uint8_t *)d_buf;
    a.off = d_off;
    a.len = d_len;
    a.n = n;
    a.init = nullptr;
    a.crc = d_adler;
    CounterLease lease;
    rc = lease.get(c, s);
    if (rc) return rc;
    a.counter = lease.p;
    a.tabs = c->d_tabs;
    hipLaunchKernelGGL(k_adler32_batch, dim3(grid_for(c, n)), dim3(MZ_WAVES_PER_WG * 64), 0, s, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

static int32_t lzma_family_batch(int xz, const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                 const uint64_t *d_out_off, const uint32_t *d_out_cap, const int64_t *d_max_out, uint32_t n,
                                 uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status, void *stream) {
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    LzmaArgs a;
    a.in = (const uint8_t *)d_in;
    a.in_off = d_in_off;
    a.in_len = d_in_len;
    a.out = (uint8_t *)d_out;
    a.out_off = d_out_off;
    a.out_cap = d_out_cap;
    a.max_out = d_max_out;
    a.n = n;
    a.out_len = d_out_len;
    a.in_used = d_in_used;
    a.crc = d_crc;
    a.status = d_status;
    CounterLease lease;
    rc = lease.get(c, s);
    if (rc) return rc;
    a.counter = lease.p;
    a.tabs = c->d_tabs;
    a.tab64 = c->d_tab64;
    /* 16 KiB LDS per wave -> 10 single-wave workgroups per CU (K3); 17.6 KiB -> 8 (.xz) */
    uint32_t resident = (uint32_t)c->cu_count * (xz ? 8u : 10u);
#ifdef MZ_LZMA_RESIDENT /* measurement builds: single-wave workgroups per CU that really fit (LDS and registers) */
    if (!xz) resident = (uint32_t)c->cu_count * MZ_LZMA_RESIDENT;
#endif
    uint32_t grid = n < resident ? n : resident;
    int slot = -1;
    void *scratch = nullptr; /* 12 KiB per resident wave: the literal model's upper half for streams with lc + lp = 4 */
    a.sprobs = nullptr;
    a.retry_list = nullptr;
    a.retry_n = nullptr;
#if defined(MZ_LZMA_NO_SLOT_KERNEL)
    const bool two_step = false;
#else
    const bool two_step = !xz;
#endif
    if (two_step) {
        /* K3: the slot kernel over every entry (16 waves per CU), then the full-model kernel over the entries it gave back */
        const uint32_t res_s = (uint32_t)c->cu_count * 4u; /* workgroups of four waves */
        const uint32_t grid_s = (n + 3u) / 4u < res_s ? (n + 3u) / 4u : res_s;
        const size_t sp_bytes = (size_t)grid_s * 4u * MZ_LZMA_SPROBS * sizeof(uint16_t);
        const size_t xp_bytes = (size_t)grid * MZ_LZMA_XPROBS * sizeof(uint16_t);
        rc = scratch_acquire(c, sp_bytes + xp_bytes + (size_t)n * 4 + 256, s, &slot, &scratch);
        if (rc) return rc;
        a.sprobs = (uint16_t *)scratch;
        a.xprobs = (uint16_t *)((uint8_t *)scratch + sp_bytes);
        uint32_t *list = (uint32_t *)((uint8_t *)scratch + sp_bytes + xp_bytes);
        a.retry_list = list + 64;
        a.retry_n = list; /* one word, zeroed with the launch */
        hipError_t he = hipMemsetAsync(list, 0, 256, s);
        if (he == hipSuccess) {
            hipLaunchKernelGGL(k_lzma_slot_batch, dim3(grid_s), dim3(256), 0, s, a);
            he = hipGetLastError();
        }
        if (he == hipSuccess) {
            a.counter = lease.p + 1; /* the second head of the lease */
            hipLaunchKernelGGL(k_lzma_batch, dim3(grid), dim3(64), 0, s, a);
            he = hipGetLastError();
        }
        rc = scratch_release(c, slot, s);
        if (he != hipSuccess) return fail("k_lzma_slot_batch", he);
        return rc;
    }
    rc = scratch_acquire(c, (size_t)grid * MZ_LZMA_XPROBS * sizeof(uint16_t), s, &slot, &scratch);
    if (rc) return rc;
    a.xprobs = (uint16_t *)scratch;
    if (xz)
        hipLaunchKernelGGL(k_xz_batch, dim3(grid), dim3(64), 0, s, a);
    else
        hipLaunchKernelGGL(k_lzma_batch, dim3(grid), dim3(64), 0, s, a);
    const hipError_t le = hipGetLastError();
    rc = scratch_release(c, slot, s);
    if (le != hipSuccess) return fail("k_lzma_batch", le);
    return rc;
}

int32_t mzhip_lzma_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                         const uint64_t *d_out_off, const uint32_t *d_out_cap, const int64_t *d_max_out, uint32_t n,
                         uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status, void *stream) {
    return lzma_family_batch(0, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_max_out, n, d_out_len, d_in_used,
                             d_crc, d_status, stream);
}

int32_t mzhip_xz_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                       const uint64_t *d_out_off, const uint32_t *d_out_cap, const int64_t *d_max_out, uint32_t n,
                       uint32_t *d_out_len, uint32_t *d_in_used, uint32_t *d_crc, int32_t *d_status, void *stream) {
    return lzma_family_batch(1, d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_max_out, n, d_out_len, d_in_used,
                             d_crc, d_status, stream);
}

int32_t mzhip_sha_batch(const void *d_buf, const uint64_t *d_off, const uint32_t *d_len, uint32_t n, uint32_t algorithm,
                        void *d_digest, void *stream) {
    if (algorithm != 20 && (algorithm < 22 || algorithm > 25)) return MZHIP_STATUS_UNSUPPORTED;
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    ShaArgs a;
    a.buf = (const uint8_t *)d_buf;
    a.off = d_off;
    a.len = d_len;
    a.n = n;
    a.algorithm = algorithm;
    a.digest = (uint8_t *)d_digest;
    if (algorithm == 20)
        hipLaunchKernelGGL(k_sha_batch<20>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    else if (algorithm == 24)
        hipLaunchKernelGGL(k_sha_batch<24>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    else if (algorithm == 25)
        hipLaunchKernelGGL(k_sha_batch<25>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    else if (algorithm == 22)
        hipLaunchKernelGGL(k_sha_batch<22>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_sha_batch<23>, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, a);
    HIP_TRY(hipGetLastError());
    return 0;
}

int32_t mzhip_deflate_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                            const uint64_t *d_out_off, const uint32_t *d_out_cap, const uint8_t *d_final, uint32_t n,
                            uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status, void *stream) {
    return mzhip_deflate_batch_level(d_in, d_in_off, d_in_len, d_out, d_out_off, d_out_cap, d_final, n, 1, 15, d_out_len, d_crc,
                                     d_status, stream);
}

int32_t mzhip_deflate_batch_level(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, void *d_out,
                                  const uint64_t *d_out_off, const uint32_t *d_out_cap, const uint8_t *d_final, uint32_t n,
                                  int32_t level, int32_t window_log2, uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status,
                                  void *stream) {
    if (n == 0) return 0;
    if (window_log2 < 9 || window_log2 > 15) return MZHIP_STATUS_UNSUPPORTED;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    DeflateArgs a;
    a.in = (const uint8_t *)d_in;
    a.in_off = d_in_off;
    a.in_len = d_in_len;
    a.out = (uint8_t *)d_out;
    a.out_off = d_out_off;
    a.out_cap = d_out_cap;
    a.final_flag = d_final;
    a.n = n;
    a.out_len = d_out_len;
    a.crc = d_crc;
    a.status = d_status;
    CounterLease lease;
    rc = lease.get(c, s);
    if (rc) return rc;
    a.counter = lease.p;
    a.tabs = c->d_tabs;
    /* compression classes (mz_strm_zlib.c:87 hands `level` to deflateInit2): 0-3 fast = one candidate per hash bucket,
     * everything else (4-9, and -1 = Z_DEFAULT_COMPRESSION) = MZ_DEF_WAYS_BEST candidates + a two-position lazy rule */
    a.ways = (level >= 0 && level <= 3) ? 1u : MZ_DEF_WAYS_BEST;
    /* the default class needs 134 KiB of dynamic LDS per workgroup: a device (or runtime) that does not grant it gets the
     * one-candidate class -- a valid stream with a worse ratio, not a launch error */
    a.parse = (a.ways > 1u && level >= 7) ? 1u : 0u; /* levels 7-9 pay for ratio as they do in zlib */
    {
        const size_t big = MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * (MZ_DEF_LDS_STRIDE + MZ_DEF_XHEAD_BYTES);
        if (a.ways > 1u && !(a.parse ? big_lds_ok(c->big_lds_deflate_cost, (const void *)k_deflate_cost_batch, big)
                                     : big_lds_ok(c->big_lds_deflate, (const void *)k_deflate_batch, big))) {
            a.ways = 1u;
            a.parse = 0u;
        }
    }
    a.max_dist = (1u << window_log2) - 262u; /* zlib's MAX_DIST(s) = w_size - MIN_LOOKAHEAD */
    const size_t lds = MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * (MZ_DEF_LDS_STRIDE + (a.ways > 1u ? MZ_DEF_XHEAD_BYTES : 0));
    uint32_t wgs = (n + MZ_WAVES_PER_WG - 1) / MZ_WAVES_PER_WG;
    uint32_t resident = (uint32_t)c->cu_count * (a.ways > 1u ? 1u : 4u); /* 38.3 / 134 KiB LDS per workgroup -> 4 / 1 per CU */
    const uint32_t grid = wgs < resident ? wgs : resident;
    int slot = -1;
    void *scratch = nullptr; /* one token block per resident wave */
    rc = scratch_acquire(c, (size_t)grid * MZ_WAVES_PER_WG * MZ_DEF_BLOCK * sizeof(uint32_t), s, &slot, &scratch);
    if (rc) return rc;
    a.tok = (uint32_t *)scratch;
    if (a.parse) hipLaunchKernelGGL(k_deflate_cost_batch, dim3(grid), dim3(MZ_WAVES_PER_WG * 64), lds, s, a);
    else hipLaunchKernelGGL(k_deflate_batch, dim3(grid), dim3(MZ_WAVES_PER_WG * 64), lds, s, a);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess && a.ways > 1u) { /* the device does not take 134 KiB of LDS per workgroup after all */
        (a.parse ? c->big_lds_deflate_cost : c->big_lds_deflate).store(-1, std::memory_order_release);
        a.ways = 1u;
        a.parse = 0u;
        hipLaunchKernelGGL(k_deflate_batch, dim3(grid), dim3(MZ_WAVES_PER_WG * 64), MZ_CRC_TAB_BYTES + MZ_WAVES_PER_WG * MZ_DEF_LDS_STRIDE, s, a);
        le = hipGetLastError();
    }
    rc = scratch_release(c, slot, s);
    if (le != hipSuccess) return fail("k_deflate_batch", le);
    return rc;
}

int32_t mzhip_lzma_encode_batch(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, uint32_t max_in_len,
                                void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap, const uint8_t *d_mode,
                                uint32_t n, uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status, void *stream) {
    return mzhip_lzma_encode_batch_preset(d_in, d_in_off, d_in_len, max_in_len, d_out, d_out_off, d_out_cap, d_mode, n, 1,
                                          d_out_len, d_crc, d_status, stream);
}

/* preset (mz_strm_lzma.c:81 hands COMPRESS_LEVEL to lzma_lzma_preset): 0-3 -> one hash candidate per position, 4-9 and
 * the default (-1 = 6) -> MZ_DEF_WAYS_BEST candidates + the two-position lazy rule, as K4's classes */
int32_t mzhip_lzma_encode_batch_preset(const void *d_in, const uint64_t *d_in_off, const uint32_t *d_in_len, uint32_t max_in_len,
                                       void *d_out, const uint64_t *d_out_off, const uint32_t *d_out_cap, const uint8_t *d_mode,
                                       uint32_t n, int32_t preset, uint32_t *d_out_len, uint32_t *d_crc, int32_t *d_status,
                                       void *stream) {
    if (n == 0) return 0;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    LzmaEncArgs a;
    a.in = (const uint8_t *)d_in;
    a.in_off = d_in_off;
    a.in_len = d_in_len;
    a.out = (uint8_t *)d_out;
    a.out_off = d_out_off;
    a.out_cap = d_out_cap;
    a.mode = d_mode;
    a.n = n;
    a.maxb = max_in_len ? (max_in_len + MZ_DEF_BLOCK - 1) / MZ_DEF_BLOCK : 1u;
    a.ways = (preset >= 0 && preset <= 3) ? 1u : MZ_DEF_WAYS_BEST;
    if (a.ways > 1u && !big_lds_ok(c->big_lds_tok, (const void *)k_lz_tokenize_batch, MZ_WAVES_PER_WG * (sizeof(mz_lz_tok_lds) + MZ_DEF_XHEAD_BYTES)))
        a.ways = 1u; /* (128 KiB of dynamic LDS per workgroup not granted: the one-candidate class) */
    a.out_len = d_out_len;
    a.crc = d_crc;
    a.status = d_status;
    a.tabs = c->d_tabs;
    /* token scratch: 4 bytes per input position of the largest entry, for every entry */
    const size_t items = (size_t)n * a.maxb;
    int slot = -1;
    void *scratch = nullptr;
    rc = scratch_acquire(c, items * MZ_DEF_BLOCK * sizeof(uint32_t) + items * sizeof(uint32_t) + 64, s, &slot, &scratch);
    if (rc) return rc;
    a.tok = (uint32_t *)scratch;
    a.ntok = a.tok + items * MZ_DEF_BLOCK;
    a.counter = a.ntok + items; /* two work counters behind the token counts */
    {
        const hipError_t me = hipMemsetAsync(a.counter, 0, 2 * sizeof(uint32_t), s);
        if (me != hipSuccess) {
            (void)scratch_release(c, slot, s);
            return fail("hipMemsetAsync", me);
        }
    }
    {
        uint32_t wgs = (uint32_t)((items + MZ_WAVES_PER_WG - 1) / MZ_WAVES_PER_WG);
        const size_t lds = MZ_WAVES_PER_WG * (sizeof(mz_lz_tok_lds) + (a.ways > 1u ? MZ_DEF_XHEAD_BYTES : 0));
        uint32_t resident = (uint32_t)c->cu_count * (a.ways > 1u ? 1u : 4u); /* 32 KiB (128 KiB) of LDS per workgroup */
        hipLaunchKernelGGL(k_lz_tokenize_batch, dim3(wgs < resident ? wgs : resident), dim3(MZ_WAVES_PER_WG * 64), lds, s, a);
        if (a.ways > 1u && hipGetLastError() != hipSuccess) { /* 128 KiB of LDS per workgroup refused after all: the one-candidate class */
            c->big_lds_tok.store(-1, std::memory_order_release);
            a.ways = 1u;
            resident = (uint32_t)c->cu_count * 4u;
            hipLaunchKernelGGL(k_lz_tokenize_batch, dim3(wgs < resident ? wgs : resident), dim3(MZ_WAVES_PER_WG * 64),
                               MZ_WAVES_PER_WG * sizeof(mz_lz_tok_lds), s, a);
        }
    }
    {
        uint32_t resident = (uint32_t)c->cu_count * 9u; /* 17 KiB LDS per single-wave workgroup */
        hipLaunchKernelGGL(k_lzma_rc_encode_batch, dim3(n < resident ? n : resident), dim3(64), 0, s, a);
    }
    const hipError_t le = hipGetLastError();
    rc = scratch_release(c, slot, s);
    if (le != hipSuccess) return fail("k_lzma_rc_encode_batch", le);
    return rc;
}

// ---- host-buffer conveniences (synchronous): staging through one scratch allocation per call

namespace {
struct Scratch {
    void *p = nullptr;
    ~Scratch() {
        if (p) (void)hipFree(p);
    }
};
// The synchronous host-buffer entry points (one entry at a time: what the vtbl shims call) run on the calling thread's
// own stream: copies and launches are ordered on it and only that stream is waited for, so two host
// threads never serialise on the null stream or on a device-wide synchronisation.
// (hipStreamPerThread itself was the first choice; with two host threads decoding entries at the same time it handed
// back garbage result words now and then -- tests/test_gpu_dropin.py::test_archives_through_unmodified_mz_zip, one run in
// three -- so the per-thread stream is one this library creates: a non-blocking stream per (host thread, device), made
// on first use.)
// A thread that exits hands its streams to a free list instead of leaking them (an application that makes a reader pool
// per archive used to leave one stream per exited thread behind); they are recycled, never destroyed: the
// scratch and work-queue caches remember stream identities ("the next launch is on the stream that used it last"), and a
// recycled stream keeps the order that reasoning relies on, where a destroyed one's handle could come back as a stranger.
struct StreamPool {
    std::mutex mu;
    std::vector<hipStream_t> idle[kMaxDevices];
};
static StreamPool *stream_pool() {
    static StreamPool *p = new StreamPool(); // (never deleted: thread_local destructors may run after static ones)
    return p;
}
struct ThreadStreams {
    hipStream_t s[kMaxDevices] = {};
    ~ThreadStreams() {
        StreamPool *p = stream_pool();
        std::lock_guard<std::mutex> g(p->mu);
        for (int d = 0; d < kMaxDevices; d++)
            if (s[d]) p->idle[d].push_back(s[d]);
    }
};
static thread_local ThreadStreams t_streams;
static hipStream_t mz_host_stream() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDevices) return nullptr;
    if (!t_streams.s[d]) {
        StreamPool *p = stream_pool();
        {
            std::lock_guard<std::mutex> g(p->mu);
            if (!p->idle[d].empty()) {
                t_streams.s[d] = p->idle[d].back();
                p->idle[d].pop_back();
            }
        }
        if (!t_streams.s[d] && hipStreamCreateWithFlags(&t_streams.s[d], hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            t_streams.s[d] = nullptr; /* the null stream still works, only slower */
        }
    }
    return t_streams.s[d];
}
#define MZ_HOST_STREAM mz_host_stream()
static inline hipError_t mz_h2d(void *dst, const void *src, size_t n) {
    return hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, MZ_HOST_STREAM); // (pageable source: staged before the call returns)
}
static inline hipError_t mz_d2h(void *dst, const void *src, size_t n) {
    const hipError_t e = hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToHost, MZ_HOST_STREAM);
    return e != hipSuccess ? e : hipStreamSynchronize(MZ_HOST_STREAM);
}
// Staging of the synchronous host-buffer calls: a buffer of the scratch cache instead of a hipMalloc / hipFree pair
// per call (hipFree alone is a device-wide synchronisation); released when the call returns, after its own sync.
struct Staging {
    DeviceCtx *c = nullptr;
    int slot = -1;
    void *p = nullptr;
    int32_t get(DeviceCtx *ctx, size_t bytes) {
        c = ctx;
        return scratch_acquire(ctx, bytes, MZ_HOST_STREAM, &slot, &p);
    }
    ~Staging() {
        if (slot >= 0) (void)scratch_release(c, slot, MZ_HOST_STREAM);
    }
};
} // namespace

// One window of a stream that is decoded window by window (the READ shim's bounded-memory path): buf[0 .. state_in->out_pos)
// is the history the caller kept (the last 32 KiB it was given, nothing at the start of the stream), the new bytes land
// behind it, at most buf_cap bytes in all.  Returns the device verdict: MZHIP_OK (stream end), MZHIP_OUT_FULL (call again
// with state_out and the tail of buf as history), MZHIP_BUF_ERROR (call again with more input from state_out's block
// header on), or a data error.  *out_len = bytes valid in buf (history included), *crc = CRC-32 of the new bytes only.
int32_t mzhip_inflate_resume_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                  const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out, uint32_t *out_len,
                                  uint32_t *in_used, uint32_t *crc) {
    return mzhip_inflate_resume_host_seg(in, in_len, buf, buf_cap, state_in, state_out, out_len, in_used, crc, 0, 0, nullptr, 0, nullptr);
}

int32_t mzhip_inflate_resume_host_seg(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                      const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out, uint32_t *out_len,
                                      uint32_t *in_used, uint32_t *crc, uint32_t seg_first, uint32_t seg_stride,
                                      uint32_t *seg_crc, uint32_t seg_cap, uint32_t *nseg) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const uint32_t hist = state_in ? state_in->out_pos : 0u;
    if (hist > buf_cap) return -102; /* MZ_PARAM_ERROR */
    // layout: [meta 128 B][in (16-aligned)][buf][segment arrays]
    const size_t in_pad = ((size_t)in_len + 15) & ~(size_t)15;
    const size_t buf_pad = ((size_t)buf_cap + 16 + 15) & ~(size_t)15;
    const size_t seg_max = (seg_stride && seg_crc) ? (size_t)buf_cap / seg_stride + 3 : 0;
    const size_t total = 128 + in_pad + buf_pad + seg_max * 16;
    if (nseg) *nseg = 0;
    Staging sc;
    rc = sc.get(c, total);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        uint64_t in_off, out_off;
        uint32_t in_len, out_cap, out_len, in_used, crc;
        int32_t status;
        mz_inflate_state rs, st;
    } m;
    memset(&m, 0, sizeof(m));
    m.in_off = 128;
    m.out_off = 128 + in_pad;
    m.in_len = in_len;
    m.out_cap = buf_cap;
    if (state_in) memcpy(&m.rs, state_in, sizeof(m.rs));
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (in_len) HIP_TRY(mz_h2d(base + 128, in, in_len));
    if (hist) HIP_TRY(mz_h2d(base + m.out_off, buf, hist));
    Meta *dm = (Meta *)base;
    rc = mzhip_inflate_resume_batch(base, &dm->in_off, &dm->in_len, base, &dm->out_off, &dm->out_cap, 1, &dm->out_len,
                                    &dm->in_used, &dm->crc, &dm->status, (const mzhip_inflate_state *)&dm->rs,
                                    /* no state asked for = the last call of a stream that ended short: the kernel then drops the
                                     * resumable rules (a stored block is taken as far as it goes) -- the pointer used to be
                                     * passed regardless, and a truncated stored stream in window mode lost its last bytes
                                     * (found by tests/test_gpu_dropin.py::test_truncation_accounting_window_mode, round 4) */
                                    state_out ? (mzhip_inflate_state *)&dm->st : nullptr, MZ_HOST_STREAM);
    if (rc) return rc;
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.out_len > hist) HIP_TRY(mz_d2h(buf + hist, base + m.out_off + hist, m.out_len - hist));
    if (seg_max && m.out_len > hist) {
        /* CRC-32 of the new bytes in the pieces the caller will hand to mz_crypt_crc32_update: the first seg_first bytes
         * (what completes the piece the previous window left open), then seg_stride at a time, the rest; computed from the
         * device's copy of the window, one launch */
        std::vector<uint64_t> off;
        std::vector<uint32_t> len;
        uint32_t pos = hist;
        uint32_t first = seg_first < m.out_len - hist ? seg_first : m.out_len - hist;
        if (first) {
            off.push_back(m.out_off + pos);
            len.push_back(first);
            pos += first;
        }
        while (pos < m.out_len) {
            const uint32_t n = m.out_len - pos < seg_stride ? m.out_len - pos : seg_stride;
            off.push_back(m.out_off + pos);
            len.push_back(n);
            pos += n;
        }
        const uint32_t ns = (uint32_t)len.size();
        if (ns <= seg_cap && ns <= seg_max) {
            uint8_t *sm = base + 128 + in_pad + buf_pad;
            uint64_t *d_off = (uint64_t *)sm;
            uint32_t *d_len = (uint32_t *)(sm + seg_max * 8), *d_crc = d_len + seg_max;
            HIP_TRY(mz_h2d(d_off, off.data(), (size_t)ns * 8));
            HIP_TRY(mz_h2d(d_len, len.data(), (size_t)ns * 4));
            rc = mzhip_crc32_batch(base, d_off, d_len, ns, nullptr, d_crc, MZ_HOST_STREAM);
            if (rc) return rc;
            HIP_TRY(mz_d2h(seg_crc, d_crc, (size_t)ns * 4));
            if (nseg) *nseg = ns;
        }
    }
    if (out_len) *out_len = m.out_len;
    if (in_used) *in_used = m.in_used;
    if (crc) *crc = m.crc;
    if (state_out) memcpy(state_out, &m.st, sizeof(m.st));
    return m.status;
}

int32_t mzhip_inflate_host2(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                            uint32_t *in_used, uint32_t *crc, uint32_t *adler) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    // layout: [meta 64 B][in (16-aligned)][out]
    const size_t in_pad = ((size_t)in_len + 15) & ~(size_t)15;
    const size_t total = 64 + in_pad + out_cap + 16;
    Staging sc;
    rc = sc.get(c, total);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        uint64_t in_off, out_off;
        uint32_t in_len, out_cap, out_len, in_used, crc;
        int32_t status;
        uint32_t adler, pad;
    } m;
    memset(&m, 0, sizeof(m));
    m.in_off = 64;
    m.out_off = 64 + in_pad;
    m.in_len = in_len;
    m.out_cap = out_cap;
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (in_len) HIP_TRY(mz_h2d(base + 64, in, in_len));
    Meta *dm = (Meta *)base;
    rc = mzhip_inflate_batch(base, &dm->in_off, &dm->in_len, base, &dm->out_off, &dm->out_cap, 1, &dm->out_len,
                             &dm->in_used, &dm->crc, &dm->status, MZ_HOST_STREAM);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
    if (adler) { /* zlib wrapper: Adler-32 of the decoded bytes, reduced on the device as well */
        HIP_TRY(mz_d2h(&m, base, sizeof(m)));
        rc = mzhip_adler32_batch(base, &dm->out_off, &dm->out_len, 1, &dm->adler, MZ_HOST_STREAM);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
    }
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.out_len && out) HIP_TRY(mz_d2h(out, base + m.out_off, m.out_len));
    if (out_len) *out_len = m.out_len;
    if (in_used) *in_used = m.in_used;
    if (crc) *crc = m.crc;
    if (adler) *adler = m.adler;
    return m.status;
}

int32_t mzhip_inflate_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                           uint32_t *in_used, uint32_t *crc) {
    return mzhip_inflate_host2(in, in_len, out, out_cap, out_len, in_used, crc, nullptr);
}

static int32_t lzma_family_host(int xz, const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                                uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const size_t in_pad = ((size_t)in_len + 15) & ~(size_t)15;
    const size_t total = 64 + in_pad + out_cap + 16;
    Staging sc;
    rc = sc.get(c, total);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        uint64_t in_off, out_off;
        int64_t max_out;
        uint32_t in_len, out_cap, out_len, in_used, crc;
        int32_t status;
    } m;
    memset(&m, 0, sizeof(m));
    m.in_off = 64;
    m.out_off = 64 + in_pad;
    m.max_out = max_out;
    m.in_len = in_len;
    m.out_cap = out_cap;
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (in_len) HIP_TRY(mz_h2d(base + 64, in, in_len));
    Meta *dm = (Meta *)base;
    rc = lzma_family_batch(xz, base, &dm->in_off, &dm->in_len, base, &dm->out_off, &dm->out_cap, &dm->max_out, 1,
                           &dm->out_len, &dm->in_used, &dm->crc, &dm->status, MZ_HOST_STREAM);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.out_len && out) HIP_TRY(mz_d2h(out, base + m.out_off, m.out_len));
    if (out_len) *out_len = m.out_len;
    if (in_used) *in_used = m.in_used;
    if (crc) *crc = m.crc;
    return m.status;
}

int32_t mzhip_lzma_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                        uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    return lzma_family_host(0, in, in_len, out, out_cap, max_out, out_len, in_used, crc);
}

int32_t mzhip_xz_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                      uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    return lzma_family_host(1, in, in_len, out, out_cap, max_out, out_len, in_used, crc);
}

uint32_t mzhip_lzma_model_bytes(void) { return (uint32_t)(MZ_LZMA_MODEL_U16 * sizeof(uint16_t)); }

// One window of one ZIP method-14 payload (include/mzhip.h).  buf[0 .. state_in->out_pos) is the dictionary so far.
int32_t mzhip_lzma_resume_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap, const mzhip_lzma_state *state_in,
                               mzhip_lzma_state *state_out, void *model, uint32_t *out_len, uint32_t *in_used) {
    static_assert(sizeof(mzhip_lzma_state) == sizeof(mz_lzma_state), "mzhip.h and lzma_core.h describe the same sixteen words");
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    if (!model || !state_in) return -102; /* MZ_PARAM_ERROR */
    const uint32_t hist = state_in->out_pos;
    if (hist > buf_cap) return -102;
    const size_t model_bytes = (MZ_LZMA_MODEL_U16 * sizeof(uint16_t) + 15) & ~(size_t)15;
    const size_t in_pad = ((size_t)in_len + 15 + 16) & ~(size_t)15;
    const size_t total = 256 + model_bytes + in_pad + (size_t)buf_cap + 16;
    Staging sc;
    rc = sc.get(c, total);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        mz_lzma_state rs, st;
        uint32_t out_len, in_used;
        int32_t status;
    } m;
    static_assert(sizeof(Meta) <= 256, "meta block");
    memset(&m, 0, sizeof(m));
    memcpy(&m.rs, state_in, sizeof(m.rs));
    uint8_t *d_model = base + 256, *d_in = d_model + model_bytes, *d_buf = d_in + in_pad;
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (state_in->flags & 1u) HIP_TRY(mz_h2d(d_model, model, MZ_LZMA_MODEL_U16 * sizeof(uint16_t)));
    if (in_len) HIP_TRY(mz_h2d(d_in, in, in_len));
    if (hist) HIP_TRY(mz_h2d(d_buf, buf, hist));
    Meta *dm = (Meta *)base;
    LzmaResumeArgs a{d_in, in_len, d_buf, buf_cap, &dm->rs, state_out ? &dm->st : nullptr, (uint16_t *)d_model,
                     &dm->out_len, &dm->in_used, &dm->status, c->d_tabs};
    hipLaunchKernelGGL(k_lzma_resume, dim3(1), dim3(64), 0, MZ_HOST_STREAM, a);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail("k_lzma_resume", le);
    HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.out_len > hist && m.out_len <= buf_cap) HIP_TRY(mz_d2h(buf + hist, d_buf + hist, m.out_len - hist));
    if (state_out) {
        memcpy(state_out, &m.st, sizeof(m.st));
        if (m.st.flags & 1u) HIP_TRY(mz_d2h(model, d_model, MZ_LZMA_MODEL_U16 * sizeof(uint16_t)));
    }
    if (out_len) *out_len = m.out_len;
    if (in_used) *in_used = m.in_used;
    return m.status;
}

// One ZIP method-14 payload from a host buffer.
int32_t mzhip_lzma_encode_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                               uint32_t *crc) {
    return mzhip_lzma_encode_host_preset(in, in_len, 1, out, out_cap, out_len, crc);
}

int32_t mzhip_lzma_encode_host_preset(const uint8_t *in, uint32_t in_len, int32_t preset, uint8_t *out, uint32_t out_cap,
                                      uint32_t *out_len, uint32_t *crc) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const size_t in_pad = ((size_t)in_len + 63) & ~(size_t)63;
    const uint32_t cap = in_len + in_len / 8 + 1024;
    Staging sc;
    rc = sc.get(c, 64 + in_pad + cap);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        uint64_t in_off, out_off;
        uint32_t in_len, out_cap, out_len, crc;
        int32_t status;
    } m;
    memset(&m, 0, sizeof(m));
    m.in_off = 64;
    m.out_off = 64 + in_pad;
    m.in_len = in_len;
    m.out_cap = cap;
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (in_len) HIP_TRY(mz_h2d(base + 64, in, in_len));
    Meta *dm = (Meta *)base;
    rc = mzhip_lzma_encode_batch_preset(base, &dm->in_off, &dm->in_len, in_len, base, &dm->out_off, &dm->out_cap, nullptr, 1,
                                        preset, &dm->out_len, &dm->crc, &dm->status, MZ_HOST_STREAM);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.status == 0 && m.out_len > out_cap) m.status = MZHIP_STATUS_OUT_FULL;
    if (m.status == 0 && m.out_len) HIP_TRY(mz_d2h(out, base + m.out_off, m.out_len));
    if (out_len) *out_len = m.out_len;
    if (crc) *crc = m.crc;
    return m.status;
}

// One segment of one ZIP method-14 payload (include/mzhip.h): in = [skip_blocks x 64 KiB of the stream's previous bytes |
// the segment].
int32_t mzhip_lzma_encode_resume_host(const uint8_t *in, uint32_t in_len, uint32_t skip_blocks, uint32_t last, int32_t preset,
                                      const mzhip_lzma_enc_state *state_in, mzhip_lzma_enc_state *state_out, void *model,
                                      uint8_t *out, uint32_t out_cap, uint32_t *out_len) {
    static_assert(sizeof(mzhip_lzma_enc_state) == sizeof(mz_lzma_enc_state), "mzhip.h and lzma_enc_core.h describe the same sixteen words");
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    if (!model || !state_in || (!last && !state_out) || (uint64_t)skip_blocks * MZ_DEF_BLOCK > in_len) return -102; /* MZ_PARAM_ERROR */
    const size_t model_bytes = (((LZ_NUM_PROBS + 1u) & ~1u) * sizeof(uint16_t) + 15) & ~(size_t)15;
    const size_t in_pad = ((size_t)in_len + 63) & ~(size_t)63;
    const uint32_t cap = in_len + in_len / 8 + 1024;
    const uint32_t maxb = in_len ? (in_len + MZ_DEF_BLOCK - 1) / MZ_DEF_BLOCK : 1u;
    const size_t tok_bytes = (size_t)maxb * MZ_DEF_BLOCK * sizeof(uint32_t) + (size_t)maxb * sizeof(uint32_t) + 64;
    Staging sc;
    rc = sc.get(c, 256 + model_bytes + in_pad + cap + 64 + tok_bytes);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    struct Meta {
        mz_lzma_enc_state rs, st;
        uint64_t in_off, out_off;
        uint32_t in_len, out_cap, out_len, crc;
        int32_t status;
    } m;
    static_assert(sizeof(Meta) <= 256, "meta block");
    memset(&m, 0, sizeof(m));
    memcpy(&m.rs, state_in, sizeof(m.rs));
    uint8_t *d_model = base + 256, *d_in = d_model + model_bytes, *d_out = d_in + in_pad;
    m.in_off = (uint64_t)(d_in - base);
    m.out_off = (uint64_t)(d_out - base);
    m.in_len = in_len;
    m.out_cap = cap;
    HIP_TRY(mz_h2d(base, &m, sizeof(m)));
    if (state_in->flags & 1u) HIP_TRY(mz_h2d(d_model, model, ((LZ_NUM_PROBS + 1u) & ~1u) * sizeof(uint16_t)));
    if (in_len) HIP_TRY(mz_h2d(d_in, in, in_len));
    Meta *dm = (Meta *)base;
    hipStream_t s = MZ_HOST_STREAM;
    LzmaEncResumeArgs r;
    LzmaEncArgs &a = r.a;
    a.in = base;
    a.in_off = &dm->in_off;
    a.in_len = &dm->in_len;
    a.out = base;
    a.out_off = &dm->out_off;
    a.out_cap = &dm->out_cap;
    a.mode = nullptr;
    a.n = 1;
    a.maxb = maxb;
    a.ways = (preset >= 0 && preset <= 3) ? 1u : MZ_DEF_WAYS_BEST;
    if (a.ways > 1u && !big_lds_ok(c->big_lds_tok, (const void *)k_lz_tokenize_batch, MZ_WAVES_PER_WG * (sizeof(mz_lz_tok_lds) + MZ_DEF_XHEAD_BYTES)))
        a.ways = 1u;
    a.out_len = &dm->out_len;
    a.crc = &dm->crc;
    a.status = &dm->status;
    a.tabs = c->d_tabs;
    a.tok = (uint32_t *)(d_out + ((cap + 63u) & ~63u));
    a.ntok = a.tok + (size_t)maxb * MZ_DEF_BLOCK;
    a.counter = a.ntok + maxb;
    HIP_TRY(hipMemsetAsync(a.counter, 0, 2 * sizeof(uint32_t), s));
    {
        const uint32_t wgs = (maxb + MZ_WAVES_PER_WG - 1) / MZ_WAVES_PER_WG;
        const size_t lds = MZ_WAVES_PER_WG * (sizeof(mz_lz_tok_lds) + (a.ways > 1u ? MZ_DEF_XHEAD_BYTES : 0));
        uint32_t resident = (uint32_t)c->cu_count * (a.ways > 1u ? 1u : 4u);
        hipLaunchKernelGGL(k_lz_tokenize_batch, dim3(wgs < resident ? wgs : resident), dim3(MZ_WAVES_PER_WG * 64), lds, s, a);
        if (a.ways > 1u && hipGetLastError() != hipSuccess) {
            c->big_lds_tok.store(-1, std::memory_order_release);
            a.ways = 1u;
            resident = (uint32_t)c->cu_count * 4u;
            hipLaunchKernelGGL(k_lz_tokenize_batch, dim3(wgs < resident ? wgs : resident), dim3(MZ_WAVES_PER_WG * 64),
                               MZ_WAVES_PER_WG * sizeof(mz_lz_tok_lds), s, a);
        }
    }
    r.skip_blocks = skip_blocks;
    r.rs = &dm->rs;
    r.st = last ? nullptr : &dm->st;
    r.model = (uint16_t *)d_model;
    hipLaunchKernelGGL(k_lzma_rc_encode_resume, dim3(1), dim3(64), 0, s, r);
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return fail("k_lzma_rc_encode_resume", le);
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(mz_d2h(&m, base, sizeof(m)));
    if (m.status == 0 && m.out_len > out_cap) m.status = MZHIP_STATUS_OUT_FULL;
    if (m.status == 0 && m.out_len) HIP_TRY(mz_d2h(out, d_out, m.out_len));
    if (m.status == 0 && !last) {
        memcpy(state_out, &m.st, sizeof(m.st));
        HIP_TRY(mz_d2h(model, d_model, ((LZ_NUM_PROBS + 1u) & ~1u) * sizeof(uint16_t)));
    }
    if (out_len) *out_len = m.out_len;
    return m.status;
}

// One .xz stream (single block, CRC32 check) from a host buffer: the input is cut into 48 KiB LZMA2 chunks that
// reset dictionary, state and properties, so every chunk is an independent stream and all of them are coded in one
// batch; framing bytes (a few per chunk) are laid out here, their CRC-32s come from the device as well.
int32_t mzhip_xz_encode_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                             uint32_t *crc) {
    return mzhip_xz_encode_host_preset(in, in_len, 1, out, out_cap, out_len, crc);
}

// part = 0: a whole .xz stream (header, one block, index, footer).  part = 1: [the stream header when `first`] + ONE block;
// *unpadded receives the block's unpadded size for the index (mzhip_xz_encode_finish_host writes it)
static int32_t xz_encode_impl(const uint8_t *in, uint32_t in_len, int32_t preset, int part, int first, uint8_t *out, uint32_t out_cap,
                              uint32_t *out_len, uint32_t *crc, uint64_t *unpadded) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const uint32_t piece = 48u << 10;
    const uint32_t np = (in_len + piece - 1) / piece;
    const uint32_t pcap = piece + piece / 8 + 1024;
    const size_t meta = (size_t)np * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 1) + 64;
    const size_t meta_pad = (meta + 63) & ~(size_t)63;
    const size_t in_pad = ((size_t)in_len + 63) & ~(size_t)63;
    Staging sc;
    rc = sc.get(c, meta_pad + in_pad + (size_t)np * pcap + 64);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    std::vector<uint8_t> hm(meta_pad, 0);
    uint64_t *h_in_off = (uint64_t *)hm.data(), *h_out_off = h_in_off + np;
    uint32_t *h_in_len = (uint32_t *)(h_out_off + np), *h_out_cap = h_in_len + np, *h_out_len = h_out_cap + np,
             *h_crc = h_out_len + np;
    int32_t *h_status = (int32_t *)(h_crc + np);
    uint8_t *h_mode = (uint8_t *)(h_status + np);
    for (uint32_t i = 0; i < np; i++) {
        h_in_off[i] = meta_pad + (uint64_t)i * piece;
        h_in_len[i] = (in_len - i * piece < piece) ? in_len - i * piece : piece;
        h_out_off[i] = meta_pad + in_pad + (uint64_t)i * pcap;
        h_out_cap[i] = pcap;
        h_mode[i] = 1;
    }
    uint32_t total_crc = 0;
    if (np) {
        HIP_TRY(mz_h2d(base, hm.data(), meta));
        HIP_TRY(mz_h2d(base + meta_pad, in, in_len));
        uint64_t *d_in_off = (uint64_t *)base, *d_out_off = d_in_off + np;
        uint32_t *d_in_len = (uint32_t *)(d_out_off + np), *d_out_cap = d_in_len + np, *d_out_len = d_out_cap + np,
                 *d_crc = d_out_len + np;
        int32_t *d_status = (int32_t *)(d_crc + np);
        uint8_t *d_mode = (uint8_t *)(d_status + np);
        rc = mzhip_lzma_encode_batch_preset(base, d_in_off, d_in_len, piece, base, d_out_off, d_out_cap, d_mode, np, preset, d_out_len,
                                     d_crc, d_status, MZ_HOST_STREAM);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(MZ_HOST_STREAM));
        HIP_TRY(mz_d2h(hm.data(), base, meta));
    }
    // ---- container (The .xz File Format 1.0.4): stream header, one block, index, footer
    uint32_t pos = 0;
    auto put = [&](const void *p, uint32_t n) -> bool {
        if (n > out_cap - pos) return false;
        memcpy(out + pos, p, n);
        pos += n;
        return true;
    };
    auto le32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); };
    auto vli = [](uint8_t *p, uint64_t v) -> uint32_t {
        uint32_t k = 0;
        while (v >= 0x80) { p[k++] = (uint8_t)(v | 0x80); v >>= 7; }
        p[k++] = (uint8_t)v;
        return k;
    };
    uint8_t hdr[12] = {0xFD, '7', 'z', 'X', 'Z', 0x00, 0x00, 0x01 /* check: CRC32 */, 0, 0, 0, 0};
    le32(hdr + 8, mzhip_crc32_host(0, hdr + 6, 2));
    uint8_t bh[12] = {0x02 /* (2 + 1) * 4 bytes */, 0x00 /* one filter, no sizes */, 0x21 /* LZMA2 */, 0x01, 0x08 /* 64 KiB */, 0, 0, 0, 0, 0, 0, 0};
    le32(bh + 8, mzhip_crc32_host(0, bh, 8));
    if ((!part || first) && !put(hdr, 12)) return MZHIP_STATUS_OUT_FULL;
    if (!put(bh, 12)) return MZHIP_STATUS_OUT_FULL;
    const uint32_t data_start = pos;
    for (uint32_t i = 0; i < np; i++) {
        if (h_status[i] != 0) return h_status[i];
        const uint32_t us = h_in_len[i], cs = h_out_len[i];
        total_crc = (i == 0) ? h_crc[0] : mzhip_crc32_combine_host(total_crc, h_crc[i], us);
        if (cs >= us || cs > 65536u) { /* stored chunk: control 0x01 (dictionary reset), size - 1 big endian, the bytes */
            uint8_t ch[3] = {0x01, (uint8_t)((us - 1) >> 8), (uint8_t)(us - 1)};
            if (!put(ch, 3) || !put(in + (size_t)i * piece, us)) return MZHIP_STATUS_OUT_FULL;
        } else { /* LZMA chunk resetting dictionary, state and properties: 0xE0 | size bits, sizes - 1, props */
            uint8_t ch[6] = {(uint8_t)(0xE0 | ((us - 1) >> 16)), (uint8_t)((us - 1) >> 8), (uint8_t)(us - 1),
                             (uint8_t)((cs - 1) >> 8), (uint8_t)(cs - 1), MZ_LZE_PROPS};
            if (!put(ch, 6) || cs > out_cap - pos) return MZHIP_STATUS_OUT_FULL;
            HIP_TRY(mz_d2h(out + pos, base + h_out_off[i], cs));
            pos += cs;
        }
    }
    const uint8_t zero4[4] = {0, 0, 0, 0};
    if (!put(zero4, 1)) return MZHIP_STATUS_OUT_FULL; /* end of the LZMA2 data */
    const uint32_t csize_blk = pos - data_start;
    if (!put(zero4, (4u - (csize_blk & 3u)) & 3u)) return MZHIP_STATUS_OUT_FULL;
    uint8_t chk[4];
    le32(chk, total_crc);
    if (!put(chk, 4)) return MZHIP_STATUS_OUT_FULL;
    if (unpadded) *unpadded = 12ull + csize_blk + 4ull;
    if (part) {
        if (out_len) *out_len = pos;
        if (crc) *crc = total_crc;
        return 0;
    }
    uint8_t idx[32];
    uint32_t k = 0;
    idx[k++] = 0x00;
    idx[k++] = 0x01;
    k += vli(idx + k, 12ull + csize_blk + 4ull);
    k += vli(idx + k, in_len);
    while (k & 3u) idx[k++] = 0;
    le32(idx + k, mzhip_crc32_host(0, idx, k));
    k += 4;
    if (!put(idx, k)) return MZHIP_STATUS_OUT_FULL;
    uint8_t ft[12];
    le32(ft + 4, k / 4 - 1);
    ft[8] = 0x00;
    ft[9] = 0x01;
    le32(ft, mzhip_crc32_host(0, ft + 4, 6));
    ft[10] = 'Y';
    ft[11] = 'Z';
    if (!put(ft, 12)) return MZHIP_STATUS_OUT_FULL;
    if (out_len) *out_len = pos;
    if (crc) *crc = total_crc;
    return 0;
}

int32_t mzhip_xz_encode_host_preset(const uint8_t *in, uint32_t in_len, int32_t preset, uint8_t *out, uint32_t out_cap,
                                    uint32_t *out_len, uint32_t *crc) {
    return xz_encode_impl(in, in_len, preset, 0, 1, out, out_cap, out_len, crc, nullptr);
}

// A .xz stream written block by block in bounded memory (include/mzhip.h): one block per call ...
int32_t mzhip_xz_encode_block_host(const uint8_t *in, uint32_t in_len, int32_t preset, int32_t first, uint8_t *out, uint32_t out_cap,
                                   uint32_t *out_len, uint32_t *crc, uint64_t *unpadded_size) {
    if (!in_len || !unpadded_size) return -102; /* MZ_PARAM_ERROR: a block holds at least one byte */
    return xz_encode_impl(in, in_len, preset, 1, first, out, out_cap, out_len, crc, unpadded_size);
}
// ... then the index over all blocks and the stream footer (host arithmetic only: a few dozen bytes)
int32_t mzhip_xz_encode_finish_host(const uint64_t *unpadded_size, const uint64_t *uncompressed_size, uint32_t nblocks, uint8_t *out,
                                    uint32_t out_cap, uint32_t *out_len) {
    std::vector<uint8_t> idx;
    auto vli = [&](uint64_t v) {
        while (v >= 0x80) { idx.push_back((uint8_t)(v | 0x80)); v >>= 7; }
        idx.push_back((uint8_t)v);
    };
    auto le32 = [](uint8_t *p, uint32_t v) { p[0] = (uint8_t)v; p[1] = (uint8_t)(v >> 8); p[2] = (uint8_t)(v >> 16); p[3] = (uint8_t)(v >> 24); };
    idx.push_back(0x00);
    vli(nblocks);
    for (uint32_t i = 0; i < nblocks; i++) {
        vli(unpadded_size[i]);
        vli(uncompressed_size[i]);
    }
    while (idx.size() & 3u) idx.push_back(0);
    uint8_t c4[4];
    le32(c4, mzhip_crc32_host(0, idx.data(), idx.size()));
    idx.insert(idx.end(), c4, c4 + 4);
    uint8_t ft[12];
    le32(ft + 4, (uint32_t)(idx.size() / 4 - 1));
    ft[8] = 0x00;
    ft[9] = 0x01;
    le32(ft, mzhip_crc32_host(0, ft + 4, 6));
    ft[10] = 'Y';
    ft[11] = 'Z';
    if (idx.size() + 12 > out_cap) return MZHIP_STATUS_OUT_FULL;
    memcpy(out, idx.data(), idx.size());
    memcpy(out + idx.size(), ft, 12);
    if (out_len) *out_len = (uint32_t)idx.size() + 12;
    return 0;
}

// One stream segment: split into 64 KiB pieces (one wave each); every piece but the last ends with an empty
// stored block so the pieces concatenate on byte boundaries; the last piece is final iff `final`.
int32_t mzhip_deflate_host2(const uint8_t *in, uint32_t in_len, uint32_t final, uint8_t *out, uint32_t out_cap,
                            uint32_t *out_len, uint32_t *crc, uint32_t *adler) {
    return mzhip_deflate_host_level(in, in_len, final, 1, 15, out, out_cap, out_len, crc, adler);
}

int32_t mzhip_deflate_host_level(const uint8_t *in, uint32_t in_len, uint32_t final, int32_t level, int32_t window_log2,
                                 uint8_t *out, uint32_t out_cap, uint32_t *out_len, uint32_t *crc, uint32_t *adler) {
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const uint32_t piece = 64u << 10;
    const uint32_t np = in_len ? (in_len + piece - 1) / piece : 1u;
    const uint32_t pcap = piece + piece / 8 + 64; /* fixed-Huffman worst case is 9/8 of the input */
    const size_t meta = (size_t)np * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 4 + 1);
    const size_t meta_pad = (meta + 63) & ~(size_t)63;
    const size_t in_pad = ((size_t)in_len + 63) & ~(size_t)63;
    Staging sc;
    rc = sc.get(c, meta_pad + in_pad + (size_t)np * pcap);
    if (rc) return rc;
    uint8_t *base = (uint8_t *)sc.p;
    uint8_t *hm = (uint8_t *)calloc(1, meta_pad);
    if (!hm) return -4;
    uint64_t *h_in_off = (uint64_t *)hm, *h_out_off = h_in_off + np;
    uint32_t *h_in_len = (uint32_t *)(h_out_off + np), *h_out_cap = h_in_len + np, *h_out_len = h_out_cap + np,
             *h_crc = h_out_len + np;
    int32_t *h_status = (int32_t *)(h_crc + np);
    uint32_t *h_adler = (uint32_t *)(h_status + np);
    uint8_t *h_final = (uint8_t *)(h_adler + np);
    for (uint32_t i = 0; i < np; i++) {
        h_in_off[i] = meta_pad + (uint64_t)i * piece;
        const uint32_t left = in_len - (in_len ? i * piece : 0);
        h_in_len[i] = left < piece ? left : piece;
        h_out_off[i] = meta_pad + in_pad + (uint64_t)i * pcap;
        h_out_cap[i] = pcap;
        h_final[i] = (uint8_t)((i + 1 == np && final) ? 1 : 0);
    }
    hipError_t he = mz_h2d(base, hm, meta);
    if (he == hipSuccess && in_len) he = mz_h2d(base + meta_pad, in, in_len);
    if (he != hipSuccess) {
        free(hm);
        return fail("hipMemcpy (deflate input)", he);
    }
    uint64_t *d_in_off = (uint64_t *)base, *d_out_off = d_in_off + np;
    uint32_t *d_in_len = (uint32_t *)(d_out_off + np), *d_out_cap = d_in_len + np, *d_out_len = d_out_cap + np,
             *d_crc = d_out_len + np;
    int32_t *d_status = (int32_t *)(d_crc + np);
    uint32_t *d_adler = (uint32_t *)(d_status + np);
    uint8_t *d_final = (uint8_t *)(d_adler + np);
    rc = mzhip_deflate_batch_level(base, d_in_off, d_in_len, base, d_out_off, d_out_cap, d_final, np, level, window_log2, d_out_len,
                                   d_crc, d_status, MZ_HOST_STREAM);
    if (rc == 0 && hipStreamSynchronize(MZ_HOST_STREAM) != hipSuccess) rc = -104;
    /* zlib wrapper: Adler-32 of the same pieces, one wave each, combined below from the checksums alone */
    if (rc == 0 && adler) rc = mzhip_adler32_batch(base, d_in_off, d_in_len, np, d_adler, MZ_HOST_STREAM);
    if (rc == 0 && adler && hipStreamSynchronize(MZ_HOST_STREAM) != hipSuccess) rc = -104;
    if (rc == 0 && mz_d2h(hm, base, meta) != hipSuccess) rc = -104;
    uint32_t total = 0, k = 0, ad = 1;
    for (uint32_t i = 0; rc == 0 && i < np; i++) {
        if (h_status[i] != 0) rc = h_status[i];
        else if (h_out_len[i] > out_cap - total) rc = MZHIP_STATUS_OUT_FULL;
        else if (mz_d2h(out + total, base + h_out_off[i], h_out_len[i]) != hipSuccess) rc = -104;
        else {
            total += h_out_len[i];
            k = (i == 0) ? h_crc[0] : mzhip_crc32_combine_host(k, h_crc[i], h_in_len[i]); /* checksums only */
            if (adler) ad = mzhip_adler32_combine_host(ad, h_adler[i], h_in_len[i]);
        }
    }
    free(hm);
    if (out_len) *out_len = total;
    if (crc) *crc = k;
    if (adler) *adler = ad;
    return rc;
}

int32_t mzhip_deflate_host(const uint8_t *in, uint32_t in_len, uint32_t final, uint8_t *out, uint32_t out_cap,
                           uint32_t *out_len, uint32_t *crc) {
    return mzhip_deflate_host2(in, in_len, final, out, out_cap, out_len, crc, nullptr);
}

__attribute__((visibility("hidden"))) uint32_t mzhip_adler32_combine(uint32_t ad1, uint32_t ad2, uint64_t len2) {
    return mzhip_adler32_combine_host(ad1, ad2, len2);
}

// Host side of mz_crypt_crc32_update.  Buffers below MZHIP_CRC_HOST_BELOW bytes are folded right here with the
// product's own slicing-by-4 tables (the same mzhip_crc_tables the kernels use): a launch plus two PCIe round trips for
// a few bytes helps nobody, and the reference calls this symbol one byte at a time from mz_strm_pkcrypt.c:79,86.
// Larger buffers go to K2 on the device.  The symbol has no error channel (mz_crypt.h:20), so a device failure neither
// aborts the host process nor corrupts the value: the bytes are folded on the host, and the failure is latched for the
// next codec-stream call of this thread to report (mzhip_take_crc_fault, checked by the READ / WRITE shims).
namespace {
const mzhip_crc_tables *host_crc_tables() {
    static mzhip_crc_tables t;
    static std::once_flag once;
    std::call_once(once, [] { mzhip_crc_tables_init(&t); });
    return &t;
}
uint32_t crc32_fold_host(uint32_t value, const uint8_t *p, size_t n) {
    const mzhip_crc_tables *t = host_crc_tables();
    uint32_t r = ~value; /* register inverted on entry and exit, mz_crypt.c:81,90 */
    while (n && ((uintptr_t)p & 3u)) {
        r = t->byte_tab[(r ^ *p++) & 255u] ^ (r >> 8);
        n--;
    }
    for (; n >= 4; n -= 4, p += 4) {
        uint32_t d;
        memcpy(&d, p, 4);
        const uint32_t x = r ^ d; /* little-endian host */
        r = t->slice[3][x & 255u] ^ t->slice[2][(x >> 8) & 255u] ^ t->slice[1][(x >> 16) & 255u] ^ t->slice[0][x >> 24];
    }
    while (n--) r = t->byte_tab[(r ^ *p++) & 255u] ^ (r >> 8);
    return ~r;
}
thread_local int32_t g_crc_fault = 0;
} // namespace

uint32_t mzhip_crc32_host(uint32_t value, const uint8_t *buf, size_t size) {
    if (size == 0) return value;
    if (size < MZHIP_CRC_HOST_BELOW) return crc32_fold_host(value, buf, size);
    DeviceCtx *c = nullptr;
    if (ctx_for_current(&c)) {
        g_crc_fault = -1; /* MZ_STREAM_ERROR: no usable HIP device (mzhip_last_error has the reason) */
        return crc32_fold_host(value, buf, size);
    }
    // segments of 256 KiB, one wave each; segment CRCs are chained with x^(8*len) shifts
    // (32-bit arithmetic on checksums only, no byte is touched on the host).
    const uint32_t seg = 256u << 10;
    const uint32_t nseg = (uint32_t)((size + seg - 1) / seg);
    const size_t meta = (size_t)nseg * (8 + 4 + 4);
    const size_t meta_pad = (meta + 63) & ~(size_t)63;
    Staging sc;
    uint64_t *h_off = (uint64_t *)malloc(meta_pad);
    bool ok = h_off != nullptr && sc.get(c, meta_pad + size) == 0;
    uint32_t v = value;
    if (ok) {
        uint8_t *base = (uint8_t *)sc.p;
        uint32_t *h_len = (uint32_t *)(h_off + nseg);
        uint32_t *h_crc = h_len + nseg;
        for (uint32_t i = 0; i < nseg; i++) {
            h_off[i] = meta_pad + (uint64_t)i * seg;
            size_t left = size - (size_t)i * seg;
            h_len[i] = (uint32_t)(left < seg ? left : seg);
        }
        ok = mz_h2d(base, h_off, meta) == hipSuccess &&
             mz_h2d(base + meta_pad, buf, size) == hipSuccess;
        uint64_t *d_off = (uint64_t *)base;
        uint32_t *d_len = (uint32_t *)(d_off + nseg);
        uint32_t *d_crc = d_len + nseg;
        ok = ok && mzhip_crc32_batch(base, d_off, d_len, nseg, nullptr, d_crc, MZ_HOST_STREAM) == 0;
        ok = ok && mz_d2h(h_crc, d_crc, nseg * sizeof(uint32_t)) == hipSuccess;
        if (ok)
            for (uint32_t i = 0; i < nseg; i++) v = mzhip_crc32_combine_host(v, h_crc[i], h_len[i]);
    }
    free(h_off);
    if (!ok) {
        if (!g_err[0]) snprintf(g_err, sizeof(g_err), "device failure in mz_crypt_crc32_update");
        g_crc_fault = -1;
        return crc32_fold_host(value, buf, size);
    }
    return v;
}

// the failure a mz_crypt_crc32_update of this thread could not report (0 = none); reading clears it
__attribute__((visibility("hidden"))) int32_t mzhip_take_crc_fault(void) {
    const int32_t f = g_crc_fault;
    g_crc_fault = 0;
    return f;
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// "Prime" path (Batching): decode every DEFLATE entry of an archive in ONE launch and keep the result
// in a host cache, so that the reference's untouched one-entry-at-a-time loop (mz_zip_entry_read ->
// mz_stream_zlib_read -> mz_crypt_crc32_update) is served at memcpy speed.  The codec stream recognises a primed
// entry by the position of its base stream (= the payload offset) plus the first payload bytes; whatever it
// returns is still CRC-checked by mz_zip.c:2116-2128 against the central directory.  CRCs are GPU-computed:
// per entry, and per 65 535-byte segment (the reader's buffer size, mz_zip_rw.c:55) for the chunked updates.

namespace {
// Page-locked host memory is expensive to make (the pages are faulted in and pinned: ~100 ms per GiB) and a process that
// primes one archive after another needs the same two blocks again and again (the file image, the decoded bytes), so
// freed blocks are kept -- four at most, 4 GiB in all -- and handed out again when they are large enough.
struct PinnedPool {
    std::mutex mu;
    struct Blk {
        void *p = nullptr;
        size_t cap = 0;
    } blk[4];
    void *get(size_t need, size_t *cap) {
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < 4; i++)
                if (blk[i].p && blk[i].cap >= need && (best < 0 || blk[i].cap < blk[best].cap)) best = i;
            if (best >= 0 && blk[best].cap <= 2 * need + ((size_t)64 << 20)) {
                void *p = blk[best].p;
                *cap = blk[best].cap;
                blk[best] = Blk();
                return p;
            }
        }
        void *p = nullptr;
        const size_t want = (need + ((size_t)2 << 20)) & ~(((size_t)2 << 20) - 1);
        /* the pages are placed where the thread that pins them runs, and that should be next to the current device (the
         * copy engines read and write this block): a thread of the library's own binds itself there for the allocation.
         * (It used to be the caller's thread, re-bound for the length of the call: runtime helper threads spawned in that
         * window inherited the narrow mask) */
        int dev = 0;
        hipError_t he = hipGetDevice(&dev);
        if (he == hipSuccess) {
            std::thread t([&] {
                (void)mzhip_bind_thread_near_device(dev, 0);
                he = hipSetDevice(dev);
                if (he == hipSuccess) he = hipHostMalloc(&p, want, hipHostMallocDefault);
            });
            t.join();
        }
        if (he != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        *cap = want;
        return p;
    }
    void put(void *p, size_t cap) {
        if (!p) return;
        void *drop = p;
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t total = cap;
            int empty = -1, smallest = -1;
            for (int i = 0; i < 4; i++) {
                if (!blk[i].p) {
                    if (empty < 0) empty = i;
                    continue;
                }
                total += blk[i].cap;
                if (smallest < 0 || blk[i].cap < blk[smallest].cap) smallest = i;
            }
            if (total <= ((size_t)4 << 30)) {
                if (empty >= 0) {
                    blk[empty].p = p;
                    blk[empty].cap = cap;
                    drop = nullptr;
                } else if (blk[smallest].cap < cap) {
                    drop = blk[smallest].p;
                    blk[smallest].p = p;
                    blk[smallest].cap = cap;
                }
            }
        }
        if (drop) (void)hipHostFree(drop);
    }
};
PinnedPool g_pinned;
double prime_now() {
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
bool prime_trace() {
    static const bool on = getenv("MZHIP_PRIME_TRACE") != nullptr;
    return on;
}
struct PrimedEntry {
    int64_t payload_off, csize, usize, out_off;
    uint32_t crc;
    int32_t status;
    int64_t seg0; // index of this entry's first segment CRC
    int32_t method;
    int32_t head_len;
    uint8_t head[256]; // the first payload bytes: a stream must present the same ones to be served
    // the entry's Hash extra field (0x1a51, mz_zip_rw.c:1398-1408), when it names SHA-1 or SHA-256 -- the two the reader
    // verifies (mz_zip_rw.c:414-419): the digest of the decoded bytes is computed on the device in the pass that decodes
    // them and compared with the field's; an entry whose digest differs is not served from the cache
    uint16_t hash_alg, hash_size; // 0: no such field
    uint8_t hash_want[32], hash_got[32];
};
// One primed archive.  Generations are reference-counted: an open stream that is being served from one pins it, so a
// later prime / clear (another archive, another thread, MZHIP_AUTOPRIME) never frees memory a stream still reads.
struct PrimeGen {
    std::vector<PrimedEntry> entries; // sorted by payload_off
    std::vector<uint32_t> seg_crc;
    uint8_t *out = nullptr; // page-locked (hipHostMalloc, from g_pinned): the D2H copy of the decoded bytes runs at link speed
    bool out_pinned = false;
    size_t out_cap = 0;
    uint64_t zip_len = 0, ident = 0; // archive identity: length + hash of its central directory and end records
    // STORE entries: no codec stream sees them (the reference's raw stream hands the bytes to mz_crypt_crc32_update,
    // mz_zip.c:2047-2049), so the CRC symbol recognises a chunk by content: the payloads are kept, cut into the reader's
    // 65 535-byte chunks with their device-computed CRCs, indexed by a fingerprint of each chunk
    struct StoreSeg {
        uint64_t off; // into store[]
        uint32_t len, crc;
    };
    std::vector<StoreSeg> store_segs;
    std::unordered_multimap<uint64_t, uint32_t> store_idx; // fingerprint -> segment
    uint8_t *store = nullptr;
    bool store_pinned = false;
    uint64_t store_entries = 0;
    // A generation is published BEFORE its entries are decoded (a reader thread that asks for an entry whose chunk of
    // the decode pipeline has not landed yet waits for exactly that chunk, not for the whole archive): state[i] says
    // whether entries[i] may be served.  Entries are immutable but for `crc`, which is written before state turns 1;
    // the STORE index is built last and looked at only once store_ready is set.
    std::unique_ptr<std::atomic<uint8_t>[]> state; // 0 pending, 1 servable, 2 not served (did not decode to its declared sizes)
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<bool> store_ready{false};
    ~PrimeGen() {
        if (out_pinned) g_pinned.put(out, out_cap);
        else free(out);
        if (store_pinned) (void)hipHostFree(store);
        else free(store);
    }
};
struct PrimeCache {
    std::vector<std::shared_ptr<PrimeGen>> gens; // newest first, at most kMaxGens
};
PrimeCache g_prime;
// readers (one lookup per entry that is opened, from as many host threads as the application has) share the lock;
// publishing, replacing and clearing a generation take it exclusively
std::shared_mutex g_prime_mu;
std::atomic<uint64_t> g_prime_hits{0}, g_prime_misses{0};
std::atomic<uint64_t> g_prime_hash_checked{0}, g_prime_hash_bad{0}; // entries whose Hash field was verified on the device / differed
std::atomic<uint64_t> g_prime_wait_ns{0}, g_prime_wait_n{0}; // MZHIP_PRIME_TRACE: lookups that had to wait for their chunk
std::atomic<int> g_any_gens{0};   // generations present at all: the streams' "is there anything to look up" (mzhip_prime_any)
std::atomic<int> g_store_gens{0}; // generations that hold STORE chunks: the CRC symbol's fast "nothing to look up"
constexpr uint32_t kSeg = 65535u;
constexpr size_t kMaxGens = 8;

uint64_t fnv1a64(const uint8_t *p, uint64_t n, uint64_t h = 0xCBF29CE484222325ull) {
    for (uint64_t i = 0; i < n; i++) h = (h ^ p[i]) * 0x100000001B3ull;
    return h;
}
// fingerprint of a chunk of at least 32 bytes: its length, first and last 16 bytes (a hint only: a hit is confirmed
// byte by byte)
uint64_t store_key(const uint8_t *p, uint32_t n) {
    uint64_t h = fnv1a64((const uint8_t *)&n, 4);
    h = fnv1a64(p, 16, h);
    return fnv1a64(p + n - 16, 16, h);
}
void count_store_gens_locked() {
    int k = 0;
    for (const auto &g : g_prime.gens) k += (g->store_ready.load() && !g->store_segs.empty()) ? 1 : 0;
    g_store_gens.store(k);
    g_any_gens.store((int)g_prime.gens.size());
}
} // namespace

extern "C" {

void mzhip_prime_clear(void) {
    (void)mzhip_prime_wait(); /* a prime that is still running reads the caller's image: it is finished, then dropped */
    std::unique_lock<std::shared_mutex> lk(g_prime_mu);
    g_prime = PrimeCache(); // generations pinned by open streams live until those streams let go
    g_prime_hits.store(0);
    g_prime_misses.store(0);
    g_store_gens.store(0);
    g_any_gens.store(0);
}

} // extern "C"

namespace {
// One slice of the primed entries, decoded on the CURRENT device of the calling thread, as a PIPELINE: the slice is
// cut into chunks of about kPrimeChunk decoded bytes, and chunk i's H2D copy (the byte range of the archive that holds
// its payloads), its launches (one per codec + the segment CRCs) and its D2H copies (results and every decoded byte)
// are queued on stream i % 3, so that H2D(i + 1), kernel(i) and D2H(i - 1) run at the same time and the PCIe link is
// busy in both directions while the kernels run (round 2: blocking copies and a device-wide synchronisation around one
// launch per codec -- 31 GB/s over the link).  Only the stream that is about to be reused is waited
// for, never the device.  ents[lo..hi) are in archive order; results land in the shared per-entry arrays.
constexpr uint64_t kPrimeChunk = 48ull << 20;
constexpr int kPrimeLanes = 3;
struct PrimeLane {
    hipStream_t s = nullptr;
    Scratch d_zip, d_out, d_meta;
    size_t zip_cap = 0, out_cap = 0, meta_cap = 0;
    uint8_t *h_meta = nullptr; // page-locked staging: launch arrays up, results down
    size_t h_cap = 0;
    // the chunk in flight on this lane
    bool busy = false;
    size_t lo = 0, hi = 0;
    uint32_t k = 0, ns = 0;
    int64_t seg0 = 0;
    std::vector<uint32_t> order;
    size_t res_off = 0; // where the result arrays start inside h_meta
    std::vector<size_t> hash_ent; // entries of the chunk whose digests were asked for, in launch order ...
    size_t hash_off = 0;          // ... and where their 32-byte digests land inside h_meta
    ~PrimeLane() {
        if (s) (void)hipStreamDestroy(s);
        if (h_meta) (void)hipHostFree(h_meta);
    }
};
struct PrimeLaneSet {
    PrimeLane lanes[kPrimeLanes];
};
std::mutex g_lane_mu;
PrimeLaneSet *g_lane_sets[kMaxDevices]; // never destroyed at exit: the runtime may be gone before static destructors run
int32_t prime_lane_reserve(void **p, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*p) HIP_TRY(hipFree(*p)); // (the lane's stream has been waited for: nothing uses the buffer)
    *p = nullptr;
    *cap = 0;
    const size_t want = (need + (need >> 2) + ((size_t)1 << 20)) & ~(((size_t)1 << 20) - 1);
    HIP_TRY(hipMalloc(p, want));
    *cap = want;
    return 0;
}
// results of the chunk that ran on lane L: wait for its stream (only this one); the entries that decoded cleanly, to
// their declared sizes, become servable (everything else goes through the ordinary per-entry path and its exact error
// behaviour) and the readers that wait for them are woken
int32_t prime_lane_collect(PrimeLane &L, PrimeGen *gen) {
    if (!L.busy) return 0;
    HIP_TRY(hipStreamSynchronize(L.s));
    const uint32_t k = L.k;
    const uint32_t *h_len = (const uint32_t *)(L.h_meta + L.res_off), *h_used = h_len + k, *h_crc = h_used + k;
    const int32_t *h_st = (const int32_t *)(h_crc + k);
    const uint32_t *h_seg = (const uint32_t *)(h_st + k);
    if (L.ns) memcpy(gen->seg_crc.data() + L.seg0, h_seg, (size_t)L.ns * 4);
    for (uint32_t i = 0; i < k; i++) {
        const size_t g = L.lo + L.order[i];
        PrimedEntry &e = gen->entries[g];
        const bool good = h_st[i] == 0 && h_len[i] == (uint32_t)e.usize && h_used[i] == (uint32_t)e.csize;
        e.crc = h_crc[i];
        e.status = good ? 0 : (h_st[i] ? h_st[i] : -3);
        if (!e.hash_alg) gen->state[g].store(good ? 1 : 2, std::memory_order_release);
    }
    /* entries with a SHA-1 / SHA-256 Hash field: servable only if the digest of what was decoded is the field's (the
     * reference compares the field's digest_size bytes, mz_zip_rw.c:446-449) */
    for (size_t j = 0; j < L.hash_ent.size(); j++) {
        PrimedEntry &e = gen->entries[L.hash_ent[j]];
        memcpy(e.hash_got, L.h_meta + L.hash_off + 32 * j, 32);
        const uint32_t full = e.hash_alg == 20 ? 20u : 32u;
        const uint32_t cmp = e.hash_size < full ? e.hash_size : full;
        const bool same = e.status == 0 && e.hash_size <= 64 && memcmp(e.hash_got, e.hash_want, cmp) == 0;
        if (e.status == 0) {
            g_prime_hash_checked.fetch_add(1, std::memory_order_relaxed);
            if (!same) g_prime_hash_bad.fetch_add(1, std::memory_order_relaxed);
        }
        gen->state[L.hash_ent[j]].store(same ? 1 : 2, std::memory_order_release);
    }
    { std::lock_guard<std::mutex> lk(gen->mu); } /* (a waiter is either before its check or inside wait()) */
    gen->cv.notify_all();
    {
        static const bool each = getenv("MZHIP_PRIME_TRACE") && atoi(getenv("MZHIP_PRIME_TRACE")) >= 2;
        if (each) fprintf(stderr, "[mzhip prime] entries %zu..%zu servable at %.2f ms\n", L.lo, L.hi, prime_now() * 1e3);
    }
    L.busy = false;
    return 0;
}
int32_t prime_slice(const uint8_t *zip, PrimeGen *gen, const std::vector<int64_t> &max_out_all, size_t lo, size_t hi) {
    std::vector<PrimedEntry> &ents = gen->entries;
    uint8_t *const h_out = gen->out;
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    if (hi <= lo) return 0;
    /* streams, device buffers and page-locked staging of the three lanes are kept between calls (one set per device
     * is parked; a second prime on the same device at the same time makes its own): hipMalloc / hipFree / hipHostMalloc
     * of ~200 MB per call cost more than the pipeline itself on a 1 GiB archive (44 ms -> see profiles/r3) */
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::unique_ptr<PrimeLaneSet> set;
    {
        std::lock_guard<std::mutex> lk(g_lane_mu);
        if (dev >= 0 && dev < kMaxDevices && g_lane_sets[dev]) {
            set.reset(g_lane_sets[dev]);
            g_lane_sets[dev] = nullptr;
        }
    }
    if (!set) {
        set.reset(new PrimeLaneSet());
        for (auto &L : set->lanes) HIP_TRY(hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
    }
    struct Park { /* back to the shelf on every way out (the lanes are idle by then: collect() waited for each stream, or
                     the set is dropped when a call failed in the middle) */
        std::unique_ptr<PrimeLaneSet> &set;
        int dev;
        bool ok = false;
        ~Park() {
            if (!ok || !set) return;
            std::lock_guard<std::mutex> lk(g_lane_mu);
            if (dev >= 0 && dev < kMaxDevices && !g_lane_sets[dev]) g_lane_sets[dev] = set.release();
        }
    } park{set, dev};
    PrimeLane *lanes = set->lanes;
    int turn = 0;
    for (size_t c0 = lo; c0 < hi;) {
        /* the next chunk: entries [c0, c1), about kPrimeChunk decoded bytes (one entry at least) */
        size_t c1 = c0;
        uint64_t acc = 0;
        while (c1 < hi && (c1 == c0 || acc + (uint64_t)ents[c1].usize + (uint64_t)ents[c1].csize <= kPrimeChunk)) {
            acc += (uint64_t)ents[c1].usize + (uint64_t)ents[c1].csize;
            c1++;
        }
        PrimeLane &L = lanes[turn];
        turn = (turn + 1) % kPrimeLanes;
        rc = prime_lane_collect(L, gen); /* the chunk this lane ran three chunks ago */
        if (rc) return rc;
        const uint32_t k = (uint32_t)(c1 - c0);
        uint64_t zlo = UINT64_MAX, zhi = 0;
        for (size_t i = c0; i < c1; i++) {
            zlo = std::min(zlo, (uint64_t)ents[i].payload_off);
            zhi = std::max(zhi, (uint64_t)(ents[i].payload_off + ents[i].csize));
        }
        const int64_t out_base = ents[c0].out_off;
        const uint64_t out_bytes = (uint64_t)(ents[c1 - 1].out_off - out_base) + (((uint64_t)ents[c1 - 1].usize + 15) & ~15ull);
        uint32_t ns = 0;
        for (size_t i = c0; i < c1; i++) ns += (uint32_t)((ents[i].usize + kSeg - 1) / kSeg);
        /* launch arrays (grouped by method: one batch launch per codec) and result arrays, one page-locked block:
         *   up:   in_off[k] out_off[k] seg_off[ns] max_out[k] (8 bytes each)  in_len[k] out_cap[k] seg_len[ns] (4 bytes each)
         *   down: out_len[k] in_used[k] crc[k] status[k] seg_crc[ns] */
        const size_t up = (size_t)k * 24 + (size_t)ns * 8 + (size_t)k * 8 + (size_t)ns * 4;
        const size_t down = (size_t)k * 16 + (size_t)ns * 4;
        const size_t up_al = (up + 255) & ~(size_t)255;
        /* ... and for the entries with a SHA-1 / SHA-256 Hash field: off[kh] (8 bytes), len[kh] (4) up, digest[kh] (32) down */
        L.hash_ent.clear();
        for (int pass = 0; pass < 2; pass++)
            for (size_t i = c0; i < c1; i++)
                if (ents[i].hash_alg == (pass == 0 ? 20 : 23)) L.hash_ent.push_back(i); /* SHA-1 first, then SHA-256: one launch each */
        const size_t kh = L.hash_ent.size();
        const size_t down_al = (down + 255) & ~(size_t)255;
        const size_t hup = kh * 12, hup_al = (hup + 255) & ~(size_t)255, hdown = kh * 32;
        const size_t meta_all = up_al + down_al + hup_al + hdown;
        if (meta_all > L.h_cap) {
            if (L.h_meta) HIP_TRY(hipHostFree(L.h_meta));
            L.h_meta = nullptr;
            L.h_cap = 0;
            const size_t want = (meta_all * 2 + 4095) & ~(size_t)4095;
            HIP_TRY(hipHostMalloc((void **)&L.h_meta, want, hipHostMallocDefault));
            L.h_cap = want;
        }
        rc = prime_lane_reserve(&L.d_zip.p, &L.zip_cap, (size_t)(zhi - zlo) + 16);
        if (!rc) rc = prime_lane_reserve(&L.d_out.p, &L.out_cap, (size_t)out_bytes + 16);
        if (!rc) rc = prime_lane_reserve(&L.d_meta.p, &L.meta_cap, meta_all + 256);
        if (rc) return rc;
        L.order.resize(k);
        for (uint32_t i = 0; i < k; i++) L.order[i] = i;
        std::stable_sort(L.order.begin(), L.order.end(), [&](uint32_t a, uint32_t b) { return ents[c0 + a].method < ents[c0 + b].method; });
        uint64_t *h_in_off = (uint64_t *)L.h_meta, *h_out_off = h_in_off + k, *h_seg_off = h_out_off + k;
        int64_t *h_max_out = (int64_t *)(h_seg_off + ns);
        uint32_t *h_in_len = (uint32_t *)(h_max_out + k), *h_out_cap = h_in_len + k, *h_seg_len = h_out_cap + k;
        for (uint32_t i = 0; i < k; i++) {
            const PrimedEntry &e = ents[c0 + L.order[i]];
            h_in_off[i] = (uint64_t)e.payload_off - zlo;
            h_in_len[i] = (uint32_t)e.csize;
            h_out_off[i] = (uint64_t)(e.out_off - out_base);
            h_out_cap[i] = (uint32_t)e.usize;
            h_max_out[i] = max_out_all[c0 + L.order[i]];
        }
        { // segments for the chunked CRC updates, in archive order (seg0 was assigned by the caller)
            uint32_t j = 0;
            for (size_t i = c0; i < c1; i++)
                for (int64_t o = 0; o < ents[i].usize; o += kSeg) {
                    h_seg_off[j] = (uint64_t)(ents[i].out_off - out_base) + (uint64_t)o;
                    h_seg_len[j] = (uint32_t)(ents[i].usize - o < kSeg ? ents[i].usize - o : kSeg);
                    j++;
                }
        }
        uint8_t *m = (uint8_t *)L.d_meta.p;
        uint64_t *d_in_off = (uint64_t *)m, *d_out_off = d_in_off + k, *d_seg_off = d_out_off + k;
        int64_t *d_max_out = (int64_t *)(d_seg_off + ns);
        uint32_t *d_in_len = (uint32_t *)(d_max_out + k), *d_out_cap = d_in_len + k, *d_seg_len = d_out_cap + k;
        uint32_t *d_out_len = (uint32_t *)(m + up_al), *d_in_used = d_out_len + k, *d_crc = d_in_used + k;
        int32_t *d_status = (int32_t *)(d_crc + k);
        uint32_t *d_seg_crc = (uint32_t *)(d_status + k);
        HIP_TRY(hipMemcpyAsync(L.d_zip.p, zip + zlo, zhi - zlo, hipMemcpyHostToDevice, L.s));
        HIP_TRY(hipMemcpyAsync(m, L.h_meta, up, hipMemcpyHostToDevice, L.s));
        for (uint32_t g0 = 0; g0 < k;) {
            uint32_t g1 = g0;
            const int32_t method = ents[c0 + L.order[g0]].method;
            while (g1 < k && ents[c0 + L.order[g1]].method == method) g1++;
            const uint32_t gn = g1 - g0;
            if (method == 8)
                rc = mzhip_inflate_batch(L.d_zip.p, d_in_off + g0, d_in_len + g0, L.d_out.p, d_out_off + g0, d_out_cap + g0, gn,
                                         d_out_len + g0, d_in_used + g0, d_crc + g0, d_status + g0, L.s);
            else
                rc = lzma_family_batch(method == 95, L.d_zip.p, d_in_off + g0, d_in_len + g0, L.d_out.p, d_out_off + g0,
                                       d_out_cap + g0, d_max_out + g0, gn, d_out_len + g0, d_in_used + g0, d_crc + g0,
                                       d_status + g0, L.s);
            if (rc) return rc;
            g0 = g1;
        }
        if (ns) {
            rc = mzhip_crc32_batch(L.d_out.p, d_seg_off, d_seg_len, ns, nullptr, d_seg_crc, L.s);
            if (rc) return rc;
        }
        if (kh) {
            /* the digests of the decoded bytes, while they are in HBM (what mz_zip_reader_entry_read hashes 65 535 bytes at a
             * time on the host, mz_zip_rw.c:465-466): one lane per entry, one launch per algorithm */
            uint64_t *h_hoff = (uint64_t *)(L.h_meta + up_al + down_al);
            uint32_t *h_hlen = (uint32_t *)(h_hoff + kh);
            size_t n1 = 0;
            for (size_t j = 0; j < kh; j++) {
                const PrimedEntry &e = ents[L.hash_ent[j]];
                h_hoff[j] = (uint64_t)(e.out_off - out_base);
                h_hlen[j] = (uint32_t)e.usize;
                n1 += e.hash_alg == 20 ? 1 : 0;
            }
            uint64_t *d_hoff = (uint64_t *)(m + up_al + down_al);
            uint32_t *d_hlen = (uint32_t *)(d_hoff + kh);
            uint8_t *d_dig = m + up_al + down_al + hup_al;
            HIP_TRY(hipMemcpyAsync(d_hoff, h_hoff, hup, hipMemcpyHostToDevice, L.s));
            if (n1) rc = mzhip_sha_batch(L.d_out.p, d_hoff, d_hlen, (uint32_t)n1, 20, d_dig, L.s);
            if (!rc && kh > n1) rc = mzhip_sha_batch(L.d_out.p, d_hoff + n1, d_hlen + n1, (uint32_t)(kh - n1), 23, d_dig + 32 * n1, L.s);
            if (rc) return rc;
            HIP_TRY(hipMemcpyAsync(L.h_meta + up_al + down_al + hup_al, d_dig, hdown, hipMemcpyDeviceToHost, L.s));
            L.hash_off = up_al + down_al + hup_al;
        }
        HIP_TRY(hipMemcpyAsync(L.h_meta + up_al, m + up_al, down, hipMemcpyDeviceToHost, L.s));
        HIP_TRY(hipMemcpyAsync(h_out + out_base, L.d_out.p, out_bytes, hipMemcpyDeviceToHost, L.s));
        L.busy = true;
        L.lo = c0;
        L.hi = c1;
        L.k = k;
        L.ns = ns;
        L.seg0 = ents[c0].seg0;
        L.res_off = up_al;
        c0 = c1;
    }
    for (int i = 0; i < kPrimeLanes; i++) { /* in the order they were started */
        rc = prime_lane_collect(lanes[(turn + i) % kPrimeLanes], gen);
        if (rc) return rc;
    }
    park.ok = true;
    return 0;
}
// STORE entries of a primed archive: their payloads are copied into the generation, cut into the reader's chunks
// (kSeg bytes from the start of each entry), and every chunk's CRC-32 is computed on the current device, one launch
// per <= 1 GiB of payload.
int32_t prime_store(const uint8_t *zip, const std::vector<std::pair<int64_t, int64_t>> &stores, PrimeGen *gen) {
    uint64_t total = 0;
    for (const auto &e : stores) total += ((uint64_t)e.second + 15) & ~15ull;
    if (!total) return 0;
    gen->store_pinned = hipHostMalloc((void **)&gen->store, total + 16, hipHostMallocDefault) == hipSuccess;
    if (!gen->store_pinned) {
        (void)hipGetLastError();
        gen->store = (uint8_t *)malloc(total + 16);
    }
    if (!gen->store) return -4;
    uint64_t pos = 0;
    for (const auto &e : stores) {
        memcpy(gen->store + pos, zip + e.first, (size_t)e.second);
        for (int64_t o = 0; o < e.second; o += kSeg)
            gen->store_segs.push_back({pos + (uint64_t)o, (uint32_t)(e.second - o < kSeg ? e.second - o : kSeg), 0u});
        pos += ((uint64_t)e.second + 15) & ~15ull;
    }
    const size_t ns = gen->store_segs.size();
    constexpr uint64_t kGroup = 1ull << 30;
    for (size_t s0 = 0; s0 < ns;) {
        size_t s1 = s0;
        const uint64_t base = gen->store_segs[s0].off;
        while (s1 < ns && gen->store_segs[s1].off + gen->store_segs[s1].len - base <= kGroup) s1++;
        const uint32_t gn = (uint32_t)(s1 - s0);
        const uint64_t bytes = gen->store_segs[s1 - 1].off + gen->store_segs[s1 - 1].len - base;
        std::vector<uint64_t> off(gn);
        std::vector<uint32_t> len(gn), crc(gn);
        for (uint32_t i = 0; i < gn; i++) {
            off[i] = gen->store_segs[s0 + i].off - base;
            len[i] = gen->store_segs[s0 + i].len;
        }
        Scratch d_buf, d_meta;
        HIP_TRY(hipMalloc(&d_buf.p, bytes + 16));
        HIP_TRY(hipMalloc(&d_meta.p, (size_t)gn * 16 + 64));
        uint64_t *d_off = (uint64_t *)d_meta.p;
        uint32_t *d_len = (uint32_t *)(d_off + gn), *d_crc = d_len + gn;
        HIP_TRY(hipMemcpy(d_buf.p, gen->store + base, bytes, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_off, off.data(), (size_t)gn * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_len, len.data(), (size_t)gn * 4, hipMemcpyHostToDevice));
        const int32_t rc = mzhip_crc32_batch(d_buf.p, d_off, d_len, gn, nullptr, d_crc, nullptr);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr)); /* the launches above are on the null stream */
        HIP_TRY(hipMemcpy(crc.data(), d_crc, (size_t)gn * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < gn; i++) gen->store_segs[s0 + i].crc = crc[i];
        s0 = s1;
    }
    for (size_t i = 0; i < ns; i++) {
        const PrimeGen::StoreSeg &sg = gen->store_segs[i];
        if (sg.len >= 32u) gen->store_idx.emplace(store_key(gen->store + sg.off, sg.len), (uint32_t)i);
    }
    gen->store_entries = stores.size();
    return 0;
}
} // namespace

extern "C" {

void mzhip_shard_bounds(const int64_t *table, int64_t n, int32_t world, int64_t *bounds) {
    /* contiguous slices balanced by compressed + uncompressed bytes (+ 64 per entry): the rule of archive.shard_bounds */
    if (world < 1) world = 1;
    double total = 0;
    for (int64_t i = 0; i < n; i++) total += (double)(table[i * 8 + 3] + table[i * 8 + 4] + 64);
    bounds[0] = 0;
    double cum = 0;
    int64_t i = 0;
    for (int32_t r = 1; r < world; r++) {
        const double target = total * r / world;
        while (i < n && cum < target) {
            cum += (double)(table[i * 8 + 3] + table[i * 8 + 4] + 64);
            i++;
        }
        /* numpy.searchsorted(cum, target, 'left') over the cumulative sums that start with 0: first index with cum >= target */
        bounds[r] = i;
    }
    bounds[world] = n;
}

} // extern "C"

namespace {
// One prime: prepared and published by the calling thread (index, entry table, page-locked output buffer), run either by
// the same thread (mzhip_prime_mem / _file: returns when every entry is decoded) or by a worker thread
// (mzhip_prime_mem_begin: returns at once, readers are served chunk by chunk as the pipeline delivers).
struct PrimeJob {
    const uint8_t *zip = nullptr;
    uint64_t zip_len = 0;
    int cur = 0; // the device of the thread that asked
    std::vector<int32_t> devs;
    std::shared_ptr<PrimeGen> gen;
    std::vector<int64_t> max_out, wtab;
    std::vector<std::pair<int64_t, int64_t>> stores; /* STORE entries: payload offset, size */
    int64_t result = 0; // entries primed, or the error
    std::string err;
    double t_start = 0;
};
std::mutex g_worker_mu;
// (on the heap and never destroyed: a process that exits with a prime still running must not meet std::terminate in the
// destructor of a joinable thread -- and the HIP runtime may be gone by the time static destructors run)
auto &g_prime_workers = *new std::vector<std::pair<std::thread, std::shared_ptr<PrimeJob>>>();

void prime_unpublish(const std::shared_ptr<PrimeGen> &gen) {
    std::unique_lock<std::shared_mutex> lk(g_prime_mu);
    auto &gens = g_prime.gens;
    for (size_t i = 0; i < gens.size();) {
        if (gens[i] == gen) gens.erase(gens.begin() + (long)i);
        else i++;
    }
    count_store_gens_locked();
}

// index the archive, lay out the generation and publish it with every entry pending.  0 = nothing to prime.
int64_t prime_prepare(const uint8_t *zip, uint64_t zip_len, const int32_t *devices, int32_t ndev, std::shared_ptr<PrimeJob> *out) {
    auto job = std::make_shared<PrimeJob>();
    job->zip = zip;
    job->zip_len = zip_len;
    job->t_start = prime_now();
    HIP_TRY(hipGetDevice(&job->cur));
    if (ndev <= 0) {
        const int32_t nd = mzhip_device_count();
        if (nd <= 0) return -104;
        for (int32_t d = 0; d < nd; d++) job->devs.push_back(d);
    } else {
        for (int32_t i = 0; i < ndev; i++) job->devs.push_back(devices ? devices[i] : i);
    }
    int64_t n = mzhip_zip_index_mem(zip, zip_len, nullptr, 0);
    if (n <= 0) return n;
    std::vector<int64_t> table((size_t)n * 8);
    mzhip_zip_index_mem(zip, zip_len, table.data(), n);
    std::vector<int64_t> rows; /* the entries a codec kernel can take, in the order their payloads lie in the file */
    for (int64_t i = 0; i < n; i++) {
        const int64_t *t = &table[(size_t)i * 8];
        if ((t[1] & 1) || t[7] < 0 || t[3] < 0 || t[4] < 0 || t[3] >= (1ll << 31) || t[4] >= (1ll << 31) ||
            (uint64_t)t[7] > zip_len || (uint64_t)t[3] > zip_len - (uint64_t)t[7])
            continue;
        if (t[0] == 0 && t[3] == t[4] && t[4] >= (int64_t)MZHIP_CRC_HOST_BELOW) job->stores.emplace_back(t[7], t[4]);
        if (t[0] != 8 && t[0] != 14 && t[0] != 95) continue;
        rows.push_back(i);
    }
    std::stable_sort(rows.begin(), rows.end(), [&](int64_t a, int64_t b) { return table[(size_t)a * 8 + 7] < table[(size_t)b * 8 + 7]; });
    auto gen = std::make_shared<PrimeGen>();
    std::vector<PrimedEntry> &ents = gen->entries;
    ents.reserve(rows.size());
    std::vector<uint16_t> h_alg((size_t)n), h_dsz((size_t)n);
    std::vector<uint8_t> h_dig((size_t)n * 64);
    if (mzhip_zip_index_hash_mem(zip, zip_len, table.data(), n, h_alg.data(), h_dsz.data(), h_dig.data()) < 0)
        std::fill(h_alg.begin(), h_alg.end(), (uint16_t)0);
    uint64_t total_out = 0;
    int64_t nseg = 0;
    for (const int64_t i : rows) {
        const int64_t *t = &table[(size_t)i * 8];
        PrimedEntry e;
        memset(&e, 0, sizeof(e));
        e.method = (int32_t)t[0];
        e.payload_off = t[7];
        e.csize = t[3];
        e.usize = t[4];
        e.out_off = (int64_t)total_out;
        e.seg0 = nseg;
        e.status = -1;
        e.head_len = (int32_t)(t[3] < (int64_t)sizeof(e.head) ? t[3] : (int64_t)sizeof(e.head));
        memcpy(e.head, zip + t[7], (size_t)e.head_len);
        if (h_alg[(size_t)i] == 20 || h_alg[(size_t)i] == 23) { /* SHA-1 / SHA-256: what mz_zip_reader_entry_open sets up (mz_zip_rw.c:414-419) */
            e.hash_alg = h_alg[(size_t)i];
            e.hash_size = h_dsz[(size_t)i];
            memcpy(e.hash_want, &h_dig[(size_t)i * 64], 32);
        }
        ents.push_back(e);
        /* TOTAL_OUT_MAX as mz_zip.c:1833-1846 sets it: the uncompressed size when the EOS flag is set */
        job->max_out.push_back((t[0] != 8 && (t[1] & 2)) ? t[4] : -1);
        job->wtab.insert(job->wtab.end(), t, t + 8);
        total_out += ((uint64_t)t[4] + 15) & ~15ull;
        nseg += (t[4] + kSeg - 1) / kSeg;
    }
    const size_t k = ents.size();
    if (k == 0 && job->stores.empty()) return 0;
    const double t_idx = prime_now();
    gen->out = (uint8_t *)g_pinned.get(total_out + 16, &gen->out_cap);
    gen->out_pinned = gen->out != nullptr;
    if (!gen->out_pinned) gen->out = (uint8_t *)malloc(total_out + 16);
    if (!gen->out) return -4;
    if (prime_trace())
        fprintf(stderr, "[mzhip prime] %zu entries, %.1f MiB to decode: index %.1f ms, pinned output %.1f ms\n", k,
                (double)total_out / 1048576.0, (t_idx - job->t_start) * 1e3, (prime_now() - t_idx) * 1e3);
    gen->seg_crc.assign((size_t)nseg, 0u);
    gen->state.reset(new std::atomic<uint8_t>[k ? k : 1]);
    for (size_t i = 0; i < k; i++) gen->state[i].store(0, std::memory_order_relaxed);
    gen->zip_len = zip_len;
    {
        /* identity = length + hash of everything from the first central-directory record to the end of the file */
        const uint64_t cd0 = (uint64_t)table[6];
        gen->ident = fnv1a64(zip + cd0, zip_len - cd0);
    }
    job->gen = gen;
    {
        std::unique_lock<std::shared_mutex> lk(g_prime_mu);
        auto &gens = g_prime.gens;
        for (size_t i = 0; i < gens.size();) { /* a re-prime of the same archive replaces its generation */
            if (gens[i]->zip_len == gen->zip_len && gens[i]->ident == gen->ident) gens.erase(gens.begin() + (long)i);
            else i++;
        }
        gens.insert(gens.begin(), gen);
        if (gens.size() > kMaxGens) gens.resize(kMaxGens);
        count_store_gens_locked();
    }
    *out = job;
    return 1;
}

// decode everything the job laid out; every entry leaves the pending state on every way out
void prime_run(const std::shared_ptr<PrimeJob> &job) {
    PrimeGen *gen = job->gen.get();
    const size_t k = gen->entries.size();
    struct Settle {
        PrimeGen *gen;
        size_t k;
        ~Settle() {
            for (size_t i = 0; i < k; i++) {
                uint8_t z = 0;
                (void)gen->state[i].compare_exchange_strong(z, 2);
            }
            { std::lock_guard<std::mutex> lk(gen->mu); }
            gen->cv.notify_all();
        }
    } settle{gen, k};
    const double t0 = prime_now();
    const int32_t world = (int32_t)std::min<size_t>(job->devs.size(), std::max<size_t>(k, 1));
    std::vector<int64_t> bounds((size_t)world + 1);
    mzhip_shard_bounds(job->wtab.data(), (int64_t)k, world, bounds.data());
    std::vector<int32_t> rcs((size_t)world, 0);
    std::vector<std::string> errs((size_t)world);
    auto work = [&](int32_t r) {
        hipError_t he = hipSetDevice(job->devs[(size_t)r]);
        if (he != hipSuccess) rcs[(size_t)r] = fail("hipSetDevice", he);
        else rcs[(size_t)r] = prime_slice(job->zip, gen, job->max_out, (size_t)bounds[(size_t)r], (size_t)bounds[(size_t)r + 1]);
        if (rcs[(size_t)r]) errs[(size_t)r] = g_err;
    };
    int now = -1;
    (void)hipGetDevice(&now);
    if (world == 1 && job->devs[0] == now) {
        work(0);
    } else { /* one host thread per device: the host side of the sharded path is C */
        std::vector<std::thread> th;
        for (int32_t r = 0; r < world; r++) th.emplace_back(work, r);
        for (auto &t : th) t.join();
        (void)hipSetDevice(job->cur);
    }
    for (int32_t r = 0; r < world; r++)
        if (rcs[(size_t)r]) {
            job->err = errs[(size_t)r];
            job->result = rcs[(size_t)r];
            prime_unpublish(job->gen);
            return;
        }
    if (prime_trace()) fprintf(stderr, "[mzhip prime] decode pipeline %.1f ms\n", (prime_now() - t0) * 1e3);
    if (!job->stores.empty()) {
        const int32_t src = prime_store(job->zip, job->stores, gen); /* on this thread's device */
        if (src) {
            job->err = g_err;
            job->result = src;
            prime_unpublish(job->gen);
            return;
        }
        gen->store_ready.store(true, std::memory_order_release);
        std::unique_lock<std::shared_mutex> lk(g_prime_mu);
        count_store_gens_locked();
    }
    int64_t good = 0;
    for (size_t i = 0; i < k; i++) good += gen->state[i].load(std::memory_order_relaxed) == 1 ? 1 : 0;
    job->result = good + (int64_t)gen->store_entries;
}

int64_t prime_wait_all() {
    std::vector<std::pair<std::thread, std::shared_ptr<PrimeJob>>> w;
    {
        std::lock_guard<std::mutex> lk(g_worker_mu);
        w.swap(g_prime_workers);
    }
    int64_t total = 0, bad = 0;
    for (auto &p : w) {
        if (p.first.joinable()) p.first.join();
        if (p.second->result < 0) {
            if (!bad) {
                bad = p.second->result;
                snprintf(g_err, sizeof(g_err), "%s", p.second->err.c_str());
            }
        } else {
            total += p.second->result;
        }
    }
    return bad ? bad : total;
}
} // namespace

extern "C" {

int64_t mzhip_prime_mem_multi(const uint8_t *zip, uint64_t zip_len, const int32_t *devices, int32_t ndev) {
    std::shared_ptr<PrimeJob> job;
    const int64_t rc = prime_prepare(zip, zip_len, devices, ndev, &job);
    if (rc <= 0) return rc;
    prime_run(job);
    if (job->result < 0) snprintf(g_err, sizeof(g_err), "%s", job->err.c_str());
    return job->result;
}

int64_t mzhip_prime_mem(const uint8_t *zip, uint64_t zip_len) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    const int32_t d = (int32_t)cur;
    return mzhip_prime_mem_multi(zip, zip_len, &d, 1);
}

int64_t mzhip_prime_mem_begin(const uint8_t *zip, uint64_t zip_len) {
    int cur = 0;
    HIP_TRY(hipGetDevice(&cur));
    const int32_t d = (int32_t)cur;
    std::shared_ptr<PrimeJob> job;
    const int64_t rc = prime_prepare(zip, zip_len, &d, 1, &job);
    if (rc <= 0) return rc;
    const int64_t n = (int64_t)job->gen->entries.size() + (int64_t)job->stores.size();
    try {
        std::lock_guard<std::mutex> lk(g_worker_mu);
        g_prime_workers.emplace_back(std::thread([job] {
                                         (void)hipSetDevice(job->cur);
                                         (void)mzhip_bind_thread_near_device(job->cur, 0); /* this thread feeds the copy engines: next to the device */
                                         prime_run(job);
                                     }),
                                     job);
    } catch (...) { /* no thread to be had: the entries must not stay pending -- decode here and now */
        prime_run(job);
        if (job->result < 0) {
            snprintf(g_err, sizeof(g_err), "%s", job->err.c_str());
            return job->result;
        }
    }
    return n;
}

int64_t mzhip_prime_wait(void) {
    const int64_t r = prime_wait_all();
    if (prime_trace() && g_prime_wait_n.load())
        fprintf(stderr, "[mzhip prime] %llu lookups waited for their chunk, %.1f ms in all\n", (unsigned long long)g_prime_wait_n.exchange(0),
                (double)g_prime_wait_ns.exchange(0) * 1e-6);
    return r;
}

static int64_t prime_file_on(const char *path, const int32_t *devices, int32_t ndev, int multi) {
    FILE *f = fopen(path, "rb");
    if (!f) return -111; /* MZ_OPEN_ERROR */
    fseeko(f, 0, SEEK_END);
    const int64_t len = (int64_t)ftello(f);
    fseeko(f, 0, SEEK_SET);
    const double t0 = prime_now();
    size_t cap = 0;
    uint8_t *buf = (uint8_t *)g_pinned.get((size_t)(len > 0 ? len : 1), &cap); /* page-locked: the H2D copies of the payload ranges run at link speed */
    const bool pinned = buf != nullptr;
    if (!pinned) buf = (uint8_t *)malloc((size_t)(len > 0 ? len : 1));
    const double t1 = prime_now();
    int64_t rc = -115; /* MZ_READ_ERROR */
    const bool got = buf && len > 0 && fread(buf, 1, (size_t)len, f) == (size_t)len;
    const double t2 = prime_now();
    if (got) rc = multi ? mzhip_prime_mem_multi(buf, (uint64_t)len, devices, ndev) : mzhip_prime_mem(buf, (uint64_t)len);
    if (prime_trace())
        fprintf(stderr, "[mzhip prime] %s: %.1f MiB image: pinned buffer %.1f ms, read %.1f ms, prime %.1f ms\n", path,
                (double)len / 1048576.0, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (prime_now() - t2) * 1e3);
    if (pinned) g_pinned.put(buf, cap);
    else free(buf);
    fclose(f);
    return rc;
}

int64_t mzhip_prime_file_multi(const char *path, const int32_t *devices, int32_t ndev) { return prime_file_on(path, devices, ndev, 1); }

int64_t mzhip_prime_file(const char *path) { return prime_file_on(path, nullptr, 0, 0); }

void mzhip_prime_hash_stats(uint64_t *checked, uint64_t *mismatched) {
    if (checked) *checked = g_prime_hash_checked.load();
    if (mismatched) *mismatched = g_prime_hash_bad.load();
}

void mzhip_prime_stats(uint64_t *entries, uint64_t *hits, uint64_t *misses) {
    std::shared_lock<std::shared_mutex> lk(g_prime_mu);
    if (entries) {
        *entries = 0;
        for (const auto &g : g_prime.gens) {
            for (size_t i = 0; i < g->entries.size(); i++) *entries += g->state[i].load(std::memory_order_relaxed) == 1 ? 1 : 0;
            *entries += g->store_entries;
        }
    }
    if (hits) *hits = g_prime_hits.load();
    if (misses) *misses = g_prime_misses.load();
}

// Used by mz_crypt_crc32_update: are these `size` bytes a chunk of a primed STORE entry?  The fingerprint finds the
// candidates, memcmp against the kept payload decides (so a file that changed between the prime and the read is
// never answered with the old CRC).  1 = yes, *crc = the chunk's CRC-32 as the device computed it.
__attribute__((visibility("hidden"))) int32_t mzhip_prime_store_crc(const uint8_t *buf, int32_t size, uint32_t *crc) {
    if (g_store_gens.load(std::memory_order_relaxed) == 0 || size < 32 || (uint32_t)size > kSeg) return 0;
    std::vector<std::shared_ptr<PrimeGen>> gens;
    {
        std::shared_lock<std::shared_mutex> lk(g_prime_mu);
        gens = g_prime.gens; /* the generations are immutable once published: search them without the lock */
    }
    const uint64_t key = store_key(buf, (uint32_t)size);
    for (const std::shared_ptr<PrimeGen> &g : gens) {
        if (!g->store_ready.load(std::memory_order_acquire)) continue; /* (its STORE index is still being built) */
        auto range = g->store_idx.equal_range(key);
        for (auto it = range.first; it != range.second; ++it) {
            const PrimeGen::StoreSeg &sg = g->store_segs[it->second];
            if (sg.len == (uint32_t)size && memcmp(buf, g->store + sg.off, (size_t)size) == 0) {
                *crc = sg.crc;
                g_prime_hits.fetch_add(1, std::memory_order_relaxed);
                return 1;
            }
        }
    }
    return 0;
}

__attribute__((visibility("hidden"))) int32_t mzhip_prime_any(void) { return g_any_gens.load(std::memory_order_relaxed) != 0; }

// Used by the READ shims: is the entry whose payload starts at `payload_off` primed?  The stream presents the payload
// bytes it has pulled so far (`head`, at least min(csize, 16) of them) and, when the zip layer set one, its
// TOTAL_IN_MAX (= the entry's compressed size, mz_zip.c:1829); offset, method, compressed size and up to 256 leading
// payload bytes must all agree.  On a hit returns 1, the cached output / sizes / CRCs, and *pin: a reference that keeps
// the generation alive until the stream hands it back with mzhip_prime_unpin() (close / delete / re-open).
__attribute__((visibility("hidden"))) int32_t mzhip_prime_lookup3(int32_t method, int64_t payload_off, const uint8_t *head,
                                                                  int32_t head_len, int64_t max_total_in, const uint8_t **data,
                                                                  int64_t *usize, int64_t *csize, uint32_t *crc,
                                                                  const uint32_t **seg_crc, void **pin, uint16_t *hash_alg,
                                                                  const uint8_t **hash_digest) {
    *pin = nullptr;
    if (hash_alg) *hash_alg = 0;
    if (hash_digest) *hash_digest = nullptr;
    std::vector<std::shared_ptr<PrimeGen>> gens;
    {
        std::shared_lock<std::shared_mutex> lk(g_prime_mu);
        if (g_prime.gens.empty()) return 0;
        gens = g_prime.gens; /* the keys of a published generation never change: search (and wait) without the lock */
    }
    /* the entry of generation g that this stream could be, settled (its chunk of the decode pipeline has landed): its
     * index, or -1.  What a stream can present is where its payload starts, the codec, the compressed size the zip
     * layer told it and the first bytes it pulled. */
    auto candidate = [&](const std::shared_ptr<PrimeGen> &g) -> int64_t {
        const std::vector<PrimedEntry> &ents = g->entries;
        size_t lo = 0, hi = ents.size();
        while (lo < hi) {
            size_t mid = (lo + hi) / 2;
            if (ents[mid].payload_off < payload_off) lo = mid + 1; else hi = mid;
        }
        if (lo == ents.size() || ents[lo].payload_off != payload_off) return -1;
        const PrimedEntry &e = ents[lo];
        const int32_t need = e.head_len < 16 ? e.head_len : 16;
        const int32_t cmp = head_len < e.head_len ? head_len : e.head_len;
        if (e.method != method || head_len < need || memcmp(head, e.head, (size_t)cmp) != 0) return -1;
        if (max_total_in > 0 && max_total_in != e.csize) return -1;
        if (g->state) { /* the entry's chunk may still be on its way: wait for that chunk */
            if (g->state[lo].load(std::memory_order_acquire) == 0) {
                const double w0 = prime_trace() ? prime_now() : 0.0;
                {
                    std::unique_lock<std::mutex> lk(g->mu);
                    g->cv.wait(lk, [&] { return g->state[lo].load(std::memory_order_acquire) != 0; });
                }
                if (prime_trace()) {
                    g_prime_wait_ns.fetch_add((uint64_t)((prime_now() - w0) * 1e9), std::memory_order_relaxed);
                    g_prime_wait_n.fetch_add(1, std::memory_order_relaxed);
                }
            }
            if (g->state[lo].load(std::memory_order_acquire) != 1) return -1; /* did not decode to its declared sizes: the ordinary path has the verdict */
        }
        return (int64_t)lo;
    };
    for (size_t gi = 0; gi < gens.size(); gi++) {
        const std::shared_ptr<PrimeGen> &g = gens[gi];
        const int64_t at = candidate(g);
        if (at < 0) continue;
        const PrimedEntry &e = g->entries[(size_t)at];
        /* Several archives may be primed at once (two versions of one file ...), and entries of two of them can agree in
         * everything a stream presents while their payloads differ past the 256th byte.  Then nobody can say whose
         * stream this is: it is not served (the ordinary per-entry path decodes what the stream really holds). */
        bool ambiguous = false;
        for (size_t gj = gi + 1; gj < gens.size() && !ambiguous; gj++) {
            const int64_t o = candidate(gens[gj]);
            if (o < 0) continue;
            const PrimedEntry &f = gens[gj]->entries[(size_t)o];
            ambiguous = f.crc != e.crc || f.usize != e.usize;
        }
        if (ambiguous) break;
        *data = g->out + e.out_off;
        *usize = e.usize;
        *csize = e.csize;
        *crc = e.crc;
        *seg_crc = g->seg_crc.data() + e.seg0;
        if (e.hash_alg && hash_alg && hash_digest) { /* (served, so the device's digest is the Hash field's) */
            *hash_alg = e.hash_alg;
            *hash_digest = e.hash_got;
        }
        g_prime_hits.fetch_add(1, std::memory_order_relaxed);
        *pin = new std::shared_ptr<PrimeGen>(g);
        return 1;
    }
    g_prime_misses.fetch_add(1, std::memory_order_relaxed);
    return 0;
}

__attribute__((visibility("hidden"))) void mzhip_prime_unpin(void *pin) {
    if (!pin) return;
    delete (std::shared_ptr<PrimeGen> *)pin; // (the last reference frees the generation: its buffers go back to pools with locks of their own)
}

// checksums only: crc(A||B) from crc(A), crc(B), |B| (shared with shim_crc32.c)
__attribute__((visibility("hidden"))) uint32_t mzhip_crc32_combine(uint32_t crc_a, uint32_t crc_b, uint64_t len_b) {
    return mzhip_crc32_combine_host(crc_a, crc_b, len_b);
}

} // extern "C"

// ---------------------------------------------------------------------------------------------------------
// Write-side prime (SURVEY 8b "Batching", config 5): compress many buffers in ONE launch per group ahead of the
// reference's untouched writer loop (mz_zip_writer_add_buffer -> mz_zip_entry_write -> mz_stream_zlib_write ->
// mz_crypt_crc32_update, one entry at a time).  The codec stream's WRITE side follows the bytes it is handed against
// the primed buffers (exact comparison, chunk by chunk); when an entry turns out to be one of them, close() emits the
// cached stream instead of launching, and the CRC updates of the 65 535-byte writer chunks (mz_zip_rw.c:55) are
// answered from device-computed segment CRCs.  Anything that diverges from the primed bytes falls back to the
// ordinary path with nothing lost.  The caller keeps the primed buffers alive and unchanged until the clear.

namespace {
struct WPrimed {
    const uint8_t *src;
    uint32_t len, out_len, crc;
    uint64_t out_off;
    int64_t seg0;
};
struct WPrimeCache {
    std::vector<WPrimed> ents;
    std::unordered_multimap<uint64_t, uint32_t> by_key;
    std::vector<uint32_t> seg_crc;
    std::vector<uint8_t *> outs; // one host buffer per launch group
    uint64_t hits = 0, misses = 0;
};
WPrimeCache g_wprime[3]; // methods 8, 14 (95 not primed: its container is laid out per entry on the host)
std::mutex g_wprime_mu;

int wprime_slot(int32_t method) { return method == 8 ? 0 : method == 14 ? 1 : -1; }

// key of an entry's first writer chunk: its length and its first and last 16 bytes
uint64_t wprime_key(const uint8_t *p, uint32_t n) {
    uint64_t a = 0, b = 0, c = 0, d = 0;
    memcpy(&a, p, 8);
    memcpy(&b, p + 8, 8);
    memcpy(&c, p + n - 16, 8);
    memcpy(&d, p + n - 8, 8);
    uint64_t h = 0x9E3779B97F4A7C15ull ^ n;
    h = (h ^ a) * 0xFF51AFD7ED558CCDull;
    h = (h ^ (h >> 32) ^ b) * 0xC4CEB9FE1A85EC53ull;
    h = (h ^ (h >> 29) ^ c) * 0xFF51AFD7ED558CCDull;
    h = (h ^ (h >> 32) ^ d) * 0xC4CEB9FE1A85EC53ull;
    return h ^ (h >> 31);
}

void wprime_clear_locked(WPrimeCache &w) {
    for (uint8_t *p : w.outs) free(p);
    w = WPrimeCache();
}
} // namespace

extern "C" {

void mzhip_prime_write_clear(void) {
    std::lock_guard<std::mutex> lk(g_wprime_mu);
    for (WPrimeCache &w : g_wprime) wprime_clear_locked(w);
}

void mzhip_prime_write_stats(uint64_t *entries, uint64_t *hits, uint64_t *misses) {
    std::lock_guard<std::mutex> lk(g_wprime_mu);
    uint64_t e = 0, h = 0, m = 0;
    for (const WPrimeCache &w : g_wprime) {
        e += w.ents.size();
        h += w.hits;
        m += w.misses;
    }
    if (entries) *entries = e;
    if (hits) *hits = h;
    if (misses) *misses = m;
}

int64_t mzhip_prime_write(int32_t method, const uint8_t *blob, const uint64_t *off, const uint32_t *len, uint32_t n) {
    const int slot = wprime_slot(method);
    if (slot < 0 || (!blob && n) || (n && (!off || !len))) return -102; /* MZ_PARAM_ERROR */
    DeviceCtx *c = nullptr;
    int32_t rc = ctx_for_current(&c);
    if (rc) return rc;
    const uint32_t kMaxLen = 8u << 20; /* what the WRITE shims hold before their first launch */
    const uint32_t piece = 64u << 10, pcap = piece + piece / 8 + 64;
    WPrimeCache fresh;
    // launch groups: bounded input bytes and bounded token scratch
    std::vector<uint32_t> ids;
    for (uint32_t i = 0; i < n; i++)
        if (len[i] >= 16u && len[i] <= kMaxLen) ids.push_back(i);
    size_t g0 = 0;
    while (g0 < ids.size()) {
        size_t g1 = g0;
        uint64_t in_bytes = 0, units = 0;
        uint32_t maxlen = 0;
        while (g1 < ids.size()) {
            const uint32_t l = len[ids[g1]];
            const uint32_t ml = l > maxlen ? l : maxlen;
            const uint64_t u = method == 8 ? units + (l + piece - 1) / piece
                                           : (uint64_t)(g1 - g0 + 1) * ((ml + piece - 1) / piece);
            if (g1 > g0 && (in_bytes + l > ((uint64_t)1 << 30) || u > 32768u)) break;
            in_bytes += (l + 63u) & ~63u;
            units = u;
            maxlen = ml;
            g1++;
        }
        const uint32_t gn = (uint32_t)(g1 - g0);
        // descriptors: method 8 = one per 64 KiB piece, method 14 = one per entry
        std::vector<uint64_t> in_off, out_off, seg_off;
        std::vector<uint32_t> in_len, out_cap, seg_len, first_unit(gn + 1);
        std::vector<uint8_t> fin;
        uint64_t ipos = 0, opos = 0;
        std::vector<uint64_t> ent_in(gn);
        for (uint32_t e = 0; e < gn; e++) {
            const uint32_t l = len[ids[g0 + e]];
            ent_in[e] = ipos;
            first_unit[e] = (uint32_t)in_off.size();
            if (method == 8) {
                for (uint32_t o = 0; o < l; o += piece) {
                    in_off.push_back(ipos + o);
                    in_len.push_back(l - o < piece ? l - o : piece);
                    out_off.push_back(opos);
                    out_cap.push_back(pcap);
                    fin.push_back(o + piece >= l ? 1 : 0);
                    opos += pcap;
                }
            } else {
                const uint32_t cap = l + l / 8 + 1024;
                in_off.push_back(ipos);
                in_len.push_back(l);
                out_off.push_back(opos);
                out_cap.push_back(cap);
                opos += (cap + 63u) & ~63u;
            }
            for (uint32_t o = 0; o < l; o += kSeg) {
                seg_off.push_back(ipos + o);
                seg_len.push_back(l - o < kSeg ? l - o : kSeg);
            }
            ipos += (l + 63u) & ~63u;
        }
        first_unit[gn] = (uint32_t)in_off.size();
        const uint32_t nu = (uint32_t)in_off.size(), ns = (uint32_t)seg_off.size();
        const uint64_t out_base = ipos; /* outputs behind the inputs in one allocation */
        for (uint64_t &o : out_off) o += out_base;
        const size_t meta = (size_t)nu * (8 + 8 + 4 + 4 + 4 + 4 + 4 + 1) + (size_t)ns * (8 + 4 + 4) + 256;
        Scratch d_data, d_meta;
        HIP_TRY(hipMalloc(&d_data.p, out_base + opos + 64));
        HIP_TRY(hipMalloc(&d_meta.p, meta));
        {   /* the group's inputs in their padded device layout, one transfer */
            uint8_t *stage = (uint8_t *)malloc(ipos + 64);
            if (!stage) return -4;
            for (uint32_t e = 0; e < gn; e++) memcpy(stage + ent_in[e], blob + off[ids[g0 + e]], len[ids[g0 + e]]);
            const hipError_t ce = hipMemcpy(d_data.p, stage, ipos, hipMemcpyHostToDevice);
            free(stage);
            if (ce != hipSuccess) return fail("hipMemcpy (buffers to prime)", ce);
        }
        uint8_t *m = (uint8_t *)d_meta.p;
        uint64_t *d_in_off = (uint64_t *)m, *d_out_off = d_in_off + nu, *d_seg_off = d_out_off + nu;
        uint32_t *d_in_len = (uint32_t *)(d_seg_off + ns), *d_out_cap = d_in_len + nu, *d_out_len = d_out_cap + nu,
                 *d_crc = d_out_len + nu;
        int32_t *d_status = (int32_t *)(d_crc + nu);
        uint32_t *d_seg_len = (uint32_t *)(d_status + nu), *d_seg_crc = d_seg_len + ns;
        uint8_t *d_fin = (uint8_t *)(d_seg_crc + ns);
        HIP_TRY(hipMemcpy(d_in_off, in_off.data(), (size_t)nu * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_out_off, out_off.data(), (size_t)nu * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_in_len, in_len.data(), (size_t)nu * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_out_cap, out_cap.data(), (size_t)nu * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_seg_off, seg_off.data(), (size_t)ns * 8, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d_seg_len, seg_len.data(), (size_t)ns * 4, hipMemcpyHostToDevice));
        if (method == 8) {
            HIP_TRY(hipMemcpy(d_fin, fin.data(), nu, hipMemcpyHostToDevice));
            rc = mzhip_deflate_batch(d_data.p, d_in_off, d_in_len, d_data.p, d_out_off, d_out_cap, d_fin, nu, d_out_len, d_crc,
                                     d_status, nullptr);
        } else {
            rc = mzhip_lzma_encode_batch(d_data.p, d_in_off, d_in_len, maxlen, d_data.p, d_out_off, d_out_cap, nullptr, nu,
                                         d_out_len, d_crc, d_status, nullptr);
        }
        if (rc) return rc;
        rc = mzhip_crc32_batch(d_data.p, d_seg_off, d_seg_len, ns, nullptr, d_seg_crc, nullptr);
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(nullptr)); /* the launches above are on the null stream */
        std::vector<uint32_t> h_len(nu), h_crc(nu), h_seg(ns);
        std::vector<int32_t> h_st(nu);
        HIP_TRY(hipMemcpy(h_len.data(), d_out_len, (size_t)nu * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_crc.data(), d_crc, (size_t)nu * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_st.data(), d_status, (size_t)nu * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(h_seg.data(), d_seg_crc, (size_t)ns * 4, hipMemcpyDeviceToHost));
        // one transfer of the whole output region, then the pieces are closed up on the host
        uint8_t *raw = (uint8_t *)malloc(opos + 64);
        if (!raw) return -4;
        hipError_t he = hipMemcpy(raw, (uint8_t *)d_data.p + out_base, opos, hipMemcpyDeviceToHost);
        if (he != hipSuccess) {
            free(raw);
            return fail("hipMemcpy (primed streams)", he);
        }
        uint64_t packed = 0;
        for (uint32_t u = 0; u < nu; u++) packed += h_len[u];
        uint8_t *host_out = (uint8_t *)malloc(packed + 64);
        if (!host_out) {
            free(raw);
            return -4;
        }
        uint64_t w = 0;
        int64_t seg_at = (int64_t)fresh.seg_crc.size(), seg_i = 0;
        for (uint32_t e = 0; e < gn; e++) {
            const uint32_t i = ids[g0 + e], l = len[i];
            const uint32_t nseg = (l + kSeg - 1) / kSeg;
            bool ok = true;
            uint32_t crc = 0;
            const uint64_t w0 = w;
            for (uint32_t u = first_unit[e]; u < first_unit[e + 1]; u++) {
                if (h_st[u] != 0 || h_len[u] > out_cap[u]) ok = false;
                if (!ok) break;
                memcpy(host_out + w, raw + (out_off[u] - out_base), h_len[u]);
                w += h_len[u];
                crc = (u == first_unit[e]) ? h_crc[u] : mzhip_crc32_combine_host(crc, h_crc[u], in_len[u]);
            }
            if (ok) {
                WPrimed pe;
                pe.src = blob + off[i];
                pe.len = l;
                pe.out_off = w0;
                pe.out_len = (uint32_t)(w - w0);
                pe.crc = crc;
                pe.seg0 = seg_at + seg_i;
                // out_off is relative to this group's buffer: remember which one through the pointer table
                pe.out_off |= (uint64_t)fresh.outs.size() << 48;
                fresh.by_key.emplace(wprime_key(pe.src, l < kSeg ? l : kSeg), (uint32_t)fresh.ents.size());
                fresh.ents.push_back(pe);
            } else {
                w = w0; /* not cached: the ordinary path and its exact behaviour */
            }
            seg_i += nseg;
        }
        fresh.seg_crc.insert(fresh.seg_crc.end(), h_seg.begin(), h_seg.end());
        fresh.outs.push_back(host_out);
        free(raw);
        g0 = g1;
    }
    std::lock_guard<std::mutex> lk(g_wprime_mu);
    wprime_clear_locked(g_wprime[slot]);
    g_wprime[slot] = std::move(fresh);
    return (int64_t)g_wprime[slot].ents.size();
}

// Used by the WRITE shims.  *id < 0: does a primed buffer start with these `size` bytes?  *id >= 0: do the bytes at
// `pos` of that buffer continue with them?  Returns 1 on a match; *have_crc says whether the chunk is one of the
// buffer's 65 535-byte segments, whose CRC-32 the device already computed.
__attribute__((visibility("hidden"))) int32_t mzhip_wprime_track(int32_t method, int64_t *id, int64_t pos, const uint8_t *buf,
                                                                 int32_t size, uint32_t *chunk_crc, int32_t *have_crc,
                                                                 const uint8_t **src) {
    const int slot = wprime_slot(method);
    *have_crc = 0;
    if (slot < 0 || size <= 0) return 0;
    std::lock_guard<std::mutex> lk(g_wprime_mu);
    WPrimeCache &w = g_wprime[slot];
    if (w.ents.empty()) return 0;
    const WPrimed *e = nullptr;
    if (*id < 0) {
        if (pos != 0 || size < 16) return 0;
        auto range = w.by_key.equal_range(wprime_key(buf, (uint32_t)size));
        for (auto it = range.first; it != range.second; ++it) {
            const WPrimed &c = w.ents[it->second];
            const uint32_t first = c.len < kSeg ? c.len : kSeg;
            if (first == (uint32_t)size && memcmp(c.src, buf, (size_t)size) == 0) {
                *id = (int64_t)it->second;
                e = &c;
                break;
            }
        }
        if (!e) {
            w.misses++;
            return 0;
        }
    } else {
        if ((uint64_t)*id >= w.ents.size()) return 0;
        e = &w.ents[(size_t)*id];
        if (pos + size > (int64_t)e->len || memcmp(e->src + pos, buf, (size_t)size) != 0) return 0;
    }
    if (pos % kSeg == 0 && ((uint32_t)size == kSeg || pos + size == (int64_t)e->len)) {
        *chunk_crc = w.seg_crc[(size_t)(e->seg0 + pos / kSeg)];
        *have_crc = 1;
        *src = e->src + pos; /* the primed bytes these were compared with: what the CRC symbol compares again */
    }
    return 1;
}

// The primed buffer behind `id`: its bytes (for a stream that diverged and must fall back), and -- when the entry
// ended exactly at the buffer's end (pos == len) -- the cached stream.  Returns 1 if the stream may be emitted.
__attribute__((visibility("hidden"))) int32_t mzhip_wprime_result(int32_t method, int64_t id, int64_t pos, const uint8_t **src,
                                                                  const uint8_t **out, uint32_t *out_len) {
    const int slot = wprime_slot(method);
    if (slot < 0) return 0;
    std::lock_guard<std::mutex> lk(g_wprime_mu);
    WPrimeCache &w = g_wprime[slot];
    if (id < 0 || (uint64_t)id >= w.ents.size()) return 0;
    const WPrimed &e = w.ents[(size_t)id];
    *src = e.src;
    if (pos != (int64_t)e.len) return 0;
    *out = w.outs[(size_t)(e.out_off >> 48)] + (e.out_off & (((uint64_t)1 << 48) - 1));
    *out_len = e.out_len;
    w.hits++;
    return 1;
}

} // extern "C"
