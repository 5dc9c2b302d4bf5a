// mzhip_kernels.hip -- the gfx950 kernels of libmzhip.so.  The per-entry algorithms live in the *_core.h / *.inc files
// next to this one; this file wraps them in kernels (work distribution, LDS carving) and includes the host runtime
// (mzhip_runtime.inc: device context, launchers of the batch C ABI of include/mzhip.h, prime cache).
//
// Launch shape of K1 (MI355X: 256 CUs x 4 SIMDs, 160 KiB LDS/CU, 8 XCDs):
//   - one wavefront per ZIP entry, 4 wavefronts per workgroup, 9.98 KiB of LDS per wave: 3.8 KiB of Huffman tables, 4.75 KiB
//     of per-lane stream rings while the lanes walk (the chase window, inflate_walk.inc), 1.2 KiB of pool -- and while the
//     window's records become bytes the pool runs on through the dead rings (5 KiB: staging bytes, pending bits,
//     back-reference list; inflate_emit.inc / inflate_commit.inc).  The LZ77 window beyond the chunk is the output buffer
//     itself.  4 workgroups = 16 waves fit a CU by LDS; the kernel is compiled for 4 waves per SIMD (<= 128 VGPRs);
//   - 131 KiB of step-record scratch in HBM per resident wave (MZ_REC_BYTES: 4 bytes + 1 byte per decode step);
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
#include <dlfcn.h>
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
/* MZ_WAVE_INDEX.  threadIdx.x >> 6 is the same in all 64 lanes, but to the compiler it is a per-lane value, and so is
 * everything computed from it: the wave's LDS slice, its record scratch -- and `rec != null`, which K1 asks before every
 * window.  That one "divergent" branch joined the window path and the step loop at the bottom of the block loop, so every
 * loop-carried value of mz_inflate_entry (bit cursor, output position, block state) lived in vector registers and all of its
 * wave-uniform control ran on the vector unit under exec masks (rounds 1 - 4; found with opt -passes='print<uniformity>',
 * profiles/r5).  Reading the index back through v_readfirstlane keeps that state on the scalar unit: K1 on 8 KiB entries
 * +6.5 %, on 64 KiB entries +0.4 %.  Only K1 and its one-block twin do it: the LZMA slot kernel LOSES 6.7 % with a uniform wave
 * index (config 4 10.38 -> 9.69 GiB/s on one box, profiles/r5/call6_probe.log -- its decisions then go through the CU's one
 * scalar unit, which round 4 measured as the slower port for it), the DEFLATE encoder does not move. */
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

// Two instantiations: RESUMABLE = false is the batch as mzhip_inflate_batch launches it (no entry is taken up or left in the
// middle: a.resume and a.stop are null, and the compiler drops the resume bookkeeping -- which block the cursor is in, where the
// step loop's unwritten tokens start, whether to stop at the next header -- from a function that already holds more uniform state
// than a wave has scalar registers); true is the window-by-window decode of the drop-in streams (mzhip_inflate_host_a with a state).
template <bool RESUMABLE>
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64, MZ_MIN_WAVES_PER_SIMD) void k_inflate_batch(InflateArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crc_tab = (uint32_t *)smem;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const int wave = (int)MZ_UNIFORM(threadIdx.x >> 6); /* wave-uniform, and known to be: MZ_WAVE_INDEX above */
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
                         (RESUMABLE && a.resume) ? a.resume + e : nullptr, (RESUMABLE && a.stop) ? a.stop + e : nullptr, &r, nullptr);
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

// K3: one wave per workgroup, the wave's whole probability model (15.6 KiB) in LDS -> 10 waves per CU.
#ifdef MZ_LZMA_WAVES /* measurement builds (profiles/r3/scripts/ab_k3.sh): waves per SIMD the register allocation is held to */
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
        // the vector-port build of the decoder (lzma_core.h: the scalar-port form and every mix of the two measured slower)
        mz_lzma_entry_v(a.in + a.in_off[e], a.in_len[e], a.out + a.out_off[e], a.out_cap[e],
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
    const uint32_t wave = threadIdx.x >> 6; /* (per-lane to the compiler, on purpose: MZ_WAVE_INDEX) */
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

// .xz in windows: ONE block's LZMA2 chunk sequence taken up where the call before left it (mz_lzma2_run, xz_core.h; the
// window mode of the drop-in READ stream for method 95, shim_lzma.c).  State record and model travel as in k_lzma_resume.
struct Lzma2RunArgs {
    const uint8_t *in;
    uint32_t in_len;
    uint8_t *buf; // [dictionary so far | room for this window]
    uint32_t buf_cap;
    const mz_lzma2_state *rs;
    mz_lzma2_state *st;
    uint16_t *model; // MZ_LZMA_MODEL_U16 probabilities, in and out
    uint32_t *out_len;
    uint32_t *in_used;
    int32_t *status;
    const mzhip_crc_tables *tabs;
    const uint64_t *tab64;
};
__global__ __launch_bounds__(64) void k_lzma2_run(Lzma2RunArgs a) {
    __shared__ __attribute__((aligned(16))) mz_xz_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        crc_tab[i] = a.tabs->byte_tab[i];
        lds.crc64_tab[i] = a.tab64[i];
    }
    __syncthreads();
    mz_lzma_result r;
    mz_lzma2_run(a.in, a.in_len, a.buf, a.buf_cap, &lds, crc_tab, a.tabs, a.model, a.rs, a.st, &r);
    *a.out_len = r.out_len; // wave-uniform results: stored by all lanes
    *a.in_used = r.in_used;
    *a.status = r.status;
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
    const uint32_t *warm;      // may be null (0): bytes in front of a piece, in the same blob, that it may match into (mz_deflate_piece)
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
    const int wave = threadIdx.x >> 6; /* (per-lane to the compiler, on purpose: MZ_WAVE_INDEX) */
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
        const uint32_t warm = a.warm ? MZ_UNIFORM(a.warm[e]) : 0u;
        mz_deflate_result r;
        mz_deflate_piece<kParse>(in - warm, MZ_UNIFORM(a.in_len[e]) + warm, warm, out, MZ_UNIFORM(a.out_cap[e]), fin,
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
    uint32_t *links;       // chain pass: 4 bytes per input position, laid out like tok; null = no stream of more than one block
    uint32_t *chain_head;  // chain pass: 1 << MZ_LZE_FAR_HBITS words per resident wave of k_lz_chain_batch
    uint32_t skip_blocks;  // blocks in front of every entry that are history only (a stream written in segments): not parsed
    uint32_t far_depth;    // links of the chain the block parse follows per position (MZ_LZE_DEPTH_FOR_PRESET)
};

// LZMA encode, pass 0: the chain pass, one wave per method-14 stream of more than one block (lzma_enc_core.h mz_lz_chain)
__global__ __launch_bounds__(64) void k_lz_chain_batch(LzmaEncArgs a) {
    MZ_LANE_DECL
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter + 2);
        if (e >= a.n) break;
        const uint32_t len = MZ_UNIFORM(a.in_len[e]);
        if (len <= MZ_DEF_BLOCK || (a.mode && MZ_UNIFORM((uint32_t)a.mode[e]) != 0u)) continue;
        const uint64_t io = a.in_off[e];
        const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
        mz_lz_chain(in, len, a.links + (size_t)e * a.maxb * MZ_DEF_BLOCK, a.chain_head + ((size_t)blockIdx.x << MZ_LZE_FAR_HBITS));
    }
}

// LZMA encode, pass 1: the LZ77 parse, one wave per 64 KiB block of any entry (work item = entry * maxb + block).
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64) void k_lz_tokenize_batch(LzmaEncArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[]; /* one head table per wave, then (ways - 1) more each */
    MZ_LANE_DECL
    const uint32_t wave = threadIdx.x >> 6; /* (per-lane to the compiler, on purpose: MZ_WAVE_INDEX) */
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
        if (lo < len && b >= a.skip_blocks) {
            const uint64_t io = a.in_off[e];
            const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
            const uint32_t far = (a.links && len > MZ_DEF_BLOCK && !(a.mode && MZ_UNIFORM((uint32_t)a.mode[e]) != 0u)) ? 1u : 0u;
            nt = mz_lz_tokenize(in, lo, (len - lo < MZ_DEF_BLOCK) ? len : lo + MZ_DEF_BLOCK, a.tok + (size_t)w * MZ_DEF_BLOCK, L,
                                MZ_UNIFORM(a.ways), xhead, far ? a.links + (size_t)e * a.maxb * MZ_DEF_BLOCK : (const uint32_t *)nullptr,
                                MZ_UNIFORM(a.far_depth));
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

// LZMA encode, pass 2 for the LZMA2 chunks of ONE .xz block (mzhip_xz_encode_*): entry 0 of `a` is the block, parsed as one
// stream (chain pass and all: a chunk's matches reach back over the chunks before it); every 64 KiB block of it is coded by
// a wave of its own with a fresh coder and model -- LZMA2 chunks that reset state and properties but keep the dictionary.
struct LzmaEncChunksArgs {
    LzmaEncArgs a;
    uint32_t nchunks;
    uint32_t chunk_cap;      // bytes of room per chunk: chunk b is written at a.out + a.out_off[0] + b * chunk_cap
    uint32_t *chunk_len;     // [nchunks]
    int32_t *chunk_status;   // [nchunks]
};
__global__ __launch_bounds__(64) void k_lzma2_chunks_encode(LzmaEncChunksArgs r) {
    __shared__ __attribute__((aligned(16))) mz_lzma_lds lds;
    __shared__ uint32_t crc_tab[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = r.a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const LzmaEncArgs &a = r.a;
    const uint64_t io = a.in_off[0], oo = a.out_off[0];
    const uint8_t *in = a.in + (((uint64_t)MZ_UNIFORM((uint32_t)(io >> 32)) << 32) | MZ_UNIFORM((uint32_t)io));
    uint8_t *out0 = a.out + (((uint64_t)MZ_UNIFORM((uint32_t)(oo >> 32)) << 32) | MZ_UNIFORM((uint32_t)oo));
    const uint32_t len = MZ_UNIFORM(a.in_len[0]);
    for (;;) {
        uint32_t b;
        MZ_WAVE_FETCH_ADD(b, a.counter + 1);
        if (b >= r.nchunks) break;
        const uint32_t hi = (len - b * MZ_DEF_BLOCK < MZ_DEF_BLOCK) ? len : (b + 1u) * MZ_DEF_BLOCK;
        mz_lzma_enc_result res;
        mz_lzma_rc_encode_x(in, hi, a.tok, a.ntok, 1u, out0 + (size_t)b * r.chunk_cap, r.chunk_cap, &lds, crc_tab, a.tabs, &res, b,
                            (const mz_lzma_enc_state *)nullptr, (mz_lzma_enc_state *)nullptr, (uint16_t *)nullptr);
        r.chunk_len[b] = res.out_len; // wave-uniform results: stored by all lanes
        r.chunk_status[b] = res.status;
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

// ---- one large entry on many waves (mzhip_inflate_parallel_host; the orchestration is inflate_parallel.inc)
struct ParFindArgs {
    const uint8_t *in;
    uint32_t in_len, b0, b1;
    uint32_t *cands;
    uint32_t cap;
    uint32_t *count;
};
// every bit offset of [b0, b1): could a block header start here (mz_block_header_plausible)?  What passes is appended (order
// does not matter, the host puts them in order: ~7 per KiB of text).
// k_find_blocks_lane: one offset per lane -- 64 lanes fetch the same dozen bytes, a quarter of them go on to four byte loads of
// the stored-block test, and every candidate is an atomic of its own: 1.25 ms for the 16.8 MB one window of text is shown.
// Kept as the statement of what is searched for (tests/test_gpu_streams.py holds the kernel below against it).
__global__ __launch_bounds__(256) void k_find_blocks_lane(ParFindArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t p = (uint64_t)a.b0 + (uint64_t)blockIdx.x * 256u + threadIdx.x; p < a.b1; p += stride)
        if (mz_block_header_plausible(a.in, a.in_len, (uint32_t)p)) {
            const uint32_t k = atomicAdd(a.count, 1u);
            if (k < a.cap) a.cands[k] = (uint32_t)p;
        }
}
// k_find_blocks: 32 offsets per lane out of one 16-byte fetch.  The tests that need no loop are made for all 32 at once on the
// 64-bit register (BTYPE = 2 is "bit k+1 clear, bit k+2 set"; HLIT / HDIST > 29 is "four bits in a row set"), a lane then
// walks the offsets that are left (7 of 32 on random bits) through the Kraft sum of the code-length code; the stored-block
// test is made once per byte position (LEN ^ NLEN of the 5 positions the lane's offsets can point at).  One atomic per wave
// and round.  Bit g of the aligned dwords q[] is stream bit g - 8 * (in & 3).
__global__ __launch_bounds__(256) void k_find_blocks(ParFindArgs a) {
    const int lane = (int)(threadIdx.x & 63u);
    if (a.in_len < 24u) return;
    const uint32_t mis = (uint32_t)((uintptr_t)a.in & 3u), sh0 = mis * 8u;
    const uint32_t *q = (const uint32_t *)(a.in - mis);
    const uint32_t nd = (a.in_len + mis + 3u) >> 2; /* dwords of q[] that hold bytes of the stream */
    /* offsets that may be looked at: [b0, lim), where the 24 bytes behind an offset are still the stream's */
    const uint64_t lim64 = 8ull * (uint64_t)(a.in_len - 23u);
    const uint64_t lim = lim64 < (uint64_t)a.b1 ? lim64 : (uint64_t)a.b1;
    if ((uint64_t)a.b0 >= lim) return;
    const uint64_t g0 = (uint64_t)a.b0 + sh0, g1 = lim + sh0; /* the same range in bits of q[] */
    const uint64_t j0 = g0 >> 5, j1 = (g1 + 31u) >> 5;       /* ... and in dwords: a lane takes one */
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t jb = j0 + (uint64_t)blockIdx.x * 256u + (threadIdx.x & ~63u); jb < j1; jb += stride) { /* (wave-uniform trip count) */
        const uint64_t j = jb + (uint64_t)lane;
        uint32_t found = 0; /* bit k: offset 32 j + k (in bits of q[]) is a candidate */
        if (j < j1) {
            uint32_t d[4];
#pragma unroll
            for (uint32_t t = 0; t < 4u; t++) d[t] = (j + t < nd) ? q[j + t] : 0u;
            const uint64_t lo = ((uint64_t)d[1] << 32) | d[0], hi = ((uint64_t)d[3] << 32) | d[2];
            /* which of the 32 offsets are inside [g0, g1) */
            uint32_t in_range = 0xFFFFFFFFu;
            if ((j << 5) < g0) in_range &= 0xFFFFFFFFu << (uint32_t)(g0 - (j << 5));
            if (((j + 1u) << 5) > g1) in_range &= 0xFFFFFFFFu >> (uint32_t)(((j + 1u) << 5) - g1);
            const uint32_t b1c = (uint32_t)~(lo >> 1), b2 = (uint32_t)(lo >> 2);
            uint32_t dyn = b1c & b2 & in_range;
            dyn &= ~(uint32_t)((lo >> 4) & (lo >> 5) & (lo >> 6) & (lo >> 7));    /* HLIT <= 29 */
            dyn &= ~(uint32_t)((lo >> 9) & (lo >> 10) & (lo >> 11) & (lo >> 12)); /* HDIST <= 29 */
            while (dyn) {
                const uint32_t k = (uint32_t)__builtin_ctz(dyn);
                dyn &= dyn - 1u;
                const uint32_t ncode = ((uint32_t)(lo >> (k + 13u)) & 15u) + 4u;
                const uint32_t s = k + 17u; /* 17 .. 48: the 3-bit lengths start here, 57 bits at most */
                uint64_t w = (lo >> s) | (hi << (64u - s));
                uint32_t kraft = 0;
                for (uint32_t i = 0; i < ncode; i++) {
                    const uint32_t l = (uint32_t)w & 7u;
                    w >>= 3;
                    kraft += l ? (128u >> l) : 0u;
                }
                if (kraft == 128u) found |= 1u << k;
            }
            /* stored blocks: BTYPE = 0 and LEN ^ NLEN = 0xFFFF at byte (g + 10) >> 3 = 4 j + 1 .. 4 j + 5 */
            const uint32_t sto = b1c & ~b2 & in_range;
            if (sto) {
#pragma unroll
                for (uint32_t t = 1; t <= 5u; t++) {
                    const uint32_t v = (uint32_t)(t < 5u ? (lo >> (8u * t)) : ((lo >> 40) | (hi << 24)));
                    const uint32_t km = t == 1u ? 0x0000003Fu : t == 2u ? 0x00003FC0u : t == 3u ? 0x003FC000u : t == 4u ? 0x3FC00000u : 0xC0000000u;
                    if (((v ^ (v >> 16)) & 0xFFFFu) == 0xFFFFu) found |= sto & km;
                }
            }
        }
        /* append: a slot per candidate, one fetch-add per wave and round */
        const uint32_t n = (uint32_t)__builtin_popcount(found);
        const uint32_t incl = mz_wave_incl_scan(n, lane);
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        if (total) { /* (wave-uniform) */
            uint32_t base = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) base = atomicAdd(a.count, total);
            __builtin_amdgcn_wave_barrier();
            base = MZ_UNIFORM(base);
            uint32_t at = base + incl - n, f = found;
            while (f) {
                const uint32_t k = (uint32_t)__builtin_ctz(f);
                f &= f - 1u;
                if (at < a.cap) a.cands[at] = (uint32_t)((j << 5) + k - sh0);
                at++;
            }
        }
    }
}

// The candidates of k_find_blocks are ~0.09 % of a text stream's bit offsets, and all but one in two hundred of them are not block
// headers: each used to cost a wave of k_inflate_blocks a header parse (120 000 waves per window against 530 blocks: 1 of the
// counting pass's 2 ms).  Here a LANE reads one candidate's header to its end -- the run-length coded lengths through the
// code-length code, bit by bit as puff.c would -- and keeps it only if the decoder would get past the header: no repeat without a
// length before it or past HLIT + HDIST, an end-of-block code, and both sets neither over-subscribed nor incomplete (a single
// 1-bit code and, for distances, no code at all excepted): the verdicts of inflate_header.inc / inflate_tables.inc, i.e. zlib's.
// Stored-block candidates pass as they are.  What is kept is appended to out[] (no order).
struct ParCheckArgs {
    const uint8_t *in;
    uint32_t in_len;
    const uint32_t *cands, *count; // what k_find_blocks found (*count may exceed cap: then cap of them are there)
    uint32_t cap;
    uint32_t *out, *out_count;
};
__device__ __forceinline__ uint64_t mz_find_bits64(const uint32_t *q, uint32_t nd, uint64_t g) {
    const uint64_t j = g >> 5;
    const uint32_t s = (uint32_t)g & 31u;
    const uint32_t d0 = j < nd ? q[j] : 0u, d1 = j + 1u < nd ? q[j + 1u] : 0u, d2 = j + 2u < nd ? q[j + 2u] : 0u;
    uint64_t w = (((uint64_t)d1 << 32) | d0) >> s;
    if (s) w |= (uint64_t)d2 << (64u - s);
    return w;
}
__global__ __launch_bounds__(256) void k_check_headers(ParCheckArgs a) {
    const int lane = (int)(threadIdx.x & 63u);
    const uint32_t n = *a.count < a.cap ? *a.count : a.cap;
    const uint32_t mis = (uint32_t)((uintptr_t)a.in & 3u), sh0 = mis * 8u;
    const uint32_t *q = (const uint32_t *)(a.in - mis);
    const uint32_t nd = (a.in_len + mis + 3u) >> 2;
    const uint64_t end_bit = 8ull * a.in_len + sh0; /* (in bits of q[]) */
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t ib = blockIdx.x * 256u + (threadIdx.x & ~63u); ib < n; ib += stride) { /* (wave-uniform trip count) */
        const uint32_t i = ib + (uint32_t)lane;
        uint32_t keep = 0, p = 0;
        if (i < n) {
            p = a.cands[i];
            const uint64_t g = (uint64_t)p + sh0;
            const uint64_t h = mz_find_bits64(q, nd, g);
            if ((((uint32_t)h >> 1) & 3u) == 0u) {
                keep = 1; /* stored: LEN / NLEN have been looked at */
            } else {
                const uint32_t nlen = (((uint32_t)h >> 3) & 31u) + 257u, ndist = (((uint32_t)h >> 8) & 31u) + 1u, ncode = (((uint32_t)h >> 13) & 15u) + 4u;
                /* the code-length code: lengths by symbol (3 bits each), how many codes of each length (5 bits each) */
                const uint64_t ord_lo = 16ull | (17ull << 5) | (18ull << 10) | (0ull << 15) | (8ull << 20) | (7ull << 25) | (9ull << 30) | (6ull << 35) | (10ull << 40) |
                                        (5ull << 45) | (11ull << 50) | (4ull << 55);
                const uint64_t ord_hi = 12ull | (3ull << 5) | (13ull << 10) | (2ull << 15) | (14ull << 20) | (1ull << 25) | (15ull << 30);
                uint64_t lw = mz_find_bits64(q, nd, g + 17u), cl = 0, bc = 0;
                for (uint32_t k = 0; k < ncode; k++) {
                    const uint32_t l = (uint32_t)lw & 7u;
                    lw >>= 3;
                    const uint32_t sym = (uint32_t)(k < 12u ? ord_lo >> (5u * k) : ord_hi >> (5u * (k - 12u))) & 31u;
                    cl |= (uint64_t)l << (3u * sym);
                    bc += 1ull << (5u * l);
                }
                /* its symbols in code order (counting sort by length): 19 x 5 bits */
                uint64_t s0 = 0, s1 = 0, offs = 0; /* offs: 5 bits per length, where the next symbol of that length goes */
                {
                    uint32_t at = 0;
                    for (uint32_t l = 1; l <= 7u; l++) {
                        offs |= (uint64_t)at << (5u * l);
                        at += (uint32_t)(bc >> (5u * l)) & 31u;
                    }
                }
                for (uint32_t sym = 0; sym < 19u; sym++) {
                    const uint32_t l = (uint32_t)(cl >> (3u * sym)) & 7u;
                    if (l) {
                        const uint32_t at = (uint32_t)(offs >> (5u * l)) & 31u;
                        offs += 1ull << (5u * l);
                        if (at < 12u) s0 |= (uint64_t)sym << (5u * at);
                        else s1 |= (uint64_t)sym << (5u * (at - 12u));
                    }
                }
                /* the lengths of the literal / length and distance codes, read for their sums only */
                uint64_t pos = g + 17u + 3u * ncode, bb = mz_find_bits64(q, nd, pos);
                uint32_t used = 0, idx = 0, prev = 0, ok = 1, has256 = 0, ll_total = 0, d_total = 0, ll_max = 0, d_max = 0;
                const uint32_t ntot = nlen + ndist;
                while (ok && idx < ntot) {
                    if (used > 48u) {
                        pos += used;
                        used = 0;
                        bb = mz_find_bits64(q, nd, pos);
                    }
                    uint32_t code = 0, first = 0, index = 0, sym = 99u, len = 1;
                    for (; len <= 7u; len++) {
                        code |= (uint32_t)(bb >> used) & 1u;
                        used++;
                        const uint32_t count = (uint32_t)(bc >> (5u * len)) & 31u;
                        if (code < first + count) { /* (code >= first always: the code-length code is complete, k_find_blocks) */
                            const uint32_t at = index + (code - first);
                            sym = (uint32_t)(at < 12u ? s0 >> (5u * at) : s1 >> (5u * (at - 12u))) & 31u;
                            break;
                        }
                        index += count;
                        first += count;
                        first <<= 1;
                        code <<= 1;
                    }
                    if (sym == 99u) { /* no code of the code-length code */
                        ok = 0;
                        break;
                    }
                    uint32_t val, rep;
                    if (sym < 16u) {
                        val = sym;
                        rep = 1;
                    } else {
                        const uint32_t ext = sym == 16u ? 2u : sym == 17u ? 3u : 7u;
                        const uint32_t xb = (uint32_t)(bb >> used) & ((1u << ext) - 1u);
                        used += ext;
                        if (sym == 16u) {
                            if (idx == 0u) {
                                ok = 0;
                                break;
                            }
                            val = prev;
                            rep = 3u + xb;
                        } else {
                            val = 0;
                            rep = (sym == 17u ? 3u : 11u) + xb;
                        }
                        if (idx + rep > ntot) {
                            ok = 0;
                            break;
                        }
                    }
                    if (val) {
                        const uint32_t n1 = idx < nlen ? (rep < nlen - idx ? rep : nlen - idx) : 0u, n2 = rep - n1;
                        ll_total += n1 << (15u - val);
                        d_total += n2 << (15u - val);
                        if (n1 && val > ll_max) ll_max = val;
                        if (n2 && val > d_max) d_max = val;
                        if (idx <= 256u && 256u < idx + rep) has256 = 1;
                    }
                    prev = val;
                    idx += rep;
                }
                if (pos + used > end_bit) ok = 0; /* (the header runs past the bytes this window was shown: the chain ends in front of it either way) */
                if (!has256) ok = 0;
                if (ll_total > 32768u || (ll_total < 32768u && ll_max != 1u)) ok = 0;
                if (d_total > 32768u || (d_total < 32768u && d_max > 1u)) ok = 0;
                keep = ok;
            }
        }
        uint64_t km = __ballot(keep != 0u);
        if (km) { /* (wave-uniform) */
            const uint32_t total = (uint32_t)__popcll(km);
            uint32_t base = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) base = atomicAdd(a.out_count, total);
            __builtin_amdgcn_wave_barrier();
            base = MZ_UNIFORM(base);
            if (keep) a.out[base + MZ_RANK_BELOW(km)] = p; /* (never more than were found: out[] is as large as cands[]) */
        }
    }
}

struct ParBlocksArgs {
    const uint8_t *in;
    uint32_t in_len;
    uint8_t *out;
    uint32_t *ptr;
    const uint32_t *bits, *pos; // per candidate: the header's bit, the out position of the block's first byte
    uint32_t n, mode;
    uint32_t *res; // 4 words per candidate (mz_inflate_one_block)
    uint32_t *counter;
    const mzhip_crc_tables *tabs;
    uint8_t *rec;
};
// a wave per candidate block (persistent waves over a work queue, like k_inflate_batch): mode 1 counts, mode 2 writes
// literals and the source map
__global__ __launch_bounds__(MZ_WAVES_PER_WG * 64, MZ_MIN_WAVES_PER_SIMD) void k_inflate_blocks(ParBlocksArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *crc_tab = (uint32_t *)smem;
    for (int i = threadIdx.x; i < 256; i += blockDim.x) crc_tab[i] = a.tabs->byte_tab[i];
    __syncthreads();
    MZ_LANE_DECL
    const int wave = (int)MZ_UNIFORM(threadIdx.x >> 6); /* wave-uniform, and known to be: MZ_WAVE_INDEX above */
    mz_inflate_lds *L = (mz_inflate_lds *)(smem + MZ_CRC_TAB_BYTES + wave * MZ_LDS_STRIDE);
    uint8_t *const rec = a.rec + ((size_t)blockIdx.x * MZ_WAVES_PER_WG + (size_t)wave) * MZ_REC_BYTES;
    for (;;) {
        uint32_t e;
        MZ_WAVE_FETCH_ADD(e, a.counter);
        if (e >= a.n) break;
        mz_inflate_one_block(a.in, a.in_len, MZ_UNIFORM(a.bits[e]), MZ_UNIFORM(a.pos[e]), a.mode, a.out, a.ptr, L, crc_tab, a.tabs, rec,
                             a.res + 4u * (size_t)e);
    }
}

struct ParJumpArgs {
    uint8_t *out;
    uint32_t *ptr;
    uint32_t hist, total;
    uint32_t *changed;
};
// one round of pointer jumping over the source map: ptr[i] = its ancestor up to four links up (in place: whatever a racing
// lane reads is an ancestor of i either way; four dependent loads per lane instead of one cut the rounds of a window from
// 14 to 5 for the same bytes moved per round).  Roots are literals (ptr[i] == i) and the history in front of the window
// (ptr[i] < hist).
__global__ __launch_bounds__(256) void k_ptr_jump(ParJumpArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    uint32_t any = 0;
    for (uint64_t i = (uint64_t)a.hist + (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.total; i += stride) {
        uint32_t p = a.ptr[i];
        if (p == (uint32_t)i || p < a.hist) continue;
        uint32_t moved = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t q = a.ptr[p];
            if (q == p) break;
            p = q;
            moved = 1;
            if (p < a.hist) break;
        }
        if (moved) {
            a.ptr[i] = p;
            any = 1;
        }
    }
    if (__any(any) && (threadIdx.x & 63) == 0) *a.changed = 1u;
}
// ... and the bytes: every position fetches its root's (a root's byte never changes, so this is in place too)
__global__ __launch_bounds__(256) void k_ptr_gather(ParJumpArgs a) {
    const uint64_t stride = (uint64_t)gridDim.x * 256u;
    for (uint64_t i = (uint64_t)a.hist + (uint64_t)blockIdx.x * 256u + threadIdx.x; i < a.total; i += stride) {
        const uint32_t p = a.ptr[i];
        if (p != (uint32_t)i) a.out[i] = a.out[p];
    }
}

// the coded pieces of one segment (mz_stream_zlib WRITE: 128 pieces of an 8 MiB segment, each in its own slot of worst-case size)
// moved together, a workgroup per piece, so that the host fetches them in ONE copy: 128 copies and waits of ~23 KB were 2 of a
// segment's 6 ms.  dst_off[] = the prefix sums of len[], made by the host from the lengths it has just read.
struct PiecePackArgs {
    const uint8_t *src;
    const uint64_t *src_off;
    const uint32_t *len;
    uint8_t *dst;
    const uint64_t *dst_off;
    uint32_t n;
};
__global__ __launch_bounds__(256) void k_pack_pieces(PiecePackArgs a) {
    for (uint32_t i = blockIdx.x; i < a.n; i += gridDim.x) {
        const uint8_t *s = a.src + a.src_off[i];
        uint8_t *d = a.dst + a.dst_off[i];
        const uint32_t n = a.len[i];
        /* the slots start on 16-byte boundaries, the places in the packed stream anywhere: dwords from the source, bytes to the target */
        for (uint32_t k = 4u * threadIdx.x; k < n; k += 1024u) {
            if (k + 4u <= n) {
                const uint32_t v = *(const uint32_t *)(s + k);
                d[k] = (uint8_t)v;
                d[k + 1u] = (uint8_t)(v >> 8);
                d[k + 2u] = (uint8_t)(v >> 16);
                d[k + 3u] = (uint8_t)(v >> 24);
            } else {
                for (uint32_t j = k; j < n; j++) d[j] = s[j];
            }
        }
    }
}

#include "mzhip_runtime.inc"
