/* wave.h -- wave64 SIMT vocabulary for the gfx950 kernels.
 *
 * The codec kernels are written "wave-synchronously": one 64-lane wavefront
 * owns one ZIP entry; code is either wave-uniform (bit cursor, block state,
 * chain walk -- lands on the scalar unit) or per-lane (candidate token decode,
 * literal scatter, cooperative match copy, CRC folding).
 *
 * The same source compiles two ways:
 *   - hipcc --offload-arch=gfx950 : the product.  A per-lane region is just
 *     straight-line code executed by all 64 lanes; cross-lane traffic uses
 *     v_readlane / ballot / LDS.
 *   - g++ -DMZHIP_HOST_EMUL       : a debugging emulation used ONLY by the CPU
 *     test-suite (tests/test_kernel_emul.py); a per-lane region becomes a loop
 *     over lane = 0..63 and per-lane variables become 64-element arrays.  It
 *     lets the kernels' control flow be checked against the oracle in a
 *     container without a GPU.  It is never linked into libmzhip.so.
 *
 * Rule that keeps the two modes equivalent: inside one MZ_LANES region a lane
 * only reads memory that no other lane writes in that same region; regions
 * that communicate through LDS / global memory are separated by MZ_WAVE_SYNC().
 */
#ifndef MZHIP_WAVE_H
#define MZHIP_WAVE_H

#include <stdint.h>

#if defined(MZHIP_HOST_EMUL)

#include <string.h>
#define MZ_DEV static inline
#define MZ_LANE_DECL
#define MZ_NOUNROLL
#define MZ_LANES for (int lane = 0; lane < 64; ++lane)
#define PV(type, name) type name[64]
#define PV2(type, name, n) type name[64][n] /* small per-lane array */
#define MZ_DEV_NOINLINE static
#define PVIN(type, name) const type (&name)[64] /* a per-lane value of the caller, as a function parameter */
typedef uintptr_t mz_lds_handle;
typedef uintptr_t mz_glb_handle;
#define MZ_LDS_HANDLE(p) ((uintptr_t)(p))
#define MZ_GLB_HANDLE(p) ((uintptr_t)(p))
#define MZ_LDS_FROM(T, h) ((T *)(h))
#define MZ_GLB_FROM(T, h) ((T *)(h))
#define P(name) name[lane]
#define MZ_READLANE(name, idx) (name[(idx)])
#define MZ_WRITELANE(name, idx, val) (name[(idx)] = (val)) /* one wave-uniform value into lane idx */
#define MZ_WRITELANE_S(name, idx, val) (name[(idx)] = (val)) /* the same where idx and val are known to live in scalar registers */
#define MZ_RANK_BELOW(m) ((uint32_t)__builtin_popcountll((uint64_t)(m) & ((1ull << lane) - 1ull))) /* set bits of the wave-uniform mask m below this lane (v_mbcnt) */
#define MZ_UNIFORM(x) (x)
#define MZ_UNIFORM64(x) (x)
#define MZ_WAVE_SYNC() ((void)0)
#define MZ_CHASE_FENCE() ((void)0)
#define MZ_BALLOT(dst, cond)                         \
    do {                                             \
        uint64_t _bal_acc = 0;                       \
        for (int lane = 0; lane < 64; ++lane)        \
            if (cond) _bal_acc |= 1ull << lane;      \
        (dst) = _bal_acc;                            \
    } while (0)
#define MZ_WAVE_XOR(dst, name)                       \
    do {                                             \
        uint32_t _xor_acc = 0;                       \
        for (int lane = 0; lane < 64; ++lane)        \
            _xor_acc ^= name[lane];                  \
        (dst) = _xor_acc;                            \
    } while (0)
#define MZ_WAVE_SUM(dst, name)                       \
    do {                                             \
        uint32_t _sum_acc = 0;                       \
        for (int lane = 0; lane < 64; ++lane)        \
            _sum_acc += name[lane];                  \
        (dst) = _sum_acc;                            \
    } while (0)
#define MZ_LDS_ATOMIC_INC(ptr) (++*(ptr))
#define MZ_LDS_ATOMIC_OR(ptr, v) (*(ptr) |= (v))
#define MZ_LDS_ATOMIC_MAX(ptr, v) (*(ptr) = (*(ptr) > (v)) ? *(ptr) : (v))
#define MZ_LDS_ATOMIC_AND(ptr, v) (*(ptr) &= (v))
/* dst[lane] = src[idx(lane)] -- a cross-lane gather (ds_bpermute on the device) */
#define MZ_GATHER(dst, src, idx_expr)                                \
    do {                                                             \
        uint32_t _gt[64];                                            \
        for (int lane = 0; lane < 64; ++lane)                        \
            _gt[lane] = src[(uint32_t)(idx_expr) & 63u];             \
        for (int lane = 0; lane < 64; ++lane)                        \
            dst[lane] = _gt[lane];                                   \
    } while (0)
/* same with a byte index (4 * lane), the native form of ds_bpermute */
#define MZ_GATHER4(dst, src, byteidx_expr)                           \
    do {                                                             \
        uint32_t _gt[64];                                            \
        for (int lane = 0; lane < 64; ++lane)                        \
            _gt[lane] = src[((uint32_t)(byteidx_expr) >> 2) & 63u];  \
        for (int lane = 0; lane < 64; ++lane)                        \
            dst[lane] = _gt[lane];                                   \
    } while (0)
/* inclusive prefix sum across the wave */
#define MZ_INCL_SCAN(dst, src)                                       \
    do {                                                             \
        uint32_t _sc = 0;                                            \
        for (int lane = 0; lane < 64; ++lane) {                      \
            _sc += src[lane];                                        \
            dst[lane] = _sc;                                         \
        }                                                            \
    } while (0)
/* inclusive prefix MINIMUM inside every row of 16 lanes: lanes 15, 31, 47, 63 end up with their row's minimum */
#define MZ_ROW16_PMIN(dst, src)                                      \
    do {                                                             \
        for (int _r = 0; _r < 64; _r += 16) {                        \
            uint32_t _m = 0xFFFFFFFFu;                               \
            for (int _k = 0; _k < 16; ++_k) {                        \
                _m = src[_r + _k] < _m ? src[_r + _k] : _m;          \
                dst[_r + _k] = _m;                                   \
            }                                                        \
        }                                                            \
    } while (0)
MZ_DEV uint32_t mz_popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
MZ_DEV uint32_t mz_ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
MZ_DEV uint32_t mz_clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); } /* v != 0 */
MZ_DEV uint32_t mz_brev32(uint32_t v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(v);
}

#else /* ---------------------------------------------------- gfx950 device */

#include <hip/hip_runtime.h>
#define MZ_DEV __device__ __forceinline__
#define MZ_LANE_DECL const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#define MZ_NOUNROLL _Pragma("nounroll")
#define MZ_LANES
#define PV(type, name) type name
#define PV2(type, name, n) type name[n]
/* a function that is NOT inlined gets generic pointers and would address LDS and global memory through flat
 * instructions; so pointers cross the call as integers and are re-made inside in their real address space
 * (infer-address-spaces then rewrites every use) */
#define MZ_DEV_NOINLINE static __device__ __attribute__((noinline))
#define PVIN(type, name) type name
typedef uint32_t mz_lds_handle; /* an LDS address is 32 bits */
typedef uint64_t mz_glb_handle;
#define MZ_LDS_HANDLE(p) ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)(p))
#define MZ_GLB_HANDLE(p) ((uint64_t)(uintptr_t)(p))
/* (arguments reach a function that is not inlined in VECTOR registers, and so do its results: to the compiler they differ from
 * lane to lane, and everything computed from them -- loop counters, branch conditions, base addresses -- would run on the
 * vector unit under exec masks.  Handles and wave-uniform scalars are therefore read back to scalar registers
 * (MZ_UNIFORM / MZ_UNIFORM64) on the callee's first line and on the caller's side of the return.) */
#define MZ_UNIFORM64(x) (((uint64_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)(x) >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x)))
#define MZ_LDS_FROM(T, h) ((T *)(__attribute__((address_space(3))) T *)(uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(h)))
#define MZ_GLB_FROM(T, h) ((T *)(__attribute__((address_space(1))) T *)MZ_UNIFORM64(h))
#define P(name) name
#define MZ_READLANE(name, idx) ((uint32_t)__builtin_amdgcn_readlane((int)(name), (int)(idx)))
#define MZ_WRITELANE(name, idx, val) ((name) = ((uint32_t)lane == (uint32_t)(idx)) ? (uint32_t)(val) : (name)) /* a select, not a branch */
#define MZ_WRITELANE_S(name, idx, val) MZ_WRITELANE(name, idx, val) /* (this clang has no v_writelane builtin: a compare and a select, as above) */
#define MZ_RANK_BELOW(m) ((uint32_t)__builtin_amdgcn_mbcnt_hi((uint32_t)((uint64_t)(m) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)(m), 0u)))
#define MZ_UNIFORM(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
/* orders this wave's memory operations for the compiler; the hardware already
 * executes one wave's LDS (and vector-memory) instructions in issue order. */
#define MZ_WAVE_SYNC()                                        \
    do {                                                      \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); \
        __builtin_amdgcn_wave_barrier();                      \
    } while (0)
/* global memory written by some lanes of this wave is about to be read through other lanes' addresses (K1's step
 * records).  Workgroup scope: the wave's stores have been issued and waited for, and the compiler keeps the order; no
 * cache maintenance is needed, because every lane of a wave sits behind the same vector L1 (gfx950, not in
 * threadgroup-split mode).  The agent-scope pair that stood here first (buffer_wbl2 + buffer_inv) wrote back and
 * invalidated caches three times per window: 7.77 ms instead of 4.75 ms on the 64 KiB probe (profiles/r3/ab_fence.log). */
#define MZ_CHASE_FENCE()                                          \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   \
    } while (0)
#define MZ_BALLOT(dst, cond) ((dst) = __ballot(cond))
/* One device-scope fetch-add per WAVE, result broadcast to every lane.  The wave barriers pin the
 * lane-0 branch: without them LLVM's jump threading fuses this `lane == 0` test with a neighbouring
 * one across the loop back-edge, lane 0 and lanes 1..63 then run different trips of the persistent
 * loop, and readfirstlane hands lanes 1..63 a stale index forever (observed on gfx950, ROCm 7.2).
 * Rule for the kernels: `if (lane == 0)` appears nowhere else -- wave-uniform values are stored by
 * all lanes (same address, same value) instead. */
#define MZ_WAVE_FETCH_ADD(dst, ptr)                                        \
    do {                                                                   \
        uint32_t _fa = 0;                                                  \
        __builtin_amdgcn_wave_barrier();                                   \
        if (lane == 0) _fa = atomicAdd((ptr), 1u);                         \
        __builtin_amdgcn_wave_barrier();                                   \
        (dst) = MZ_UNIFORM(_fa);                                           \
    } while (0)
#define MZ_WAVE_XOR(dst, name)                                          \
    do {                                                                \
        uint32_t _xor_acc = (name);                                     \
        for (int _xo = 32; _xo > 0; _xo >>= 1)                          \
            _xor_acc ^= (uint32_t)__shfl_xor((int)_xor_acc, _xo, 64);   \
        (dst) = MZ_UNIFORM(_xor_acc);                                   \
    } while (0)
#define MZ_LDS_ATOMIC_INC(ptr) atomicAdd((ptr), 1u)
#define MZ_LDS_ATOMIC_OR(ptr, v) atomicOr((ptr), (v))
#define MZ_LDS_ATOMIC_MAX(ptr, v) atomicMax((ptr), (v))
#define MZ_LDS_ATOMIC_AND(ptr, v) atomicAnd((ptr), (v))
#define MZ_GATHER(dst, src, idx_expr) ((dst) = (uint32_t)__shfl((int)(src), (int)(idx_expr), 64))
#define MZ_GATHER4(dst, src, byteidx_expr) ((dst) = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(byteidx_expr), (int)(src)))
/* inclusive wave64 prefix sum on the DPP network: Kogge-Stone inside each row of 16 lanes
 * (row_shr 1,2,4,8), then row_bcast:15 / row_bcast:31 to carry row totals (gfx9 DPP controls). */
#define MZ_DPP(x, ctrl, rmask) ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), (ctrl), (rmask), 0xf, true))
__device__ __forceinline__ uint32_t mz_wave_incl_scan(uint32_t x, int lane) {
    (void)lane;
    x += MZ_DPP(x, 0x111, 0xf); /* row_shr:1, lanes shifted in from outside the row read 0 */
    x += MZ_DPP(x, 0x112, 0xf); /* row_shr:2 */
    x += MZ_DPP(x, 0x114, 0xf); /* row_shr:4 */
    x += MZ_DPP(x, 0x118, 0xf); /* row_shr:8 */
    x += MZ_DPP(x, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */
    x += MZ_DPP(x, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3 */
    return x;
}
/* inclusive prefix minimum inside every row of 16 lanes (row_shr 1, 2, 4, 8; a lane with nothing to its left keeps its own) */
#define MZ_DPP_KEEP(x, ctrl) ((uint32_t)__builtin_amdgcn_update_dpp((int)(x), (int)(x), (ctrl), 0xf, 0xf, false))
__device__ __forceinline__ uint32_t mz_row16_prefix_min(uint32_t x) {
    uint32_t y;
    y = MZ_DPP_KEEP(x, 0x111); x = y < x ? y : x;
    y = MZ_DPP_KEEP(x, 0x112); x = y < x ? y : x;
    y = MZ_DPP_KEEP(x, 0x114); x = y < x ? y : x;
    y = MZ_DPP_KEEP(x, 0x118); x = y < x ? y : x;
    return x;
}
#define MZ_ROW16_PMIN(dst, src) ((dst) = mz_row16_prefix_min(src))
#define MZ_INCL_SCAN(dst, src) ((dst) = mz_wave_incl_scan((src), lane))
#define MZ_WAVE_SUM(dst, name) ((dst) = (uint32_t)__builtin_amdgcn_readlane((int)mz_wave_incl_scan((name), lane), 63))
MZ_DEV uint32_t mz_popc64(uint64_t v) { return (uint32_t)__popcll(v); }
MZ_DEV uint32_t mz_ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
MZ_DEV uint32_t mz_clz64(uint64_t v) { return (uint32_t)__builtin_clzll(v); } /* v != 0 */
MZ_DEV uint32_t mz_brev32(uint32_t v) { return __brev(v); }

#endif

/* Byte-granular moves of 1 / 2 / 4 / 8 bytes at ANY address (LDS or global): gfx950 serves unaligned ds_read/ds_write
 * b16..b64 and global loads / stores natively (hipcc emits the single instruction for these memcpys); the host
 * emulation is plain memcpy. */
MZ_DEV uint64_t mz_ld8(const uint8_t *p) { uint64_t v; __builtin_memcpy(&v, p, 8); return v; }
MZ_DEV uint32_t mz_ld4(const uint8_t *p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }
MZ_DEV uint32_t mz_ld2(const uint8_t *p) { uint16_t v; __builtin_memcpy(&v, p, 2); return v; }
MZ_DEV void mz_st8(uint8_t *p, uint64_t v) { __builtin_memcpy(p, &v, 8); }
MZ_DEV void mz_st4(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
MZ_DEV void mz_st2(uint8_t *p, uint32_t v) { uint16_t w = (uint16_t)v; __builtin_memcpy(p, &w, 2); }

/* status words shared by every kernel; numerically the zlib / MZ_* codes the
 * reference surfaces (mz.h:20-26, mz_strm_zlib.c:186-189). */
#define MZHIP_OK 0
#define MZHIP_DATA_ERROR (-3)
#define MZHIP_BUF_ERROR (-5)
#define MZHIP_OUT_FULL (-200)
#define MZHIP_UNSUPPORTED (-109) /* MZ_SUPPORT_ERROR */

#endif
