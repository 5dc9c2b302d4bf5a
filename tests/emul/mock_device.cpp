// mock_device.cpp -- TEST INFRASTRUCTURE: the host-buffer entry points of include/mzhip.h implemented on the 64-lane
// HOST EMULATION of the device cores (emul.cpp), so that the C shims (shim_zlib.c, shim_lzma.c, shim_crc32.c,
// shim_autoprime.c -- the code that mirrors mz_strm_zlib.c / mz_strm_lzma.c call for call) can be exercised behind the
// reference's unmodified zip layer in a container WITHOUT a GPU.  tests/emul/Makefile links it into
// tests/emul/_build/libmockdrop.so, which only tests/test_shims_emul.py loads.  It is never part of libmzhip.so: the
// product has no CPU path, and a GPU box runs the same tests against the real device (tests/test_gpu_dropin.py,
// test_gpu_wrappers.py).  What it cannot cover: the prime caches and the .xz writer, whose host code lives in
// mzhip_kernels.hip.
#include <atomic>

#include "emul.cpp"

#include "../../include/mzhip.h"

#define MOCK_API extern "C" __attribute__((visibility("default")))

MOCK_API int32_t mzhip_device_count(void) { return 1; }
MOCK_API int32_t mzhip_init(int32_t) { return 0; }
MOCK_API const char *mzhip_last_error(void) { return ""; }
MOCK_API const char *mzhip_version(void) { return "mock (host emulation of the device cores)"; }

static int32_t mock_inflate_whole(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                                  uint32_t *in_used, uint32_t *crc, uint32_t *adler) {
    uint32_t ol = 0, iu = 0, k = 0;
    const uint8_t dummy = 0;
    const int32_t st = emul_inflate(in ? in : &dummy, in_len, out, out_cap, &ol, &iu, &k);
    if (out_len) *out_len = ol;
    if (in_used) *in_used = iu;
    if (crc) *crc = k;
    if (adler) *adler = emul_adler32(out, ol);
    return st;
}
static int32_t mock_inflate_resume(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                   const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out, uint32_t *out_len,
                                   uint32_t *in_used, uint32_t *crc) {
    uint32_t ol = 0, iu = 0, k = 0;
    const uint8_t dummy = 0;
    const int32_t st = emul_inflate_resume(in ? in : &dummy, in_len, buf, buf_cap, (const uint32_t *)state_in, (uint32_t *)state_out, &ol,
                                           &iu, &k);
    if (out_len) *out_len = ol;
    if (in_used) *in_used = iu;
    if (crc) *crc = k;
    return st;
}
// ... with the CRCs of the new bytes in the caller's pieces (the device does this with one k_crc32_batch launch)
static std::atomic<int> g_mock_seg_calls{0};
MOCK_API int mzmock_seg_calls(void) { return g_mock_seg_calls; }
static int32_t mock_inflate_window(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                   const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out, uint32_t *out_len,
                                   uint32_t *in_used, uint32_t *crc, uint32_t *adler, uint32_t seg_first, uint32_t seg_stride,
                                   uint32_t *seg_crc, uint32_t seg_cap, uint32_t *nseg) {
    const uint32_t hist = state_in ? state_in->out_pos : 0u;
    uint32_t ol = 0;
    const int32_t st = mock_inflate_resume(in, in_len, buf, buf_cap, state_in, state_out, &ol, in_used, crc);
    if (out_len) *out_len = ol;
    if (nseg) *nseg = 0;
    if (adler) *adler = ol > hist ? emul_adler32(buf + hist, ol - hist) : 1u;
    if (seg_stride && seg_crc && ol > hist) {
        uint32_t pos = hist, n = 0;
        const uint32_t first = seg_first < ol - hist ? seg_first : ol - hist;
        const uint32_t count = (first ? 1u : 0u) + (ol - hist - first + seg_stride - 1) / seg_stride;
        if (count <= seg_cap) {
            if (first) {
                seg_crc[n++] = emul_crc32(buf + pos, first);
                pos += first;
            }
            while (pos < ol) {
                const uint32_t k = ol - pos < seg_stride ? ol - pos : seg_stride;
                seg_crc[n++] = emul_crc32(buf + pos, k);
                pos += k;
            }
            if (nseg) *nseg = n;
            g_mock_seg_calls++;
        }
    }
    return st;
}
// include/mzhip.h: a whole entry (no states) or one window of it, one entry point
MOCK_API int32_t mzhip_inflate_host_a(const mzhip_inflate_host_args *ap) {
    if (!ap || ap->size < offsetof(mzhip_inflate_host_args, buf) + sizeof(void *) || ap->size > 4096u || (ap->size & 3u)) return -102;
    mzhip_inflate_host_args a;
    memset(&a, 0, sizeof(a));
    memcpy(&a, ap, ap->size < sizeof(a) ? ap->size : sizeof(a));
    if (!a.state_in && !a.state_out) {
        if (a.nseg) *a.nseg = 0;
        return mock_inflate_whole(a.in, a.in_len, a.buf, a.buf_cap, a.out_len, a.in_used, a.crc, a.adler);
    }
    return mock_inflate_window(a.in, a.in_len, a.buf, a.buf_cap, a.state_in, a.state_out, a.out_len, a.in_used, a.crc, a.adler, a.seg_first,
                               a.seg_stride, a.seg_crc, a.seg_cap, a.nseg);
}

// one window of one large entry, every block by a "wave" of its own (inflate_parallel.inc): the same orchestration as the
// device's, its three steps as loops over the emulated device functions
#ifndef MZ_LARGE_WINDOW
#define MZ_LARGE_WINDOW (2u << 20)
#define MZ_LARGE_SHOW_MAX (1u << 20)
#define MZ_LARGE_MIN_ROOM (128u << 10)
#endif
#include "inflate_parallel.inc"
namespace {
struct MockParBackend {
    const uint8_t *in;
    uint32_t in_len;
    uint8_t *buf;
    std::vector<uint32_t> ptr;
    int32_t find(uint32_t b0, uint32_t b1, std::vector<uint32_t> &c) {
        c.resize(1u << 16);
        uint32_t n = emul_find_blocks(in, in_len, b0, b1, c.data(), (uint32_t)c.size());
        if (n > c.size()) {
            c.resize(n);
            n = emul_find_blocks(in, in_len, b0, b1, c.data(), (uint32_t)c.size());
        }
        c.resize(n);
        return 0;
    }
    int32_t parse(uint32_t mode, const uint32_t *bits, const uint32_t *pos, uint32_t n, MzParBlock *res) {
        for (uint32_t i = 0; i < n; i++) {
            uint32_t r[4];
            emul_inflate_block(in, in_len, bits[i], pos[i], mode, buf, ptr.data(), r);
            res[i].status = (int32_t)r[0];
            res[i].end_bit = r[1];
            res[i].out_end = r[2];
            res[i].last = r[3];
        }
        return 0;
    }
    int32_t resolve(uint32_t hist, uint32_t total, uint32_t *rounds) {
        for (uint32_t changed = 1; changed; (*rounds)++) {
            changed = 0;
            for (uint32_t i = hist; i < total; i++) {
                const uint32_t p = ptr[i];
                if (p >= hist && ptr[p] != p) {
                    ptr[i] = ptr[p];
                    changed = 1;
                }
            }
        }
        for (uint32_t i = hist; i < total; i++) buf[i] = buf[ptr[i]];
        return 0;
    }
};
} // namespace
// (mzhip_window_alloc / _free: the product's pool over the stand-in's hipHostMalloc, which MZMOCK_NO_PINNED makes fail)
// one large entry where it lies ("device-resident": the mock's device memory is host memory), inflate_parallel.inc's
// mz_large_entry over the emulated device functions; MZMOCK_LARGE_WINDOW / _SHOW shrink the windows for the tests
namespace {
struct MockLargeBackend {
    const uint8_t *in;
    uint8_t *out;
    int32_t window(uint32_t vbyte, uint32_t vlen, uint32_t obase, uint32_t buf_cap, uint32_t start_bit, uint32_t hist, MzParWindow *w) {
        MockParBackend be;
        be.in = in + vbyte;
        be.in_len = vlen;
        be.buf = out + obase;
        be.ptr.assign((size_t)buf_cap + 64, 0u);
        return mz_parallel_window(be, vlen, buf_cap, start_bit, hist, w);
    }
    int32_t serial(uint32_t vbyte, uint32_t vlen, uint32_t obase, uint32_t buf_cap, uint32_t hdr_bit, uint32_t bit, uint32_t hist, uint32_t flags,
                   MzSerialResult *o) {
        uint32_t a[4] = {hdr_bit, bit, hist, flags}, b[4] = {0, 0, 0, 0}, crc = 0;
        o->status = emul_inflate_resume(in + vbyte, vlen, out + obase, buf_cap, a, b, &o->out_len, &o->in_used, &crc);
        o->hdr_bit = b[0];
        o->bit = b[1];
        o->out_pos = b[2];
        o->ok = b[3] & 1u;
        return 0;
    }
};
} // namespace
static uint32_t g_mock_large[3];
MOCK_API void mzmock_large_stats(uint32_t *v) { memcpy(v, g_mock_large, sizeof(g_mock_large)); }
MOCK_API int32_t mzhip_inflate_large(const void *d_in, uint32_t in_len, void *d_out, uint32_t out_cap, uint32_t *out_len, uint32_t *in_used,
                                     uint32_t *crc, int32_t *status, void *) {
    MockLargeBackend be;
    be.in = (const uint8_t *)d_in;
    be.out = (uint8_t *)d_out;
    MzLargeResult r;
    const int32_t rc = mz_large_entry(be, in_len, out_cap, &r);
    if (rc) return rc;
    g_mock_large[0] = r.par_windows;
    g_mock_large[1] = r.par_blocks;
    g_mock_large[2] = r.serial_calls;
    if (out_len) *out_len = r.out_len;
    if (in_used) *in_used = r.in_used;
    if (crc) *crc = emul_crc32(be.out, r.out_len);
    if (status) *status = r.status;
    return 0;
}
static std::atomic<int> g_mock_par_blocks{0}, g_mock_par_calls{0};
MOCK_API int mzmock_par_blocks(void) { return g_mock_par_blocks; }
MOCK_API int mzmock_par_calls(void) { return g_mock_par_calls; }
MOCK_API int32_t mzhip_inflate_parallel_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                             const mzhip_inflate_state *state_in, mzhip_inflate_state *state_out, uint32_t *out_len,
                                             uint32_t *blocks, uint32_t *ended, uint32_t *crc, uint32_t *adler, uint32_t seg_first,
                                             uint32_t seg_stride, uint32_t *seg_crc, uint32_t seg_cap, uint32_t *nseg) {
    const uint32_t hist = state_in ? state_in->out_pos : 0u;
    if (crc) *crc = 0u;
    if (adler) *adler = 1u;
    const uint32_t start = state_in && (state_in->flags & 1u) ? state_in->hdr_bit : 0u;
    if (nseg) *nseg = 0;
    if (blocks) *blocks = 0;
    if (ended) *ended = 0;
    if (out_len) *out_len = hist;
    if (state_in && (state_in->flags & 1u) && state_in->bit != state_in->hdr_bit) return 0; /* (inside a block: stream order) */
    if (hist > buf_cap) return -102;
    MockParBackend be;
    be.in = in;
    be.in_len = in_len;
    be.buf = buf;
    be.ptr.assign((size_t)buf_cap + 64, 0u);
    MzParWindow w;
    const int32_t rc = mz_parallel_window(be, in_len, buf_cap, start, hist, &w);
    if (rc) return rc;
    g_mock_par_calls++;
    g_mock_par_blocks += (int)w.blocks;
    if (blocks) *blocks = w.blocks;
    if (ended) *ended = w.ended;
    if (out_len) *out_len = w.total;
    if (state_out) {
        state_out->hdr_bit = state_out->bit = w.next_bit;
        state_out->out_pos = w.total;
        state_out->flags = 1u;
    }
    const uint32_t ol = w.total;
    if (crc && ol > hist) *crc = emul_crc32(buf + hist, ol - hist);
    if (adler && ol > hist) *adler = emul_adler32(buf + hist, ol - hist);
    if (seg_stride && seg_crc && ol > hist) {
        uint32_t pos = hist, n = 0;
        const uint32_t first = seg_first < ol - hist ? seg_first : ol - hist;
        const uint32_t count = (first ? 1u : 0u) + (ol - hist - first + seg_stride - 1) / seg_stride;
        if (count <= seg_cap) {
            if (first) {
                seg_crc[n++] = emul_crc32(buf + pos, first);
                pos += first;
            }
            while (pos < ol) {
                const uint32_t k = ol - pos < seg_stride ? ol - pos : seg_stride;
                seg_crc[n++] = emul_crc32(buf + pos, k);
                pos += k;
            }
            if (nseg) *nseg = n;
        }
    }
    return 0;
}

// one stream segment = 64 KiB pieces, every piece but the last closed on a byte boundary (as mzhip_deflate_host_a does)
static int32_t mock_deflate_segment(const uint8_t *in, uint32_t in_len, uint32_t final, int32_t level, int32_t window_log2,
                                    uint8_t *out, uint32_t out_cap, uint32_t *out_len, uint32_t *crc, uint32_t *adler) {
    const uint32_t piece = 64u << 10;
    const uint32_t np = in_len ? (in_len + piece - 1) / piece : 1u;
    uint32_t total = 0, k = 0, ad = 1;
    emul_deflate_window((uint32_t)window_log2);
    const uint8_t dummy = 0;
    for (uint32_t i = 0; i < np; i++) {
        const uint32_t off = in_len ? i * piece : 0u, len = in_len - off < piece ? in_len - off : piece;
        uint32_t ol = 0, pc = 0;
        const int32_t st = ((level >= 0 && level <= 3) ? emul_deflate : (level >= 7) ? emul_deflate_best : emul_deflate_lazy)(in_len ? in + off : &dummy, len, out + total, out_cap - total, (i + 1 == np && final) ? 1u : 0u, &ol, &pc);
        if (st) return st;
        total += ol;
        k = i == 0 ? pc : mzhip_crc32_combine_host(k, pc, len);
        if (adler) ad = mzhip_adler32_combine_host(ad, emul_adler32(in_len ? in + off : &dummy, len), len);
    }
    if (out_len) *out_len = total;
    if (crc) *crc = k;
    if (adler) *adler = ad;
    return 0;
}
MOCK_API int32_t mzhip_deflate_host_a(const mzhip_deflate_host_args *ap) {
    if (!ap || ap->size < offsetof(mzhip_deflate_host_args, out) + sizeof(void *) || ap->size > 4096u || (ap->size & 3u)) return -102;
    mzhip_deflate_host_args a;
    memset(&a, 0, sizeof(a));
    memcpy(&a, ap, ap->size < sizeof(a) ? ap->size : sizeof(a));
    return mock_deflate_segment(a.in, a.in_len, a.final, a.level, a.window_log2 ? a.window_log2 : 15, a.out, a.out_cap, a.out_len, a.crc, a.adler);
}

MOCK_API int32_t mzhip_lzma_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                                 uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    uint32_t ol = 0, iu = 0, k = 0;
    const int32_t st = emul_lzma(in, in_len, out, out_cap, max_out, &ol, &iu, &k);
    if (out_len) *out_len = ol;
    if (in_used) *in_used = iu;
    if (crc) *crc = k;
    return st;
}
MOCK_API uint32_t mzhip_lzma_model_bytes(void) { return emul_lzma_model_u16() * 2u; }
MOCK_API uint32_t mzhip_lzma_encode_history_bytes(void) { return emul_lzma_encode_history_bytes(); }
static std::atomic<int> g_mock_lzma_windows{0};
MOCK_API int mzmock_lzma_windows(void) { return g_mock_lzma_windows; }
MOCK_API int32_t mzhip_lzma_resume_host(const uint8_t *in, uint32_t in_len, uint8_t *buf, uint32_t buf_cap,
                                        const mzhip_lzma_state *state_in, mzhip_lzma_state *state_out, void *model,
                                        uint32_t *out_len, uint32_t *in_used) {
    uint32_t ol = 0, iu = 0;
    g_mock_lzma_windows++;
    const int32_t st = emul_lzma_resume(in, in_len, buf, buf_cap, (const uint32_t *)state_in, (uint32_t *)state_out,
                                        (uint16_t *)model, &ol, &iu);
    if (out_len) *out_len = ol;
    if (in_used) *in_used = iu;
    return st;
}
static std::atomic<int> g_mock_lzma2_windows{0};
MOCK_API int mzmock_lzma2_windows(void) { return g_mock_lzma2_windows; }
MOCK_API int32_t mzhip_lzma2_run_host(const mzhip_lzma2_run_args *ap) {
    if (!ap || ap->size < offsetof(mzhip_lzma2_run_args, in_used) + sizeof(void *)) return -102;
    mzhip_lzma2_run_args a;
    memset(&a, 0, sizeof(a));
    memcpy(&a, ap, ap->size < sizeof(a) ? ap->size : sizeof(a));
    if (!a.model || !a.state_in || !a.state_out || !a.buf || a.state_in->out_pos > a.buf_cap) return -102;
    uint32_t ol = 0, iu = 0;
    g_mock_lzma2_windows++;
    const uint8_t dummy = 0;
    const int32_t st = emul_lzma2_run(a.in ? a.in : &dummy, a.in_len, a.buf, a.buf_cap, (const uint32_t *)a.state_in, (uint32_t *)a.state_out,
                                      (uint16_t *)a.model, &ol, &iu);
    if (a.out_len) *a.out_len = ol;
    if (a.in_used) *a.in_used = iu;
    return st;
}
static std::atomic<int> g_mock_lzma_segments{0};
MOCK_API int mzmock_lzma_segments(void) { return g_mock_lzma_segments; }
MOCK_API int32_t mzhip_lzma_encode_resume_host(const uint8_t *in, uint32_t in_len, uint32_t skip_blocks, uint32_t last, int32_t preset,
                                               const mzhip_lzma_enc_state *state_in, mzhip_lzma_enc_state *state_out, void *model,
                                               uint8_t *out, uint32_t out_cap, uint32_t *out_len) {
    uint32_t ol = 0;
    g_mock_lzma_segments++;
    emul_set_far_depth(MZ_LZE_DEPTH_FOR_PRESET(preset));
    const int32_t st = emul_lzma_encode_resume(in, in_len, skip_blocks, last, (preset >= 0 && preset <= 3) ? 1u : MZ_DEF_WAYS_BEST,
                                               (const uint32_t *)state_in, (uint32_t *)state_out, (uint16_t *)model, out, out_cap, &ol);
    if (out_len) *out_len = ol;
    return st;
}
MOCK_API int32_t mzhip_xz_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, int64_t max_out,
                               uint32_t *out_len, uint32_t *in_used, uint32_t *crc) {
    uint32_t ol = 0, iu = 0, k = 0;
    const int32_t st = emul_xz(in, in_len, out, out_cap, max_out, &ol, &iu, &k);
    if (out_len) *out_len = ol;
    if (in_used) *in_used = iu;
    if (crc) *crc = k;
    return st;
}
MOCK_API int32_t mzhip_lzma_encode_host(const uint8_t *in, uint32_t in_len, uint8_t *out, uint32_t out_cap, uint32_t *out_len,
                                        uint32_t *crc) {
    const uint8_t dummy = 0;
    return emul_lzma_encode(in ? in : &dummy, in_len, 0u, out, out_cap, out_len, crc);
}
MOCK_API int32_t mzhip_xz_encode_host(const uint8_t *, uint32_t, uint8_t *, uint32_t, uint32_t *, uint32_t *) {
    return MZHIP_STATUS_UNSUPPORTED; /* the .xz container is laid out by host code inside mzhip_kernels.hip */
}
MOCK_API int32_t mzhip_lzma_encode_host_preset(const uint8_t *in, uint32_t in_len, int32_t preset, uint8_t *out, uint32_t out_cap,
                                               uint32_t *out_len, uint32_t *crc) {
    const uint8_t dummy = 0;
    emul_set_far_depth(MZ_LZE_DEPTH_FOR_PRESET(preset));
    return emul_lzma_encode_ways(in ? in : &dummy, in_len, 0u, (preset >= 0 && preset <= 3) ? 1u : MZ_DEF_WAYS_BEST, out, out_cap, out_len,
                                 crc);
}
MOCK_API int32_t mzhip_xz_encode_host_preset(const uint8_t *, uint32_t, int32_t, uint8_t *, uint32_t, uint32_t *, uint32_t *) {
    return MZHIP_STATUS_UNSUPPORTED;
}
MOCK_API int32_t mzhip_xz_encode_block_host(const uint8_t *, uint32_t, int32_t, int32_t, uint8_t *, uint32_t, uint32_t *, uint32_t *,
                                            uint64_t *) {
    return MZHIP_STATUS_UNSUPPORTED;
}
MOCK_API int32_t mzhip_xz_encode_finish_host(const uint64_t *, const uint64_t *, uint32_t, uint8_t *, uint32_t, uint32_t *) {
    return MZHIP_STATUS_UNSUPPORTED;
}

static std::atomic<int> g_mock_crc_calls{0}; // checksum calls that would have been a launch on the device
MOCK_API int mzmock_crc_host_calls(void) { return g_mock_crc_calls; }
MOCK_API uint32_t mzhip_crc32_host(uint32_t value, const uint8_t *buf, size_t size) {
    uint32_t v = value;
    if (size >= MZHIP_CRC_HOST_BELOW) g_mock_crc_calls++;
    for (size_t pos = 0; pos < size;) { /* the emulated kernel takes 32-bit lengths */
        const size_t n = size - pos < ((size_t)1 << 30) ? size - pos : ((size_t)1 << 30);
        v = emul_crc32_super(buf + pos, (uint32_t)n, v);
        pos += n;
    }
    return v;
}
extern "C" uint32_t mzhip_adler32_combine(uint32_t a, uint32_t b, uint64_t len_b) { return mzhip_adler32_combine_host(a, b, len_b); }

// ---- the READ-side prime cache: the PRODUCT's code (minizip-ng_amd/csrc/mzhip_prime.inc -- generations, the three-lane decode
// pipeline, look-ups, the windows shim_autoprime.c rolls over large archives with) over a synchronous stand-in for the HIP
// runtime: "device memory" is host memory, a "stream" completes every operation when it is queued, the batch launchers run the
// emulated device cores entry by entry.  What the stand-in cannot show is overlap and device failures; what it does show is every
// line of host logic between the unmodified zip layer and the kernels, in a container without a GPU and under the sanitizers.
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
namespace {
typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
thread_local int t_mock_dev = 0;
std::atomic<long> g_mock_dev_allocs{0}, g_mock_host_allocs{0}; // outstanding (tests: nothing leaks)
hipError_t hipGetDevice(int *d) { *d = t_mock_dev; return hipSuccess; }
hipError_t hipSetDevice(int d) { t_mock_dev = d; return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "mock"; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n ? n : 1); if (*p) g_mock_dev_allocs++; return *p ? hipSuccess : 2; }
hipError_t hipFree(void *p) { if (p) g_mock_dev_allocs--; free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) {
    if (getenv("MZMOCK_NO_PINNED")) { *p = nullptr; return 2; }
    *p = malloc(n ? n : 1);
    if (*p) g_mock_host_allocs++;
    return *p ? hipSuccess : 2;
}
hipError_t hipHostFree(void *p) { if (p) g_mock_host_allocs--; free(p); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = malloc(1); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, int, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, int) { if (n) memmove(d, s, n); return hipSuccess; }
thread_local char g_err[256] = "";
int32_t fail(const char *what, hipError_t) { snprintf(g_err, sizeof(g_err), "%s failed (mock)", what); return -104; }
#define HIP_TRY(expr)                          \
    do {                                       \
        hipError_t _e = (expr);                \
        if (_e != hipSuccess) return fail(#expr, _e); \
    } while (0)
struct DeviceCtx { int unused; };
DeviceCtx g_mock_ctx;
constexpr int kMaxDevices = 16;
int32_t ctx_for_current(DeviceCtx **out) { *out = &g_mock_ctx; return 0; }
struct Scratch {
    void *p = nullptr;
    ~Scratch() { if (p) (void)hipFree(p); }
};
// one launch per codec = a loop over the emulated cores
int32_t lzma_family_batch(int xz, const void *d_in, const uint64_t *in_off, const uint32_t *in_len, void *d_out, const uint64_t *out_off,
                          const uint32_t *out_cap, const int64_t *max_out, uint32_t n, uint32_t *out_len, uint32_t *in_used, uint32_t *crc,
                          int32_t *status, hipStream_t) {
    for (uint32_t i = 0; i < n; i++)
        status[i] = (xz ? emul_xz : emul_lzma)((const uint8_t *)d_in + in_off[i], in_len[i], (uint8_t *)d_out + out_off[i], out_cap[i],
                                               max_out ? max_out[i] : -1, &out_len[i], &in_used[i], &crc[i]);
    return 0;
}
} // namespace
MOCK_API long mzmock_outstanding_allocs(void) { return g_mock_dev_allocs.load() + g_mock_host_allocs.load(); }
MOCK_API int32_t mzhip_bind_thread_near_device(int32_t, int32_t) { return 0; }
static std::atomic<int> g_mock_batch_launches{0};
MOCK_API int mzmock_batch_launches(void) { return g_mock_batch_launches.load(); }
MOCK_API int32_t mzhip_inflate_batch(const void *d_in, const uint64_t *in_off, const uint32_t *in_len, void *d_out, const uint64_t *out_off,
                                     const uint32_t *out_cap, uint32_t n, uint32_t *out_len, uint32_t *in_used, uint32_t *crc,
                                     int32_t *status, void *) {
    g_mock_batch_launches++;
    const uint8_t dummy = 0;
    for (uint32_t i = 0; i < n; i++)
        status[i] = emul_inflate(in_len[i] ? (const uint8_t *)d_in + in_off[i] : &dummy, in_len[i], (uint8_t *)d_out + out_off[i], out_cap[i],
                                 &out_len[i], &in_used[i], &crc[i]);
    return 0;
}
MOCK_API int32_t mzhip_crc32_batch(const void *d_buf, const uint64_t *off, const uint32_t *len, uint32_t n, const uint32_t *init, uint32_t *crc,
                                   void *) {
    for (uint32_t i = 0; i < n; i++) crc[i] = emul_crc32_super((const uint8_t *)d_buf + off[i], len[i], init ? init[i] : 0u);
    return 0;
}
MOCK_API int32_t mzhip_sha_batch(const void *d_buf, const uint64_t *off, const uint32_t *len, uint32_t n, uint32_t algorithm, void *d_digest,
                                 void *) {
    for (uint32_t i = 0; i < n; i++) {
        uint8_t dg[64];
        memset(dg, 0, sizeof(dg));
        emul_sha((const uint8_t *)d_buf + off[i], len[i], (int)algorithm, dg);
        memcpy((uint8_t *)d_digest + 32 * (size_t)i, dg, 32);
    }
    return 0;
}
#define mzhip_prime_any mzhip_prime_any_real
#include "mzhip_prime.inc"
#undef mzhip_prime_any
extern "C" int32_t mzhip_take_crc_fault(void) { return 0; }
extern "C" void mzhip_thread_use_device(int32_t) {}
MOCK_API uint64_t mzhip_crc_faults(void) { return 0; }
extern "C" int32_t mzhip_wprime_track(int32_t, int64_t *, int64_t, const uint8_t *, int32_t, uint32_t *, int32_t *have_crc,
                                      const uint8_t **) {
    *have_crc = 0;
    return 0;
}
// "is any archive primed?": MZMOCK_PRIME_ANY=1 says yes even when nothing is, so that the streams' short first pull and the
// ordinary decode behind it run on the CPU as well
extern "C" int32_t mzhip_prime_any(void) {
    static const int on = getenv("MZMOCK_PRIME_ANY") != nullptr;
    return on || mzhip_prime_any_real();
}
extern "C" int32_t mzhip_wprime_result(int32_t, int64_t, int64_t, const uint8_t **, const uint8_t **, uint32_t *) { return 0; }
