// cal_kernels.hip -- kernels of exactly known HBM traffic, for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
// in the access widths K1 .. K6 use (MI355X_MICROARCH.md, HBM: "calibrate on a known byte count in your own access
// pattern").  Not part of the product library: built into profiles/cal/libcal.so by profiles/cal/build.sh and driven by
// profiles/pmc_calibrate.py.  Every kernel touches each byte of an n-byte buffer exactly once, coalesced, grid-stride.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <class T>
__global__ __launch_bounds__(256) void k_cal_read(const T *p, size_t n, T *sink) {
    T acc = T();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == (T)0x5Au && sink) *sink = acc; // (keeps the loads alive; practically never taken)
}
template <class T>
__global__ __launch_bounds__(256) void k_cal_write(T *p, size_t n, T v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
struct u128 {
    uint32_t a, b, c, d;
};
__global__ __launch_bounds__(256) void k_cal_read16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x5A5A5A5Au && sink) *sink = acc;
}
__global__ __launch_bounds__(256) void k_cal_write16(uint4 *p, size_t n, uint32_t v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(v, v, v, v);
}

extern "C" int cal_run(int which, void *buf, size_t bytes, void *sink, hipStream_t s) {
    const unsigned grid = 256u * 32u;
    switch (which) {
    case 0: k_cal_read<uint8_t><<<grid, 256, 0, s>>>((const uint8_t *)buf, bytes, (uint8_t *)sink); break;
    case 1: k_cal_read<uint32_t><<<grid, 256, 0, s>>>((const uint32_t *)buf, bytes / 4, (uint32_t *)sink); break;
    case 2: k_cal_read16<<<grid, 256, 0, s>>>((const uint4 *)buf, bytes / 16, (uint32_t *)sink); break;
    case 3: k_cal_write<uint8_t><<<grid, 256, 0, s>>>((uint8_t *)buf, bytes, (uint8_t)7); break;
    case 4: k_cal_write<uint32_t><<<grid, 256, 0, s>>>((uint32_t *)buf, bytes / 4, 7u); break;
    case 5: k_cal_write16<<<grid, 256, 0, s>>>((uint4 *)buf, bytes / 16, 7u); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
