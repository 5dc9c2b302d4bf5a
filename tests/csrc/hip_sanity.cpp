// Minimal stand-alone HIP sanity program (bring-up aid): malloc, memcpy, a trivial kernel, sync.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k_add(int *p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] += 1;
}
int main() {
    int n = 1024, *d = nullptr, h[1024];
    for (int i = 0; i < n; i++) h[i] = i;
    printf("malloc\n"); fflush(stdout);
    if (hipMalloc(&d, n * 4) != hipSuccess) return 1;
    printf("memcpy\n"); fflush(stdout);
    if (hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice) != hipSuccess) return 2;
    printf("launch\n"); fflush(stdout);
    hipLaunchKernelGGL(k_add, dim3(4), dim3(256), 0, 0, d, n);
    printf("sync\n"); fflush(stdout);
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    printf("ok %d %d\n", h[0], h[1023]);
    return 0;
}
