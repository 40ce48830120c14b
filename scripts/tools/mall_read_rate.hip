// Read-only streaming rate of a buffer of S bytes read R times inside ONE launch (grid-stride, 16 bytes per lane): what the
// memory-side cache (Infinity Cache, 256 MB) delivers next to HBM.  hipcc --offload-arch=gfx950 -O3 mall_read_rate.hip -o /tmp/mall && /tmp/mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(256) void k_read(uint4 const *__restrict__ p, size_t n16, int reps, unsigned *sink) {
    unsigned acc = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int r = 0; r < reps; ++r) {
        // every pass starts at another phase, so that a CU does not re-read its own lines out of its L2
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x + (size_t)r * 977 * 256; i < n16 + (size_t)r * 977 * 256; i += stride) {
            const uint4 v = p[i % n16];
            acc ^= v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    if (acc == 0x12345678u) atomicAdd(sink, 1u);
}
int main() {
    const size_t maxb = (size_t)4 << 30;
    uint4 *buf; unsigned *sink;
    hipMalloc(&buf, maxb); hipMalloc(&sink, 4);
    hipMemset(buf, 1, maxb); hipMemset(sink, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t sizes[] = {(size_t)8 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)192 << 20,
                            (size_t)256 << 20, (size_t)384 << 20, (size_t)512 << 20, (size_t)1 << 30, (size_t)4 << 30};
    for (size_t s : sizes) {
        const size_t n16 = s / 16;
        const int reps = (int)(((size_t)16 << 30) / s);  // 16 GB of reads per launch
        const int grid = 256 * 8;
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n16, 1, sink);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, buf, n16, reps, sink);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        printf("buffer %6zu MB read %4d x in one launch: %8.1f GB/s (%.3f ms)\n", s >> 20, reps, (double)s * reps / (ms * 1e-3) / 1e9, ms);
    }
    return 0;
}
