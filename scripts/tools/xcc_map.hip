// Diagnostic: which XCD (XCC_ID hardware register) does block b of a launch run on?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int *out, int spin) {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    if (threadIdx.x == 0) out[blockIdx.x] = (int)(v & 0xf);
    // keep the block resident for a while so that the whole grid is co-resident
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
}
int main() {
    for (int grid : {2048, 1024, 4096}) {
        int *d;
        hipMalloc(&d, grid * sizeof(int));
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 200000);
        hipDeviceSynchronize();
        std::vector<int> h(grid);
        hipMemcpy(h.data(), d, grid * sizeof(int), hipMemcpyDeviceToHost);
        int agree = 0;
        for (int b = 0; b < grid; ++b) agree += h[b] == b % 8;
        printf("grid %d: block b on XCC b%%8 for %d of %d blocks; first 24:", grid, agree, grid);
        for (int b = 0; b < 24; ++b) printf(" %d", h[b]);
        printf("\n");
        hipFree(d);
    }
    return 0;
}
