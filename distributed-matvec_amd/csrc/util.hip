// util.hip -- measurement helpers that are not part of the hot path.
// ls_amd_stream_copy: a plain streaming copy (16 bytes per lane, grid-stride), the "device copy kernel" SURVEY.md 8(d) asks
// the attainable HBM rate of the box to be measured with; bench.py reports it next to the roofline.
#include <hip/hip_runtime.h>

#include <cstdint>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(int64_t n16, u32x4 const *__restrict__ src, u32x4 *__restrict__ dst) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) {
        const u32x4 v = __builtin_nontemporal_load(src + i);
        __builtin_nontemporal_store(v, dst + i);
    }
}

extern "C" int ls_amd_stream_copy(void *d_dst, void const *d_src, int64_t bytes, void *stream) {
    const int64_t n16 = bytes / 16;
    if (n16 <= 0) return 0;
    int64_t blocks = (n16 + 255) / 256;
    if (blocks > ((int64_t)1 << 30)) blocks = (int64_t)1 << 30; // one 16-byte element per thread: plain grids stream best here
    hipLaunchKernelGGL(k_stream_copy, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n16, (u32x4 const *)d_src, (u32x4 *)d_dst);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
