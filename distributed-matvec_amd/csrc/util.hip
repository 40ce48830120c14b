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

// ls_amd_stream_read: the read-only counterpart (each thread reads `per_thread` 16-byte elements a block-stride apart and keeps
// an xor of them; one 4-byte store per thread that saw a non-zero): the attainable READ rate of the box.  The pull kernels are
// > 90 % reads, so this, not the copy, is the line their traffic should be held against.
__global__ __launch_bounds__(256) void k_stream_read(int64_t n16, int per_thread, u32x4 const *__restrict__ src, unsigned *__restrict__ sink) {
    const int64_t base = (int64_t)blockIdx.x * 256 * per_thread + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
    for (int q = 0; q < per_thread; ++q) {
        const int64_t i = base + (int64_t)q * 256;
        if (i < n16) acc ^= __builtin_nontemporal_load(src + i);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9e3779b9u) sink[0] = 1; // practically never: keeps the loads alive
}
extern "C" int ls_amd_stream_read(void const *d_src, int64_t bytes, int per_thread, void *d_sink, void *stream) {
    const int64_t n16 = bytes / 16;
    if (n16 <= 0) return 0;
    if (per_thread < 1) per_thread = 1;
    const int64_t per_block = (int64_t)256 * per_thread, blocks = (n16 + per_block - 1) / per_block;
    hipLaunchKernelGGL(k_stream_read, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n16, per_thread, (u32x4 const *)d_src, (unsigned *)d_sink);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
