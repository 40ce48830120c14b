// repl.hip -- helpers of the replicated-x driver (dist.c): the owner-grouping pass of a CHUNK of y and the event plumbing of
// the chunked return / the adaptive split.  Streaming kernels, nothing of the hot row path (k_rows.hip, k_pull.hip).
//
// Reference replaced: /root/reference/src/BlockToHashed.chpl:87-208 restricted to a rank's rows -- "group my results by owner"
// before they travel back (DistributedMatrixVector.chpl:739-853: consumers drain while producers still compute; here the rows
// of chunk c return to their owners while chunk c + 1 is gathered).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

#include "lsk.h"

#define RP_CHECK(expr)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) {                                                                          \
            size_t cap_ = 0;                                                                             \
            char *buf_ = lsk_error_buffer(&cap_);                                                        \
            snprintf(buf_, cap_, "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return -1;                                                                                   \
        }                                                                                                \
    } while (0)

// out[i] = src[perm[i]] for i in [lo_k, hi_k), k = blockIdx.y: one launch for the P owner groups of a chunk of rows
template <typename I, typename T>
__global__ __launch_bounds__(256) void k_gather_perm_ranges(lsk_ranges R, I const *__restrict__ perm, T const *__restrict__ src, T *__restrict__ out) {
    const int64_t lo = R.lo[blockIdx.y], hi = R.hi[blockIdx.y];
    for (int64_t i = lo + (int64_t)blockIdx.x * 256 + threadIdx.x; i < hi; i += (int64_t)gridDim.x * 256)
        out[i] = src[__builtin_nontemporal_load(perm + i)];
}

extern "C" int lsk_gather_perm_ranges(lsk_ranges const *ranges, void const *perm, int perm_is_64, int elt_size, void const *src, void *out, void *stream) {
    lsk_ranges R = *ranges;
    if (R.n < 1) return 0;
    if (R.n > 64) { size_t cap = 0; char *b = lsk_error_buffer(&cap); snprintf(b, cap, "lsk_gather_perm_ranges: more than 64 ranges"); return -1; }
    int64_t longest = 0;
    for (int k = 0; k < R.n; ++k) if (R.hi[k] - R.lo[k] > longest) longest = R.hi[k] - R.lo[k];
    if (longest <= 0) return 0;
    int64_t blocks = (longest + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    const dim3 g((unsigned)blocks, (unsigned)R.n), b(256);
    hipStream_t s = (hipStream_t)stream;
    if (elt_size == 8) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm_ranges<int64_t, double>), g, b, 0, s, R, (int64_t const *)perm, (double const *)src, (double *)out);
        else hipLaunchKernelGGL((k_gather_perm_ranges<int32_t, double>), g, b, 0, s, R, (int32_t const *)perm, (double const *)src, (double *)out);
    } else if (elt_size == 16) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm_ranges<int64_t, double2>), g, b, 0, s, R, (int64_t const *)perm, (double2 const *)src, (double2 *)out);
        else hipLaunchKernelGGL((k_gather_perm_ranges<int32_t, double2>), g, b, 0, s, R, (int32_t const *)perm, (double2 const *)src, (double2 *)out);
    } else { size_t cap = 0; char *bf = lsk_error_buffer(&cap); snprintf(bf, cap, "lsk_gather_perm_ranges: element size %d", elt_size); return -1; }
    RP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int lsk_event_query(void *ev) {
    const hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) return 0;
    if (e == hipErrorNotReady) return 1;
    (void)hipGetLastError();
    return -1;
}
extern "C" int lsk_stream_wait_event(void *stream, void *ev) {
    RP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0));
    return 0;
}
