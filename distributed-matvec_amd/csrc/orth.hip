// orth.hip -- the vector algebra of the eigensolver callers (Diagonalize / PRIMME-style Lanczos and Davidson steps): classical
// Gram-Schmidt of one vector against a block of basis vectors, fused so that the block is streamed as few times as the
// arithmetic allows.  With the slot cache the matvec of a projected basis takes 66 ms on chain_40_symm while two CGS passes in
// torch.mv form (h = V w; w -= V^T h; twice) read the 6.9 GB vectors of the basis four times -- 79 ms per step: the
// orthogonalisation had become the larger half of the solver.
//
// ls_amd_orth_pass(m, n, V, ldv, w, h_in, out):
//     if h_in:  w <- w - sum_k h_in[k] V[k]                    (update)
//     out[k]  = <V[k], w>  for k < m  over the UPDATED w       (the next pass's coefficients = the loss of orthogonality)
//     out[m]  = <w, w>
// One pass over V and w.  Pass 1 (h_in = NULL) gives h and ||w||^2; pass 2 (h_in = h) applies it and returns h2 and the new
// norm in the same sweep -- when ||h2|| is at rounding level (the usual case) the solver is done after TWO reads of V, and
// a third pass (h_in = h2) is only taken when orthogonality was really lost.  Real f64, m <= 32 rows.
#include <hip/hip_runtime.h>

#include <cstdint>

constexpr int kOrthBlock = 256;
constexpr int kOrthMaxRows = 32;

// ALIGNED: V, w 16-byte aligned and ldv even -> the two elements of a thread travel as one 16-byte load per row
template <bool ALIGNED>
__device__ __forceinline__ double2 orth_load2(double const *p) {
    if (ALIGNED) return *reinterpret_cast<double2 const *>(p);
    return make_double2(p[0], p[1]);
}
template <bool ALIGNED>
__global__ __launch_bounds__(kOrthBlock) void k_orth_pass(int m, int64_t n, double const *__restrict__ V, int64_t ldv, double *__restrict__ w,
                                                          double const *__restrict__ h_in, double *__restrict__ out) {
    __shared__ double s_h[kOrthMaxRows];
    __shared__ double s_red[kOrthBlock / 64][kOrthMaxRows + 1];
    if (threadIdx.x < m) s_h[threadIdx.x] = h_in ? h_in[threadIdx.x] : 0.0;
    __syncthreads();
    double acc[kOrthMaxRows + 1];
#pragma unroll
    for (int k = 0; k <= kOrthMaxRows; ++k) acc[k] = 0.0;
    const bool update = h_in != nullptr;
    // two consecutive elements per thread (16-byte loads when the rows are 16-byte aligned: ldv and n even, pointers aligned)
    // every block walks ONE contiguous range of columns (a grid-stride loop makes every block jump gridDim x 4 KB per iteration in
    // each of the m + 1 streams: a new page per row and iteration -- 3.9 TB/s against 5.4 TB/s of torch.mv on the same data)
    const int64_t pairs = n >> 1;
    const int64_t per_block = ((pairs + gridDim.x - 1) / gridDim.x + kOrthBlock - 1) / kOrthBlock * kOrthBlock;
    const int64_t p0 = (int64_t)blockIdx.x * per_block, p1 = p0 + per_block < pairs ? p0 + per_block : pairs;
    for (int64_t p = p0 + threadIdx.x; p < p1; p += kOrthBlock) {
        const int64_t i = 2 * p;
        const double2 wv = orth_load2<ALIGNED>(w + i);
        double w0 = wv.x, w1 = wv.y;
        if (update) {
#pragma unroll 4
            for (int k = 0; k < m; ++k) {
                const double hk = s_h[k];
                const double2 v = orth_load2<ALIGNED>(V + (int64_t)k * ldv + i);
                w0 -= hk * v.x;
                w1 -= hk * v.y;
            }
            if (ALIGNED) *reinterpret_cast<double2 *>(w + i) = make_double2(w0, w1);
            else { w[i] = w0; w[i + 1] = w1; }
        }
#pragma unroll
        for (int k = 0; k < kOrthMaxRows; ++k)
            if (k < m) { // (second read of the block when updating: L1 / L2)
                const double2 v = orth_load2<ALIGNED>(V + (int64_t)k * ldv + i);
                acc[k] += v.x * w0 + v.y * w1;
            }
        acc[kOrthMaxRows] += w0 * w0 + w1 * w1;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t i = n - 1;
        double w0 = w[i];
        if (update) { for (int k = 0; k < m; ++k) w0 -= s_h[k] * V[(int64_t)k * ldv + i]; w[i] = w0; }
#pragma unroll
        for (int k = 0; k < kOrthMaxRows; ++k) // (compile-time indices: a run-time index would send acc[] to scratch)
            if (k < m) acc[k] += V[(int64_t)k * ldv + i] * w0;
        acc[kOrthMaxRows] += w0 * w0;
    }
    // wave reduction, then one atomic per block and row
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k <= kOrthMaxRows; ++k) {
        if (k < m || k == kOrthMaxRows) {
            double v = acc[k];
            for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
            if (lane == 0) s_red[wave][k] = v;
        }
    }
    __syncthreads();
    if (threadIdx.x <= kOrthMaxRows && (threadIdx.x < m || threadIdx.x == kOrthMaxRows)) {
        double v = 0.0;
        for (int q = 0; q < kOrthBlock / 64; ++q) v += s_red[q][threadIdx.x];
        unsafeAtomicAdd(out + (threadIdx.x == kOrthMaxRows ? m : threadIdx.x), v);
    }
}

extern "C" int ls_amd_orth_max_rows(void) { return kOrthMaxRows; }
extern "C" int ls_amd_internal_error(char const *fmt, ...); // host.c: formats into ls_amd_last_error(), returns -1

extern "C" int ls_amd_orth_pass(int m, int64_t n, double const *d_V, int64_t ldv, double *d_w, double const *d_h_in, double *d_out, void *stream) {
    if (m < 0 || m > kOrthMaxRows || n < 0 || ldv < n)
        return ls_amd_internal_error("ls_amd_orth_pass: bad arguments (m = %d of at most %d rows, n = %lld, ldv = %lld)", m, kOrthMaxRows, (long long)n, (long long)ldv);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(d_out, 0, sizeof(double) * (size_t)(m + 1), s);
    if (e != hipSuccess) return ls_amd_internal_error("ls_amd_orth_pass: hipMemsetAsync: %s", hipGetErrorString(e));
    if (n == 0) return 0;
    int64_t blocks = ((n >> 1) + kOrthBlock - 1) / kOrthBlock;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 16) blocks = 256 * 16; // 16 blocks per CU, grid-stride: few atomics, long streams
    const bool aligned = ((uintptr_t)d_V % 16 == 0) && ((uintptr_t)d_w % 16 == 0) && (ldv % 2 == 0);
    if (aligned) hipLaunchKernelGGL(k_orth_pass<true>, dim3((unsigned)blocks), dim3(kOrthBlock), 0, s, m, n, d_V, ldv, d_w, d_h_in, d_out);
    else hipLaunchKernelGGL(k_orth_pass<false>, dim3((unsigned)blocks), dim3(kOrthBlock), 0, s, m, n, d_V, ldv, d_w, d_h_in, d_out);
    e = hipGetLastError();
    return e == hipSuccess ? 0 : ls_amd_internal_error("ls_amd_orth_pass: launch (m = %d, n = %lld, ldv = %lld): %s", m, (long long)n, (long long)ldv, hipGetErrorString(e));
}

// Thick restart: V[:m_out] <- S^T V[:m_in] in place (S: m_in x m_out, row-major, device), column by column -- a thread holds the
// m_in values of its column in registers, so the rows it overwrites (m_out <= m_in) have been read.  torch.mm on this shape (a
// 12 x 8 matrix against 861 M columns) took 83 ms per restart on chain_40_symm; this reads m_in and writes m_out vectors once.
__global__ __launch_bounds__(kOrthBlock) void k_basis_rotate(int m_in, int m_out, int64_t n, double *__restrict__ V, int64_t ldv, double const *__restrict__ S) {
    __shared__ double s_S[kOrthMaxRows * kOrthMaxRows];
    for (int k = threadIdx.x; k < m_in * m_out; k += kOrthBlock) s_S[k] = S[k];
    __syncthreads();
    const int64_t per_block = ((n + gridDim.x - 1) / gridDim.x + kOrthBlock - 1) / kOrthBlock * kOrthBlock;
    const int64_t i0 = (int64_t)blockIdx.x * per_block, i1 = i0 + per_block < n ? i0 + per_block : n;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kOrthBlock) {
        double v[kOrthMaxRows];
#pragma unroll
        for (int k = 0; k < kOrthMaxRows; ++k) v[k] = k < m_in ? V[(int64_t)k * ldv + i] : 0.0;
        for (int o = 0; o < m_out; ++o) {
            double y = 0.0;
#pragma unroll
            for (int k = 0; k < kOrthMaxRows; ++k)
                if (k < m_in) y += s_S[k * m_out + o] * v[k];
            V[(int64_t)o * ldv + i] = y;
        }
    }
}
extern "C" int ls_amd_basis_rotate(int m_in, int m_out, int64_t n, double *d_V, int64_t ldv, double const *d_S, void *stream) {
    // (m_out <= m_in <= kOrthMaxRows also bounds the m_in x m_out coefficients by the kernel's LDS copy, s_S)
    if (m_in < 1 || m_in > kOrthMaxRows || m_out < 1 || m_out > m_in || m_in * m_out > kOrthMaxRows * kOrthMaxRows || n < 0 || ldv < n)
        return ls_amd_internal_error("ls_amd_basis_rotate: bad arguments (m_in = %d, m_out = %d, at most %d rows, n = %lld, ldv = %lld)", m_in, m_out, kOrthMaxRows, (long long)n, (long long)ldv);
    if (n == 0) return 0;
    int64_t blocks = (n + kOrthBlock - 1) / kOrthBlock;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_basis_rotate, dim3((unsigned)blocks), dim3(kOrthBlock), 0, (hipStream_t)stream, m_in, m_out, n, d_V, ldv, d_S);
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : ls_amd_internal_error("ls_amd_basis_rotate: launch (m_in = %d, m_out = %d, n = %lld): %s", m_in, m_out, (long long)n, hipGetErrorString(e));
}

