// k_plan.hip -- plan-time and API helpers: norms, index tables, rank directories, scans, enumeration (StatesEnumeration.chpl:158-224),
// layout converters (BlockToHashed / HashedToBlock), batched externs, host test hooks of the device helpers.  Split out of kernels.hip
// in round 6; shared device helpers: lsk_dev.hpp.
#include "lsk_dev.hpp"

// ---- host test hooks / profiling entries of the helpers in lsk_dev.hpp ------------------------------------------------------
extern "C" int lsk_ablate_mask(void) {
    if (!kAblate) return 0;
    char const *e = getenv("LS_AMD_ABLATE");
    return e ? atoi(e) : 0;
}

// host test hook: mode-3 orbit minimum of `a` on a ring of L sites (32-bit words for L <= 32, as the kernels choose)
extern "C" uint64_t lsk_test_rep_trivial_dihedral(uint64_t a, int L, int inv, int reflect) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    if (L <= 32) return (uint64_t)rep_trivial_dihedral<uint32_t>((uint32_t)a, L, (uint32_t)mask, inv != 0, reflect != 0);
    return rep_trivial_dihedral<uint64_t>(a, L, mask, inv != 0, reflect != 0);
}

// Profiling entry (scripts/k4_rate.py): K4 alone over the packets of a ring -- every row's state with each adjacent pair
// (b, b + 1 mod L) flipped, all lanes busy -- to price it outside the tile kernels.  variant 0: the whole orbit minimum;
// 1: the two run searches only; 2: the packets only (loop and flip, no K4).
template <typename W>
__global__ __launch_bounds__(kBlock) void k_bench_k4(int L, int inv, int reflect, int variant, int64_t n,
                                                     uint64_t const *__restrict__ reps, uint64_t *__restrict__ out) {
    const W mask = (W)(L >= 64 ? ~0ULL : ((1ULL << L) - 1));
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const W a = (W)reps[i];
        W acc = 0;
        for (int b = 0; b < L; ++b) {
            const W pm = (W)(((W)1 << b) | ((W)1 << (b + 1 == L ? 0 : b + 1)));
            const W beta = a ^ pm;
            if (variant == 0) acc ^= rep_trivial_dihedral<W>(beta, L, mask, inv != 0, reflect != 0);
            else if (variant == 1) {
                int l0, l1;
                const W r0 = longest_runs<W>((W)(~beta & mask), L, mask, l0);
                const W r1 = longest_runs<W>(beta, L, mask, l1);
                acc ^= r0 + r1 + (W)(l0 + 64 * l1);
            } else acc ^= beta + (W)b;
        }
        out[i] = (uint64_t)acc;
    }
}

extern "C" int lsk_bench_k4(int L, int inv, int reflect, int variant, int64_t n, uint64_t const *reps, uint64_t *out, void *stream) {
    if (n <= 0 || L < 2 || L > 64) return 0;
    const dim3 g((unsigned)grid_for(n)), b(kBlock);
    if (L <= 32) hipLaunchKernelGGL(k_bench_k4<uint32_t>, g, b, 0, (hipStream_t)stream, L, inv, reflect, variant, n, reps, out);
    else hipLaunchKernelGGL(k_bench_k4<uint64_t>, g, b, 0, (hipStream_t)stream, L, inv, reflect, variant, n, reps, out);
    LSK_LAUNCH_CHECK();
    return 0;
}

// rowtab2[r] = rowtab[r] | rowtab[rev_tw(r)] << 32 (tw <= 8): out[2^tw]
extern "C" int lsk_torus_rowtab(int tw, uint32_t *out);

extern "C" int lsk_torus_rowtab2(int tw, uint64_t *out) {
    if (tw < 1 || tw > 8) return -1;
    uint32_t t[256];
    lsk_torus_rowtab(tw, t);
    for (uint32_t r = 0; r < (1u << tw); ++r) {
        uint32_t q = 0;
        for (int i = 0; i < tw; ++i) if (r & (1u << i)) q |= 1u << (tw - 1 - i);
        out[r] = (uint64_t)t[r] | ((uint64_t)t[q] << 32);
    }
    return 0;
}

// host test hook: the factorised minimum over the images `present` of one base word (64-bit arithmetic)
extern "C" uint64_t lsk_test_torus_min_d2(uint64_t v, int L, int tw, int inv, int present, uint64_t const *rowtab2, uint64_t best) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    uint64_t col0 = 0;
    for (int y = 0; y < L / tw; ++y) col0 |= 1ULL << (y * tw);
    return torus_min_d2<uint64_t>(v, L, tw, mask, col0, inv != 0, present, rowtab2, best);
}

// the row table of torus_min for rows of tw <= 8 bits: out[2^tw]
extern "C" int lsk_torus_rowtab(int tw, uint32_t *out) {
    if (tw < 1 || tw > 8) return -1;
    const uint32_t rmask = (1u << tw) - 1u;
    for (uint32_t r = 0; r <= rmask; ++r) {
        uint32_t mn = r, mx = r, amn = 0, amx = 0;
        for (int i = 1; i < tw; ++i) {
            const uint32_t q = ((r << i) | (r >> (tw - i))) & rmask;
            mn = q < mn ? q : mn;
            mx = q > mx ? q : mx;
        }
        for (int i = 0; i < tw; ++i) {
            const uint32_t q = i == 0 ? r : (((r << i) | (r >> (tw - i))) & rmask);
            if (q == mn) amn |= 1u << i;
            if (q == mx) amx |= 1u << i;
        }
        out[r] = mn | (amn << 8) | (mx << 16) | (amx << 24);
    }
    return 0;
}

// host test hook: torus_min of one word (64-bit arithmetic)
extern "C" uint64_t lsk_test_torus_min(uint64_t v, int L, int tw, int inv, uint32_t const *rowtab, uint64_t best) {
    const uint64_t mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    uint64_t col0 = 0;
    for (int y = 0; y < L / tw; ++y) col0 |= 1ULL << (y * tw);
    return torus_min<uint64_t>(v, L, tw, mask, col0, inv != 0, rowtab, best);
}

// ---------------------------------------------------------------------------------------------
// out[i] = src[perm[i]]: the hashed -> block permutation of the replicated-x exchange (P ascending streams interleaved).
// Two outputs per thread so that f64 results leave as 16-byte stores; perm is read with 8- / 16-byte loads.
// ---------------------------------------------------------------------------------------------
template <typename I, typename T>
__global__ __launch_bounds__(kBlock) void k_gather_perm(int64_t n, I const *__restrict__ perm, T const *__restrict__ src,
                                                        T *__restrict__ out) {
    const int64_t pairs = n >> 1;
    for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < pairs; k += (int64_t)gridDim.x * kBlock) {
        const I p0 = __builtin_nontemporal_load(perm + 2 * k), p1 = __builtin_nontemporal_load(perm + 2 * k + 1);
        const T a = src[p0], b = src[p1];
        out[2 * k] = a;
        out[2 * k + 1] = b;
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) out[n - 1] = src[perm[n - 1]];
}
__global__ __launch_bounds__(kBlock) void k_iota(int64_t n, int64_t base, int64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = base + i;
}
extern "C" int lsk_iota_i64(int64_t n, int64_t base, int64_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_iota, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, base, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_narrow_i32(int64_t n, int64_t const *__restrict__ in, int32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (int32_t)in[i];
}
extern "C" int lsk_narrow_i32(int64_t n, int64_t const *in, int32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_narrow_i32, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, in, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_axpy1(int64_t n, double const *__restrict__ a, double *__restrict__ y) {
    const int64_t m = CPLX ? 2 * n : n;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < m; i += (int64_t)gridDim.x * kBlock) y[i] += a[i];
}
// y += a (the accumulate semantics of operators without diagonal terms, DMV:1062-1063, in the replicated-x driver)
extern "C" int lsk_add_into(int cplx, int64_t n, void const *a, void *y, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_axpy1<true>, dim3(grid_for(2 * n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)a, (double *)y);
    else hipLaunchKernelGGL(k_axpy1<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)a, (double *)y);
    LSK_LAUNCH_CHECK();
    return 0;
}
extern "C" int lsk_gather_perm(int64_t n, void const *perm, int perm_is_64, int elt_size, void const *src, void *out, void *stream) {
    if (n == 0) return 0;
    const int64_t blocks = (n / 2 + kBlock - 1) / kBlock;
    dim3 g((unsigned)(blocks < 1 ? 1 : (blocks > (1 << 20) ? (1 << 20) : blocks))), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (elt_size == 8) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm<int64_t, double>), g, b, 0, s, n, (int64_t const *)perm, (double const *)src, (double *)out);
        else hipLaunchKernelGGL((k_gather_perm<int32_t, double>), g, b, 0, s, n, (int32_t const *)perm, (double const *)src, (double *)out);
    } else if (elt_size == 16) {
        if (perm_is_64) hipLaunchKernelGGL((k_gather_perm<int64_t, double2>), g, b, 0, s, n, (int64_t const *)perm, (double2 const *)src, (double2 *)out);
        else hipLaunchKernelGGL((k_gather_perm<int32_t, double2>), g, b, 0, s, n, (int32_t const *)perm, (double2 const *)src, (double2 *)out);
    } else { snprintf(g_err, sizeof(g_err), "lsk_gather_perm: element size %d", elt_size); return -1; }
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// plan-time helpers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_norms(lsk_basis bs, lsk_group_elem const *__restrict__ elems, int64_t n,
                                                  uint64_t const *__restrict__ reps, double *__restrict__ norms) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t rep; double chr, chi, stab;
        state_info(bs, elems, reps[i], rep, chr, chi, stab);
        double n2 = stab * bs.inv_order;
        norms[i] = n2 > 1e-12 ? sqrt(n2) : 0.0;
    }
}
extern "C" int lsk_norms(lsk_basis bs, int64_t n, uint64_t const *reps, double *norms, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_norms, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, bs, bs.elems, n, reps, norms);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_check_combinadic(uint64_t const *__restrict__ g_binom, int64_t n,
                                                             uint64_t const *__restrict__ reps, int *flag) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (rank_combinadic(reps[i], s_binom) != i) atomicExch(flag, 1);
}
extern "C" int lsk_check_combinadic(lsk_index ix, int hamming_weight, int64_t n, uint64_t const *reps, int *d_flag,
                                    void *stream) {
    (void)hamming_weight;
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_check_combinadic, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ix.binom, n, reps, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---- rank directory (lsk_rankdir): build + self-check -----------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_rankdir_mark(int64_t n, uint64_t const *__restrict__ reps, int sites, int weight,
                                                         uint64_t const *__restrict__ g_binom, lsk_rankdir *__restrict__ dir, int *flag) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t s = reps[i];
        if (__popcll(s) != weight || (sites < 64 && (s >> sites) != 0)) { atomicExch(flag, 1); continue; }
        const uint64_t g = (uint64_t)rank_combinadic(s, s_binom);
        atomicOr((unsigned long long *)&dir[g >> 6].bits, 1ULL << (g & 63));
    }
}
struct ScanDirPopcIn {
    lsk_rankdir const *dir;
    __device__ int64_t operator()(int64_t i) const { return (int64_t)__popcll(dir[i].bits); }
};
__global__ __launch_bounds__(kBlock) void k_rankdir_prefix(int64_t entries, int64_t const *__restrict__ pre, lsk_rankdir *__restrict__ dir) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock) {
        dir[i].prefix = (uint32_t)pre[i];
        dir[i].pad = 0;
    }
}
__global__ __launch_bounds__(kBlock) void k_rankdir_check(lsk_index ix, int64_t n, uint64_t const *__restrict__ reps, int *flag) {
    extern __shared__ uint64_t s_db[];
    rankdir_load(ix, s_db);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (rankdir_index(ix, reps[i], s_db) != i) atomicExch(flag, 1);
}

// table[b] = first i with (reps[i] >> shift) >= b.  Element i owns the buckets (bucket(i - 1), bucket(i)]; the representatives of
// a projected basis are far from uniform over the top bits (92 % of the buckets of chain_32_symm are empty, in runs of millions),
// so a run longer than 64 buckets is filled by the WHOLE WAVE of its owner instead of one lane (chain_40_symm: 331 -> 23 ms per table).
__global__ __launch_bounds__(kBlock) void k_build_table(int64_t n, uint64_t const *__restrict__ reps, int shift,
                                                        int64_t nbuckets, uint32_t *__restrict__ table) {
    const int lane = threadIdx.x & 63;
    const int64_t total = n + 1, rounded = (total + 63) & ~(int64_t)63;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < rounded; i += (int64_t)gridDim.x * kBlock) {
        int64_t lo = 0, hi = -1; // (an empty range for the lanes past the end: they still take part in the wave's long runs)
        if (i < total) {
            lo = (i == 0) ? 0 : (int64_t)(reps[i - 1] >> shift) + 1;
            hi = (i == n) ? nbuckets : (int64_t)(reps[i] >> shift);
        }
        const bool is_long = hi - lo >= 64;
        if (!is_long) for (int64_t b = lo; b <= hi; ++b) table[b] = (uint32_t)i;
        unsigned long long m = __ballot(is_long);
        while (m) { // wave-uniform: every lane helps to fill the long runs of the wave, one after the other
            const int l = __builtin_ctzll(m);
            m &= m - 1;
            const int64_t rlo = (int64_t)readlane_t<uint64_t>((uint64_t)lo, l), rhi = (int64_t)readlane_t<uint64_t>((uint64_t)hi, l);
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)i, l);
            for (int64_t b = rlo + lane; b <= rhi; b += 64) table[b] = v;
        }
    }
}
extern "C" int lsk_build_table(int64_t n, uint64_t const *reps, int shift, int64_t nbuckets, uint32_t *table,
                               void *stream) {
    hipLaunchKernelGGL(k_build_table, dim3(grid_for(n + 1)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, shift, nbuckets, table);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// batched externs on device pointers
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_state_info(lsk_basis bs, lsk_group_elem const *__restrict__ elems,
                                                       int64_t n, uint64_t const *__restrict__ alphas,
                                                       uint64_t *__restrict__ betas, double *__restrict__ chars,
                                                       double *__restrict__ norms) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t rep; double chr, chi, stab;
        uint64_t a = alphas[i];
        if (bs.proj == LSK_PROJ_NONE) { rep = a; chr = 1.0; chi = 0.0; stab = 1.0 / bs.inv_order; }
        else state_info(bs, elems, a, rep, chr, chi, stab);
        double n2 = stab * bs.inv_order;
        betas[i] = rep;
        chars[2 * i] = chr;
        chars[2 * i + 1] = chi;
        norms[i] = n2 > 1e-12 ? sqrt(n2) : 0.0;
    }
}
extern "C" int lsk_state_info(lsk_basis bs, int64_t n, uint64_t const *alphas, uint64_t *betas, double *characters,
                              double *norms, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_state_info, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, bs, bs.elems, n, alphas, betas, characters, norms);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_state_index(lsk_index ix, int64_t n, uint64_t const *__restrict__ spins,
                                                        int64_t *__restrict__ indices) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    if (ix.kind == LSK_INDEX_COMBINADIC) load_binom(s_binom, ix.binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t s = spins[i];
        int64_t idx;
        if (ix.kind == LSK_INDEX_IDENTITY) idx = (int64_t)s < ix.count ? (int64_t)s : -1;
        else if (ix.kind == LSK_INDEX_COMBINADIC) {
            idx = rank_combinadic(s, s_binom);
            // membership: the basis is the first `count` states of one popcount class
            if (idx >= ix.count || __popcll(s) != __popcll(ix.reps[0])) idx = -1;
        } else idx = search_index(ix, s);
        indices[i] = idx;
    }
}
extern "C" int lsk_state_index(lsk_index ix, int64_t n, uint64_t const *spins, int64_t *indices, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_state_index, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, ix, n, spins, indices);
    LSK_LAUNCH_CHECK();
    return 0;
}
// Plan time, replicated-x exchange of unprojected bases: which blocks of 2^shift rows of the GLOBAL vector do the rows
// alphas[0, n) read?  Every non-zero off-diagonal group of a row -> partner state -> index in the global basis -> one bit.
// (The rows' own neighbourhood -- the LDS windows of the staged kernels -- is added by the host.)
__global__ __launch_bounds__(kBlock) void k_reach_blocks(int n_groups, lsk_group const *__restrict__ groups, lsk_term const *__restrict__ off,
                                                         lsk_index ix, int64_t n, uint64_t const *__restrict__ alphas, int shift,
                                                         uint32_t *__restrict__ bitmap) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    if (ix.kind == LSK_INDEX_COMBINADIC) load_binom(s_binom, ix.binom);
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t a = alphas[i];
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            if (cr == 0.0 && ci == 0.0) continue;
            const uint64_t s = a ^ G.x;
            int64_t idx;
            if (ix.kind == LSK_INDEX_IDENTITY) idx = (int64_t)s < ix.count ? (int64_t)s : -1;
            else if (ix.kind == LSK_INDEX_COMBINADIC) {
                idx = rank_combinadic(s, s_binom);
                if (idx >= ix.count || __popcll(s) != __popcll(ix.reps[0])) idx = -1;
            } else idx = search_index(ix, s);
            if (idx < 0) continue; // outside the basis: the matvec reports it (DMV:115-118)
            const int64_t b = idx >> shift;
            const uint32_t bit = 1u << (b & 31);
            if (!(bitmap[b >> 5] & bit)) atomicOr(bitmap + (b >> 5), bit); // plain read first: nearly every block is marked early
        }
    }
}
extern "C" int lsk_reach_blocks(lsk_operator op, lsk_index ix_global, int64_t n, uint64_t const *alphas, int shift, uint32_t *bitmap,
                                void *stream) {
    if (n == 0 || op.n_groups == 0) return 0;
    hipLaunchKernelGGL(k_reach_blocks, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, ix_global, n,
                       alphas, shift, bitmap);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_offdiag_counts(int n_groups, lsk_group const *__restrict__ groups,
                                                           lsk_term const *__restrict__ off, int64_t n,
                                                           uint64_t const *__restrict__ alphas,
                                                           int64_t *__restrict__ counts) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = alphas[i];
        int c = 0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            c += (cr != 0.0 || ci != 0.0);
        }
        counts[i] = c;
    }
}
extern "C" int lsk_offdiag_counts(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t *counts, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_offdiag_counts, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, n, alphas, counts);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_offdiag_fill(int n_groups, lsk_group const *__restrict__ groups,
                                                         lsk_term const *__restrict__ off, int64_t n,
                                                         uint64_t const *__restrict__ alphas,
                                                         int64_t const *__restrict__ offsets,
                                                         uint64_t *__restrict__ betas, double *__restrict__ coeffs,
                                                         double const *__restrict__ xs) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = alphas[i];
        int64_t o = offsets[i];
        double xv = xs ? xs[i] : 1.0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            if (cr != 0.0 || ci != 0.0) {
                betas[o] = a ^ G.x;
                coeffs[2 * o] = cr * xv;
                coeffs[2 * o + 1] = ci * xv;
                ++o;
            }
        }
    }
}
extern "C" int lsk_offdiag_fill(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t const *offsets,
                                uint64_t *betas, double *coeffs, double const *xs, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_offdiag_fill, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_groups, op.groups, op.off, n, alphas, offsets, betas, coeffs, xs);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_diag_coeffs(int n_diag, lsk_term const *__restrict__ diag, int64_t n,
                                                        uint64_t const *__restrict__ alphas, double *__restrict__ ys,
                                                        double const *__restrict__ xs) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        double dr, di;
        term_sum<true>(diag, 0, n_diag, alphas[i], dr, di);
        ys[i] = xs ? dr * xs[i] : dr;
    }
}
extern "C" int lsk_diag_coeffs(lsk_operator op, int64_t n, uint64_t const *alphas, double *ys, double const *xs,
                               void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_diag_coeffs, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, op.n_diag, op.diag, n, alphas, ys, xs);
    LSK_LAUNCH_CHECK();
    return 0;
}

// Exclusive prefix sum of int64 values (the layout converters, the enumeration, ls_chpl_operator_apply_off_diag): three small
// kernels per level -- per-block scan of kScanPer elements + block totals, the totals scanned recursively, the offsets added
// back.  (Hand-written: hipCUB's DeviceScan brought 225 trampoline kernels into the library for these three call sites.)
constexpr int kScanItems = 8;
constexpr int kScanPer = kBlock * kScanItems;
struct ScanArrayIn {
    int64_t const *v;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return v[i]; }
};
struct ScanMaskIn { // 1 where masks[i] == p
    uint8_t const *masks;
    uint8_t p;
    __device__ __forceinline__ int64_t operator()(int64_t i) const { return masks[i] == p ? 1 : 0; }
};
template <typename In>
__global__ __launch_bounds__(kBlock) void k_scan_block(In in, int64_t n, int64_t *__restrict__ out, int64_t *__restrict__ totals) {
    __shared__ int64_t s_wave[kBlock / 64];
    const int64_t base = (int64_t)blockIdx.x * kScanPer + (int64_t)threadIdx.x * kScanItems;
    int64_t v[kScanItems], sum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { v[k] = base + k < n ? in(base + k) : 0; sum += v[k]; }
    // inclusive scan of the per-thread sums inside the wave, then across the four waves
    int64_t inc = sum;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int64_t o = __shfl_up(inc, d);
        if (lane >= d) inc += o;
    }
    if (lane == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    int64_t wave_off = 0, total = 0;
    for (int w = 0; w < kBlock / 64; ++w) { if (w < (int)(threadIdx.x >> 6)) wave_off += s_wave[w]; total += s_wave[w]; }
    int64_t run = wave_off + inc - sum;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) { if (base + k < n) out[base + k] = run; run += v[k]; }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}
__global__ __launch_bounds__(kBlock) void k_scan_add(int64_t n, int64_t *__restrict__ out, int64_t const *__restrict__ offsets) {
    const int64_t off = offsets[blockIdx.x];
    const int64_t base = (int64_t)blockIdx.x * kScanPer;
    for (int k = threadIdx.x; k < kScanPer; k += kBlock)
        if (base + k < n) out[base + k] += off;
}
// scratch for n elements: totals of every level, one after the other
static int64_t scan_scratch_elems(int64_t n) {
    int64_t e = 0;
    while (n > 1) { n = (n + kScanPer - 1) / kScanPer; e += n; if (n == 1) break; }
    return e > 0 ? e : 1;
}
template <typename In>
static int scan_level(In in, int64_t n, int64_t *out, int64_t *scratch, hipStream_t s) {
    const int64_t blocks = (n + kScanPer - 1) / kScanPer;
    hipLaunchKernelGGL((k_scan_block<In>), dim3((unsigned)blocks), dim3(kBlock), 0, s, in, n, out, scratch);
    LSK_LAUNCH_CHECK();
    if (blocks > 1) {
        ScanArrayIn t{scratch};
        if (scan_level<ScanArrayIn>(t, blocks, scratch, scratch + blocks, s) != 0) return -1; // in place: totals -> their exclusive sums
        hipLaunchKernelGGL(k_scan_add, dim3((unsigned)blocks), dim3(kBlock), 0, s, n, out, scratch);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
template <typename In>
static int exclusive_scan(In in, int64_t n, int64_t *out, int64_t *scratch, hipStream_t s) {
    if (n <= 0) return 0;
    if ((n + kScanPer - 1) / kScanPer > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "scan too large"); return -1; }
    return scan_level<In>(in, n, out, scratch, s);
}
int lsk_internal_exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, hipStream_t s) {
    if (n == 0) return 0;
    int64_t *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(n)));
    ScanArrayIn src{in};
    const int rc = exclusive_scan<ScanArrayIn>(src, n, out, scratch, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(scratch);
    if (rc != 0) return -1;
    LSK_CHECK(e2);
    return 0;
}
extern "C" int lsk_exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, void *stream) {
    return lsk_internal_exclusive_scan_i64(n, in, out, (hipStream_t)stream);
}
extern "C" int lsk_rankdir_build(int64_t n, uint64_t const *reps, int sites, int weight, uint64_t const *d_binom, int64_t entries,
                                 lsk_rankdir *dir, int *d_flag, void *stream) {
    if (entries <= 0 || sites < 1 || sites > 64 || weight < 0 || weight >= LSK_BINOM_K) { snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: bad arguments"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    LSK_CHECK(hipMemsetAsync(dir, 0, sizeof(lsk_rankdir) * (size_t)entries, s));
    if (n > 0) {
        hipLaunchKernelGGL(k_rankdir_mark, dim3(grid_for(n)), dim3(kBlock), 0, s, n, reps, sites, weight, d_binom, dir, d_flag);
        LSK_LAUNCH_CHECK();
    }
    int64_t *pre = nullptr, *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&pre, 8 * (size_t)entries));
    if (hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(entries)) != hipSuccess) { (void)hipFree(pre); snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: out of memory"); return -1; }
    ScanDirPopcIn in{dir};
    int rc = exclusive_scan<ScanDirPopcIn>(in, entries, pre, scratch, s);
    if (rc == 0) {
        hipLaunchKernelGGL(k_rankdir_prefix, dim3(grid_for(entries)), dim3(kBlock), 0, s, entries, pre, dir);
        if (hipGetLastError() != hipSuccess) rc = -1;
    }
    if (rc == 0 && n > 0) {
        lsk_index ix;
        memset(&ix, 0, sizeof(ix));
        ix.kind = LSK_INDEX_SEARCH; ix.count = n; ix.reps = reps; ix.binom = d_binom; ix.dir = dir; ix.dir_sites = sites; ix.dir_weight = weight;
        hipLaunchKernelGGL(k_rankdir_check, dim3(grid_for(n)), dim3(kBlock), sizeof(uint64_t) * (size_t)sites * (size_t)(weight + 1), s, ix, n, reps, d_flag);
        if (hipGetLastError() != hipSuccess) rc = -1;
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(pre);
    (void)hipFree(scratch);
    if (rc != 0) { snprintf(g_err, sizeof(g_err), "lsk_rankdir_build: launch failed"); return -1; }
    LSK_CHECK(e2);
    return 0;
}

// ---- all-destinations directory (lsk_gdir): every rank derives it alone -- the owner of a state is a hash of the state ----------
// thread = one word of 64 consecutive global ranks: unrank the first, Gosper-step through the rest, mark each in its owner's entry
__global__ __launch_bounds__(kBlock) void k_gdir_mark(lsk_gdir gd, Owner ow, uint64_t const *__restrict__ g_binom, int64_t words,
                                                      lsk_rankdir *__restrict__ entries) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < words; w += (int64_t)gridDim.x * kBlock) {
        const int64_t g0 = w << 6, g1 = g0 + 64 < gd.n_ranks ? g0 + 64 : gd.n_ranks;
        uint64_t s = unrank_combinadic(g0, gd.weight, s_binom);
        lsk_rankdir *row = entries + w * (int64_t)gd.P;
        for (int64_t g = g0; g < g1; ++g) {
            row[owner_of(s, ow)].bits |= 1ULL << (g - g0); // (this thread owns the whole row)
            s = next_fixed_hamming(s);
        }
    }
}
struct ScanGdirPopcIn {
    lsk_rankdir const *entries;
    int64_t P, d;
    __device__ int64_t operator()(int64_t w) const { return (int64_t)__popcll(entries[w * P + d].bits); }
};
__global__ __launch_bounds__(kBlock) void k_gdir_prefix(int64_t words, int64_t P, int64_t d, int64_t const *__restrict__ pre,
                                                        lsk_rankdir *__restrict__ entries) {
    for (int64_t w = (int64_t)blockIdx.x * kBlock + threadIdx.x; w < words; w += (int64_t)gridDim.x * kBlock) {
        entries[w * P + d].prefix = (uint32_t)pre[w];
        entries[w * P + d].pad = 0;
    }
}
extern "C" int lsk_gdir_build(lsk_gdir gd, lsk_rankdir *entries, uint64_t const *d_binom, void *stream) {
    if (gd.P < 1 || gd.P > LSK_MAX_PARTS || gd.sites < 1 || gd.sites > 64 || gd.weight < 0 || gd.weight >= LSK_BINOM_K - 1 || gd.n_ranks < 1) {
        snprintf(g_err, sizeof(g_err), "lsk_gdir_build: bad arguments"); return -1;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t words = (gd.n_ranks + 63) >> 6;
    LSK_CHECK(hipMemsetAsync(entries, 0, sizeof(lsk_rankdir) * (size_t)words * (size_t)gd.P, s));
    hipLaunchKernelGGL(k_gdir_mark, dim3(grid_for(words)), dim3(kBlock), 0, s, gd, make_owner(gd.P), d_binom, words, entries);
    LSK_LAUNCH_CHECK();
    int64_t *pre = nullptr, *scratch = nullptr;
    LSK_CHECK(hipMalloc((void **)&pre, 8 * (size_t)words));
    if (hipMalloc((void **)&scratch, 8 * (size_t)scan_scratch_elems(words)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(pre); snprintf(g_err, sizeof(g_err), "lsk_gdir_build: out of memory"); return -1; }
    int rc = 0;
    for (int d = 0; d < gd.P && rc == 0; ++d) { // one exclusive scan of the popcounts per destination
        ScanGdirPopcIn in{entries, gd.P, d};
        rc = exclusive_scan<ScanGdirPopcIn>(in, words, pre, scratch, s);
        if (rc == 0) {
            hipLaunchKernelGGL(k_gdir_prefix, dim3(grid_for(words)), dim3(kBlock), 0, s, words, (int64_t)gd.P, (int64_t)d, pre, entries);
            if (hipGetLastError() != hipSuccess) rc = -1;
        }
    }
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(pre);
    (void)hipFree(scratch);
    if (rc != 0) { snprintf(g_err, sizeof(g_err), "lsk_gdir_build: launch failed"); return -1; }
    LSK_CHECK(e2);
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_gdir_check(lsk_gdir gd, Owner ow, int part, int64_t n, uint64_t const *__restrict__ reps,
                                                       uint64_t const *__restrict__ g_binom, int *flag) {
    extern __shared__ uint64_t s_db[];
    gdir_load(gd, g_binom, s_db);
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        if (owner_of(reps[i], ow) != part || gdir_index(gd, reps[i], part, s_db) != i) atomicExch(flag, 1);
}
extern "C" int lsk_gdir_check(lsk_gdir gd, int part, int64_t n, uint64_t const *reps, uint64_t const *d_binom, int *d_flag, void *stream) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_gdir_check, dim3(grid_for(n)), dim3(kBlock), sizeof(uint64_t) * (size_t)gd.sites * (size_t)(gd.weight + 1), (hipStream_t)stream,
                       gd, make_owner(gd.P), part, n, reps, d_binom, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// enumeration of representatives (enumerateStates, StatesEnumeration.chpl:158-224,516-585)
// Candidate c in [0, n_candidates) is the c-th state in ascending order of the candidate space:
//   fixed Hamming weight -> unrank_combinadic(c) ; otherwise the integer c.
// Each thread owns kEnumChunk consecutive candidates; survivors are flagged in a 64-bit mask, the
// masks' popcounts are scanned, and a second (cheap) kernel writes the survivors in order.
// ---------------------------------------------------------------------------------------------
constexpr int kEnumChunk = 64;

__global__ __launch_bounds__(kBlock) void k_enum_flags(lsk_basis bs, lsk_group_elem const *__restrict__ elems,
                                                       uint64_t const *__restrict__ g_binom, int64_t n_cand,
                                                       int64_t n_threads, uint64_t *__restrict__ flags,
                                                       int64_t *__restrict__ counts) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_threads; t += (int64_t)gridDim.x * kBlock) {
        int64_t c0 = t * kEnumChunk;
        int64_t c1 = c0 + kEnumChunk < n_cand ? c0 + kEnumChunk : n_cand;
        uint64_t s = bs.hamming_weight >= 0 ? unrank_combinadic(c0, bs.hamming_weight, s_binom) : (uint64_t)c0;
        uint64_t m = 0;
        for (int64_t c = c0; c < c1; ++c) {
            bool keep = true;
            if (bs.proj == LSK_PROJ_FULL) {
                // trivial sector: every orbit has non-zero norm, so "is its own orbit minimum" is the whole test
                if (bs.k4_mode != 0) keep = bs.number_sites <= 32 ? rep_trivial<uint32_t>(bs, elems, (uint32_t)s) == (uint32_t)s
                                                                   : rep_trivial<uint64_t>(bs, elems, s) == s;
                else keep = is_representative(bs, elems, s);
            }
            if (keep) m |= 1ULL << (c - c0);
            if (c + 1 < c1) s = (bs.hamming_weight > 0) ? next_fixed_hamming(s) : s + 1;
        }
        flags[t] = m;
        counts[t] = __popcll(m);
    }
}
__global__ __launch_bounds__(kBlock) void k_enum_write(lsk_basis bs, uint64_t const *__restrict__ g_binom,
                                                       int64_t n_cand, int64_t n_threads,
                                                       uint64_t const *__restrict__ flags,
                                                       int64_t const *__restrict__ offsets,
                                                       uint64_t *__restrict__ out) {
    __shared__ uint64_t s_binom[64 * LSK_BINOM_K];
    load_binom(s_binom, g_binom);
    for (int64_t t = (int64_t)blockIdx.x * kBlock + threadIdx.x; t < n_threads; t += (int64_t)gridDim.x * kBlock) {
        uint64_t m = flags[t];
        if (!m) continue;
        int64_t c0 = t * kEnumChunk;
        int64_t c1 = c0 + kEnumChunk < n_cand ? c0 + kEnumChunk : n_cand;
        uint64_t s = bs.hamming_weight >= 0 ? unrank_combinadic(c0, bs.hamming_weight, s_binom) : (uint64_t)c0;
        int64_t o = offsets[t];
        for (int64_t c = c0; c < c1; ++c) {
            if ((m >> (c - c0)) & 1) out[o++] = s;
            if (c + 1 < c1) s = (bs.hamming_weight > 0) ? next_fixed_hamming(s) : s + 1;
        }
    }
}

extern "C" int lsk_enumerate(lsk_basis bs, uint64_t const *d_binom, int64_t n_cand, uint64_t **d_states,
                             int64_t *count, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    *d_states = nullptr;
    *count = 0;
    if (n_cand <= 0) { LSK_CHECK(hipMalloc((void **)d_states, 8)); return 0; }
    int64_t n_threads = (n_cand + kEnumChunk - 1) / kEnumChunk;
    uint64_t *flags = nullptr;
    int64_t *counts = nullptr, *offsets = nullptr;
    LSK_CHECK(hipMalloc((void **)&flags, 8 * n_threads));
    LSK_CHECK(hipMalloc((void **)&counts, 8 * n_threads));
    LSK_CHECK(hipMalloc((void **)&offsets, 8 * n_threads));
    hipLaunchKernelGGL(k_enum_flags, dim3(grid_for(n_threads)), dim3(kBlock), 0, s, bs, bs.elems, d_binom, n_cand, n_threads, flags, counts);
    LSK_LAUNCH_CHECK();
    if (lsk_internal_exclusive_scan_i64(n_threads, counts, offsets, s) != 0) return -1;
    int64_t last_off = 0, last_cnt = 0;
    LSK_CHECK(hipMemcpy(&last_off, offsets + (n_threads - 1), 8, hipMemcpyDeviceToHost));
    LSK_CHECK(hipMemcpy(&last_cnt, counts + (n_threads - 1), 8, hipMemcpyDeviceToHost));
    int64_t total = last_off + last_cnt;
    LSK_CHECK(hipMalloc((void **)d_states, total > 0 ? 8 * total : 8));
    hipLaunchKernelGGL(k_enum_write, dim3(grid_for(n_threads)), dim3(kBlock), 0, s, bs, d_binom, n_cand, n_threads, flags, offsets, *d_states);
    LSK_LAUNCH_CHECK();
    LSK_CHECK(hipStreamSynchronize(s));
    (void)hipFree(flags); (void)hipFree(counts); (void)hipFree(offsets);
    *count = total;
    return 0;
}

// masks[i] = owner of states[i]  (_enumStatesComputeMasksAndCounts, StatesEnumeration.chpl:138-156)
__global__ __launch_bounds__(kBlock) void k_masks(int64_t n, uint64_t const *__restrict__ states, Owner ow,
                                                  uint8_t *__restrict__ masks) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        masks[i] = (uint8_t)owner_of(states[i], ow);
}
extern "C" int lsk_masks(int64_t n, uint64_t const *states, int P, uint8_t *masks, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_masks, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, states, make_owner(P), masks);
    LSK_LAUNCH_CHECK();
    return 0;
}

__global__ __launch_bounds__(kBlock) void k_mask_counts(int64_t n, uint8_t const *__restrict__ masks,
                                                        unsigned long long *__restrict__ counts) {
    __shared__ unsigned s_cnt[LSK_MAX_PARTS];
    for (int d = threadIdx.x; d < LSK_MAX_PARTS; d += kBlock) s_cnt[d] = 0;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        atomicAdd(&s_cnt[masks[i]], 1u);
    __syncthreads();
    for (int d = threadIdx.x; d < LSK_MAX_PARTS; d += kBlock)
        if (s_cnt[d]) atomicAdd(&counts[d], (unsigned long long)s_cnt[d]);
}
extern "C" int lsk_mask_counts(int64_t n, uint8_t const *masks, int P, int64_t *h_counts, void *stream) {
    unsigned long long *d = nullptr;
    LSK_CHECK(hipMalloc((void **)&d, 8 * LSK_MAX_PARTS));
    LSK_CHECK(hipMemsetAsync(d, 0, 8 * LSK_MAX_PARTS, (hipStream_t)stream));
    if (n > 0) {
        hipLaunchKernelGGL(k_mask_counts, dim3(grid_for(n, kBlock * 16)), dim3(kBlock), 0, (hipStream_t)stream, n, masks, d);
        LSK_LAUNCH_CHECK();
    }
    unsigned long long h[LSK_MAX_PARTS];
    LSK_CHECK(hipStreamSynchronize((hipStream_t)stream));
    LSK_CHECK(hipMemcpy(h, d, 8 * LSK_MAX_PARTS, hipMemcpyDeviceToHost));
    (void)hipFree(d);
    for (int p = 0; p < P; ++p) h_counts[p] = (int64_t)h[p];
    return 0;
}

// Layout converters.  position[i] = rank of element i among the elements with the same mask that
// precede it (exclusive scan of the indicator, one pass per partition); block->hashed then is
// dest[mask[i]][position[i]] = src[i] and hashed->block its inverse.  Stable by construction, which
// is what keeps every hashed part ascending (BlockToHashed.chpl:87-208, HashedToBlock.chpl:67-153).
template <int ELT>
__global__ __launch_bounds__(kBlock) void k_permute(int64_t n, uint8_t const *__restrict__ masks, uint8_t p,
                                                    int64_t const *__restrict__ pos, char const *src, char *dst,
                                                    int to_hashed) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        if (masks[i] != p) continue;
        int64_t j = pos[i];
        if (ELT == 8) {
            if (to_hashed) ((uint64_t *)dst)[j] = ((uint64_t const *)src)[i];
            else ((uint64_t *)dst)[i] = ((uint64_t const *)src)[j];
        } else {
            if (to_hashed) ((ulonglong2 *)dst)[j] = ((ulonglong2 const *)src)[i];
            else ((ulonglong2 *)dst)[i] = ((ulonglong2 const *)src)[j];
        }
    }
}
static int permute_by_masks(int64_t n, uint8_t const *masks, int P, int elt_size, void const *block_const,
                            void *block_mut, void *const *parts, int to_hashed, hipStream_t s) {
    if (n == 0) return 0;
    if (elt_size != 8 && elt_size != 16) { snprintf(g_err, sizeof(g_err), "layout converters support 8/16-byte elements"); return -1; }
    int64_t *pos = nullptr, *tmp = nullptr;
    LSK_CHECK(hipMalloc((void **)&pos, 8 * n));
    if (hipMalloc((void **)&tmp, 8 * (size_t)scan_scratch_elems(n)) != hipSuccess) { (void)hipFree(pos); snprintf(g_err, sizeof(g_err), "layout converter: no memory"); return -1; }
    for (int p = 0; p < P; ++p) {
        ScanMaskIn f{masks, (uint8_t)p};
        if (exclusive_scan<ScanMaskIn>(f, n, pos, tmp, s) != 0) { (void)hipFree(pos); (void)hipFree(tmp); return -1; }
        char const *src = to_hashed ? (char const *)block_const : (char const *)parts[p];
        char *dst = to_hashed ? (char *)parts[p] : (char *)block_mut;
        if (elt_size == 8) hipLaunchKernelGGL(k_permute<8>, dim3(grid_for(n)), dim3(kBlock), 0, s, n, masks, (uint8_t)p, pos, src, dst, to_hashed);
        else hipLaunchKernelGGL(k_permute<16>, dim3(grid_for(n)), dim3(kBlock), 0, s, n, masks, (uint8_t)p, pos, src, dst, to_hashed);
        LSK_LAUNCH_CHECK();
    }
    LSK_CHECK(hipStreamSynchronize(s));
    if (tmp) (void)hipFree(tmp);
    (void)hipFree(pos);
    return 0;
}
extern "C" int lsk_block_to_hashed(int64_t n, uint8_t const *masks, int P, int elt_size, void const *src,
                                   void *const *h_dest, void *stream) {
    return permute_by_masks(n, masks, P, elt_size, src, nullptr, h_dest, 1, (hipStream_t)stream);
}
extern "C" int lsk_hashed_to_block(int64_t n, uint8_t const *masks, int P, int elt_size, void const *const *h_src,
                                   void *dest, void *stream) {
    return permute_by_masks(n, masks, P, elt_size, nullptr, dest, (void *const *)h_src, 0, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// deterministic vectors for tests / bench: the value depends only on (basis state, seed), so every
// partitioning of the same basis sees the same logical vector.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_fill_random(int64_t n, uint64_t const *__restrict__ states, uint64_t seed,
                                                        int cplx, double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t h = hash64_01(states[i] ^ (seed * 0x9e3779b97f4a7c15ULL + 0x632be59bd9b4e019ULL));
        double re = (double)(h >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        if (cplx) {
            uint64_t h2 = hash64_01(h ^ 0xd6e8feb86659fd93ULL);
            out[2 * i] = re;
            out[2 * i + 1] = (double)(h2 >> 11) * (1.0 / 9007199254740992.0) - 0.5;
        } else out[i] = re;
    }
}
extern "C" int lsk_fill_random(int64_t n, uint64_t const *states, uint64_t seed, int cplx, void *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_fill_random, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, states, seed, cplx, (double *)out);
    LSK_LAUNCH_CHECK();
    return 0;
}

