// k_rows.hip (one of the translation units kernels.hip was split into, round 6) -- hand-written HIP (gfx950 / CDNA4) kernels of the matrix-free y <- H x hot path and
// the thin extern-"C" shim (lsk_*) the C host side calls.  No MFMA anywhere: this path is irregular
// integer/bit work plus gather/scatter, bounded by HBM and (for symmetry-projected bases) integer ALU.
//
// Reference call sites replaced:
//   K1 localDiagonalBatch            /root/reference/src/DistributedMatrixVector.chpl:36-53
//   K2 computeOffDiag                /root/reference/src/BatchedOperator.chpl:82-116
//   K3 spin-inversion canonicalise   /root/reference/src/BatchedOperator.chpl:139-153
//   K4 symmetry projection           /root/reference/src/BatchedOperator.chpl:163-203
//   K5 hash64_01 % numLocales        /root/reference/src/StatesEnumeration.chpl:122-136
//   K6 radixOneStep                  /root/reference/src/DistributedMatrixVector.chpl:265-311
//   K7 ls_hs_state_index             /root/reference/src/DistributedMatrixVector.chpl:96-103
//   K8 ConcurrentAccessor.localAdd   /root/reference/src/ConcurrentAccessor.chpl:48-54
#include <hip/hip_runtime.h>
#include "lsk_dev.hpp"

// ---------------------------------------------------------------------------------------------
// K1: diagonal pass  y[i] = d(sigma_i) x[i]
// ---------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_diag(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag,
                                                 int64_t n, uint64_t const *__restrict__ reps,
                                                 double const *__restrict__ x, double *__restrict__ y) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t a = reps[i];
        double dr, di;
        diag_coeff<uint64_t, false>(runs, n_diag, diag, a, dr, di);
        if (CPLX) {
            double xr = x[2 * i], xi = x[2 * i + 1];
            y[2 * i] = dr * xr - di * xi;
            y[2 * i + 1] = dr * xi + di * xr;
        } else {
            y[i] = dr * x[i];
        }
    }
}

extern "C" int lsk_diag(lsk_operator op, int cplx, int64_t n, uint64_t const *reps, void const *x, void *y,
                        void *stream) {
    if (n == 0 || op.n_diag == 0) return 0;
    dim3 g(grid_for(n)), b(kBlock);
    if (cplx) hipLaunchKernelGGL(k_diag<true>, g, b, 0, (hipStream_t)stream, op.runs, op.n_diag, op.diag, n, reps, (double const *)x, (double *)y);
    else hipLaunchKernelGGL(k_diag<false>, g, b, 0, (hipStream_t)stream, op.runs, op.n_diag, op.diag, n, reps, (double const *)x, (double *)y);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Direct fused kernel: one partition, no permutation symmetries.  One row per lane, uniform loop
// over flip-mask groups.  Consecutive lanes hold consecutive basis states, so for a given group
// the active lanes' targets are (piecewise) consecutive as well: the scatter / gather coalesces.
//   PUSH: y[idx(beta)] += c x[i]                 (K2 + K3 + K7 + K8 fused; y holds the diagonal part)
//   PULL: y[i] = d x[i] + sum_g <i|H_g|i ^ x_g> x[idx(i ^ x_g)]   (no atomics, y written once).  The coefficient is the one the
//         reference's row expansion of the PARTNER state gives to row i -- group g evaluated at i ^ x_g -- so ANY operator on an
//         unprojected basis can be pulled, Hermitian or not (round 6; for a Hermitian one it equals conj(c)).  Inversion sectors
//         (INV) keep conj(c) of the row's own expansion: the projected matrix is pulled through its Hermiticity.
// Exchange runs (adjacent transpositions, e.g. the open bonds of a chain) take a branch-free inner
// loop: the rank of the target differs from the row's own rank by +-C(lo, k) with k = number of set
// bits below lo, which is carried incrementally; inactive lanes gather their own x and add 0, so
// the compiler can unroll and keep several gathers in flight.
// Row tiles come from a host-built tile map (lsk_tile_entry, lsk.h): block b runs on XCD b % 8 and
// walks that XCD's list of tiles, so the traversal order -- which decides what the XCD's L2 can
// reuse -- is data, not code.
// ---------------------------------------------------------------------------------------------
// DIRECTED: some run of the operator consists of directed pairs (lsk.h, LSK_GROUP_HOP_*): an instantiation of its own, so that the
// run loop of every other operator stays what it was -- at 82 SGPRs instead of <= 80 the occupancy API over-reports the resident
// blocks of this persistent kernel by one and a straggler round costs a third of its speed (measured in round 6: 14.0 -> 18.2 ms on
// chain_32; tests/test_host_tables.py::test_hot_kernel_register_budget)
template <typename W, bool CPLX, int INDEX, bool INV, bool PULL, bool REAL, bool DIRECTED>
__global__ __launch_bounds__(kBlock) void k_direct(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                   lsk_term const *__restrict__ off, int n_diag,
                                                   lsk_term const *__restrict__ diag, lsk_basis bs,
                                                   lsk_index ix, uint64_t const *__restrict__ tilemap,
                                                   int64_t slots_per_xcd, uint64_t const *__restrict__ reps,
                                                   double const *__restrict__ x, double *y, int *err, int gx,
                                                   int64_t const *__restrict__ row_gidx) {
    typedef typename WordTraits<W>::binom_t BT;
    typedef WordTraits<W> WT;
    __shared__ BT s_binom[INDEX == LSK_INDEX_COMBINADIC ? 64 * LSK_BINOM_K : 1];
    if (INDEX == LSK_INDEX_COMBINADIC) {
        for (int k = threadIdx.x; k < 64 * LSK_BINOM_K; k += blockDim.x) s_binom[k] = (BT)ix.binom[k];
        __syncthreads();
    }
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3; // grid is a multiple of 8
    const W site_mask = (W)bs.site_mask;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t]; // (first row, number of rows <= kBlock): lsk_tile_entry
        if ((uint64_t)threadIdx.x >= (slot >> 48)) continue;
        const int64_t i = (int64_t)(slot & 0xffffffffffffULL) + threadIdx.x;
        const W a = (W)__builtin_nontemporal_load(reps + i);
        // replicated-x mode (gx): rows are one hash partition, x is the whole vector in global ascending
        // order; ig = global index of this row (closed form, or precomputed for searched bases)
        int64_t ig = i;
        if (gx & 1) {
            if (INDEX == LSK_INDEX_COMBINADIC) ig = rank_combinadic_w<W, BT>(a, s_binom);
            else if (INDEX == LSK_INDEX_IDENTITY) ig = (int64_t)a;
            else ig = row_gidx[i];
        }
        double xr, xi = 0.0;
        if (CPLX) { xr = x[2 * ig]; xi = x[2 * ig + 1]; } else xr = x[ig];
        double accr = 0.0, acci = 0.0;
        if (PULL && n_diag == 0) { // no diagonal pass in the reference either: y is accumulated into (DMV:1062-1063)
            if (CPLX) { accr = y[2 * i]; acci = y[2 * i + 1]; } else accr = y[i];
        }
        if (PULL && n_diag > 0) {
            double dr, di;
            diag_coeff<W, REAL>(runs, n_diag, diag, a, dr, di);
            accr = dr * xr - (CPLX ? di * xi : 0.0);
            if (CPLX) acci = dr * xi + di * xr;
        }
        int g_begin = 0;
        if (INDEX == LSK_INDEX_COMBINADIC) {
            // ---- exchange runs: branch-free ------------------------------------------------------
            g_begin = runs.n_run_groups;
            const W tdiff = a ^ (a >> 1);
            for (int r = 0; r < runs.n_runs; ++r) {
                const int lo0 = runs.lo0[r], cnt = DIRECTED ? (runs.cnt[r] & 0xffff) : runs.cnt[r];
                // direction of the run's pairs (lsk.h): 0 exchange | 1 the LOWER site alone is the source pattern | 2 the upper one.  A row is
                // a TARGET of its partner's expansion in pull form, a source in push form: the pattern asked of the row flips with PULL
                const int dir = DIRECTED ? (runs.cnt[r] >> 16) : 0;
                const W want = (dir == 0) ? (W)0 : (W)(((dir == 1) != PULL) ? ~(W)0 : (W)0); // bit lo of an active row (dir != 0)
                const W adir = dir == 0 ? (W)~(W)0 : (W)~(a ^ want);                          // bit lo set <=> the row has the asked pattern
                const W tsel = (!DIRECTED || dir == 0) ? tdiff : (W)(tdiff & adir);           // bit lo set <=> pair lo is active for this row
                const double vr = runs.v_re[r], vi = REAL ? 0.0 : runs.v_im[r];
                int k = WT::popc(a & (W)(((uint64_t)1 << lo0) - 1));
                int lo_begin = lo0, lo_end = lo0 + cnt;
                if (sizeof(W) == 4 && PULL && REAL && !(gx & 1)) {
                    // Far pairs (lo >= hb): the 64 consecutive states of a wave nearly always agree on every
                    // bit >= hb, so such a pair is anti-aligned for the whole wave or for none of it.  The
                    // test, the bit count below the pair and the rank shift are then wave-uniform (scalar
                    // unit), an aligned pair issues no gather at all, and an anti-aligned one costs an
                    // add, an address and an fma per lane.
                    const int hb = (gx >> 24) & 63;
                    const int split = hb == 0 ? lo_end : (hb < lo0 ? lo0 : (hb > lo_end ? lo_end : hb));
                    const uint32_t a0 = __builtin_amdgcn_readfirstlane((uint32_t)a);
                    const bool uni = split < lo_end &&
                                     __builtin_amdgcn_ballot_w64((((uint32_t)a ^ a0) >> split) != 0) == 0;
                    if (uni) {
                        uint32_t m = (a0 ^ (a0 >> 1)) & (uint32_t)((((uint64_t)1 << lo_end) - 1) & ~(((uint64_t)1 << split) - 1));
                        if (DIRECTED && dir != 0) m &= ~(a0 ^ (uint32_t)want); // directed pairs: only rows with the asked pattern take part
                        const uint32_t i32 = (uint32_t)ig;
                        while (m) {
                            double xv[4], xw[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                xv[u] = 0.0;
                                xw[u] = 0.0;
                                if (m) {
                                    const int lo = __builtin_ctz(m);
                                    m &= m - 1;
                                    const int kk = bs.hamming_weight - __popc(a0 >> lo); // set bits below lo
                                    const uint32_t d = (uint32_t)s_binom[lo * LSK_BINOM_K + kk];
                                    const uint32_t idx = ((a0 >> lo) & 1) ? i32 + d : i32 - d;
                                    if (CPLX) {
                                        const double2 q = reinterpret_cast<double2 const *>(x)[idx];
                                        xv[u] = q.x;
                                        xw[u] = q.y;
                                    } else xv[u] = x[idx];
                                }
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                accr = fma(vr, xv[u], accr);
                                if (CPLX) acci = fma(vr, xw[u], acci);
                            }
                        }
                        lo_end = split;
                    }
                }
#pragma unroll 4
                for (int lo = lo_begin; lo < lo_end; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tsel >> lo) & 1;
                    const BT d = s_binom[lo * LSK_BINOM_K + k];
                    k += bit ? 1 : 0;
                    if (sizeof(W) == 4) {
                        const uint32_t i32 = (uint32_t)ig;
                        uint32_t idx = bit ? i32 + (uint32_t)d : i32 - (uint32_t)d;
                        if (PULL) {
                            idx = act ? idx : i32;
                            if (CPLX) {
                                double yr = x[2 * (size_t)idx], yi = x[2 * (size_t)idx + 1];
                                // v * x[idx]: an exchange group has the SAME amplitude for both patterns, so <i|H|j> = v (vi == 0 for a Hermitian H)
                                accr += act ? (vr * yr - vi * yi) : 0.0;
                                acci += act ? (vr * yi + vi * yr) : 0.0;
                            } else {
                                double yv = x[idx];
                                accr = fma(act ? vr : 0.0, yv, accr);
                            }
                        } else if (act) {
                            if (CPLX) {
                                atomic_add_f64(y + 2 * (size_t)idx, vr * xr - vi * xi);
                                atomic_add_f64(y + 2 * (size_t)idx + 1, vr * xi + vi * xr);
                            } else atomic_add_f64(y + idx, vr * xr);
                        }
                    } else {
                        int64_t idx = bit ? ig + (int64_t)d : ig - (int64_t)d;
                        if (PULL) {
                            idx = act ? idx : ig;
                            if (CPLX) {
                                double yr = x[2 * idx], yi = x[2 * idx + 1];
                                accr += act ? (vr * yr - vi * yi) : 0.0;
                                acci += act ? (vr * yi + vi * yr) : 0.0;
                            } else {
                                double yv = x[idx];
                                accr = fma(act ? vr : 0.0, yv, accr);
                            }
                        } else if (act) {
                            if (CPLX) {
                                atomic_add_f64(y + 2 * idx, vr * xr - vi * xi);
                                atomic_add_f64(y + 2 * idx + 1, vr * xi + vi * xr);
                            } else atomic_add_f64(y + idx, vr * xr);
                        }
                    }
                }
            }
        }
        // ---- everything else: generic groups ----------------------------------------------------
        for (int g = g_begin; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            // pull on an unprojected basis: <i|H_g|i ^ x_g>, the coefficient of the PARTNER's row expansion (any operator)
            constexpr bool PARTNER = PULL && !INV;
            group_coeff<REAL>(G, off, (uint64_t)(PARTNER ? a ^ (W)G.x : a), cr, ci);
            if (cr == 0.0 && (REAL || ci == 0.0)) continue;
            if (PARTNER) ci = -ci; // (the accumulation below multiplies by conj(c): written for the Hermitian shortcut of the INV sectors)
            W beta = a ^ (W)G.x;
            bool flipped = false;
            if (INV) { // K3
                W f = beta ^ site_mask;
                if (f < beta) { beta = f; flipped = true; cr *= (double)bs.spin_inversion; ci *= (double)bs.spin_inversion; }
            }
            int64_t idx;
            if (INDEX == LSK_INDEX_IDENTITY) idx = (int64_t)beta;
            else if (INDEX == LSK_INDEX_COMBINADIC) {
                if (G.adj >= 0 && !flipped && WT::popc(a & (W)G.x) == 1) {
                    // adjacent transposition: rank changes by C(lo, #set bits below lo)
                    int k = WT::popc(a & (W)(((uint64_t)1 << G.adj) - 1));
                    int64_t d = (int64_t)s_binom[G.adj * LSK_BINOM_K + k];
                    idx = ((a >> G.adj) & 1) ? ig + d : ig - d;
                } else {
                    // a state of another Hamming weight is outside the basis: ls_hs_state_index would
                    // return a negative index and the reference halts (DMV:115-118)
                    // (pull over a NON-Hermitian operator: a partner outside the basis that maps INTO it contributes nothing -- x has no
                    // such entry -- and its flag goes to a word nobody reads (lsk_direct, pull == 2); whether the operator maps the basis
                    // OUT of itself, the reference's halt, is what lsk_direct_validate checks at plan time.  Hermitian operators keep the
                    // run-time flag: the two are the same event.  No switch in the kernel: it has no scalar register to spare)
                    if (WT::popc(beta) != bs.hamming_weight) { atomicExch(err, 1); continue; }
                    idx = rank_combinadic_w<W, BT>(beta, s_binom);
                }
            } else {
                idx = search_index(ix, (uint64_t)beta);
                if (idx < 0) { atomicExch(err, 1); continue; } // DMV:115-118
            }
            if (PULL) {
                // conj(c) * x[idx]
                if (CPLX) {
                    double yr = x[2 * idx], yi = x[2 * idx + 1];
                    accr += cr * yr + ci * yi;
                    acci += cr * yi - ci * yr;
                } else accr += cr * x[idx];
            } else {
                if (CPLX) {
                    atomic_add_f64(y + 2 * idx, cr * xr - ci * xi);
                    atomic_add_f64(y + 2 * idx + 1, cr * xi + ci * xr);
                } else atomic_add_f64(y + idx, cr * xr);
            }
        }
        if (PULL) {
            if (CPLX) { y[2 * i] = accr; y[2 * i + 1] = acci; } else __builtin_nontemporal_store(accr, y + i);
        }
    }
}

// first pair index handled wave-uniformly by the 32-bit pull row kernels.  Measured on chain_32: 14 is best for k_direct,
// 12 (= every pair outside the LDS window) for k_chain_t.
constexpr int kDirectHighPair = 14;
template <typename W, bool CPLX, int INDEX, bool INV, bool PULL, bool DIRECTED>
static int launch_direct4(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_tilemap tm, uint64_t const *reps,
                          void const *x, void *y, int *d_err, void *stream, int gx, int64_t const *row_gidx) {
    int64_t gb = tm.slots_per_xcd * 8;
    // (f64 vectors only ever meet real operators: the plan refuses the other combination, so it is not instantiated)
    constexpr bool kCplxOp = CPLX;
    int64_t cap;
    if constexpr (kCplxOp) cap = op.is_real ? resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, true, DIRECTED>, gb) : resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, false, DIRECTED>, gb);
    else cap = resident_grid(k_direct<W, CPLX, INDEX, INV, PULL, true, DIRECTED>, gb);
    cap &= ~(int64_t)7; // XCD dealing needs a multiple of 8
    if (cap < 8) cap = 8;
    if (gb > cap) gb = cap; // persistent: one block per 256-row tile costs more than it gains here (13.3 -> 15.5 ms on chain_32)
    dim3 g((unsigned)gb), b(kBlock);
    gx = (gx & 1) | (kDirectHighPair << 24);
    if (op.is_real || !kCplxOp)
        hipLaunchKernelGGL((k_direct<W, CPLX, INDEX, INV, PULL, true, DIRECTED>), g, b, 0, (hipStream_t)stream, op.runs,
                           op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, ix, tm.entries, tm.slots_per_xcd, reps,
                           (double const *)x, (double *)y, d_err, gx, row_gidx);
    else if constexpr (kCplxOp)
        hipLaunchKernelGGL((k_direct<W, CPLX, INDEX, INV, PULL, false, DIRECTED>), g, b, 0, (hipStream_t)stream, op.runs,
                           op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, ix, tm.entries, tm.slots_per_xcd, reps,
                           (double const *)x, (double *)y, d_err, gx, row_gidx);
    LSK_LAUNCH_CHECK();
    return 0;
}
template <typename W, bool CPLX, int INDEX, bool INV, bool PULL>
static int launch_direct3(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_tilemap tm, uint64_t const *reps,
                          void const *x, void *y, int *d_err, void *stream, int gx, int64_t const *row_gidx) {
    bool directed = false; // (the host forms no directed runs on inversion bases: detect_runs)
    for (int r = 0; r < op.runs.n_runs; ++r) directed = directed || (op.runs.cnt[r] >> 16) != 0;
    if constexpr (INDEX == LSK_INDEX_COMBINADIC) { // (the run loop only exists for closed-form ranks: elsewhere every group is generic)
        if constexpr (!INV) {
            if (directed) return launch_direct4<W, CPLX, INDEX, INV, PULL, true>(op, bs, ix, tm, reps, x, y, d_err, stream, gx, row_gidx);
        } else if (directed) { snprintf(g_err, sizeof(g_err), "lsk_direct: directed runs on an inversion basis"); return -1; }
    }
    return launch_direct4<W, CPLX, INDEX, INV, PULL, false>(op, bs, ix, tm, reps, x, y, d_err, stream, gx, row_gidx);
}
template <typename W, bool CPLX, int INDEX>
static int launch_direct2(lsk_operator op, lsk_basis bs, lsk_index ix, int pull, lsk_tilemap n,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                          int64_t const *row_gidx) {
    const bool inv = bs.proj == LSK_PROJ_INVERSION;
    if (inv) {
        if (pull) return launch_direct3<W, CPLX, INDEX, true, true>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
        return launch_direct3<W, CPLX, INDEX, true, false>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
    }
    if (pull) return launch_direct3<W, CPLX, INDEX, false, true>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
    return launch_direct3<W, CPLX, INDEX, false, false>(op, bs, ix, n, reps, x, y, d_err, stream, gx, row_gidx);
}
template <bool CPLX, int INDEX>
static int launch_direct1(lsk_operator op, lsk_basis bs, lsk_index ix, int pull, lsk_tilemap n,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                          int64_t const *row_gidx) {
    // 32-bit states: every site, and every rank, fits 32 bits (C(32, 16) < 2^31)
    if constexpr (INDEX == LSK_INDEX_COMBINADIC)
        if (bs.number_sites <= 32) return launch_direct2<uint32_t, CPLX, INDEX>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    return launch_direct2<uint64_t, CPLX, INDEX>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
}
static int direct_dispatch(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int pull, lsk_tilemap n,
                           uint64_t const *reps, void const *x, void *y, int *d_err, void *stream, int gx,
                           int64_t const *row_gidx) {
    if (n.slots_per_xcd == 0) return 0;
    if (!n.entries) { snprintf(g_err, sizeof(g_err), "lsk_direct: no tile map"); return -1; }
    if (bs.proj == LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "lsk_direct: basis needs projection"); return -1; }
    switch (ix.kind) {
    case LSK_INDEX_IDENTITY:
        return cplx ? launch_direct1<true, LSK_INDEX_IDENTITY>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_IDENTITY>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    case LSK_INDEX_COMBINADIC:
        return cplx ? launch_direct1<true, LSK_INDEX_COMBINADIC>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_COMBINADIC>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    default:
        return cplx ? launch_direct1<true, LSK_INDEX_SEARCH>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx)
                    : launch_direct1<false, LSK_INDEX_SEARCH>(op, bs, ix, pull, n, reps, x, y, d_err, stream, gx, row_gidx);
    }
}
// Plan-time check of a pull plan over a NON-Hermitian operator: the reference's row expansion halts when some basis state is
// mapped out of the basis (negative index, DMV:115-118), and a push matvec reports exactly that.  A gather cannot see it -- it
// only ever asks which partners map INTO a row -- so the forward expansion is checked once, when the plan is made.
template <int INDEX>
__global__ __launch_bounds__(kBlock) void k_direct_validate(int n_groups, lsk_group const *__restrict__ groups, lsk_term const *__restrict__ off,
                                                            lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *__restrict__ reps, int *err) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t a = reps[i];
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            double cr, ci;
            group_coeff<false>(G, off, a, cr, ci);
            if (cr == 0.0 && ci == 0.0) continue;
            const uint64_t beta = a ^ G.x;
            bool inside;
            if (INDEX == LSK_INDEX_IDENTITY) inside = (beta & ~bs.site_mask) == 0;
            else if (INDEX == LSK_INDEX_COMBINADIC) inside = __popcll(beta) == bs.hamming_weight && (beta & ~bs.site_mask) == 0;
            else inside = search_index(ix, beta) >= 0;
            if (!inside) atomicExch(err, 1);
        }
    }
}
extern "C" int lsk_direct_validate(lsk_operator op, lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *reps, int *d_err, void *stream) {
    if (n <= 0 || op.n_groups <= 0) return 0;
    const dim3 g((unsigned)grid_for(n)), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    if (ix.kind == LSK_INDEX_IDENTITY) hipLaunchKernelGGL(k_direct_validate<LSK_INDEX_IDENTITY>, g, b, 0, s, op.n_groups, op.groups, op.off, bs, ix, n, reps, d_err);
    else if (ix.kind == LSK_INDEX_COMBINADIC) hipLaunchKernelGGL(k_direct_validate<LSK_INDEX_COMBINADIC>, g, b, 0, s, op.n_groups, op.groups, op.off, bs, ix, n, reps, d_err);
    else hipLaunchKernelGGL(k_direct_validate<LSK_INDEX_SEARCH>, g, b, 0, s, op.n_groups, op.groups, op.off, bs, ix, n, reps, d_err);
    LSK_LAUNCH_CHECK();
    return 0;
}
extern "C" int lsk_direct(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int pull, lsk_tilemap tm,
                          uint64_t const *reps, void const *x, void *y, int *d_err, void *stream) {
    // pull: 0 push | 1 pull, partners outside the basis are flagged (Hermitian operators) | 2 pull of a non-Hermitian operator: the
    // flag of a partner outside the basis lands in d_err[1], which nobody reads (d_err points at two ints)
    return direct_dispatch(op, bs, ix, cplx, pull != 0, tm, reps, x, y, pull == 2 ? d_err + 1 : d_err, stream, 0, nullptr);
}
// replicated-x pull: `reps` = the n rows of one partition, `x` = whole vector in global order, `ix` = index
// of the GLOBAL basis, row_gidx[i] = global index of row i (only read for SEARCH indices)
// ---------------------------------------------------------------------------------------------
// Staged PUSH (k_push_t; round 6): the reference's own formulation on one partition -- y[idx(beta)] += c x[i] with f64 atomics -- with
// the two things the generic row kernel leaves on the table:
//   * a tile of TILE consecutive rows keeps an LDS copy of y[tile - HALO, tile + TILE + HALO): a contribution whose target lies within
//     HALO rows of its source (the adjacent pairs at the low sites: |shift| = C(lo, k) <= 462 for lo < 12) is an LDS add, and the window
//     goes out once, one atomic per touched row, instead of one global atomic per contribution (chain_32: 6 of 16.5 per row);
//   * the diagonal part goes into the same window (y is cleared by the caller instead of being assigned by k_diag).
// Everything else is k_direct's push form: exchange runs branch-free with the rank shift C(lo, k) carried along, other groups by a full
// re-ranking, far targets by global atomics.  Real operators with undirected runs on the full fixed-weight basis (combinadic index),
// f64 or c128 vectors; LS_AMD_ROW_KERNEL=generic keeps k_direct.
// ---------------------------------------------------------------------------------------------
constexpr int kPushHalo = 512;
template <typename W, bool CPLX, int TILE>
__global__ __launch_bounds__(kBlock) void k_push_t(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                   lsk_term const *__restrict__ off, int n_diag, lsk_term const *__restrict__ diag,
                                                   lsk_basis bs, lsk_index ix, uint64_t const *__restrict__ tilemap, int64_t slots_per_xcd,
                                                   int64_t n, uint64_t const *__restrict__ reps, double const *__restrict__ x,
                                                   double *__restrict__ y, int *err) {
    typedef typename WordTraits<W>::binom_t BT;
    typedef WordTraits<W> WT;
    constexpr int HALO = kPushHalo;
    constexpr int WINDOW = TILE + 2 * HALO;
    constexpr int NC = CPLX ? 2 : 1;
    __shared__ BT s_binom[64 * LSK_BINOM_K];
    __shared__ double s_y[WINDOW * NC];
    for (int k = threadIdx.x; k < 64 * LSK_BINOM_K; k += kBlock) s_binom[k] = (BT)ix.binom[k];
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        const int cnt = (int)(slot >> 48);
        if (cnt == 0) continue; // block-uniform
        const int64_t i0 = (int64_t)(slot & 0xffffffffffffULL);
        const int64_t w0 = i0 - HALO;
        __syncthreads(); // the previous window has gone out (and the binomials are loaded)
        for (int j = threadIdx.x; j < WINDOW * NC; j += kBlock) s_y[j] = 0.0;
        __syncthreads();
#pragma unroll 1
        for (int sub = 0; sub < TILE / kBlock; ++sub) {
            const int r = sub * kBlock + threadIdx.x;
            if (r >= cnt) continue; // (no barrier inside)
            const int64_t i = i0 + r;
            const int own = HALO + r;
            const W a = (W)__builtin_nontemporal_load(reps + i);
            double xr, xi = 0.0;
            if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; } else xr = x[i];
            auto add = [&](int64_t idx, double vr, double vi) { // y[idx] += v: the window when idx lies inside it
                const int64_t o = idx - w0;
                if (o >= 0 && o < WINDOW) {
                    if (CPLX) { atomicAdd(&s_y[2 * o], vr); atomicAdd(&s_y[2 * o + 1], vi); } else atomicAdd(&s_y[o], vr);
                } else if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                else atomic_add_f64(y + idx, vr);
            };
            if (n_diag > 0) {
                double dr, di;
                diag_coeff<W, true>(runs, n_diag, diag, a, dr, di);
                if (CPLX) { atomicAdd(&s_y[2 * own], dr * xr); atomicAdd(&s_y[2 * own + 1], dr * xi); } else atomicAdd(&s_y[own], dr * xr);
            }
            // ---- exchange runs: the rank shift C(lo, k) carried along ------------------------------------------------------------
            const W tdiff = a ^ (a >> 1);
            for (int q = 0; q < runs.n_runs; ++q) {
                const int lo0 = runs.lo0[q], rc = runs.cnt[q];
                const double v = runs.v_re[q];
                int k = WT::popc(a & (W)(((uint64_t)1 << lo0) - 1));
#pragma unroll 4
                for (int lo = lo0; lo < lo0 + rc; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const int64_t d = (int64_t)s_binom[lo * LSK_BINOM_K + k];
                    k += bit ? 1 : 0;
                    if (act) add(bit ? i + d : i - d, v * xr, v * xi);
                }
            }
            // ---- everything else: generic groups ---------------------------------------------------------------------------------
            for (int g = runs.n_run_groups; g < n_groups; ++g) {
                lsk_group const G = groups[g];
                double cr, ci;
                group_coeff<true>(G, off, (uint64_t)a, cr, ci);
                if (cr == 0.0) continue;
                const W beta = a ^ (W)G.x;
                int64_t idx;
                if (G.adj >= 0 && WT::popc(a & (W)G.x) == 1) {
                    const int k = WT::popc(a & (W)(((uint64_t)1 << G.adj) - 1));
                    const int64_t d = (int64_t)s_binom[G.adj * LSK_BINOM_K + k];
                    idx = ((a >> G.adj) & 1) ? i + d : i - d;
                } else {
                    if (WT::popc(beta) != bs.hamming_weight) { atomicExch(err, 1); continue; } // DMV:115-118
                    idx = rank_combinadic_w<W, BT>(beta, s_binom);
                }
                add(idx, cr * xr, cr * xi);
            }
        }
        __syncthreads();
        // the window goes out: one atomic per touched row (the halos overlap the neighbouring tiles' windows and their far targets)
        for (int j = threadIdx.x; j < WINDOW; j += kBlock) {
            const int64_t row = w0 + j;
            if (row < 0 || row >= n) continue;
            if (CPLX) {
                const double vr = s_y[2 * j], vi = s_y[2 * j + 1];
                if (vr != 0.0) atomic_add_f64(y + 2 * row, vr);
                if (vi != 0.0) atomic_add_f64(y + 2 * row + 1, vi);
            } else {
                const double vr = s_y[j];
                if (vr != 0.0) atomic_add_f64(y + row, vr);
            }
        }
    }
}
extern "C" int lsk_push_tile_rows(int cplx) { return cplx ? 512 : 1024; }
// y must hold zeros (n_diag > 0: the diagonal part is added here) or what the matvec accumulates into (n_diag == 0, DMV:1062-1063)
extern "C" int lsk_push_staged(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, lsk_tilemap tm, int64_t n, uint64_t const *reps,
                               void const *x, void *y, int *d_err, void *stream) {
    if (n == 0 || tm.slots_per_xcd == 0) return 0;
    if (!tm.entries || ix.kind != LSK_INDEX_COMBINADIC || bs.proj != LSK_PROJ_NONE || !op.is_real) { snprintf(g_err, sizeof(g_err), "lsk_push_staged: plan out of range"); return -1; }
    const int64_t gb = tm.slots_per_xcd * 8; // one block per tile
    dim3 g((unsigned)gb), b(kBlock);
#define LSK_PT_ARGS op.runs, op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, ix, tm.entries, tm.slots_per_xcd, n, reps, (double const *)x, (double *)y, d_err
    if (bs.number_sites <= 32) {
        if (cplx) hipLaunchKernelGGL((k_push_t<uint32_t, true, 512>), g, b, 0, (hipStream_t)stream, LSK_PT_ARGS);
        else hipLaunchKernelGGL((k_push_t<uint32_t, false, 1024>), g, b, 0, (hipStream_t)stream, LSK_PT_ARGS);
    } else if (cplx) hipLaunchKernelGGL((k_push_t<uint64_t, true, 512>), g, b, 0, (hipStream_t)stream, LSK_PT_ARGS);
    else hipLaunchKernelGGL((k_push_t<uint64_t, false, 1024>), g, b, 0, (hipStream_t)stream, LSK_PT_ARGS);
#undef LSK_PT_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

extern "C" int lsk_direct_gx(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, lsk_tilemap tm, uint64_t const *reps,
                             int64_t const *row_gidx, void const *x_global, void *y, int *d_err, void *stream) {
    return direct_dispatch(op, bs, ix, cplx, 1, tm, reps, x_global, y, d_err, stream, 1, row_gidx);
}

// ---------------------------------------------------------------------------------------------
// Staged row kernel for chain-like operators (pull): the full fixed-Hamming-weight basis (row i = i-th state), exchange
// runs of adjacent pairs plus at most two other exchange pairs whose partner ranks the plan caches (anything else stays
// with k_direct).
//
// The generic row kernel above is bound by the vector-memory address unit (TA busy > 80 %: every gather is a wave
// instruction whatever it hits).  An adjacent pair (lo, lo + 1) moves a state by C(lo, k) <= C(11, 5) = 462 ranks when
// lo < 12, so those twelve gathers stay inside a window of the block's own rows +- 512: the block loads the window into
// LDS once (coalesced 16-byte loads -- the price of the old own-x load) and reads the near partners from LDS.  Pairs
// >= 12: the 64 consecutive states of a wave agree on every bit >= 12 in 92 % of the waves, so the anti-alignment test,
// the bit count below the pair and the rank shift are the same for the whole wave: an aligned far pair issues nothing,
// an anti-aligned one costs add + address + fma.  Waves that straddle two high parts take the per-lane loop.
// Measured on chain_32 f64 (gpurun_out/r2/ablate_sweep.log, pairs dropped from the top): streaming part 2.85 ms, the 12
// LDS pairs +2.2 ms, pairs 12..19 +1.25 ms (mostly L2 hits), pairs 20..30 +3.1-3.75 ms (every gather misses the L2:
// 27 GB at the fabric rate), the cached ring-closing pair +1.0-1.6 ms.
// ---------------------------------------------------------------------------------------------
constexpr int kChainHalo = 512;    // >= C(11, 5)
constexpr int kChainLdsPairs = 12; // pairs lo < 12 are served from the LDS window
constexpr int kChainFar = 12;      // far-pair gathers in flight per row before the first wait (8 / 10 / 12: 7.67 / 7.66 / 7.63 ms)

// W = state word (u32 up to 32 sites, u64 up to 64), R = rank type (u32 while the basis has < 2^32 - 1 states, else u64),
// CPLX = complex128 vectors (real operator; the window holds double2, gathers are 16 bytes per lane), TILE rows per
// block iteration (1024 for f64, 512 for c128: 5 blocks per CU).
__device__ __forceinline__ double cx_load_nt(double const *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double2 cx_load_nt(double2 const *p) {
    typedef double d2v __attribute__((ext_vector_type(2)));
    const d2v v = __builtin_nontemporal_load(reinterpret_cast<d2v const *>(p));
    return make_double2(v.x, v.y);
}
// Far pairs (>= hb) of a wave-row are priced lane-parallel: lane l looks at pair split + l of the wave-uniform state a0
// (anti-aligned?, bits below, binomial from LDS, sign), and the loop over the anti-aligned ones only does ballot-mask
// ctz + v_readlane + add + gather.  (The round-1 kernel did that arithmetic on the scalar unit, ~340 scalar
// instructions per 64 rows; moving it to the vector lanes changed nothing measurable: 10.70 vs 10.71 ms.)
constexpr int kChainFarC = 6; // complex: 6 x 16 bytes in flight per lane

// launch bounds, second argument = waves per SIMD the register allocation must allow.  256-thread blocks are admitted per CU
// up to floor(800 / (ceil(sgpr / 16) * 16 + 16)): at 98 SGPRs the 7th block does not fit while the occupancy API still
// answers 7 -- a straggler round of blocks, measured +24 % (13.1 vs 10.7 ms on chain_32)
// (f64: 7 blocks = what 22.5 KB of LDS admit; c128: bounds 4 / 5 / 6 measure 14.56 / 14.55 / 14.57 ms; the 64-bit f64
// instantiations -- 33..64 sites, no in-tree config -- need 6: at 7 they spill 20-40 bytes per lane to scratch)
template <typename W, typename R, bool CPLX, int TILE, bool REC>
__global__ __launch_bounds__(kBlock, (CPLX || sizeof(R) == 8 ? 5 : (sizeof(W) == 8 ? 6 : 7))) void k_chain_t(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag,
                                                    int hamming_weight, uint4 const *__restrict__ g_img, int img16, int kc,
                                                    int near_off,
                                                    uint64_t const *__restrict__ tilemap, int64_t slots_per_xcd, int64_t n,
                                                    uint64_t const *__restrict__ reps, void const *__restrict__ x_v,
                                                    void *__restrict__ y_v, int hb, int n_cached,
                                                    R const *__restrict__ cache, double cv0, double cv1, int64_t row0,
                                                    int64_t n_x) {
    typedef typename ChainX<CPLX>::type X;
    typedef WordTraits<W> WT;
    constexpr int NB = ChainTraits<W, R>::NB;
    // complex vectors: 11 LDS pairs and a 256-row halo (>= C(10, 5)) keep the block at 20.7 KB like the f64 one, i.e. more
    // resident blocks per CU; pair 11 then gathers from global memory (its partners are <= 462 rows away: L1 / L2 hits)
    constexpr int HALO = CPLX ? 256 : kChainHalo;
    constexpr int LDSP = CPLX ? 11 : kChainLdsPairs;
    constexpr int WINDOW = TILE + 2 * HALO + 2;
    constexpr int FAR = CPLX ? kChainFarC : kChainFar;
    constexpr R kNone = ~(R)0;
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    // profiling builds only (make ablate, LS_AMD_ABLATE & 128): ONE MORE 8-byte stream per row, prefetched like the records -- what a
    // byte per row costs this kernel, i.e. what computing sigma instead of loading it could save at best (scripts/chain_stream_cost.py)
    const bool extra_stream = kAblate && (n_cached & 0x100);
    // (profiling builds, LS_AMD_ABLATE & 4096 / 8192: the wave-uniform far gathers of pairs >= 20 / of all far pairs as non-temporal loads)
    const int nt_from = !kAblate ? 99 : ((n_cached & 0x200) ? 20 : ((n_cached & 0x400) ? 0 : 99));
    if (kAblate) n_cached &= 0xff;
    uint64_t ex_next = 0;
    // LDS image made once by the host (chain_lds_image): the binomial table in the rank type, NB rows of kc = weight + 2
    // columns, then the near-pair table (below); copied with 16-byte loads -- one block per tile means once per 1024 rows
    extern __shared__ uint4 s_img[];
    R const *const s_binom = reinterpret_cast<R const *>(s_img);
    uint2 const *const s_near = reinterpret_cast<uint2 const *>(reinterpret_cast<char const *>(s_img) + near_off);
    __shared__ X s_x[WINDOW + 1]; // last slot: 0, read by the lanes whose near pair is aligned
    for (int k = threadIdx.x; k < img16; k += kBlock) s_img[k] = g_img[k];
    (void)NB;
    if (threadIdx.x == 0) s_x[WINDOW] = cx_zero<X>();
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    const int lane = threadIdx.x & 63;
    tilemap += (int64_t)xcd * slots_per_xcd;
    uint32_t const *__restrict__ reps32 = reinterpret_cast<uint32_t const *>(reps);
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        const int cnt = (int)(slot >> 48);
        if (cnt == 0) continue; // block-uniform
        const int64_t i0 = (int64_t)(slot & 0xffffffffffffULL);
        const int64_t w0 = (row0 + i0 - HALO) & ~(int64_t)1; // first row of the window (even; may be < 0)
        W a_next = 0;
        R t0_next = kNone, t1_next = kNone;
        // REC: `reps` is the plan's fused record array, row -> sigma (low word) | partner rank of the first cached pair
        // (high word, ~0 = none): one 8-byte load per row instead of a 4-byte state and a 4-byte cache load
        auto load_row = [&](int64_t row, W &a_out, R &t0_out, R &t1_out) {
            if (REC) {
                const uint64_t rec = __builtin_nontemporal_load(reps + row);
                a_out = (W)(uint32_t)rec;
                t0_out = (R)(uint32_t)(rec >> 32);
                if (n_cached == 0) t0_out = kNone;
            } else {
                if (sizeof(W) == 4) a_out = (W)__builtin_nontemporal_load(reps32 + 2 * row); // low word only
                else a_out = (W)__builtin_nontemporal_load(reps + row);
                if (n_cached > 0) t0_out = __builtin_nontemporal_load(cache + row);
            }
            if (n_cached > 1) t1_out = __builtin_nontemporal_load(cache + (size_t)n + (size_t)row);
            if (kAblate && extra_stream) ex_next = __builtin_nontemporal_load(reps + (row < (n >> 1) ? row + (n >> 1) : row - (n >> 1)));
        };
        // Lanes past the end of a partial tile stay ACTIVE as ghosts of the tile's last row (they recompute it and
        // store nothing): the far pairs are priced lane-parallel, which needs every lane of a live wave.
        const int wave0 = (int)(threadIdx.x & ~63u);
        if (wave0 < cnt) { // first row of this thread: requested before the window is staged
            const int64_t rr = i0 + ((int)threadIdx.x < cnt ? (int)threadIdx.x : cnt - 1);
            load_row(rr, a_next, t0_next, t1_next);
        }
        __syncthreads(); // every wave is done with the previous window (and s_binom is loaded)
        if (CPLX) {
            for (int j = threadIdx.x; j < WINDOW; j += kBlock) {
                const int64_t row = w0 + j;
                s_x[j] = (row >= 0 && row < n_x) ? x[row] : cx_zero<X>();
            }
        } else {
            double const *xd = (double const *)x_v;
            double *sd = (double *)s_x;
            for (int j = 2 * threadIdx.x; j < WINDOW; j += 2 * kBlock) {
                const int64_t row = w0 + j;
                double2 v;
                if (row >= 0 && row + 1 < n_x) v = *reinterpret_cast<double2 const *>(xd + row);
                else { v.x = (row >= 0 && row < n_x) ? xd[row] : 0.0; v.y = (row + 1 >= 0 && row + 1 < n_x) ? xd[row + 1] : 0.0; }
                sd[j] = v.x;
                sd[j + 1] = v.y;
            }
        }
        __syncthreads();
        const int own0 = (int)(row0 + i0 - w0);
        X y_pending = cx_zero<X>(); // stored one iteration late (see k_chain)
        int64_t i_pending = -1;
#pragma unroll 1
        for (int sub = 0; sub < TILE / kBlock; ++sub) {
            const int r = sub * kBlock + threadIdx.x;
            if (i_pending >= 0) cx_store_nt(y + i_pending, y_pending);
            i_pending = -1;
            const W a = a_next;
            const R t0 = t0_next, t1 = t1_next;
            const uint64_t ex = ex_next;
            if (sub + 1 < TILE / kBlock && (sub + 1) * kBlock + wave0 < cnt) {
                const int64_t in = i0 + (r + kBlock < cnt ? r + kBlock : cnt - 1);
                load_row(in, a_next, t0_next, t1_next);
            }
            if (sub * kBlock + wave0 >= cnt) continue; // wave-uniform: the whole wave is past the end
            const bool ghost = r >= cnt;
            const int64_t i = i0 + (ghost ? cnt - 1 : r);
            const R ig = (R)(row0 + i);
            X g0 = cx_zero<X>(), g1 = cx_zero<X>();
            if (n_cached > 0) g0 = x[t0 != kNone ? t0 : ig];
            if (n_cached > 1) g1 = x[t1 != kNone ? t1 : ig];
            const int jr = own0 + (int)(i - i0);
            const X xr = s_x[jr];
            double dr, di;
            diag_coeff<W, true>(runs, n_diag, diag, a, dr, di);
            X acc = cx_scale(dr, xr);
            const W tdiff = a ^ (a >> 1);
            // The rows of a wave ascend, so every lane shares the common prefix of the first and the last state: pairs at or
            // above `ubit` (one past the highest bit on which those two differ) are wave-uniform.  92 % of the waves of
            // chain_32 have ubit <= 12; the others used to fall back to the per-lane loop for ALL their far pairs and now
            // only walk the pairs below ubit per lane.
            const W a0 = readfirstlane_t<W>(a);
            const W adiff = a0 ^ readlane_t<W>(a, 63);
            const int ubit = adiff == 0 ? 0 : (int)(8 * sizeof(W)) - (sizeof(W) == 4 ? __clz((int)(uint32_t)adiff) : __clzll((long long)(uint64_t)adiff));
            const R ig0 = readfirstlane_t<R>(ig);
            const uint32_t dl = (uint32_t)(ig - ig0); // 0..63: the far gathers address x as (uniform base) + dl
            for (int q = 0; q < runs.n_runs; ++q) {
                const int lo0 = runs.lo0[q];
                int lo_end = lo0 + runs.cnt[q];
                const double vr = runs.v_re[q];
                int k = WT::popc(a & (W)(((uint64_t)1 << lo0) - 1));
                int lo = lo0;
                const int e1 = lo_end < LDSP ? lo_end : LDSP;
                const int near_end = lo0 > e1 ? lo0 : e1;
                int split = hb == 0 ? lo_end : (hb < near_end ? near_end : (hb > lo_end ? lo_end : hb));
                if (hb != 0 && split < ubit) split = ubit < lo_end ? ubit : lo_end;
                // ---- far pairs: the 64 states of the wave agree on every bit >= split ------------------------------
                const bool uni = split < lo_end;
                unsigned long long m = 0;
                R off = 0;
                X xv[FAR];
#pragma unroll
                for (int u = 0; u < FAR; ++u) xv[u] = cx_zero<X>();
                if (uni) {
                    // lane l prices pair p = split + l of the common state
                    const int p = split + lane;
                    const bool in_run = p < lo_end;
                    const int ps = in_run ? p : 0;
                    const W hi = a0 >> ps;
                    const bool bit = hi & 1;
                    const bool act = in_run && (((hi >> 1) & 1) != (W)bit);
                    const int kk = hamming_weight - WT::popc(hi); // set bits below p
                    const R d = s_binom[ps * kc + (kk < 0 ? 0 : kk)];
                    off = bit ? d : (R)(0 - d);
                    m = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
                    for (int u = 0; u < FAR; ++u) {
                        if (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            X const *const src = x + (size_t)(R)(ig0 + readlane_t<R>(off, l)) + dl;
                            if (kAblate && split + l >= nt_from) xv[u] = cx_load_nt(src); else xv[u] = *src;
                        }
                    }
                    lo_end = split;
                }
                // ---- near pairs: partner inside the LDS window ----------------------------------------
                if (lo0 == 0 && e1 == LDSP) {
                    // The usual case, the run covers every LDS pair: the byte displacements of four pairs at a time come
                    // from a table indexed by the five state bits they touch and the number of set bits below them
                    // (chain_lds_image); an aligned pair holds a displacement that clamps to the zero slot.  3-4 VALU
                    // instructions per pair instead of 12 (two bit tests, binomial address, sign, select, scale).
                    const uint32_t al = (uint32_t)a;
                    const uint32_t jb = (uint32_t)jr * (uint32_t)sizeof(X);
                    constexpr uint32_t ZOFF = (uint32_t)WINDOW * (uint32_t)sizeof(X);
                    char const *const sb = reinterpret_cast<char const *>(s_x);
                    const uint2 q0 = s_near[al & 31u];
                    const uint2 q1 = s_near[32 + 32 * __popc(al & 15u) + ((al >> 4) & 31u)];
                    const uint2 q2 = s_near[192 + 32 * __popc(al & 255u) + ((al >> 8) & 31u)];
                    const uint32_t wq[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
                    // LDS reads in flight before their fmas (pair order kept).  Measured on chain_32 f64: 1 / 2 / 3 / 4 in flight
                    // 7.64 / 7.63 / 7.93 / 7.93 ms -- the registers of a deeper batch cost more than its latency hiding buys;
                    // c128 is indifferent (14.55-14.67 ms with or without the table)
                    constexpr int NBATCH = 2;
#pragma unroll
                    for (int p0 = 0; p0 < LDSP; p0 += NBATCH) {
                        X nv[NBATCH];
#pragma unroll
                        for (int u = 0; u < NBATCH; ++u) {
                            const int p = p0 + u;
                            const uint32_t w = wq[(p < LDSP ? p : 0) >> 1];
                            const int32_t d = (p & 1) ? ((int32_t)w >> 16) : (int32_t)(int16_t)(w & 0xffffu);
                            uint32_t o = jb + (uint32_t)d;
                            o = o < ZOFF ? o : ZOFF;
                            nv[u] = p < LDSP ? *reinterpret_cast<X const *>(sb + o) : cx_zero<X>();
                        }
#pragma unroll
                        for (int u = 0; u < NBATCH; ++u)
                            if (p0 + u < LDSP) cx_fma(vr, nv[u], acc);
                    }
                    lo = LDSP;
                    k = __popc(al & ((1u << LDSP) - 1u));
                }
#pragma unroll 4
                for (; lo < e1; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const int d = (int)s_binom[lo * kc + k];
                    k += bit ? 1 : 0;
                    const int j = bit ? jr + d : jr - d;
                    cx_fma(vr, s_x[act ? j : WINDOW], acc);
                }
#pragma unroll
                for (int u = 0; u < FAR; ++u) cx_fma(vr, xv[u], acc); // zero-filled slots included: counting the gathers and
                                                                      // branching around idle fmas measured slower (7.72 vs 7.59 ms)
                while (m) { // more than FAR anti-aligned far pairs
                    X xw[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        xw[u] = cx_zero<X>();
                        if (m) {
                            const int l = __builtin_ctzll(m);
                            m &= m - 1;
                            xw[u] = (x + (size_t)(R)(ig0 + readlane_t<R>(off, l)))[dl];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) cx_fma(vr, xw[u], acc);
                }
                // ---- middle pairs (and far pairs of a wave that straddles two high parts) --------------
#pragma unroll 4
                for (; lo < lo_end; ++lo) {
                    const bool bit = (a >> lo) & 1;
                    const bool act = (tdiff >> lo) & 1;
                    const R d = s_binom[lo * kc + k];
                    k += bit ? 1 : 0;
                    R idx = bit ? (R)(ig + d) : (R)(ig - d);
                    idx = act ? idx : ig;
                    cx_fma(act ? vr : 0.0, x[idx], acc);
                }
            }
            cx_fma(t0 != kNone ? cv0 : 0.0, g0, acc);
            cx_fma(t1 != kNone ? cv1 : 0.0, g1, acc);
            if (kAblate && extra_stream && ex == 0x0123456789abcdefULL) cx_fma(1.0, xr, acc); // (practically never: keeps the load alive)
            y_pending = acc;
            i_pending = ghost ? -1 : i;
        }
        if (i_pending >= 0) cx_store_nt(y + i_pending, y_pending);
    }
}

// cache[i] = rank of reps[i] ^ xmask when exactly one of the two bits of xmask is set in reps[i], else ~0;
// *flag is raised when a partner falls outside the basis (the caller then does not use the cache, and
// the generic path reports the error at run time as the reference does)
template <typename W, typename R>
__global__ __launch_bounds__(kBlock) void k_chain_cache(int64_t n, uint64_t const *__restrict__ reps, uint64_t xmask, uint64_t inv_mask,
                                                        int hamming_weight, uint64_t const *__restrict__ g_binom,
                                                        R *__restrict__ out, int *__restrict__ flag) {
    constexpr int NB = ChainTraits<W, R>::NB;
    __shared__ R s_binom[NB * LSK_BINOM_K];
    for (int k = threadIdx.x; k < NB * LSK_BINOM_K; k += blockDim.x) s_binom[k] = (R)g_binom[k];
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const W a = (W)reps[i];
        R t = ~(R)0;
        if (WordTraits<W>::popc(a & (W)xmask) == 1) {
            W beta = a ^ (W)xmask;
            if (inv_mask) { const W f = beta ^ (W)inv_mask; beta = f < beta ? f : beta; } // inversion sector: the canonical state of {beta, ~beta}
            if (WordTraits<W>::popc(beta) != hamming_weight) atomicExch(flag, 1);
            else t = (R)rank_combinadic_w<W, R>(beta, s_binom);
        }
        out[i] = t;
    }
}
extern "C" int lsk_chain_cache(lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *reps, uint64_t xmask, void *out,
                               int wide_ranks, int *d_flag, void *stream) {
    if (n == 0) return 0;
    dim3 g(grid_for(n)), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const uint64_t inv_mask = bs.proj == LSK_PROJ_INVERSION ? bs.site_mask : 0;
    if (bs.number_sites <= 32 && !wide_ranks)
        hipLaunchKernelGGL((k_chain_cache<uint32_t, uint32_t>), g, b, 0, s, n, reps, xmask, inv_mask, bs.hamming_weight, ix.binom, (uint32_t *)out, d_flag);
    else if (!wide_ranks)
        hipLaunchKernelGGL((k_chain_cache<uint64_t, uint32_t>), g, b, 0, s, n, reps, xmask, inv_mask, bs.hamming_weight, ix.binom, (uint32_t *)out, d_flag);
    else
        hipLaunchKernelGGL((k_chain_cache<uint64_t, uint64_t>), g, b, 0, s, n, reps, xmask, inv_mask, bs.hamming_weight, ix.binom, (uint64_t *)out, d_flag);
    LSK_LAUNCH_CHECK();
    return 0;
}

// fused per-row record of the 32-bit instantiation: sigma | partner rank << 32 (see k_chain_t, REC)
__global__ __launch_bounds__(kBlock) void k_chain_pack(int64_t n, uint64_t const *__restrict__ reps,
                                                       uint32_t const *__restrict__ cache, uint64_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
        out[i] = (uint64_t)(uint32_t)reps[i] | ((uint64_t)(cache ? cache[i] : 0xffffffffu) << 32);
}
extern "C" int lsk_chain_pack(int64_t n, uint64_t const *reps, void const *cache, uint64_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_chain_pack, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, (uint32_t const *)cache, out);
    LSK_LAUNCH_CHECK();
    return 0;
}

// binomial table in the rank type of the staged kernel: the u64 table as it is, or a u32 copy made once per device table
__global__ void k_binom_narrow(uint64_t const *__restrict__ in, uint32_t *__restrict__ out, int n) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) out[k] = (uint32_t)in[k];
}
template <typename R> static R const *chain_binom(uint64_t const *g_binom, hipStream_t stream);
template <> uint64_t const *chain_binom<uint64_t>(uint64_t const *g_binom, hipStream_t) { return g_binom; }
template <> uint32_t const *chain_binom<uint32_t>(uint64_t const *g_binom, hipStream_t stream) {
    static uint64_t const *src = nullptr;
    static uint32_t *narrow = nullptr;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    if (src != g_binom || !narrow) {
        if (!narrow && hipMalloc((void **)&narrow, sizeof(uint32_t) * 64 * LSK_BINOM_K) != hipSuccess) return nullptr;
        hipLaunchKernelGGL(k_binom_narrow, dim3(4), dim3(kBlock), 0, stream, g_binom, narrow, 64 * LSK_BINOM_K);
        src = g_binom;
    }
    return narrow;
}

// LDS image of k_chain_t, made once per (rank type, rows, weight, vector type) and kept on the device:
//   [rows][kc] binomials C(n, k), k < kc = weight + 2, in the rank type (low 32 bits for 32-bit ranks), padded to 16 bytes;
//   near-pair table: for g = 0..2 (pairs 4g..4g+3), kidx = number of set bits below bit 4g (0..4g), pat = state bits
//   4g..4g+4: four int16 = signed byte displacement of the partner inside the LDS window of x (elem bytes per row,
//   +C(lo, k) rows when the lower bit of the pair is set, -C(lo, k) when the upper one is), 0x7000 for an aligned pair
//   or a pair >= ldsp (added to any row offset it lands past the window and clamps to the zero slot).
//   Entry index = {0, 32, 192}[g] + 32 kidx + pat.
static int chain_near_fill(uint64_t const (*C)[LSK_BINOM_K], int elem, int ldsp, int16_t *near) {
    int const base[3] = {0, 32, 192};
    for (int g = 0; g < 3; ++g)
        for (int kidx = 0; kidx <= 4 * g; ++kidx)
            for (int pat = 0; pat < 32; ++pat)
                for (int p = 0; p < 4; ++p) {
                    int const lo = 4 * g + p;
                    int const bit = (pat >> p) & 1, nxt = (pat >> (p + 1)) & 1;
                    int const k = kidx + __builtin_popcount(pat & ((1 << p) - 1));
                    int64_t const d = (int64_t)C[lo][k < LSK_BINOM_K ? k : 0] * elem;
                    int16_t v = 0x7000;
                    if (lo < ldsp && bit != nxt && k <= lo) {
                        if (d >= 0x7000) { snprintf(g_err, sizeof(g_err), "chain_near_fill: displacement out of range"); return -1; }
                        v = (int16_t)(bit ? d : -d);
                    }
                    near[4 * (base[g] + 32 * kidx + pat) + p] = v;
                }
    return 0;
}
static void chain_binomials(uint64_t (*C)[LSK_BINOM_K]) {
    for (int n = 0; n < 64; ++n)
        for (int k = 0; k < LSK_BINOM_K; ++k) C[n][k] = k == 0 ? 1 : (n == 0 ? 0 : C[n - 1][k - 1] + C[n - 1][k]);
}
// host test hook: the near-pair table alone (480 entries of four int16)
extern "C" int lsk_test_chain_near_table(int elem, int ldsp, int16_t *out) {
    uint64_t C[64][LSK_BINOM_K];
    chain_binomials(C);
    return chain_near_fill(C, elem, ldsp, out);
}
struct ChainImage { int rsize, rows, kc, elem, ldsp, device; uint4 *dev; int bytes, near_off; };
static int chain_lds_image(int rsize, int rows, int weight, int elem, int ldsp, ChainImage *out) {
    static std::vector<ChainImage> cache;
    static std::mutex lock;
    std::lock_guard<std::mutex> guard(lock);
    int kc = (weight < 0 ? rows : weight) + 2;
    if (kc > LSK_BINOM_K) kc = LSK_BINOM_K;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = 0; // the image lives in the memory of the device it was made on
    for (ChainImage const &c : cache)
        if (c.rsize == rsize && c.rows == rows && c.kc == kc && c.elem == elem && c.ldsp == ldsp && c.device == device) { *out = c; return 0; }
    uint64_t C[64][LSK_BINOM_K];
    chain_binomials(C);
    int const near_off = (rows * kc * rsize + 15) & ~15;
    int const bytes = near_off + 480 * 8;
    std::vector<unsigned char> img((size_t)bytes, 0);
    for (int n = 0; n < rows; ++n)
        for (int k = 0; k < kc; ++k) {
            if (rsize == 4) reinterpret_cast<uint32_t *>(img.data())[n * kc + k] = (uint32_t)C[n][k];
            else reinterpret_cast<uint64_t *>(img.data())[n * kc + k] = C[n][k];
        }
    if (chain_near_fill(C, elem, ldsp, reinterpret_cast<int16_t *>(img.data() + near_off)) != 0) return -1;
    ChainImage c = {rsize, rows, kc, elem, ldsp, device, nullptr, bytes, near_off};
    if (hipMalloc((void **)&c.dev, (size_t)bytes) != hipSuccess) { snprintf(g_err, sizeof(g_err), "chain_lds_image: no device memory"); return -1; }
    if (hipMemcpy(c.dev, img.data(), (size_t)bytes, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(c.dev); snprintf(g_err, sizeof(g_err), "chain_lds_image: copy failed"); return -1; }
    cache.push_back(c);
    *out = c;
    return 0;
}

template <typename W, typename R, bool CPLX, int TILE, bool REC>
static int launch_chain(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_tilemap tm, int64_t n, uint64_t const *reps,
                        int64_t row0, int64_t n_x, void const *x, void *y, int n_cached, void const *cache, double cv0,
                        double cv1, void *stream) {
    int64_t gb = tm.slots_per_xcd * 8;
    ChainImage img;
    if (chain_lds_image((int)sizeof(R), ChainTraits<W, R>::NB, bs.hamming_weight, CPLX ? 16 : 8, CPLX ? 11 : kChainLdsPairs, &img) != 0) return -1;
    // One block per tile (block b -> tile b / 8 of XCD list b % 8), NOT a persistent grid: measured on chain_32 8.47 vs 10.85 ms
    // (f64) and 15.7 vs 20.8 ms (c128).  Persistent blocks start together and stay phase-locked (window load, LDS pairs, far
    // gathers), so the phases' costs add up; blocks dispatched one by one as others retire drift apart and the memory phases of
    // some overlap the LDS / ALU phases of others.
    hipLaunchKernelGGL((k_chain_t<W, R, CPLX, TILE, REC>), dim3((unsigned)gb), dim3(kBlock), (size_t)img.bytes, (hipStream_t)stream, op.runs,
                       op.n_diag, op.diag, bs.hamming_weight, img.dev, img.bytes / 16, img.kc, img.near_off, tm.entries, tm.slots_per_xcd, n, reps, x, y,
                       kChainLdsPairs, n_cached | ((kAblate && (bs.debug_ablate & 128)) ? 0x100 : 0) | ((kAblate && (bs.debug_ablate & 4096)) ? 0x200 : 0) |
                           ((kAblate && (bs.debug_ablate & 8192)) ? 0x400 : 0), (R const *)cache, cv0, cv1, row0, n_x);
    LSK_LAUNCH_CHECK();
    return 0;
}

// rows per tile: 1024 (f64) / 512 (c128).  Measured r2 on chain_32: doubling them (fewer blocks, 1.5x instead of 2x window
// loads, but 5 instead of 7 blocks per CU) is slower, 8.72 vs 8.26 ms (f64), 15.5 vs 15.4 ms (c128).
extern "C" int lsk_chain_tile_rows(int cplx) { return cplx ? 512 : 1024; }

extern "C" int lsk_chain(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int wide_ranks, int fused_records, lsk_tilemap tm,
                         int64_t n, uint64_t const *reps, int64_t row0, int64_t n_x, void const *x, void *y, int n_cached,
                         void const *cache, double cv0, double cv1, void *stream) {
    if (n == 0 || tm.slots_per_xcd == 0) return 0;
    const bool narrow = bs.number_sites <= 32 && !wide_ranks;
#define LSK_CHAIN_ARGS op, bs, ix, tm, n, reps, row0, n_x, x, y, n_cached, cache, cv0, cv1, stream
    if (fused_records) { // `reps` is the record array made by lsk_chain_pack (32-bit states and ranks only)
        if (!narrow || cplx) { snprintf(g_err, sizeof(g_err), "lsk_chain: fused records need 32-bit states and ranks and f64 vectors"); return -1; }
        return launch_chain<uint32_t, uint32_t, false, 1024, true>(LSK_CHAIN_ARGS);
    }
    if (narrow) {
        if (!cplx) { snprintf(g_err, sizeof(g_err), "lsk_chain: 32-bit states and ranks with f64 vectors run on fused records (lsk_chain_pack)"); return -1; }
        return launch_chain<uint32_t, uint32_t, true, 512, false>(LSK_CHAIN_ARGS);
    }
    if (!wide_ranks) return cplx ? launch_chain<uint64_t, uint32_t, true, 512, false>(LSK_CHAIN_ARGS) : launch_chain<uint64_t, uint32_t, false, 1024, false>(LSK_CHAIN_ARGS);
    return cplx ? launch_chain<uint64_t, uint64_t, true, 512, false>(LSK_CHAIN_ARGS) : launch_chain<uint64_t, uint64_t, false, 1024, false>(LSK_CHAIN_ARGS);
#undef LSK_CHAIN_ARGS
}

// ---------------------------------------------------------------------------------------------
// Staged row kernel for ANY set of exchange pairs (k_pairs_t): real Hermitian operators whose off-diagonal part is a sum of
// v_p (|01><10| + |10><01|) over arbitrary site pairs (i_p, j_p) and whose diagonal is a sum of vz_p s_i s_j over the same
// pairs -- the Heisenberg / XXZ model on any lattice (square, kagome, J1-J2 rings, ...), on the full fixed-weight basis of
// <= 32 sites, pull form.  k_chain_t is the special case "adjacent pairs", which is faster still for rings; everything that is
// not a ring used to run the generic row kernel (k_direct: one full combinadic re-ranking per non-adjacent non-zero).
//
// A state is high | low with low = bits 0..10.  In the ascending order all 11-bit words of weight kl under one `high` are a
// contiguous block of C(11, kl) <= 462 rows (kl = weight - popcount(high)), and rank = blockstart(high) + rank_low[low]
// (rank_low: 2048 x u16 in LDS).  The 64 rows of a wave are consecutive, so they almost always lie in ONE block; a wave that
// straddles blocks is processed block segment by block segment (same code, partial lane mask).  Per segment the pairs fall
// into three classes:
//   NEAR      both sites in `low`: the partner is in the same block, at rank_low[low ^ m] - rank_low[low] rows: LDS window
//             of x (+-512 rows), two LDS reads and a handful of VALU instructions per pair and lane;
//   HIGH      both sites in `high`: anti-alignment and the rank shift are the same for every lane of the segment: priced once,
//             lane-parallel (lane l <-> pair l: the shift is a sum over the set bits between the two sites), then a loop over the
//             anti-aligned pairs only -- readlane, add, gather of 64 CONSECUTIVE elements of x, fma;
//   STRADDLE  i in `low`, j in `high`: the partner block (high ^ bit j, kl -+ 1) is the same for every lane: its start is priced
//             once, lane-parallel; per lane the partner is blockstart + rank_low[low ^ bit i] (a gather inside a <= 3.7 KB block).
// The diagonal comes out of the same loops: d(a) = sum_p vz_p - 2 sum_{p anti-aligned} vz_p.
// ---------------------------------------------------------------------------------------------
constexpr int kPairLowBits = 11;
constexpr int kPairHalo = 512; // >= C(11, 5)
constexpr int kPairFar = 8;    // gathers of HIGH pairs in flight per lane
enum { PAIR_NEAR = 0, PAIR_STRADDLE = 1, PAIR_HIGH = 2 };

// SW = state word: u32 for <= 32 sites (the plan's 4-byte copy of the states), u64 for 33..64 sites (round 6: the caller's 8-byte
// representatives as they are; ranks stay 32-bit -- fewer than 2^32 states -- so every binomial a valid state touches fits u32)
template <typename SW, bool CPLX, int TILE>
__global__ __launch_bounds__(kBlock, (CPLX ? 4 : 6)) void k_pairs_t(lsk_pairplan pp, int hamming_weight, uint64_t const *__restrict__ tilemap,
                                                              int64_t slots_per_xcd, int64_t n, void const *__restrict__ x_v,
                                                              void *__restrict__ y_v) {
    typedef typename ChainX<CPLX>::type X;
    constexpr int HALO = kPairHalo;
    constexpr int WINDOW = TILE + 2 * HALO + 2;
    constexpr int LOWMASK = (1 << kPairLowBits) - 1;
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    __shared__ X s_x[WINDOW + 1]; // last slot: 0
    __shared__ uint16_t s_rl[1 << kPairLowBits];
    constexpr int NBITS = 8 * (int)sizeof(SW);
    constexpr SW ONE = (SW)1;
    auto popc = [](SW v) { return sizeof(SW) == 4 ? __popc((uint32_t)v) : __popcll((uint64_t)v); };
    auto ctz = [](SW v) { return sizeof(SW) == 4 ? __builtin_ctz((uint32_t)v) : __builtin_ctzll((uint64_t)v); };
    SW const *__restrict__ states = (SW const *)pp.states;
    __shared__ uint32_t s_binom[NBITS * LSK_PAIR_KC];
    __shared__ lsk_pair s_pairs[LSK_MAX_PAIRS];
    const int n_near = pp.n_near, n_str = pp.n_str, n_high = pp.n_high;
    const int kc = LSK_PAIR_KC;
    for (int k = threadIdx.x; k < (1 << kPairLowBits); k += kBlock) s_rl[k] = pp.rank_low[k];
    for (int k = threadIdx.x; k < NBITS * kc; k += kBlock) s_binom[k] = pp.binom[k];
    for (int k = threadIdx.x; k < n_near + n_str + n_high; k += kBlock) s_pairs[k] = pp.pairs[k];
    if (threadIdx.x == 0) s_x[WINDOW] = cx_zero<X>();
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    const int lane = threadIdx.x & 63;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        const int cnt = (int)(slot >> 48);
        if (cnt == 0) continue;
        const int64_t i0 = (int64_t)(slot & 0xffffffffffffULL);
        const int64_t w0 = (i0 - HALO) & ~(int64_t)1;
        __syncthreads(); // every wave is done with the previous window (and the tables are loaded)
        if (CPLX) {
            for (int j = threadIdx.x; j < WINDOW; j += kBlock) {
                const int64_t row = w0 + j;
                s_x[j] = (row >= 0 && row < n) ? x[row] : cx_zero<X>();
            }
        } else {
            double const *xd = (double const *)x_v;
            double *sd = (double *)s_x;
            for (int j = 2 * threadIdx.x; j < WINDOW; j += 2 * kBlock) {
                const int64_t row = w0 + j;
                double2 v;
                if (row >= 0 && row + 1 < n) v = *reinterpret_cast<double2 const *>(xd + row);
                else { v.x = (row >= 0 && row < n) ? xd[row] : 0.0; v.y = (row + 1 >= 0 && row + 1 < n) ? xd[row + 1] : 0.0; }
                sd[j] = v.x;
                sd[j + 1] = v.y;
            }
        }
        __syncthreads();
        const int own0 = (int)(i0 - w0);
        const int wave0 = (int)(threadIdx.x & ~63u);
#pragma unroll 1
        for (int sub = 0; sub < TILE / kBlock; ++sub) {
            if (sub * kBlock + wave0 >= cnt) break; // wave-uniform: the whole wave is past the end
            const int r = sub * kBlock + threadIdx.x;
            const bool ghost = r >= cnt; // lanes past the end stay active as copies of the last row (they store nothing)
            const int64_t i = i0 + (ghost ? cnt - 1 : r);
            const SW a = __builtin_nontemporal_load(states + i);
            const uint32_t ig = (uint32_t)i;
            const uint32_t low = (uint32_t)a & LOWMASK;
            const SW hi = a >> kPairLowBits;
            const int jr = own0 + (int)(i - i0);
            const X xr = s_x[jr];
            X acc = cx_zero<X>();
            double dsub = 0.0; // sum of vz over this row's anti-aligned pairs
            // ---- NEAR pairs ---------------------------------------------------------------------------------------------
            const int rl0 = (int)s_rl[low];
            for (int p = 0; p < n_near; ++p) {
                lsk_pair const P = s_pairs[p];
                const uint32_t m = (1u << P.i) | (1u << P.j);
                const bool act = __popc(low & m) == 1;
                const int r1 = (int)s_rl[low ^ m];
                const int o = jr + (r1 - rl0);
                cx_fma(P.v, s_x[act ? o : WINDOW], acc);
                dsub += act ? P.vz : 0.0;
            }
            // ---- block segments: lanes that share `hi` -----------------------------------------------------------------------
            unsigned long long pending = __builtin_amdgcn_ballot_w64(true);
            while (pending) {
                const int l0 = __builtin_ctzll(pending);
                const SW href = readlane_t<SW>(hi, l0);
                const bool inseg = hi == href;
                pending &= ~__builtin_amdgcn_ballot_w64(inseg);
                const int kl = hamming_weight - popc(href); // set bits of `low`, the same for every lane of the segment
                double dz_u = 0.0;
                // ---- HIGH pairs: priced once, lane l <-> pair pass + l -------------------------------------------------------
                for (int pass = 0; pass < n_high; pass += 64) {
                    const int q = pass + lane;
                    const bool in = q < n_high;
                    lsk_pair const P = s_pairs[n_near + n_str + (in ? q : 0)];
                    const int pi = P.i - kPairLowBits, pj = P.j - kPairLowBits; // positions inside `hi`
                    const uint32_t bi = (uint32_t)(href >> pi) & 1u, bj = (uint32_t)(href >> pj) & 1u;
                    const bool act = in && bi != bj;
                    // rank of the configuration "bit at i" minus rank of "bit at j": only the set bits at or above i matter
                    int k = kl + popc(href & ((ONE << pi) - ONE)); // set bits of the state below site i
                    SW between = href & ((ONE << pj) - ONE) & ~(((SW)2 << pi) - ONE);
                    int64_t lowcfg = (int64_t)s_binom[P.i * kc + min(k + 1, kc - 1)], highcfg = 0;
                    int tt = 0;
                    while (between) {
                        const int b = ctz(between) + kPairLowBits;
                        between &= between - ONE;
                        ++tt;
                        lowcfg += (int64_t)s_binom[b * kc + min(k + 1 + tt, kc - 1)];
                        highcfg += (int64_t)s_binom[b * kc + min(k + tt, kc - 1)];
                    }
                    highcfg += (int64_t)s_binom[P.j * kc + min(k + tt + 1, kc - 1)];
                    const int32_t delta = (int32_t)(bi ? highcfg - lowcfg : lowcfg - highcfg); // partner rank - own rank
                    unsigned long long m = __builtin_amdgcn_ballot_w64(act);
                    while (m) {
                        X xv[kPairFar];
                        double vv[kPairFar];
#pragma unroll
                        for (int u = 0; u < kPairFar; ++u) {
                            xv[u] = cx_zero<X>();
                            vv[u] = 0.0;
                            if (m) {
                                const int l = __builtin_ctzll(m);
                                m &= m - 1;
                                const int32_t d = __builtin_amdgcn_readlane(delta, l);
                                vv[u] = readlane_f64(P.v, l);
                                dz_u += readlane_f64(P.vz, l);
                                xv[u] = x[inseg ? (uint32_t)(ig + (uint32_t)d) : ig];
                            }
                        }
#pragma unroll
                        for (int u = 0; u < kPairFar; ++u) cx_fma(inseg ? vv[u] : 0.0, xv[u], acc);
                    }
                }
                // ---- STRADDLE pairs: the partner block is priced once, the place inside it per lane ----------------------------
                for (int pass = 0; pass < n_str; pass += 64) {
                    const int q = pass + lane;
                    const bool in = q < n_str;
                    lsk_pair const P = s_pairs[n_near + (in ? q : 0)];
                    const int pj = P.j - kPairLowBits;
                    const uint32_t bj = (uint32_t)(href >> pj) & 1u;
                    const SW h2 = href ^ (ONE << pj);
                    const int kl2 = bj ? kl + 1 : kl - 1; // a bit comes down into `low`, or leaves it
                    uint32_t base = 0;
                    {
                        SW hb = h2;
                        int idx = kl2;
                        while (hb) {
                            const int b = ctz(hb) + kPairLowBits;
                            hb &= hb - ONE;
                            ++idx;
                            base += s_binom[b * kc + min(max(idx, 0), kc - 1)];
                        }
                    }
                    const int np = min(64, n_str - pass);
                    for (int l = 0; l < np; ++l) {
                        const int pi = __builtin_amdgcn_readlane((int)P.i, l);
                        const uint32_t sbj = (uint32_t)__builtin_amdgcn_readlane((int)bj, l);
                        const uint32_t sbase = (uint32_t)__builtin_amdgcn_readlane((int)base, l);
                        const double sv = readlane_f64(P.v, l), svz = readlane_f64(P.vz, l);
                        const bool act = inseg && ((low >> pi) & 1u) != sbj;
                        const uint32_t idx = act ? sbase + (uint32_t)s_rl[low ^ (1u << pi)] : ig;
                        cx_fma(act ? sv : 0.0, x[idx], acc);
                        dsub += act ? svz : 0.0;
                    }
                }
                dsub += inseg ? dz_u : 0.0;
            }
            cx_fma(pp.dsum - 2.0 * dsub, xr, acc);
            if (!ghost) cx_store_nt(y + i, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same operators FAR FROM HALF FILLING (k_pairs_row; round 6): the blocks k_pairs_t walks shrink to a handful of rows there
// (C(11, kl), kl ~ 11 w / L) and the generic row kernel, which re-ranks every partner from scratch behind a divergent branch,
// used to be the faster of the two (6 x 6 square lattice, weight 6: 0.25 against 1.09 ms).  One row per lane, every pair
// branch-free, the rank of a partner in O(1):
//   the particles of a row sit at pos_0 < pos_1 < ..., rank = sum_t C(pos_t, t + 1).  A pair (i < j) whose two bits differ moves ONE
//   particle between i and j past the m particles in between, and each of those changes its index by one:
//     particle k at i -> j (ends as particle k + m):  d = C(j, k+m+1) - C(i, k+1) - sum_{t = k+1 .. k+m} [C(pos_t, t+1) - C(pos_t, t)]
//     particle k+m at j -> i (ends as particle k):    d = C(i, k+1) - C(j, k+m+1) - sum_{t = k .. k+m-1} [C(pos_t, t+1) - C(pos_t, t+2)]
//   with k = particles below i.  The two sums are differences of per-row prefix arrays A[], B[] (weight + 1 entries each) that the
//   lane writes to LDS once per row (entry u of lane l at [u][l]: conflict-free), so a pair costs three popcounts, two reads of the
//   binomial table, two of the prefix arrays, one gather of x and an fma -- and nothing depends on the previous pair: the
//   compiler keeps several gathers in flight.  32-bit arithmetic throughout (fewer than 2^32 states; everything is additive).
// ---------------------------------------------------------------------------------------------
template <typename SW, bool CPLX>
__global__ __launch_bounds__(kBlock) void k_pairs_row(lsk_pairplan pp, int hamming_weight, uint64_t const *__restrict__ tilemap,
                                                      int64_t slots_per_xcd, void const *__restrict__ x_v, void *__restrict__ y_v) {
    typedef typename ChainX<CPLX>::type X;
    constexpr int NBITS = 8 * (int)sizeof(SW);
    constexpr int kc = LSK_PAIR_KC;
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    extern __shared__ uint32_t s_prefix[]; // A: [hamming_weight + 1][kBlock], then B the same
    __shared__ uint32_t s_binom[NBITS * kc];
    auto popc = [](SW v) { return sizeof(SW) == 4 ? __popc((uint32_t)v) : __popcll((uint64_t)v); };
    auto ctz = [](SW v) { return sizeof(SW) == 4 ? __builtin_ctz((uint32_t)v) : __builtin_ctzll((uint64_t)v); };
    for (int k = threadIdx.x; k < NBITS * kc; k += kBlock) s_binom[k] = pp.binom[k];
    __syncthreads();
    uint32_t *const sA = s_prefix + threadIdx.x;
    uint32_t *const sB = sA + (hamming_weight + 1) * kBlock;
    SW const *__restrict__ states = (SW const *)pp.states;
    lsk_pair_row const *__restrict__ rows = pp.rows;
    const int n_pairs = pp.n_near + pp.n_str + pp.n_high;
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t t = blockIdx.x >> 3; t < slots_per_xcd; t += blocks_per_xcd) {
        const uint64_t slot = tilemap[t];
        if ((uint64_t)threadIdx.x >= (slot >> 48)) continue; // (no block-wide barrier below: every lane owns its LDS column)
        const int64_t i = (int64_t)(slot & 0xffffffffffffULL) + threadIdx.x;
        const SW a = __builtin_nontemporal_load(states + i);
        const uint32_t ig = (uint32_t)i;
        {
            uint32_t A = 0, B = 0;
            SW s = a;
            sA[0] = 0;
            sB[0] = 0;
            for (int u = 0; u < hamming_weight; ++u) { // (the same count for every row of the basis)
                const int p = ctz(s);
                s &= s - (SW)1;
                uint32_t const *c = s_binom + p * kc + u; // C(p, u), C(p, u + 1), C(p, u + 2)
                A += c[1] - c[0];
                B += c[1] - c[2];
                sA[(u + 1) * kBlock] = A;
                sB[(u + 1) * kBlock] = B;
            }
        }
        const X xr = x[i];
        X acc = cx_zero<X>();
        double dsub = 0.0; // sum of vz over this row's anti-aligned pairs
#pragma unroll 4
        for (int p = 0; p < n_pairs; ++p) {
            lsk_pair_row const R = rows[p]; // (wave-uniform: scalar loads)
            const bool up = ((a >> R.i) & (SW)1) != 0; // the particle sits at i
            const bool act = up != (((a >> R.j) & (SW)1) != 0);
            const int k = popc(a & (SW)((((SW)1) << R.i) - (SW)1)), m = popc(a & (SW)R.between);
            const uint32_t e = s_binom[R.j * kc + k + m + 1] - s_binom[R.i * kc + k + 1];
            uint32_t const *arr = up ? sA : sB;
            const int u0 = up ? k + 1 : k;
            const uint32_t passed = arr[(u0 + m) * kBlock] - arr[u0 * kBlock];
            const uint32_t d = (up ? e : 0u - e) - passed;
            const uint32_t idx = act ? ig + d : ig;
            cx_fma(act ? R.v : 0.0, x[idx], acc);
            dsub += act ? R.vz : 0.0;
        }
        cx_fma(pp.dsum - 2.0 * dsub, xr, acc);
        cx_store_nt(y + i, acc);
    }
}
// ---------------------------------------------------------------------------------------------
// ... and walking the PARTICLES of the row instead of the pairs of the operator (k_pairs_site): a pair is active iff exactly one of
// its sites is occupied, i.e. iff it is found from an occupied site p looking at an EMPTY neighbour q -- weight x degree candidates
// per row instead of all pairs (6 x 6 square lattice at weight 6: 24 instead of 72; the kernel is bound by VALU issue, so that is
// what counts).  The neighbours of a site are a table in LDS (lsk_pair_site, D slots per site); with c = particles below q and the
// moving particle t at p:  q > p: it ends as particle c - 1, d = C(q, c) - C(p, t+1) - (A[c] - A[t+1]);
//                           q < p: it ends as particle c,     d = C(q, c+1) - C(p, t+1) - (B[t] - B[c])      (A, B as above).
// ---------------------------------------------------------------------------------------------
template <typename SW, bool CPLX, int D, bool UNIFORM>
__global__ __launch_bounds__(kBlock) void k_pairs_site(lsk_pairplan pp, int hamming_weight, uint64_t const *__restrict__ tilemap,
                                                       int64_t slots_per_xcd, void const *__restrict__ x_v, void *__restrict__ y_v) {
    typedef typename ChainX<CPLX>::type X;
    constexpr int NBITS = 8 * (int)sizeof(SW);
    constexpr int kc = LSK_PAIR_KC;
    constexpr int DW = D / 4; // 32-bit words of neighbours per site
    X const *__restrict__ x = (X const *)x_v;
    X *__restrict__ y = (X *)y_v;
    extern __shared__ uint4 s_dyn4[]; // the neighbour table (lsk.h), then the prefix arrays A, B: [hamming_weight + 1][kBlock] each
    __shared__ uint32_t s_binom[NBITS * kc];
    auto popc = [](SW v) { return sizeof(SW) == 4 ? __popc((uint32_t)v) : __popcll((uint64_t)v); };
    auto ctz = [](SW v) { return sizeof(SW) == 4 ? __builtin_ctz((uint32_t)v) : __builtin_ctzll((uint64_t)v); };
    uint32_t *const s_tab = reinterpret_cast<uint32_t *>(s_dyn4);
    const int tab_words = (pp.site_words + 3) & ~3;
    for (int k = threadIdx.x; k < NBITS * kc; k += kBlock) s_binom[k] = pp.binom[k];
    for (int k = threadIdx.x; k < pp.site_words; k += kBlock) s_tab[k] = pp.sites[k];
    __syncthreads();
    uint32_t const *const s_nb = s_tab, *const s_cls = s_tab + pp.n_sites * DW;
    double2 const *const s_amp = reinterpret_cast<double2 const *>(s_tab + ((2 * pp.n_sites * DW + 3) & ~3));
    uint32_t *const sA = s_tab + tab_words + threadIdx.x;
    uint32_t *const sB = sA + (hamming_weight + 1) * kBlock;
    // one J for all bonds: scalar registers
    double2 const *g_amp = reinterpret_cast<double2 const *>(pp.sites + ((2 * pp.n_sites * DW + 3) & ~3));
    const double v0 = g_amp[0].x, vz0 = g_amp[0].y;
    SW const *__restrict__ states = (SW const *)pp.states;
    const int xcd = blockIdx.x & 7;
    const int64_t blocks_per_xcd = gridDim.x >> 3;
    tilemap += (int64_t)xcd * slots_per_xcd;
    for (int64_t tl = blockIdx.x >> 3; tl < slots_per_xcd; tl += blocks_per_xcd) {
        const uint64_t slot = tilemap[tl];
        if ((uint64_t)threadIdx.x >= (slot >> 48)) continue; // (no block-wide barrier below: every lane owns its LDS column)
        const int64_t i = (int64_t)(slot & 0xffffffffffffULL) + threadIdx.x;
        const SW a = __builtin_nontemporal_load(states + i);
        const uint32_t ig = (uint32_t)i;
        {
            uint32_t A = 0, B = 0;
            SW s = a;
            sA[0] = 0;
            sB[0] = 0;
            for (int u = 0; u < hamming_weight; ++u) {
                const int p = ctz(s);
                s &= s - (SW)1;
                uint32_t const *c = s_binom + p * kc + u;
                A += c[1] - c[0];
                B += c[1] - c[2];
                sA[(u + 1) * kBlock] = A;
                sB[(u + 1) * kBlock] = B;
            }
        }
        const X xr = x[i];
        X acc = cx_zero<X>();
        double dsub = 0.0; // sum of vz over the active pairs (UNIFORM: their number)
        SW s = a;
        // (two particles per trip only with one J for all bonds: the amplitude reads of the general form push the kernel past 80 scalar
        // registers otherwise, where the occupancy API over-reports the resident blocks of a persistent grid -- lsk_dev.hpp)
        constexpr int kParticlesPerTrip = UNIFORM ? 2 : 1;
#pragma unroll kParticlesPerTrip
        for (int t = 0; t < hamming_weight; ++t) {
            const int p = ctz(s);
            s &= s - (SW)1;
            const uint32_t base = ig - s_binom[p * kc + t + 1]; // the rank without the term of the moving particle
            const uint32_t At1 = sA[(t + 1) * kBlock], Bt = sB[t * kBlock];
            uint32_t nbw[DW], clw[DW];
#pragma unroll
            for (int j = 0; j < DW; ++j) {
                nbw[j] = s_nb[p * DW + j];
                clw[j] = UNIFORM ? 0u : s_cls[p * DW + j];
            }
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int q = (int)((nbw[d >> 2] >> (8 * (d & 3))) & 255u);
                const bool act = ((a >> q) & (SW)1) == 0; // the neighbour is empty (a padding slot holds p itself: occupied)
                const bool up = q > p;
                const int c = popc(a & (SW)((((SW)1) << q) - (SW)1)); // particles below q
                const uint32_t cq = s_binom[q * kc + (up ? c : c + 1)];
                const uint32_t pr = (up ? sA : sB)[c * kBlock];
                const uint32_t passed = up ? pr - At1 : Bt - pr;
                const uint32_t idx = act ? base + cq - passed : ig;
                if (UNIFORM) {
                    const double on = act ? 1.0 : 0.0;
                    cx_fma(on, x[idx], acc);
                    dsub += on;
                } else {
                    const double2 amp = s_amp[(clw[d >> 2] >> (8 * (d & 3))) & 255u];
                    cx_fma(act ? amp.x : 0.0, x[idx], acc);
                    dsub += act ? amp.y : 0.0;
                }
            }
        }
        if (UNIFORM) {
            acc = cx_scale(v0, acc);
            dsub *= vz0;
        }
        cx_fma(pp.dsum - 2.0 * dsub, xr, acc);
        cx_store_nt(y + i, acc);
    }
}
template <typename SW, bool CPLX, int D, bool UNIFORM>
static int launch_pairs_site2(lsk_pairplan pp, int hamming_weight, lsk_tilemap tm, void const *x, void *y, void *stream) {
    const size_t lds = sizeof(uint32_t) * (size_t)((pp.site_words + 3) & ~3) + 2 * (size_t)(hamming_weight + 1) * kBlock * sizeof(uint32_t);
    const int64_t gb = tm.slots_per_xcd * 8;
    const int cap = resident_grid(k_pairs_site<SW, CPLX, D, UNIFORM>, gb, lds);
    const unsigned g = (unsigned)(gb < cap ? gb : cap);
    hipLaunchKernelGGL((k_pairs_site<SW, CPLX, D, UNIFORM>), dim3(g), dim3(kBlock), lds, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, x, y);
    LSK_LAUNCH_CHECK();
    return 0;
}
template <typename SW, bool CPLX, int D>
static int launch_pairs_site(lsk_pairplan pp, int hamming_weight, lsk_tilemap tm, void const *x, void *y, void *stream) {
    return pp.n_classes == 1 ? launch_pairs_site2<SW, CPLX, D, true>(pp, hamming_weight, tm, x, y, stream)
                             : launch_pairs_site2<SW, CPLX, D, false>(pp, hamming_weight, tm, x, y, stream);
}
template <typename SW, bool CPLX>
static int launch_pairs_row(lsk_pairplan pp, int hamming_weight, lsk_tilemap tm, void const *x, void *y, void *stream) {
    const size_t lds = 2 * (size_t)(hamming_weight + 1) * kBlock * sizeof(uint32_t);
    const int64_t gb = tm.slots_per_xcd * 8;
    const int cap = resident_grid(k_pairs_row<SW, CPLX>, gb, lds);
    const unsigned g = (unsigned)(gb < cap ? gb : cap);
    hipLaunchKernelGGL((k_pairs_row<SW, CPLX>), dim3(g), dim3(kBlock), lds, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, x, y);
    LSK_LAUNCH_CHECK();
    return 0;
}

extern "C" int lsk_pairs_tile_rows(int cplx) { return cplx ? 512 : 1024; }
extern "C" int lsk_pairs(lsk_pairplan pp, int hamming_weight, int cplx, lsk_tilemap tm, int64_t n, void const *x, void *y, void *stream) {
    if (n == 0 || tm.slots_per_xcd == 0) return 0;
    if (pp.n_near + pp.n_str + pp.n_high > LSK_MAX_PAIRS || hamming_weight + 2 > LSK_PAIR_KC) { snprintf(g_err, sizeof(g_err), "lsk_pairs: plan out of range"); return -1; }
    if (pp.sites) { // one row per lane, walking the particles (tiles of kBlock rows)
        if (pp.n_sites < 1 || pp.n_sites > (pp.wide ? 64 : 32) || (pp.degree != 4 && pp.degree != 8) || pp.n_classes < 1 ||
            pp.n_classes > LSK_PAIR_SITE_MAX_CLASSES) { snprintf(g_err, sizeof(g_err), "lsk_pairs: neighbour table out of range"); return -1; }
#define LSK_SITE(SW, CPLX) (pp.degree == 4 ? launch_pairs_site<SW, CPLX, 4>(pp, hamming_weight, tm, x, y, stream) : launch_pairs_site<SW, CPLX, 8>(pp, hamming_weight, tm, x, y, stream))
        if (pp.wide) return cplx ? LSK_SITE(uint64_t, true) : LSK_SITE(uint64_t, false);
        return cplx ? LSK_SITE(uint32_t, true) : LSK_SITE(uint32_t, false);
#undef LSK_SITE
    }
    if (pp.rows) { // far from half filling: one row per lane (tiles of kBlock rows)
        if (pp.wide) return cplx ? launch_pairs_row<uint64_t, true>(pp, hamming_weight, tm, x, y, stream) : launch_pairs_row<uint64_t, false>(pp, hamming_weight, tm, x, y, stream);
        return cplx ? launch_pairs_row<uint32_t, true>(pp, hamming_weight, tm, x, y, stream) : launch_pairs_row<uint32_t, false>(pp, hamming_weight, tm, x, y, stream);
    }
    const int64_t gb = tm.slots_per_xcd * 8; // one block per tile
    if (pp.wide) { // 33..64 sites: 8-byte states
        if (cplx) hipLaunchKernelGGL((k_pairs_t<uint64_t, true, 512>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
        else hipLaunchKernelGGL((k_pairs_t<uint64_t, false, 1024>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
    } else if (cplx) hipLaunchKernelGGL((k_pairs_t<uint32_t, true, 512>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
    else hipLaunchKernelGGL((k_pairs_t<uint32_t, false, 1024>), dim3((unsigned)gb), dim3(kBlock), 0, (hipStream_t)stream, pp, hamming_weight, tm.entries, tm.slots_per_xcd, n, x, y);
    LSK_LAUNCH_CHECK();
    return 0;
}
// states[i] = (u32)reps[i]: the 4-byte state array the kernel streams
__global__ __launch_bounds__(kBlock) void k_narrow_states(int64_t n, uint64_t const *__restrict__ reps, uint32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) out[i] = (uint32_t)reps[i];
}
extern "C" int lsk_narrow_states(int64_t n, uint64_t const *reps, uint32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_narrow_states, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, out);
    LSK_LAUNCH_CHECK();
    return 0;
}



