// k_pull.hip -- symmetry-projected bases, pull formulation (BatchedOperator.chpl:163-213 + DMV:73-127 as a gather): k_tile_pull (value
// table), the static index table, k_pull_t / k_pull_gather / k_pull_count.  Split out of kernels.hip in round 6; helpers: lsk_dev.hpp.
#include "lsk_dev.hpp"

// ---------------------------------------------------------------------------------------------
// Staged PULL kernel for symmetry-projected bases (Hermitian operators):
//   y[r] = d(r) x[r] + sum_j conj(H~[r'_j, r]) x[r'_j],   H~[r', r] = c conj(chi0) n(r') / n(r)
// Same stage A as k_tile (LDS term list per 256-row tile), stage B projects every packet, looks the
// representative up in the GLOBAL basis, gathers x there and accumulates into a per-tile LDS copy of
// y (ds_add_f64) -- no global atomics, y written once.  With one partition "global" == "local"; with
// one partition per GPU x is the all-gathered vector in global ascending order (replicated-x mode).
// ---------------------------------------------------------------------------------------------
constexpr int kGCPull = 4;
constexpr int kCapPull = kBlock * kGCPull;
// Near window of the pull kernel.  In the sorted array of representatives the partners of a row cluster around the row
// itself: on the symmetric chains 48 % of all projected states |rep(beta)> lie within 512 entries of the tile that
// generated them (measured with the oracle, tests/test_partner_locality.py).  The tile therefore
// stages the representatives [tile - 512, tile + 256 + 512) in LDS as 32-bit offsets from the first of them and
// resolves those partners with a binary search there; their values come from the index-ordered (prescaled) x, whose
// lines are shared by the whole neighbourhood in L1/L2.  Only the others pay the random 16-byte request into the hash
// table, which is what bounds this kernel (one fabric request per probe).
constexpr int kPullHalo = 512;
constexpr int kPullWin = kBlock + 2 * kPullHalo;
constexpr uint32_t kWinAbsent = 0xffffffffu; // offsets >= 2^32 - 1 are treated as "not in the window" (hash path)
// position of offset d in the ascending window w[0, n), or -1.  n <= 2047.
__host__ __device__ __forceinline__ int window_find(uint32_t const *w, int n, uint32_t d) {
    int pos = 0; // lower bound: first entry >= d
#pragma unroll
    for (int step = 1024; step >= 1; step >>= 1)
        if (pos + step <= n && w[pos + step - 1] < d) pos += step;
    return (pos < n && w[pos] == d) ? pos : -1;
}
__host__ __device__ __forceinline__ uint32_t window_offset(uint64_t rep, uint64_t v0) {
    const uint64_t d = rep - v0; // rep >= v0 inside the window (ascending)
    return d >= (uint64_t)kWinAbsent ? kWinAbsent : (uint32_t)d;
}

extern "C" int lsk_test_window_find(uint64_t const *reps, int n, uint64_t key) {
    if (n < 1 || n > kPullWin) return -2;
    uint32_t w[kPullWin];
    for (int i = 0; i < n; ++i) w[i] = window_offset(reps[i], reps[0]);
    if (key < reps[0]) return -1;
    const uint32_t d = window_offset(key, reps[0]);
    return d == kWinAbsent ? -1 : window_find(w, n, d);
}

template <typename W, bool PM1, bool CPLX, bool REAL>
__global__ __launch_bounds__(kBlock) void k_tile_pull(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                      lsk_term const *__restrict__ off, int n_diag,
                                                      lsk_term const *__restrict__ diag, lsk_basis bs,
                                                      lsk_group_elem const *__restrict__ elems, lsk_index ixg,
                                                      int64_t row0, int64_t row1,
                                                      uint64_t const *__restrict__ reps,
                                                      double const *__restrict__ norms_local,
                                                      double const *__restrict__ norms_global,
                                                      int64_t const *__restrict__ row_gidx,
                                                      uint64_t const *__restrict__ tab, int tab_bits,
                                                      uint64_t const *__restrict__ greps, int64_t n_global,
                                                      double const *__restrict__ xs, int halo,
                                                      double const *__restrict__ x, double *__restrict__ y, int *err) {
    constexpr int ES = CPLX ? 4 : 2; // u64 words per hash entry
    __shared__ uint32_t s_win[kPullWin];
    __shared__ uint64_t s_beta[kCapPull];
    constexpr bool RC = REAL && PM1; // conj(H~) stays real: real coefficients and +-1 characters
    __shared__ double s_coef[kCapPull * (RC ? 1 : 2)];
    __shared__ uint16_t s_row[kCapPull];
    __shared__ double s_acc[kBlock * (CPLX ? 2 : 1)];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double inv_na = 0.0;
        if (valid) {
            a = reps[i];
            double na = norms_local[i];
            inv_na = na > 0.0 ? 1.0 / na : 0.0;
        }
        if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0;
        // near window: global indices [gbase, gbase + wn) around the tile (the global index of its first row)
        int64_t gbase = 0;
        int wn = 0;
        uint64_t v0 = 0;
        if (halo > 0) {
            const int64_t ig0 = row_gidx ? row_gidx[t0] : t0;
            gbase = ig0 > halo ? ig0 - halo : 0;
            const int64_t left = n_global - gbase;
            wn = (int)(left < (int64_t)(kBlock + 2 * halo) ? left : (int64_t)(kBlock + 2 * halo));
            v0 = greps[gbase];
            for (int w = tid; w < wn; w += kBlock) s_win[w] = window_offset(greps[gbase + w], v0);
        }
        for (int g0 = 0; g0 < n_groups; g0 += kGCPull) {
            if (tid == 0) s_n = 0;
            __syncthreads();
            const int g1 = min(g0 + kGCPull, n_groups);
            for (int g = g0; g < g1; ++g) {
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                const bool act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                const unsigned long long ball = __ballot(act);
                int base = 0;
                if (lane == 0 && ball) base = atomicAdd(&s_n, __popcll(ball));
                base = __shfl(base, 0);
                if (act) {
                    const int slot = base + __popcll(ball & ((1ULL << lane) - 1));
                    s_beta[slot] = a ^ G.x;
                    s_row[slot] = (uint16_t)tid;
                    // conj(c) / n(alpha)
                    if (RC) s_coef[slot] = cr * inv_na;
                    else { s_coef[2 * slot] = cr * inv_na; s_coef[2 * slot + 1] = -ci * inv_na; }
                }
            }
            __syncthreads();
            const int n = (kAblate && (bs.debug_ablate & 1)) ? 0 : s_n;
            // ---- stage B1: K4 on every packet; representative and conj(H~) go back into the list ---------
            for (int e = tid; e < n; e += kBlock) {
                uint64_t beta = s_beta[e];
                double hr, hi = 0.0; // conj(H~) so far
                if (RC) hr = s_coef[e]; else { hr = s_coef[2 * e]; hi = s_coef[2 * e + 1]; }
                if (kAblate && (bs.debug_ablate & 4)) {
                    beta = a; // a key that exists (this thread's own row)
                } else if (bs.k4_mode != 0) {
                    beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // x is pre-multiplied by norm(rep)
                } else {
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    double n2 = stab * bs.inv_order;
                    if (!(n2 > 1e-12)) { s_row[e] = 0xffff; continue; } // zero-norm orbit: contributes nothing (DMV:110)
                    const double nb = sqrt(n2);
                    beta = (uint64_t)rep;
                    // times chi0 = conj(conj(chi0)) = (chr, -chi), times norm(rep)
                    double tr = (hr * chr + hi * chi) * nb, ti = (hi * chr - hr * chi) * nb;
                    hr = tr; hi = ti;
                    if (RC) s_coef[e] = hr; else { s_coef[2 * e] = hr; s_coef[2 * e + 1] = hi; }
                }
                s_beta[e] = beta;
            }
            // ---- stage B2: gathers.  A thread's packets are independent: issue all home-slot loads first
            // (kGCPull requests in flight per lane), then resolve and accumulate -----------------------------
            if (!(kAblate && (bs.debug_ablate & 2))) {
                const uint64_t hmask = (1ULL << tab_bits) - 1;
                uint64_t key[kGCPull], slot[kGCPull];
                ulonglong2 first[kGCPull];
                double im0[kGCPull];
                bool live[kGCPull];
                int pos[kGCPull];
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) { // the (independent) window searches first: LDS only
                    const int e = tid + k * kBlock;
                    live[k] = e < n && s_row[e] != 0xffff;
                    key[k] = 0; slot[k] = 0; im0[k] = 0.0; pos[k] = -1;
                    first[k] = make_ulonglong2(0, 0);
                    if (live[k]) {
                        key[k] = s_beta[e];
                        if (wn > 0 && key[k] >= v0) {
                            const uint32_t d = window_offset(key[k], v0);
                            if (d != kWinAbsent) pos[k] = window_find(s_win, wn, d);
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) { // then every global load of this thread's packets
                    if (!live[k]) continue;
                    if (pos[k] >= 0) { // near partner: value from the index-ordered vector; looks like a home-slot hit below
                        const int64_t j = gbase + pos[k];
                        first[k].x = key[k];
                        if (CPLX) { first[k].y = (unsigned long long)__double_as_longlong(xs[2 * j]); im0[k] = xs[2 * j + 1]; }
                        else first[k].y = (unsigned long long)__double_as_longlong(xs[j]);
                    } else {
                        slot[k] = hash_slot(key[k], tab_bits);
                        first[k] = *(ulonglong2 const *)(tab + slot[k] * ES);
                        if (CPLX) im0[k] = __longlong_as_double((long long)tab[slot[k] * ES + 2]);
                    }
                }
#pragma unroll
                for (int k = 0; k < kGCPull; ++k) {
                    if (!live[k]) continue;
                    const int e = tid + k * kBlock;
                    double xr, xi = 0.0;
                    if (first[k].x == key[k]) {
                        xr = __longlong_as_double((long long)first[k].y);
                        if (CPLX) xi = im0[k];
                    } else {
                        // collision (load factor <= 0.5: ~1 in 4): continue the probe sequence
                        bool found = false;
                        uint64_t sl = slot[k];
                        uint64_t cur = first[k].x;
                        while (cur != kHashEmpty) {
                            sl = (sl + 1) & hmask;
                            const ulonglong2 en = *(ulonglong2 const *)(tab + sl * ES);
                            cur = en.x;
                            if (cur == key[k]) {
                                xr = __longlong_as_double((long long)en.y);
                                if (CPLX) xi = __longlong_as_double((long long)tab[sl * ES + 2]);
                                found = true;
                                break;
                            }
                        }
                        if (!found) { atomicExch(err, 1); continue; }
                    }
                    double hr, hi = 0.0;
                    if (RC) hr = s_coef[e]; else { hr = s_coef[2 * e]; hi = s_coef[2 * e + 1]; }
                    const int r = s_row[e];
                    if (CPLX) {
                        atomicAdd(&s_acc[2 * r], hr * xr - hi * xi);
                        atomicAdd(&s_acc[2 * r + 1], hr * xi + hi * xr);
                    } else {
                        atomicAdd(&s_acc[r], hr * xr);
                    }
                }
            }
            __syncthreads();
        }
        if (valid) {
            const int64_t ig = row_gidx ? row_gidx[i] : i;
            double dr = 0.0, di = 0.0;
            if (n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
            if (CPLX) {
                const double xr = x[2 * ig], xi = x[2 * ig + 1];
                double yr = dr * xr - di * xi + s_acc[2 * tid], yi = dr * xi + di * xr + s_acc[2 * tid + 1];
                if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; } // accumulate, DMV:1062-1063
                y[2 * i] = yr; y[2 * i + 1] = yi;
            } else {
                double yr = dr * x[ig] + s_acc[tid];
                if (n_diag == 0) yr += y[i];
                y[i] = yr;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(kBlock) void k_hash_insert(int64_t n, uint64_t const *__restrict__ reps, int bits,
                                                        int es, uint64_t *tab, uint32_t *__restrict__ slot_of) {
    const uint64_t mask = (1ULL << bits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const uint64_t key = reps[i];
        uint64_t slot = hash_slot(key, bits);
        for (;;) {
            unsigned long long old = atomicCAS((unsigned long long *)(tab + slot * es), (unsigned long long)kHashEmpty,
                                               (unsigned long long)key);
            if (old == kHashEmpty || old == key) break;
            slot = (slot + 1) & mask;
        }
        slot_of[i] = (uint32_t)slot;
    }
}
__global__ __launch_bounds__(kBlock) void k_hash_clear(int64_t entries, int es, uint64_t *__restrict__ tab) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock)
        tab[i * es] = kHashEmpty;
}
// values: tab[slot_of[i]] <- x[i] * norms[i]   (norms == NULL: unscaled)
// xs (may be NULL): the same scaled values in index order, for the near window of k_tile_pull
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_hash_fill(int64_t n, uint32_t const *__restrict__ slot_of,
                                                      double const *__restrict__ x, double const *__restrict__ norms,
                                                      uint64_t *__restrict__ tab, double *__restrict__ xs) {
    constexpr int ES = CPLX ? 4 : 2;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double nb = norms ? norms[i] : 1.0;
        double *val = (double *)(tab + (size_t)slot_of[i] * ES + 1);
        if (CPLX) {
            const double vr = x[2 * i] * nb, vi = x[2 * i + 1] * nb;
            val[0] = vr; val[1] = vi;
            if (xs) { xs[2 * i] = vr; xs[2 * i + 1] = vi; }
        } else {
            const double v = x[i] * nb;
            val[0] = v;
            if (xs) xs[i] = v;
        }
    }
}
extern "C" int lsk_hash_build(int cplx, int64_t n, uint64_t const *reps, int bits, void *tab, uint32_t *slot_of,
                              void *stream) {
    const int es = cplx ? 4 : 2;
    const int64_t entries = (int64_t)1 << bits;
    hipLaunchKernelGGL(k_hash_clear, dim3(grid_for(entries)), dim3(kBlock), 0, (hipStream_t)stream, entries, es, (uint64_t *)tab);
    LSK_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_hash_insert, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, reps, bits, es, (uint64_t *)tab, slot_of);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int lsk_hash_fill(int cplx, int64_t n, uint32_t const *slot_of, void const *x, double const *norms, void *tab,
                             void *xs, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_hash_fill<true>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, slot_of, (double const *)x, norms, (uint64_t *)tab, (double *)xs);
    else hipLaunchKernelGGL(k_hash_fill<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, slot_of, (double const *)x, norms, (uint64_t *)tab, (double *)xs);
    LSK_LAUNCH_CHECK();
    return 0;
}

extern "C" int lsk_tile_pull(lsk_operator op, lsk_basis bs, lsk_index ix_global, int cplx, int64_t row0, int64_t row1,
                             uint64_t const *reps, double const *norms_local, double const *norms_global,
                             int64_t const *row_gidx, void const *tab, int tab_bits, uint64_t const *reps_global,
                             int64_t n_global, void const *xs_global, int halo, void const *x_global, void *y,
                             int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (bs.proj != LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull is for projected bases"); return -1; }
    if (halo < 0 || halo > kPullHalo || (halo > 0 && (!reps_global || !xs_global || n_global <= 0))) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull: bad near window"); return -1; }
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
#define LSK_TP_ARGS op.runs, op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs, bs.elems, ix_global, row0, row1, reps, \
        norms_local, norms_global, row_gidx, (uint64_t const *)tab, tab_bits, reps_global, n_global, (double const *)xs_global, halo, \
        (double const *)x_global, (double *)y, d_err
#define LSK_TP_LAUNCH(W, PM1)                                                                                   \
    do {                                                                                                        \
        if (cplx) {                                                                                             \
            if (op.is_real) { g.x = tile_grid(k_tile_pull<W, PM1, true, true>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, true, true>), g, b, 0, s, LSK_TP_ARGS); } \
            else { g.x = tile_grid(k_tile_pull<W, PM1, true, false>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, true, false>), g, b, 0, s, LSK_TP_ARGS); } \
        } else {                                                                                                \
            g.x = tile_grid(k_tile_pull<W, PM1, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_pull<W, PM1, false, true>), g, b, 0, s, LSK_TP_ARGS); /* f64: real operators only */ \
        }                                                                                                       \
    } while (0)
    if (bs.number_sites <= 32) { if (bs.chars_pm1) LSK_TP_LAUNCH(uint32_t, true); else LSK_TP_LAUNCH(uint32_t, false); }
    else { if (bs.chars_pm1) LSK_TP_LAUNCH(uint64_t, true); else LSK_TP_LAUNCH(uint64_t, false); }
#undef LSK_TP_LAUNCH
#undef LSK_TP_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Static index table {representative -> 32-bit payload} (lsk_gtab, lsk.h) and the INDEXED mode of the staged pull
// kernel.  The value table above costs one request per far partner but must be rewritten every matvec -- a pass of N
// random 16-byte writes on EVERY rank (chain_40_symm: 43 of 373 ms on one GPU, and undivided by P in the replicated-x
// exchange).  The index table is built once: a far partner costs two dependent requests (bucket, then x[slot]), nothing is
// refreshed, and x is read wherever it already lies -- index order on one device, or the blocks of the replicated-x
// exchange as they arrive from their owners (slot = owner * max_count + local index), which removes the hashed -> block
// permutation pass as well.  Per-rank work then shrinks with P.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_gtab_clear(int64_t entries, uint64_t *__restrict__ tab) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < entries; i += (int64_t)gridDim.x * kBlock) tab[i] = kGtEmpty;
}
__global__ __launch_bounds__(kBlock) void k_gtab_insert(lsk_gtab t, uint64_t *tab, int64_t n, uint64_t const *__restrict__ reps,
                                                        uint32_t const *__restrict__ payload, int *flag) {
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, reps[i], b, tag);
        const uint32_t pay = payload ? payload[i] : (uint32_t)i;
        bool placed = false;
        for (int d = 0; d <= kGtMaxDist && !placed; ++d) {
            const unsigned long long e = ((unsigned long long)gt_hi(tag, d) << 32) | pay;
            for (int sl = 0; sl < 2 && !placed; ++sl)
                placed = atomicCAS((unsigned long long *)(tab + 2 * b + sl), (unsigned long long)kGtEmpty, e) == kGtEmpty;
            b = (b + 1) & bmask;
        }
        if (!placed) atomicExch(flag, 1);
    }
}
// ---- value table (lsk_vtab): the index table with 32-byte buckets {entry0, entry1, x[slot0], x[slot1]} ------------------------------
// Same shape, same placement as the index table it is copied from (bucket b of one = bucket b of the other), so the probe
// sequence is identical; what changes is that the bucket brings the partner's VALUE with it.  Today a far partner of the
// projected pull kernel costs a bucket probe and then a dependent gather of x[slot] -- two random 64-byte requests where the
// fabric serves ~56 G/s (chain_40_symm: 1.0 TB of requests per matvec for 27.6 GB of compulsory bytes).  Here it costs one.
// Price: 32 instead of 16 bytes per bucket (chain_40_symm: 27.6 GB) and a refresh per matvec -- in TABLE order, so the writes
// stream and only the reads of x are random (N requests; the round-2 value table was refreshed in ROW order: N random 16-byte
// WRITES, 43 ms on chain_40_symm).  The static half of a bucket is never rewritten.
__global__ __launch_bounds__(kBlock) void k_vtab_from_gtab(int64_t buckets, uint64_t const *__restrict__ gt, uint64_t *__restrict__ vt) {
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < buckets; b += (int64_t)gridDim.x * kBlock) {
        const ulonglong2 e = *(ulonglong2 const *)(gt + 2 * b);
        *(ulonglong2 *)(vt + 4 * b) = e;
        *(ulonglong2 *)(vt + 4 * b + 2) = make_ulonglong2(0, 0);
    }
}
template <typename X8>
__global__ __launch_bounds__(kBlock) void k_vtab_refresh(int64_t buckets, uint64_t *__restrict__ vt, double const *__restrict__ xsrc) {
    for (int64_t b = (int64_t)blockIdx.x * kBlock + threadIdx.x; b < buckets; b += (int64_t)gridDim.x * kBlock) {
        typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
        const u64x2 raw = __builtin_nontemporal_load((u64x2 const *)(vt + 4 * b));
        const ulonglong2 e = make_ulonglong2(raw.x, raw.y);
        double2 v;
        v.x = e.x != kGtEmpty ? xsrc[(uint32_t)e.x] : 0.0;
        v.y = e.y != kGtEmpty ? xsrc[(uint32_t)e.y] : 0.0;
        __builtin_nontemporal_store(v.x, (double *)(vt + 4 * b + 2));
        __builtin_nontemporal_store(v.y, (double *)(vt + 4 * b + 3));
    }
}
extern "C" int lsk_vtab_build(lsk_gtab t, uint64_t *vt, void *stream) {
    const int64_t nb = (int64_t)1 << t.bbits;
    hipLaunchKernelGGL(k_vtab_from_gtab, dim3(grid_for(nb)), dim3(kBlock), 0, (hipStream_t)stream, nb, t.entries, vt);
    LSK_LAUNCH_CHECK();
    return 0;
}
extern "C" int lsk_vtab_refresh(lsk_gtab t, uint64_t *vt, void const *xsrc, void *stream) {
    const int64_t nb = (int64_t)1 << t.bbits;
    int64_t blocks = (nb + kBlock - 1) / kBlock;
    if (blocks > ((int64_t)1 << 30)) blocks = (int64_t)1 << 30; // a plain grid: one bucket per thread streams best
    hipLaunchKernelGGL(k_vtab_refresh<double>, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, nb, vt, (double const *)xsrc);
    LSK_LAUNCH_CHECK();
    return 0;
}
// value (and slot) of the key with home bucket b / tag: `cur` / `vals` are the home bucket's two halves, already loaded
__device__ __forceinline__ double vt_resolve(lsk_gtab const &t, uint64_t const *__restrict__ vt, uint64_t b, uint32_t tag, ulonglong2 cur,
                                             double2 vals, uint32_t &slot) {
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int d = 0;; ++d) {
        const uint32_t want = gt_hi(tag, d);
        if ((uint32_t)(cur.x >> 32) == want && cur.x != kGtEmpty) { slot = (uint32_t)cur.x; return vals.x; }
        if ((uint32_t)(cur.y >> 32) == want && cur.y != kGtEmpty) { slot = (uint32_t)cur.y; return vals.y; }
        if (cur.x == kGtEmpty || cur.y == kGtEmpty || d == kGtMaxDist) { slot = 0xffffffffu; return 0.0; }
        b = (b + 1) & bmask;
        cur = *(ulonglong2 const *)(vt + 4 * b);
        vals = *(double2 const *)(vt + 4 * b + 2);
    }
}

__global__ __launch_bounds__(kBlock) void k_gtab_lookup(lsk_gtab t, int64_t n, uint64_t const *__restrict__ keys,
                                                        uint32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, keys[i], b, tag);
        out[i] = gt_resolve(t, t.entries, b, tag, *(ulonglong2 const *)(t.entries + 2 * b));
    }
}
extern "C" int lsk_gtab_bits(int L, int64_t n, int64_t max_bytes) {
    if (L < 1 || L > 64 || n < 0) return -1;
    int bb = 2;
    while (((int64_t)2 << bb) < 2 * n) ++bb; // two entries per bucket, load factor <= 0.5
    if (bb < L - 24) bb = L - 24;            // tag (L - bbits bits) + displacement (8) + payload (32) must fit 64 bits
    if (bb > L) bb = L;
    if (bb > 40 || ((int64_t)16 << bb) > max_bytes) return -1;
    return bb;
}
extern "C" int lsk_gtab_build(lsk_gtab t, uint64_t *entries, int64_t n, uint64_t const *reps, uint32_t const *payload,
                              int *d_flag, void *stream) {
    if (t.tbits != t.L - t.bbits || t.tbits < 0 || t.tbits > 24 || n >= 0xffffffffLL) { snprintf(g_err, sizeof(g_err), "lsk_gtab_build: bad table shape"); return -1; }
    const int64_t ne = (int64_t)2 << t.bbits;
    hipLaunchKernelGGL(k_gtab_clear, dim3(grid_for(ne)), dim3(kBlock), 0, (hipStream_t)stream, ne, entries);
    LSK_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_gtab_insert, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, t, entries, n, reps, payload, d_flag);
        LSK_LAUNCH_CHECK();
    }
    return 0;
}
extern "C" int lsk_gtab_lookup(lsk_gtab t, int64_t n, uint64_t const *keys, uint32_t *out, void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_gtab_lookup, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, t, n, keys, out);
    LSK_LAUNCH_CHECK();
    return 0;
}
// host-side sequential build with the placement rule of k_gtab_insert (tests: no device needed); -1 when a key does not fit
extern "C" int lsk_test_gtab_build_host(lsk_gtab t, uint64_t *h, int64_t n, uint64_t const *reps, uint32_t const *payload) {
    const int64_t ne = (int64_t)2 << t.bbits;
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int64_t i = 0; i < ne; ++i) h[i] = kGtEmpty;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t b;
        uint32_t tag;
        gt_split(t, reps[i], b, tag);
        bool placed = false;
        for (int d = 0; d <= kGtMaxDist && !placed; ++d) {
            for (int sl = 0; sl < 2 && !placed; ++sl)
                if (h[2 * b + sl] == kGtEmpty) { h[2 * b + sl] = ((uint64_t)gt_hi(tag, d) << 32) | (payload ? payload[i] : (uint32_t)i); placed = true; }
            b = (b + 1) & bmask;
        }
        if (!placed) return -1;
    }
    return 0;
}
extern "C" int64_t lsk_test_gtab_find(lsk_gtab t, uint64_t const *h, uint64_t key) {
    uint64_t b;
    uint32_t tag;
    gt_split(t, key, b, tag);
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int d = 0; d <= kGtMaxDist; ++d) {
        const uint32_t want = gt_hi(tag, d);
        for (int sl = 0; sl < 2; ++sl) {
            const uint64_t e = h[2 * b + sl];
            if (e == kGtEmpty) return -1;
            if ((uint32_t)(e >> 32) == want) return (int64_t)(uint32_t)e;
        }
        b = (b + 1) & bmask;
    }
    return -1;
}

template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scale(int64_t n, double const *__restrict__ x, double const *__restrict__ norms,
                                                  double *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
        const double nb = norms[i];
        if (CPLX) { out[2 * i] = x[2 * i] * nb; out[2 * i + 1] = x[2 * i + 1] * nb; } else out[i] = x[i] * nb;
    }
}
extern "C" int lsk_scale(int cplx, int64_t n, void const *x, double const *norms, void *out, void *stream) {
    if (n == 0) return 0;
    if (cplx) hipLaunchKernelGGL(k_scale<true>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)x, norms, (double *)out);
    else hipLaunchKernelGGL(k_scale<false>, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, (double const *)x, norms, (double *)out);
    LSK_LAUNCH_CHECK();
    return 0;
}
__global__ __launch_bounds__(kBlock) void k_scatter_owned(int64_t n, uint32_t const *__restrict__ perm, int64_t base, int64_t count,
                                                          uint64_t const *__restrict__ src, uint64_t *__restrict__ dst) {
    for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n; g += (int64_t)gridDim.x * kBlock) {
        const int64_t j = (int64_t)perm[g] - base;
        if (j >= 0 && j < count) dst[j] = src[g];
    }
}
extern "C" int lsk_scatter_owned(int64_t n, uint32_t const *perm, int64_t base, int64_t count, uint64_t const *src, uint64_t *dst,
                                 void *stream) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(k_scatter_owned, dim3(grid_for(n)), dim3(kBlock), 0, (hipStream_t)stream, n, perm, base, count, src, dst);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// INDEXED pull kernels of the projected bases (k_pull_t / k_pull_gather).
//
// One block per 256-row tile, the packet list PER WAVE: each wave owns a 256-slot ring of the LDS list and the rows of its
// 64 lanes.  Stage A appends the packets of three flip-mask groups (<= 192) behind what is left in the ring, stage B takes
// full chunks of 64 packets out of it -- K4 with every lane busy -- and leaves the remainder (< 64) for the next round; the
// tile ends with one partial chunk.  No block barrier inside a tile except around the shared near window, so the four waves
// of a block drift apart and the ALU phase (K4) of one overlaps the look-ups of another.
//
// Stage B per packet: K4 (orbit minimum [+ character, norm]) -> SLOT of the representative:
//   near partners: the tile stages the sorted representatives [tile - halo, tile + 256 + halo) as a two-way hash set in LDS
//     (nw_*, below): one ds_read_b64 instead of the 11-step binary search of round 3; global index -> slot (perm[g] or g);
//   far partners: ONE 16-byte bucket of the static index table (lsk_gtab) -> slot.
// What happens with the slot is the SINK:
//   SINK_FUSED   : value = xsrc[slot], ds_add_f64 into the tile's LDS copy of y, y written once;
//   SINK_VALUE   : (round 6; f64, one partition) as FUSED, but a far partner costs ONE fabric request instead of two dependent
//                  ones: the table bucket is 32 bytes -- the two index entries AND the values x[slot] of both (lsk_vtab, below) --
//                  fetched by two 16-byte loads of one 64-byte line; the values are refreshed once per matvec in TABLE order
//                  (k_vtab_refresh: streaming over the table, one random read of x per representative);
//   SINK_RESOLVE : the slot (and, unless every packet has the same real amplitude, its coefficient) is written to the
//                  per-wave packet stream of lsk_pullbuf and NOTHING of x is read -- this half of the matvec runs while the
//                  blocks of x are still on the wire (ls_amd_repl_matvec, dist.c); k_pull_gather then streams the slots,
//                  gathers x and accumulates.  The stream is recomputed every matvec: the path stays matrix-free.
// ---------------------------------------------------------------------------------------------
constexpr int kWvRing = 256; // slots per wave: < 64 left over + 3 groups x 64 lanes
constexpr int kWvGroups = 3;
enum { K4_TRIVIAL = 0, K4_PM1 = 1, K4_GENERAL = 2 };     // what K4 has to deliver (lsk_basis.k4_mode != 0 -> TRIVIAL)
enum { COEF_UNI = 0, COEF_REAL = 1, COEF_CPLX = 2 };      // per-packet coefficient: none (one real amplitude), f64, 2 x f64
enum { SINK_FUSED = 0, SINK_RESOLVE = 1, SINK_VALUE = 2 };
constexpr uint32_t kNoSlot = 0xffffffffu;

// Near window as a hash set in LDS: kNwSets sets of two 4-byte entries.  h = d * odd constant is a bijection of the 32-bit
// offset d = rep - v0, set = top 10 bits, entry = (low 22 bits of h) << 10 | position in the window (< 1024) -- so set and
// tag together identify d and a match cannot be a false positive.  A set that is already full DROPS the third arrival: the
// window is only an accelerator, whatever it does not answer goes through the static index table (which holds every
// representative).  At <= 768 staged entries ~2 % are dropped.
constexpr int kNwSets = 1024;
constexpr int kNwMaxWin = 1024;
constexpr uint32_t kNwEmpty = 0xffffffffu;
__host__ __device__ __forceinline__ uint32_t nw_mix(uint32_t d) { return d * 0x9E3779B1u; }
__host__ __device__ __forceinline__ uint32_t nw_entry(uint32_t h, int pos) { return (h << 10) | (uint32_t)pos; }
__device__ __forceinline__ void nw_insert(uint32_t *tab, uint32_t d, int pos) {
    const uint32_t h = nw_mix(d), e = nw_entry(h, pos);
    uint32_t *s = tab + 2 * (h >> 22);
    if (atomicCAS(s, kNwEmpty, e) != kNwEmpty) (void)atomicCAS(s + 1, kNwEmpty, e);
}
__host__ __device__ __forceinline__ int nw_match(uint32_t e0, uint32_t e1, uint32_t h) {
    const uint32_t want = h << 10;
    if (((e0 ^ want) >> 10) == 0 && e0 != kNwEmpty) return (int)(e0 & 1023u);
    if (((e1 ^ want) >> 10) == 0 && e1 != kNwEmpty) return (int)(e1 & 1023u);
    return -1;
}
__device__ __forceinline__ int nw_find(uint32_t const *tab, uint32_t d) {
    const uint32_t h = nw_mix(d);
    const uint2 e = *reinterpret_cast<uint2 const *>(tab + 2 * (h >> 22));
    return nw_match(e.x, e.y, h);
}
// host mirror (tests, no device): stage reps[0, n) (ascending, n <= 1024) with the rule of the kernel, sequentially, then look
// `key` up: its position, -1 when the window does not answer (absent, or dropped from a full set), -2 on bad arguments
extern "C" int lsk_test_nw_find(uint64_t const *reps, int n, uint64_t key) {
    if (n < 1 || n > kNwMaxWin) return -2;
    static thread_local uint32_t tab[2 * kNwSets];
    for (int i = 0; i < 2 * kNwSets; ++i) tab[i] = kNwEmpty;
    for (int i = 0; i < n; ++i) {
        const uint32_t d = window_offset(reps[i], reps[0]);
        if (d == kWinAbsent) continue;
        const uint32_t h = nw_mix(d), e = nw_entry(h, i);
        uint32_t *s = tab + 2 * (h >> 22);
        if (s[0] == kNwEmpty) s[0] = e; else if (s[1] == kNwEmpty) s[1] = e;
    }
    if (key < reps[0]) return -1;
    const uint32_t d = window_offset(key, reps[0]);
    if (d == kWinAbsent) return -1;
    const uint32_t h = nw_mix(d);
    return nw_match(tab[2 * (h >> 22)], tab[2 * (h >> 22) + 1], h);
}

// Measured (profiles/r4_pull_xcd_chunk_ab.txt; C = 0 / 16 / 64 / 256 / 1024 / 4096): chain_36_symm cached gather 4.18 / 3.88 / 3.76 /
// 3.67 / 3.65 / 3.88 ms, fused 18.08 / 17.58 / 17.24 / 17.38 / 17.28 / 18.08 ms; chain_40_symm cached 65.7 / 63.6 / 59.9 / 60.1 / 60.0 /
// 61.5 ms, fused 282.5 / 285.0 / 280.9 / 287.3 / 279.5 / 280.1 ms.
extern "C" int64_t lsk_test_pull_tile_of_block(int64_t b, int64_t n_tiles, int C) { return pull_tile_of_block(b, n_tiles, C); }
constexpr int kPullXcdChunk = 256;
static int pull_xcd_chunk() { return kPullXcdChunk; }

template <typename W, int K4M, int COEF, bool CPLX, int SINK>
__global__ __launch_bounds__(kBlock, (COEF == COEF_CPLX ? 4 : 6)) void k_pull_t(lsk_runs runs, int n_groups, lsk_group const *__restrict__ groups,
                                                   lsk_term const *__restrict__ off, int n_diag,
                                                   lsk_term const *__restrict__ diag, lsk_basis bs,
                                                   lsk_group_elem const *__restrict__ elems, int64_t row0, int64_t row1,
                                                   uint64_t const *__restrict__ reps,
                                                   double const *__restrict__ norms_local, lsk_pullidx ix,
                                                   uint64_t const *__restrict__ greps, int64_t n_global,
                                                   double const *__restrict__ xsrc, int halo, double uni_v,
                                                   double *__restrict__ y, lsk_pullbuf buf, int *err, int xcd_chunk) {
    typedef typename ChainX<CPLX>::type X;
    constexpr bool REAL = COEF != COEF_CPLX;
    constexpr bool VALUE = SINK == SINK_VALUE;
    constexpr bool FUSED = SINK == SINK_FUSED || VALUE;
    static_assert(!(VALUE && CPLX), "the value table holds f64 values");
    constexpr int NC = COEF == COEF_UNI ? 0 : (COEF == COEF_REAL ? 1 : 2);
    X const *__restrict__ xv = (X const *)xsrc;
    constexpr int kCap = (kBlock / 64) * kWvRing;
    __shared__ uint32_t s_nw[2 * kNwSets];
    extern __shared__ uint32_t s_nwslot[]; // [kNwMaxWin] when ix.perm != NULL (launch-time size): slot of every window entry
    __shared__ W s_beta[kCap];
    __shared__ double s_coef[NC ? kCap * NC : 1];
    __shared__ uint8_t s_row[kCap]; // row inside the wave (0..63)
    __shared__ double s_acc[FUSED ? kBlock * (CPLX ? 2 : 1) : 1];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kWvRing; // this wave's ring
    uint64_t const *__restrict__ tab = VALUE ? ix.vtab : ix.tab.entries;
    const int64_t n_tiles = (row1 - row0 + kBlock - 1) / kBlock;
    for (int64_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const int64_t t0 = row0 + pull_tile_of_block(tb, n_tiles, gridDim.x >= n_tiles ? xcd_chunk : 0) * kBlock;
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double inv_na = 0.0;
        if (valid) {
            a = reps[i];
            const double na = norms_local[i];
            inv_na = na > 0.0 ? 1.0 / na : 0.0;
        }
        if (FUSED) { if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0; }
        // the diagonal coefficient now, not in the epilogue: the run tables then do not stay in scalar registers across stage B
        double dr = 0.0, di = 0.0;
        if (FUSED && valid && n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
        int64_t gbase = 0;
        int wn = 0;
        uint64_t v0 = 0;
        if (halo > 0) {
            const int64_t ig0 = ix.row_g0 + t0;
            gbase = ig0 > halo ? ig0 - halo : 0;
            const int64_t left = n_global - gbase;
            wn = (int)(left < (int64_t)(kBlock + 2 * halo) ? left : (int64_t)(kBlock + 2 * halo));
            v0 = greps[gbase];
            uint4 *const z = reinterpret_cast<uint4 *>(s_nw);
            for (int w = tid; w < 2 * kNwSets / 4; w += kBlock) z[w] = make_uint4(kNwEmpty, kNwEmpty, kNwEmpty, kNwEmpty);
            __syncthreads();
            for (int w = tid; w < wn; w += kBlock) {
                const uint32_t d = window_offset(greps[gbase + w], v0);
                if (d != kWinAbsent) nw_insert(s_nw, d, w);
                // replicated-x exchange: the slot of a near partner comes out of LDS (a coalesced load per window entry here)
                // instead of one dependent, uncoalesced load of perm[] per near packet
                if (ix.perm) s_nwslot[w] = ix.perm[gbase + w];
            }
        }
        __syncthreads(); // the window is staged
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kWvRing
        // packet stream of this wave's 64 rows (SINK_RESOLVE)
        const int64_t wg = ((t0 - buf.row0) >> 6) + wave;
        const int64_t sbase = buf.offs ? buf.offs[wg] : wg * buf.cap; // exact layout (slot cache) | `cap` packets of room each
        int emitted = 0;
        // K chunks at once: the packets at ring positions head + 64 k + lane (the last chunk holds m <= 64 of them):
        // K4 -> slot [-> value -> ds_add_f64], the loads of the K packets of a lane issued together
        auto chunks = [&](auto KC, int m) {
            constexpr int K = decltype(KC)::value;
            uint64_t beta[K], bkt[K];
            double hr[K], hi[K];
            int r[K], pos[K];
            bool live[K];
            uint32_t tag[K], slot[K];
            ulonglong2 first[K];
            double2 fval[VALUE ? K : 1]; // SINK_VALUE: the value half of the home bucket
#pragma unroll
            for (int k = 0; k < K; ++k) {
                live[k] = k + 1 < K || lane < m;
                const int e = rb + ((head + 64 * k + lane) & (kWvRing - 1));
                beta[k] = live[k] ? (uint64_t)s_beta[e] : 0;
                hr[k] = 1.0; hi[k] = 0.0;
                if (NC == 1) hr[k] = s_coef[e];
                if (NC == 2) { hr[k] = s_coef[2 * e]; hi[k] = s_coef[2 * e + 1]; }
                r[k] = (int)s_row[e];
            }
            // the three x-independent steps of a chunk: K4, near window (LDS), first-level load (perm entry | home bucket)
            auto step_k4 = [&](int k) {
                if (kAblate && (bs.debug_ablate & 4)) return; // profiling builds: no K4 (the look-ups then mostly miss)
                if (K4M == K4_TRIVIAL) {
                    beta[k] = (uint64_t)rep_trivial<W>(bs, elems, (W)beta[k]); // xsrc is pre-multiplied by norm(rep)
                } else if (live[k]) {
                    W rep; double chr, chi, stab;
                    state_info_w<W, K4M == K4_PM1>(bs, elems, (W)beta[k], rep, chr, chi, stab);
                    const double n2 = stab * bs.inv_order;
                    if (!(n2 > 1e-12)) live[k] = false; // zero-norm orbit: contributes nothing (DMV:110)
                    else {
                        const double nb = sqrt(n2);
                        beta[k] = (uint64_t)rep;
                        const double tr = (hr[k] * chr + hi[k] * chi) * nb, ti = (hi[k] * chr - hr[k] * chi) * nb;
                        hr[k] = tr; hi[k] = ti;
                    }
                }
            };
            auto step_window = [&](int k) { // near window: LDS only
                pos[k] = -1; bkt[k] = 0; tag[k] = 0; slot[k] = kNoSlot;
                first[k] = make_ulonglong2(0, 0);
                if (kAblate && (bs.debug_ablate & 2)) { // profiling builds: K4 kept alive, no look-up, no accumulation
                    if (beta[k] == 0x123456789abcdefULL) atomicExch(err, 2);
                    live[k] = false;
                }
                if (live[k]) {
                    if (wn > 0 && beta[k] >= v0 && !(kAblate && (bs.debug_ablate & 32))) {
                        const uint32_t d = window_offset(beta[k], v0);
                        if (d != kWinAbsent) pos[k] = nw_find(s_nw, d);
                    }
                    if (pos[k] < 0) gt_split(ix.tab, beta[k], bkt[k], tag[k]);
                }
            };
            auto step_first = [&](int k) { // first-level loads: perm entry (near) or home bucket (far)
                if (!live[k]) return;
                if (pos[k] >= 0) slot[k] = ix.perm ? s_nwslot[pos[k]] : (uint32_t)(gbase + pos[k]);
                else if constexpr (VALUE) { // both halves of the 32-byte bucket at once: one line, one fabric request
                    first[k] = *(ulonglong2 const *)(tab + 4 * bkt[k]);
                    fval[k] = *(double2 const *)(tab + 4 * bkt[k] + 2);
                } else first[k] = *(ulonglong2 const *)(tab + 2 * bkt[k]);
            };
            // (step by step over the chunks.  Chunk by chunk instead -- the home-bucket load of chunk k in flight while chunk k + 1 runs
            // its K4 -- measured no different: chain_36_symm 17.61 vs 17.72 ms, chain_40_symm 276.5 vs 277.9 ms,
            // profiles/r5_pull_skew_ab.txt: the kernel is at the fabric's random-request rate, not at a latency it could hide)
#pragma unroll
            for (int k = 0; k < K; ++k) step_k4(k);
#pragma unroll
            for (int k = 0; k < K; ++k) step_window(k);
#pragma unroll
            for (int k = 0; k < K; ++k) step_first(k);
            [[maybe_unused]] double far_val[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                far_val[k] = 0.0;
                if (!live[k] || pos[k] >= 0) continue;
                if constexpr (VALUE) far_val[k] = vt_resolve(ix.tab, tab, bkt[k], tag[k], first[k], fval[k], slot[k]);
                else slot[k] = gt_resolve(ix.tab, tab, bkt[k], tag[k], first[k]);
                if (slot[k] == kNoSlot) { atomicExch(err, 1); live[k] = false; }
            }
            if constexpr (FUSED) {
                X val[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if constexpr (VALUE) { // far partners brought their value with the bucket; near ones read x next to the tile
                        val[k] = !live[k] ? 0.0 : (pos[k] >= 0 ? xv[slot[k]] : far_val[k]);
                    } else val[k] = (live[k] && !(kAblate && (bs.debug_ablate & 64))) ? xv[slot[k]] : cx_zero<X>();
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (!live[k]) continue;
                    const int ra = (wave << 6) + r[k];
                    if constexpr (CPLX) {
                        atomicAdd(&s_acc[2 * ra], hr[k] * val[k].x - hi[k] * val[k].y);
                        atomicAdd(&s_acc[2 * ra + 1], hr[k] * val[k].y + hi[k] * val[k].x);
                    } else if constexpr (NC == 0) {
                        atomicAdd(&s_acc[ra], val[k]);
                    } else {
                        atomicAdd(&s_acc[ra], hr[k] * val[k]);
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (k + 1 == K && lane >= m) continue;
                    const int64_t o = sbase + emitted + 64 * k + lane;
                    __builtin_nontemporal_store(live[k] ? slot[k] : kNoSlot, buf.slots + o);
                    __builtin_nontemporal_store((uint8_t)r[k], buf.rows + o);
                    if (NC == 1) __builtin_nontemporal_store(hr[k], buf.coefs + o);
                    if (NC == 2) { __builtin_nontemporal_store(hr[k], buf.coefs + 2 * o); __builtin_nontemporal_store(hi[k], buf.coefs + 2 * o + 1); }
                }
                emitted += 64 * (K - 1) + m;
            }
        };
        const W tdiff = (W)a ^ (W)((W)a >> 1); // bit b set: sites b, b + 1 differ (adjacent exchange groups)
        for (int g0 = 0; g0 < n_groups; g0 += kWvGroups) {
            const int g1 = min(g0 + kWvGroups, n_groups);
            for (int g = g0; g < g1; ++g) { // stage A: append
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                bool act;
                if (NC == 0) { // every group is an exchange pair with the amplitude uni_v
                    act = valid && (G.adj >= 0 ? (bool)((tdiff >> G.adj) & 1) : WordTraits<W>::popc((W)a & (W)G.x) == 1);
                } else {
                    if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                    act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                }
                const unsigned long long ball = __ballot(act);
                if (act) {
                    const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kWvRing - 1));
                    s_beta[slot] = (W)(a ^ G.x);
                    s_row[slot] = (uint8_t)lane;
                    if (NC == 1) s_coef[slot] = cr * inv_na;
                    if (NC == 2) { s_coef[2 * slot] = cr * inv_na; s_coef[2 * slot + 1] = -ci * inv_na; }
                }
                cnt += __popcll(ball);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (kAblate && (bs.debug_ablate & 1)) { head = (head + cnt) & (kWvRing - 1); cnt = 0; } // profiling builds: stage A only
            // stage B on full chunks: two at a time while the ring has them (trivial sectors; the element loops of the other
            // sectors are long enough by themselves, and two chunks of their state do not fit the scalar registers)
            constexpr int KMAX = K4M == K4_TRIVIAL ? 2 : 1;
            while (cnt >= 64 * KMAX) {
                chunks(std::integral_constant<int, KMAX>(), 64);
                head = (head + 64 * KMAX) & (kWvRing - 1);
                cnt -= 64 * KMAX;
            }
            if constexpr (KMAX == 2)
                if (cnt >= 64) {
                    chunks(std::integral_constant<int, 1>(), 64);
                    head = (head + 64) & (kWvRing - 1);
                    cnt -= 64;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (cnt > 0) chunks(std::integral_constant<int, 1>(), cnt);
        if constexpr (FUSED) {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const int64_t ig = ix.row_g0 + i;
                const uint32_t own = ix.perm ? ix.perm[ig] : (uint32_t)ig;
                const double back = K4M == K4_TRIVIAL ? inv_na : 1.0; // xsrc holds x * norm(rep) in the prescaling K4 modes
                const double sc = NC == 0 ? uni_v * inv_na : 1.0;     // one amplitude for every packet: applied once per row
                if constexpr (CPLX) {
                    const X xo = xv[own];
                    const double xr = xo.x * back, xi = xo.y * back;
                    double yr = dr * xr - di * xi + sc * s_acc[2 * tid], yi = dr * xi + di * xr + sc * s_acc[2 * tid + 1];
                    if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; } // accumulate, DMV:1062-1063
                    y[2 * i] = yr; y[2 * i + 1] = yi;
                } else {
                    double yr = n_diag > 0 ? dr * (xv[own] * back) + sc * s_acc[tid] : sc * s_acc[tid];
                    if (n_diag == 0) yr += y[i];
                    y[i] = yr;
                }
            }
        } else if (lane == 0) buf.counts[wg] = (uint32_t)emitted;
        __syncthreads(); // every wave is done with the window
    }
}

// Second half of the split matvec: the packet stream of k_pull_t<..., SINK_RESOLVE> -> x[slot] -> y.  One wave per 64 rows,
// no barrier at all: a wave only touches the LDS accumulators of its own rows.  GU chunks of 64 packets in flight per wave.
template <int COEF, bool CPLX>
__global__ __launch_bounds__(kBlock) void k_pull_gather(lsk_runs runs, int n_diag, lsk_term const *__restrict__ diag, int k4_mode,
                                                        int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                        double const *__restrict__ norms_local, lsk_pullidx ix,
                                                        double const *__restrict__ xsrc, double uni_v, double *__restrict__ y,
                                                        lsk_pullbuf buf, int xcd_chunk) {
    typedef typename ChainX<CPLX>::type X;
    constexpr bool REAL = COEF != COEF_CPLX;
    constexpr int NC = COEF == COEF_UNI ? 0 : (COEF == COEF_REAL ? 1 : 2);
    constexpr int GU = 4;
    X const *__restrict__ xv = (X const *)xsrc;
    __shared__ double s_acc[kBlock * (CPLX ? 2 : 1)];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int64_t n_tiles = (row1 - row0 + kBlock - 1) / kBlock;
    for (int64_t tb = blockIdx.x; tb < n_tiles; tb += gridDim.x) {
        const int64_t t0 = row0 + pull_tile_of_block(tb, n_tiles, gridDim.x >= n_tiles ? xcd_chunk : 0) * kBlock;
        if ((t0 + (wave << 6)) >= row1) continue; // wave-uniform
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        if (CPLX) { s_acc[2 * tid] = 0.0; s_acc[2 * tid + 1] = 0.0; } else s_acc[tid] = 0.0;
        const int64_t wg = ((t0 - buf.row0) >> 6) + wave;
        const int64_t sbase = buf.offs ? buf.offs[wg] : wg * buf.cap;
        const int n = (int)buf.counts[wg];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int c = 0; c < n; c += 64 * GU) {
            uint32_t slot[GU];
            int r[GU];
            double hr[GU], hi[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                const int p = c + 64 * u + lane;
                slot[u] = kNoSlot; r[u] = 0; hr[u] = 1.0; hi[u] = 0.0;
                if (p < n) {
                    const int64_t o = sbase + p;
                    slot[u] = __builtin_nontemporal_load(buf.slots + o);
                    r[u] = (int)__builtin_nontemporal_load(buf.rows + o);
                    if (NC == 1) hr[u] = __builtin_nontemporal_load(buf.coefs + o);
                    if (NC == 2) { hr[u] = __builtin_nontemporal_load(buf.coefs + 2 * o); hi[u] = __builtin_nontemporal_load(buf.coefs + 2 * o + 1); }
                }
            }
            X val[GU];
#pragma unroll
            for (int u = 0; u < GU; ++u) val[u] = slot[u] != kNoSlot ? xv[slot[u]] : cx_zero<X>();
#pragma unroll
            for (int u = 0; u < GU; ++u) {
                if (slot[u] == kNoSlot) continue;
                const int ra = (wave << 6) + r[u];
                if constexpr (CPLX) {
                    atomicAdd(&s_acc[2 * ra], hr[u] * val[u].x - hi[u] * val[u].y);
                    atomicAdd(&s_acc[2 * ra + 1], hr[u] * val[u].y + hi[u] * val[u].x);
                } else if constexpr (NC == 0) {
                    atomicAdd(&s_acc[ra], val[u]);
                } else {
                    atomicAdd(&s_acc[ra], hr[u] * val[u]);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (valid) {
            const uint64_t a = reps[i];
            const double na = norms_local[i];
            const double inv_na = na > 0.0 ? 1.0 / na : 0.0;
            const int64_t ig = ix.row_g0 + i;
            const uint32_t own = ix.perm ? ix.perm[ig] : (uint32_t)ig;
            const double back = k4_mode != 0 ? inv_na : 1.0;
            const double sc = NC == 0 ? uni_v * inv_na : 1.0;
            double dr = 0.0, di = 0.0;
            if (n_diag > 0) diag_coeff<uint64_t, REAL>(runs, n_diag, diag, a, dr, di);
            if constexpr (CPLX) {
                const X xo = xv[own];
                const double xr = xo.x * back, xi = xo.y * back;
                double yr = dr * xr - di * xi + sc * s_acc[2 * tid], yi = dr * xi + di * xr + sc * s_acc[2 * tid + 1];
                if (n_diag == 0) { yr += y[2 * i]; yi += y[2 * i + 1]; }
                y[2 * i] = yr; y[2 * i + 1] = yi;
            } else {
                double yr = n_diag > 0 ? dr * (xv[own] * back) + sc * s_acc[tid] : sc * s_acc[tid];
                if (n_diag == 0) yr += y[i];
                y[i] = yr;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // this wave's accumulators are free again
        __builtin_amdgcn_wave_barrier();
    }
}

// kind of K4 work and of per-packet coefficient for (operator, basis) -- one decision for the fused, the resolve and the
// gather kernel
static void pull_kinds(lsk_operator const &op, lsk_basis const &bs, int &k4m, int &coef) {
    if (bs.k4_mode != 0) { k4m = K4_TRIVIAL; coef = !op.is_real ? COEF_CPLX : (op.uni ? COEF_UNI : COEF_REAL); }
    else if (bs.chars_pm1) { k4m = K4_PM1; coef = op.is_real ? COEF_REAL : COEF_CPLX; }
    else { k4m = K4_GENERAL; coef = COEF_CPLX; }
}
extern "C" int64_t lsk_pullbuf_cap(lsk_operator op) { return (int64_t)64 * (op.n_groups > 0 ? op.n_groups : 1); }
extern "C" int lsk_pullbuf_coef_doubles(lsk_operator op, lsk_basis bs) {
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    return coef == COEF_UNI ? 0 : (coef == COEF_REAL ? 1 : 2);
}

template <typename W, int K4M, int COEF, bool CPLX, int SINK>
static void launch_pull_t(lsk_operator const &op, lsk_basis const &bs, int64_t row0, int64_t row1, uint64_t const *reps,
                          double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global, void const *xsrc,
                          int halo, void *y, lsk_pullbuf buf, int *d_err, hipStream_t s) {
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    dim3 g((unsigned)tile_grid(k_pull_t<W, K4M, COEF, CPLX, SINK>, work_blocks)), b(kBlock);
    const size_t dyn = ix.perm ? sizeof(uint32_t) * kNwMaxWin : 0;
    hipLaunchKernelGGL((k_pull_t<W, K4M, COEF, CPLX, SINK>), g, b, dyn, s, op.runs, op.n_groups, op.groups, op.off, op.n_diag, op.diag, bs,
                       bs.elems, row0, row1, reps, norms_local, ix, reps_global, n_global, (double const *)xsrc, halo, op.uni_v,
                       (double *)y, buf, d_err, pull_xcd_chunk());
}
template <typename W, bool CPLX, int SINK>
static int dispatch_pull_t(lsk_operator const &op, lsk_basis const &bs, int64_t row0, int64_t row1, uint64_t const *reps,
                           double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                           void const *xsrc, int halo, void *y, lsk_pullbuf buf, int *d_err, hipStream_t s) {
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
#define LSK_PT(K4M, COEF) launch_pull_t<W, K4M, COEF, CPLX, SINK>(op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, xsrc, halo, y, buf, d_err, s)
    if (coef == COEF_CPLX) {
        if constexpr (!CPLX && (SINK == SINK_FUSED || SINK == SINK_VALUE)) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull: complex coefficients need c128 vectors"); return -1; }
        else { if (k4m == K4_TRIVIAL) LSK_PT(K4_TRIVIAL, COEF_CPLX); else if (k4m == K4_PM1) LSK_PT(K4_PM1, COEF_CPLX); else LSK_PT(K4_GENERAL, COEF_CPLX); }
    } else if (coef == COEF_REAL) { if (k4m == K4_TRIVIAL) LSK_PT(K4_TRIVIAL, COEF_REAL); else LSK_PT(K4_PM1, COEF_REAL); }
    else LSK_PT(K4_TRIVIAL, COEF_UNI);
#undef LSK_PT
    return 0;
}
static int pull_args_ok(lsk_basis const &bs, int halo, uint64_t const *reps_global, int64_t n_global, char const *who) {
    if (bs.proj != LSK_PROJ_FULL) { snprintf(g_err, sizeof(g_err), "%s is for projected bases", who); return -1; }
    if (halo < 0 || kBlock + 2 * halo > kNwMaxWin || (halo > 0 && (!reps_global || n_global <= 0))) { snprintf(g_err, sizeof(g_err), "%s: bad near window", who); return -1; }
    return 0;
}
extern "C" int lsk_pull_max_halo(void) { return (kNwMaxWin - kBlock) / 2; }

extern "C" int lsk_tile_pull_idx(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                                 double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                                 void const *xsrc, int halo, void *y, int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (pull_args_ok(bs, halo, reps_global, n_global, "lsk_tile_pull_idx") != 0) return -1;
    lsk_pullbuf none;
    memset(&none, 0, sizeof(none));
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define LSK_PA op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, xsrc, halo, y, none, d_err, s
    if (ix.vtab) { // value table (f64, one partition): one fabric request per far partner
        if (cplx || ix.perm) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull_idx: the value table serves f64 vectors of one partition"); return -1; }
        rc = bs.number_sites <= 32 ? dispatch_pull_t<uint32_t, false, SINK_VALUE>(LSK_PA) : dispatch_pull_t<uint64_t, false, SINK_VALUE>(LSK_PA);
    } else if (bs.number_sites <= 32) rc = cplx ? dispatch_pull_t<uint32_t, true, SINK_FUSED>(LSK_PA) : dispatch_pull_t<uint32_t, false, SINK_FUSED>(LSK_PA);
    else rc = cplx ? dispatch_pull_t<uint64_t, true, SINK_FUSED>(LSK_PA) : dispatch_pull_t<uint64_t, false, SINK_FUSED>(LSK_PA);
#undef LSK_PA
    if (rc != 0) return -1;
    LSK_LAUNCH_CHECK();
    return 0;
}

// Packets that stage A of k_pull_t generates for every 64 rows (the streams' exact lengths): out[w] for the rows
// [row0 + 64 w, row0 + 64 w + 64).  Same activity test as stage A -- dead packets (zero-norm orbits) keep their place in a
// stream, so this is what the resolve kernel emits.
template <int COEF>
__global__ __launch_bounds__(kBlock) void k_pull_count(int n_groups, lsk_group const *__restrict__ groups, lsk_term const *__restrict__ off,
                                                       int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                       int64_t *__restrict__ out) {
    constexpr bool REAL = COEF != COEF_CPLX;
    const int lane = threadIdx.x & 63;
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t w0 = t0 + (threadIdx.x & ~63u);
        if (w0 >= row1) continue;
        const int64_t i = t0 + threadIdx.x;
        const bool valid = i < row1;
        const uint64_t a = valid ? reps[i] : 0;
        const uint64_t tdiff = a ^ (a >> 1);
        int cnt = 0;
        for (int g = 0; g < n_groups; ++g) {
            lsk_group const G = groups[g];
            bool act;
            if (COEF == COEF_UNI) act = valid && (G.adj >= 0 ? (bool)((tdiff >> G.adj) & 1) : __popcll(a & G.x) == 1);
            else {
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
            }
            cnt += __popcll(__ballot(act));
        }
        if (lane == 0) out[(w0 - row0) >> 6] = cnt;
    }
}
// out[0, streams] <- exclusive offsets of the packet streams of rows [row0, row1) (streams = ceil(rows / 64); out[streams] =
// total); `out` has ((streams + 3) & ~3) + 1 entries and the entries behind out[streams] repeat the total: the last 256-row
// tile of the resolve / gather kernels runs four waves whatever the row count, and each reads its offset.  Synchronises the stream
extern "C" int lsk_tile_pull_stream_offsets(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps,
                                            int64_t *out, void *stream) {
    if (row1 <= row0) return 0;
    const int64_t streams = (row1 - row0 + 63) / 64;
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    hipStream_t s = (hipStream_t)stream;
    const int64_t padded = (streams + 3) & ~(int64_t)3;
    LSK_CHECK(hipMemsetAsync(out, 0, 8 * (size_t)(padded + 1), s));
    const dim3 g((unsigned)grid_for(row1 - row0)), b(kBlock);
    if (coef == COEF_UNI) hipLaunchKernelGGL(k_pull_count<COEF_UNI>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    else if (coef == COEF_REAL) hipLaunchKernelGGL(k_pull_count<COEF_REAL>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    else hipLaunchKernelGGL(k_pull_count<COEF_CPLX>, g, b, 0, s, op.n_groups, op.groups, op.off, row0, row1, reps, out);
    LSK_LAUNCH_CHECK();
    return lsk_internal_exclusive_scan_i64(padded + 1, out, out, s);
}

// first half of the split matvec: rows [row0, row1) -> packet stream in `buf` (buf.row0 = the row that owns stream 0; a
// multiple of 64 rows below row0).  Reads neither x nor y.
extern "C" int lsk_tile_pull_resolve(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps,
                                     double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                                     int halo, lsk_pullbuf buf, int *d_err, void *stream) {
    if (row1 <= row0) return 0;
    if (pull_args_ok(bs, halo, reps_global, n_global, "lsk_tile_pull_resolve") != 0) return -1;
    if (!buf.slots || !buf.rows || !buf.counts || (!buf.offs && buf.cap < lsk_pullbuf_cap(op)) || ((row0 - buf.row0) & 255) != 0 || row0 < buf.row0 ||
        (lsk_pullbuf_coef_doubles(op, bs) > 0 && !buf.coefs)) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull_resolve: bad packet buffer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
#define LSK_PA op, bs, row0, row1, reps, norms_local, ix, reps_global, n_global, nullptr, halo, nullptr, buf, d_err, s
    if (bs.number_sites <= 32) rc = dispatch_pull_t<uint32_t, false, SINK_RESOLVE>(LSK_PA);
    else rc = dispatch_pull_t<uint64_t, false, SINK_RESOLVE>(LSK_PA);
#undef LSK_PA
    if (rc != 0) return -1;
    LSK_LAUNCH_CHECK();
    return 0;
}
// second half: y[row0, row1) from the packet stream and xsrc
extern "C" int lsk_tile_pull_gather(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                                    double const *norms_local, lsk_pullidx ix, void const *xsrc, lsk_pullbuf buf, void *y,
                                    void *stream) {
    if (row1 <= row0) return 0;
    int k4m, coef;
    pull_kinds(op, bs, k4m, coef);
    if (coef == COEF_CPLX && !cplx) { snprintf(g_err, sizeof(g_err), "lsk_tile_pull_gather: complex coefficients need c128 vectors"); return -1; }
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
    dim3 g(1), b(kBlock);
#define LSK_PG(COEF, CPLX)                                                                                              \
    do {                                                                                                                \
        g.x = (unsigned)tile_grid(k_pull_gather<COEF, CPLX>, work_blocks);                                              \
        hipLaunchKernelGGL((k_pull_gather<COEF, CPLX>), g, b, 0, s, op.runs, op.n_diag, op.diag, bs.k4_mode, row0, row1, reps, \
                           norms_local, ix, (double const *)xsrc, op.uni_v, (double *)y, buf, pull_xcd_chunk());        \
    } while (0)
    if (coef == COEF_CPLX) LSK_PG(COEF_CPLX, true);
    else if (coef == COEF_REAL) { if (cplx) LSK_PG(COEF_REAL, true); else LSK_PG(COEF_REAL, false); }
    else { if (cplx) LSK_PG(COEF_UNI, true); else LSK_PG(COEF_UNI, false); }
#undef LSK_PG
    LSK_LAUNCH_CHECK();
    return 0;
}



