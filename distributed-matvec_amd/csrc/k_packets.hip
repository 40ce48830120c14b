// k_packets.hip -- the packet path (the reference's formulation, DMV:265-311, 663-853): producers k_tile / k_tile_wv / k_tile_st,
// consumers k_scatter* / k_window.  Split out of kernels.hip in round 6; shared device helpers: lsk_dev.hpp.
#include "lsk_dev.hpp"

// ---------------------------------------------------------------------------------------------
// Staged ("tile") kernel: symmetry projection and/or hash-partitioned output.
// A 256-row tile expands kGC flip-mask groups at a time into an LDS term list (stage A, K2), the
// list is then processed densely, one packet per lane (stage B): K3/K4 projection, K5 owner hash,
// and either K7+K8 (own partition) or a rank inside the (tile, destination) bucket.  Buckets are
// reserved in the send buffer with ONE global atomic per (tile-chunk, destination) and written out
// from LDS (K6: the radix partition by destination happens here, in LDS).
// ---------------------------------------------------------------------------------------------

constexpr uint32_t kDead = 0xffffffffu;

// GC = flip-mask groups expanded per LDS list: 8 for cheap packets (fewer barriers: chain_28, P = 8: 11.0 vs 14.0 ms with 4), 4 for
// symmetry-projected bases (20 instead of 40 KB of LDS per block: twice the blocks per CU to hide K4 and the index look-ups:
// chain_36_symm push 46.1 -> 31.8 ms)
template <typename W, bool PM1, bool CPLX, bool REAL, int GC>
__global__ __launch_bounds__(kBlock) void k_tile(int n_groups, lsk_group const *__restrict__ groups,
                                                 lsk_term const *__restrict__ off, lsk_basis bs,
                                                 lsk_group_elem const *__restrict__ elems, lsk_index ix,
                                                 int count_only, Owner owner, int me, int64_t row0, int64_t row1,
                                                 uint64_t const *__restrict__ reps,
                                                 double const *__restrict__ norms,
                                                 double const *__restrict__ x, double *y,
                                                 unsigned long long *cursors,
                                                 lsk_round_layout const *__restrict__ layout, char *send,
                                                 unsigned long long *counts, int *err) {
    constexpr int kCap = kBlock * GC;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[kCap * (CPLX ? 2 : 1)];
    __shared__ uint32_t s_meta[kCap];
    __shared__ unsigned s_cnt[LSK_MAX_PARTS];
    __shared__ unsigned long long s_base[LSK_MAX_PARTS];
    __shared__ int s_n;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int P = (int)owner.P;

    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double xr = 0.0, xi = 0.0;
        if (valid) {
            a = reps[i];
            if (count_only) xr = 1.0; // the packet set must not depend on x (exact send counts)
            else if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; } else xr = x[i];
            if (!count_only && bs.proj == LSK_PROJ_FULL) { // fold 1 / norm(alpha) into x  (BatchedOperator.chpl:198-202)
                double na = norms[i];
                double s = na > 0.0 ? 1.0 / na : 0.0;
                xr *= s;
                xi *= s;
            }
        }
        for (int g0 = 0; g0 < n_groups; g0 += GC) {
            if (tid == 0) s_n = 0;
            for (int d = tid; d < P; d += kBlock) s_cnt[d] = 0;
            __syncthreads();
            // ---- stage A: expand terms of kGC groups into the LDS list --------------------------
            const int g1 = min(g0 + GC, n_groups);
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
                    if (CPLX) {
                        s_val[2 * slot] = cr * xr - ci * xi;
                        s_val[2 * slot + 1] = cr * xi + ci * xr;
                    } else s_val[slot] = cr * xr;
                }
            }
            __syncthreads();
            const int n = (kAblate && (bs.debug_ablate & 1)) ? 0 : s_n; // LS_AMD_ABLATE (profiling only): 1 no stage B, 8 drop own packets, 16 no packet writes
            // ---- stage B: project, hash, scatter locally or rank into a destination bucket --------
            for (int e = tid; e < n; e += kBlock) {
                uint64_t beta = s_beta[e];
                double vr, vi = 0.0;
                if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e];
                bool dead = false;
                if (bs.proj == LSK_PROJ_INVERSION) {
                    uint64_t f = beta ^ bs.site_mask;
                    if (f < beta) { beta = f; vr *= (double)bs.spin_inversion; vi *= (double)bs.spin_inversion; }
                } else if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) {
                    beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // norm(rep) applied at index time
                } else if (bs.proj == LSK_PROJ_FULL) {
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    double n2 = stab * bs.inv_order;
                    if (n2 > 1e-12) {
                        double nb = sqrt(n2);
                        beta = (uint64_t)rep;
                        if (CPLX) {
                            double tr = (vr * chr - vi * chi) * nb, ti = (vr * chi + vi * chr) * nb;
                            vr = tr; vi = ti;
                        } else vr = vr * chr * nb;
                    } else dead = true; // zero-norm orbit: c == 0 => skipped (DMV:110)
                }
                uint32_t meta = kDead;
                if (!dead) {
                    const int dest = owner_of(beta, owner);
                    if (count_only) {
                        atomicAdd(&s_cnt[dest], 1u);
                    } else if (dest == me) {
                        if (kAblate && (bs.debug_ablate & 8)) { s_meta[e] = kDead; continue; }
                        int64_t idx = search_index(ix, beta);
                        if (idx < 0) atomicExch(err, 1);
                        else {
                            if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) { double nb = norms[idx]; vr *= nb; vi *= nb; }
                            if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                            else atomic_add_f64(y + idx, vr);
                        }
                    } else {
                        unsigned rank = atomicAdd(&s_cnt[dest], 1u);
                        meta = ((uint32_t)dest << 16) | rank;
                        s_beta[e] = beta;
                        if (CPLX) { s_val[2 * e] = vr; s_val[2 * e + 1] = vi; } else s_val[e] = vr;
                    }
                }
                s_meta[e] = meta;
            }
            __syncthreads();
            if (count_only) {
                for (int d = tid; d < P; d += kBlock)
                    if (s_cnt[d]) atomicAdd(&counts[d], (unsigned long long)s_cnt[d]);
            } else if (P > 1) {
                for (int d = tid; d < P; d += kBlock)
                    if (s_cnt[d]) s_base[d] = atomicAdd(&cursors[d], (unsigned long long)s_cnt[d]);
                __syncthreads();
                for (int e = tid; e < n; e += kBlock) {
                    const uint32_t meta = s_meta[e];
                    if (meta == kDead || (kAblate && (bs.debug_ablate & 16))) continue;
                    const int dest = (int)(meta >> 16);
                    const unsigned long long pos = s_base[dest] + (meta & 0xffffu);
                    uint64_t *ob = (uint64_t *)(send + layout->beta_off[dest]);
                    double *ov = (double *)(send + layout->val_off[dest]);
                    ob[pos] = s_beta[e];
                    if (CPLX) { ov[2 * pos] = s_val[2 * e]; ov[2 * pos + 1] = s_val[2 * e + 1]; }
                    else ov[pos] = s_val[e];
                }
            }
            __syncthreads();
        }
    }
}

extern "C" int lsk_tile(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int count_only, int P, int me,
                        int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x,
                        void *y, unsigned long long *d_cursors, lsk_round_layout const *d_layout, void *d_send,
                        unsigned long long *d_counts, int *d_err, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (P > LSK_MAX_PARTS || P < 1) { snprintf(g_err, sizeof(g_err), "lsk_tile: bad partition count %d", P); return -1; }
    if (!count_only && ix.kind != LSK_INDEX_SEARCH) { snprintf(g_err, sizeof(g_err), "lsk_tile needs a SEARCH index"); return -1; }
    Owner ow = make_owner(P);
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
#define LSK_TILE_ARGS op.n_groups, op.groups, op.off, bs, bs.elems, ix, count_only, ow, me, row0, row1, reps, norms, \
        (double const *)x, (double *)y, d_cursors, d_layout, (char *)d_send, d_counts, d_err
#define LSK_TILE_LAUNCH(W, PM1, GC)                                                                            \
    do {                                                                                                   \
        if (cplx) {                                                                                        \
            if (op.is_real) { g.x = tile_grid(k_tile<W, PM1, true, true, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, true, true, GC>), g, b, 0, s, LSK_TILE_ARGS); } \
            else { g.x = tile_grid(k_tile<W, PM1, true, false, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, true, false, GC>), g, b, 0, s, LSK_TILE_ARGS); } \
        } else { /* f64 vectors: real operators only (the plan refuses the rest) */                        \
            g.x = tile_grid(k_tile<W, PM1, false, true, GC>, work_blocks); hipLaunchKernelGGL((k_tile<W, PM1, false, true, GC>), g, b, 0, s, LSK_TILE_ARGS); \
        }                                                                                                  \
    } while (0)
    const bool narrow = bs.number_sites <= 32 && bs.proj == LSK_PROJ_FULL;
    if (narrow) { if (bs.chars_pm1) LSK_TILE_LAUNCH(uint32_t, true, 4); else LSK_TILE_LAUNCH(uint32_t, false, 4); }
    else if (bs.proj == LSK_PROJ_FULL) { if (bs.chars_pm1) LSK_TILE_LAUNCH(uint64_t, true, 4); else LSK_TILE_LAUNCH(uint64_t, false, 4); }
    else LSK_TILE_LAUNCH(uint64_t, true, 8); // no projection: PM1 is irrelevant
#undef LSK_TILE_LAUNCH
#undef LSK_TILE_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Packet producer with per-WAVE packet rings and a DETERMINISTIC send layout (k_tile_wv; P <= 64).
//
// k_tile above synchronises its four waves three times per list (stage A | stage B | bucket reservation | write-out), ranks
// the packets of a list with 64-lane LDS atomics on P addresses and reserves the buckets with global atomics on P cursors:
// 75 % of its wave cycles wait (profiles/r2_packets_chain28_P8_sq_counters.txt).  Here a wave owns its 64 rows, a 256-slot
// ring of the LDS list and -- in lane d -- the cursor of destination d inside the round's send segment:
//   * the plan's count pass (COUNT) leaves the number of packets of every (wave, destination) in wtab[wave][P]; the host
//     turns them into exclusive offsets along the waves of a round, so the position of every packet in the send buffer is
//     fixed by the plan: no cursor atomics, no bucket reservation, and the packet order (hence the order in which the
//     consumer's atomics arrive) no longer depends on the block schedule;
//   * stage A appends the packets of three flip-mask groups to the ring, stage B takes chunks of 64 out of it: projection
//     (inversion | orbit minimum | state_info), owner hash, then a loop over the DISTINCT destinations of the chunk:
//     ballot + mbcnt = rank inside the chunk, readlane of the destination's cursor and segment offsets, one store of beta
//     and one of the value straight into the send segment (consecutive chunks continue the same run of every segment).
//   No block barrier anywhere; LDS holds only the rings (16 KB f64, 24 KB c128 per block).
// ---------------------------------------------------------------------------------------------
constexpr int kTwRing = 256;
constexpr int kTwGroups = 3;
__device__ __forceinline__ int64_t readlane_i64(int64_t v, int lane) {
    return (int64_t)readlane_t<uint64_t>((uint64_t)v, lane);
}
// PK12 (pre-indexed packets; unprojected fixed-weight bases): the local index of EVERY packet at its destination -- the own
// partition included -- is read off the all-destinations directory gd (one 16-byte load), remote packets leave as (u32 index,
// value) and the consumer neither ranks nor searches.
template <typename W, bool PM1, bool CPLX, bool REAL, bool COUNT, bool PK12>
__global__ __launch_bounds__(kBlock) void k_tile_wv(int n_groups, lsk_group const *__restrict__ groups,
                                                    lsk_term const *__restrict__ off, lsk_basis bs,
                                                    lsk_group_elem const *__restrict__ elems, lsk_index ix, lsk_gdir gd, Owner owner,
                                                    int me, int64_t row0, int64_t row1, uint64_t const *__restrict__ reps,
                                                    double const *__restrict__ norms, double const *__restrict__ x, double *y,
                                                    uint32_t *__restrict__ wtab, lsk_round_layout const *__restrict__ layout,
                                                    char *send, int *err, lsk_gtab own_gt) {
    constexpr int kCap = (kBlock / 64) * kTwRing;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[COUNT ? 1 : kCap * (CPLX ? 2 : 1)];
    // PK12: colex rank of beta when it is one binomial away from alpha's (exchange on adjacent sites: rank(alpha) +- C(lo, k)), so that
    // the rank sum of the directory look-up runs once per row, not once per packet; kNoRank: the full sum (see k_tile_st)
    __shared__ uint32_t s_rank[(!COUNT && PK12) ? kCap : 1];
    extern __shared__ uint64_t s_db[]; // rank directory of the own partition: binomials of the closed-form rank (0 bytes without one)
    if (!COUNT && PK12) { gdir_load(gd, ix.binom, s_db); __syncthreads(); }
    else if (!COUNT && ix.dir) { rankdir_load(ix, s_db); __syncthreads(); }
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kTwRing;
    const int P = (int)owner.P;
    // lane d keeps what the wave knows about destination d: segment offsets of the round and the running cursor
    int64_t seg_b = 0, seg_v = 0;
    if (!COUNT && lane < P) { seg_b = layout->beta_off[lane]; seg_v = layout->val_off[lane]; }
    for (int64_t t0 = row0 + (int64_t)blockIdx.x * kBlock; t0 < row1; t0 += (int64_t)gridDim.x * kBlock) {
        if (t0 + (wave << 6) >= row1) continue; // wave-uniform: nothing below synchronises the block
        const int64_t i = t0 + tid;
        const bool valid = i < row1;
        uint64_t a = 0;
        double xr = 0.0, xi = 0.0;
        if (valid) {
            a = reps[i];
            if (COUNT) xr = 1.0; // the packet set must not depend on x (exact send counts)
            else {
                if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; } else xr = x[i];
                if (bs.proj == LSK_PROJ_FULL) { // fold 1 / norm(alpha) into x  (BatchedOperator.chpl:198-202)
                    const double na = norms[i];
                    const double s = na > 0.0 ? 1.0 / na : 0.0;
                    xr *= s;
                    xi *= s;
                }
            }
        }
        const bool narrow_ranks = !COUNT && PK12 && bs.proj == LSK_PROJ_NONE && gd.n_ranks <= 0xffffffffLL;
        uint64_t ga = 0; // colex rank of alpha
        if (narrow_ranks && valid) {
            const int kc = gd.weight + 1;
            uint64_t t = a;
            int k = 1;
            while (t && k < kc) { ga += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
        }
        const int64_t wg = ((t0 - row0) >> 6) + wave; // this wave's 64 rows inside the round
        uint32_t cur = 0;                             // lane d: packets so far (COUNT) | next position in segment d
        if (!COUNT && lane < P) cur = wtab[wg * P + lane];
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kTwRing
        auto chunk = [&](int m) {
            bool live = lane < m;
            const int e = rb + ((head + lane) & (kTwRing - 1));
            uint64_t beta = live ? s_beta[e] : 0;
            double vr = 0.0, vi = 0.0;
            if (!COUNT) { if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e]; }
            if (bs.proj == LSK_PROJ_INVERSION) {
                const uint64_t f = beta ^ bs.site_mask;
                if (f < beta) { beta = f; vr *= (double)bs.spin_inversion; vi *= (double)bs.spin_inversion; }
            } else if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) {
                beta = (uint64_t)rep_trivial<W>(bs, elems, (W)beta); // norm(rep) applied at index time
            } else if (bs.proj == LSK_PROJ_FULL) {
                if (live && !(kAblate && (bs.debug_ablate & 256))) { // (profiling builds: no K4)
                    W rep; double chr, chi, stab;
                    state_info_w<W, PM1>(bs, elems, (W)beta, rep, chr, chi, stab);
                    const double n2 = stab * bs.inv_order;
                    if (n2 > 1e-12) {
                        const double nb = sqrt(n2);
                        beta = (uint64_t)rep;
                        if (CPLX) { const double tr = (vr * chr - vi * chi) * nb, ti = (vr * chi + vi * chr) * nb; vr = tr; vi = ti; }
                        else vr = vr * chr * nb;
                    } else live = false; // zero-norm orbit: c == 0 => skipped (DMV:110)
                }
            }
            const int dest = live ? ((kAblate && (bs.debug_ablate & 2048)) ? (int)(beta % owner.P) : owner_of(beta, owner)) : -1; // (profiling builds: no hash)
            bool remote = live;
            uint32_t pidx = 0; // PK12: the packet's index inside its destination's block
            if (!COUNT && PK12) {
                if (live) {
                    const uint32_t rk = s_rank[e];
                    const int64_t idx = rk != kNoRank ? gdir_index_of_rank(gd, (uint64_t)rk, dest) : gdir_index(gd, beta, dest, s_db);
                    if (idx < 0) { // not a basis state (DMV:115-118): the flag halts the matvec; the slot the count pass reserved for the
                        atomicExch(err, 1); // packet is still filled -- with (index 0, value 0) -- so that no consumer meets a stale key
                        if (dest == me) remote = false;
                        else { vr = 0.0; vi = 0.0; }
                    } else if (dest == me) {
                        remote = false;
                        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                        else atomic_add_f64(y + idx, vr);
                    } else pidx = (uint32_t)idx;
                }
            } else if (!COUNT) {
                if (kAblate && (bs.debug_ablate & 1024) && live && dest == me) remote = false; // (profiling builds: own packets dropped)
                else if (live && dest == me) {
                    remote = false;
                    // (own-partition packets are indexed HERE, by the 1 / P of the lanes they fall on while the rest of the wave waits: the
                    // dependent loads of a binary search cost the whole wave -- ablation, chain_36_symm x 8: 13 of 62 ms; one probe of
                    // the partition's hash index when the plan has it)
                    const int64_t idx = ix.dir ? rankdir_index(ix, beta, s_db) : (own_gt.entries ? gtab_index(own_gt, beta) : search_index(ix, beta));
                    if (idx < 0) atomicExch(err, 1);
                    else {
                        if (bs.proj == LSK_PROJ_FULL && bs.k4_mode != 0) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
                        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
                        else atomic_add_f64(y + idx, vr);
                    }
                }
            }
            // one pass per distinct destination of the chunk (<= min(P, 64))
            unsigned long long rem = __ballot(remote);
            while (rem) {
                const int l = __builtin_ctzll(rem);
                const int d = __builtin_amdgcn_readlane(dest, l);
                const bool mine = remote && dest == d;
                const unsigned long long mm = __ballot(mine);
                if (!COUNT) {
                    const uint32_t base = (uint32_t)__builtin_amdgcn_readlane((int)cur, d);
                    const int64_t ob = readlane_i64(seg_b, d), ov = readlane_i64(seg_v, d);
                    if (mine && !(kAblate && (bs.debug_ablate & 512))) { // (profiling builds: nothing stored)
                        const size_t pos = (size_t)base + (size_t)__popcll(mm & ((1ULL << lane) - 1));
                        if (PK12) reinterpret_cast<uint32_t *>(send + ob)[pos] = pidx;
                        else reinterpret_cast<uint64_t *>(send + ob)[pos] = beta;
                        double *pv = reinterpret_cast<double *>(send + ov);
                        if (CPLX) { pv[2 * pos] = vr; pv[2 * pos + 1] = vi; } else pv[pos] = vr;
                    }
                }
                if (lane == d) cur += (uint32_t)__popcll(mm);
                rem &= ~mm;
            }
        };
        for (int g0 = 0; g0 < n_groups; g0 += kTwGroups) {
            const int g1 = min(g0 + kTwGroups, n_groups);
            for (int g = g0; g < g1; ++g) { // stage A: append
                lsk_group const G = groups[g];
                double cr = 0.0, ci = 0.0;
                if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                const bool act = valid && (cr != 0.0 || (!REAL && ci != 0.0));
                const unsigned long long ball = __ballot(act);
                if (act) {
                    const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kTwRing - 1));
                    s_beta[slot] = a ^ G.x;
                    if (!COUNT) {
                        if (CPLX) { s_val[2 * slot] = cr * xr - ci * xi; s_val[2 * slot + 1] = cr * xi + ci * xr; }
                        else s_val[slot] = cr * xr;
                        if (PK12) {
                            uint32_t rk = kNoRank;
                            if (narrow_ranks && G.fast == LSK_GROUP_EXCHANGE && G.adj >= 0) {
                                const uint64_t c = s_db[G.adj * (gd.weight + 1) + __popcll(a & ((1ULL << G.adj) - 1))];
                                rk = (uint32_t)(((a >> G.adj) & 1ULL) ? ga + c : ga - c); // the lower site's bit moves up | the upper one's down
                            }
                            s_rank[slot] = rk;
                        }
                    }
                }
                cnt += __popcll(ball);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            while (cnt >= 64) {
                chunk(64);
                head = (head + 64) & (kTwRing - 1);
                cnt -= 64;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (cnt > 0) chunk(cnt);
        if (COUNT && lane < P) wtab[wg * P + lane] = cur;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
}

extern "C" int lsk_tile_wv_max_parts(void) { return 64; }
// rows [row0, row1) of partition `me` (row0 = first row of the round: wave w of the round owns rows row0 + 64 w ...).
// count_only: wtab[w][P] <- packets of wave w per destination (the own partition included); otherwise wtab holds the
// exclusive offsets of every (wave, destination) inside the round's segments and the packets are written to d_send.
extern "C" int lsk_tile_wv(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_gdir gd, int cplx, int count_only, int P, int me,
                           int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x, void *y,
                           uint32_t *d_wtab, lsk_round_layout const *d_layout, void *d_send, int *d_err, lsk_gtab own_gt, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (P > lsk_tile_wv_max_parts() || P < 1 || !d_wtab) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv: bad partition count %d or no wave table", P); return -1; }
    const bool pk12 = !count_only && gd.entries != nullptr;
    if (!count_only && !pk12 && ix.kind != LSK_INDEX_SEARCH) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv needs a SEARCH index"); return -1; }
    if (pk12 && (bs.proj == LSK_PROJ_FULL || gd.P != P || !ix.binom)) { snprintf(g_err, sizeof(g_err), "lsk_tile_wv: pre-indexed packets need an unprojected basis and a directory over %d partitions", P); return -1; }
    Owner ow = make_owner(P);
    dim3 g(1), b(kBlock);
    const int64_t work_blocks = (row1 - row0 + kBlock - 1) / kBlock;
    hipStream_t s = (hipStream_t)stream;
    const size_t dyn_db = pk12 ? sizeof(uint64_t) * (size_t)gd.sites * (size_t)(gd.weight + 1)
                               : (ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0);
#define LSK_TW_ARGS op.n_groups, op.groups, op.off, bs, bs.elems, ix, gd, ow, me, row0, row1, reps, norms, (double const *)x, (double *)y, \
        d_wtab, d_layout, (char *)d_send, d_err, own_gt
#define LSK_TW_ONE(W, PM1, CPLX, REAL)                                                                                           \
    do {                                                                                                                         \
        if (count_only) { g.x = tile_grid(k_tile_wv<W, PM1, CPLX, REAL, true, false>, work_blocks); hipLaunchKernelGGL((k_tile_wv<W, PM1, CPLX, REAL, true, false>), g, b, 0, s, LSK_TW_ARGS); } \
        else { g.x = tile_grid(k_tile_wv<W, PM1, CPLX, REAL, false, false>, work_blocks); hipLaunchKernelGGL((k_tile_wv<W, PM1, CPLX, REAL, false, false>), g, b, dyn_db, s, LSK_TW_ARGS); } \
    } while (0)
#define LSK_TW_LAUNCH(W, PM1)                                                                                  \
    do {                                                                                                       \
        if (cplx) { if (op.is_real) LSK_TW_ONE(W, PM1, true, true); else LSK_TW_ONE(W, PM1, true, false); }    \
        else LSK_TW_ONE(W, PM1, false, true); /* f64 vectors: real operators only (the plan refuses the rest) */ \
    } while (0)
    if (pk12) { // unprojected bases only: W / PM1 are irrelevant
        if (cplx) {
            if (op.is_real) { g.x = tile_grid(k_tile_wv<uint64_t, true, true, true, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, true, true, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
            else { g.x = tile_grid(k_tile_wv<uint64_t, true, true, false, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, true, false, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
        } else { g.x = tile_grid(k_tile_wv<uint64_t, true, false, true, false, true>, work_blocks); hipLaunchKernelGGL((k_tile_wv<uint64_t, true, false, true, false, true>), g, b, dyn_db, s, LSK_TW_ARGS); }
        LSK_LAUNCH_CHECK();
        return 0;
    }
    const bool narrow = bs.number_sites <= 32 && bs.proj == LSK_PROJ_FULL;
    if (narrow) { if (bs.chars_pm1) LSK_TW_LAUNCH(uint32_t, true); else LSK_TW_LAUNCH(uint32_t, false); }
    else if (bs.proj == LSK_PROJ_FULL) { if (bs.chars_pm1) LSK_TW_LAUNCH(uint64_t, true); else LSK_TW_LAUNCH(uint64_t, false); }
    else LSK_TW_LAUNCH(uint64_t, true); // no projection: PM1 is irrelevant
#undef LSK_TW_LAUNCH
#undef LSK_TW_ONE
#undef LSK_TW_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}


// ---------------------------------------------------------------------------------------------
// Consumer side (K7 + K8): received packets -> local index -> atomic add
// ---------------------------------------------------------------------------------------------
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter(lsk_index ix, int64_t n, uint64_t const *__restrict__ betas,
                                                    double const *__restrict__ vals, double *y,
                                                    double const *__restrict__ norms, int *err, int xcd_chunk) {
    // XCD-chunked block -> packets map (pull_tile_of_block): the look-ups of neighbouring packet blocks read neighbouring table /
    // representative lines, which then meet in ONE L2 instead of eight
    extern __shared__ uint64_t s_db[]; // rank directory: the binomials of the closed-form rank (launch-time size, 0 without one)
    if (ix.dir) { rankdir_load(ix, s_db); __syncthreads(); }
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * k]; vi = vals[2 * k + 1]; } else vr = vals[k];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)betas[k] : (ix.dir ? rankdir_index(ix, betas[k], s_db) : search_index(ix, betas[k]));
        if (idx < 0) { atomicExch(err, 1); continue; }
        if (norms) { double nb = norms[idx]; vr *= nb; vi *= nb; }
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter(lsk_index ix, int cplx, int64_t n, uint64_t const *betas, void const *vals, void *y,
                           double const *norms, int *d_err, void *stream) {
    if (n == 0) return 0;
    if (ix.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter: SEARCH/IDENTITY index only"); return -1; }
    dim3 g(grid_for(n)), b(kBlock);
    // one block per 256 packets, 64 consecutive blocks per XCD (chain_28 x 8 partitions: consumers 14.85 -> 14.37 ms; chunks of
    // 1 / 16 / 256 / 1024: 14.65 / 14.42 / 14.36 / 14.39 -- profiles/r4_scatter_xcd_chunk_ab.txt)
    constexpr int chunk = 64;
    { const int64_t nb = (n + kBlock - 1) / kBlock; g.x = (unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30)); }
    const size_t dyn = ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter<true>, g, b, dyn, (hipStream_t)stream, ix, n, betas, (double const *)vals, (double *)y, norms, d_err, chunk);
    else hipLaunchKernelGGL(k_scatter<false>, g, b, dyn, (hipStream_t)stream, ix, n, betas, (double const *)vals, (double *)y, norms, d_err, chunk);
    LSK_LAUNCH_CHECK();
    return 0;
}

// Fused consumers: ONE launch over all segments of a round's receive buffer (chain_28 x 8 partitions ran 56 launches of 0.26 ms
// per matvec, each with its own ramp and tail).  Block b takes packets [256 b, 256 b + 256) of the concatenated count space;
// the segment of a packet is found in an LDS copy of the (<= 65) segment starts.
__device__ __forceinline__ int seg_of(int64_t const *s_start, int n, int64_t k) {
    int lo = 0, hi = n; // largest s with start[s] <= k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_start[mid] <= k) lo = mid; else hi = mid;
    }
    return lo;
}
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_idx(lsk_segs segs, char const *__restrict__ base, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        const uint32_t idx = reinterpret_cast<uint32_t const *>(base + segs.key_off[sg])[j];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double *y = reinterpret_cast<double *>(segs.y[sg]);
        if (CPLX) {
            const double vr = vals[2 * j], vi = vals[2 * j + 1];
            if (vr == 0.0 && vi == 0.0) continue; // DMV:110
            atomic_add_f64(y + 2 * (size_t)idx, vr);
            atomic_add_f64(y + 2 * (size_t)idx + 1, vi);
        } else {
            const double vr = vals[j];
            if (vr == 0.0) continue;
            atomic_add_f64(y + idx, vr);
        }
    }
}
extern "C" int lsk_scatter_idx(int cplx, lsk_segs const *segs, void const *base, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_idx: %d segments", segs->n); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    if (cplx) hipLaunchKernelGGL(k_scatter_idx<true>, g, b, 0, (hipStream_t)stream, *segs, (char const *)base, 64);
    else hipLaunchKernelGGL(k_scatter_idx<false>, g, b, 0, (hipStream_t)stream, *segs, (char const *)base, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}
// ... and for packets that carry the state (projected bases, or no room for the all-destinations directory): one index, one y
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_segs(lsk_index ix, lsk_gtab gt, lsk_segs segs, char const *__restrict__ base,
                                                         double const *__restrict__ norms, int *err, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    extern __shared__ uint64_t s_db[];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    if (ix.dir) rankdir_load(ix, s_db);
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    double *y = reinterpret_cast<double *>(segs.y[0]);
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * j]; vi = vals[2 * j + 1]; } else vr = vals[j];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        const uint64_t beta = reinterpret_cast<uint64_t const *>(base + segs.key_off[sg])[j];
        const int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)beta : (ix.dir ? rankdir_index(ix, beta, s_db) : (gt.entries ? gtab_index(gt, beta) : search_index(ix, beta)));
        if (idx < 0) { atomicExch(err, 1); continue; }
        if (norms) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter_segs(lsk_index ix, lsk_gtab gt, int cplx, lsk_segs const *segs, void const *base, double const *norms, int *d_err, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_segs: %d segments", segs->n); return -1; }
    if (ix.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter_segs: SEARCH/IDENTITY index only"); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    const size_t dyn = ix.dir ? sizeof(uint64_t) * (size_t)ix.dir_sites * (size_t)(ix.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter_segs<true>, g, b, dyn, (hipStream_t)stream, ix, gt, *segs, (char const *)base, norms, d_err, 64);
    else hipLaunchKernelGGL(k_scatter_segs<false>, g, b, dyn, (hipStream_t)stream, ix, gt, *segs, (char const *)base, norms, d_err, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ... and with the segments of one producer going to DIFFERENT partitions of this process (P logical partitions on one device: the
// "exchange" is a pointer hand-off): index and norms of a segment come from a device array of per-partition contexts
template <bool CPLX>
__global__ __launch_bounds__(kBlock) void k_scatter_parts(lsk_part_ctx const *__restrict__ parts, lsk_index any, lsk_segs segs,
                                                          char const *__restrict__ base, int *err, int xcd_chunk) {
    __shared__ int64_t s_start[LSK_MAX_SEGS + 1];
    extern __shared__ uint64_t s_db[];
    for (int i = threadIdx.x; i <= segs.n; i += kBlock) s_start[i] = segs.start[i];
    if (any.dir) rankdir_load(any, s_db); // (sites, weight and the binomials are those of the basis: the same for every partition)
    __syncthreads();
    const int64_t n = segs.start[segs.n];
    const int64_t n_blocks = (n + kBlock - 1) / kBlock;
    for (int64_t kb = blockIdx.x; kb < n_blocks; kb += gridDim.x) {
        const int64_t k = pull_tile_of_block(kb, n_blocks, gridDim.x >= n_blocks ? xcd_chunk : 0) * kBlock + threadIdx.x;
        if (k >= n) continue;
        const int sg = seg_of(s_start, segs.n, k);
        const int64_t j = k - s_start[sg];
        double const *vals = reinterpret_cast<double const *>(base + segs.val_off[sg]);
        double vr, vi = 0.0;
        if (CPLX) { vr = vals[2 * j]; vi = vals[2 * j + 1]; } else vr = vals[j];
        if (vr == 0.0 && vi == 0.0) continue; // DMV:110
        const uint64_t beta = reinterpret_cast<uint64_t const *>(base + segs.key_off[sg])[j];
        lsk_part_ctx const *pc = parts + segs.part[sg];
        lsk_index ix = any; // kind, binom, dir_sites, dir_weight: common; the rest per partition
        ix.shift = pc->ix.shift; ix.count = pc->ix.count; ix.reps = pc->ix.reps; ix.table = pc->ix.table; ix.dir = pc->ix.dir; ix.kind = pc->ix.kind;
        const int64_t idx = ix.kind == LSK_INDEX_IDENTITY ? (int64_t)beta : (ix.dir ? rankdir_index(ix, beta, s_db) : (pc->gt.entries ? gtab_index(pc->gt, beta) : search_index(ix, beta)));
        if (idx < 0) { atomicExch(err, 1); continue; }
        double const *norms = pc->norms;
        if (norms) { const double nb = norms[idx]; vr *= nb; vi *= nb; }
        double *y = reinterpret_cast<double *>(segs.y[sg]);
        if (CPLX) { atomic_add_f64(y + 2 * idx, vr); atomic_add_f64(y + 2 * idx + 1, vi); }
        else atomic_add_f64(y + idx, vr);
    }
}
extern "C" int lsk_scatter_parts(lsk_part_ctx const *d_parts, lsk_index any, int cplx, lsk_segs const *segs, void const *base, int *d_err, void *stream) {
    if (segs->n < 1 || segs->n > LSK_MAX_SEGS) { snprintf(g_err, sizeof(g_err), "lsk_scatter_parts: %d segments", segs->n); return -1; }
    if (any.kind == LSK_INDEX_COMBINADIC) { snprintf(g_err, sizeof(g_err), "lsk_scatter_parts: SEARCH/IDENTITY index only"); return -1; }
    const int64_t n = segs->start[segs->n];
    if (n <= 0) return 0;
    const int64_t nb = (n + kBlock - 1) / kBlock;
    dim3 g((unsigned)(nb < ((int64_t)1 << 30) ? nb : ((int64_t)1 << 30))), b(kBlock);
    const size_t dyn = any.dir ? sizeof(uint64_t) * (size_t)any.dir_sites * (size_t)(any.dir_weight + 1) : 0;
    if (cplx) hipLaunchKernelGGL(k_scatter_parts<true>, g, b, dyn, (hipStream_t)stream, d_parts, any, *segs, (char const *)base, d_err, 64);
    else hipLaunchKernelGGL(k_scatter_parts<false>, g, b, dyn, (hipStream_t)stream, d_parts, any, *segs, (char const *)base, d_err, 64);
    LSK_LAUNCH_CHECK();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Packets in SORTED STREAMS: consumers without atomics (unprojected fixed-weight bases, exchange operators).
//
// What bounds every consumer above is one fabric atomic per packet (~40 G/s: chain_28 x 8 partitions 13 of 22.6 ms).  But the
// packets of an exchange pair are not random: for a fixed pair (i, j) and a fixed pattern of alpha on it (01 or 10) the map
// alpha -> beta = alpha ^ x adds a CONSTANT, so it is monotone; the rows of a producer ascend, the states of a destination
// ascend, hence the destination indices of the packets of one STREAM = (pair, pattern) ascend along the producer's rows.
// A producer that writes every (destination, stream) as its own run of the send segment -- in row order -- hands the consumer
// 2 n_groups SORTED runs per source.  The consumer (k_window) then owns a WINDOW of W consecutive rows of y: it finds the
// sub-run of every stream that falls into the window by binary search on the keys, streams those packets (coalesced 12-byte
// reads) into an LDS copy of the window (ds_add_f64) and adds the window to y once: no global atomics, no fabric request per
// packet, and the packet order inside y's window no longer matters.
//
// Producer (k_tile_st): k_tile_wv's wave rings, but a wave walks a TILE of tile_rows rows (64 at a time, the ring carried over)
// and keeps one cursor per CLASS = (destination, stream) in LDS, initialised from the plan's table ttab[tile][class] = absolute
// position of the tile's first packet of that class inside the destination's segment (count pass + host scan, like wtab).  The
// rank of a packet among the packets of its class inside a chunk of 64 comes from one ballot per class BIT (9 ballots for
// 8 destinations x 56 streams) instead of one pass per destination; ring order = (group, lane) order, so the packets of one
// class leave in row order: every stream is EXACTLY sorted.  Own-partition packets take the same way (no atomics here either).
// ---------------------------------------------------------------------------------------------
constexpr int kStRing = 256;
constexpr int kStMaxClasses = 1024; // LDS: 4 waves x classes x 4 bytes of cursors
#define LSK_WAVE_SYNC()                                              \
    do {                                                             \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       \
        __builtin_amdgcn_wave_barrier();                             \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");       \
    } while (0)
template <bool CPLX, bool REAL, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_tile_st(int n_groups, lsk_group const *__restrict__ groups,
                                                    lsk_term const *__restrict__ off, lsk_gdir gd,
                                                    uint64_t const *__restrict__ g_binom, Owner owner, int S, int cbits,
                                                    int tile_rows, int64_t row0, int64_t row1, int64_t n_tiles,
                                                    uint64_t const *__restrict__ reps, double const *__restrict__ x,
                                                    uint32_t *__restrict__ ttab, lsk_round_layout const *__restrict__ layout,
                                                    char *send, int *err, int xcd_chunk) {
    constexpr int kWaves = kBlock / 64;
    constexpr int kCap = kWaves * kStRing;
    __shared__ uint64_t s_beta[kCap];
    __shared__ double s_val[COUNT ? 1 : kCap * (CPLX ? 2 : 1)];
    __shared__ uint8_t s_sid[kCap];
    // global (colex) rank of beta when it is one binomial away from alpha's: an exchange on ADJACENT sites (lo, lo + 1) moves the
    // (k + 1)-th set bit by one place, rank(beta) = rank(alpha) +- C(lo, k), k = set bits of alpha below lo -- the 14-step rank
    // sum of the directory look-up then runs once per row instead of once per packet (kNoRank: the full sum, e.g. the bond
    // that closes a ring; bases with >= 2^32 states always take it)
    __shared__ uint32_t s_rank[COUNT ? 1 : kCap];
    extern __shared__ uint64_t s_dyn[]; // [binomials of the directory][key offsets P][value offsets P][cursors: waves x classes u32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int rb = wave * kStRing;
    const int P = (int)owner.P, C = P * S;
    const int ndb = COUNT ? 0 : gd.sites * (gd.weight + 1);
    uint64_t *s_db = s_dyn;
    int64_t *s_koff = reinterpret_cast<int64_t *>(s_dyn + ndb);
    int64_t *s_voff = s_koff + (COUNT ? 0 : P);
    uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_voff + (COUNT ? 0 : P)) + wave * C;
    if (!COUNT) {
        gdir_load(gd, g_binom, s_db);
        for (int d = tid; d < P; d += kBlock) { s_koff[d] = layout->beta_off[d]; s_voff[d] = layout->val_off[d]; }
        __syncthreads();
    }
    const int64_t n_work = (n_tiles + kWaves - 1) / kWaves; // a block takes kWaves consecutive tiles, one per wave
    for (int64_t wb = blockIdx.x; wb < n_work; wb += gridDim.x) {
        // (consecutive tiles on ONE XCD: they append to the same lines of every stream, which then fill up inside one L2)
        const int64_t tile = pull_tile_of_block(wb, n_work, gridDim.x >= n_work ? xcd_chunk : 0) * kWaves + wave;
        if (tile >= n_tiles) continue; // wave-uniform: nothing below synchronises the block
        const int64_t t0 = row0 + tile * tile_rows;
        const int64_t t1 = t0 + tile_rows < row1 ? t0 + tile_rows : row1;
        uint32_t *trow = ttab + (size_t)tile * C;
        for (int c = lane; c < C; c += 64) s_cur[c] = COUNT ? 0u : trow[c];
        LSK_WAVE_SYNC();
        int head = 0, cnt = 0; // wave-uniform: the ring holds [head, head + cnt) mod kStRing
        auto chunk = [&](int m) {
            const bool live = lane < m;
            const int e = rb + ((head + lane) & (kStRing - 1));
            const uint64_t beta = live ? s_beta[e] : 0;
            const int dest = live ? owner_of(beta, owner) : 0;
            const uint32_t cls = (uint32_t)dest * (uint32_t)S + (live ? (uint32_t)s_sid[e] : 0u);
            // rank among the packets of the same class in this chunk: lanes that agree with me on every class bit
            unsigned long long same = __ballot(live);
            for (int b = 0; b < cbits; ++b) {
                const bool bit = (cls >> b) & 1u;
                const unsigned long long bb = __ballot(live && bit);
                same &= bit ? bb : ~bb;
            }
            const uint32_t rank = (uint32_t)__popcll(same & ((1ULL << lane) - 1));
            const uint32_t n_same = (uint32_t)__popcll(same);
            const uint32_t base = live ? s_cur[cls] : 0u;
            LSK_WAVE_SYNC(); // every lane has read its cursor before the last lane of a class moves it
            if (live && rank + 1 == n_same) s_cur[cls] = base + n_same;
            if (!COUNT && live) {
                double vr, vi = 0.0;
                if (CPLX) { vr = s_val[2 * e]; vi = s_val[2 * e + 1]; } else vr = s_val[e];
                const uint32_t rk = s_rank[e];
                int64_t idx;
                if (gd.entries) idx = rk != kNoRank ? gdir_index_of_rank(gd, (uint64_t)rk, dest) : gdir_index(gd, beta, dest, s_db);
                else if (rk != kNoRank) idx = (int64_t)rk; // GLOBAL-RANK keys (lsk_wdests): no directory on the producer's side at all
                else { // (a pair that is not adjacent, e.g. the bond that closes a ring: the full rank sum)
                    idx = -1;
                    if (__popcll(beta) == gd.weight && (gd.sites >= 64 || (beta >> gd.sites) == 0)) {
                        const int kc = gd.weight + 1;
                        uint64_t t = beta, g = 0;
                        int k = 1;
                        while (t) { g += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
                        if ((int64_t)g < gd.n_ranks) idx = (int64_t)g;
                    }
                }
                if (idx < 0) { // not a basis state (DMV:115-118): the flag halts the matvec; the slot the count pass reserved is still
                    atomicExch(err, 1); // filled -- (index 0, value 0) -- so that no consumer meets a stale key
                    idx = 0; vr = 0.0; vi = 0.0;
                }
                const size_t pos = (size_t)base + rank;
                reinterpret_cast<uint32_t *>(send + s_koff[dest])[pos] = (uint32_t)idx;
                double *pv = reinterpret_cast<double *>(send + s_voff[dest]);
                if (CPLX) { pv[2 * pos] = vr; pv[2 * pos + 1] = vi; } else pv[pos] = vr;
            }
            LSK_WAVE_SYNC();
        };
        for (int64_t r0 = t0; r0 < t1; r0 += 64) {
            const int64_t i = r0 + lane;
            const bool valid = i < t1;
            uint64_t a = 0;
            double xr = 0.0, xi = 0.0;
            if (valid) {
                a = reps[i];
                if (COUNT) xr = 1.0; // the packet set must not depend on x (exact send counts)
                else if (CPLX) { xr = x[2 * i]; xi = x[2 * i + 1]; }
                else xr = x[i];
            }
            uint64_t ga = 0; // colex rank of alpha (states of another weight never reach the look-up: their rows have no packets)
            const bool narrow_ranks = !COUNT && gd.n_ranks <= 0xffffffffLL;
            if (narrow_ranks && valid) {
                const int kc = gd.weight + 1;
                uint64_t t = a;
                int k = 1;
                while (t && k < kc) { ga += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
            }
            for (int g0 = 0; g0 < n_groups; g0 += kTwGroups) {
                const int g1 = min(g0 + kTwGroups, n_groups);
                for (int g = g0; g < g1; ++g) { // stage A: append (beta, value, stream)
                    lsk_group const G = groups[g];
                    double cr = 0.0, ci = 0.0;
                    if (valid) group_coeff<REAL>(G, off, a, cr, ci);
                    // (every group is an exchange pair: a packet exists iff alpha is anti-aligned on it, whatever its amplitude)
                    const bool act = valid && __popcll(a & G.x) == 1;
                    const unsigned long long ball = __ballot(act);
                    if (act) {
                        const int slot = rb + ((head + cnt + __popcll(ball & ((1ULL << lane) - 1))) & (kStRing - 1));
                        const int up = (int)((a >> (__ffsll((unsigned long long)G.x) - 1)) & 1ULL); // the lower site's bit moves up
                        s_beta[slot] = a ^ G.x;
                        s_sid[slot] = (uint8_t)(2 * g + up);
                        if (!COUNT) {
                            if (CPLX) { s_val[2 * slot] = cr * xr - ci * xi; s_val[2 * slot + 1] = cr * xi + ci * xr; }
                            else s_val[slot] = cr * xr;
                            uint32_t rk = kNoRank;
                            if (narrow_ranks && G.adj >= 0) {
                                const uint64_t c = s_db[G.adj * (gd.weight + 1) + __popcll(a & ((1ULL << G.adj) - 1))];
                                rk = (uint32_t)(up ? ga + c : ga - c);
                            }
                            s_rank[slot] = rk;
                        }
                    }
                    cnt += __popcll(ball);
                }
                LSK_WAVE_SYNC();
                while (cnt >= 64) {
                    chunk(64);
                    head = (head + 64) & (kStRing - 1);
                    cnt -= 64;
                }
            }
        }
        if (cnt > 0) chunk(cnt);
        if (COUNT) for (int c = lane; c < C; c += 64) trow[c] = s_cur[c];
        LSK_WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// The same producer WITHOUT the ring (k_tile_sd; round 6, the default -- LS_AMD_STREAM_PRODUCER=ring keeps k_tile_st): a wave takes
// 64 rows and walks the groups with all of them; the packets of group g belong to the 2 P classes (destination, pattern) of that group only, so their ranks inside
// the class cost log2(2 P) ballots instead of log2(2 P n_groups), nothing is staged in LDS but the cursors, and the lanes that write
// one class in one step are neighbours in the stream (the next 64 rows append to the same lines).  Same packets, same order inside
// every stream, same ttab -- the count pass and the consumers do not know which producer ran.  Lanes whose row is aligned on the
// pair idle for that step: at half filling half of them.
// ---------------------------------------------------------------------------------------------
constexpr int kSdChunks = 4; // 64-row chunks a wave holds in registers while it walks the groups
template <bool CPLX, bool REAL, bool COUNT>
__global__ __launch_bounds__(kBlock) void k_tile_sd(int n_groups, lsk_group const *__restrict__ groups,
                                                    lsk_term const *__restrict__ off, lsk_gdir gd,
                                                    uint64_t const *__restrict__ g_binom, Owner owner, int S, int sbits,
                                                    int tile_rows, int64_t row0, int64_t row1, int64_t n_tiles,
                                                    uint64_t const *__restrict__ reps, double const *__restrict__ x,
                                                    uint32_t *__restrict__ ttab, lsk_round_layout const *__restrict__ layout,
                                                    char *send, int *err, int xcd_chunk) {
    constexpr int kWaves = kBlock / 64;
    extern __shared__ uint64_t s_dyn[]; // [binomials of the directory][key offsets P][value offsets P][cursors: waves x classes u32]
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int P = (int)owner.P, C = P * S;
    const int ndb = COUNT ? 0 : gd.sites * (gd.weight + 1);
    uint64_t *s_db = s_dyn;
    int64_t *s_koff = reinterpret_cast<int64_t *>(s_dyn + ndb);
    int64_t *s_voff = s_koff + (COUNT ? 0 : P);
    volatile uint32_t *s_cur = reinterpret_cast<uint32_t *>(s_voff + (COUNT ? 0 : P)) + wave * C; // (a wave's LDS accesses complete in order)
    if (!COUNT) {
        gdir_load(gd, g_binom, s_db);
        for (int d = tid; d < P; d += kBlock) { s_koff[d] = layout->beta_off[d]; s_voff[d] = layout->val_off[d]; }
        __syncthreads();
    }
    const unsigned long long below = (1ULL << lane) - 1;
    const bool narrow_ranks = !COUNT && gd.n_ranks <= 0xffffffffLL;
    const int64_t n_work = (n_tiles + kWaves - 1) / kWaves;
    for (int64_t wb = blockIdx.x; wb < n_work; wb += gridDim.x) {
        const int64_t tile = pull_tile_of_block(wb, n_work, gridDim.x >= n_work ? xcd_chunk : 0) * kWaves + wave;
        if (tile >= n_tiles) continue; // wave-uniform: nothing below synchronises the block
        const int64_t t0 = row0 + tile * tile_rows;
        const int64_t t1 = t0 + tile_rows < row1 ? t0 + tile_rows : row1;
        uint32_t *trow = ttab + (size_t)tile * C;
        for (int c = lane; c < C; c += 64) s_cur[c] = COUNT ? 0u : trow[c];
        // GROUPS OUTSIDE, rows inside: while a group is walked only its 2 P classes are being appended to, by this wave and by the
        // waves of the neighbouring tiles -- the open lines of an XCD fit its L2 and leave it complete (with the rows outside every
        // line of all P S classes stays half-written for the whole tile: measured 2.24 GB written per launch for 0.87 GB of packets)
        for (int64_t q0 = t0; q0 < t1; q0 += 64 * kSdChunks) {
            uint64_t a[kSdChunks], ga[kSdChunks];
            double xr[kSdChunks], xi[kSdChunks];
            bool valid[kSdChunks];
#pragma unroll
            for (int u = 0; u < kSdChunks; ++u) {
                const int64_t i = q0 + 64 * u + lane;
                valid[u] = i < t1;
                a[u] = 0;
                xr[u] = 0.0;
                xi[u] = 0.0;
                if (valid[u]) {
                    a[u] = reps[i];
                    if (COUNT) xr[u] = 1.0;
                    else if (CPLX) { xr[u] = x[2 * i]; xi[u] = x[2 * i + 1]; }
                    else xr[u] = x[i];
                }
                ga[u] = 0; // colex rank of alpha
                if (narrow_ranks && valid[u]) {
                    const int kc = gd.weight + 1;
                    uint64_t t = a[u];
                    int k = 1;
                    while (t && k < kc) { ga[u] += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
                }
            }
            for (int g = 0; g < n_groups; ++g) {
                lsk_group const G = groups[g];
#pragma unroll
                for (int u = 0; u < kSdChunks; ++u) {
                    // (every group is an exchange pair: a packet exists iff alpha is anti-aligned on it, whatever its amplitude)
                    const bool act = valid[u] && __popcll(a[u] & G.x) == 1;
                    unsigned long long same = __ballot(act);
                    if (same == 0) continue;
                    const uint64_t beta = a[u] ^ G.x;
                    const int up = (int)((a[u] >> (__ffsll((unsigned long long)G.x) - 1)) & 1ULL); // the lower site's bit moves up
                    const int dest = owner_of(beta, owner);
                    const uint32_t sc = (uint32_t)dest * 2u + (uint32_t)up;
                    for (int b = 0; b < sbits; ++b) {
                        const bool bit = (sc >> b) & 1u;
                        const unsigned long long bb = __ballot(act && bit);
                        same &= bit ? bb : ~bb;
                    }
                    const uint32_t rank = (uint32_t)__popcll(same & below), n_same = (uint32_t)__popcll(same);
                    const uint32_t cls = (uint32_t)dest * (uint32_t)S + 2u * (uint32_t)g + (uint32_t)up;
                    if (act) {
                        const uint32_t base = s_cur[cls];
                        if (rank + 1 == n_same) s_cur[cls] = base + n_same;
                        if (!COUNT) {
                            double cr, ci;
                            group_coeff<REAL>(G, off, a[u], cr, ci);
                            double vr = cr * xr[u] - (CPLX ? ci * xi[u] : 0.0), vi = CPLX ? cr * xi[u] + ci * xr[u] : 0.0;
                            int64_t idx;
                            if (narrow_ranks && G.adj >= 0) {
                                // an exchange on ADJACENT sites moves one particle by one place: rank(beta) = rank(alpha) +- C(lo, k)
                                const uint64_t c = s_db[G.adj * (gd.weight + 1) + __popcll(a[u] & ((1ULL << G.adj) - 1))];
                                const uint32_t rk = (uint32_t)(up ? ga[u] + c : ga[u] - c);
                                idx = gd.entries ? gdir_index_of_rank(gd, (uint64_t)rk, dest) : (int64_t)rk;
                            } else if (gd.entries) idx = gdir_index(gd, beta, dest, s_db);
                            else { // GLOBAL-RANK keys, a pair that is not adjacent (e.g. the bond that closes a ring): the full rank sum
                                idx = -1;
                                if (__popcll(beta) == gd.weight && (gd.sites >= 64 || (beta >> gd.sites) == 0)) {
                                    const int kc = gd.weight + 1;
                                    uint64_t t = beta, gsum = 0;
                                    int k = 1;
                                    while (t) { gsum += s_db[(__ffsll((unsigned long long)t) - 1) * kc + k]; ++k; t &= t - 1; }
                                    if ((int64_t)gsum < gd.n_ranks) idx = (int64_t)gsum;
                                }
                            }
                            if (idx < 0) { // not a basis state (DMV:115-118): the flag halts the matvec; the reserved slot is still filled
                                atomicExch(err, 1);
                                idx = 0; vr = 0.0; vi = 0.0;
                            }
                            const size_t pos = (size_t)base + rank;
                            reinterpret_cast<uint32_t *>(send + s_koff[dest])[pos] = (uint32_t)idx;
                            double *pv = reinterpret_cast<double *>(send + s_voff[dest]);
                            if (CPLX) { pv[2 * pos] = vr; pv[2 * pos + 1] = vi; } else pv[pos] = vr;
                        }
                    }
                }
            }
        }
        if (COUNT) for (int c = lane; c < C; c += 64) trow[c] = s_cur[c];
    }
}

extern "C" int lsk_tile_st_max_classes(void) { return kStMaxClasses; }
// rows [row0, row1) of one partition, tile t = rows [row0 + t tile_rows, ...).  count_only: d_ttab[tile][P * S] <- packets of every
// (tile, class = destination * S + stream), stream = 2 * group + (bit of alpha at the pair's lower site); otherwise d_ttab holds
// the position of the tile's first packet of every class inside the destination's segment of *d_layout, and the packets --
// (u32 index at the destination, value), the own partition's included -- are written to d_send.
extern "C" int lsk_tile_st(lsk_operator op, lsk_gdir gd, uint64_t const *d_binom, int cplx, int count_only, int P, int S,
                           int tile_rows, int64_t row0, int64_t row1, uint64_t const *reps, void const *x, uint32_t *d_ttab,
                           lsk_round_layout const *d_layout, void *d_send, int *d_err, void *stream) {
    if (row1 <= row0 || op.n_groups == 0) return 0;
    if (S != 2 * op.n_groups || S > 256 || P < 1 || P * S > kStMaxClasses || tile_rows < 64 || (tile_rows & 63) || !d_ttab ||
        (!count_only && (gd.P != P || !d_binom || (!gd.entries && gd.n_ranks > 0xffffffffLL) /* global-rank keys are 32-bit */))) {
        snprintf(g_err, sizeof(g_err), "lsk_tile_st: bad arguments (P = %d, S = %d, tile_rows = %d)", P, S, tile_rows);
        return -1;
    }
    const int C = P * S;
    int cbits = 0;
    while ((1 << cbits) < C) ++cbits;
    Owner ow = make_owner(P);
    const int64_t n_tiles = (row1 - row0 + tile_rows - 1) / tile_rows;
    const int64_t n_work = (n_tiles + kBlock / 64 - 1) / (kBlock / 64);
    dim3 g(1), b(kBlock);
    hipStream_t s = (hipStream_t)stream;
    const size_t dyn = sizeof(uint32_t) * (size_t)(kBlock / 64) * (size_t)C +
                       (count_only ? 0 : sizeof(uint64_t) * ((size_t)gd.sites * (size_t)(gd.weight + 1) + 2 * (size_t)P));
#define LSK_ST_ARGS op.n_groups, op.groups, op.off, gd, d_binom, ow, S, cbits, tile_rows, row0, row1, n_tiles, reps, (double const *)x, \
        d_ttab, d_layout, (char *)d_send, d_err, 64
    // LS_AMD_STREAM_PRODUCER=ring: the round-5 producer (wave rings).  Default since round 6: k_tile_sd.  (The count pass and the
    // producer agree on nothing but the packet order, which is the same.)
    char const *prod_env = getenv("LS_AMD_STREAM_PRODUCER");
    const int direct = (prod_env && strcmp(prod_env, "ring") == 0) ? 0 : 1;
    if (direct) {
        int sbits = 0;
        while ((1 << sbits) < 2 * P) ++sbits;
        const size_t dyn_d = dyn; // the same image: binomials, offsets, cursors
#define LSK_SD_ARGS op.n_groups, op.groups, op.off, gd, d_binom, ow, S, sbits, tile_rows, row0, row1, n_tiles, reps, (double const *)x, \
        d_ttab, d_layout, (char *)d_send, d_err, 64
#define LSK_SD_ONE(CPLX, REAL)                                                                                                    \
    do {                                                                                                                          \
        if (count_only) { g.x = tile_grid(k_tile_sd<CPLX, REAL, true>, n_work); hipLaunchKernelGGL((k_tile_sd<CPLX, REAL, true>), g, b, dyn_d, s, LSK_SD_ARGS); } \
        else { g.x = tile_grid(k_tile_sd<CPLX, REAL, false>, n_work); hipLaunchKernelGGL((k_tile_sd<CPLX, REAL, false>), g, b, dyn_d, s, LSK_SD_ARGS); } \
    } while (0)
        if (cplx) { if (op.is_real) LSK_SD_ONE(true, true); else LSK_SD_ONE(true, false); }
        else LSK_SD_ONE(false, true);
#undef LSK_SD_ONE
#undef LSK_SD_ARGS
        LSK_LAUNCH_CHECK();
        return 0;
    }
#define LSK_ST_ONE(CPLX, REAL)                                                                                                    \
    do {                                                                                                                          \
        if (count_only) { g.x = tile_grid(k_tile_st<CPLX, REAL, true>, n_work); hipLaunchKernelGGL((k_tile_st<CPLX, REAL, true>), g, b, dyn, s, LSK_ST_ARGS); } \
        else { g.x = tile_grid(k_tile_st<CPLX, REAL, false>, n_work); hipLaunchKernelGGL((k_tile_st<CPLX, REAL, false>), g, b, dyn, s, LSK_ST_ARGS); } \
    } while (0)
    if (cplx) { if (op.is_real) LSK_ST_ONE(true, true); else LSK_ST_ONE(true, false); }
    else LSK_ST_ONE(false, true); // f64 vectors: real operators only (the plan refuses the rest)
#undef LSK_ST_ONE
#undef LSK_ST_ARGS
    LSK_LAUNCH_CHECK();
    return 0;
}

// Consumer of the sorted streams.  Block -> (destination partition, wpb consecutive windows of W rows of its y).  n_src source
// segments per destination, S streams each: soff[s] .. soff[s + 1] = packets of stream s inside the segment, keys ascending.
constexpr int kWinRows = 2048;    // doubles of one window's accumulator (c128: 1024 rows)
constexpr int kWinStreams = 512;  // run bounds kept in LDS per pass over the streams
constexpr int kWinRuns = 4;       // runs a wave has in flight (8: the registers cost more occupancy than the loads win -- c128 5.8 -> 7.5 ms)
constexpr int kWinDir = 512;      // entries of the destination's rank directory a window keeps in LDS (global-rank keys)
__device__ __forceinline__ uint32_t lower_bound_u32(uint32_t const *__restrict__ k, uint32_t lo, uint32_t hi, uint32_t v) {
    while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if (k[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first position >= lo whose key is >= v (the window's end is a few dozen packets on: gallop, then search)
__device__ __forceinline__ uint32_t gallop_u32(uint32_t const *__restrict__ k, uint32_t lo, uint32_t end, uint32_t v) {
    uint32_t step = 64;
    while ((uint64_t)lo + step <= (uint64_t)end && k[lo + step - 1] < v) { lo += step; step <<= 1; }
    const uint32_t hi = (uint64_t)lo + step < (uint64_t)end ? lo + step : end;
    return lower_bound_u32(k, lo, hi, v);
}
template <bool CPLX, int THREADS>
__global__ __launch_bounds__(THREADS) void k_window(lsk_wdests dests, lsk_wsrc const *__restrict__ srcs, int n_src, int S, int wpb) {
    constexpr int W = CPLX ? kWinRows / 2 : kWinRows;
    constexpr int WAVES = THREADS / 64;
    __shared__ double s_acc[kWinRows];
    __shared__ uint32_t s_lo[kWinStreams];
    __shared__ uint16_t s_len[kWinStreams]; // (the keys of a stream are distinct: a window holds <= W of them)
    __shared__ uint32_t const *s_keys[LSK_MAX_SEGS];
    __shared__ double const *s_vals[LSK_MAX_SEGS];
    __shared__ uint32_t s_gb[2]; // global-rank keys: the ranks of the window's first row and of the row behind its last
    // ... and the slice of the destination's rank directory those ranks span (round 6): W rows of a hash partition span ~ P W ranks
    // = P W / 64 entries, read once, coalesced -- the rank -> row translation of a packet is an LDS read instead of a dependent
    // 16-byte load behind the key's.  A window that spans more (P > 15) keeps the global loads.
    __shared__ uint2 s_dir[2 * kWinDir]; // as (bits, prefix) of 32 ranks each: the look-up is three 32-bit instructions
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int d = 0;
    while (d + 1 < dests.n && (int64_t)blockIdx.x >= dests.first_block[d + 1]) ++d; // block-uniform
    const int64_t n = dests.count[d];
    double *__restrict__ y = reinterpret_cast<double *>(dests.y[d]);
    lsk_rankdir const *__restrict__ dir = dests.dir[d]; // NULL: the keys are indices at the destination
    const bool gkeys = dir != nullptr;
    const int64_t wb = (int64_t)blockIdx.x - dests.first_block[d];
    const int T = n_src * S;
    lsk_wsrc const *__restrict__ segs = srcs + (size_t)d * n_src;
    for (int q = tid; q < n_src; q += THREADS) { s_keys[q] = segs[q].keys; s_vals[q] = segs[q].vals; }
    for (int win = 0; win < wpb; ++win) {
        const int64_t w0 = (wb * wpb + win) * W;
        if (w0 >= n) break;
        const int64_t w1 = w0 + W < n ? w0 + W : n;
        for (int i = tid; i < kWinRows; i += THREADS) s_acc[i] = 0.0;
        if (gkeys) { // the window in KEY space: [rank of row w0, rank of row w1) -- rows ascend, so do their ranks
            if (tid < 2) {
                const int64_t row = tid == 0 ? w0 : w1;
                uint64_t g = (uint64_t)dests.n_ranks;
                if (row < n) {
                    uint64_t t = dests.reps[d][row];
                    g = 0;
                    int k = 1;
                    while (t) { g += dests.binom[(__ffsll((unsigned long long)t) - 1) * LSK_BINOM_K + k]; ++k; t &= t - 1; }
                }
                s_gb[tid] = (uint32_t)(g > 0xffffffffULL ? 0xffffffffULL : g);
            }
            __syncthreads();
        }
        const uint32_t k0 = gkeys ? s_gb[0] : (uint32_t)w0, k1 = gkeys ? s_gb[1] : (uint32_t)w1;
        const uint32_t e0 = k0 >> 6, ne = gkeys ? (uint32_t)(((uint64_t)k1 + 63) >> 6) - e0 : 0u;
        const bool ldir = gkeys && ne <= (uint32_t)kWinDir;
        if (ldir) // (visible to everybody after the barrier behind the run searches)
            for (uint32_t j = tid; j < ne; j += THREADS) {
                const lsk_rankdir en = dir[e0 + j];
                const uint32_t lo32 = (uint32_t)en.bits;
                s_dir[2 * j] = make_uint2(lo32, en.prefix);
                s_dir[2 * j + 1] = make_uint2((uint32_t)(en.bits >> 32), en.prefix + (uint32_t)__popc(lo32));
            }
        const bool carry = win > 0 && T <= kWinStreams; // the end of the previous window's run is the start of this one's
        for (int t0 = 0; t0 < T; t0 += kWinStreams) {
            const int tn = T - t0 < kWinStreams ? T - t0 : kWinStreams;
            for (int t = tid; t < tn; t += THREADS) {
                const int q = (t0 + t) / S, s = (t0 + t) - q * S;
                uint32_t const *__restrict__ keys = segs[q].keys;
                uint32_t const *__restrict__ soff = segs[q].soff;
                const uint32_t e = soff[s + 1];
                const uint32_t lo = carry ? s_lo[t] + s_len[t] : lower_bound_u32(keys, soff[s], e, k0);
                uint32_t hi = w1 >= n ? e : gallop_u32(keys, lo, e, k1);
                if (hi - lo > (uint32_t)W) hi = lo + (uint32_t)W; // (only after a failed directory look-up: the flag is up anyway)
                s_lo[t] = lo;
                s_len[t] = (uint16_t)(hi - lo);
            }
            __syncthreads();
            // a wave takes the runs t = wave, wave + WAVES, ..., kWinRuns at a time: all their loads are issued before the first add
            for (int t = wave; t < tn; t += WAVES * kWinRuns) {
                uint32_t const *kp[kWinRuns];
                double const *vp[kWinRuns];
                uint32_t len[kWinRuns], longest = 0;
#pragma unroll
                for (int u = 0; u < kWinRuns; ++u) {
                    const int tu = t + WAVES * u;
                    const bool has = tu < tn;
                    const int q = has ? (t0 + tu) / S : 0;
                    const uint32_t lo = has ? s_lo[tu] : 0u;
                    len[u] = has ? (uint32_t)s_len[tu] : 0u;
                    kp[u] = s_keys[q] + lo;
                    vp[u] = s_vals[q] + (size_t)lo * (CPLX ? 2 : 1);
                    longest = len[u] > longest ? len[u] : longest;
                }
                for (uint32_t it = (uint32_t)lane; it < longest + (uint32_t)lane; it += 64) { // (wave-uniform trip count)
                    uint32_t key[kWinRuns];
                    double vr[kWinRuns], vi[kWinRuns];
#pragma unroll
                    for (int u = 0; u < kWinRuns; ++u) {
                        key[u] = 0xffffffffu; vr[u] = 0.0; vi[u] = 0.0;
                        if (it < len[u]) {
                            key[u] = kp[u][it];
                            if (CPLX) { vr[u] = vp[u][2 * (size_t)it]; vi[u] = vp[u][2 * (size_t)it + 1]; } else vr[u] = vp[u][it];
                        }
                    }
                    if (gkeys && ldir) { // rank -> row of y[d] out of the window's slice of the directory
#pragma unroll
                        for (int u = 0; u < kWinRuns; ++u) {
                            if (it >= len[u]) continue;
                            const uint32_t j = (key[u] >> 5) - 2 * e0; // (a key outside the window's ranks -- a misplaced segment -- raises the flag)
                            const uint2 en = j < 2 * ne ? s_dir[j] : make_uint2(0u, 0u);
                            const uint32_t b = key[u] & 31u;
                            if (!((en.x >> b) & 1u)) { atomicExch(dests.err, 1); key[u] = 0xffffffffu; continue; } // not a state of this partition
                            key[u] = en.y + (uint32_t)__popc(en.x & ((1u << b) - 1u));
                        }
                    } else if (gkeys) { // ... or ONE 16-byte entry of the directory per key (ascending keys: neighbouring lanes read the same
                        ulonglong2 en[kWinRuns]; // or the next entry)
#pragma unroll
                        for (int u = 0; u < kWinRuns; ++u)
                            en[u] = (it < len[u] && (int64_t)key[u] < dests.n_ranks) // (a key that is no rank at all reads nothing)
                                        ? *reinterpret_cast<ulonglong2 const *>(dir + (key[u] >> 6)) : make_ulonglong2(0, 0);
#pragma unroll
                        for (int u = 0; u < kWinRuns; ++u) {
                            if (it >= len[u]) continue;
                            const uint64_t bit = 1ULL << (key[u] & 63);
                            if (!(en[u].x & bit)) { atomicExch(dests.err, 1); key[u] = 0xffffffffu; continue; } // not a state of this partition
                            key[u] = (uint32_t)en[u].y + (uint32_t)__popcll(en[u].x & (bit - 1));
                        }
                    }
#pragma unroll
                    for (int u = 0; u < kWinRuns; ++u) {
                        const uint32_t o = key[u] - (uint32_t)w0; // (a key outside the window -- only after a failed look-up -- is dropped)
                        if (it < len[u] && o < (uint32_t)W) {
                            if (CPLX) { atomicAdd(&s_acc[2 * o], vr[u]); atomicAdd(&s_acc[2 * o + 1], vi[u]); } else atomicAdd(&s_acc[o], vr[u]);
                        }
                    }
                }
            }
            __syncthreads();
        }
        const int64_t m = (w1 - w0) * (CPLX ? 2 : 1);
        double *__restrict__ yw = y + w0 * (CPLX ? 2 : 1);
        for (int64_t i = tid; i < m; i += THREADS) yw[i] += s_acc[i];
        __syncthreads();
    }
}
extern "C" int lsk_window_rows(int cplx) { return cplx ? kWinRows / 2 : kWinRows; }
// y[d][key] += value for every packet of every stream of every source segment: dests (by value) names the destination vectors
// and the first block of each (ceil(ceil(count / rows) / wpb) blocks per destination); d_srcs is [dests.n][n_src].
extern "C" int lsk_window(int cplx, lsk_wdests const *dests, lsk_wsrc const *d_srcs, int n_src, int S, int wpb, void *stream) {
    if (dests->n < 1 || dests->n > LSK_MAX_SEGS || n_src < 1 || S < 1 || wpb < 1) { snprintf(g_err, sizeof(g_err), "lsk_window: bad arguments"); return -1; }
    const int64_t nb = dests->first_block[dests->n];
    if (nb <= 0) return 0;
    if (nb > ((int64_t)1 << 31) - 1) { snprintf(g_err, sizeof(g_err), "lsk_window: too many windows"); return -1; }
    // threads of a consumer block: 8 waves share a window's LDS where a window has many short runs (f64, >= 256 streams per window:
    // chain_28 x 8 3.21 -> 2.98 ms, chain_30 x 8 16.2 -> 14.7), 4 otherwise (c128 5.3 vs 6.1 ms, two partitions 3.1 vs 3.3).
    // LS_AMD_WINDOW_THREADS=256|512 overrides (A/B).
    char const *te = getenv("LS_AMD_WINDOW_THREADS");
    const int threads = te ? (atoi(te) >= 512 ? 512 : 256) : ((!cplx && (int64_t)n_src * S >= 256) ? 512 : 256);
    dim3 g((unsigned)nb), b((unsigned)threads);
    if (threads == 512) {
        if (cplx) hipLaunchKernelGGL((k_window<true, 512>), g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
        else hipLaunchKernelGGL((k_window<false, 512>), g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
    } else if (cplx) hipLaunchKernelGGL((k_window<true, 256>), g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
    else hipLaunchKernelGGL((k_window<false, 256>), g, b, 0, (hipStream_t)stream, *dests, d_srcs, n_src, S, wpb);
    LSK_LAUNCH_CHECK();
    return 0;
}


