// lsk_dev.hpp -- what every kernel translation unit shares: the error buffer and launch macros, grid helpers, and the device
// helpers of the hot path (owner hash, index look-ups, term evaluation, K4 orbit minima, lane utilities).  Everything in here is
// a template, `static`, `constexpr` or `__forceinline__`: each .hip file gets its own copy, nothing is exported.
//   k_runtime.hip  runtime shim (memory, streams, events)
//   k_rows.hip     K1 k_diag, k_direct, k_chain_t, k_pairs_t (one partition, unprojected bases)
//   k_packets.hip  producers k_tile / k_tile_wv / k_tile_st, consumers k_scatter* / k_window
//   k_pull.hip     projected bases, pull: k_tile_pull, static index table, k_pull_t / k_pull_gather / k_pull_count
//   k_plan.hip     plan-time and API helpers: norms, index tables, directories, scans, enumeration, layout converters, test hooks
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <type_traits>

#include "lsk.h"

extern thread_local char g_err[512]; // k_runtime.hip; lsk_last_error()

#define LSK_CHECK(expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) {                                                                  \
            snprintf(g_err, sizeof(g_err), "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr,    \
                     hipGetErrorString(e_));                                                     \
            return -1;                                                                           \
        }                                                                                        \
    } while (0)

#define LSK_LAUNCH_CHECK() LSK_CHECK(hipGetLastError())

// host-side scan shared by the plan-time code and the slot cache (k_plan.hip)
int lsk_internal_exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 8; // 256 CUs x 8 resident blocks: grid-stride beyond this

// Persistent (grid-stride) launches must not exceed what is co-resident, or the surplus blocks run as a
// second, mostly idle wave (measured: +33 % on the row kernel when 7 instead of 8 blocks fit per CU).
// resident_grid() asks the runtime once per kernel; note that on ROCm 7.2 the answer is one block per
// CU too high for 256-thread kernels with 81..96 SGPRs (MI355X_MICROARCH.md), so the hot kernels are
// kept at <= 80 SGPRs (asserted in tests/test_host_tables.py::test_hot_kernel_register_budget).
#include <map>
#include <mutex>
#include <vector>
static int g_num_cus = 0;
template <typename K>
static int resident_grid(K kernel, int64_t work_blocks, size_t dyn_lds = 0, int block = kBlock) {
    static std::map<std::pair<void const *, size_t>, int> cache;
    static std::mutex lock; // plans may be created / launched from several host threads (loop-back communicators)
    std::lock_guard<std::mutex> guard(lock);
    if (g_num_cus == 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0) g_num_cus = 256;
    }
    const std::pair<void const *, size_t> key((void const *)kernel, dyn_lds);
    auto it = cache.find(key);
    int per_cu;
    if (it == cache.end()) {
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, block, dyn_lds) != hipSuccess || nb < 1) nb = 1;
        if (nb > 8) nb = 8;
        cache[key] = nb;
        per_cu = nb;
    } else per_cu = it->second;
    int64_t g = (int64_t)g_num_cus * per_cu;
    if (work_blocks < g) g = work_blocks;
    if (g < 1) g = 1;
    return (int)g;
}

// Grid of the staged (tile) kernels: one block per tile up to the whole tile count.  They walk rows with a plain grid
// stride (no XCD tile lists), so nothing needs them to be persistent, and their 106 SGPRs put them where the occupancy
// API over-reports the resident blocks by one (MI355X_MICROARCH.md): a CUs x API-answer grid runs a straggler round
// with one block per CU (measured r2: k_tile_pull on chain_36_symm 34.8 -> 24.5 ms with the plain grid).
template <typename K>
static int tile_grid(K, int64_t work_blocks) {
    if (work_blocks < 1) work_blocks = 1;
    if (work_blocks > (int64_t)1 << 30) work_blocks = (int64_t)1 << 30;
    return (int)work_blocks;
}

// LS_AMD_ABLATE (lsk_basis.debug_ablate) switches stages of the tile / pull kernels off to price them --
// profiling builds only (make ABLATE=1): the shipped kernels carry none of these branches.
#ifndef LSK_ABLATE
#define LSK_ABLATE 0
#endif
constexpr bool kAblate = LSK_ABLATE != 0;

static inline int grid_for(int64_t n, int per_block = kBlock) {
    int64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxGrid) b = kMaxGrid;
    return (int)b;
}

// K5: splitmix64 finaliser (StatesEnumeration.chpl:122-127)
__device__ __forceinline__ uint64_t hash64_01(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return x;
}

// owner = hash % P with 32-bit arithmetic only (a true modulo: P = 3 must work, not just 2^k)
struct Owner {
    uint32_t P;
    uint32_t pmask;    // P - 1 when P is a power of two, else 0xffffffff
    uint32_t two32mod; // 2^32 mod P
};
static inline Owner make_owner(int P) {
    Owner o;
    o.P = (uint32_t)P;
    o.pmask = ((P & (P - 1)) == 0) ? (uint32_t)(P - 1) : 0xffffffffu;
    o.two32mod = (uint32_t)((1ULL << 32) % (uint64_t)P);
    return o;
}
__device__ __forceinline__ int owner_of(uint64_t s, Owner o) {
    if (o.P == 1) return 0; // one locale: nothing to hash (wave-uniform)
    uint64_t h = hash64_01(s);
    if (o.pmask != 0xffffffffu) return (int)((uint32_t)h & o.pmask);
    uint32_t hi = (uint32_t)(h >> 32), lo = (uint32_t)h;
    uint32_t r = hi % o.P;
    return (int)((r * o.two32mod + lo % o.P) % o.P);
}

// K8: relaxed, agent-scope f64 add.  unsafeAtomicAdd lowers to global_atomic_add_f64 on gfx950 for
// coarse-grained (hipMalloc) memory -- checked in the ISA dump, see DESIGN.md.
__device__ __forceinline__ void atomic_add_f64(double *p, double v) { unsafeAtomicAdd(p, v); }

template <bool REAL>
__device__ __forceinline__ void term_sum(lsk_term const *__restrict__ terms, int b, int e, uint64_t a,
                                         double &cr, double &ci) {
    cr = 0.0;
    ci = 0.0;
    for (int t = b; t < e; ++t) {
        lsk_term T = terms[t];
        if ((a & T.m) == T.r) {
            bool neg = __popcll(a & T.s) & 1;
            cr += neg ? -T.v_re : T.v_re;
            if (!REAL) ci += neg ? -T.v_im : T.v_im;
        }
    }
}

// K2: coefficient of one flip-mask group on state a
template <bool REAL>
__device__ __forceinline__ void group_coeff(lsk_group const &G, lsk_term const *__restrict__ off,
                                            uint64_t a, double &cr, double &ci) {
    if (G.fast == LSK_GROUP_EXCHANGE) {
        bool act = __popcll(a & G.x) == 1;
        cr = act ? G.v_re : 0.0;
        ci = (!REAL && act) ? G.v_im : 0.0;
        return;
    }
    // (directed pairs, LSK_GROUP_HOP_*: their terms are evaluated like any generic group's -- a branch of their own costs k_direct four
    // scalar registers and, at 82, a third of its speed; the classification only serves the run detection of the host)
    term_sum<REAL>(off, G.begin, G.end, a, cr, ci);
}

// combinadic rank among equal-popcount integers (ls_hs_fixed_hamming_state_to_index, FFI.chpl:165)
__device__ __forceinline__ int64_t rank_combinadic(uint64_t s, uint64_t const *binom) {
    int64_t idx = 0;
    int k = 1;
    while (s) {
        int p = __ffsll((unsigned long long)s) - 1;
        idx += (int64_t)binom[p * LSK_BINOM_K + k];
        ++k;
        s &= s - 1;
    }
    return idx;
}
__device__ __forceinline__ uint64_t unrank_combinadic(int64_t idx, int hamming, uint64_t const *binom) {
    uint64_t s = 0;
    int p = 63;
    for (int k = hamming; k >= 1; --k) {
        while (p > k - 1 && (int64_t)binom[p * LSK_BINOM_K + k] > idx) --p;
        // p is now the largest position with C(p, k) <= idx
        s |= 1ULL << p;
        idx -= (int64_t)binom[p * LSK_BINOM_K + k];
        --p;
    }
    return s;
}
// Gosper's hack (StatesEnumeration.chpl:31-34)
__device__ __forceinline__ uint64_t next_fixed_hamming(uint64_t v) {
    uint64_t t = v | (v - 1);
    return (t + 1) | (((~t & (t + 1)) - 1) >> (__ffsll((unsigned long long)v)));
}

// K7: prefix-bucket table + binary search in the ascending representatives
__device__ __forceinline__ int64_t search_index(lsk_index const &ix, uint64_t s) {
    uint64_t b = s >> ix.shift;
    uint32_t lo = ix.table[b], hi = ix.table[b + 1];
    const uint32_t end = hi;
    while (lo < hi) {
        uint32_t mid = lo + ((hi - lo) >> 1);
        if (ix.reps[mid] < s) lo = mid + 1; else hi = mid;
    }
    return (lo < end && ix.reps[lo] == s) ? (int64_t)lo : -1;
}

// Static index table (lsk_gtab, lsk.h; built by k_pull.hip): {state -> payload} in 16-byte buckets of two entries
constexpr uint64_t kGtEmpty = ~0ULL;
constexpr int kGtMaxDist = 255;
// an L-bit bijection (odd multiplications mod 2^L and xor-shifts): bucket and tag together identify the key
__host__ __device__ __forceinline__ uint64_t gt_mix(uint64_t k, int L) {
    const uint64_t m = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    const int s = (L + 1) >> 1;
    k = (k * 0x9E3779B97F4A7C15ULL) & m;
    k ^= k >> s;
    k = (k * 0xD6E8FEB86659FD93ULL) & m;
    k ^= k >> s;
    return k;
}
__host__ __device__ __forceinline__ void gt_split(lsk_gtab const &t, uint64_t key, uint64_t &bucket, uint32_t &tag) {
    const uint64_t h = gt_mix(key, t.L);
    bucket = h >> t.tbits;
    tag = (uint32_t)(h & ((1ULL << t.tbits) - 1));
}
// upper word of an entry: tag << 8 | displacement
__host__ __device__ __forceinline__ uint32_t gt_hi(uint32_t tag, int dist) { return (tag << 8) | (uint32_t)dist; }

// payload of `key`, or 0xffffffff; `first` is the home bucket when the caller has already loaded it
__device__ __forceinline__ uint32_t gt_resolve(lsk_gtab const &t, uint64_t const *__restrict__ tab, uint64_t b, uint32_t tag,
                                               ulonglong2 cur) {
    const uint64_t bmask = (1ULL << t.bbits) - 1;
    for (int d = 0;; ++d) {
        const uint32_t want = gt_hi(tag, d);
        if ((uint32_t)(cur.x >> 32) == want && cur.x != kGtEmpty) return (uint32_t)cur.x;
        if ((uint32_t)(cur.y >> 32) == want && cur.y != kGtEmpty) return (uint32_t)cur.y;
        if (cur.x == kGtEmpty || cur.y == kGtEmpty || d == kGtMaxDist) return 0xffffffffu; // inserts never skip an empty slot
        b = (b + 1) & bmask;
        cur = *(ulonglong2 const *)(tab + 2 * b);
    }
}

// index of state s in the partition the table was built from, or -1 (payload = index)
__device__ __forceinline__ int64_t gtab_index(lsk_gtab const &t, uint64_t s) {
    if (t.L < 64 && (s >> t.L) != 0) return -1;
    uint64_t b;
    uint32_t tag;
    gt_split(t, s, b, tag);
    const uint32_t pay = gt_resolve(t, t.entries, b, tag, *(ulonglong2 const *)(t.entries + 2 * b));
    return pay == 0xffffffffu ? -1 : (int64_t)pay;
}

// Rank directory (lsk_rankdir, lsk.h): LDS copy of the binomials a rank needs -- C(p, k), p < sites, k <= weight -- and the look-up
__device__ __forceinline__ int rankdir_lds_entries(lsk_index const &ix) { return ix.dir ? ix.dir_sites * (ix.dir_weight + 1) : 0; }
__device__ __forceinline__ void rankdir_load(lsk_index const &ix, uint64_t *s_db) { // (the caller synchronises the block)
    const int kc = ix.dir_weight + 1;
    for (int i = threadIdx.x; i < ix.dir_sites * kc; i += blockDim.x) s_db[i] = ix.binom[(i / kc) * LSK_BINOM_K + (i % kc)];
}
__device__ __forceinline__ int64_t rankdir_index(lsk_index const &ix, uint64_t s, uint64_t const *s_db) {
    const int kc = ix.dir_weight + 1;
    if (__popcll(s) != ix.dir_weight || (ix.dir_sites < 64 && (s >> ix.dir_sites) != 0)) return -1;
    uint64_t g = 0;
    int k = 1;
    while (s) {
        const int p = __ffsll((unsigned long long)s) - 1;
        g += s_db[p * kc + k];
        ++k;
        s &= s - 1;
    }
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(ix.dir + (g >> 6));
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}

// All-destinations directory (lsk_gdir, lsk.h): index of state s inside the block of partition d, or -1 (s_db as above: the
// binomials C(p, k), p < sites, k <= weight)
__device__ __forceinline__ int64_t gdir_index(lsk_gdir const &gd, uint64_t s, int d, uint64_t const *s_db) {
    const int kc = gd.weight + 1;
    if (__popcll(s) != gd.weight || (gd.sites < 64 && (s >> gd.sites) != 0)) return -1;
    uint64_t g = 0;
    int k = 1;
    while (s) {
        const int p = __ffsll((unsigned long long)s) - 1;
        g += s_db[p * kc + k];
        ++k;
        s &= s - 1;
    }
    if ((int64_t)g >= gd.n_ranks) return -1;
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(gd.entries + (g >> 6) * (uint64_t)gd.P + (uint64_t)d);
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}
// the same look-up for a state whose global rank g is already known
constexpr uint32_t kNoRank = 0xffffffffu;
__device__ __forceinline__ int64_t gdir_index_of_rank(lsk_gdir const &gd, uint64_t g, int d) {
    if ((int64_t)g >= gd.n_ranks) return -1;
    const ulonglong2 e = *reinterpret_cast<ulonglong2 const *>(gd.entries + (g >> 6) * (uint64_t)gd.P + (uint64_t)d);
    const uint64_t bit = 1ULL << (g & 63);
    if (!(e.x & bit)) return -1;
    return (int64_t)(uint32_t)e.y + __popcll(e.x & (bit - 1));
}
__device__ __forceinline__ void gdir_load(lsk_gdir const &gd, uint64_t const *__restrict__ g_binom, uint64_t *s_db) { // (the caller synchronises)
    const int kc = gd.weight + 1;
    for (int i = threadIdx.x; i < gd.sites * kc; i += blockDim.x) s_db[i] = g_binom[(i / kc) * LSK_BINOM_K + (i % kc)];
}

// Open-addressing hash table {representative -> x * norm(rep)} used by the staged pull kernel.  The
// uncoalesced per-lane loads of a search (table + ~5 probes + value = 8 line requests per packet) were what
// bounded k_tile_pull (L1/TA issue: one line per lane per cycle); a hit in the home slot costs ONE 16-byte
// request.  Keys are inserted once per plan (linear probing, load factor <= 0.5), values are refreshed
// every matvec through slot_of[i].  Entry = {key, re[, im, pad]}: 2 (f64) or 4 (c128) u64 words.
constexpr uint64_t kHashEmpty = ~0ULL;
__device__ __forceinline__ uint64_t hash_slot(uint64_t key, int bits) {
    return (key * 0x9E3779B97F4A7C15ULL) >> (64 - bits);
}
template <int ES>
__device__ __forceinline__ bool hash_lookup(uint64_t const *__restrict__ tab, int bits, uint64_t key, double &vr,
                                            double &vi) {
    const uint64_t mask = (1ULL << bits) - 1;
    uint64_t slot = hash_slot(key, bits);
    for (;;) {
        if (ES == 2) {
            const ulonglong2 e = *(ulonglong2 const *)(tab + slot * 2);
            if (e.x == key) { vr = __longlong_as_double((long long)e.y); vi = 0.0; return true; }
            if (e.x == kHashEmpty) return false;
        } else {
            const ulonglong2 e = *(ulonglong2 const *)(tab + slot * 4);
            if (e.x == key) {
                vr = __longlong_as_double((long long)e.y);
                vi = __longlong_as_double((long long)tab[slot * 4 + 2]);
                return true;
            }
            if (e.x == kHashEmpty) return false;
        }
        slot = (slot + 1) & mask;
    }
}

// one symmetry-group element applied to a state
__device__ __forceinline__ uint64_t delta_swap(uint64_t x, uint64_t m, int d) {
    uint64_t t = ((x >> d) ^ x) & m;
    return x ^ t ^ (t << d);
}
__device__ __forceinline__ uint64_t apply_elem(lsk_group_elem const &e, uint64_t x, int L, uint64_t mask) {
    if (e.kind == LSK_ELEM_BENES) {
        if (e.masks[0]) x = delta_swap(x, e.masks[0], 32);
        if (e.masks[1]) x = delta_swap(x, e.masks[1], 16);
        if (e.masks[2]) x = delta_swap(x, e.masks[2], 8);
        if (e.masks[3]) x = delta_swap(x, e.masks[3], 4);
        if (e.masks[4]) x = delta_swap(x, e.masks[4], 2);
        if (e.masks[5]) x = delta_swap(x, e.masks[5], 1);
        if (e.masks[6]) x = delta_swap(x, e.masks[6], 2);
        if (e.masks[7]) x = delta_swap(x, e.masks[7], 4);
        if (e.masks[8]) x = delta_swap(x, e.masks[8], 8);
        if (e.masks[9]) x = delta_swap(x, e.masks[9], 16);
        if (e.masks[10]) x = delta_swap(x, e.masks[10], 32);
        return x;
    }
    if (e.kind == LSK_ELEM_REVROT) x = __brevll(x) >> (64 - L);
    int k = e.k;
    if (k == 0) return x;
    return ((x >> k) | (x << (L - k))) & mask;
}

// 32-bit variant of apply_elem for bases with <= 32 sites: permutations only move the low 32 bits, so
// the distance-32 Benes stages are empty and the remaining masks live in the low words.
__device__ __forceinline__ uint32_t delta_swap32(uint32_t x, uint32_t m, int d) {
    uint32_t t = ((x >> d) ^ x) & m;
    return x ^ t ^ (t << d);
}
__device__ __forceinline__ uint32_t apply_elem32(lsk_group_elem const &e, uint32_t x, int L, uint32_t mask) {
    if (e.kind == LSK_ELEM_BENES) {
        // only the low words of the masks are read (4-byte scalar loads: half the scalar registers of the 8-byte ones)
        uint32_t const *const m32 = reinterpret_cast<uint32_t const *>(e.masks);
        if (m32[2]) x = delta_swap32(x, m32[2], 16);
        if (m32[4]) x = delta_swap32(x, m32[4], 8);
        if (m32[6]) x = delta_swap32(x, m32[6], 4);
        if (m32[8]) x = delta_swap32(x, m32[8], 2);
        if (m32[10]) x = delta_swap32(x, m32[10], 1);
        if (m32[12]) x = delta_swap32(x, m32[12], 2);
        if (m32[14]) x = delta_swap32(x, m32[14], 4);
        if (m32[16]) x = delta_swap32(x, m32[16], 8);
        if (m32[18]) x = delta_swap32(x, m32[18], 16);
        return x;
    }
    if (e.kind == LSK_ELEM_REVROT) x = __brev(x) >> (32 - L);
    int k = e.k;
    if (k == 0) return x;
    return ((x >> k) | (x << (L - k))) & mask;
}
template <typename W> __device__ __forceinline__ W apply_elem_w(lsk_group_elem const &e, W x, int L, W mask);
template <> __device__ __forceinline__ uint64_t apply_elem_w<uint64_t>(lsk_group_elem const &e, uint64_t x, int L, uint64_t mask) {
    return apply_elem(e, x, L, mask);
}
template <> __device__ __forceinline__ uint32_t apply_elem_w<uint32_t>(lsk_group_elem const &e, uint32_t x, int L, uint32_t mask) {
    return apply_elem32(e, x, L, mask);
}

// K4: ls_hs_state_info -- orbit minimum, conj(character) of a minimising element, stabiliser sum.
// One pass over the permutations; the optional global spin flip is folded in by canonicalising every
// image to "top site bit clear" (t ^ mask < t iff the top bit of t is set), which halves the work.
// The stabiliser sum needs no `g(a) == a` tests: the elements that map a onto its representative
// form the coset g0 Stab(a), so  sum_{s in Stab(a)} chi(s) = conj(chi(g0)) * sum_{g: g(a) = rep} chi(g),
// i.e. it is accumulated over the ties with the running minimum.
// PM1: every character (and the inversion character) is +-1 -> integer accumulation.
template <typename W, bool PM1>
__device__ __forceinline__ void state_info_w(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                             W a, W &rep, double &chr, double &chi, double &stab) {
    W best = ~(W)0;
    int info = 0;
    int si = 0;
    double sr = 0.0, sim = 0.0;
    const int inv = bs.spin_inversion;
    const int L = bs.number_sites;
    const W mask = (W)bs.site_mask;
    for (int g = 0; g < bs.n_elems; ++g) {
        lsk_group_elem const &e = elems[g];
        W t = apply_elem_w<W>(e, a, L, mask);
        int top = 0;
        if (inv != 0) {
            top = (int)((t >> (L - 1)) & 1);
            t = top ? (W)(t ^ mask) : t;
        }
        const bool less = t < best, eq = t == best;
        if (PM1) {
            int ch = (int)e.ch_re;
            ch = top ? ch * inv : ch;
            si = less ? ch : (eq ? si + ch : si);
        } else {
            double cr = e.ch_re, ci = e.ch_im;
            if (top) { cr *= (double)inv; ci *= (double)inv; }
            sr = less ? cr : (eq ? sr + cr : sr);
            sim = less ? ci : (eq ? sim + ci : sim);
        }
        best = less ? t : best;
        info = less ? (2 * g + top) : info;
    }
    rep = best;
    lsk_group_elem const &e0 = elems[info >> 1];
    double c0r = e0.ch_re, c0i = e0.ch_im;
    if (info & 1) { c0r *= (double)inv; c0i *= (double)inv; }
    chr = c0r;
    chi = -c0i;
    if (PM1) stab = c0r * (double)si;
    else stab = c0r * sr + c0i * sim; // Re(conj(chi0) * S)
}
// K4, trivial sector, cyclic / dihedral group (mode 3): orbit minimum WITHOUT visiting every rotation.
// The smallest rotation (as an integer, site L-1 = MSB) starts with the longest cyclic run of zeros, so
//   1. R <- start positions (MSB ends) of the longest zero runs: R_1 = z, R_{j+1} = R_j & rotl(z, j)
//      with z = ~a; the loop runs (longest run) times -- ~5-8 on typical states instead of L;
//   2. only those start positions are candidates (usually 1-2): rotate each to the top and take the min;
//   3. reflections: the runs of rev(a) are the mirrored runs of a, so their candidates come from R
//      by one rotate + bit-reverse, no second search;
//   4. global spin flip: the flipped images start with a run of *ones* of a, so only the family whose
//      longest run is longer (both on a tie) can contain the minimum.
// (host-callable as well: tests/test_host_tables.py checks it against the brute-force orbit minimum through
// lsk_test_rep_trivial_dihedral)
__host__ __device__ __forceinline__ int k4_ctz32(uint32_t v) { return __builtin_ctz(v); }
__host__ __device__ __forceinline__ int k4_ctz64(uint64_t v) { return __builtin_ctzll(v); }
__host__ __device__ __forceinline__ uint32_t k4_brev32(uint32_t v) { return __builtin_bitreverse32(v); }
__host__ __device__ __forceinline__ uint64_t k4_brev64(uint64_t v) { return __builtin_bitreverse64(v); }
template <typename W>
__host__ __device__ __forceinline__ W rotl_sites(W x, int s, int L, W mask) {
    return s == 0 ? x : (W)(((x << s) | (x >> (L - s))) & mask);
}
// rotl_sites without the final mask: for the run searches, where the result only meets words inside the mask
template <typename W>
__host__ __device__ __forceinline__ W rotl_raw(W x, int s, int L) {
    return s == 0 ? x : (W)((x << s) | (x >> (L - s)));
}
template <typename W>
__host__ __device__ __forceinline__ W rev_sites(W x, int L) {
    if (sizeof(W) == 4) return (W)(k4_brev32((uint32_t)x) >> (32 - L));
    return (W)(k4_brev64((uint64_t)x) >> (64 - L));
}
// Start positions (MSB ends) of the longest cyclic runs of set bits of z, and their length.  The run length is found by
// doubling and refining instead of one rotation per unit of length: R2 = z & rot(z,1), R4 = R2 & rot(R2,2), R8 = R4 &
// rot(R4,4) hold the starts of runs >= 2, 4, 8; below 8 two more rotations settle the exact length (R6 = R4 & rot(R2,4), then
// one step of 1) -- five rotations, no data-dependent trip count, where the step-by-step loop makes every lane of a wave wait
// for the longest run among 64 packets (8.5 steps on half-filled 36-site states against 5.3 on average).  Runs >= 8
// continue step by step from R8.
template <typename W>
__host__ __device__ __forceinline__ W longest_runs(W z, int L, W mask, int &len) {
    if (z == 0) { len = 0; return (W)1; }          // no zero site at all: every rotation is the same word
    if (z == mask) { len = L; return (W)1; }        // all sites zero
#ifdef LSK_K4_STEPWISE
    const bool stepwise = true; // A/B builds: the round-2 loop, one rotation per unit of run length
#else
    const bool stepwise = false;
#endif
    if (stepwise || L < 9) { // tiny rings: the doubling steps would wrap around the ring
        W R = z;
        int s = 1;
        for (;;) {
            const W T = R & rotl_raw<W>(z, s, L);
            if (T == 0) break;
            R = T;
            ++s;
        }
        len = s;
        return R;
    }
    const W R2 = z & rotl_raw<W>(z, 1, L);
    const W R4 = R2 & rotl_raw<W>(R2, 2, L);
    const W R8 = R4 & rotl_raw<W>(R4, 4, L);
    W R;
    int s;
    if (R8 != 0) { // rare per packet; the tail of the old loop
        R = R8;
        s = 8;
        for (;;) {
            const W T = R & rotl_raw<W>(z, s, L);
            if (T == 0 || s + 1 >= L) break;
            R = T;
            ++s;
        }
        len = s;
        return R;
    }
    const bool c4 = R4 != 0, c2 = R2 != 0;
    R = c4 ? R4 : (c2 ? R2 : z);
    s = c4 ? 4 : (c2 ? 2 : 1);
    const W R6 = R4 & rotl_raw<W>(R2, 4, L);
    if (R6 != 0) { R = R6; s = 6; }
    const W T = R & rotl_raw<W>(z, s, L);
    if (T != 0) { R = T; ++s; }
    len = s;
    return R;
}
// minimum over the rotations that put the MSB end of a longest run on top and, with reflections, over the mirrored
// words that start with the same run: the run whose MSB end is p has the candidate c = rotl(word, L-1-p); mirrored, the
// run leads again when its LSB end is on top of rev(word), and that word is rev(rotl(c, len)) -- one more rotation of c
// and a bit reversal, inside the same loop (round 2 ran a second loop over rev(word) and the mirrored end positions).
template <typename W>
__host__ __device__ __forceinline__ W family_min(W word, W R, int len, int L, W mask, bool reflect, W best) {
    const int lr = len >= L ? 0 : len; // len == L only for the all-equal words, whose rotations coincide
    while (R) {
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)R) : k4_ctz64((uint64_t)R);
        R &= R - 1;
        const W c = rotl_sites<W>(word, L - 1 - p, L, mask); // site p becomes the top site
        best = c < best ? c : best;
        if (reflect) {
            const W m = rev_sites<W>(rotl_sites<W>(c, lr, L, mask), L);
            best = m < best ? m : best;
        }
    }
    return best;
}
// (Both families through ONE loop -- a lane walking the starts of the first, then of the second, with the second's
// candidates as complements of the rotations of a -- measured slower: 15.7 vs 14.4 ms for the packets of chain_36_symm,
// scripts/k4_rate.py; the selects per iteration cost more than the shorter trip count saves.)
template <typename W>
__host__ __device__ __forceinline__ W rep_trivial_dihedral(W a, int L, W mask, bool inv, bool reflect) {
    if (!inv) {
        int len0;
        const W R0 = longest_runs<W>((W)(~a & mask), L, mask, len0); // zero runs of a
        return family_min<W>(a, R0, len0, L, mask, reflect, ~(W)0);
    }
    // With the global spin flip the minimum starts with the longest run of EQUAL bits of a, zeros or ones (a run of ones
    // leads the flipped word).  Those runs are the zero runs of the transition word t = a ^ rotl(a, 1) (t_p = 1 where
    // a changes between p-1 and p), one site shorter and with the same MSB ends -- so ONE run search finds the longest
    // runs of both kinds, ties between the kinds included, and ONE loop walks their starts; the kind of a run is the top bit
    // of the rotated word.  (Until late round 3: a search for zeros, one for ones, and two candidate passes that nearly
    // every wave entered both of; K4 alone 7.0 ms for the packets of chain_36_symm.)
    const W zt = (W)(~(a ^ rotl_sites<W>(a, 1, L, mask)) & mask);
    W R;
    int ell; // length of the longest run of equal bits
    if (zt == 0) { R = mask; ell = 1; }            // a alternates: every site starts a run of one
    else if (zt == mask) { R = (W)1; ell = L; }    // all sites equal
    else { int lt; R = longest_runs<W>(zt, L, mask, lt); ell = lt + 1; }
    const int lr = ell >= L ? 0 : ell;
    W best = ~(W)0;
    while (R) {
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)R) : k4_ctz64((uint64_t)R);
        R &= R - 1;
        const W r = rotl_sites<W>(a, L - 1 - p, L, mask); // site p becomes the top site
        const W flip = ((r >> (L - 1)) & 1) ? mask : (W)0;  // a run of ones: its flipped image competes
        const W c = r ^ flip;
        best = c < best ? c : best;
        if (reflect) {
            const W m = rev_sites<W>(rotl_sites<W>(r, lr, L, mask), L) ^ flip;
            best = m < best ? m : best;
        }
    }
    return best;
}


// Minimum of a word over the tw * th translations of a tw x th torus (site = y tw + x) and, with `inv`, of its complement:
// the rows are the digits of the word, so the minimum puts on top the smallest value ANY row takes under ANY rotation inside
// the row.  rowtab[r] (made by the host, lsk_torus_rowtab) holds for a row value r: bits 0-7 the minimum over its rotations,
// 8-15 the set of rotation amounts that reach it, 16-23 the maximum, 24-31 the amounts that reach that (the rows of the
// complemented word are the complements, so its row minima are the complements of the maxima).  One table load per row, then
// only the (row, amount) pairs that put the overall row minimum on top are built and compared -- 1-2 of the 2 tw th
// candidates on a half-filled 6 x 6 lattice -- and none at all when an earlier coset already has a smaller top row.
// (tw <= 8.  Until mid round 4: tw * th steps of "rotate the rows by one site / the word by one row" per coset.)
template <typename W>
__host__ __device__ __forceinline__ W torus_min(W v, int L, int tw, W mask, W col0, bool inv, uint32_t const *__restrict__ rowtab, W best) {
    const int th = L / tw;
    const uint32_t rmask = (1u << tw) - 1u;
    uint32_t mstar = 0xffffffffu;
    for (int k = 0; k < th; ++k) {
        const uint32_t e = rowtab[(uint32_t)(v >> (k * tw)) & rmask];
        uint32_t m = e & 0xffu;
        if (inv) { const uint32_t m2 = ~(e >> 16) & rmask; m = m2 < m ? m2 : m; }
        mstar = m < mstar ? m : mstar;
    }
    if ((W)mstar > (W)(best >> (L - tw))) return best; // (best == ~0 at the start: never true)
    // the candidates as site masks: bit k tw + i of cand[fam] = "rotate the rows by i, then row k to the top".  Collected
    // first and then popped ONE PER LANE AND ITERATION: the lanes of a wave hold different words, and a loop over
    // (row, family) with the construction inside its body made every wave run that body for nearly all 2 th combinations
    // (some lane always matches) -- 62.8 ms per matvec on heisenberg_square_6x6 -- instead of for the 1-3 its lanes need.
    const W nv = (W)(~v & mask);
    W cand0 = 0, cand1 = 0;
    for (int k = 0; k < th; ++k) {
        const uint32_t e = rowtab[(uint32_t)(v >> (k * tw)) & rmask];
        if ((e & 0xffu) == mstar) cand0 |= (W)((e >> 8) & 0xffu) << (k * tw);
        if (inv && (~(e >> 16) & rmask) == mstar) cand1 |= (W)(e >> 24) << (k * tw);
    }
    const uint32_t inv_tw = 65536u / (uint32_t)tw + 1u; // p / tw for p < 64, tw <= 8
    while (cand0 | cand1) {
        const bool first = cand0 != 0;
        const W cm = first ? cand0 : cand1;
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)cm) : k4_ctz64((uint64_t)cm);
        if (first) cand0 &= cand0 - 1; else cand1 &= cand1 - 1;
        const int k = (int)(((uint32_t)p * inv_tw) >> 16), i = p - k * tw;
        const W word = first ? v : nv;
        const W lo = (W)(col0 * (W)((1u << i) - 1u)); // columns 0 .. i-1 of every row (no carries: 2^i - 1 < 2^tw)
        W c = (W)((((W)(word << i)) & (W)~lo & mask) | ((W)(word >> (tw - i)) & lo)); // (i == 0: lo == 0, the first term is the word)
        c = rotl_sites<W>(c, tw * (th - 1 - k), L, mask);
        best = c < best ? c : best;
    }
    return best;
}
// ---- K4 mode 5: the point group of a rectangular / square torus, factorised -------------------------------------
// The cosets T g of such a lattice group are (modulo translations) the elements of D2 = {1, r, o, r o} -- r reverses every row
// (x -> tw-1-x), o reverses the order of the rows (y -> th-1-y), r o is the reversal of the whole word -- and, on a square
// torus, those times the transpose: D4.  Mode 4 sends the word through ONE compiled 11-stage network PER coset (8 x ~110 VALU
// instructions on 64-bit words) and runs torus_min on each image (8 x ~350).  Here
//   * only the transpose is a network; r is floor(tw / 2) delta swaps, r o one bit reversal, o = (r o) r;
//   * the four images of a base word are made of TWO row alphabets: {rows of v} for v and o(v), {reversed rows} for r(v) and
//     r o (v); rowtab2[row] carries the torus_min fields of the row (low half) AND of the reversed row (high half), so ONE pass
//     over the rows finds the smallest top row any of the four images can reach (and of their complements under the spin flip);
//   * only the alphabets that reach it build candidates, for their two words each.
// `present`: bit 0 = identity, 1 = r, 2 = o, 3 = r o (the images that belong to the group).
template <typename W>
__host__ __device__ __forceinline__ W rowrev_w(W v, int tw, W col0) {
    for (int j = 0; 2 * j + 1 < tw; ++j) { // swap columns j and tw-1-j of every row
        const int d = tw - 1 - 2 * j;
        const W t = (W)(((v >> d) ^ v) & (W)(col0 << j));
        v ^= (W)(t | (W)(t << d));
    }
    return v;
}
template <typename W>
__host__ __device__ __forceinline__ W torus_candidate(W word, int i, int kpos, int L, int tw, int th, W mask, W col0) {
    const W lo = (W)((W)(col0 << i) - col0); // columns 0 .. i-1 of every row: 2^i - 1 per row, no borrow between rows (i < tw)
    const W c = (W)((((W)(word << i)) & (W)~lo & mask) | ((W)(word >> (tw - i)) & lo)); // every row rotated by i
    return rotl_sites<W>(c, tw * (th - 1 - kpos), L, mask);                              // row kpos to the top
}
template <typename W>
__host__ __device__ __forceinline__ W torus_min_d2(W v, int L, int tw, W mask, W col0, bool inv, int present,
                                                   uint64_t const *__restrict__ rowtab2, W best) {
    const int th = L / tw;
    const uint32_t rmask = (1u << tw) - 1u;
    const bool needA = present & 5, needB = present & 10;
    // pass 1: the smallest top row of each alphabet (A: rows as they are, B: rows reversed; 1: of the complemented word)
    uint32_t mA0 = 0xffffffffu, mA1 = 0xffffffffu, mB0 = 0xffffffffu, mB1 = 0xffffffffu;
    for (int k = 0; k < th; ++k) {
        const uint64_t e = rowtab2[(uint32_t)(v >> (k * tw)) & rmask];
        const uint32_t ea = (uint32_t)e, eb = (uint32_t)(e >> 32);
        uint32_t m = ea & 0xffu; mA0 = m < mA0 ? m : mA0;
        m = ~(ea >> 16) & rmask;  mA1 = m < mA1 ? m : mA1;
        m = eb & 0xffu;           mB0 = m < mB0 ? m : mB0;
        m = ~(eb >> 16) & rmask;  mB1 = m < mB1 ? m : mB1;
    }
    if (!needA) mA0 = mA1 = 0xffffffffu;
    if (!needB) mB0 = mB1 = 0xffffffffu;
    if (!inv) mA1 = mB1 = 0xffffffffu;
    uint32_t mstar = mA0 < mA1 ? mA0 : mA1;
    mstar = mB0 < mstar ? mB0 : mstar;
    mstar = mB1 < mstar ? mB1 : mstar;
    if ((W)mstar > (W)(best >> (L - tw))) return best;
    // pass 2, ONE loop over the rows for all four alphabets (a loop per alphabet made every wave walk the rows four times: some
    // lane always needs each of them): bit k tw + i of a mask = "rotate the rows by i, then row k (base order) to the top"
    W cA0 = 0, cA1 = 0, cB0 = 0, cB1 = 0;
    for (int k = 0; k < th; ++k) {
        const uint64_t e = rowtab2[(uint32_t)(v >> (k * tw)) & rmask];
        const uint32_t ea = (uint32_t)e, eb = (uint32_t)(e >> 32);
        const int sh = k * tw;
        if ((ea & 0xffu) == mstar && mA0 == mstar) cA0 |= (W)((ea >> 8) & 0xffu) << sh;
        if ((~(ea >> 16) & rmask) == mstar && mA1 == mstar) cA1 |= (W)(ea >> 24) << sh;
        if ((eb & 0xffu) == mstar && mB0 == mstar) cB0 |= (W)((eb >> 8) & 0xffu) << sh;
        if ((~(eb >> 16) & rmask) == mstar && mB1 == mstar) cB1 |= (W)(eb >> 24) << sh;
    }
    // the words: rows in the base order (v | r(v)) and in reversed order (o(v) = rev(r(v)) | r o (v) = rev(v))
    const W rv = rowrev_w<W>(v, tw, col0);
    const W brv = rev_sites<W>(rv, L), bv = rev_sites<W>(v, L);
    const bool fwdA = present & 1, fwdB = present & 2, bwdA = present & 4, bwdB = present & 8;
    const uint32_t inv_tw = 65536u / (uint32_t)tw + 1u; // p / tw for p < 64, tw <= 8
    // ONE pop loop over the candidates of all alphabets (its trip count is the largest number of candidates a lane of the wave holds)
    while (cA0 | cA1 | cB0 | cB1) {
        const bool a0 = cA0 != 0, a1 = !a0 && cA1 != 0, b0 = !a0 && !a1 && cB0 != 0;
        const bool isB = !a0 && !a1, cpl = a1 || (isB && !b0);
        const W cm = a0 ? cA0 : (a1 ? cA1 : (b0 ? cB0 : cB1));
        const int p = sizeof(W) == 4 ? k4_ctz32((uint32_t)cm) : k4_ctz64((uint64_t)cm);
        const W rest = (W)(cm & (cm - 1));
        if (a0) cA0 = rest; else if (a1) cA1 = rest; else if (b0) cB0 = rest; else cB1 = rest;
        const int k = (int)(((uint32_t)p * inv_tw) >> 16), i = p - k * tw;
        const W flip = cpl ? mask : (W)0;
        if (isB ? fwdB : fwdA) { const W c = torus_candidate<W>((W)((isB ? rv : v) ^ flip), i, k, L, tw, th, mask, col0); best = c < best ? c : best; }
        if (isB ? bwdB : bwdA) { const W c = torus_candidate<W>((W)((isB ? bv : brv) ^ flip), i, th - 1 - k, L, tw, th, mask, col0); best = c < best ? c : best; }
    }
    return best;
}


// K4, trivial sector: only the orbit minimum.  mode 2 generates the L rotations incrementally
// (rotr by one site = shift + move bit 0 to bit L-1) for a and, with reflections, for rev(a).
template <typename W>
__device__ __forceinline__ W rep_trivial(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems, W a) {
    const int L = bs.number_sites;
    const W mask = (W)bs.site_mask;
    const bool inv = bs.spin_inversion != 0;
    W best = ~(W)0;
    if (bs.k4_mode == 3) return rep_trivial_dihedral<W>(a, L, mask, inv, bs.reflect != 0);
    if (bs.k4_mode == 5) { // D2 / D4 point group of a torus, factorised: one network (the transpose) instead of one per coset
        const W col0 = (W)bs.tcol0;
        best = torus_min_d2<W>(a, L, bs.tw, mask, col0, inv, bs.d4_mask & 15, bs.trow2, best);
        if (bs.d4_mask >> 4) best = torus_min_d2<W>(apply_elem_w<W>(bs.cosets[0], a, L, mask), L, bs.tw, mask, col0, inv, bs.d4_mask >> 4, bs.trow2, best);
        return best;
    }
    if (bs.k4_mode == 4) {
        // translations of a tw x th torus as a subgroup: one compiled network per coset representative (the point group), then
        // tw * th cheap steps -- rotate every row by one site; after tw of them the word is back, rotate it by one row
        const int tw = bs.tw, th = L / tw;
        const W col0 = (W)bs.tcol0, ncol0 = (W)(~col0 & mask);
        if (bs.trow) { // tw <= 8: the row table picks the few translations that can be minimal (torus_min)
            for (int r = 0; r < bs.n_cosets; ++r)
                best = torus_min<W>(apply_elem_w<W>(bs.cosets[r], a, L, mask), L, tw, mask, col0, inv, bs.trow, best);
            return best;
        }
        for (int r = 0; r < bs.n_cosets; ++r) {
            W b = apply_elem_w<W>(bs.cosets[r], a, L, mask);
            for (int j = 0; j < th; ++j) {
                for (int i = 0; i < tw; ++i) {
                    W c = b;
                    if (inv) c = ((b >> (L - 1)) & 1) ? (W)(b ^ mask) : b;
                    best = c < best ? c : best;
                    b = (W)(((W)(b << 1) & ncol0) | ((W)(b >> (tw - 1)) & col0));
                }
                b = rotl_sites<W>(b, tw, L, mask);
            }
        }
        return best;
    }
    if (bs.k4_mode == 2) {
        W r = a;
        for (int pass = 0; pass <= bs.reflect; ++pass) {
#pragma unroll 4
            for (int k = 0; k < L; ++k) {
                W c = r;
                if (inv) c = ((r >> (L - 1)) & 1) ? (W)(r ^ mask) : r;
                best = c < best ? c : best;
                r = (W)(r >> 1) | (W)((r & 1) << (L - 1));
            }
            if (sizeof(W) == 4) r = (W)(__brev((uint32_t)a) >> (32 - L));
            else r = (W)(__brevll((uint64_t)a) >> (64 - L));
        }
        return best;
    }
    for (int g = 0; g < bs.n_elems; ++g) {
        W t = apply_elem_w<W>(elems[g], a, L, mask);
        if (inv) t = ((t >> (L - 1)) & 1) ? (W)(t ^ mask) : t;
        best = t < best ? t : best;
    }
    return best;
}

__device__ __forceinline__ void state_info(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                           uint64_t a, uint64_t &rep, double &chr, double &chi, double &stab) {
    if (bs.chars_pm1) state_info_w<uint64_t, true>(bs, elems, a, rep, chr, chi, stab);
    else state_info_w<uint64_t, false>(bs, elems, a, rep, chr, chi, stab);
}

// ls_hs_is_representative with early exit: false as soon as some element maps below a
__device__ __forceinline__ bool is_representative(lsk_basis const &bs, lsk_group_elem const *__restrict__ elems,
                                                  uint64_t a) {
    double st = 0.0;
    const int inv = bs.spin_inversion;
    for (int g = 0; g < bs.n_elems; ++g) {
        lsk_group_elem const &e = elems[g];
        uint64_t t = apply_elem(e, a, bs.number_sites, bs.site_mask);
        if (t < a) return false;
        if (t == a) st += e.ch_re;
        if (inv != 0) {
            uint64_t tf = t ^ bs.site_mask;
            if (tf < a) return false;
            if (tf == a) st += e.ch_re * (double)inv;
        }
    }
    return st * bs.inv_order > 1e-12;
}

__device__ __forceinline__ void load_binom(uint64_t *s_binom, uint64_t const *__restrict__ g_binom) {
    for (int i = threadIdx.x; i < 64 * LSK_BINOM_K; i += blockDim.x) s_binom[i] = g_binom[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// word-width helpers: bases with <= 32 sites run the row kernels on 32-bit states (half the VALU work)
// ---------------------------------------------------------------------------------------------
template <typename W> struct WordTraits;
template <> struct WordTraits<uint32_t> {
    typedef uint32_t binom_t;
    static __device__ __forceinline__ int popc(uint32_t v) { return __popc(v); }
    static __device__ __forceinline__ int ctz(uint32_t v) { return __ffs((int)v) - 1; }
};
template <> struct WordTraits<uint64_t> {
    typedef uint64_t binom_t;
    static __device__ __forceinline__ int popc(uint64_t v) { return __popcll(v); }
    static __device__ __forceinline__ int ctz(uint64_t v) { return __ffsll((unsigned long long)v) - 1; }
};
template <typename W, typename BT>
__device__ __forceinline__ int64_t rank_combinadic_w(W s, BT const *binom) {
    int64_t idx = 0;
    int k = 1;
    while (s) {
        int p = WordTraits<W>::ctz(s);
        idx += (int64_t)binom[p * LSK_BINOM_K + k];
        ++k;
        s &= s - 1;
    }
    return idx;
}

// diagonal coefficient with the zz-run shortcut: sum_b v (-1)^{[bits b, b+1 differ]} = v (cnt - 2 #differ)
template <typename W, bool REAL>
__device__ __forceinline__ void diag_coeff(lsk_runs const &runs, int n_diag, lsk_term const *__restrict__ diag,
                                           W a, double &dr, double &di) {
    dr = 0.0;
    di = 0.0;
    if (runs.n_zz > 0) {
        const W t = a ^ (a >> 1);
        for (int r = 0; r < runs.n_zz; ++r) {
            const W m = (W)(((uint64_t)1 << runs.zz_cnt[r]) - 1) << runs.zz_lo0[r];
            dr += runs.zz_v[r] * (double)(runs.zz_cnt[r] - 2 * WordTraits<W>::popc(t & m));
        }
    }
    if (runs.n_zz_terms < n_diag) {
        double gr, gi;
        term_sum<REAL>(diag, runs.n_zz_terms, n_diag, (uint64_t)a, gr, gi);
        dr += gr;
        di += gi;
    }
}


// ---- lane / complex helpers shared by the row, pull and packet kernels (were next to k_chain_t)
template <typename W, typename R> struct ChainTraits;
template <> struct ChainTraits<uint32_t, uint32_t> { static constexpr int NB = 32; };
template <> struct ChainTraits<uint64_t, uint32_t> { static constexpr int NB = 64; };
template <> struct ChainTraits<uint64_t, uint64_t> { static constexpr int NB = 64; };

template <typename T> __device__ __forceinline__ T readlane_t(T v, int lane);
template <> __device__ __forceinline__ uint32_t readlane_t<uint32_t>(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
template <> __device__ __forceinline__ uint64_t readlane_t<uint64_t>(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
template <typename T> __device__ __forceinline__ T readfirstlane_t(T v);
template <> __device__ __forceinline__ uint32_t readfirstlane_t<uint32_t>(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
template <> __device__ __forceinline__ uint64_t readfirstlane_t<uint64_t>(uint64_t v) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <bool CPLX> struct ChainX { typedef double type; };
template <> struct ChainX<true> { typedef double2 type; };
__device__ __forceinline__ void cx_fma(double c, double v, double &acc) { acc = fma(c, v, acc); }
__device__ __forceinline__ void cx_fma(double c, double2 v, double2 &acc) { acc.x = fma(c, v.x, acc.x); acc.y = fma(c, v.y, acc.y); }
__device__ __forceinline__ double cx_scale(double c, double v) { return c * v; }
__device__ __forceinline__ double2 cx_scale(double c, double2 v) { return make_double2(c * v.x, c * v.y); }
template <typename X> __device__ __forceinline__ X cx_zero();
template <> __device__ __forceinline__ double cx_zero<double>() { return 0.0; }
template <> __device__ __forceinline__ double2 cx_zero<double2>() { return make_double2(0.0, 0.0); }
__device__ __forceinline__ void cx_store_nt(double *p, double v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void cx_store_nt(double2 *p, double2 v) {
    __builtin_nontemporal_store(v.x, &p->x);
    __builtin_nontemporal_store(v.y, &p->y);
}


// block -> tile of the pull kernels.  Blocks b = x (mod 8) run on XCD x; a plain grid therefore deals every XCD every eighth
// tile, and each of the eight L2s fetches its own copy of the partner sectors that neighbouring tiles share.  With a chunk of C
// tiles per XCD the blocks of one XCD walk C consecutive tiles before they jump by 8 C.  (The last, incomplete round of
// chunks keeps the identity.)
__host__ __device__ __forceinline__ int64_t pull_tile_of_block(int64_t b, int64_t n_tiles, int C) {
    if (C <= 1) return b;
    const int64_t round = 8 * (int64_t)C, full = n_tiles / round * round;
    if (b >= full) return b;
    const int64_t x = b & 7, j = b >> 3, q = j / C, r = j - q * C;
    return (q * 8 + x) * C + r;
}

