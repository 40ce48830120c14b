/* CPU oracle, part 2 (C + OpenMP): restatement of the reference's matvec hot path.
 *
 * TEST INFRASTRUCTURE ONLY -- never linked into, imported by or called from the product
 * (distributed-matvec_amd/).  Used by tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py, as the checker / the timed CPU baseline ("port").
 *
 * PARITY UNPINNED by reference artefacts: the per-row arithmetic the reference calls
 * (ls_internal_operator_apply_{diag,off_diag}_x1, ls_hs_state_info, ls_hs_state_index,
 * ls_hs_is_representative: /root/reference/src/FFI.chpl:173-184,219-225) lives in the
 * un-vendored lattice-symmetries-haskell @ 14e7319 and the golden HDF5 files are absent
 * (/root/reference/Makefile:128-146).  This file restates (a) those externs from their call
 * sites and the executable spec /root/reference/src/BatchedOperator.chpl:11-36, and (b) the
 * Chapel control flow around them.  It is cross-checked against the independent dense
 * Kronecker/projector construction in oracle/model.py (tests/test_oracle_*.py).
 *
 * Each function cites the reference lines it follows.
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef double _Complex c128;

typedef struct {
    int n;
    c128 *v;
    uint64_t *m, *r, *x, *s;
} lso_terms;

typedef struct {
    int number_sites;
    int hamming_weight; /* -1: unrestricted */
    int spin_inversion; /* 0: none */
    int group_order;    /* permutation part only, >= 1 (identity first) */
    int32_t *perms;     /* [group_order][number_sites] */
    c128 *chars;        /* [group_order] */
    uint64_t *tables;   /* [group_order][8][256] byte-sliced permutation tables */
    lso_terms diag, off;
    int n_groups;       /* distinct flip masks in `off` (terms sorted by x) */
    int *group_begin;   /* [n_groups + 1] */
} lso_model;

/* ------------------------------------------------------------------------------------ */
/* helpers                                                                              */
/* ------------------------------------------------------------------------------------ */

static void terms_copy(lso_terms *t, int n, const double *v, const uint64_t *m, const uint64_t *r,
                       const uint64_t *x, const uint64_t *s) {
    t->n = n;
    t->v = (c128 *)malloc(sizeof(c128) * (n > 0 ? n : 1));
    t->m = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
    t->r = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
    t->x = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
    t->s = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) {
        t->v[i] = v[2 * i] + v[2 * i + 1] * I;
        t->m[i] = m[i];
        t->r[i] = r[i];
        t->x[i] = x[i];
        t->s[i] = s[i];
    }
}

static void binom_init(void);

static uint64_t perm_naive(const int32_t *p, int L, uint64_t s) {
    uint64_t out = 0;
    for (int i = 0; i < L; ++i) out |= ((s >> p[i]) & 1ULL) << i;
    return out;
}

lso_model *lso_model_create(int number_sites, int hamming_weight, int spin_inversion,
                            int group_order, const int32_t *perms, const double *chars, int n_diag,
                            const double *dv, const uint64_t *dm, const uint64_t *dr,
                            const uint64_t *dx, const uint64_t *ds, int n_off, const double *ov,
                            const uint64_t *om, const uint64_t *orr, const uint64_t *ox,
                            const uint64_t *os) {
    lso_model *M = (lso_model *)calloc(1, sizeof(lso_model));
    binom_init(); /* before any parallel region touches the table */
    M->number_sites = number_sites;
    M->hamming_weight = hamming_weight;
    M->spin_inversion = spin_inversion;
    M->group_order = group_order;
    M->perms = (int32_t *)malloc(sizeof(int32_t) * group_order * number_sites);
    memcpy(M->perms, perms, sizeof(int32_t) * group_order * number_sites);
    M->chars = (c128 *)malloc(sizeof(c128) * group_order);
    for (int g = 0; g < group_order; ++g) M->chars[g] = chars[2 * g] + chars[2 * g + 1] * I;
    M->tables = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)group_order * 8 * 256);
    for (int g = 0; g < group_order; ++g)
        for (int b = 0; b < 8; ++b)
            for (int val = 0; val < 256; ++val)
                M->tables[((size_t)g * 8 + b) * 256 + val] =
                    perm_naive(M->perms + (size_t)g * number_sites, number_sites,
                               (uint64_t)val << (8 * b));
    terms_copy(&M->diag, n_diag, dv, dm, dr, dx, ds);
    terms_copy(&M->off, n_off, ov, om, orr, ox, os);
    /* group boundaries: terms must arrive sorted by x */
    M->group_begin = (int *)malloc(sizeof(int) * (n_off + 2));
    M->n_groups = 0;
    for (int i = 0; i < n_off; ++i) {
        if (i == 0 || M->off.x[i] != M->off.x[i - 1]) M->group_begin[M->n_groups++] = i;
        if (i > 0 && M->off.x[i] < M->off.x[i - 1]) {
            fprintf(stderr, "lso_model_create: off-diagonal terms must be sorted by x\n");
            abort();
        }
    }
    M->group_begin[M->n_groups] = n_off;
    return M;
}

void lso_model_destroy(lso_model *M) {
    if (!M) return;
    free(M->perms); free(M->chars); free(M->tables); free(M->group_begin);
    free(M->diag.v); free(M->diag.m); free(M->diag.r); free(M->diag.x); free(M->diag.s);
    free(M->off.v); free(M->off.m); free(M->off.r); free(M->off.x); free(M->off.s);
    free(M);
}

int lso_max_number_off_diag(const lso_model *M) { return M->n_groups; }
int lso_number_diag_terms(const lso_model *M) { return M->diag.n; }

static inline uint64_t perm_apply(const lso_model *M, int g, uint64_t s) {
    const uint64_t *T = M->tables + (size_t)g * 8 * 256;
    uint64_t out = 0;
    for (int b = 0; b < 8; ++b) out |= T[b * 256 + ((s >> (8 * b)) & 0xFF)];
    return out;
}

static inline int has_permutations(const lso_model *M) { return M->group_order > 1; }
static inline int requires_projection(const lso_model *M) {
    return has_permutations(M) || M->spin_inversion != 0;
}
static inline uint64_t site_mask(const lso_model *M) {
    return M->number_sites >= 64 ? ~0ULL : ((1ULL << M->number_sites) - 1);
}

/* /root/reference/src/StatesEnumeration.chpl:122-127 */
uint64_t lso_hash64_01(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return x;
}
/* /root/reference/src/StatesEnumeration.chpl:133-136 */
static inline int locale_idx_of(uint64_t s, int P) { return (int)(lso_hash64_01(s) % (uint64_t)P); }

void lso_locale_idx_of(int64_t n, const uint64_t *states, int num_locales, uint8_t *keys) {
    for (int64_t i = 0; i < n; ++i) keys[i] = (uint8_t)locale_idx_of(states[i], num_locales);
}

/* /root/reference/src/StatesEnumeration.chpl:31-34 */
static inline uint64_t next_state_fixed_hamming(uint64_t v) {
    uint64_t t = v | (v - 1);
    return (t + 1) | (((~t & (t + 1)) - 1) >> (__builtin_ctzll(v) + 1));
}

static uint64_t binom_tab[65][65];
static int binom_ready = 0;
static void binom_init(void) {
    if (binom_ready) return;
    for (int n = 0; n <= 64; ++n) {
        binom_tab[n][0] = 1;
        for (int k = 1; k <= 64; ++k)
            binom_tab[n][k] = (n == 0) ? 0 : binom_tab[n - 1][k - 1] + (k <= n - 1 ? binom_tab[n - 1][k] : 0);
    }
    binom_ready = 1;
}

/* ls_hs_fixed_hamming_state_to_index (/root/reference/src/FFI.chpl:165): combinadic rank among
 * equal-popcount integers in ascending order (usage /root/reference/src/StatesEnumeration.chpl:77-88) */
int64_t lso_fixed_hamming_state_to_index(uint64_t s) {
    binom_init();
    int64_t idx = 0;
    int k = 1;
    while (s) {
        int p = __builtin_ctzll(s);
        idx += (int64_t)binom_tab[p][k];
        ++k;
        s &= s - 1;
    }
    return idx;
}
/* batch form for the sampled-row parity tests (one call instead of one ctypes call per state) */
void lso_fixed_hamming_state_to_index_batch(int64_t n, const uint64_t *states, int64_t *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = lso_fixed_hamming_state_to_index(states[i]);
}
/* ls_hs_fixed_hamming_index_to_state (/root/reference/src/FFI.chpl:166) */
uint64_t lso_fixed_hamming_index_to_state(int64_t idx, int hamming_weight) {
    binom_init();
    uint64_t s = 0;
    for (int k = hamming_weight; k >= 1; --k) {
        int p = k - 1;
        while (p + 1 <= 63 && (int64_t)binom_tab[p + 1][k] <= idx) ++p;
        s |= 1ULL << p;
        idx -= (int64_t)binom_tab[p][k];
    }
    return s;
}

/* ------------------------------------------------------------------------------------ */
/* restated externs                                                                     */
/* ------------------------------------------------------------------------------------ */

static inline c128 diag_coeff(const lso_model *M, uint64_t a) {
    c128 acc = 0;
    const lso_terms *t = &M->diag;
    for (int k = 0; k < t->n; ++k)
        if ((a & t->m[k]) == t->r[k])
            acc += (__builtin_popcountll(a & t->s[k]) & 1) ? -t->v[k] : t->v[k];
    return acc;
}

/* ls_internal_operator_apply_diag_x1: ys[i] = d(alphas[i]) * xs[i]; xs == NULL -> ys[i] = d
 * (/root/reference/src/DistributedMatrixVector.chpl:43-45, BatchedOperator.chpl:229-230) */
void lso_apply_diag_x1(const lso_model *M, int64_t n, const uint64_t *alphas, double *ys,
                       const double *xs) {
    for (int64_t i = 0; i < n; ++i) {
        double d = creal(diag_coeff(M, alphas[i]));
        ys[i] = xs ? d * xs[i] : d;
    }
}

/* ls_internal_operator_apply_off_diag_x1, semantics spelled out by localCompressMultiply
 * (/root/reference/src/BatchedOperator.chpl:11-36): per row, every flip-mask group with a
 * non-zero coefficient is appended as (beta, c * xs[row]); offsets = running count. */
void lso_apply_off_diag_x1(const lso_model *M, int64_t n, const uint64_t *alphas, uint64_t *betas,
                           c128 *coeffs, int64_t *offsets, const double *xs) {
    const lso_terms *t = &M->off;
    int64_t off = 0;
    offsets[0] = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t a = alphas[i];
        for (int g = 0; g < M->n_groups; ++g) {
            c128 acc = 0;
            for (int k = M->group_begin[g]; k < M->group_begin[g + 1]; ++k)
                if ((a & t->m[k]) == t->r[k])
                    acc += (__builtin_popcountll(a & t->s[k]) & 1) ? -t->v[k] : t->v[k];
            if (acc != 0) {
                betas[off] = a ^ t->x[M->group_begin[g]];
                coeffs[off] = xs ? acc * xs[i] : acc;
                ++off;
            }
        }
        offsets[i + 1] = off;
    }
}
/* the same with complex xs (north-star c128 mode; the reference's extern is f64-only,
 * /root/reference/src/FFI.chpl:225) */
void lso_apply_off_diag_x1_c128(const lso_model *M, int64_t n, const uint64_t *alphas,
                                uint64_t *betas, c128 *coeffs, int64_t *offsets, const c128 *xs) {
    lso_apply_off_diag_x1(M, n, alphas, betas, coeffs, offsets, NULL);
    if (xs)
        for (int64_t i = 0; i < n; ++i)
            for (int64_t k = offsets[i]; k < offsets[i + 1]; ++k) coeffs[k] *= xs[i];
}

/* ls_hs_state_info (/root/reference/src/FFI.chpl:181-184; usage BatchedOperator.chpl:184-203).
 * beta = min_g g(alpha) over permutations x optional global flip; character = conj(chi(g0));
 * norm = sqrt((1/|G|) sum_{g in Stab(alpha)} chi(g)). */
void lso_state_info(const lso_model *M, int64_t n, const uint64_t *alphas, uint64_t *betas,
                    c128 *characters, double *norms) {
    const uint64_t mask = site_mask(M);
    const int inv = M->spin_inversion;
    const double order = (double)M->group_order * (inv ? 2.0 : 1.0);
    for (int64_t i = 0; i < n; ++i) {
        uint64_t a = alphas[i];
        uint64_t best = ~0ULL;
        c128 bestc = 1;
        c128 stab = 0;
        for (int g = 0; g < M->group_order; ++g) {
            uint64_t t = perm_apply(M, g, a);
            c128 ch = M->chars[g];
            if (t == a) stab += ch;
            if (t < best) { best = t; bestc = ch; }
            if (inv) {
                uint64_t tf = t ^ mask;
                c128 chf = ch * (double)inv;
                if (tf == a) stab += chf;
                if (tf < best) { best = tf; bestc = chf; }
            }
        }
        double n2 = creal(stab) / order;
        betas[i] = best;
        characters[i] = conj(bestc);
        norms[i] = n2 > 1e-12 ? sqrt(n2) : 0.0;
    }
}

/* ls_hs_is_representative (/root/reference/src/FFI.chpl:177-179; usage
 * StatesEnumeration.chpl:176-188): flag = alpha is its orbit minimum; norm as above. */
void lso_is_representative(const lso_model *M, int64_t n, const uint64_t *alphas, uint8_t *flags,
                           double *norms) {
    for (int64_t i = 0; i < n; ++i) {
        uint64_t b; c128 ch; double nr;
        lso_state_info(M, 1, alphas + i, &b, &ch, &nr);
        flags[i] = (b == alphas[i]);
        norms[i] = nr;
    }
}

/* ls_hs_state_index (/root/reference/src/FFI.chpl:173-175; usage DMV:102): position in the sorted
 * representatives, negative if absent. */
static inline int64_t binary_search(const uint64_t *reps, int64_t count, uint64_t s) {
    int64_t lo = 0, hi = count;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (reps[mid] < s) lo = mid + 1; else hi = mid;
    }
    return (lo < count && reps[lo] == s) ? lo : -1;
}
/* Bucketed form of the same search ("a (possibly bucketed) binary search"): a table of lower bounds by
 * the top bits of the state, built once per representatives array, then a binary search inside the bucket (~8 states).
 * Identical results; it only keeps the CPU baseline of bench.py from being dominated by 25 cache-missing probes per
 * look-up.  The table belongs to the basis (built with it, outside any timed region: local_matvec builds it on first
 * use and the bench's warm-up call pays for that). */
static struct { const uint64_t *reps; int64_t count; int shift; uint32_t *table; uint64_t nbuckets; } g_index;
static void ensure_index(const uint64_t *reps, int64_t count) {
    if (g_index.reps == reps && g_index.count == count && g_index.table) return;
    free(g_index.table);
    g_index.table = NULL;
    g_index.reps = reps; g_index.count = count;
    if (count < (1 << 12) || count >= 0xffffffffLL) return; /* small arrays: plain search */
    int bits = 0;
    while ((1LL << (bits + 1)) <= count) ++bits;
    bits -= 3;
    int top = 64 - __builtin_clzll(reps[count - 1] | 1); /* bits needed for the largest state */
    if (bits > top) bits = top;
    g_index.shift = top - bits;
    g_index.nbuckets = (1ULL << bits);
    g_index.table = (uint32_t *)malloc(4 * (g_index.nbuckets + 2));
    int64_t i = 0;
    for (uint64_t b = 0; b <= g_index.nbuckets; ++b) {
        while (i < count && (reps[i] >> g_index.shift) < b) ++i;
        g_index.table[b] = (uint32_t)i;
    }
    g_index.table[g_index.nbuckets + 1] = (uint32_t)count;
}
static inline int64_t indexed_search(const uint64_t *reps, int64_t count, uint64_t s) {
    if (!g_index.table || g_index.reps != reps) return binary_search(reps, count, s);
    uint64_t b = s >> g_index.shift;
    if (b >= g_index.nbuckets) return -1;
    int64_t lo = g_index.table[b], hi = g_index.table[b + 1];
    const int64_t end = hi;
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (reps[mid] < s) lo = mid + 1; else hi = mid;
    }
    return (lo < end && reps[lo] == s) ? lo : -1;
}
/* the bucket table of `reps` built ahead of a timed matvec (bench.py's cpu_baseline: the table belongs to the basis, as
 * ls_hs_basis_build leaves the reference's index structures behind before any matvec is timed) */
void lso_prepare_index(const uint64_t *reps, int64_t count) { ensure_index(reps, count); }
void lso_state_index(const uint64_t *reps, int64_t count, int64_t n, const uint64_t *spins,
                     int64_t *indices) {
    for (int64_t i = 0; i < n; ++i) indices[i] = binary_search(reps, count, spins[i]);
}

/* ------------------------------------------------------------------------------------ */
/* enumeration (numLocales == 1 view of /root/reference/src/StatesEnumeration.chpl:158-224)  */
/* ------------------------------------------------------------------------------------ */

static uint64_t min_state(const lso_model *M) {
    return M->hamming_weight >= 0 ? (M->hamming_weight == 0 ? 0 : ((1ULL << M->hamming_weight) - 1)) : 0;
}
/* ls_hs_max_state_estimate [upstream-memory]: with spin inversion the highest admissible state has
 * the top site bit clear (a state and its flip are one orbit; the smaller one has bit L-1 == 0),
 * which is what makes `high = min(high, high ^ mask)` at StatesEnumeration.chpl:211-216 a no-op
 * for the reference's own call; chain_10 then ends at 496. */
static uint64_t max_state(const lso_model *M) {
    const int L = M->number_sites - (M->spin_inversion ? 1 : 0);
    if (M->hamming_weight >= 0)
        return M->hamming_weight == 0 ? 0 : ((1ULL << M->hamming_weight) - 1) << (L - M->hamming_weight);
    return L >= 64 ? ~0ULL : ((1ULL << L) - 1);
}

/* returns the number of basis states in [lower, upper] (inclusive; pass lower > upper to use
 * the whole range); writes them when out != NULL. */
int64_t lso_enumerate(const lso_model *M, uint64_t lower, uint64_t upper, uint64_t *out) {
    if (lower > upper) { lower = min_state(M); upper = max_state(M); }
    const int fixed = M->hamming_weight >= 0;
    int64_t count = 0;
    if (!has_permutations(M)) {
        /* _enumerateStatesUnprojected :201-224 -- spin inversion cuts the range */
        uint64_t high = upper;
        if (M->spin_inversion) {
            uint64_t alt = high ^ site_mask(M);
            if (alt < high) high = alt;
        }
        uint64_t v = lower;
        if (v > high) return 0;
        for (;;) {
            if (out) out[count] = v;
            ++count;
            if (v == high) break;
            v = fixed ? next_state_fixed_hamming(v) : v + 1;
            if (v > high) break;
        }
        return count;
    }
    /* _enumerateStatesProjected :158-200 */
    uint64_t v = lower;
    for (;;) {
        uint8_t flag; double norm;
        lso_is_representative(M, 1, &v, &flag, &norm);
        if (flag && norm > 0) { if (out) out[count] = v; ++count; }
        if (v == upper) break;
        v = fixed ? next_state_fixed_hamming(v) : v + 1;
    }
    return count;
}

/* parallel variant for big symmetric bases: splits the index range like
 * determineEnumerationRanges (/root/reference/src/StatesEnumeration.chpl:94-113) */
int64_t lso_enumerate_parallel(const lso_model *M, uint64_t *out, int64_t capacity, int num_threads) {
    if (!has_permutations(M) || M->hamming_weight < 0) {
        int64_t n = lso_enumerate(M, 1, 0, NULL);
        if (out && n <= capacity) lso_enumerate(M, 1, 0, out);
        return n;
    }
    const int hw = M->hamming_weight;
    int64_t lo = lso_fixed_hamming_state_to_index(min_state(M));
    int64_t hi = lso_fixed_hamming_state_to_index(max_state(M));
    int64_t total = hi - lo + 1;
    int nchunks = 64 * (num_threads > 0 ? num_threads : 1);
    if (nchunks > total) nchunks = (int)total;
    int64_t *counts = (int64_t *)calloc(nchunks + 1, sizeof(int64_t));
    uint64_t **bufs = (uint64_t **)calloc(nchunks, sizeof(uint64_t *));
#pragma omp parallel for schedule(dynamic, 1) num_threads(num_threads > 0 ? num_threads : 1)
    for (int c = 0; c < nchunks; ++c) {
        int64_t a = lo + total * c / nchunks, b = lo + total * (c + 1) / nchunks - 1;
        uint64_t sa = lso_fixed_hamming_index_to_state(a, hw), sb = lso_fixed_hamming_index_to_state(b, hw);
        int64_t n = lso_enumerate(M, sa, sb, NULL);
        bufs[c] = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
        lso_enumerate(M, sa, sb, bufs[c]);
        counts[c] = n;
    }
    int64_t n = 0;
    for (int c = 0; c < nchunks; ++c) {
        if (out && n + counts[c] <= capacity) memcpy(out + n, bufs[c], 8 * counts[c]);
        n += counts[c];
        free(bufs[c]);
    }
    free(bufs); free(counts);
    return n;
}

/* ------------------------------------------------------------------------------------ */
/* BatchedOperator.computeOffDiag  (/root/reference/src/BatchedOperator.chpl:82-213)        */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    int64_t batch;
    uint64_t *spins1, *spins2;
    c128 *coeffs1, *coeffs2;
    double *norms;
    uint8_t *keys;
    int64_t *offsets;
} batched_op;

static void batched_init(batched_op *b, const lso_model *M, int64_t batch) {
    int64_t nt = M->n_groups > 1 ? M->n_groups : 1;
    int64_t cap = batch * (nt + 1); /* :63-64 */
    b->batch = batch;
    b->spins1 = (uint64_t *)malloc(8 * cap);
    b->spins2 = (uint64_t *)malloc(8 * cap);
    b->coeffs1 = (c128 *)malloc(16 * cap);
    b->coeffs2 = (c128 *)malloc(16 * cap);
    b->norms = (double *)malloc(8 * cap);
    b->keys = (uint8_t *)malloc(cap);
    b->offsets = (int64_t *)malloc(8 * (batch + 1));
}
static void batched_free(batched_op *b) {
    free(b->spins1); free(b->spins2); free(b->coeffs1); free(b->coeffs2);
    free(b->norms); free(b->keys); free(b->offsets);
}

/* xs_is_complex selects the c128 north-star mode */
static int64_t compute_off_diag(batched_op *b, const lso_model *M, int64_t count,
                                const uint64_t *alphas, const void *xs, int xs_is_complex,
                                int num_locales, uint64_t **betas_out, c128 **cs_out) {
    if (!requires_projection(M) || !has_permutations(M)) {
        /* branches (i) :89-116 and (ii) :119-161 */
        if (xs_is_complex)
            lso_apply_off_diag_x1_c128(M, count, alphas, b->spins1, b->coeffs1, b->offsets, (const c128 *)xs);
        else
            lso_apply_off_diag_x1(M, count, alphas, b->spins1, b->coeffs1, b->offsets, (const double *)xs);
        int64_t total = b->offsets[count];
        if (M->spin_inversion) {
            const uint64_t mask = site_mask(M);
            for (int64_t i = 0; i < total; ++i) { /* :144-151 */
                uint64_t cur = b->spins1[i], inv = cur ^ mask;
                if (inv < cur) { b->spins1[i] = inv; b->coeffs1[i] *= (double)M->spin_inversion; }
            }
        }
        for (int64_t i = 0; i < total; ++i)
            b->keys[i] = (uint8_t)(num_locales > 1 ? locale_idx_of(b->spins1[i], num_locales) : 0);
        *betas_out = b->spins1; *cs_out = b->coeffs1;
        return total;
    }
    /* branch (iii) :163-213 */
    if (xs_is_complex)
        lso_apply_off_diag_x1_c128(M, count, alphas, b->spins2, b->coeffs2, b->offsets, (const c128 *)xs);
    else
        lso_apply_off_diag_x1(M, count, alphas, b->spins2, b->coeffs2, b->offsets, (const double *)xs);
    int64_t total = b->offsets[count];
    memcpy(b->spins2 + total, alphas, 8 * count); /* :178-182 */
    lso_state_info(M, total + count, b->spins2, b->spins1, b->coeffs1, b->norms);
    for (int64_t i = 0; i < count; ++i)
        for (int64_t k = b->offsets[i]; k < b->offsets[i + 1]; ++k)
            b->coeffs1[k] *= b->coeffs2[k] * b->norms[k] / b->norms[total + i]; /* :198-202 */
    for (int64_t i = 0; i < total; ++i)
        b->keys[i] = (uint8_t)(num_locales > 1 ? locale_idx_of(b->spins1[i], num_locales) : 0);
    *betas_out = b->spins1; *cs_out = b->coeffs1;
    return total;
}

/* ------------------------------------------------------------------------------------ */
/* localProcess + ConcurrentAccessor  (DMV:73-127, ConcurrentAccessor.chpl:48-54)        */
/* ------------------------------------------------------------------------------------ */

static inline void atomic_add_f64(double *p, double v) {
#pragma omp atomic
    *p += v;
}

static int local_process(const lso_model *M, int identity_index, const uint64_t *reps, int64_t count,
                         void *y, int y_is_complex, const uint64_t *betas, const c128 *cs, int64_t size) {
    for (int64_t k = 0; k < size; ++k) {
        c128 c = cs[k];
        int64_t idx;
        if (identity_index) idx = (int64_t)betas[k]; /* DMV:86-95 */
        else {
            if (c == 0) continue; /* DMV:110 */
            idx = indexed_search(reps, count, betas[k]);
            if (idx < 0) {        /* DMV:115-118 halt */
                fprintf(stderr, "lso: invalid index for state %llu with coeff (%g,%g)\n",
                        (unsigned long long)betas[k], creal(c), cimag(c));
                return -1;
            }
        }
        if (y_is_complex) {
            double *yy = (double *)y + 2 * idx;
            atomic_add_f64(yy, creal(c));
            atomic_add_f64(yy + 1, cimag(c));
        } else {
            atomic_add_f64((double *)y + idx, creal(c)); /* cast c128 -> eltType, DMV:91,109 */
        }
    }
    return 0;
}

static int state_index_is_identity(const lso_model *M) {
    return M->hamming_weight < 0 && !requires_projection(M);
}

/* ------------------------------------------------------------------------------------ */
/* localMatrixVector, numLocales == 1   (DMV:1055-1070 -> :59-71 and :856-1053)            */
/* ------------------------------------------------------------------------------------ */

static int local_matvec(const lso_model *M, int64_t count, const uint64_t *reps, const void *x,
                        void *y, int is_complex, int num_threads) {
    if (num_threads <= 0) {
#ifdef _OPENMP
        num_threads = omp_get_max_threads();
#else
        num_threads = 1;
#endif
    }
    /* localDiagonal: y[i] = d * x[i]  (assignment; only when there are diagonal terms) */
    if (M->diag.n > 0) {
#pragma omp parallel for schedule(static) num_threads(num_threads)
        for (int64_t i = 0; i < count; ++i) {
            c128 d = diag_coeff(M, reps[i]);
            if (is_complex) ((c128 *)y)[i] = d * ((const c128 *)x)[i];
            else ((double *)y)[i] = creal(d) * ((const double *)x)[i];
        }
    }
    if (M->n_groups == 0 || count == 0) return 0;
    /* chunking rule DMV:871-883 with kRemoteBufferSize = 150000 (DMV:456) */
    const int64_t T = M->n_groups;
    const int64_t buf = 150000 > T ? 150000 : T;
    int64_t num_chunks = (count * T + buf - 1) / buf;
    if (num_chunks < 10 * (int64_t)num_threads) num_chunks = 10 * (int64_t)num_threads;
    if (num_chunks > count) num_chunks = count;
    const int64_t chunk_size = (count + num_chunks - 1) / num_chunks;
    const int ident = state_index_is_identity(M);
    if (!ident) ensure_index(reps, count);
    int failed = 0;
#pragma omp parallel num_threads(num_threads)
    {
        batched_op b;
        batched_init(&b, M, chunk_size);
#pragma omp for schedule(dynamic, 1) /* atomic chunk counter, DMV:670-679 */
        for (int64_t c = 0; c < num_chunks; ++c) {
            /* RangeChunk.chunks(0..#count, numChunks) with the default remainder policy */
            int64_t lo = (count * c) / num_chunks, hi = (count * (c + 1)) / num_chunks;
            if (hi <= lo) continue;
            uint64_t *betas; c128 *cs;
            const void *xs = is_complex ? (const void *)((const c128 *)x + lo)
                                        : (const void *)((const double *)x + lo);
            int64_t n = compute_off_diag(&b, M, hi - lo, reps + lo, xs, is_complex, 1, &betas, &cs);
            if (local_process(M, ident, reps, count, y, is_complex, betas, cs, n) != 0) {
#pragma omp atomic write
                failed = 1;
            }
        }
        batched_free(&b);
    }
    return failed ? -1 : 0;
}

int lso_local_matvec_f64(const lso_model *M, int64_t count, const uint64_t *reps, const double *x,
                         double *y, int num_threads) {
    return local_matvec(M, count, reps, x, y, 0, num_threads);
}
int lso_local_matvec_c128(const lso_model *M, int64_t count, const uint64_t *reps, const c128 *x,
                          c128 *y, int num_threads) {
    return local_matvec(M, count, reps, x, y, 1, num_threads);
}

/* ------------------------------------------------------------------------------------ */
/* matrixVectorProduct over P hash partitions (DMV:1072-1093), executed sequentially:      */
/* every "locale" produces chunks, buckets them by key (radixOneStep DMV:265-311) and the  */
/* owner applies localProcess (own bucket DMV:695-715, remote buckets via Consumer         */
/* DMV:818-852).  Communication is replaced by direct calls; arithmetic is unchanged.      */
/* ------------------------------------------------------------------------------------ */

static int partitioned_matvec(const lso_model *M, int P, const int64_t *counts,
                              const uint64_t *const *reps, const void *const *x, void *const *y,
                              int is_complex) {
    const int64_t T = M->n_groups > 0 ? M->n_groups : 1;
    for (int p = 0; p < P; ++p) {
        if (M->diag.n > 0)
            for (int64_t i = 0; i < counts[p]; ++i) {
                c128 d = diag_coeff(M, reps[p][i]);
                if (is_complex) ((c128 *)y[p])[i] = d * ((const c128 *)x[p])[i];
                else ((double *)y[p])[i] = creal(d) * ((const double *)x[p])[i];
            }
    }
    if (M->n_groups == 0) return 0;
    /* DEVIATION: the reference takes the identity shortcut (DMV:86-95) whenever the basis flag is
     * set, which for numLocales > 1 would index a partition-local array with a global state; a
     * hashed partition needs the search, so the shortcut is only taken for P == 1. */
    const int ident = state_index_is_identity(M) && P == 1;
    for (int p = 0; p < P; ++p) {
        const int64_t count = counts[p];
        if (count == 0) continue;
        const int64_t buf = 150000 > T ? 150000 : T;
        int64_t num_chunks = (count * T + buf - 1) / buf;
        if (num_chunks < 10) num_chunks = 10;
        if (num_chunks > count) num_chunks = count;
        const int64_t chunk_size = (count + num_chunks - 1) / num_chunks;
        batched_op b;
        batched_init(&b, M, chunk_size);
        uint64_t *sb = (uint64_t *)malloc(8 * chunk_size * (T + 1));
        c128 *sc = (c128 *)malloc(16 * chunk_size * (T + 1));
        for (int64_t c = 0; c < num_chunks; ++c) {
            int64_t lo = (count * c) / num_chunks, hi = (count * (c + 1)) / num_chunks;
            if (hi <= lo) continue;
            uint64_t *betas; c128 *cs;
            const void *xs = is_complex ? (const void *)((const c128 *)x[p] + lo)
                                        : (const void *)((const double *)x[p] + lo);
            int64_t n = compute_off_diag(&b, M, hi - lo, reps[p] + lo, xs, is_complex, P, &betas, &cs);
            for (int dest = 0; dest < P; ++dest) { /* bucket by key, then owner processes */
                int64_t m = 0;
                for (int64_t k = 0; k < n; ++k)
                    if (b.keys[k] == dest) { sb[m] = betas[k]; sc[m] = cs[k]; ++m; }
                if (local_process(M, ident, reps[dest], counts[dest], y[dest], is_complex, sb, sc, m) != 0) {
                    free(sb); free(sc); batched_free(&b);
                    return -1;
                }
            }
        }
        free(sb); free(sc);
        batched_free(&b);
    }
    return 0;
}

int lso_matvec_partitioned_f64(const lso_model *M, int P, const int64_t *counts,
                               const uint64_t *const *reps, const double *const *x, double *const *y) {
    return partitioned_matvec(M, P, counts, reps, (const void *const *)x, (void *const *)y, 0);
}
int lso_matvec_partitioned_c128(const lso_model *M, int P, const int64_t *counts,
                                const uint64_t *const *reps, const c128 *const *x, c128 *const *y) {
    return partitioned_matvec(M, P, counts, reps, (const void *const *)x, (void *const *)y, 1);
}

/* ------------------------------------------------------------------------------------ */
/* layout converters: semantics of arrFromBlockToHashed / arrFromHashedToBlock             */
/* (/root/reference/src/BlockToHashed.chpl:87-208, HashedToBlock.chpl:67-153): stable        */
/* partition by masks and its inverse (k-way unmerge).  elt_size in bytes.                 */
/* ------------------------------------------------------------------------------------ */

void lso_block_to_hashed(int64_t n, const uint8_t *masks, int P, int elt_size, const void *src,
                         void *const *dest, int64_t *counts_out) {
    int64_t *w = (int64_t *)calloc(P, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        int k = masks[i];
        memcpy((char *)dest[k] + w[k] * elt_size, (const char *)src + i * elt_size, elt_size);
        ++w[k];
    }
    if (counts_out) memcpy(counts_out, w, sizeof(int64_t) * P);
    free(w);
}
void lso_hashed_to_block(int64_t n, const uint8_t *masks, int P, int elt_size,
                         const void *const *src, void *dest) {
    int64_t *w = (int64_t *)calloc(P, sizeof(int64_t));
    for (int64_t i = 0; i < n; ++i) {
        int k = masks[i];
        memcpy((char *)dest + i * elt_size, (const char *)src[k] + w[k] * elt_size, elt_size);
        ++w[k];
    }
    free(w);
}

int lso_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
