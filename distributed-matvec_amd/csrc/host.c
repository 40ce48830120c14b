/* host.c -- C host side of the MI355X-native matvec: the ls_hs_* ABI subset, plans, and the
 * ls_chpl_* exports of the reference's shared library.  Plain C11; every device action goes through
 * the extern-"C" shim in the k_*.hip translation units (lsk.h).  No compute on the CPU: the functions here build
 * tables (symmetry-group closure, Benes networks, term grouping), size buffers and launch kernels.
 *
 * Reference behaviour mirrored (see include/ls_chpl.h, include/ls_amd.h for per-symbol citations):
 *   localMatrixVector / matrixVectorProduct   /root/reference/src/DistributedMatrixVector.chpl:1055-1110
 *   localOffDiagonalNoQueue (sizing, rounds)  /root/reference/src/DistributedMatrixVector.chpl:856-1053
 *   enumerateStates                           /root/reference/src/StatesEnumeration.chpl:516-603
 *   Diagonalize's PRIMME callback             /root/reference/src/Diagonalize.chpl:134-162
 */
#define _GNU_SOURCE
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/ls_amd.h"
#include "../../include/ls_chpl.h"
#include "../../include/ls_hs.h"
#include "lsk.h"

/* ============================================================================================ */
/* errors                                                                                       */
/* ============================================================================================ */
static __thread char g_last_error[1024] = "";
static ls_amd_error_handler g_handler = NULL;

static int set_error(char const *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return -1;
}
static int dev_error(void) { return set_error("%s", lsk_last_error()); }
/* for the other host files (dist.c): same buffer, same -1 */
int ls_amd_internal_error(char const *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return -1;
}
#define DEV(expr) do { if ((expr) != 0) return dev_error(); } while (0)
/* a constructor that succeeded leaves no message behind (a failed attempt before it must not read as this call's error) */
void ls_amd_internal_clear_error(void) { g_last_error[0] = 0; }

char const *ls_amd_last_error(void) { return g_last_error; }
void ls_amd_set_error_handler(ls_amd_error_handler handler) { g_handler = handler; }

/* the reference's `halt(...)` */
static void halt_with(char const *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    snprintf(g_last_error, sizeof(g_last_error), "%s", buf);
    if (g_handler) { g_handler(buf); return; }
    fprintf(stderr, "[Error]   halt: %s\n", buf);
    abort();
}

/* ============================================================================================ */
/* device helpers re-exported                                                                   */
/* ============================================================================================ */
/* object registry                                                                              */
/* ============================================================================================ */
/* Everything this library knows about a basis / an operator beyond the struct prefix the reference declares
 * (/root/reference/src/FFI.chpl:94-119) lives in a side table keyed by the object's address -- NOT in a trailing struct
 * field: an ls_hs_operator built by the real lattice-symmetries-haskell has "other stuff" of unknown layout behind the
 * prefix, and reading a field of ours there would be out of bounds.  Objects created here are registered by their
 * constructors; foreign ones by ls_amd_adopt_basis / ls_amd_adopt_operator. */
typedef struct { void const *key; void *val; int kind; } reg_slot;
enum { REG_BASIS = 1, REG_OPERATOR = 2 };
static reg_slot *g_reg = NULL;
static size_t g_reg_cap = 0, g_reg_used = 0; /* used counts live entries and tombstones */
static pthread_mutex_t g_reg_lock = PTHREAD_MUTEX_INITIALIZER;
static void const *const REG_TOMB = (void const *)(uintptr_t)1;

static size_t reg_hash(void const *k, size_t cap) { return (size_t)(((uintptr_t)k >> 4) * 0x9E3779B97F4A7C15ULL) & (cap - 1); }
static void reg_put_nolock(void const *key, void *val, int kind);
static void reg_grow(void) {
    reg_slot *old = g_reg;
    size_t const oc = g_reg_cap;
    g_reg_cap = oc ? 2 * oc : 64;
    g_reg = (reg_slot *)calloc(g_reg_cap, sizeof(reg_slot));
    g_reg_used = 0;
    for (size_t i = 0; i < oc; ++i)
        if (old[i].key && old[i].key != REG_TOMB) reg_put_nolock(old[i].key, old[i].val, old[i].kind);
    free(old);
}
static void reg_put_nolock(void const *key, void *val, int kind) {
    if (2 * (g_reg_used + 1) > g_reg_cap) reg_grow();
    size_t i = reg_hash(key, g_reg_cap);
    while (g_reg[i].key && g_reg[i].key != REG_TOMB && g_reg[i].key != key) i = (i + 1) & (g_reg_cap - 1);
    if (!g_reg[i].key) ++g_reg_used;
    g_reg[i].key = key;
    g_reg[i].val = val;
    g_reg[i].kind = kind;
}
static void reg_put(void const *key, void *val, int kind) {
    pthread_mutex_lock(&g_reg_lock);
    reg_put_nolock(key, val, kind);
    pthread_mutex_unlock(&g_reg_lock);
}
static void *reg_find(void const *key, int *kind) {
    void *v = NULL;
    pthread_mutex_lock(&g_reg_lock);
    if (g_reg_cap) {
        size_t i = reg_hash(key, g_reg_cap);
        while (g_reg[i].key) {
            if (g_reg[i].key == key) { v = g_reg[i].val; if (kind) *kind = g_reg[i].kind; break; }
            i = (i + 1) & (g_reg_cap - 1);
        }
    }
    pthread_mutex_unlock(&g_reg_lock);
    return v;
}
static void *reg_get(void const *key) { return reg_find(key, NULL); }
static void reg_del(void const *key) {
    pthread_mutex_lock(&g_reg_lock);
    if (g_reg_cap) {
        size_t i = reg_hash(key, g_reg_cap);
        while (g_reg[i].key) {
            if (g_reg[i].key == key) { g_reg[i].key = REG_TOMB; g_reg[i].val = NULL; break; }
            i = (i + 1) & (g_reg_cap - 1);
        }
    }
    pthread_mutex_unlock(&g_reg_lock);
}
static void halt_with(char const *fmt, ...);
struct ls_amd_basis_ext;
struct ls_amd_operator_ext;
static struct ls_amd_basis_ext *basis_ext_of(ls_hs_basis const *b);
static struct ls_amd_operator_ext *operator_ext_of(ls_hs_operator const *op);
#define BEXT(b) basis_ext_of(b)
#define OEXT(op) operator_ext_of(op)

/* ============================================================================================ */
int ls_amd_device_count(void) { return lsk_device_count(); }
int ls_amd_set_device(int device) { DEV(lsk_set_device(device)); return 0; }
int ls_amd_malloc(void **d_ptr, size_t bytes) { DEV(lsk_malloc(d_ptr, bytes)); return 0; }
int ls_amd_free(void *d_ptr) { DEV(lsk_free(d_ptr)); return 0; }
int ls_amd_memcpy_h2d(void *d, void const *h, size_t bytes) { DEV(lsk_h2d(d, h, bytes)); return 0; }
int ls_amd_memcpy_d2h(void *h, void const *d, size_t bytes) { DEV(lsk_d2h(h, d, bytes)); return 0; }
int ls_amd_memcpy_d2d(void *d, void const *s, size_t bytes, void *stream) { DEV(lsk_d2d_async(d, s, bytes, stream)); return 0; }
int ls_amd_memset(void *d, int value, size_t bytes, void *stream) { DEV(lsk_memset_async(d, value, bytes, stream)); return 0; }
int ls_amd_synchronize(void *stream) { DEV(lsk_sync(stream)); return 0; }

uint64_t ls_amd_hash64_01(uint64_t x) {
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return x;
}
int ls_amd_locale_idx_of(uint64_t basis_state, int num_locales) {
    return (int)(ls_amd_hash64_01(basis_state) % (uint64_t)num_locales);
}

/* ============================================================================================ */
/* binomials                                                                                    */
/* ============================================================================================ */
static uint64_t g_binom[64 * LSK_BINOM_K];
static int g_binom_ready = 0;
static uint64_t *g_d_binom = NULL;

static void binom_fill(void) {
    for (int n = 0; n < 64; ++n)
        for (int k = 0; k < LSK_BINOM_K; ++k) {
            uint64_t v;
            if (k == 0) v = 1;
            else if (n == 0) v = 0;
            else v = g_binom[(n - 1) * LSK_BINOM_K + (k - 1)] + g_binom[(n - 1) * LSK_BINOM_K + k];
            g_binom[n * LSK_BINOM_K + k] = v;
        }
    g_binom_ready = 1;
}
static pthread_once_t g_binom_once = PTHREAD_ONCE_INIT;
static void binom_init(void) { pthread_once(&g_binom_once, binom_fill); } /* (first use may come from several host threads at once) */
static uint64_t binom(int n, int k) {
    binom_init();
    if (k < 0 || n < 0 || k > n) return 0;
    if (n - k < k) k = n - k;
    if (n < 64 && k < LSK_BINOM_K) return g_binom[n * LSK_BINOM_K + k];
    /* n == 64 or beyond table: compute directly (only used for sizes) */
    long double r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;
    return (uint64_t)(r + 0.5L);
}
static pthread_mutex_t g_binom_lock = PTHREAD_MUTEX_INITIALIZER;
static int device_binom_unlocked(uint64_t const **out);
static int device_binom(uint64_t const **out) { /* first use may come from several host threads at once */
    pthread_mutex_lock(&g_binom_lock);
    int const rc = device_binom_unlocked(out);
    pthread_mutex_unlock(&g_binom_lock);
    return rc;
}
static int device_binom_unlocked(uint64_t const **out) {
    binom_init();
    if (!g_d_binom) {
        void *p;
        DEV(lsk_malloc(&p, sizeof(g_binom)));
        DEV(lsk_h2d(p, g_binom, sizeof(g_binom)));
        g_d_binom = (uint64_t *)p;
    }
    *out = g_d_binom;
    return 0;
}

ptrdiff_t ls_hs_fixed_hamming_state_to_index(uint64_t s) {
    binom_init();
    ptrdiff_t idx = 0;
    int k = 1;
    while (s) {
        int p = __builtin_ctzll(s);
        idx += (ptrdiff_t)binom(p, k);
        ++k;
        s &= s - 1;
    }
    return idx;
}
uint64_t ls_hs_fixed_hamming_index_to_state(ptrdiff_t idx, int hamming_weight) {
    uint64_t s = 0;
    int p = 63;
    for (int k = hamming_weight; k >= 1; --k) {
        while (p > k - 1 && (ptrdiff_t)binom(p, k) > idx) --p;
        s |= 1ULL << p;
        idx -= (ptrdiff_t)binom(p, k);
        --p;
    }
    return s;
}

/* ============================================================================================ */
/* basis                                                                                        */
/* ============================================================================================ */
struct ls_amd_basis_ext {
    int hamming_weight; /* -1 unrestricted */
    int n_generators;
    int *gen_perms;     /* [n_generators][L] */
    int *gen_sectors;
    int order;          /* permutation group order (>= 1, identity first) */
    int *perms;         /* [order][L] */
    lsk_group_elem *elems; /* [order] host copy (characters, networks) */
    lsk_group_elem *d_elems;
    /* K4 mode 4 (lsk_basis in lsk.h): the group contains every translation of a tw x (L / tw) torus; coset_ids = one element
     * of every right coset T g (indices into elems), found once per basis.  tw == 0: not looked for yet, -1: no such subgroup */
    int tw, n_cosets;
    int *coset_ids;
    lsk_group_elem *d_cosets;
    uint32_t *d_trow;    /* mode 4, tw <= 8: the row table of torus_min on the device */
    int d4_mask;         /* mode 5: which of 1, r, o, r o (bits 0-3) and of their products with the transpose (bits 4-7) the cosets are;
                          * 0 = not examined, -1 = the cosets are not of that form */
    lsk_group_elem d4_transpose; /* the transpose network (the only compiled network of mode 5) */
    lsk_group_elem *d_d4_net;
    uint64_t *d_trow2;   /* mode 5: row table with the reversed row's fields in the high half */
    int owns_representatives;
    uint64_t *d_reps_cache; /* device copy of `representatives` for the host-pointer entry points */
    uint64_t d_reps_count;
    /* plans cached by the host-pointer entry points (ls_chpl_matrix_vector_product, ls_chpl_primme_matvec), keyed by the
     * OPERATOR and the communicator they were built for: several operators share one basis (H and an observable,
     * ls_hs_clone_operator), and each needs its own device term tables */
    struct host_plan_slot {
        ls_hs_operator const *op;
        void *comm;    /* ls_amd_comm* (NULL: single process) */
        void *plan;    /* ls_amd_plan* when comm == NULL */
        void *dist;    /* ls_amd_dist* otherwise */
        uint64_t stamp;
        /* persistent device copies of the caller's host vectors (allocated on first use, kept with the plan; the second pair
         * only for blocks of columns: the upload of column k + 1 overlaps kernel and download of column k) */
        void *d_x[2], *d_y[2];
        int64_t stage_n;
    } host_plans[4];
    uint64_t host_plan_clock;
    uint32_t *d_index_table; /* search table over d_reps_cache (ls_hs_state_index) */
    int index_kind, index_shift;
    int refcount;           /* the creator's reference + one per operator built on the basis */
    int adopted;            /* the struct belongs to somebody else (ls_amd_adopt_basis): never freed here */
};

static struct ls_amd_basis_ext g_dummy_basis_ext;
static struct ls_amd_basis_ext *basis_ext_of(ls_hs_basis const *b) {
    struct ls_amd_basis_ext *e = (struct ls_amd_basis_ext *)reg_get(b);
    if (!e) {
        halt_with("ls_hs_basis %p was not created by this library: register it with ls_amd_adopt_basis first", (void const *)b);
        g_dummy_basis_ext.hamming_weight = -1;
        return &g_dummy_basis_ext;
    }
    return e;
}

void ls_hs_init(void) {}
void ls_hs_exit(void) {}

static int perm_order(int const *p, int L) {
    int *q = (int *)malloc(sizeof(int) * L), *t = (int *)malloc(sizeof(int) * L);
    memcpy(q, p, sizeof(int) * L);
    int n = 1;
    for (;;) {
        int ident = 1;
        for (int i = 0; i < L; ++i) if (q[i] != i) { ident = 0; break; }
        if (ident) break;
        for (int i = 0; i < L; ++i) t[i] = q[p[i]]; /* compose(p, q)[i] = q[p[i]] */
        memcpy(q, t, sizeof(int) * L);
        ++n;
    }
    free(q); free(t);
    return n;
}

/* Benes network for y[i] = x[src[i]], i < 64.  masks[level] = input stage of recursion level
 * `level` (distance 32 >> level), masks[10 - level] = its output stage, masks[5] = the middle. */
static void benes_route(int base, int n, int const *src, int level, uint64_t *masks) {
    if (n == 2) {
        if (src[0] == 1) masks[5] |= 1ULL << base;
        return;
    }
    int const h = n / 2;
    int inv[64], incolor[64], outcolor[64], lower[32], upper[32];
    for (int i = 0; i < n; ++i) { inv[src[i]] = i; incolor[i] = -1; outcolor[i] = -1; }
    for (int o = 0; o < n; ++o) {
        if (outcolor[o] != -1) continue;
        int cur = o, c = 0;
        while (outcolor[cur] == -1) {
            outcolor[cur] = c;
            int s = src[cur];
            incolor[s] = c;
            int sp = s ^ h;
            incolor[sp] = 1 - c;
            int o2 = inv[sp];
            outcolor[o2] = 1 - c;
            cur = o2 ^ h;
        }
    }
    for (int j = 0; j < h; ++j) {
        if (incolor[j] == 1) masks[level] |= 1ULL << (base + j);
        if (outcolor[j] == 1) masks[10 - level] |= 1ULL << (base + j);
    }
    for (int i = 0; i < h; ++i) {
        int fo_l = (outcolor[i] == 0) ? i : i + h;
        int fo_u = (outcolor[i] == 1) ? i : i + h;
        lower[i] = src[fo_l] & (h - 1);
        upper[i] = src[fo_u] & (h - 1);
    }
    benes_route(base, h, lower, level + 1, masks);
    benes_route(base + h, h, upper, level + 1, masks);
}

static void compile_elem(int const *perm, int L, double ch_re, double ch_im, lsk_group_elem *e) {
    memset(e, 0, sizeof(*e));
    e->ch_re = ch_re;
    e->ch_im = ch_im;
    /* rotation: perm[i] = (i + k) mod L  -> y = rotr_L(x, k) */
    int k = perm[0];
    int is_rot = 1;
    for (int i = 0; i < L; ++i) if (perm[i] != (i + k) % L) { is_rot = 0; break; }
    if (is_rot) { e->kind = LSK_ELEM_ROT; e->k = k; return; }
    /* reflection + rotation: perm[i] = (k' - i) mod L -> y = rotr_L(rev_L(x), (L - 1 - k') mod L) */
    int kp = perm[0];
    int is_rev = 1;
    for (int i = 0; i < L; ++i) if (perm[i] != ((kp - i) % L + L) % L) { is_rev = 0; break; }
    if (is_rev) { e->kind = LSK_ELEM_REVROT; e->k = ((L - 1 - kp) % L + L) % L; return; }
    int src[64];
    for (int i = 0; i < 64; ++i) src[i] = i < L ? perm[i] : i;
    e->kind = LSK_ELEM_BENES;
    benes_route(0, 64, src, 0, e->masks);
}

static uint64_t host_delta_swap(uint64_t x, uint64_t m, int d) {
    uint64_t t = ((x >> d) ^ x) & m;
    return x ^ t ^ (t << d);
}
/* CPU mirror of apply_elem in lsk_dev.hpp -- used by the test hooks and table checks only */
static uint64_t host_apply_elem(lsk_group_elem const *e, uint64_t x, int L) {
    static int const dist[LSK_BENES_STAGES] = {32, 16, 8, 4, 2, 1, 2, 4, 8, 16, 32};
    uint64_t const mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    if (e->kind == LSK_ELEM_BENES) {
        for (int s = 0; s < LSK_BENES_STAGES; ++s)
            if (e->masks[s]) x = host_delta_swap(x, e->masks[s], dist[s]);
        return x;
    }
    if (e->kind == LSK_ELEM_REVROT) {
        uint64_t r = 0;
        for (int i = 0; i < L; ++i) r |= ((x >> i) & 1ULL) << (L - 1 - i);
        x = r;
    }
    if (e->k == 0) return x;
    return ((x >> e->k) | (x << (L - e->k))) & mask;
}

/* The kernels loop over the group elements of a projected basis per candidate state; the largest groups of the reference's inputs
 * have 288 elements (heisenberg_square_6x6), the automorphisms of a 64-site lattice a few thousand */
enum { LS_AMD_MAX_GROUP_ORDER = 1 << 20 }; /* closure is linear in the order (hash-table membership): a million elements take milliseconds */
static uint64_t perm_hash(int const *p, int L) {
    uint64_t h = 0x9e3779b97f4a7c15ULL;
    for (int i = 0; i < L; ++i) { h ^= (uint64_t)(p[i] + 1); h *= 0xff51afd7ed558ccdULL; h ^= h >> 29; }
    return h;
}
static int close_group(struct ls_amd_basis_ext *ext, int L) {
    int const ng = ext->n_generators;
    int cap = 64, order = 1;
    int *perms = (int *)malloc(sizeof(int) * cap * L);
    double *chars = (double *)malloc(sizeof(double) * 2 * cap);
    double *gch = (double *)malloc(sizeof(double) * 2 * (ng > 0 ? ng : 1));
    for (int i = 0; i < L; ++i) perms[i] = i;
    chars[0] = 1.0; chars[1] = 0.0;
    for (int g = 0; g < ng; ++g) {
        int const *p = ext->gen_perms + g * L;
        int n = perm_order(p, L);
        int k = ((ext->gen_sectors[g] % n) + n) % n;
        double ang = -2.0 * M_PI * (double)k / (double)n;
        double re = cos(ang), im = sin(ang);
        /* snap the exact values so real sectors stay exactly real */
        if (fabs(re) < 1e-15) re = 0.0;
        if (fabs(im) < 1e-15) im = 0.0;
        if (fabs(fabs(re) - 1.0) < 1e-15) re = re > 0 ? 1.0 : -1.0;
        if (fabs(fabs(im) - 1.0) < 1e-15) im = im > 0 ? 1.0 : -1.0;
        gch[2 * g] = re; gch[2 * g + 1] = im;
    }
    int *cand = (int *)malloc(sizeof(int) * L);
    /* membership by an open-addressing table of element numbers keyed by a hash of the permutation: the closure is linear in
     * the group order (a linear scan per candidate made generators of a huge group -- a typo in one permutation of a YAML file
     * is enough: two random permutations generate S_L -- run ~2^40 comparisons before the size limit below was reached) */
    size_t tcap = 256;
    int32_t *table = (int32_t *)malloc(sizeof(int32_t) * tcap);
    for (size_t i = 0; i < tcap; ++i) table[i] = -1;
    table[perm_hash(perms, L) & (tcap - 1)] = 0;
    for (int head = 0; head < order; ++head) { /* BFS: `order` grows while we scan */
        for (int g = 0; g < ng; ++g) {
            int const *p = ext->gen_perms + g * L;
            int const *e = perms + (size_t)head * L;
            for (int i = 0; i < L; ++i) cand[i] = e[p[i]]; /* compose(g, e) */
            double cre = chars[2 * head] * gch[2 * g] - chars[2 * head + 1] * gch[2 * g + 1];
            double cim = chars[2 * head] * gch[2 * g + 1] + chars[2 * head + 1] * gch[2 * g];
            int found = -1;
            size_t slot = perm_hash(cand, L) & (tcap - 1);
            for (; table[slot] >= 0; slot = (slot + 1) & (tcap - 1))
                if (memcmp(perms + (size_t)table[slot] * L, cand, sizeof(int) * L) == 0) { found = table[slot]; break; }
            if (found >= 0) {
                if (fabs(chars[2 * found] - cre) > 1e-9 || fabs(chars[2 * found + 1] - cim) > 1e-9) {
                    free(perms); free(chars); free(gch); free(cand); free(table);
                    return set_error("symmetry sectors are incompatible with the group structure");
                }
                continue;
            }
            if (order >= LS_AMD_MAX_GROUP_ORDER) {
                free(perms); free(chars); free(gch); free(cand); free(table);
                return set_error("symmetry group too large (more than %d elements; note: one wrong entry in a permutation makes the generators span nearly S_L)", LS_AMD_MAX_GROUP_ORDER);
            }
            if (order == cap) {
                cap *= 2;
                perms = (int *)realloc(perms, sizeof(int) * cap * L);
                chars = (double *)realloc(chars, sizeof(double) * 2 * cap);
            }
            memcpy(perms + (size_t)order * L, cand, sizeof(int) * L);
            chars[2 * order] = cre; chars[2 * order + 1] = cim;
            table[slot] = order;
            ++order;
            if ((size_t)order * 2 > tcap) { /* keep the table at most half full */
                tcap *= 2;
                table = (int32_t *)realloc(table, sizeof(int32_t) * tcap);
                for (size_t i = 0; i < tcap; ++i) table[i] = -1;
                for (int j = 0; j < order; ++j) {
                    size_t s2 = perm_hash(perms + (size_t)j * L, L) & (tcap - 1);
                    while (table[s2] >= 0) s2 = (s2 + 1) & (tcap - 1);
                    table[s2] = j;
                }
            }
        }
    }
    free(table);
    ext->order = order;
    ext->perms = perms;
    ext->elems = (lsk_group_elem *)malloc(sizeof(lsk_group_elem) * order);
    for (int j = 0; j < order; ++j) {
        double re = chars[2 * j], im = chars[2 * j + 1];
        if (fabs(im) < 1e-14) im = 0.0;
        if (fabs(re) < 1e-14) re = 0.0;
        compile_elem(perms + (size_t)j * L, L, re, im, ext->elems + j);
    }
    free(chars); free(gch); free(cand);
    return 0;
}

ls_hs_basis *ls_hs_create_spin_basis(int number_sites, int hamming_weight, int spin_inversion,
                                     int number_generators, int const *permutations,
                                     int const *sectors) {
    if (number_sites < 1 || number_sites > 64) { set_error("number_sites must be in [1, 64] (number_words == 1)"); return NULL; }
    if (hamming_weight > number_sites) { set_error("hamming_weight > number_sites"); return NULL; }
    if (hamming_weight >= LSK_BINOM_K) { set_error("hamming_weight too large for the binomial table"); return NULL; }
    if (spin_inversion != 0 && spin_inversion != 1 && spin_inversion != -1) { set_error("spin_inversion must be 0, 1 or -1"); return NULL; }
    if (spin_inversion != 0 && hamming_weight >= 0 && 2 * hamming_weight != number_sites) {
        set_error("spin inversion requires hamming_weight == number_sites / 2"); return NULL;
    }
    for (int g = 0; g < number_generators; ++g) {
        uint64_t seen = 0;
        for (int i = 0; i < number_sites; ++i) {
            int v = permutations[g * number_sites + i];
            if (v < 0 || v >= number_sites || ((seen >> v) & 1)) { set_error("generator %d is not a permutation", g); return NULL; }
            seen |= 1ULL << v;
        }
    }
    ls_hs_basis *b = (ls_hs_basis *)calloc(1, sizeof(ls_hs_basis));
    struct ls_amd_basis_ext *ext = (struct ls_amd_basis_ext *)calloc(1, sizeof(*ext));
    reg_put(b, ext, REG_BASIS);
    ext->refcount = 1;
    b->number_sites = number_sites;
    b->number_particles = -1;
    b->number_up = hamming_weight >= 0 ? hamming_weight : -1;
    b->particle_type = LS_HS_SPIN;
    b->spin_inversion = spin_inversion;
    b->kernels = NULL;
    ext->hamming_weight = hamming_weight < 0 ? -1 : hamming_weight;
    ext->n_generators = number_generators;
    ext->gen_perms = (int *)malloc(sizeof(int) * (size_t)(number_generators > 0 ? number_generators : 1) * number_sites);
    ext->gen_sectors = (int *)malloc(sizeof(int) * (size_t)(number_generators > 0 ? number_generators : 1));
    if (number_generators > 0) {
        memcpy(ext->gen_perms, permutations, sizeof(int) * (size_t)number_generators * number_sites);
        memcpy(ext->gen_sectors, sectors, sizeof(int) * (size_t)number_generators);
    }
    if (close_group(ext, number_sites) != 0) { ls_hs_destroy_basis(b); return NULL; }
    b->requires_projection = ext->order > 1 || spin_inversion != 0;
    b->state_index_is_identity = ext->hamming_weight < 0 && !b->requires_projection;
    return b;
}

ls_hs_basis *ls_hs_clone_basis(ls_hs_basis const *basis) {
    struct ls_amd_basis_ext const *e = BEXT(basis);
    ls_hs_basis *b = ls_hs_create_spin_basis(basis->number_sites, e->hamming_weight, basis->spin_inversion,
                                             e->n_generators, e->gen_perms, e->gen_sectors);
    if (!b) return NULL;
    if (basis->representatives.elts) { /* deep copy, like upstream's clone keeps the built states */
        size_t bytes = 8 * basis->representatives.num_elts;
        b->representatives.elts = malloc(bytes ? bytes : 8);
        memcpy(b->representatives.elts, basis->representatives.elts, bytes);
        b->representatives.num_elts = basis->representatives.num_elts;
        b->representatives.freer = (void *)free;
        BEXT(b)->owns_representatives = 1;
    }
    return b;
}

#define HOST_PLAN_SLOTS ((int)(sizeof(((struct ls_amd_basis_ext *)0)->host_plans) / sizeof(struct host_plan_slot)))
static void host_plan_slot_drop(struct host_plan_slot *sl) {
    if (sl->plan) ls_amd_plan_destroy((ls_amd_plan *)sl->plan);
    if (sl->dist) ls_amd_dist_destroy((ls_amd_dist *)sl->dist);
    for (int i = 0; i < 2; ++i) { if (sl->d_x[i]) lsk_free(sl->d_x[i]); if (sl->d_y[i]) lsk_free(sl->d_y[i]); }
    memset(sl, 0, sizeof(*sl));
}
/* Forget the cached plans of one operator (op != NULL), of one communicator (comm != NULL), or all of them.  The slots are
 * DETACHED under the registry lock -- a basis cannot disappear, and no second thread can pick the same slot, while they are
 * read -- and destroyed after it is released (destroying a plan takes that lock again). */
static int detach_host_plans_locked(struct ls_amd_basis_ext *e, ls_hs_operator const *op, void const *comm,
                                    struct host_plan_slot *out, int n, int cap) {
    for (int i = 0; i < HOST_PLAN_SLOTS && n < cap; ++i) {
        struct host_plan_slot *sl = &e->host_plans[i];
        if (!sl->op) continue;
        if ((op && sl->op != op) || (comm && sl->comm != comm)) continue;
        out[n++] = *sl;
        memset(sl, 0, sizeof(*sl));
    }
    return n;
}
static void basis_drop_host_plans(struct ls_amd_basis_ext *e, ls_hs_operator const *op, void const *comm) {
    struct host_plan_slot dropped[HOST_PLAN_SLOTS];
    pthread_mutex_lock(&g_reg_lock);
    int const n = detach_host_plans_locked(e, op, comm, dropped, 0, HOST_PLAN_SLOTS);
    pthread_mutex_unlock(&g_reg_lock);
    for (int i = 0; i < n; ++i) host_plan_slot_drop(&dropped[i]);
}
/* a communicator is going away (ls_amd_comm_destroy, dist.c): no cached ls_amd_dist may keep pointing at it */
void ls_amd_internal_forget_comm(void const *comm) {
    pthread_mutex_lock(&g_reg_lock);
    size_t bases = 0;
    for (size_t i = 0; i < g_reg_cap; ++i)
        if (g_reg[i].key && g_reg[i].key != REG_TOMB && g_reg[i].kind == REG_BASIS) ++bases;
    int const cap = (int)(bases ? bases : 1) * HOST_PLAN_SLOTS;
    struct host_plan_slot *dropped = (struct host_plan_slot *)calloc((size_t)cap, sizeof(*dropped));
    int n = 0;
    for (size_t i = 0; i < g_reg_cap; ++i)
        if (g_reg[i].key && g_reg[i].key != REG_TOMB && g_reg[i].kind == REG_BASIS)
            n = detach_host_plans_locked((struct ls_amd_basis_ext *)g_reg[i].val, NULL, comm, dropped, n, cap);
    pthread_mutex_unlock(&g_reg_lock);
    for (int i = 0; i < n; ++i) host_plan_slot_drop(&dropped[i]);
    free(dropped);
}

static void basis_drop_device_caches(ls_hs_basis *b) {
    struct ls_amd_basis_ext *e = BEXT(b);
    basis_drop_host_plans(e, NULL, NULL);
    if (e->d_reps_cache) { lsk_free(e->d_reps_cache); e->d_reps_cache = NULL; e->d_reps_count = 0; }
    if (e->d_index_table) { lsk_free(e->d_index_table); e->d_index_table = NULL; }
    e->index_kind = -1;
}

void ls_hs_destroy_basis(ls_hs_basis *b) {
    if (!b) return;
    struct ls_amd_basis_ext *e = (struct ls_amd_basis_ext *)reg_get(b);
    if (!e) return; /* not ours (or already gone): leave the struct alone */
    /* (atomic: threads that clone / destroy operators of one basis -- the reference's tasks do -- must not lose a count) */
    if (__atomic_sub_fetch(&e->refcount, 1, __ATOMIC_ACQ_REL) > 0) return; /* operators built on this basis still share it */
    basis_drop_device_caches(b);
    if (e->d_elems) lsk_free(e->d_elems);
    if (e->d_cosets) lsk_free(e->d_cosets);
    if (e->d_trow) lsk_free(e->d_trow);
    if (e->d_d4_net) lsk_free(e->d_d4_net);
    if (e->d_trow2) lsk_free(e->d_trow2);
    if (e->owns_representatives && b->representatives.elts) free(b->representatives.elts);
    free(e->gen_perms); free(e->gen_sectors); free(e->perms); free(e->elems); free(e->coset_ids);
    reg_del(b);
    int const adopted = e->adopted;
    free(e);
    if (!adopted) free(b);
}

uint64_t ls_hs_min_state_estimate(ls_hs_basis const *b) {
    int h = BEXT(b)->hamming_weight;
    return h > 0 ? ((1ULL << h) - 1) : 0;
}
/* with spin inversion the highest admissible state has the top site bit clear (see oracle notes) */
uint64_t ls_hs_max_state_estimate(ls_hs_basis const *b) {
    int const L = b->number_sites - (b->spin_inversion != 0 ? 1 : 0);
    int h = BEXT(b)->hamming_weight;
    if (h >= 0) return h == 0 ? 0 : ((1ULL << h) - 1) << (L - h);
    return L >= 64 ? ~0ULL : ((1ULL << L) - 1);
}
int ls_hs_basis_number_bits(ls_hs_basis const *b) { return b->number_sites; }
int ls_hs_basis_number_words(ls_hs_basis const *b) { return (b->number_sites + 63) / 64; }
bool ls_hs_basis_has_fixed_hamming_weight(ls_hs_basis const *b) { return BEXT(b)->hamming_weight >= 0; }
bool ls_hs_basis_has_spin_inversion_symmetry(ls_hs_basis const *b) { return b->spin_inversion != 0; }
bool ls_hs_basis_has_permutation_symmetries(ls_hs_basis const *b) { return BEXT(b)->order > 1; }
bool ls_hs_basis_requires_projection(ls_hs_basis const *b) { return b->requires_projection; }

void ls_hs_unchecked_set_representatives(ls_hs_basis *b, chpl_external_array const *states) {
    basis_drop_device_caches(b);
    if (BEXT(b)->owns_representatives && b->representatives.elts) free(b->representatives.elts);
    b->representatives = *states;
    BEXT(b)->owns_representatives = 0;
}

int ls_amd_basis_group_order(ls_hs_basis const *b) { return BEXT(b)->order; }
uint64_t ls_amd_basis_apply_group_element(ls_hs_basis const *b, int element, uint64_t state) {
    return host_apply_elem(BEXT(b)->elems + element, state, b->number_sites);
}
int ls_amd_basis_group_character(ls_hs_basis const *b, int element, double *re, double *im) {
    if (element < 0 || element >= BEXT(b)->order) return set_error("element out of range");
    *re = BEXT(b)->elems[element].ch_re;
    *im = BEXT(b)->elems[element].ch_im;
    return 0;
}

static pthread_mutex_t g_device_tables_lock = PTHREAD_MUTEX_INITIALIZER;
/* Looks for the translation subgroup of a lattice group (K4 mode 4).  Every element is taken by its ACTION on the one-hot
 * states (img[g][s] = position of the bit of g(1 << s)), so no convention about permutation arrays enters: t is a translation
 * of the w x h torus (s = y w + x) when img_t[s] = ((y + dy) % h) w + (x + dx) % w; the subgroup qualifies when all w h of them
 * are in the group.  Then one element of every right coset T g is kept (t o g covers the coset: img_{t o g}[s] = img_t[img_g[s]]).
 * Sets e->tw (-1: none), e->n_cosets, e->coset_ids. */
static void find_translation_cosets(struct ls_amd_basis_ext *e, int L) {
    e->tw = -1;
    int const order = e->order;
    if (order < 4 || order > 8192 || L < 4) return;
    unsigned char *img = (unsigned char *)malloc((size_t)order * (size_t)L);
    for (int g = 0; g < order; ++g)
        for (int s = 0; s < L; ++s) img[(size_t)g * L + s] = (unsigned char)__builtin_ctzll(host_apply_elem(&e->elems[g], 1ULL << s, L));
    /* index of an action among the group's (open addressing over an FNV hash) */
    int cap = 1;
    while (cap < 4 * order) cap *= 2;
    int *table = (int *)malloc(sizeof(int) * (size_t)cap);
    for (int i = 0; i < cap; ++i) table[i] = -1;
#define IMG_HASH(ptr, out_h) do { uint64_t h_ = 1469598103934665603ULL; for (int s_ = 0; s_ < L; ++s_) { h_ ^= (ptr)[s_]; h_ *= 1099511628211ULL; } (out_h) = (int)(h_ & (uint64_t)(cap - 1)); } while (0)
    for (int g = 0; g < order; ++g) {
        int h;
        IMG_HASH(img + (size_t)g * L, h);
        while (table[h] >= 0) h = (h + 1) & (cap - 1);
        table[h] = g;
    }
    unsigned char *cand = (unsigned char *)malloc((size_t)L);
    int *trans = (int *)malloc(sizeof(int) * (size_t)L);
    int best_w = -1;
    for (int w = 2; w <= L / 2 && best_w < 0; ++w) {
        if (L % w) continue;
        int const hgt = L / w;
        int found = 0;
        for (int dy = 0; dy < hgt; ++dy)
            for (int dx = 0; dx < w; ++dx) {
                for (int s = 0; s < L; ++s) cand[s] = (unsigned char)((((s / w) + dy) % hgt) * w + ((s % w) + dx) % w);
                int h, id = -1;
                IMG_HASH(cand, h);
                while (table[h] >= 0) {
                    if (memcmp(img + (size_t)table[h] * L, cand, (size_t)L) == 0) { id = table[h]; break; }
                    h = (h + 1) & (cap - 1);
                }
                if (id >= 0) trans[found++] = id;
            }
        if (found == L) best_w = w;
    }
    if (best_w > 0) {
        unsigned char *covered = (unsigned char *)calloc((size_t)order, 1);
        int *ids = (int *)malloc(sizeof(int) * (size_t)order);
        int nc = 0, ok = 1;
        for (int g = 0; g < order && ok; ++g) {
            if (covered[g]) continue;
            ids[nc++] = g;
            for (int t = 0; t < L && ok; ++t) {
                unsigned char const *it = img + (size_t)trans[t] * L, *ig = img + (size_t)g * L;
                for (int s = 0; s < L; ++s) cand[s] = it[ig[s]];
                int h, id = -1;
                IMG_HASH(cand, h);
                while (table[h] >= 0) {
                    if (memcmp(img + (size_t)table[h] * L, cand, (size_t)L) == 0) { id = table[h]; break; }
                    h = (h + 1) & (cap - 1);
                }
                if (id < 0) ok = 0; /* not closed: cannot happen for a group; be safe */
                else covered[id] = 1;
            }
        }
        if (ok && nc * L == order) {
            e->tw = best_w;
            e->n_cosets = nc;
            e->coset_ids = (int *)malloc(sizeof(int) * (size_t)nc);
            memcpy(e->coset_ids, ids, sizeof(int) * (size_t)nc);
        }
        free(covered); free(ids);
    }
#undef IMG_HASH
    free(img); free(table); free(cand); free(trans);
}

/* K4 mode 5: are the cosets T g found above the point group of the torus itself?  Canonical images of a site (y, x): r = (y, tw-1-x),
 * o = (th-1-y, x), and on a square torus the transpose (x, y) applied first; a coset matches the canonical element c when
 * img_g = img_t o img_c for some translation t.  Every coset must match a different c.  Sets e->d4_mask (> 0, or -1). */
static void classify_d4_cosets(struct ls_amd_basis_ext *e, int L) {
    e->d4_mask = -1;
    int const tw = e->tw;
    if (tw <= 0 || tw > 8 || L % tw || e->n_cosets < 1 || e->n_cosets > 8) return;
    int const th = L / tw;
    int mask = 0;
    for (int r = 0; r < e->n_cosets; ++r) {
        unsigned char ig[64];
        for (int s = 0; s < L; ++s) ig[s] = (unsigned char)__builtin_ctzll(host_apply_elem(&e->elems[e->coset_ids[r]], 1ULL << s, L));
        int found = -1;
        for (int c = 0; c < 8 && found < 0; ++c) {
            if ((c & 4) && tw != th) continue;
            for (int dy = 0; dy < th && found < 0; ++dy)
                for (int dx = 0; dx < tw && found < 0; ++dx) {
                    int ok = 1;
                    for (int s = 0; s < L && ok; ++s) {
                        int y = s / tw, x = s % tw;
                        if (c & 4) { int const t = y; y = x; x = t; } /* transpose first */
                        if (c & 1) x = tw - 1 - x;                     /* r */
                        if (c & 2) y = th - 1 - y;                     /* o */
                        ok = ig[s] == ((y + dy) % th) * tw + (x + dx) % tw;
                    }
                    if (ok) found = c;
                }
        }
        if (found < 0 || (mask >> found) & 1) return;
        mask |= 1 << found;
    }
    if (!(mask & 1)) return; /* (the coset of the identity is always there) */
    if (mask >> 4) { /* the transpose as a compiled network: y[i] = x[perm[i]], an involution */
        int perm[64];
        for (int s = 0; s < L; ++s) perm[s] = (s % tw) * tw + s / tw;
        compile_elem(perm, L, 1.0, 0.0, &e->d4_transpose);
    }
    e->d4_mask = mask;
}
int ls_amd_test_d4_mask(ls_hs_basis const *b) {
    struct ls_amd_basis_ext *e = BEXT(b);
    if (ls_amd_test_translation_cosets(b, NULL) <= 0) return 0;
    pthread_mutex_lock(&g_device_tables_lock);
    if (e->d4_mask == 0) classify_d4_cosets(e, b->number_sites);
    pthread_mutex_unlock(&g_device_tables_lock);
    return e->d4_mask > 0 ? e->d4_mask : 0;
}

/* host-only test hooks of K4 mode 4: the shape found for a basis (returns tw, or -1), and the orbit minimum of `state` computed
 * the way the kernels do it (coset networks + row / word rotations; the global spin flip folded in by canonicalising to "top
 * site clear" when the basis has one) */
int ls_amd_test_translation_cosets(ls_hs_basis const *b, int *n_cosets) {
    struct ls_amd_basis_ext *e = BEXT(b);
    pthread_mutex_lock(&g_device_tables_lock);
    if (e->tw == 0) find_translation_cosets(e, b->number_sites);
    pthread_mutex_unlock(&g_device_tables_lock);
    if (n_cosets) *n_cosets = e->tw > 0 ? e->n_cosets : 0;
    return e->tw;
}
uint64_t ls_amd_test_rep_by_cosets(ls_hs_basis const *b, uint64_t a) {
    struct ls_amd_basis_ext *e = BEXT(b);
    int const L = b->number_sites;
    if (ls_amd_test_translation_cosets(b, NULL) <= 0) return ~0ULL;
    uint64_t const mask = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
    int const tw = e->tw, th = L / tw;
    uint64_t col0 = 0, best = ~0ULL;
    for (int y = 0; y < th; ++y) col0 |= 1ULL << (y * tw);
    char const *k4env = getenv("LS_AMD_K4");
    int const d4 = (k4env && strcmp(k4env, "cosets") == 0) ? 0 : ls_amd_test_d4_mask(b);
    if (d4 > 0) { /* what the kernels run (mode 5): one transpose network, the rest factorised */
        uint64_t rowtab2[256];
        lsk_torus_rowtab2(tw, rowtab2);
        best = lsk_test_torus_min_d2(a, L, tw, b->spin_inversion != 0, d4 & 15, rowtab2, best);
        if (d4 >> 4) best = lsk_test_torus_min_d2(host_apply_elem(&e->d4_transpose, a, L), L, tw, b->spin_inversion != 0, d4 >> 4, rowtab2, best);
        return best;
    }
    if (tw <= 8) { /* mode 4: the row table picks the translations that can be minimal */
        uint32_t rowtab[256];
        lsk_torus_rowtab(tw, rowtab);
        for (int r = 0; r < e->n_cosets; ++r)
            best = lsk_test_torus_min(host_apply_elem(&e->elems[e->coset_ids[r]], a, L), L, tw, b->spin_inversion != 0, rowtab, best);
        return best;
    }
    for (int r = 0; r < e->n_cosets; ++r) {
        uint64_t v = host_apply_elem(&e->elems[e->coset_ids[r]], a, L);
        for (int j = 0; j < th; ++j) {
            for (int i = 0; i < tw; ++i) {
                uint64_t c = v;
                if (b->spin_inversion != 0 && ((v >> (L - 1)) & 1)) c = v ^ mask;
                if (c < best) best = c;
                v = ((v << 1) & ~col0 & mask) | ((v >> (tw - 1)) & col0);
            }
            v = ((v << tw) | (v >> (L - tw))) & mask;
        }
    }
    return best;
}

static int basis_device_unlocked(ls_hs_basis const *b, lsk_basis *out) {
    struct ls_amd_basis_ext *e = BEXT(b);
    if (!e->d_elems) {
        void *p;
        DEV(lsk_malloc(&p, sizeof(lsk_group_elem) * e->order));
        DEV(lsk_h2d(p, e->elems, sizeof(lsk_group_elem) * e->order));
        e->d_elems = (lsk_group_elem *)p;
    }
    out->number_sites = b->number_sites;
    out->hamming_weight = e->hamming_weight;
    out->spin_inversion = b->spin_inversion;
    out->n_elems = e->order;
    out->proj = e->order > 1 ? LSK_PROJ_FULL : (b->spin_inversion != 0 ? LSK_PROJ_INVERSION : LSK_PROJ_NONE);
    out->site_mask = b->number_sites >= 64 ? ~0ULL : ((1ULL << b->number_sites) - 1);
    out->inv_order = 1.0 / ((double)e->order * (b->spin_inversion != 0 ? 2.0 : 1.0));
    out->elems = e->d_elems;
    out->chars_pm1 = 1;
    int trivial = b->spin_inversion >= 0;
    for (int g = 0; g < e->order; ++g) {
        if (e->elems[g].ch_im != 0.0 || fabs(e->elems[g].ch_re) != 1.0) out->chars_pm1 = 0;
        if (e->elems[g].ch_im != 0.0 || e->elems[g].ch_re != 1.0) trivial = 0;
    }
    out->k4_mode = 0;
    out->reflect = 0;
    out->debug_ablate = lsk_ablate_mask(); /* 0 in the shipped library; LS_AMD_ABLATE in profiling builds (make ablate) */
    out->tw = out->n_cosets = 0;
    out->tcol0 = 0;
    out->cosets = NULL;
    out->trow = NULL;
    out->d4_mask = 0;
    out->trow2 = NULL;
    /* LS_AMD_K4 (test hook): general = the element loop with characters and norms even in trivial sectors; brute = trivial
     * sectors by the plain loop over every element (no run pruning, no translation cosets) */
    char const *k4env = getenv("LS_AMD_K4");
    int const k4_general = k4env && strcmp(k4env, "general") == 0, k4_brute = k4env && strcmp(k4env, "brute") == 0;
    int const k4_cosets = k4env && strcmp(k4env, "cosets") == 0; /* keep mode 4 (one network per coset) where mode 5 applies: A/B */
    if (trivial && e->order > 1 && !k4_general) {
        out->k4_mode = 1;
        /* full cyclic group of the ring (every rotation k = 0..L-1), optionally with all reflections? */
        int const L = b->number_sites;
        uint64_t rot = 0, rev = 0;
        int other = 0;
        for (int g = 0; g < e->order; ++g) {
            if (e->elems[g].kind == LSK_ELEM_ROT) rot |= 1ULL << e->elems[g].k;
            else if (e->elems[g].kind == LSK_ELEM_REVROT) rev |= 1ULL << e->elems[g].k;
            else other = 1;
        }
        uint64_t const full = L >= 64 ? ~0ULL : ((1ULL << L) - 1);
        if (!other && rot == full && (rev == 0 || rev == full) && e->order == L * (rev ? 2 : 1) && L >= 3) {
            out->k4_mode = k4_brute ? 2 : 3; /* 3: longest-zero-run candidate pruning */
            out->reflect = rev ? 1 : 0;
        } else if (!k4_brute) {
            /* a lattice group: its translation subgroup is walked with cheap bit operations, only the coset representatives
             * (the point group) go through compiled networks */
            if (e->tw == 0) find_translation_cosets(e, L);
            if (e->tw > 0) {
                if (!e->d_cosets) {
                    lsk_group_elem *tmp = (lsk_group_elem *)malloc(sizeof(lsk_group_elem) * (size_t)e->n_cosets);
                    for (int r = 0; r < e->n_cosets; ++r) tmp[r] = e->elems[e->coset_ids[r]];
                    void *p = NULL;
                    int const bad = lsk_malloc(&p, sizeof(lsk_group_elem) * (size_t)e->n_cosets) != 0 ||
                                    lsk_h2d(p, tmp, sizeof(lsk_group_elem) * (size_t)e->n_cosets) != 0;
                    free(tmp);
                    if (bad) { if (p) lsk_free(p); return dev_error(); }
                    e->d_cosets = (lsk_group_elem *)p;
                }
                if (!e->d_trow && e->tw <= 8) {
                    uint32_t rowtab[256];
                    void *p = NULL;
                    lsk_torus_rowtab(e->tw, rowtab);
                    if (lsk_malloc(&p, sizeof(rowtab)) != 0 || lsk_h2d(p, rowtab, sizeof(uint32_t) << e->tw) != 0) { if (p) lsk_free(p); return dev_error(); }
                    e->d_trow = (uint32_t *)p;
                }
                out->trow = e->d_trow;
                out->k4_mode = 4;
                out->tw = e->tw;
                out->n_cosets = e->n_cosets;
                out->cosets = e->d_cosets;
                for (int y = 0; y < L / e->tw; ++y) out->tcol0 |= 1ULL << (y * e->tw);
                /* mode 5: the cosets are the point group of the torus -- one network (the transpose) instead of n_cosets */
                if (e->d4_mask == 0) classify_d4_cosets(e, L);
                if (e->d4_mask > 0 && !k4_cosets) {
                    if (!e->d_trow2) {
                        uint64_t rowtab2[256];
                        void *p = NULL;
                        lsk_torus_rowtab2(e->tw, rowtab2);
                        if (lsk_malloc(&p, sizeof(rowtab2)) != 0 || lsk_h2d(p, rowtab2, sizeof(uint64_t) << e->tw) != 0) { if (p) lsk_free(p); return dev_error(); }
                        e->d_trow2 = (uint64_t *)p;
                    }
                    if ((e->d4_mask >> 4) && !e->d_d4_net) {
                        void *p = NULL;
                        if (lsk_malloc(&p, sizeof(lsk_group_elem)) != 0 || lsk_h2d(p, &e->d4_transpose, sizeof(lsk_group_elem)) != 0) { if (p) lsk_free(p); return dev_error(); }
                        e->d_d4_net = (lsk_group_elem *)p;
                    }
                    out->k4_mode = 5;
                    out->d4_mask = e->d4_mask;
                    out->trow2 = e->d_trow2;
                    out->cosets = e->d_d4_net; /* NULL when no image involves the transpose */
                    out->n_cosets = e->d_d4_net ? 1 : 0;
                }
            }
        }
    }
    return 0;
}

/* ============================================================================================ */
/* operator                                                                                     */
/* ============================================================================================ */
struct ls_amd_operator_ext {
    int n_groups;
    lsk_group *groups; /* host */
    lsk_term *off;     /* host, grouped */
    lsk_term *diag;    /* host */
    int n_off, n_diag;
    int is_real, is_hermitian;
    lsk_runs runs;
    /* device mirrors */
    lsk_group *d_groups;
    lsk_term *d_off, *d_diag;
    int adopted; /* foreign struct (ls_amd_adopt_operator): only the side tables are ours */
};

static struct ls_amd_operator_ext g_dummy_operator_ext;
static struct ls_amd_operator_ext *operator_ext_of(ls_hs_operator const *op) {
    struct ls_amd_operator_ext *e = (struct ls_amd_operator_ext *)reg_get(op);
    if (!e) {
        halt_with("ls_hs_operator %p was not created by this library: register it with ls_amd_adopt_operator first", (void const *)op);
        return &g_dummy_operator_ext;
    }
    return e;
}

typedef struct { double re, im; uint64_t m, r, x, s; } raw_term;

static int raw_cmp(void const *pa, void const *pb) {
    raw_term const *a = (raw_term const *)pa, *b = (raw_term const *)pb;
    if (a->x != b->x) return a->x < b->x ? -1 : 1;
    if (a->m != b->m) return a->m < b->m ? -1 : 1;
    if (a->r != b->r) return a->r < b->r ? -1 : 1;
    if (a->s != b->s) return a->s < b->s ? -1 : 1;
    return 0;
}

static void eval_terms(lsk_term const *t, int b, int e, uint64_t a, double *re, double *im) {
    double cr = 0, ci = 0;
    for (int k = b; k < e; ++k)
        if ((a & t[k].m) == t[k].r) {
            int neg = __builtin_popcountll(a & t[k].s) & 1;
            cr += neg ? -t[k].v_re : t[k].v_re;
            ci += neg ? -t[k].v_im : t[k].v_im;
        }
    *re = cr; *im = ci;
}

static uint64_t deposit_bits(uint64_t pattern, uint64_t support) { /* software pdep */
    uint64_t out = 0;
    for (int i = 0; support; ++i) {
        uint64_t low = support & -support;
        if ((pattern >> i) & 1) out |= low;
        support ^= low;
    }
    return out;
}

static void classify_groups(struct ls_amd_operator_ext *ext) {
    ext->is_hermitian = 1;
    /* diagonal must be real for a Hermitian operator */
    for (int k = 0; k < ext->n_diag; ++k) if (ext->diag[k].v_im != 0.0) ext->is_hermitian = 0;
    for (int g = 0; g < ext->n_groups; ++g) {
        lsk_group *G = &ext->groups[g];
        uint64_t support = G->x;
        for (int k = G->begin; k < G->end; ++k) support |= ext->off[k].m | ext->off[k].s;
        int nb = __builtin_popcountll(support);
        G->adj = -1;
        if (__builtin_popcountll(G->x) == 2) {
            int lo = __builtin_ctzll(G->x);
            if (G->x == (3ULL << lo)) G->adj = lo;
        }
        G->fast = LSK_GROUP_GENERIC;
        G->v_re = G->v_im = 0.0;
        if (nb <= 16) {
            /* exact truth table over the support */
            int herm = 1;
            for (uint64_t pat = 0; pat < (1ULL << nb); ++pat) {
                uint64_t a = deposit_bits(pat, support);
                double fr, fi, gr, gi;
                eval_terms(ext->off, G->begin, G->end, a, &fr, &fi);
                eval_terms(ext->off, G->begin, G->end, a ^ G->x, &gr, &gi);
                if (fabs(gr - fr) > 1e-12 * (1 + fabs(fr)) || fabs(gi + fi) > 1e-12 * (1 + fabs(fi))) herm = 0;
            }
            if (!herm) ext->is_hermitian = 0;
            if (nb == 2 && support == G->x) {
                uint64_t b0 = G->x & -G->x, b1 = G->x ^ b0;
                double f00r, f00i, f01r, f01i, f10r, f10i, f11r, f11i;
                eval_terms(ext->off, G->begin, G->end, 0, &f00r, &f00i);
                eval_terms(ext->off, G->begin, G->end, b0, &f01r, &f01i);
                eval_terms(ext->off, G->begin, G->end, b1, &f10r, &f10i);
                eval_terms(ext->off, G->begin, G->end, b0 | b1, &f11r, &f11i);
                if (f00r == 0 && f00i == 0 && f11r == 0 && f11i == 0 && f01r == f10r && f01i == f10i) {
                    G->fast = LSK_GROUP_EXCHANGE;
                    G->v_re = f01r;
                    G->v_im = f01i;
                } else if (f00r == 0 && f00i == 0 && f11r == 0 && f11i == 0 && ((f01r == 0 && f01i == 0) != (f10r == 0 && f10i == 0))) {
                    /* a directed pair (sigma^+ sigma^-): one of the two patterns alone is a source */
                    int const lo_src = !(f01r == 0 && f01i == 0); /* alpha = the pair's lower site alone has a coefficient */
                    G->fast = lo_src ? LSK_GROUP_HOP_LO : LSK_GROUP_HOP_HI;
                    G->v_re = lo_src ? f01r : f10r;
                    G->v_im = lo_src ? f01i : f10i;
                }
            }
        } else {
            /* sampled check (fixed seed) */
            uint64_t z = 0x9e3779b97f4a7c15ULL;
            for (int it = 0; it < 4096; ++it) {
                z = ls_amd_hash64_01(z + 0x9e3779b97f4a7c15ULL * (uint64_t)(it + 1));
                double fr, fi, gr, gi;
                eval_terms(ext->off, G->begin, G->end, z, &fr, &fi);
                eval_terms(ext->off, G->begin, G->end, z ^ G->x, &gr, &gi);
                if (fabs(gr - fr) > 1e-12 * (1 + fabs(fr)) || fabs(gi + fi) > 1e-12 * (1 + fabs(fi))) { ext->is_hermitian = 0; break; }
            }
        }
    }
}

typedef struct { int lo; int index; double re, im; int kind; } run_item;
static int run_item_cmp(void const *pa, void const *pb) {
    run_item const *a = (run_item const *)pa, *b = (run_item const *)pb;
    return a->lo < b->lo ? -1 : (a->lo > b->lo ? 1 : 0);
}

/* Recognise exchange runs among the groups and zz runs among the diagonal terms (lsk.h) and move
 * them to the front of their arrays.  `number_sites`/`inversion`: runs must not touch the top site
 * of a spin-inversion basis, where the image state may need the global flip. */
static void detect_runs(struct ls_amd_operator_ext *ext, int number_sites, int inversion) {
    lsk_runs *R = &ext->runs;
    memset(R, 0, sizeof(*R));
    /* ---- off-diagonal exchange runs ---- */
    int ng = ext->n_groups;
    run_item *items = (run_item *)malloc(sizeof(run_item) * (ng > 0 ? ng : 1));
    int ni = 0;
    for (int g = 0; g < ng; ++g) {
        lsk_group const *G = &ext->groups[g];
        if (G->fast == LSK_GROUP_GENERIC || G->adj < 0) continue; /* exchange pairs and directed pairs (HOP_*) on adjacent sites */
        if (inversion && G->fast != LSK_GROUP_EXCHANGE) continue;  /* (directed runs: unprojected bases only -- k_direct's DIRECTED instantiation) */
        if (inversion && G->adj + 1 >= number_sites - 1) continue;
        run_item it = {G->adj, g, G->v_re, G->v_im, G->fast};
        items[ni++] = it;
    }
    qsort(items, ni, sizeof(run_item), run_item_cmp);
    char *taken = (char *)calloc(ng > 0 ? ng : 1, 1);
    lsk_group *reordered = (lsk_group *)malloc(sizeof(lsk_group) * (ng > 0 ? ng : 1));
    int w = 0;
    for (int i = 0; i < ni && R->n_runs < LSK_MAX_RUNS;) {
        int j = i + 1;
        while (j < ni && items[j].lo == items[j - 1].lo + 1 && items[j].re == items[i].re && items[j].im == items[i].im && items[j].kind == items[i].kind) ++j;
        if (j - i >= 2) {
            int r = R->n_runs++;
            int const dir = items[i].kind == LSK_GROUP_HOP_LO ? 1 : (items[i].kind == LSK_GROUP_HOP_HI ? 2 : 0);
            R->lo0[r] = items[i].lo; R->cnt[r] = (j - i) | (dir << 16); R->v_re[r] = items[i].re; R->v_im[r] = items[i].im;
            for (int k = i; k < j; ++k) { reordered[w++] = ext->groups[items[k].index]; taken[items[k].index] = 1; }
        }
        i = j;
    }
    R->n_run_groups = w;
    for (int g = 0; g < ng; ++g) if (!taken[g]) reordered[w++] = ext->groups[g];
    memcpy(ext->groups, reordered, sizeof(lsk_group) * ng);
    free(items); free(taken); free(reordered);
    /* ---- diagonal zz runs ---- */
    int nd = ext->n_diag;
    items = (run_item *)malloc(sizeof(run_item) * (nd > 0 ? nd : 1));
    ni = 0;
    for (int t = 0; t < nd; ++t) {
        lsk_term const *T = &ext->diag[t];
        if (T->m != 0 || T->r != 0 || T->v_im != 0.0 || __builtin_popcountll(T->s) != 2) continue;
        int lo = __builtin_ctzll(T->s);
        if (T->s != (3ULL << lo)) continue;
        run_item it = {lo, t, T->v_re, 0.0, 0};
        items[ni++] = it;
    }
    qsort(items, ni, sizeof(run_item), run_item_cmp);
    taken = (char *)calloc(nd > 0 ? nd : 1, 1);
    lsk_term *dre = (lsk_term *)malloc(sizeof(lsk_term) * (nd > 0 ? nd : 1));
    w = 0;
    for (int i = 0; i < ni && R->n_zz < LSK_MAX_RUNS;) {
        int j = i + 1;
        while (j < ni && items[j].lo == items[j - 1].lo + 1 && items[j].re == items[i].re) ++j;
        if (j - i >= 2) {
            int r = R->n_zz++;
            R->zz_lo0[r] = items[i].lo; R->zz_cnt[r] = j - i; R->zz_v[r] = items[i].re;
            for (int k = i; k < j; ++k) { dre[w++] = ext->diag[items[k].index]; taken[items[k].index] = 1; }
        }
        i = j;
    }
    R->n_zz_terms = w;
    for (int t = 0; t < nd; ++t) if (!taken[t]) dre[w++] = ext->diag[t];
    memcpy(ext->diag, dre, sizeof(lsk_term) * nd);
    free(items); free(taken); free(dre);
}

static ls_hs_nonbranching_terms *make_nbt(lsk_term const *t, uint64_t const *xs, int n, int nbits) {
    if (n == 0) return NULL; /* the reference tests `p == nil` (ForeignTypes.chpl:219-222) */
    ls_hs_nonbranching_terms *nb = (ls_hs_nonbranching_terms *)calloc(1, sizeof(*nb));
    ls_hs_scalar *v = (ls_hs_scalar *)malloc(sizeof(ls_hs_scalar) * n);
    uint64_t *m = (uint64_t *)malloc(8 * n), *l = (uint64_t *)malloc(8 * n), *r = (uint64_t *)malloc(8 * n),
             *x = (uint64_t *)malloc(8 * n), *s = (uint64_t *)malloc(8 * n);
    for (int i = 0; i < n; ++i) {
        v[i].re = t[i].v_re; v[i].im = t[i].v_im;
        m[i] = t[i].m; r[i] = t[i].r; x[i] = xs ? xs[i] : 0; s[i] = t[i].s;
        l[i] = t[i].r ^ (x[i] & t[i].m);
    }
    nb->number_terms = n; nb->number_bits = nbits;
    nb->v = v; nb->m = m; nb->l = l; nb->r = r; nb->x = x; nb->s = s;
    return nb;
}
static void free_nbt(ls_hs_nonbranching_terms *nb) {
    if (!nb) return;
    free((void *)nb->v); free((void *)nb->m); free((void *)nb->l); free((void *)nb->r);
    free((void *)nb->x); free((void *)nb->s); free(nb);
}

/* term tables of an operator: filter / normalise / merge the raw terms, split them into diagonal and off-diagonal ones,
 * group the latter by flip mask, classify.  *offx_out (malloc'ed, n_off entries): flip mask of every off-diagonal term. */
static struct ls_amd_operator_ext *build_operator_ext(int number_sites, int has_inversion, int number_terms, double const *v,
                                                      uint64_t const *m, uint64_t const *r, uint64_t const *x,
                                                      uint64_t const *s, uint64_t **offx_out) {
    uint64_t const site_mask = number_sites >= 64 ? ~0ULL : ((1ULL << number_sites) - 1);
    raw_term *raw = (raw_term *)malloc(sizeof(raw_term) * (number_terms > 0 ? number_terms : 1));
    int n = 0;
    double vmax = 0;
    for (int i = 0; i < number_terms; ++i) {
        if ((m[i] | r[i] | x[i] | s[i]) & ~site_mask) { free(raw); set_error("term %d touches sites outside the basis", i); return NULL; }
        if (r[i] & ~m[i]) continue; /* can never be active */
        raw_term t = {v[2 * i], v[2 * i + 1], m[i], r[i], x[i], s[i]};
        /* sign bits inside the projector have a fixed value when the term is active */
        if (__builtin_popcountll(t.r & t.s) & 1) { t.re = -t.re; t.im = -t.im; }
        t.s &= ~t.m;
        raw[n++] = t;
        double a = fabs(t.re) + fabs(t.im);
        if (a > vmax) vmax = a;
    }
    qsort(raw, n, sizeof(raw_term), raw_cmp);
    /* merge equal keys */
    int w = 0;
    for (int i = 0; i < n;) {
        raw_term acc = raw[i];
        int j = i + 1;
        while (j < n && raw_cmp(&raw[i], &raw[j]) == 0) { acc.re += raw[j].re; acc.im += raw[j].im; ++j; }
        if (fabs(acc.re) + fabs(acc.im) > 1e-15 * vmax) raw[w++] = acc;
        i = j;
    }
    n = w;
    struct ls_amd_operator_ext *ext = (struct ls_amd_operator_ext *)calloc(1, sizeof(*ext));
    int nd = 0;
    while (nd < n && raw[nd].x == 0) ++nd; /* sorted by x: the diagonal terms come first */
    ext->n_diag = nd;
    ext->n_off = n - nd;
    ext->diag = (lsk_term *)malloc(sizeof(lsk_term) * (nd > 0 ? nd : 1));
    ext->off = (lsk_term *)malloc(sizeof(lsk_term) * (ext->n_off > 0 ? ext->n_off : 1));
    uint64_t *offx = (uint64_t *)malloc(8 * (ext->n_off > 0 ? ext->n_off : 1));
    ext->is_real = 1;
    for (int i = 0; i < n; ++i) {
        lsk_term t = {raw[i].re, raw[i].im, raw[i].m, raw[i].r, raw[i].s};
        if (raw[i].im != 0.0) ext->is_real = 0;
        if (i < nd) ext->diag[i] = t; else { ext->off[i - nd] = t; offx[i - nd] = raw[i].x; }
    }
    ext->groups = (lsk_group *)malloc(sizeof(lsk_group) * (ext->n_off > 0 ? ext->n_off : 1));
    ext->n_groups = 0;
    for (int i = 0; i < ext->n_off; ++i) {
        if (i == 0 || offx[i] != offx[i - 1]) {
            lsk_group G;
            memset(&G, 0, sizeof(G));
            G.x = offx[i];
            G.begin = i;
            ext->groups[ext->n_groups++] = G;
        }
        ext->groups[ext->n_groups - 1].end = i + 1;
    }
    classify_groups(ext);
    detect_runs(ext, number_sites, has_inversion);
    free(raw);
    if (offx_out) *offx_out = offx; else free(offx);
    return ext;
}

/* The operator SHARES its basis (one more reference), as upstream's ls_hs_operator does: representatives set or built
 * on the basis after the operator exists are seen by the operator, and nothing is duplicated (the representatives of
 * heisenberg_chain_32 are 4.8 GB of host memory). */
ls_hs_operator *ls_hs_create_operator_from_terms(ls_hs_basis const *basis, int number_terms,
                                                 double const *v, uint64_t const *m, uint64_t const *r,
                                                 uint64_t const *x, uint64_t const *s) {
    struct ls_amd_basis_ext *be = (struct ls_amd_basis_ext *)reg_get(basis);
    if (!be) { set_error("unknown basis: create it with ls_hs_create_spin_basis or register it with ls_amd_adopt_basis"); return NULL; }
    uint64_t *offx = NULL;
    struct ls_amd_operator_ext *ext = build_operator_ext(basis->number_sites, basis->spin_inversion != 0, number_terms, v, m, r,
                                                         x, s, &offx);
    if (!ext) return NULL;
    ls_hs_operator *op = (ls_hs_operator *)calloc(1, sizeof(*op));
    reg_put(op, ext, REG_OPERATOR);
    op->basis = (ls_hs_basis *)basis;
    __atomic_add_fetch(&be->refcount, 1, __ATOMIC_RELAXED);
    op->diag_terms = make_nbt(ext->diag, NULL, ext->n_diag, basis->number_sites);
    op->off_diag_terms = make_nbt(ext->off, offx, ext->n_off, basis->number_sites);
    free(offx);
    return op;
}

/* ---- foreign objects ------------------------------------------------------------------------ */
/* A basis struct somebody else owns (e.g. built by lattice-symmetries-haskell): only the reference's prefix is read
 * (number_sites, number_up = Hamming weight or -1, spin_inversion, representatives; FFI.chpl:94-105).  The symmetry
 * group is not part of that prefix, so the caller passes the generators it built the basis from (the YAML's
 * `symmetries`), in the convention of ls_hs_create_spin_basis. */
int ls_amd_adopt_basis(ls_hs_basis const *basis, int number_generators, int const *permutations, int const *sectors) {
    if (!basis) return set_error("null basis");
    if (reg_get(basis)) return set_error("basis %p is already registered", (void const *)basis);
    if (basis->particle_type != LS_HS_SPIN) return set_error("only spin bases are supported (DistributedMatrixVector.chpl works on spin configs)");
    ls_hs_basis *tmp = ls_hs_create_spin_basis(basis->number_sites, basis->number_up >= 0 ? basis->number_up : -1, basis->spin_inversion,
                                               number_generators, permutations, sectors);
    if (!tmp) return -1;
    if (tmp->requires_projection != basis->requires_projection) {
        ls_hs_destroy_basis(tmp);
        return set_error("generators do not reproduce the basis: requires_projection differs");
    }
    /* move the private part over to the foreign address */
    struct ls_amd_basis_ext *e = (struct ls_amd_basis_ext *)reg_get(tmp);
    reg_del(tmp);
    free(tmp);
    e->adopted = 1;
    e->owns_representatives = 0;
    reg_put(basis, e, REG_BASIS);
    return 0;
}
/* An operator struct somebody else owns: the term tables are rebuilt from its off_diag_terms / diag_terms
 * (ls_hs_nonbranching_terms, number_words == 1) and kept in the side table. */
int ls_amd_adopt_operator(ls_hs_operator const *op) {
    if (!op || !op->basis) return set_error("null operator");
    if (reg_get(op)) return set_error("operator %p is already registered", (void const *)op);
    if (!reg_get(op->basis)) return set_error("the operator's basis is unknown: ls_amd_adopt_basis first");
    ls_hs_nonbranching_terms const *parts[2] = {op->diag_terms, op->off_diag_terms};
    int n = 0;
    for (int k = 0; k < 2; ++k)
        if (parts[k]) {
            if (parts[k]->number_bits > 64) return set_error("bases with more than 64 bits are not yet implemented"); /* DMV:1099 */
            n += parts[k]->number_terms;
        }
    double *v = (double *)malloc(16 * (size_t)(n > 0 ? n : 1));
    uint64_t *m = (uint64_t *)malloc(8 * (size_t)(n > 0 ? n : 1)), *r = (uint64_t *)malloc(8 * (size_t)(n > 0 ? n : 1)),
             *x = (uint64_t *)malloc(8 * (size_t)(n > 0 ? n : 1)), *s = (uint64_t *)malloc(8 * (size_t)(n > 0 ? n : 1));
    int k = 0;
    for (int q = 0; q < 2; ++q)
        for (int i = 0; parts[q] && i < parts[q]->number_terms; ++i, ++k) {
            v[2 * k] = parts[q]->v[i].re; v[2 * k + 1] = parts[q]->v[i].im;
            m[k] = parts[q]->m[i]; r[k] = parts[q]->r[i]; x[k] = parts[q]->x[i]; s[k] = parts[q]->s[i];
        }
    struct ls_amd_operator_ext *ext = build_operator_ext(op->basis->number_sites, op->basis->spin_inversion != 0, n, v, m, r, x, s, NULL);
    free(v); free(m); free(r); free(x); free(s);
    if (!ext) return -1;
    ext->adopted = 1;
    reg_put(op, ext, REG_OPERATOR);
    __atomic_add_fetch(&BEXT(op->basis)->refcount, 1, __ATOMIC_RELAXED);
    return 0;
}
/* forget an adopted basis / operator (device tables are released; the foreign struct is not touched) */
void ls_amd_release(void const *object) {
    int kind = 0;
    if (!reg_find(object, &kind)) return;
    if (kind == REG_BASIS) ls_hs_destroy_basis((ls_hs_basis *)object);
    else ls_hs_destroy_operator((ls_hs_operator *)object);
}

ls_hs_operator *ls_hs_clone_operator(ls_hs_operator const *op) {
    struct ls_amd_operator_ext const *e = OEXT(op);
    int n = e->n_diag + e->n_off;
    double *v = (double *)malloc(16 * (n > 0 ? n : 1));
    uint64_t *m = (uint64_t *)malloc(8 * (n > 0 ? n : 1)), *r = (uint64_t *)malloc(8 * (n > 0 ? n : 1)),
             *x = (uint64_t *)malloc(8 * (n > 0 ? n : 1)), *s = (uint64_t *)malloc(8 * (n > 0 ? n : 1));
    int k = 0;
    for (int i = 0; i < e->n_diag; ++i, ++k) {
        v[2 * k] = e->diag[i].v_re; v[2 * k + 1] = e->diag[i].v_im;
        m[k] = e->diag[i].m; r[k] = e->diag[i].r; x[k] = 0; s[k] = e->diag[i].s;
    }
    for (int g = 0; g < e->n_groups; ++g)
        for (int i = e->groups[g].begin; i < e->groups[g].end; ++i, ++k) {
            v[2 * k] = e->off[i].v_re; v[2 * k + 1] = e->off[i].v_im;
            m[k] = e->off[i].m; r[k] = e->off[i].r; x[k] = e->groups[g].x; s[k] = e->off[i].s;
        }
    ls_hs_operator *c = ls_hs_create_operator_from_terms(op->basis, n, v, m, r, x, s);
    free(v); free(m); free(r); free(x); free(s);
    return c;
}

void ls_hs_destroy_operator(ls_hs_operator *op) {
    if (!op) return;
    struct ls_amd_operator_ext *e = (struct ls_amd_operator_ext *)reg_get(op);
    if (!e) return;
    { /* plans cached for this operator hold its device tables (and its address may be reused by the next operator) */
        struct ls_amd_basis_ext *be = op->basis ? (struct ls_amd_basis_ext *)reg_get(op->basis) : NULL;
        if (be) basis_drop_host_plans(be, op, NULL);
    }
    if (e->d_groups) lsk_free(e->d_groups);
    if (e->d_off) lsk_free(e->d_off);
    if (e->d_diag) lsk_free(e->d_diag);
    free(e->groups); free(e->off); free(e->diag);
    reg_del(op);
    int const adopted = e->adopted;
    free(e);
    ls_hs_destroy_basis(op->basis); /* drops the operator's reference */
    if (adopted) return;
    free_nbt(op->diag_terms);
    free_nbt(op->off_diag_terms);
    free(op);
}

int ls_hs_operator_max_number_off_diag(ls_hs_operator const *op) { return OEXT(op)->n_groups; }
bool ls_hs_operator_is_hermitian(ls_hs_operator const *op) { return OEXT(op)->is_hermitian != 0; }
bool ls_hs_operator_is_real(ls_hs_operator const *op) { return OEXT(op)->is_real != 0; }

static int operator_device_unlocked(ls_hs_operator const *op, lsk_operator *out) {
    struct ls_amd_operator_ext *e = OEXT(op);
    if (!e->d_groups) {
        void *p;
        DEV(lsk_malloc(&p, sizeof(lsk_group) * (e->n_groups > 0 ? e->n_groups : 1)));
        DEV(lsk_h2d(p, e->groups, sizeof(lsk_group) * e->n_groups));
        e->d_groups = (lsk_group *)p;
        DEV(lsk_malloc(&p, sizeof(lsk_term) * (e->n_off > 0 ? e->n_off : 1)));
        DEV(lsk_h2d(p, e->off, sizeof(lsk_term) * e->n_off));
        e->d_off = (lsk_term *)p;
        DEV(lsk_malloc(&p, sizeof(lsk_term) * (e->n_diag > 0 ? e->n_diag : 1)));
        DEV(lsk_h2d(p, e->diag, sizeof(lsk_term) * e->n_diag));
        e->d_diag = (lsk_term *)p;
    }
    out->n_diag = e->n_diag;
    out->n_off = e->n_off;
    out->n_groups = e->n_groups;
    out->is_real = e->is_real;
    out->diag = e->d_diag;
    out->off = e->d_off;
    out->groups = e->d_groups;
    out->runs = e->runs;
    /* one real amplitude on every off-diagonal group, all of them exchange pairs: the projected pull kernels then carry no
     * coefficient per packet */
    out->uni = e->is_real && e->n_groups > 0;
    out->uni_v = e->n_groups > 0 ? e->groups[0].v_re : 0.0;
    for (int g = 0; g < e->n_groups && out->uni; ++g)
        if (e->groups[g].fast != LSK_GROUP_EXCHANGE || e->groups[g].v_im != 0.0 || e->groups[g].v_re != out->uni_v ||
            __builtin_popcountll(e->groups[g].x) != 2)
            out->uni = 0;
    return 0;
}

/* the device mirrors of a basis / an operator are uploaded on first use; several host threads may create plans on the same
 * objects at once (one communicator per thread) */
static int operator_device(ls_hs_operator const *op, lsk_operator *out) {
    pthread_mutex_lock(&g_device_tables_lock);
    int const rc = operator_device_unlocked(op, out);
    pthread_mutex_unlock(&g_device_tables_lock);
    return rc;
}
static int basis_device(ls_hs_basis const *b, lsk_basis *out) {
    pthread_mutex_lock(&g_device_tables_lock);
    int const rc = basis_device_unlocked(b, out);
    pthread_mutex_unlock(&g_device_tables_lock);
    return rc;
}

/* ============================================================================================ */
/* kernel table                                                                                 */
/* ============================================================================================ */
static ls_chpl_kernels g_kernels;
void ls_hs_internal_set_chpl_kernels(ls_chpl_kernels const *kernels) { g_kernels = *kernels; }
ls_chpl_kernels const *ls_hs_internal_get_chpl_kernels(void) { return &g_kernels; }

void ls_chpl_init_kernels(void) {
    ls_chpl_kernels k;
    k.enumerate_states = (void *)ls_chpl_enumerate_representatives;
    k.operator_apply_off_diag = (void *)ls_chpl_operator_apply_off_diag;
    k.operator_apply_diag = (void *)ls_chpl_operator_apply_diag;
    k.matrix_vector_product = (void *)ls_chpl_matrix_vector_product;
    ls_hs_internal_set_chpl_kernels(&k);
}

static int g_initialised = 0;
void ls_chpl_init(void) {
    if (g_initialised) return;
    if (lsk_device_count() < 1) { halt_with("ls_chpl_init: no HIP device available"); return; }
    char const *dev = getenv("LS_AMD_DEVICE");
    if (dev && lsk_set_device(atoi(dev)) != 0) { halt_with("ls_chpl_init: %s", lsk_last_error()); return; }
    ls_chpl_init_kernels();
    g_initialised = 1;
}
void ls_chpl_finalize(void) {
    if (g_d_binom) { lsk_free(g_d_binom); g_d_binom = NULL; }
    g_initialised = 0;
}

void ls_hs_basis_build(ls_hs_basis *basis) {
    if (basis->representatives.elts) return;
    typedef void (*enum_fn)(ls_hs_basis *, uint64_t, uint64_t, chpl_external_array *);
    enum_fn f = (enum_fn)g_kernels.enumerate_states;
    if (!f) { ls_chpl_init_kernels(); f = (enum_fn)g_kernels.enumerate_states; }
    chpl_external_array arr = {NULL, 0, NULL};
    f(basis, ls_hs_min_state_estimate(basis), ls_hs_max_state_estimate(basis), &arr);
    if (!arr.elts) return;
    basis_drop_device_caches(basis);
    basis->representatives = arr;
    BEXT(basis)->owns_representatives = 1;
}

/* ============================================================================================ */
/* plans                                                                                        */
/* ============================================================================================ */
typedef struct part_state {
    int64_t count;
    uint64_t const *d_reps; /* borrowed */
    lsk_index index;
    uint32_t *d_table;      /* owned */
    lsk_rankdir *d_dir;     /* owned: rank directory (hash partitions of unprojected fixed-weight bases), else NULL */
    double *d_norms;        /* owned (FULL projection) */
    int rounds;
    int64_t *send_counts;   /* [rounds][P] host */
    lsk_round_layout *d_layouts; /* [rounds] device */
    int64_t max_send_bytes;
    int64_t *h_beta_off, *h_val_off; /* [rounds][P] host copies of the layout */
    /* packet producer with per-wave rings (lsk_tile_wv, P <= 64): position of every (wave of 64 rows, destination) inside the
     * round's send segments, fixed by the plan's count pass -- [sum over rounds of waves][P] u32 */
    uint32_t *d_wtab;     /* owned */
    int64_t *wtab_first;  /* [rounds] first wave of the round in d_wtab */
    /* sorted packet streams (lsk_tile_st / lsk_window): position of every (tile, destination, stream) inside the destination's
     * segment -- [sum over rounds of tiles][P * S] u32 -- and the stream starts of every segment, [rounds][P][S + 1] */
    uint32_t *d_ttab;     /* owned */
    int64_t *ttab_first;  /* [rounds + 1] first tile of the round in d_ttab */
    uint32_t *d_soff;     /* owned */
    uint32_t *h_soff;     /* owned: host copy (dist.c exchanges it at set-up) */
    struct ls_amd_gtab *scatter_gt; /* shared (refcount): {representative -> index} for the consumers of state-carrying packets, or NULL */
} part_state;

#ifndef LS_AMD_PULL_VALUES_DEFAULT
#define LS_AMD_PULL_VALUES_DEFAULT 0 /* flipped to 1 where the value table measures faster INCLUDING its refresh */
#endif
enum { FAMILY_DIRECT_PUSH = 0, FAMILY_DIRECT_PULL = 1, FAMILY_TILE = 2, FAMILY_TILE_PULL = 3,
       FAMILY_REPL_DIRECT = 4, FAMILY_REPL_TILE = 5 };

struct ls_amd_plan {
    ls_hs_operator const *op;
    lsk_operator dop;
    lsk_basis dbs;
    int cplx, P, me, n_local;
    int family;
    part_state *parts; /* [n_local] */
    void *d_send;      /* local mode: staging for one round */
    int64_t send_capacity;
    unsigned long long *d_cursors; /* [P] */
    unsigned long long *d_counts;  /* [P] */
    int *d_err;
    int64_t nnz;
    /* pre-indexed packets (packet plans over unprojected fixed-weight bases): all-destinations rank directory; the producer
     * writes (u32 index at the destination, value) and the consumers are search-free (lsk_gdir, lsk.h) */
    lsk_gdir gd;
    lsk_rankdir *d_gdir; /* owned */
    lsk_part_ctx *d_part_ctx; /* owned: all partitions in this process -- index / norms of every destination for the fused consumer */
    int part_ctx_any;         /* a partition whose index carries a rank directory (else 0) */
    int key_bytes;       /* 4: pre-indexed packets, 8: packets carry the state */
    /* sorted packet streams (all partitions in this process; unprojected fixed-weight bases, exchange operators): every source
     * partition keeps its round in its own buffer and ONE consumer launch per round adds windows of every y in LDS -- no atomics */
    int streams, st_S, st_tile_rows, st_wpb, st_rounds;
    int stream_gkeys;        /* the keys of the streams are global colex ranks; every partition then carries its own rank directory */
    void **d_send_parts;     /* [P] owned */
    lsk_wsrc *d_wsrcs;       /* owned: device [rounds][P destinations][P sources] */
    /* replicated-x mode: index / norms of the GLOBAL basis, global index of every local row */
    lsk_index gindex;
    uint32_t *d_gtable;
    int64_t *d_row_gidx;
    double *d_norms_global;
    /* tile map of the row kernels (lsk_tilemap in lsk.h) */
    lsk_tilemap tilemap;
    void *d_tilemap;
    int has_pairs; /* staged row kernel for arbitrary exchange pairs (lsk_pairs): non-ring lattices */
    int has_push_staged; /* staged push kernel (lsk_push_staged): LDS window of y per tile; y is cleared, not assigned by k_diag */
    int one_rank_full_basis; /* FAMILY_TILE with one rank that owns the full fixed-weight basis (plan_setup_part) */
    lsk_pairplan pairs;
    void *d_pair_recs, *d_rank_low, *d_pair_binom, *d_states32, *d_pair_rows, *d_pair_sites;
    int has_chain; /* staged row kernel (lsk_chain) */
    int chain_cached;      /* leading non-adjacent exchange groups whose partner ranks are cached */
    void *d_chain_cache;   /* [chain_cached][count] u32, or u64 when chain_wide */
    int chain_wide;        /* 64-bit ranks (>= 2^32 - 1 states) */
    uint64_t *d_chain_rec; /* fused records (lsk_chain_pack) replacing reps + the first cached pair */
    double chain_v[2];
    int64_t chain_row0; /* global rank of the first local row (replicated-x plans) */
    void *d_htab;        /* hash table {rep -> x * norm(rep)} of the tile-pull families */
    uint32_t *d_slot_of; /* slot of every (global) representative */
    int htab_bits;
    void *d_xs;          /* x * norm in index order (K4 modes that prescale x): values of the near window; else x itself */
    int pull_halo;       /* near window of the staged pull kernel (entries either side of a tile), 0 = off */
    /* indexed mode of the staged pull kernel (lsk_tile_pull_idx): static {rep -> slot} table shared by every plan over the
     * same global basis and partition layout; nothing is refreshed per matvec */
    int idx_mode;
    struct ls_amd_gtab *gtab;
    uint64_t *d_vtab;    /* owned: value table over the same buckets (f64, one partition, matrix-free), else NULL */
    int64_t row_g0;      /* global index of the first local row (replicated-x plans over a contiguous block) */
    /* split matvec of the indexed mode (lsk_tile_pull_resolve | lsk_tile_pull_gather): packet streams of the first
     * split_rows rows (a multiple of 256, or all of them); 0 = the fused kernel only */
    lsk_pullbuf pbuf;
    int64_t split_rows;
    int64_t split_active; /* rows [0, split_active) are resolved ahead this matvec (<= split_rows, whole 256-row tiles or all of
                           * them; 0 = follow split_rows): the replicated-x driver resolves only as many rows as hide its exchange */
    /* slot cache (ls_amd_plan_cache_slots): the packet streams of the split form are plan data -- they depend on the operator,
     * the basis and the partition layout only -- so a plan that is applied many times (an eigensolver) resolves them ONCE and
     * every later matvec is the gather kernel alone.  Opt-in: the path is then no longer matrix-free (5 bytes per non-zero of
     * HBM for one-amplitude operators, 13 / 21 with real / complex coefficients). */
    int slot_cache, slot_cache_valid;
    int64_t slot_cache_bytes;
    void const *y_checked[2]; /* y pointers whose memory kind was looked at (push plans: ls_amd_internal_check_y) */
    int leaves_basis;         /* pull plan over a non-Hermitian operator whose row expansion leaves the basis (found at plan time) */
    /* kernel timing ring */
    int t_capacity, t_count;
    void **t_start, **t_stop;
    /* stage timers (the reference's kDisplayTimings tree, DMV:1028-1052): event pairs tagged with a stage */
    int st_capacity, st_count;
    void **st_start, **st_stop;
    unsigned char *st_stage;
    double st_ms[8];
    int64_t st_calls[8];
    int64_t st_matvecs;
};

/* ---- static index tables (lsk_gtab) ---------------------------------------------------------------------------------
 * One table per (global basis, partition layout), shared by reference: every plan of a process over the same device array
 * of representatives and the same masks gets the same table and the same global-row -> slot permutation (the loop-back
 * ranks of the tests and of scripts/loopback_bench.py are threads of one process: eight private copies of a 17 GB table
 * would not fit one device; separate processes each build their own). */
typedef struct ls_amd_gtab {
    uint64_t const *reps; /* key: device array of the global representatives (borrowed) */
    uint8_t const *masks; /* key: owner of every global row (NULL: one partition, slot = global index) */
    int64_t n;
    int P, L;
    lsk_gtab tab;
    uint64_t *d_entries;
    uint32_t *d_perm;     /* [n] global row -> slot owner * max_count + local index (NULL when masks == NULL) */
    int64_t counts[LSK_MAX_PARTS], max_count;
    uint64_t finger[8];   /* key, part two: the representatives at eight sampled positions -- a caller that frees the array and
                           * gets the same address back for another basis of the same size (caching allocators do that) must not
                           * be handed the old table */
    int refs;
    struct ls_amd_gtab *next;
} ls_amd_gtab;
static ls_amd_gtab *g_gtabs = NULL;
static lsk_gtab no_gtab(void) { lsk_gtab t; memset(&t, 0, sizeof(t)); return t; }
static pthread_mutex_t g_gtab_lock = PTHREAD_MUTEX_INITIALIZER;

static void gtab_free(ls_amd_gtab *t) {
    if (t->d_entries) lsk_free(t->d_entries);
    if (t->d_perm) lsk_free(t->d_perm);
    free(t);
}
void ls_amd_internal_gtab_release(ls_amd_gtab *t) {
    if (!t) return;
    pthread_mutex_lock(&g_gtab_lock);
    if (--t->refs == 0) {
        for (ls_amd_gtab **q = &g_gtabs; *q; q = &(*q)->next)
            if (*q == t) { *q = t->next; break; }
        gtab_free(t);
    }
    pthread_mutex_unlock(&g_gtab_lock);
}
static int gtab_build(ls_amd_gtab *t, void *stream) {
    int64_t const n = t->n;
    int const P = t->P;
    if (t->masks) {
        /* slot of every global row: the hashed -> block merge of the slot numbers (arrFromHashedToBlock, HashedToBlock.chpl:67-153) */
        DEV(lsk_mask_counts(n, t->masks, P, t->counts, stream));
        for (int p = 0; p < P; ++p) if (t->counts[p] > t->max_count) t->max_count = t->counts[p];
        if ((int64_t)P * t->max_count >= 0xffffffffLL) return set_error("indexed pull: more than 2^32 - 1 slots");
        void *pos[LSK_MAX_PARTS];
        memset(pos, 0, sizeof(pos));
        void *perm64 = NULL, *perm32 = NULL;
        int rc = lsk_malloc(&perm64, 8 * (size_t)(n > 0 ? n : 1));
        for (int p = 0; p < P && rc == 0; ++p) {
            rc = lsk_malloc(&pos[p], 8 * (size_t)(t->counts[p] > 0 ? t->counts[p] : 1));
            if (rc == 0) rc = lsk_iota_i64(t->counts[p], (int64_t)p * t->max_count, (int64_t *)pos[p], stream);
        }
        if (rc == 0) rc = lsk_hashed_to_block(n, t->masks, P, 8, (void const *const *)pos, perm64, stream);
        if (rc == 0) rc = lsk_malloc(&perm32, 4 * (size_t)(n > 0 ? n : 1));
        if (rc == 0) rc = lsk_narrow_i32(n, (int64_t const *)perm64, (int32_t *)perm32, stream);
        if (rc == 0) rc = lsk_sync(stream);
        for (int p = 0; p < P; ++p) if (pos[p]) lsk_free(pos[p]);
        if (perm64) lsk_free(perm64);
        if (rc != 0) { if (perm32) lsk_free(perm32); return dev_error(); }
        t->d_perm = (uint32_t *)perm32;
    } else {
        t->counts[0] = n;
        t->max_count = n;
    }
    int64_t const max_bytes = (int64_t)1 << 40;
    int const bb = lsk_gtab_bits(t->L, n, max_bytes);
    if (bb < 0) return set_error("indexed pull: no admissible index-table size for %lld keys of %d bits", (long long)n, t->L);
    t->tab.L = t->L; t->tab.bbits = bb; t->tab.tbits = t->L - bb;
    void *p, *flag;
    DEV(lsk_malloc(&p, (size_t)16 << bb));
    t->d_entries = (uint64_t *)p;
    t->tab.entries = t->d_entries;
    DEV(lsk_malloc(&flag, sizeof(int)));
    int zero = 0, bad = 0;
    int rc = lsk_h2d(flag, &zero, sizeof(int)) || lsk_gtab_build(t->tab, t->d_entries, n, t->reps, t->d_perm, (int *)flag, stream) ||
             lsk_sync(stream) || lsk_d2h(&bad, flag, sizeof(int));
    lsk_free(flag);
    if (rc) return dev_error();
    if (bad) return set_error("indexed pull: a key could not be placed within 255 buckets of its home");
    return 0;
}
/* the table of (d_reps, n, d_masks, P), built on first use.  Holds g_gtab_lock while building: the other ranks of a
 * loop-back group wait for the one that got there first. */
static int gtab_fingerprint(uint64_t const *d_reps, int64_t n, void *stream, uint64_t *f) {
    memset(f, 0, 8 * sizeof(uint64_t));
    if (n <= 0) return 0;
    DEV(lsk_sync(stream));
    for (int k = 0; k < 8; ++k) DEV(lsk_d2h(&f[k], d_reps + (n - 1) * k / 7, sizeof(uint64_t)));
    return 0;
}
int ls_amd_internal_gtab_acquire(ls_amd_gtab **out, int L, uint64_t const *d_reps, int64_t n, uint8_t const *d_masks, int P,
                                 void *stream) {
    *out = NULL;
    uint64_t finger[8];
    if (gtab_fingerprint(d_reps, n, stream, finger) != 0) return -1;
    pthread_mutex_lock(&g_gtab_lock);
    for (ls_amd_gtab *t = g_gtabs; t; t = t->next)
        if (t->reps == d_reps && t->n == n && t->masks == d_masks && t->P == P && t->L == L && memcmp(t->finger, finger, sizeof(finger)) == 0) {
            ++t->refs;
            pthread_mutex_unlock(&g_gtab_lock);
            *out = t;
            return 0;
        }
    ls_amd_gtab *t = (ls_amd_gtab *)calloc(1, sizeof(*t));
    t->reps = d_reps; t->n = n; t->masks = d_masks; t->P = P; t->L = L;
    memcpy(t->finger, finger, sizeof(finger));
    if (gtab_build(t, stream) != 0) {
        gtab_free(t);
        pthread_mutex_unlock(&g_gtab_lock);
        return -1;
    }
    t->refs = 1;
    t->next = g_gtabs;
    g_gtabs = t;
    pthread_mutex_unlock(&g_gtab_lock);
    *out = t;
    return 0;
}
uint32_t const *ls_amd_internal_gtab_perm(ls_amd_gtab const *t) { return t->d_perm; }
int64_t ls_amd_internal_gtab_max_count(ls_amd_gtab const *t) { return t->max_count; }
int64_t const *ls_amd_internal_gtab_counts(ls_amd_gtab const *t) { return t->counts; }
int64_t ls_amd_internal_gtab_bytes(ls_amd_gtab const *t) { return (int64_t)16 << t->tab.bbits; }

static int timing_begin(ls_amd_plan *pl, void *stream) {
    if (pl->t_count >= pl->t_capacity) return -1;
    if (lsk_event_record(pl->t_start[pl->t_count], stream) != 0) return -1;
    return pl->t_count;
}
static void timing_end(ls_amd_plan *pl, int slot, void *stream) {
    if (slot < 0) return;
    if (lsk_event_record(pl->t_stop[slot], stream) == 0) pl->t_count = slot + 1;
}

/* ---- stage timers ----------------------------------------------------------------------------- */
enum { ST_DIAG = 0, ST_REFRESH = 1, ST_ROWS = 2, ST_GENERATE = 3, ST_EXCHANGE = 4, ST_SCATTER = 5, ST_RETURN = 6, ST_COUNT = 7 };
static int stage_begin(ls_amd_plan *pl, int stage, void *stream) {
    if (pl->st_count >= pl->st_capacity) return -1;
    if (lsk_event_record(pl->st_start[pl->st_count], stream) != 0) return -1;
    pl->st_stage[pl->st_count] = (unsigned char)stage;
    return pl->st_count;
}
static void stage_end(ls_amd_plan *pl, int slot, void *stream) {
    if (slot < 0) return;
    if (lsk_event_record(pl->st_stop[slot], stream) == 0) pl->st_count = slot + 1;
}
int ls_amd_internal_stage_begin(ls_amd_plan *pl, int stage, void *stream) { return stage_begin(pl, stage, stream); }
void ls_amd_internal_stage_end(ls_amd_plan *pl, int slot, void *stream) { stage_end(pl, slot, stream); }
void ls_amd_internal_count_matvec(ls_amd_plan *pl) { if (pl->st_capacity) ++pl->st_matvecs; }
/* Plans that accumulate into the caller's y with hardware f64 atomics (push: k_direct, the packet consumers) need y in
 * coarse-grained memory (hipMalloc): on fine-grained memory -- hipMallocManaged unless advised otherwise -- the unsafe-fp-atomics
 * `global_atomic_add_f64` may lose updates silently ("Atomics").  A managed y is refused here instead;
 * the host-pointer boundary (ls_chpl_matrix_vector_product) stages managed memory and never gets here with it.
 * LS_AMD_ALLOW_MANAGED_Y=1: the caller vouches for hipMemAdvise(..., hipMemAdviseSetCoarseGrain, ...). */
int ls_amd_internal_check_y(ls_amd_plan *pl, void const *y) {
    if (pl->family != FAMILY_DIRECT_PUSH && pl->family != FAMILY_TILE) return 0;
    if (!y || y == pl->y_checked[0] || y == pl->y_checked[1]) return 0;
    if (lsk_pointer_kind(y) == LSK_PTR_MANAGED) {
        char const *e = getenv("LS_AMD_ALLOW_MANAGED_Y");
        if (!e || atoi(e) == 0)
            return set_error("y is managed (hipMallocManaged) memory: this plan accumulates with hardware f64 atomics, which need "
                             "coarse-grained memory -- pass hipMalloc memory, go through ls_chpl_matrix_vector_product (which stages it), "
                             "or advise the range coarse-grained and set LS_AMD_ALLOW_MANAGED_Y=1");
    }
    pl->y_checked[1] = pl->y_checked[0];
    pl->y_checked[0] = y;
    return 0;
}
/* folds the recorded event pairs into the per-stage totals (synchronises on the events) */
static int stage_collect(ls_amd_plan *pl) {
    for (int i = 0; i < pl->st_count; ++i) {
        float ms = 0;
        DEV(lsk_event_elapsed_ms(pl->st_start[i], pl->st_stop[i], &ms));
        pl->st_ms[pl->st_stage[i]] += ms;
        pl->st_calls[pl->st_stage[i]] += 1;
    }
    pl->st_count = 0;
    return 0;
}
int ls_amd_plan_enable_stage_timing(ls_amd_plan *pl, int max_events) {
    for (int i = 0; i < pl->st_capacity; ++i) { lsk_event_destroy(pl->st_start[i]); lsk_event_destroy(pl->st_stop[i]); }
    free(pl->st_start); free(pl->st_stop); free(pl->st_stage);
    pl->st_start = pl->st_stop = NULL; pl->st_stage = NULL;
    pl->st_capacity = pl->st_count = 0;
    memset(pl->st_ms, 0, sizeof(pl->st_ms)); memset(pl->st_calls, 0, sizeof(pl->st_calls));
    pl->st_matvecs = 0;
    if (max_events <= 0) return 0;
    pl->st_start = (void **)calloc(max_events, sizeof(void *));
    pl->st_stop = (void **)calloc(max_events, sizeof(void *));
    pl->st_stage = (unsigned char *)calloc(max_events, 1);
    for (int i = 0; i < max_events; ++i) {
        DEV(lsk_event_create(&pl->st_start[i]));
        DEV(lsk_event_create(&pl->st_stop[i]));
        pl->st_capacity = i + 1;
    }
    return 0;
}
int ls_amd_plan_stage_times(ls_amd_plan *pl, double *ms, int64_t *calls, int64_t *matvecs) {
    if (stage_collect(pl) != 0) return -1;
    for (int k = 0; k < ST_COUNT; ++k) { ms[k] = pl->st_ms[k]; calls[k] = pl->st_calls[k]; }
    if (matvecs) *matvecs = pl->st_matvecs;
    return 0;
}
/* the reference prints this tree when kDisplayTimings is set (DMV:1028-1052); stages are mapped to what replaced them */
int ls_amd_plan_timing_report(ls_amd_plan *pl, char *buf, size_t cap) {
    if (stage_collect(pl) != 0) return -1;
    double total = 0;
    for (int k = 0; k < ST_COUNT; ++k) total += pl->st_ms[k];
    int64_t const m = pl->st_matvecs > 0 ? pl->st_matvecs : 1;
#define PER(k) (pl->st_ms[k] / (double)m), (long long)pl->st_calls[k]
    snprintf(buf, cap,
             "matrixVectorProduct [%s]: %.3f ms per matvec over %lld matvecs (device time of the stages, HIP events)\n"
             " \xe2\x94\x9c\xe2\x94\x80 localDiagonal (k_diag):                         %9.3f ms  (%lld launches)\n"
             " \xe2\x94\x9c\xe2\x94\x80 x preparation (table refresh | x n(rep) | permutation): %1.3f ms  (%lld launches)\n"
             " \xe2\x94\x9c\xe2\x94\x80 row kernel (fused computeOffDiag + localProcess): %7.3f ms  (%lld launches)\n"
             " \xe2\x94\x9c\xe2\x94\x80 producers (k_tile: computeOffDiag, stateInfo,\n"
             " \xe2\x94\x82   localeIdxOf, radixOneStep, own localProcess):    %9.3f ms  (%lld launches)\n"
             " \xe2\x94\x9c\xe2\x94\x80 exchange wait on the compute stream (all-to-all-v): %5.3f ms  (%lld waits)\n"
             " \xe2\x94\x9c\xe2\x94\x80 consumers (k_scatter: indexing + accessing):    %9.3f ms  (%lld launches)\n"
             " \xe2\x94\x94\xe2\x94\x80 replicated-x: y rows grouped by owner + returned: %8.3f ms  (%lld passes)\n",
             ls_amd_plan_kernel_name(pl), total / (double)m, (long long)pl->st_matvecs, PER(ST_DIAG), PER(ST_REFRESH), PER(ST_ROWS),
             PER(ST_GENERATE), PER(ST_EXCHANGE), PER(ST_SCATTER), PER(ST_RETURN));
#undef PER
    return 0;
}

int ls_amd_plan_enable_timing(ls_amd_plan *pl, int max_samples) {
    for (int i = 0; i < pl->t_capacity; ++i) { lsk_event_destroy(pl->t_start[i]); lsk_event_destroy(pl->t_stop[i]); }
    free(pl->t_start); free(pl->t_stop);
    pl->t_start = pl->t_stop = NULL;
    pl->t_capacity = pl->t_count = 0;
    if (max_samples <= 0) return 0;
    pl->t_start = (void **)calloc(max_samples, sizeof(void *));
    pl->t_stop = (void **)calloc(max_samples, sizeof(void *));
    for (int i = 0; i < max_samples; ++i) {
        DEV(lsk_event_create(&pl->t_start[i]));
        DEV(lsk_event_create(&pl->t_stop[i]));
        pl->t_capacity = i + 1;
    }
    return 0;
}
int ls_amd_plan_kernel_times(ls_amd_plan *pl, float *ms, int capacity, int *count) {
    int n = pl->t_count < capacity ? pl->t_count : capacity;
    for (int i = 0; i < n; ++i) DEV(lsk_event_elapsed_ms(pl->t_start[i], pl->t_stop[i], &ms[i]));
    *count = n;
    pl->t_count = 0;
    return 0;
}
int ls_amd_fill_random(int64_t n, uint64_t const *d_states, uint64_t seed, ls_amd_dtype dtype, void *d_out,
                       void *stream) {
    DEV(lsk_fill_random(n, d_states, seed, dtype == LS_AMD_C128, d_out, stream));
    return 0;
}

static int prescaled_x(ls_amd_plan *pl, int64_t n, uint64_t const *d_reps, void const *d_x, double const *d_norms,
                       void const **out, void *stream);

/* near window of the staged pull kernels: entries either side of a 256-row tile that are resolved in LDS (LS_AMD_PULL_HALO;
 * 0 = every partner through the table).  The indexed kernels hold the window as a hash set of <= 1024 entries and measure
 * alike from 64 to 512 (profiles/r3_indexed_ab_*): 256; the value-table kernel keeps its sorted window of +-512. */
static int pull_halo_setting(int indexed) {
    int const mx = indexed ? lsk_pull_max_halo() : 512;
    char const *e = getenv("LS_AMD_PULL_HALO");
    int h = e ? atoi(e) : (indexed ? 256 : 512);
    if (h < 0) h = 0;
    if (h > mx) h = mx;
    return h;
}

/* ---- split matvec: packet streams ------------------------------------------------------------------------------------- */
static void split_free(ls_amd_plan *pl) {
    if (pl->pbuf.offs) lsk_free((void *)pl->pbuf.offs);
    if (pl->pbuf.slots) lsk_free(pl->pbuf.slots);
    if (pl->pbuf.rows) lsk_free(pl->pbuf.rows);
    if (pl->pbuf.coefs) lsk_free(pl->pbuf.coefs);
    if (pl->pbuf.counts) lsk_free(pl->pbuf.counts);
    memset(&pl->pbuf, 0, sizeof(pl->pbuf));
    pl->split_rows = 0;
    pl->slot_cache = pl->slot_cache_valid = 0;
}
/* room for the packet streams of as many of the plan's rows as `max_bytes` allow (all of them if it can); returns the number
 * of rows covered (0: none -- the plan keeps the fused kernel) */
int64_t ls_amd_internal_plan_split_enable(ls_amd_plan *pl, int64_t max_bytes) {
    split_free(pl);
    if (!pl->idx_mode || pl->dbs.proj != LSK_PROJ_FULL || pl->parts[0].count <= 0) return 0;
    int64_t const cap = lsk_pullbuf_cap(pl->dop);
    int const nc = lsk_pullbuf_coef_doubles(pl->dop, pl->dbs);
    int64_t const per_stream = cap * (4 + 1 + 8 * nc) + 4; /* 64 rows */
    int64_t streams = (pl->parts[0].count + 63) / 64;
    streams = (streams + 3) & ~(int64_t)3; /* whole 256-row tiles */
    if (max_bytes > 0 && streams * per_stream > max_bytes) streams = (max_bytes / per_stream) & ~(int64_t)3;
    while (streams >= 4) {
        void *a = NULL, *b = NULL, *c = NULL, *d = NULL;
        if (lsk_malloc(&a, (size_t)(streams * cap * 4)) == 0 && lsk_malloc(&b, (size_t)(streams * cap)) == 0 &&
            (nc == 0 || lsk_malloc(&c, (size_t)(streams * cap * 8 * nc)) == 0) && lsk_malloc(&d, (size_t)(streams * 4)) == 0) {
            pl->pbuf.slots = (uint32_t *)a; pl->pbuf.rows = (uint8_t *)b; pl->pbuf.coefs = (double *)c; pl->pbuf.counts = (uint32_t *)d;
            pl->pbuf.cap = cap;
            pl->pbuf.row0 = 0;
            pl->split_rows = streams * 64 < pl->parts[0].count ? streams * 64 : pl->parts[0].count;
            return pl->split_rows;
        }
        if (a) lsk_free(a);
        if (b) lsk_free(b);
        if (c) lsk_free(c);
        if (d) lsk_free(d);
        streams = (streams / 2) & ~(int64_t)3; /* no room: half as many rows */
    }
    return 0;
}
int64_t ls_amd_internal_plan_split_rows(ls_amd_plan const *pl) { return pl->split_rows; }
/* rows resolved ahead by the NEXT ls_amd_internal_repl_split_begin: a prefix of the rows the packet buffer covers (rounded down
 * to whole 256-row tiles; <= 0 or >= split_rows: all of them).  A cached plan keeps all its rows. */
static int64_t split_rows_now(ls_amd_plan const *pl) {
    if (pl->slot_cache || pl->split_active <= 0 || pl->split_active >= pl->split_rows) return pl->split_rows;
    return pl->split_active;
}
void ls_amd_internal_plan_split_set_active(ls_amd_plan *pl, int64_t rows) {
    pl->split_active = rows <= 0 ? 0 : (rows & ~(int64_t)255);
    if (rows > 0 && pl->split_active < 256) pl->split_active = 256;
}
int64_t ls_amd_internal_plan_split_active(ls_amd_plan const *pl) { return split_rows_now(pl); }
int ls_amd_internal_plan_slot_cached(ls_amd_plan const *pl) { return pl->slot_cache != 0; }
/* Streams laid out back to back at their exact lengths (a count pass of stage A + a scan): about half of what the worst-case
 * stride of the split form reserves.  Covers every row if `max_bytes` (<= 0: no ceiling) and the device allow, else the longest
 * prefix of whole 256-row tiles that fits -- the rows behind it keep the fused kernel.  Returns the rows covered (0: none, nothing
 * is left allocated), -1 on a device error. */
static int64_t split_enable_exact(ls_amd_plan *pl, int64_t max_bytes) {
    split_free(pl);
    part_state *ps = &pl->parts[0];
    int64_t const streams_all = (ps->count + 63) / 64;
    int const nc = lsk_pullbuf_coef_doubles(pl->dop, pl->dbs);
    int64_t const per_packet = 4 + 1 + 8 * nc;
    void *offs = NULL;
    /* (offsets and counts for WHOLE tiles of four streams: every wave of the last tile reads its offset and stores its count) */
    if (lsk_malloc(&offs, 8 * (size_t)(((streams_all + 3) & ~(int64_t)3) + 1)) != 0) return 0;
    if (lsk_tile_pull_stream_offsets(pl->dop, pl->dbs, 0, ps->count, ps->d_reps, (int64_t *)offs, NULL) != 0) { lsk_free(offs); return -1; }
    int64_t streams = streams_all;
    for (;;) {
        /* largest number of streams (whole tiles, or all of them) whose packets fit the ceiling: offs is ascending */
        int64_t total = 0;
        if (lsk_d2h(&total, (char *)offs + 8 * (size_t)streams, 8) != 0) { lsk_free(offs); return -1; }
        if (max_bytes > 0 && total * per_packet + 12 * streams + 8 > max_bytes) {
            int64_t lo = 0, hi = streams / 4; /* in tiles: lo fits, hi does not */
            while (hi - lo > 1) {
                int64_t const mid = lo + (hi - lo) / 2;
                int64_t t = 0;
                if (lsk_d2h(&t, (char *)offs + 8 * (size_t)(4 * mid), 8) != 0) { lsk_free(offs); return -1; }
                if (t * per_packet + 12 * 4 * mid + 8 <= max_bytes) lo = mid; else hi = mid;
            }
            streams = 4 * lo;
            if (streams <= 0) { lsk_free(offs); return 0; }
            if (lsk_d2h(&total, (char *)offs + 8 * (size_t)streams, 8) != 0) { lsk_free(offs); return -1; }
        }
        if (total >= ((int64_t)1 << 40)) { lsk_free(offs); return 0; }
        void *a = NULL, *b = NULL, *c = NULL, *d = NULL;
        if (lsk_malloc(&a, (size_t)(4 * total + 4)) == 0 && lsk_malloc(&b, (size_t)(total + 4)) == 0 &&
            (nc == 0 || lsk_malloc(&c, (size_t)(8 * nc * total + 8)) == 0) && lsk_malloc(&d, (size_t)(4 * ((streams + 3) & ~(int64_t)3))) == 0) {
            pl->pbuf.slots = (uint32_t *)a; pl->pbuf.rows = (uint8_t *)b; pl->pbuf.coefs = (double *)c; pl->pbuf.counts = (uint32_t *)d;
            pl->pbuf.offs = (int64_t const *)offs;
            pl->pbuf.cap = lsk_pullbuf_cap(pl->dop);
            pl->pbuf.row0 = 0;
            pl->split_rows = streams * 64 < ps->count ? streams * 64 : ps->count;
            pl->slot_cache_bytes = total * per_packet + 12 * streams + 8;
            return pl->split_rows;
        }
        if (a) lsk_free(a);
        if (b) lsk_free(b);
        if (c) lsk_free(c);
        if (d) lsk_free(d);
        /* the device has no room for this many: three quarters of it, in whole tiles */
        max_bytes = (total * per_packet + 12 * streams + 8) / 4 * 3;
        if (max_bytes < ((int64_t)1 << 20)) { lsk_free(offs); return 0; }
    }
}
/* Keep the resolved packet streams across matvecs (see slot_cache above; include/ls_amd.h).  Returns the number of rows whose
 * streams are cached: all of them, or the prefix that fits `max_bytes` / the device -- the rest runs the fused kernel. */
int64_t ls_amd_plan_cache_slots(ls_amd_plan *pl, int64_t max_bytes) {
    if (!pl || !pl->idx_mode || pl->dbs.proj != LSK_PROJ_FULL || pl->parts[0].count <= 0 ||
        (pl->family != FAMILY_TILE_PULL && pl->family != FAMILY_REPL_TILE)) return 0;
    int64_t const rows = split_enable_exact(pl, max_bytes);
    if (rows < 0) return dev_error();
    if (rows == 0) return 0;
    pl->slot_cache = 1;
    pl->slot_cache_valid = 0;
    return rows;
}
int ls_amd_plan_slot_cache_rows(ls_amd_plan const *pl, int64_t *rows, int64_t *bytes) {
    if (rows) *rows = pl->slot_cache ? pl->split_rows : 0;
    if (bytes) *bytes = pl->slot_cache ? pl->slot_cache_bytes : 0;
    return 0;
}

/* LS_AMD_PACKETS=block keeps the block-wide packet lists of k_tile (cursor atomics); default: per-wave rings (k_tile_wv) */
static int packets_wave_rings(void) {
    char const *e = getenv("LS_AMD_PACKETS");
    return !(e && strcmp(e, "block") == 0);
}
static int64_t rows_per_round_default(void) {
    char const *e = getenv("LS_AMD_ROWS_PER_ROUND");
    if (e) { long long v = atoll(e); if (v > 0) return (int64_t)v; }
    return (int64_t)1 << 24;
}

static int build_search_index(part_state *ps, int number_sites, void *stream) {
    if (ps->count >= 0xffffffffLL) return set_error("partitions with >= 2^32 states are not supported");
    int bits = 0;
    while ((1LL << (bits + 1)) <= ps->count) ++bits; /* floor(log2(count)) */
    bits -= 3;
    if (bits < 1) bits = 1;
    if (bits > 26) bits = 26;
    if (bits > number_sites) bits = number_sites;
    int shift = number_sites - bits;
    int64_t nbuckets = (int64_t)1 << bits;
    void *p;
    DEV(lsk_malloc(&p, 4 * (size_t)(nbuckets + 2)));
    ps->d_table = (uint32_t *)p;
    DEV(lsk_build_table(ps->count, ps->d_reps, shift, nbuckets, ps->d_table, stream));
    ps->index.kind = LSK_INDEX_SEARCH;
    ps->index.shift = shift;
    ps->index.table = ps->d_table;
    return 0;
}

static int upload(void **slot, void const *host, size_t bytes) {
    DEV(lsk_malloc(slot, bytes ? bytes : 8));
    DEV(lsk_h2d(*slot, host, bytes));
    return 0;
}

/* Tile map of the row kernels (lsk_tilemap in lsk.h): the traversal order is data, not code.
 *   chunk == 0: XCD k gets the k-th contiguous eighth of the row tiles;
 *   chunk  > 0: chunks of `chunk` consecutive tiles are dealt to the XCDs round-robin, so all XCDs advance through the
 *               same region of the vector together (the shared Infinity Cache sees one moving front) while each keeps
 *               runs of consecutive tiles for its own L2.
 * Orders that were measured and removed (r2, chain_32 f64, gpurun_out/r2/order_sweep.log): sets closed under the flips of
 * the top t site bits, per XCD or chip-wide (t = 6..12, 0.26 - 8 M rows per set): 10.97 - 11.67 ms against 11.09 ms
 * contiguous and 10.60 ms chunked -- the far-pair gathers are bound by the L2-miss request rate, which no order of
 * 1024-row tiles changes (a set small enough for one 4 MiB L2 is 70 tiles; 224 tiles are in flight per XCD). */
typedef struct { uint64_t *e; int64_t n, cap; } tile_list;
static void tile_push(tile_list *l, int64_t row, int64_t cnt) {
    if (l->n == l->cap) { l->cap = l->cap ? 2 * l->cap : 1024; l->e = (uint64_t *)realloc(l->e, sizeof(uint64_t) * (size_t)l->cap); }
    l->e[l->n++] = (uint64_t)row | ((uint64_t)cnt << 48);
}
static int tilemap_host(int64_t n, int TILE, int64_t chunk, uint64_t **out, int64_t *slots_out) {
    tile_list lists[8];
    memset(lists, 0, sizeof(lists));
    int64_t const tiles = (n + TILE - 1) / TILE;
    if (chunk > 0) {
        for (int64_t q = 0; q < tiles; ++q)
            tile_push(&lists[(q / chunk) % 8], q * TILE, n - q * TILE < TILE ? n - q * TILE : TILE);
    } else
        for (int k = 0; k < 8; ++k) /* XCD k: the k-th contiguous eighth of the tiles */
            for (int64_t q = tiles * k / 8; q < tiles * (k + 1) / 8; ++q)
                tile_push(&lists[k], q * TILE, n - q * TILE < TILE ? n - q * TILE : TILE);
    int64_t slots = 0, total = 0;
    for (int k = 0; k < 8; ++k) if (lists[k].n > slots) slots = lists[k].n;
    uint64_t *flat = (uint64_t *)calloc((size_t)(8 * slots > 0 ? 8 * slots : 1), sizeof(uint64_t));
    for (int k = 0; k < 8; ++k) {
        for (int64_t q = 0; q < lists[k].n; ++q) { flat[k * slots + q] = lists[k].e[q]; total += (int64_t)(lists[k].e[q] >> 48); }
        free(lists[k].e);
    }
    if (total != n) {
        free(flat);
        return set_error("internal error: tile map covers %lld of %lld rows", (long long)total, (long long)n);
    }
    *out = flat;
    *slots_out = slots;
    return 0;
}
static int build_tilemap(ls_amd_plan *pl, int64_t n, int TILE) {
    uint64_t *flat = NULL;
    int64_t slots = 0;
    /* Default for the staged kernel: chunks of 256 tiles -- measured on chain_32 with one block per tile
     * (gpurun_out/r2: 9.35 ms contiguous eighths, 8.98 / 8.56 / 8.47 / 8.47 / 8.49 / 8.53 / 8.85 / 9.83 ms at
     * 4 / 32 / 128 / 256 / 512 / 1024 / 4096 / 16384) */
    char const *e = getenv("LS_AMD_TILE_CHUNK");
    int64_t const chunk = e ? atoll(e) : (TILE >= 512 ? 256 : 0);
    if (tilemap_host(n, TILE, chunk > 0 ? chunk : 0, &flat, &slots) < 0) return -1;
    int const up = upload(&pl->d_tilemap, flat, sizeof(uint64_t) * (size_t)(8 * slots > 0 ? 8 * slots : 1));
    free(flat);
    if (up) return -1;
    pl->tilemap.entries = (uint64_t const *)pl->d_tilemap;
    pl->tilemap.slots_per_xcd = slots;
    return 0;
}
/* test hook (ls_amd.h): the row kernel of a one-process plan skips ONE row -- the first tile of XCD 0's list loses its last row, so
 * that row's y is never written (pull) or its contributions never leave (push).  Memory-safe: a shorter tile reads nothing new.
 * Returns 1 when plan data was corrupted, 0 when the plan has neither a tile map nor per-row norms (packet plans). */
int ls_amd_test_corrupt_plan(ls_amd_plan *pl) {
    if (!pl) return 0;
    if (pl->family == FAMILY_TILE_PULL && pl->n_local == 1 && pl->parts[0].d_norms && pl->parts[0].count > 0) {
        /* projected pull plans walk their tiles by arithmetic, not through the map: ONE per-row norm is doubled instead -- that
         * row's y and the contributions it makes to its partners come out wrong (plan data the kernel reads; memory-safe) */
        double v = 0;
        double *p = pl->parts[0].d_norms + pl->parts[0].count / 2;
        if (lsk_device_sync() != 0 || lsk_d2h(&v, p, sizeof(v)) != 0) return 0;
        v *= 2.0;
        return lsk_h2d(p, &v, sizeof(v)) == 0;
    }
    if (!pl->d_tilemap || pl->tilemap.slots_per_xcd <= 0) return 0;
    uint64_t e = 0;
    if (lsk_device_sync() != 0 || lsk_d2h(&e, pl->d_tilemap, sizeof(e)) != 0) return 0;
    uint64_t const rows = e >> 48;
    if (rows < 2) return 0;
    e = (e & 0xffffffffffffULL) | ((rows - 1) << 48);
    if (lsk_h2d(pl->d_tilemap, &e, sizeof(e)) != 0) return 0;
    return 1;
}
/* test hook (host only): the tile map of n rows.  Returns slots per XCD (< 0 on error); *entries is malloc'ed, free
 * with ls_amd_test_free. */
int ls_amd_test_window_find(uint64_t const *reps, int n, uint64_t key) { return lsk_test_window_find(reps, n, key); }
int ls_amd_test_nw_find(uint64_t const *reps, int n, uint64_t key) { return lsk_test_nw_find(reps, n, key); }
int ls_amd_test_chain_near_table(int elem, int ldsp, int16_t *out) { return lsk_test_chain_near_table(elem, ldsp, out); }
uint64_t ls_amd_test_rep_trivial_dihedral(uint64_t a, int L, int inv, int reflect) { return lsk_test_rep_trivial_dihedral(a, L, inv, reflect); }
int ls_amd_bench_k4(int L, int inv, int reflect, int variant, int64_t n, uint64_t const *d_reps, uint64_t *d_out, void *stream) {
    if (lsk_bench_k4(L, inv, reflect, variant, n, d_reps, d_out, stream) != 0) { set_error("%s", lsk_last_error()); return -1; }
    return 0;
}
int64_t ls_amd_test_tilemap(int64_t n, int tile_rows, int64_t chunk, uint64_t **entries) {
    int64_t slots = 0;
    *entries = NULL;
    if (tilemap_host(n, tile_rows, chunk, entries, &slots) < 0) return -1;
    return slots;
}
void ls_amd_test_free(void *p) { free(p); }
/* host-only test hooks of the static index table (lsk_gtab): shape for n keys of L bits, sequential build into a malloc'ed
 * array of 2 << bbits entries (*entries, ls_amd_test_free), lookup */
int ls_amd_test_gtab_bits(int L, int64_t n) { return lsk_gtab_bits(L, n, (int64_t)1 << 40); }
int ls_amd_test_gtab_build(int L, int bbits, int64_t n, uint64_t const *reps, uint32_t const *payload, uint64_t **entries) {
    lsk_gtab t;
    t.entries = NULL; t.L = L; t.bbits = bbits; t.tbits = L - bbits;
    *entries = (uint64_t *)malloc(sizeof(uint64_t) * ((size_t)2 << bbits));
    if (lsk_test_gtab_build_host(t, *entries, n, reps, payload) != 0) { free(*entries); *entries = NULL; return -1; }
    return 0;
}
int64_t ls_amd_test_gtab_find(int L, int bbits, uint64_t const *entries, uint64_t key) {
    lsk_gtab t;
    t.entries = NULL; t.L = L; t.bbits = bbits; t.tbits = L - bbits;
    return lsk_test_gtab_find(t, entries, key);
}

/* Staged row kernel (k_chain_t, lsk.h): pull, f64 or c128 vectors, <= 64 sites, the full fixed-weight basis without
 * symmetries, a real operator made of exchange runs plus at most two other exchange pairs.  LS_AMD_ROW_KERNEL=generic keeps k_direct. */
/* Staged push (k_push_t, lsk.h): the full fixed-weight basis without symmetries, a real operator with at least one undirected
 * exchange run (what gives the near targets an LDS window is worth having for).  LS_AMD_ROW_KERNEL=generic keeps k_direct. */
static int push_staged_eligible(ls_amd_plan const *pl) {
    ls_hs_operator const *op = pl->op;
    struct ls_amd_operator_ext const *ext = OEXT(op);
    char const *e = getenv("LS_AMD_ROW_KERNEL");
    if (e && strcmp(e, "generic") == 0) return 0;
    if (op->basis->number_sites > 64 || op->basis->spin_inversion != 0 || pl->dbs.proj != LSK_PROJ_NONE || !ext->is_real ||
        ext->runs.n_runs <= 0 || BEXT(op->basis)->hamming_weight < 0)
        return 0;
    for (int q = 0; q < ext->runs.n_runs; ++q) if (ext->runs.cnt[q] >> 16) return 0; /* directed runs: k_direct's DIRECTED instantiation */
    return 1;
}
static int chain_eligible(ls_amd_plan const *pl) {
    ls_hs_operator const *op = pl->op;
    struct ls_amd_operator_ext const *ext = OEXT(op);
    char const *e = getenv("LS_AMD_ROW_KERNEL"); /* auto (default) | generic: k_direct | pairs: k_pairs_t where it applies | pairrows / pairsites: k_pairs_row / k_pairs_site where they apply */
    if (e && (strcmp(e, "generic") == 0 || strcmp(e, "pairs") == 0 || strcmp(e, "pairrows") == 0 || strcmp(e, "pairsites") == 0)) return 0;
    int const L = op->basis->number_sites;
    int const inv = op->basis->spin_inversion != 0;
    /* Inversion sectors WITHOUT permutations (round 6; BASELINE config 1's sector, BatchedOperator.chpl:119-161): at half filling the
     * canonical state of {sigma, ~sigma} is the one with the top site clear, so the basis is the full set of weight-L/2 words of
     * L - 1 sites in the same colex order (index kind COMBINADIC with Leff = L - 1), every pair below the top site acts as in the
     * plain sector, and a pair that touches the top site ALWAYS lands on a flipped state: partner = rank(~beta), amplitude
     * s v -- a cached pair.  The ring then is a run of L - 2 bonds + two cached pairs, exactly what k_chain_t already takes. */
    if (inv && (pl->dbs.proj != LSK_PROJ_INVERSION || pl->family != FAMILY_DIRECT_PULL || (L & 1) || BEXT(op->basis)->hamming_weight != L / 2)) return 0;
    if (L > 64 || (!inv && pl->dbs.proj != LSK_PROJ_NONE) ||
        !ext->is_real || !ext->is_hermitian || ext->runs.n_runs <= 0 || ext->n_diag <= 0)
        return 0;
    int extra = 0; /* the run bond (L - 2, L - 1) leaves its run and joins the cached pairs */
    if (inv) for (int q = 0; q < ext->runs.n_runs; ++q) if (ext->runs.lo0[q] + ext->runs.cnt[q] - 1 == L - 2) extra = 1;
    if (ext->n_groups - ext->runs.n_run_groups + extra > 2) return 0;
    for (int g = ext->runs.n_run_groups; g < ext->n_groups; ++g) {
        lsk_group const *G = &ext->groups[g];
        if (G->fast != LSK_GROUP_EXCHANGE || G->v_im != 0.0 || __builtin_popcountll(G->x) != 2) return 0;
    }
    return 1;
}
/* tile map with 1024-row tiles + the cache of partner ranks of the exchange pairs outside the runs (for a ring: the
 * bond that closes it) -- 4 bytes per row instead of a ranking loop of `weight` steps per row and matvec.
 * Leaves pl->has_chain == 0 (and no tile map) when a partner leaves the basis: k_direct reports that at run time,
 * as the reference does. */
static int setup_chain(ls_amd_plan *pl, lsk_index index, int64_t n, uint64_t const *d_reps, void *stream) {
    struct ls_amd_operator_ext const *ext = OEXT(pl->op);
    int nc = ext->n_groups - ext->runs.n_run_groups;
    /* the cached pairs: flip mask + amplitude.  Inversion sectors (chain_eligible): the run bond that touches the top site becomes
     * a cached pair, pairs that touch the top site carry the sector's sign, and THIS plan's copy of the run table loses that bond */
    int const Ls = pl->op->basis->number_sites;
    int const inv = pl->dbs.proj == LSK_PROJ_INVERSION;
    uint64_t const top = 1ULL << (Ls - 1);
    uint64_t cx[3];
    double cv[3];
    int k = 0;
    if (inv)
        for (int q = 0; q < pl->dop.runs.n_runs; ++q)
            if (pl->dop.runs.lo0[q] + pl->dop.runs.cnt[q] - 1 == Ls - 2) {
                cx[k] = 3ULL << (Ls - 2);
                cv[k++] = pl->dop.runs.v_re[q] * (double)pl->op->basis->spin_inversion;
                if (--pl->dop.runs.cnt[q] == 0) { /* (a run of that one bond: gone) */
                    for (int r = q; r + 1 < pl->dop.runs.n_runs; ++r) {
                        pl->dop.runs.lo0[r] = pl->dop.runs.lo0[r + 1]; pl->dop.runs.cnt[r] = pl->dop.runs.cnt[r + 1];
                        pl->dop.runs.v_re[r] = pl->dop.runs.v_re[r + 1]; pl->dop.runs.v_im[r] = pl->dop.runs.v_im[r + 1];
                    }
                    --pl->dop.runs.n_runs;
                }
                break;
            }
    for (int g = ext->runs.n_run_groups; g < ext->n_groups && k < 3; ++g) {
        cx[k] = ext->groups[g].x;
        cv[k++] = ext->groups[g].v_re * ((inv && (ext->groups[g].x & top)) ? (double)pl->op->basis->spin_inversion : 1.0);
    }
    nc = k;
    if (nc > 2) return 0;
    /* ranks are 32-bit while the whole basis (index.count states: x is indexed by global rank) has < 2^32 - 1 states;
     * LS_AMD_CHAIN_WIDE=1 forces the 64-bit instantiation (test hook: no in-tree config is that large) */
    char const *ew = getenv("LS_AMD_CHAIN_WIDE");
    pl->chain_wide = index.count >= 0xffffffffLL || (ew && atoi(ew) != 0 && pl->op->basis->number_sites > 32);
    size_t const es = pl->chain_wide ? sizeof(uint64_t) : sizeof(uint32_t);
    if (nc > 0 && n > 0) {
        void *q;
        if (lsk_malloc(&q, es * (size_t)nc * (size_t)n) != 0) return 0; /* no room for the cache: k_direct */
        pl->d_chain_cache = q;
        int zero = 0, flag = 0;
        DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
        for (int c = 0; c < nc; ++c)
            DEV(lsk_chain_cache(pl->dbs, index, n, d_reps, cx[c],
                                (char *)pl->d_chain_cache + es * (size_t)c * (size_t)n, pl->chain_wide, pl->d_err, stream));
        DEV(lsk_sync(stream));
        DEV(lsk_d2h(&flag, pl->d_err, sizeof(int)));
        DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
        if (flag) {
            lsk_free(pl->d_chain_cache);
            pl->d_chain_cache = NULL;
            return 0;
        }
        pl->chain_cached = nc;
        pl->chain_v[0] = cv[0];
        pl->chain_v[1] = nc > 1 ? cv[1] : 0.0;
    }
    if (build_tilemap(pl, n, lsk_chain_tile_rows(pl->cplx)) != 0) return -1;
    /* 32-bit states and ranks, f64 vectors: state and partner rank of the first cached pair are fused into one 8-byte
     * record per row (8 instead of 8 + 4 bytes per row from HBM, one load instead of two: measured on chain_32
     * 8.59 -> 8.41 ms; c128: 15.40 -> 15.70 ms, so complex vectors keep the two arrays).  This shape has no unfused
     * instantiation (it sat at 98 SGPRs, one resident block per CU short); a second cached pair stays in the cache
     * array, whose layout [pair][row] is what the kernel indexes.  No room for the records: the generic row kernel. */
    if (!pl->cplx && pl->op->basis->number_sites <= 32 && !pl->chain_wide && n > 0) {
        void *q;
        if (lsk_malloc(&q, sizeof(uint64_t) * (size_t)n) != 0) {
            if (pl->d_chain_cache) { lsk_free(pl->d_chain_cache); pl->d_chain_cache = NULL; }
            if (pl->d_tilemap) { lsk_free(pl->d_tilemap); pl->d_tilemap = NULL; memset(&pl->tilemap, 0, sizeof(pl->tilemap)); }
            pl->chain_cached = 0;
            return 0;
        }
        pl->d_chain_rec = (uint64_t *)q;
        DEV(lsk_chain_pack(n, d_reps, pl->chain_cached ? pl->d_chain_cache : NULL, (uint64_t *)q, stream));
        DEV(lsk_sync(stream));
        if (pl->chain_cached <= 1 && pl->d_chain_cache) { lsk_free(pl->d_chain_cache); pl->d_chain_cache = NULL; }
    }
    pl->has_chain = 1;
    return 0;
}

/* Staged row kernel for arbitrary exchange pairs (k_pairs_t, lsk.h): pull, the full fixed-weight basis of <= 32 sites without
 * symmetries, a real Hermitian operator whose off-diagonal groups are all exchange pairs (any two sites, any real amplitude)
 * and whose diagonal terms are zz terms on those same pairs -- Heisenberg / XXZ on any lattice.  Rings and open chains keep
 * k_chain_t.  LS_AMD_ROW_KERNEL=generic keeps the generic row kernel (k_direct), =pairs sends rings here too. */
static int pair_cmp(void const *pa, void const *pb) {
    lsk_pair const *a = (lsk_pair const *)pa, *b = (lsk_pair const *)pb;
    int const ca = a->j < 11 ? 0 : (a->i < 11 ? 1 : 2), cb = b->j < 11 ? 0 : (b->i < 11 ? 1 : 2);
    if (ca != cb) return ca - cb;
    if (a->i != b->i) return (int)a->i - (int)b->i;
    return (int)a->j - (int)b->j;
}
static int setup_pairs(ls_amd_plan *pl, int64_t n, uint64_t const *d_reps, void *stream) {
    ls_hs_operator const *op = pl->op;
    struct ls_amd_operator_ext const *ext = OEXT(op);
    int const L = op->basis->number_sites, hw = BEXT(op->basis)->hamming_weight;
    char const *e = getenv("LS_AMD_ROW_KERNEL");
    if (e && strcmp(e, "generic") == 0) return 0;
    int const wide = L > 32; /* 33..64 sites (round 6): the kernel streams the 8-byte representatives themselves */
    if (L > 64 || hw < 0 || hw + 2 > LSK_PAIR_KC || op->basis->spin_inversion != 0 || pl->dbs.proj != LSK_PROJ_NONE || !ext->is_real ||
        !ext->is_hermitian || ext->n_groups < 1 || ext->n_groups > LSK_MAX_PAIRS || n <= 0 || n >= 0xffffffffLL ||
        ext->n_diag <= 0) /* (no diagonal terms: y is accumulated into, DMV:1062-1063 -- left to the generic kernel) */
        return 0;
    /* The kernel walks the basis in blocks "all 11-bit low words of weight kl under one high part": C(11, kl) rows, kl ~ 11 hw / L on
     * average.  Far from half filling the blocks are a handful of rows, a wave spans many of them and prices its HIGH / STRADDLE pairs
     * once per block: measured on a 36-site square lattice (profiles/r6_widen_bench.txt) weight 6 -- kl ~ 1.8 -- 1.09 ms against 0.25 ms
     * of the generic row kernel, weight 9 -- kl ~ 2.75 -- 15.3 against 14.3 ms.  Such bases keep k_direct (LS_AMD_ROW_KERNEL=pairs forces
     * the staged kernel: tests). */
    /* Round 6: those bases take the one-row-per-lane variant of the same plan instead (k_pairs_row: O(1) rank shifts out of per-row
     * prefix arrays in LDS), which leaves the generic kernel behind at any filling; LS_AMD_ROW_KERNEL=pairrows forces it everywhere. */
    int const force_sites = e && strcmp(e, "pairsites") == 0;
    int const by_row = force_sites || (e && strcmp(e, "pairrows") == 0) || (!(e && strcmp(e, "pairs") == 0) && (11 * hw < 4 * L || 11 * (L - hw) < 4 * L));
    lsk_pair recs[LSK_MAX_PAIRS];
    memset(recs, 0, sizeof(recs));
    for (int g = 0; g < ext->n_groups; ++g) {
        lsk_group const *G = &ext->groups[g];
        if (G->fast != LSK_GROUP_EXCHANGE || G->v_im != 0.0 || __builtin_popcountll(G->x) != 2) return 0;
        recs[g].i = (uint8_t)__builtin_ctzll(G->x);
        recs[g].j = (uint8_t)(63 - __builtin_clzll(G->x));
        recs[g].v = G->v_re;
    }
    double dsum = 0.0;
    for (int t = 0; t < ext->n_diag; ++t) { /* every diagonal term must be a zz term on one of the pairs */
        lsk_term const *T = &ext->diag[t];
        if (T->m != 0 || T->r != 0 || T->v_im != 0.0) return 0;
        int g = 0;
        while (g < ext->n_groups && ext->groups[g].x != T->s) ++g;
        if (g == ext->n_groups) return 0;
        recs[g].vz += T->v_re;
        dsum += T->v_re;
    }
    qsort(recs, (size_t)ext->n_groups, sizeof(lsk_pair), pair_cmp);
    lsk_pairplan pp;
    memset(&pp, 0, sizeof(pp));
    for (int g = 0; g < ext->n_groups; ++g) {
        if (recs[g].j < 11) ++pp.n_near; else if (recs[g].i < 11) ++pp.n_str; else ++pp.n_high;
    }
    pp.dsum = dsum;
    uint16_t rank_low[2048];
    for (int w = 0; w < 2048; ++w) {
        uint64_t r = 0;
        int k = 0;
        for (int b = 0; b < 11; ++b) if ((w >> b) & 1) r += binom(b, ++k);
        rank_low[w] = (uint16_t)r;
    }
    int const nbits = wide ? 64 : 32;
    uint32_t bin[64 * LSK_PAIR_KC]; /* (entries no valid state of a basis with < 2^32 states can touch are truncated: never read for a live lane) */
    for (int a = 0; a < nbits; ++a) for (int k = 0; k < LSK_PAIR_KC; ++k) bin[a * LSK_PAIR_KC + k] = (uint32_t)binom(a, k);
    void *q = NULL;
    if (!wide) {
        if (lsk_malloc(&q, 4 * (size_t)n) != 0) return 0; /* no room for the 4-byte states: the generic kernel */
        pl->d_states32 = q;
    }
    if (upload(&pl->d_pair_recs, recs, sizeof(lsk_pair) * (size_t)ext->n_groups) != 0 || upload(&pl->d_rank_low, rank_low, sizeof(rank_low)) != 0 ||
        upload(&pl->d_pair_binom, bin, sizeof(uint32_t) * (size_t)nbits * LSK_PAIR_KC) != 0)
        return -1;
    if (!wide) DEV(lsk_narrow_states(n, d_reps, (uint32_t *)pl->d_states32, stream));
    if (by_row) {
        lsk_pair_row rr[LSK_MAX_PAIRS];
        memset(rr, 0, sizeof(rr));
        for (int g = 0; g < ext->n_groups; ++g) {
            int const i = recs[g].i, j = recs[g].j;
            rr[g].i = i;
            rr[g].j = j;
            rr[g].between = ((1ULL << j) - 1) & ~((2ULL << i) - 1);
            rr[g].v = recs[g].v;
            rr[g].vz = recs[g].vz;
        }
        if (upload(&pl->d_pair_rows, rr, sizeof(lsk_pair_row) * (size_t)ext->n_groups) != 0) return -1;
        pp.rows = (lsk_pair_row const *)pl->d_pair_rows;
        /* ... or walking the particles of the row: weight x degree candidates instead of all pairs (k_pairs_site is bound by VALU
         * issue, ~30 instructions per candidate against ~47 per pair of k_pairs_row).  LS_AMD_ROW_KERNEL=pairsites forces it while
         * the degree fits the table, =pairrows keeps the pair loop. */
        int deg[64] = {0}, dmax = 0;
        for (int g = 0; g < ext->n_groups; ++g) { ++deg[recs[g].i]; ++deg[recs[g].j]; }
        for (int q = 0; q < L; ++q) if (deg[q] > dmax) dmax = deg[q];
        int const D = dmax <= 4 ? 4 : 8;
        if (dmax <= LSK_PAIR_SITE_MAX_DEGREE && !(e && strcmp(e, "pairrows") == 0) && (force_sites || 2 * hw * D <= 3 * ext->n_groups)) {
            /* amplitude classes: distinct (v, vz) */
            double cv[LSK_PAIR_SITE_MAX_CLASSES], cz[LSK_PAIR_SITE_MAX_CLASSES];
            uint8_t cls_of[LSK_MAX_PAIRS];
            int n_cls = 0, ok = 1;
            for (int g = 0; g < ext->n_groups && ok; ++g) {
                int c = 0;
                while (c < n_cls && !(cv[c] == recs[g].v && cz[c] == recs[g].vz)) ++c;
                if (c == n_cls) {
                    if (n_cls == LSK_PAIR_SITE_MAX_CLASSES) { ok = 0; break; }
                    cv[c] = recs[g].v;
                    cz[c] = recs[g].vz;
                    ++n_cls;
                }
                cls_of[g] = (uint8_t)c;
            }
            if (ok) {
                int const DW = D / 4, amp_at = (2 * L * DW + 3) & ~3, words = amp_at + 4 * n_cls;
                uint32_t *tab = (uint32_t *)calloc((size_t)words, sizeof(uint32_t));
                if (!tab) return -1;
                uint8_t *nb = (uint8_t *)tab, *cl = (uint8_t *)(tab + L * DW);
                for (int q = 0; q < L; ++q) for (int d = 0; d < D; ++d) nb[q * D + d] = (uint8_t)q; /* padding: the site itself */
                int fill[64] = {0};
                for (int g = 0; g < ext->n_groups; ++g) {
                    int const ends[2] = {recs[g].i, recs[g].j};
                    for (int side = 0; side < 2; ++side) {
                        int const p = ends[side], d = fill[p]++;
                        nb[p * D + d] = (uint8_t)ends[1 - side];
                        cl[p * D + d] = cls_of[g];
                    }
                }
                double *amp = (double *)(tab + amp_at);
                for (int c = 0; c < n_cls; ++c) { amp[2 * c] = cv[c]; amp[2 * c + 1] = cz[c]; }
                int const up = upload(&pl->d_pair_sites, tab, sizeof(uint32_t) * (size_t)words);
                free(tab);
                if (up != 0) return -1;
                pp.sites = (uint32_t const *)pl->d_pair_sites;
                pp.n_classes = n_cls;
                pp.site_words = words;
                pp.n_sites = L;
                pp.degree = D;
            }
        }
    }
    pp.pairs = (lsk_pair const *)pl->d_pair_recs;
    pp.rank_low = (uint16_t const *)pl->d_rank_low;
    pp.binom = (uint32_t const *)pl->d_pair_binom;
    pp.states = wide ? (void const *)d_reps : (void const *)pl->d_states32;
    pp.wide = wide;
    if (build_tilemap(pl, n, by_row ? 256 : lsk_pairs_tile_rows(pl->cplx)) != 0) return -1;
    pl->pairs = pp;
    pl->has_pairs = 1;
    return 0;
}

/* Pre-indexed packets: for a hash partition of an UNPROJECTED fixed-weight basis every rank can
 * derive, alone, where any state sits inside its owner's block -- global colex rank (closed form) -> entry [rank / 64][owner] of the
 * all-destinations directory -> prefix + popcount.  The producer then sends (u32 index, value) = 12 bytes (20 for c128) instead
 * of (u64 state, value) = 16 (24), the exchange shrinks by a quarter (a sixth), and the consumer is one streaming read + one
 * atomic per packet.  Costs P / 4 bytes of HBM per basis state on every rank (chain_32 at 8 ranks: 1.2 GB; the per-rank data of the
 * packet strategy stays O(N / P) otherwise), so it is taken only while it fits LS_AMD_PACKET_INDEX_MAX bytes (default: a quarter
 * of the free HBM); LS_AMD_PACKET_INDEX=0 keeps the state-carrying packets (also the path of every projected basis, whose
 * representatives no closed form ranks), =1 takes the indexed ones also for logical partitions inside one process. */
static int packets_wave_rings(void);
/* dist.c: the packet layout is one decision of all ranks -- a rank whose peers could not build the directory re-creates its plan
 * with state-carrying packets (thread-local: loop-back ranks are threads) */
static __thread int g_no_packet_index = 0;
void ls_amd_internal_set_no_packet_index(int v) { g_no_packet_index = v; }
/* bytes of the key array (states or indices) of a segment of c packets: the values behind it stay 8-byte aligned */
static int64_t segment_key_bytes(ls_amd_plan const *pl, int64_t c) { return pl->key_bytes == 4 ? ((4 * c + 7) & ~(int64_t)7) : 8 * c; }
/* Sorted packet streams (k_packets.hip, k_tile_st / k_window) for the partitions of ONE process: every off-diagonal group must be an
 * exchange pair -- a packet exists iff alpha is anti-aligned on the pair, and beta - alpha is then one of two constants -- on an
 * unprojected fixed-weight basis (the conditions of the pre-indexed packets, which setup_packet_index checks).
 * LS_AMD_PACKET_STREAMS=0: the atomics of lsk_scatter_parts / lsk_scatter_idx instead (A/B). */
/* A plan that owns ONE partition is driven by generate / exchange / scatter: only a driver that knows the streams (dist.c: the
 * stream offsets of every source are exchanged at set-up, the own segment is consumed out of the send buffer) asks for them */
static __thread int g_want_streams = 0;
void ls_amd_internal_set_want_streams(int v) { g_want_streams = v; }
/* the static part of the conditions (operator and basis): dist.c sizes its rounds by it before the plan exists */
int ls_amd_internal_streams_eligible(ls_hs_operator const *op, int P) {
    char const *e = getenv("LS_AMD_PACKET_STREAMS");
    if (e && atoi(e) == 0) return 0;
    e = getenv("LS_AMD_PACKET_INDEX"); /* (the streams are made of pre-indexed packets: switched off with them) */
    if (e && atoi(e) == 0) return 0;
    ls_hs_basis const *b = op->basis;
    /* (P == 1: every packet is an own-partition packet, whose atomics in the producer meet ~8 packets per line of y -- measured on
     * chain_32 with one rank: 97 ms with them against 148 ms through 2 x 119 GB of streams) */
    if (BEXT(b)->order > 1 || b->spin_inversion != 0 || BEXT(b)->hamming_weight < 0 || P < 2 || P > LSK_MAX_SEGS) return 0;
    struct ls_amd_operator_ext const *oe = OEXT(op);
    if (oe->n_groups < 1 || oe->n_groups > 128 || P * 2 * oe->n_groups > lsk_tile_st_max_classes()) return 0;
    for (int g = 0; g < oe->n_groups; ++g)
        if (oe->groups[g].fast != LSK_GROUP_EXCHANGE || __builtin_popcountll(oe->groups[g].x) != 2) return 0;
    return 1;
}
static __thread int g_no_streams = 0; /* set while a plan whose stream buffers did not fit is created again with the atomic consumers */
static int streams_wanted(ls_amd_plan const *pl) {
    if (g_no_streams || (pl->me >= 0 && !g_want_streams) || pl->family != FAMILY_TILE || pl->dbs.proj != LSK_PROJ_NONE) return 0;
    return ls_amd_internal_streams_eligible(pl->op, pl->P);
}
static int setup_packet_index(ls_amd_plan *pl, uint64_t const *const *d_reps, int64_t const *counts, void *stream) {
    ls_hs_basis const *b = pl->op->basis;
    int const L = b->number_sites, h = BEXT(b)->hamming_weight, P = pl->P;
    char const *e = getenv("LS_AMD_PACKET_INDEX");
    int const for_streams = streams_wanted(pl); /* the streams need every packet's index at its destination */
    if ((e && atoi(e) == 0) || g_no_packet_index) return 0;
    /* Default: only where packets cross a wire (one partition per process).  With all partitions in one process the "exchange" is a
     * pointer hand-off, and on one device the directory costs the producers more than it saves the consumers (chain_28 x 8: 23.5
     * against 22.6 ms, chain_30 x 8: 97.7 against 92.2 -- profiles/r5_packets_preindexed_ab.txt: the consumers are bound by their
     * atomics, not by the rank directory look-up they lose).  LS_AMD_PACKET_INDEX=1 forces it there too (tests, A/B). */
    if (pl->me < 0 && !(e && atoi(e) == 1) && !for_streams) return 0;
    if (pl->dbs.proj == LSK_PROJ_FULL || h < 0 || h >= LSK_BINOM_K - 1 || L > 64 || P > lsk_tile_wv_max_parts() || P > LSK_MAX_SEGS ||
        !packets_wave_rings()) return 0;
    for (int i = 0; i < pl->n_local; ++i) if (counts[i] >= 0xffffffffLL) return 0;
    int const sites = L - (b->spin_inversion != 0 ? 1 : 0); /* inversion sectors: the canonical states have the top site bit clear */
    uint64_t const n_ranks = binom(sites, h);
    if (n_ranks == 0 || n_ranks >= ((uint64_t)1 << 40)) return 0;
    int64_t const words = (int64_t)((n_ranks + 63) / 64);
    size_t const bytes = sizeof(lsk_rankdir) * (size_t)words * (size_t)P;
    size_t fr = 0, tot = 0, ceiling;
    if (lsk_mem_info(&fr, &tot) != 0) return 0;
    e = getenv("LS_AMD_PACKET_INDEX_MAX");
    ceiling = e && atoll(e) > 0 ? (size_t)atoll(e) : fr / 4;
    uint64_t const *d_binom;
    if (device_binom(&d_binom) != 0) return -1;
    /* Sorted streams with GLOBAL-RANK keys (round 6): the key of a packet is the colex rank of beta among all states of
     * the weight -- the producer needs NO directory (the rank is alpha's +- one binomial on adjacent pairs, a rank sum otherwise) and
     * the consumer translates rank -> row through its OWN partition's rank directory (lsk_rankdir, built with the partition's index:
     * 1 / 4 byte per global state whatever P).  The all-destinations directory -- P / 4 bytes per global state on EVERY rank, 1.2 GB
     * for chain_32 at 8 ranks -- is then not built at all.  Needs < 2^32 global ranks (u32 keys); LS_AMD_STREAM_KEYS=index keeps the
     * round-5 form (index at the destination out of the all-destinations directory). */
    e = getenv("LS_AMD_STREAM_KEYS");
    if (for_streams && n_ranks < 0xffffffffULL && !(e && strcmp(e, "index") == 0)) {
        memset(&pl->gd, 0, sizeof(pl->gd));
        pl->gd.entries = NULL; pl->gd.P = P; pl->gd.sites = L; pl->gd.weight = h; pl->gd.n_ranks = (int64_t)n_ranks;
        pl->key_bytes = 4;
        pl->stream_gkeys = 1;
        pl->streams = 1;
        pl->st_S = 2 * OEXT(pl->op)->n_groups;
        pl->st_tile_rows = 256;
        return 0;
    }
    if (bytes > ceiling) return 0; /* does not fit: the packets carry the state and the consumers rank it (O(N / P) memory) */
    void *p = NULL;
    if (lsk_malloc(&p, bytes) != 0) return 0;
    lsk_gdir gd;
    memset(&gd, 0, sizeof(gd));
    gd.entries = (lsk_rankdir const *)p; gd.P = P; gd.sites = L; gd.weight = h; gd.n_ranks = (int64_t)n_ranks;
    int zero = 0, flag = 1;
    int ok = lsk_h2d(pl->d_err, &zero, sizeof(int)) == 0 && lsk_gdir_build(gd, (lsk_rankdir *)p, d_binom, stream) == 0;
    /* self-check: every state this process owns must come back as (its partition, its position) */
    for (int i = 0; ok && i < pl->n_local; ++i)
        ok = lsk_gdir_check(gd, pl->me < 0 ? i : pl->me, counts[i], d_reps[i], d_binom, pl->d_err, stream) == 0;
    ok = ok && lsk_sync(stream) == 0 && lsk_d2h(&flag, pl->d_err, sizeof(int)) == 0 && flag == 0;
    (void)lsk_h2d(pl->d_err, &zero, sizeof(int));
    if (!ok) { lsk_free(p); return 0; } /* not the full fixed-weight basis (or no room for the scans): state-carrying packets */
    pl->gd = gd;
    pl->d_gdir = (lsk_rankdir *)p;
    pl->key_bytes = 4;
    if (for_streams) {
        pl->streams = 1;
        pl->st_S = 2 * OEXT(pl->op)->n_groups;
        pl->st_tile_rows = 256;
    }
    return 0;
}

static int plan_setup_part(ls_amd_plan *pl, part_state *ps, int part_id, int num_rounds, void *stream) {
    ls_hs_basis const *b = pl->op->basis;
    int const L = b->number_sites;
    int const h = BEXT(b)->hamming_weight;
    uint64_t const *d_binom;
    if (device_binom(&d_binom) != 0) return -1;
    ps->index.count = ps->count;
    ps->index.reps = ps->d_reps;
    ps->index.binom = d_binom;
    ps->index.table = NULL;
    ps->index.shift = 0;
    ps->index.kind = LSK_INDEX_SEARCH;
    int closed_form = 0;
    if (pl->P == 1 && pl->dbs.proj != LSK_PROJ_FULL && pl->family != FAMILY_TILE) {
        int const Leff = L - (b->spin_inversion != 0 ? 1 : 0);
        if (h < 0) {
            if (Leff < 63 && ps->count == ((int64_t)1 << Leff)) { ps->index.kind = LSK_INDEX_IDENTITY; closed_form = 1; }
        } else if ((uint64_t)ps->count == binom(Leff, h) && ps->count > 0) {
            /* the basis should be the full fixed-Hamming prefix: verify on the device */
            int zero = 0, flag = 0;
            DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
            DEV(lsk_check_combinadic(ps->index, h, ps->count, ps->d_reps, pl->d_err, stream));
            DEV(lsk_sync(stream));
            DEV(lsk_d2h(&flag, pl->d_err, sizeof(int)));
            DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
            if (!flag) { ps->index.kind = LSK_INDEX_COMBINADIC; closed_form = 1; }
        }
    }
    if (pl->P == 1 && pl->family == FAMILY_TILE && pl->dbs.proj == LSK_PROJ_NONE && b->spin_inversion == 0 && h >= 0 &&
        (uint64_t)ps->count == binom(L, h) && ps->count > 0) {
        /* ONE rank drives this plan through generate / exchange / scatter (dist.c) and owns the whole basis: every packet is its own.
         * When the basis is the full fixed-weight set the producer can be the staged push kernel (k_push_t), which scatters them
         * itself -- verified here, decided at the end of the set-up (ls_amd_plan_create). */
        int zero = 0, flag = 0;
        lsk_index cix = ps->index;
        cix.kind = LSK_INDEX_COMBINADIC;
        DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
        DEV(lsk_check_combinadic(cix, h, ps->count, ps->d_reps, pl->d_err, stream));
        DEV(lsk_sync(stream));
        DEV(lsk_d2h(&flag, pl->d_err, sizeof(int)));
        DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
        pl->one_rank_full_basis = !flag;
    }
    if (!closed_form && ps->count > 0) {
        if (build_search_index(ps, L, stream) != 0) return -1;
    }
    /* Packet plans over a hash partition of an unprojected fixed-weight basis: the global rank of a state is closed-form, the
     * local index is a popcount away (lsk_rankdir) -- one 16-byte load per packet in k_scatter / k_tile_wv instead of the prefix
     * table and the 3-4 dependent probes of the binary search.  C(L, h) / 4 bytes per partition; the search index stays as the
     * fallback (no room, a state of another weight) and for every other caller.  chain_28 x 8 partitions 23.7 -> 22.3 ms, x 2
     * partitions 12.0 -> 10.2 ms, chain_30 x 8 98.0 -> 91.7 ms (profiles/r4_packets_rank_directory_ab.txt).  (Inversion sectors qualify: the canonical
     * state of a pair has the same weight at half filling, which is the only filling they exist at.) */
    if (pl->family == FAMILY_TILE && !pl->d_gdir /* pre-indexed packets need no per-partition directory */ &&
        pl->dbs.proj != LSK_PROJ_FULL && h >= 0 && h < LSK_BINOM_K - 1 && L <= 64 && ps->count > 0 &&
        ps->count < 0xffffffffLL && ps->index.kind == LSK_INDEX_SEARCH) {
        uint64_t const n_global = binom(L, h);
        if (n_global > 0 && n_global < ((uint64_t)1 << 40)) {
            int64_t const entries = (int64_t)((n_global + 63) / 64);
            void *p = NULL;
            if (lsk_malloc(&p, sizeof(lsk_rankdir) * (size_t)entries) == 0) {
                int zero = 0, flag = 1;
                if (lsk_h2d(pl->d_err, &zero, sizeof(int)) == 0 &&
                    lsk_rankdir_build(ps->count, ps->d_reps, L, h, d_binom, entries, (lsk_rankdir *)p, pl->d_err, stream) == 0 &&
                    lsk_d2h(&flag, pl->d_err, sizeof(int)) == 0 && flag == 0) {
                    ps->d_dir = (lsk_rankdir *)p;
                    ps->index.dir = ps->d_dir;
                    ps->index.dir_sites = L;
                    ps->index.dir_weight = h;
                } else lsk_free(p);
                (void)lsk_h2d(pl->d_err, &zero, sizeof(int));
            }
        }
    }
    if (pl->family == FAMILY_TILE_PULL) {
        void *p;
        DEV(lsk_malloc(&p, 8 * (size_t)(ps->count > 0 ? ps->count : 1)));
        ps->d_norms = (double *)p;
        DEV(lsk_norms(pl->dbs, ps->count, ps->d_reps, ps->d_norms, stream));
        ps->rounds = 1;
        return 0;
    }
    if (pl->family != FAMILY_TILE) return 0;

    if (pl->dbs.proj == LSK_PROJ_FULL) {
        void *p;
        DEV(lsk_malloc(&p, 8 * (size_t)(ps->count > 0 ? ps->count : 1)));
        ps->d_norms = (double *)p;
        DEV(lsk_norms(pl->dbs, ps->count, ps->d_reps, ps->d_norms, stream));
    }
    int rounds = num_rounds;
    if (rounds <= 0) {
        int64_t rpr = rows_per_round_default();
        rounds = (int)((ps->count + rpr - 1) / rpr);
        if (rounds < 1) rounds = 1;
    }
    ps->rounds = rounds;
    int const P = pl->P;
    ps->send_counts = (int64_t *)calloc((size_t)rounds * P, sizeof(int64_t));
    ps->h_beta_off = (int64_t *)calloc((size_t)rounds * P, sizeof(int64_t));
    ps->h_val_off = (int64_t *)calloc((size_t)rounds * P, sizeof(int64_t));
    lsk_round_layout *layouts = (lsk_round_layout *)calloc(rounds, sizeof(lsk_round_layout));
    int const w = pl->cplx ? 16 : 8;
    unsigned long long hc[LSK_MAX_PARTS];
    /* the per-wave producer (deterministic send layout) whenever lane d of a wave can stand for destination d */
    int const wave_rings = !pl->streams && P <= lsk_tile_wv_max_parts() && packets_wave_rings();
    uint32_t *h_wtab = NULL;
    int64_t n_waves = 0;
    /* sorted streams: count pass per (tile, destination, stream); the host turns the counts into positions inside the destination's
     * segment, streams one after the other, tiles in row order inside a stream */
    uint32_t *h_ttab = NULL, *h_soff = NULL, *sbase = NULL;
    uint64_t *running = NULL;
    int const S = pl->st_S, C = P * S, TR = pl->st_tile_rows;
    int64_t n_tiles = 0;
    if (pl->streams) {
        uint64_t const *d_binom;
        if (device_binom(&d_binom) != 0) { free(layouts); return -1; }
        ps->ttab_first = (int64_t *)calloc((size_t)rounds + 1, sizeof(int64_t));
        for (int r = 0; r < rounds; ++r) {
            int64_t row0 = ps->count * r / rounds, row1 = ps->count * (r + 1) / rounds;
            ps->ttab_first[r] = n_tiles;
            n_tiles += (row1 - row0 + TR - 1) / TR;
        }
        ps->ttab_first[rounds] = n_tiles;
        void *pt;
        size_t const bytes = sizeof(uint32_t) * (size_t)(n_tiles > 0 ? n_tiles : 1) * (size_t)C;
        if (lsk_malloc(&pt, bytes) != 0) { free(layouts); return dev_error(); }
        ps->d_ttab = (uint32_t *)pt;
        if (lsk_memset_async(pt, 0, bytes, stream) != 0) { free(layouts); return dev_error(); }
        for (int r = 0; r < rounds; ++r) {
            int64_t row0 = ps->count * r / rounds, row1 = ps->count * (r + 1) / rounds;
            if (lsk_tile_st(pl->dop, pl->gd, d_binom, pl->cplx, 1, P, S, TR, row0, row1, ps->d_reps, NULL,
                            ps->d_ttab + (size_t)ps->ttab_first[r] * C, NULL, NULL, pl->d_err, stream) != 0) { free(layouts); return dev_error(); }
        }
        h_ttab = (uint32_t *)malloc(bytes);
        h_soff = (uint32_t *)calloc((size_t)rounds * P * (S + 1), sizeof(uint32_t));
        sbase = (uint32_t *)calloc((size_t)C, sizeof(uint32_t));
        running = (uint64_t *)calloc((size_t)C, sizeof(uint64_t));
        if (!h_ttab || !h_soff || !sbase || !running || lsk_sync(stream) != 0 || lsk_d2h(h_ttab, pt, bytes) != 0) {
            free(h_ttab); free(h_soff); free(sbase); free(running); free(layouts);
            return dev_error();
        }
    }
    if (wave_rings) {
        ps->wtab_first = (int64_t *)calloc((size_t)rounds + 1, sizeof(int64_t));
        for (int r = 0; r < rounds; ++r) {
            int64_t row0 = ps->count * r / rounds, row1 = ps->count * (r + 1) / rounds;
            ps->wtab_first[r] = n_waves;
            n_waves += (row1 - row0 + 63) / 64;
        }
        ps->wtab_first[rounds] = n_waves;
        void *pw;
        size_t const bytes = sizeof(uint32_t) * (size_t)(n_waves > 0 ? n_waves : 1) * (size_t)P;
        if (lsk_malloc(&pw, bytes) != 0) { free(layouts); return dev_error(); }
        ps->d_wtab = (uint32_t *)pw;
        if (lsk_memset_async(pw, 0, bytes, stream) != 0) { free(layouts); return dev_error(); }
        for (int r = 0; r < rounds; ++r) {
            int64_t row0 = ps->count * r / rounds, row1 = ps->count * (r + 1) / rounds;
            if (lsk_tile_wv(pl->dop, pl->dbs, ps->index, pl->gd, pl->cplx, 1, P, part_id, row0, row1, ps->d_reps, ps->d_norms, NULL, NULL,
                            ps->d_wtab + (size_t)ps->wtab_first[r] * P, NULL, NULL, pl->d_err, no_gtab(), stream) != 0) { free(layouts); return dev_error(); }
        }
        h_wtab = (uint32_t *)malloc(bytes);
        if (!h_wtab || lsk_sync(stream) != 0 || lsk_d2h(h_wtab, pw, bytes) != 0) { free(h_wtab); free(layouts); return dev_error(); }
    }
    for (int r = 0; r < rounds; ++r) {
        int64_t row0 = ps->count * r / rounds, row1 = ps->count * (r + 1) / rounds;
        if (pl->streams) {
            memset(running, 0, sizeof(uint64_t) * (size_t)C);
            for (int64_t t = ps->ttab_first[r]; t < ps->ttab_first[r + 1]; ++t) { /* exclusive prefix over the tiles, class by class */
                uint32_t *row = h_ttab + (size_t)t * C;
                for (int c = 0; c < C; ++c) { uint32_t const k = row[c]; row[c] = (uint32_t)running[c]; running[c] += k; }
            }
            int too_many = 0;
            for (int d = 0; d < P; ++d) { /* the streams of a segment one after the other */
                uint64_t base = 0;
                uint32_t *so = h_soff + ((size_t)r * P + d) * (S + 1);
                for (int q = 0; q < S; ++q) {
                    so[q] = (uint32_t)base;
                    sbase[d * S + q] = (uint32_t)base;
                    base += running[d * S + q];
                    if (base > 0xffffffffULL) too_many = 1;
                }
                so[S] = (uint32_t)base;
                hc[d] = base;
            }
            if (too_many) {
                free(h_ttab); free(h_soff); free(sbase); free(running); free(layouts);
                return set_error("more than 2^32 packets for one destination in one round: raise the number of rounds");
            }
            for (int64_t t = ps->ttab_first[r]; t < ps->ttab_first[r + 1]; ++t) {
                uint32_t *row = h_ttab + (size_t)t * C;
                for (int c = 0; c < C; ++c) row[c] += sbase[c];
            }
        } else if (wave_rings) {
            /* counts -> exclusive offsets along the waves of the round, destination by destination; the own partition's
             * packets never enter the send buffer */
            for (int d = 0; d < P; ++d) hc[d] = 0;
            for (int64_t wv = ps->wtab_first[r]; wv < ps->wtab_first[r + 1]; ++wv) {
                uint32_t *row = h_wtab + (size_t)wv * P;
                for (int d = 0; d < P; ++d) {
                    uint32_t const c = row[d];
                    if (d != part_id && hc[d] + c > 0xffffffffULL) { free(h_wtab); free(layouts); return set_error("more than 2^32 packets for one destination in one round: raise the number of rounds"); }
                    row[d] = (uint32_t)hc[d];
                    hc[d] += c;
                }
            }
        } else {
            if (lsk_memset_async(pl->d_counts, 0, 8 * LSK_MAX_PARTS, stream) != 0 ||
                lsk_tile(pl->dop, pl->dbs, ps->index, pl->cplx, 1, P, part_id, row0, row1, ps->d_reps, ps->d_norms,
                         NULL, NULL, pl->d_cursors, NULL, NULL, pl->d_counts, pl->d_err, stream) != 0 ||
                lsk_sync(stream) != 0 || lsk_d2h(hc, pl->d_counts, 8 * (size_t)P) != 0) { free(layouts); return dev_error(); }
        }
        int64_t off = 0;
        for (int d = 0; d < P; ++d) {
            pl->nnz += (int64_t)hc[d];
            int64_t c = (d == part_id && !pl->streams) ? 0 : (int64_t)hc[d]; /* (sorted streams: the own partition's packets take the buffer too) */
            ps->send_counts[(size_t)r * P + d] = c;
            int64_t const keys = segment_key_bytes(pl, c); /* u64 states, or u32 indices padded to 8 bytes */
            layouts[r].beta_off[d] = off;
            layouts[r].val_off[d] = off + keys;
            ps->h_beta_off[(size_t)r * P + d] = off;
            ps->h_val_off[(size_t)r * P + d] = off + keys;
            off += keys + w * c;
        }
        if (off > ps->max_send_bytes) ps->max_send_bytes = off;
    }
    if (wave_rings) {
        int const rcw = lsk_h2d(ps->d_wtab, h_wtab, sizeof(uint32_t) * (size_t)(n_waves > 0 ? n_waves : 1) * (size_t)P);
        free(h_wtab);
        if (rcw != 0) { free(layouts); return dev_error(); }
    }
    if (pl->streams) {
        void *po = NULL;
        size_t const sob = sizeof(uint32_t) * (size_t)rounds * (size_t)P * (size_t)(S + 1);
        int const rcs = lsk_h2d(ps->d_ttab, h_ttab, sizeof(uint32_t) * (size_t)(n_tiles > 0 ? n_tiles : 1) * (size_t)C) != 0 ||
                        lsk_malloc(&po, sob) != 0 || lsk_h2d(po, h_soff, sob) != 0;
        ps->d_soff = (uint32_t *)po; /* owned by the plan from here on */
        ps->h_soff = h_soff;
        free(h_ttab); free(sbase); free(running);
        if (rcs != 0) { free(layouts); return dev_error(); }
    }
    void *p;
    if (lsk_malloc(&p, sizeof(lsk_round_layout) * (size_t)rounds) != 0) { free(layouts); return dev_error(); }
    ps->d_layouts = (lsk_round_layout *)p; /* owned by the plan from here on */
    int const rc = lsk_h2d(p, layouts, sizeof(lsk_round_layout) * (size_t)rounds);
    free(layouts);
    return rc != 0 ? dev_error() : 0;
}

/* windows of y per consumer block: the end of one window's run is the start of the next one's (one binary search saved per
 * stream), as long as >= 8192 blocks keep the 2048 block slots busy without a long tail.  (Measured on chain_28 x 8: 1 / 4 / 8
 * windows per block 2.96 / 2.86 / 3.06 ms -- no knob; the test hook forces the carry path on small bases.) */
static __thread int g_test_stream_wpb = 0;
void ls_amd_test_set_stream_windows_per_block(int n) { g_test_stream_wpb = n; }
static int stream_windows_per_block(int64_t windows) {
    if (g_test_stream_wpb > 0) return g_test_stream_wpb;
    int64_t const w = windows / 8192;
    return w < 1 ? 1 : (w > 8 ? 8 : (int)w);
}
/* sorted streams, all partitions in this process: one send buffer per source partition (a round of all sources is consumed by
 * ONE launch, so y is read and written once per round) and the consumer's view of every (round, destination, source) segment */
static __thread int g_test_fail_stream_buffers = 0;
void ls_amd_test_fail_stream_buffers(int on) { g_test_fail_stream_buffers = on; }
static int setup_streams(ls_amd_plan *pl, int rounds) {
    int const P = pl->P, S = pl->st_S;
    if (g_test_fail_stream_buffers) return set_error("test hook: no room for the stream buffers");
    pl->st_rounds = rounds;
    pl->d_send_parts = (void **)calloc((size_t)P, sizeof(void *));
    for (int p = 0; p < P; ++p) {
        if (pl->parts[p].rounds != rounds) return set_error("internal error: partitions disagree on the number of rounds");
        if (lsk_malloc(&pl->d_send_parts[p], (size_t)(pl->parts[p].max_send_bytes > 0 ? pl->parts[p].max_send_bytes : 8)) != 0) return dev_error();
    }
    size_t const n = (size_t)rounds * (size_t)P * (size_t)P;
    lsk_wsrc *h = (lsk_wsrc *)calloc(n, sizeof(lsk_wsrc));
    for (int r = 0; r < rounds; ++r)
        for (int d = 0; d < P; ++d)
            for (int q = 0; q < P; ++q) {
                part_state const *src = &pl->parts[q];
                lsk_wsrc *w = h + ((size_t)r * P + d) * P + q;
                w->keys = (uint32_t const *)((char const *)pl->d_send_parts[q] + src->h_beta_off[(size_t)r * P + d]);
                w->vals = (double const *)((char const *)pl->d_send_parts[q] + src->h_val_off[(size_t)r * P + d]);
                w->soff = src->d_soff + ((size_t)r * P + d) * (size_t)(S + 1);
            }
    void *pw = NULL;
    int const bad = lsk_malloc(&pw, sizeof(lsk_wsrc) * n) != 0 || lsk_h2d(pw, h, sizeof(lsk_wsrc) * n) != 0;
    free(h);
    pl->d_wsrcs = (lsk_wsrc *)pw;
    if (bad) return dev_error();
    /* windows per block: the end of one window's run is the start of the next one's (one binary search saved per stream) as long
     * as the launch still has several blocks per CU */
    int64_t windows = 0;
    int const W = lsk_window_rows(pl->cplx);
    for (int d = 0; d < P; ++d) windows += (pl->parts[d].count + W - 1) / W;
    pl->st_wpb = stream_windows_per_block(windows);
    return 0;
}

int ls_amd_plan_create(ls_amd_plan **out, ls_hs_operator const *op, ls_amd_dtype dtype,
                       int num_partitions, int my_partition, uint64_t const *const *d_reps,
                       int64_t const *counts, int num_rounds, ls_amd_mode mode, void *stream) {
    *out = NULL;
    if (!op || !op->basis) return set_error("null operator");
    if (ls_hs_basis_number_words(op->basis) != 1) return set_error("bases with more than 64 bits are not yet implemented"); /* DMV:1099 */
    if (num_partitions < 1 || num_partitions > LSK_MAX_PARTS) return set_error("num_partitions must be in [1, %d]", LSK_MAX_PARTS);
    if (my_partition >= num_partitions) return set_error("my_partition out of range");
    if (dtype == LS_AMD_F64 && !OEXT(op)->is_real) return set_error("an operator with complex coefficients needs dtype c128");
    ls_amd_plan *pl = (ls_amd_plan *)calloc(1, sizeof(*pl));
    pl->op = op;
    pl->cplx = dtype == LS_AMD_C128;
    pl->P = num_partitions;
    pl->me = my_partition;
    pl->n_local = my_partition < 0 ? num_partitions : 1;
    if (operator_device(op, &pl->dop) != 0 || basis_device(op->basis, &pl->dbs) != 0) { free(pl); return -1; }
    if (dtype == LS_AMD_F64) {
        /* f64 vectors: characters must be real as well (the reference casts c128 -> f64, DMV:91,109) */
        for (int g = 0; g < BEXT(op->basis)->order; ++g)
            if (BEXT(op->basis)->elems[g].ch_im != 0.0) { free(pl); return set_error("complex characters need dtype c128"); }
    }
    /* kernel family */
    /* a plan that owns ONE partition is driven by generate / exchange / scatter (one locale per process): always the
     * packet kernels, also when numLocales == 1 (the reference's matrixVectorProduct works there too, DMV:1072-1093) */
    int const force_tile = my_partition >= 0;
    if (force_tile) pl->family = FAMILY_TILE;
    else if (num_partitions == 1 && pl->dbs.proj != LSK_PROJ_FULL) {
        ls_amd_mode m = mode;
        if (m == LS_AMD_MODE_AUTO) {
            char const *e = getenv("LS_AMD_MODE");
            if (e && strcmp(e, "pull") == 0) m = LS_AMD_MODE_PULL;
            else if (e && strcmp(e, "push") == 0) m = LS_AMD_MODE_PUSH;
            /* measured: pull is 2.6x push on chain_32.  Round 6: on an unprojected basis ANY operator can be pulled -- row i takes
             * <i|H_g|i ^ x_g> from the partner's row expansion (k_direct, k_rows.hip) -- so non-Hermitian operators leave the
             * atomics too; only the inversion sectors still pull through the Hermiticity of the projected matrix */
            else m = (OEXT(op)->is_hermitian || pl->dbs.proj == LSK_PROJ_NONE) ? LS_AMD_MODE_PULL : LS_AMD_MODE_PUSH;
        }
        if (m == LS_AMD_MODE_PULL && !OEXT(op)->is_hermitian && pl->dbs.proj != LSK_PROJ_NONE) {
            if (mode == LS_AMD_MODE_PULL) { free(pl); return set_error("pull mode needs a Hermitian operator (or an unprojected basis)"); }
            m = LS_AMD_MODE_PUSH;
        }
        pl->family = m == LS_AMD_MODE_PULL ? FAMILY_DIRECT_PULL : FAMILY_DIRECT_PUSH;
    } else if (num_partitions == 1 && pl->dbs.proj == LSK_PROJ_FULL) {
        /* projected basis on one device: staged pull (no global atomics) when H is Hermitian */
        ls_amd_mode m = mode;
        if (m == LS_AMD_MODE_AUTO) {
            char const *e = getenv("LS_AMD_MODE");
            if (e && strcmp(e, "push") == 0) m = LS_AMD_MODE_PUSH;
            else m = LS_AMD_MODE_PULL;
        }
        if (m == LS_AMD_MODE_PULL && !OEXT(op)->is_hermitian) {
            if (mode == LS_AMD_MODE_PULL) { free(pl); return set_error("pull mode needs a Hermitian operator"); }
            m = LS_AMD_MODE_PUSH;
        }
        pl->family = m == LS_AMD_MODE_PULL ? FAMILY_TILE_PULL : FAMILY_TILE;
        if (pl->family == FAMILY_TILE_PULL) {
            /* Default: the INDEXED mode -- static {rep -> index} table + x[index] (two dependent requests per far partner,
             * nothing rewritten per matvec).  Measured r3 (profiles/r3_indexed_ab_*): chain_36_symm 24.8 -> 20.4 ms,
             * chain_40_symm 373.5 -> 323.0 ms per matvec against the {rep -> x n(rep)} value table (one request per far
             * partner, but N random 16-byte writes per matvec: 2.8 / 42.5 ms), which stays as LS_AMD_PULL_INDEXED=0 and
             * as the fallback for bases whose keys do not fit the 8-byte entries (lsk_gtab_bits). */
            char const *e = getenv("LS_AMD_PULL_INDEXED");
            pl->idx_mode = e ? atoi(e) != 0 : 1;
            if (pl->idx_mode && (counts[0] >= 0xffffffffLL || lsk_gtab_bits(op->basis->number_sites, counts[0], (int64_t)1 << 40) < 0)) pl->idx_mode = 0;
        }
    } else pl->family = FAMILY_TILE;

    void *p;
    if (lsk_malloc(&p, 8 * LSK_MAX_PARTS) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    pl->d_cursors = (unsigned long long *)p;
    if (lsk_malloc(&p, 8 * LSK_MAX_PARTS) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    pl->d_counts = (unsigned long long *)p;
    if (lsk_malloc(&p, 2 * sizeof(int)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); } /* [0]: the error flag; [1]: a word nobody reads (lsk_direct) */
    pl->d_err = (int *)p;
    int zero = 0;
    int const zero2[2] = {0, 0};
    if (lsk_h2d(pl->d_err, zero2, sizeof(zero2)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }

    pl->key_bytes = 8;
    if (pl->family == FAMILY_TILE && setup_packet_index(pl, d_reps, counts, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
    pl->parts = (part_state *)calloc(pl->n_local, sizeof(part_state));
    int const num_rounds_arg = num_rounds;
    if (pl->streams && num_rounds <= 0) { /* one consumer launch per round over all sources: every partition runs the same rounds */
        int64_t mx = 0, rpr = rows_per_round_default();
        for (int i = 0; i < pl->n_local; ++i) if (counts[i] > mx) mx = counts[i];
        num_rounds = (int)((mx + rpr - 1) / rpr);
        if (num_rounds < 1) num_rounds = 1;
    }
    for (int i = 0; i < pl->n_local; ++i) {
        part_state *ps = &pl->parts[i];
        ps->count = counts[i];
        ps->d_reps = d_reps[i];
        int pid = my_partition < 0 ? i : my_partition;
        if (plan_setup_part(pl, ps, pid, num_rounds, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
        if (ps->max_send_bytes > pl->send_capacity) pl->send_capacity = ps->max_send_bytes;
    }
    int gkeys_ok = 1;
    if (pl->stream_gkeys) for (int i = 0; i < pl->n_local; ++i) if (pl->parts[i].count > 0 && !pl->parts[i].index.dir) gkeys_ok = 0;
    if (pl->streams && !gkeys_ok) {
        /* a partition without its rank directory (not the hash partition of the full fixed-weight basis, or no room): the same plan with
         * the atomic consumers */
        ls_amd_plan_destroy(pl);
        if (g_no_streams) return -1;
        g_no_streams = 1;
        int const rc = ls_amd_plan_create(out, op, dtype, num_partitions, my_partition, d_reps, counts, num_rounds_arg, mode, stream);
        g_no_streams = 0;
        return rc;
    }
    if (pl->streams && my_partition < 0 && setup_streams(pl, num_rounds) != 0) {
        /* one buffer per source partition did not fit (or a table could not be uploaded): the same plan with ONE shared buffer and
         * the atomic consumers -- the O(N / P) form must not fail where it used to work */
        ls_amd_plan_destroy(pl);
        if (g_no_streams) return -1;
        g_no_streams = 1;
        int const rc = ls_amd_plan_create(out, op, dtype, num_partitions, my_partition, d_reps, counts, num_rounds_arg, mode, stream);
        g_no_streams = 0;
        return rc;
    }
    if (pl->streams && my_partition >= 0) { /* one partition per process: the driver owns the buffers (dist.c) */
        pl->st_rounds = pl->parts[0].rounds;
        pl->st_wpb = stream_windows_per_block((pl->parts[0].count + lsk_window_rows(pl->cplx) - 1) / lsk_window_rows(pl->cplx));
    }
    if (pl->family == FAMILY_DIRECT_PULL || pl->family == FAMILY_DIRECT_PUSH) {
        int const combinadic = pl->parts[0].index.kind == LSK_INDEX_COMBINADIC;
        part_state *ps0 = &pl->parts[0];
        if (pl->family == FAMILY_DIRECT_PULL && combinadic && chain_eligible(pl) &&
            setup_chain(pl, ps0->index, ps0->count, ps0->d_reps, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
        if (!pl->has_chain && pl->family == FAMILY_DIRECT_PULL && combinadic && setup_pairs(pl, ps0->count, ps0->d_reps, stream) != 0) {
            ls_amd_plan_destroy(pl);
            return -1;
        }
        if (pl->family == FAMILY_DIRECT_PUSH && combinadic && push_staged_eligible(pl)) {
            if (build_tilemap(pl, ps0->count, lsk_push_tile_rows(pl->cplx)) != 0) { ls_amd_plan_destroy(pl); return -1; }
            pl->has_push_staged = 1;
        }
        if (!pl->has_chain && !pl->has_pairs && !pl->has_push_staged && build_tilemap(pl, ps0->count, 256) != 0) { ls_amd_plan_destroy(pl); return -1; }
        if (pl->family == FAMILY_DIRECT_PULL && !OEXT(op)->is_hermitian) {
            /* a gather never sees a state that is mapped OUT of the basis (the reference's halt, DMV:115-118): checked once, here;
             * ls_amd_plan_check reports it after every matvec like the push kernels' error flag */
            int flag = 0;
            if (lsk_direct_validate(pl->dop, pl->dbs, ps0->index, ps0->count, ps0->d_reps, pl->d_err, stream) != 0 || lsk_sync(stream) != 0 ||
                lsk_d2h(&flag, pl->d_err, sizeof(int)) != 0 || lsk_h2d(pl->d_err, &zero, sizeof(int)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
            pl->leaves_basis = flag != 0;
        }
    }
    if (pl->family == FAMILY_TILE && my_partition < 0 && pl->send_capacity > 0 && !pl->streams) {
        if (lsk_malloc(&pl->d_send, (size_t)pl->send_capacity) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    }
    if (pl->family == FAMILY_TILE && pl->P == 1 && my_partition >= 0 && pl->one_rank_full_basis && pl->parts[0].rounds == 1 &&
        push_staged_eligible(pl)) {
        /* one rank, every packet its own: the producer of the only round is the staged push kernel (round 6: chain_32 through the
         * C host's packet driver 87.6 -> see profiles/r6_staged_push_ab.txt) */
        if (build_tilemap(pl, pl->parts[0].count, lsk_push_tile_rows(pl->cplx)) != 0) { ls_amd_plan_destroy(pl); return -1; }
        pl->has_push_staged = 1;
    }
    if (pl->family == FAMILY_TILE && pl->key_bytes == 8 && pl->P > 1) {
        /* State-carrying packets into a SEARCHED index (projected bases): the consumer's look-up is one 16-byte probe of a static
         * {representative -> index} table per partition instead of the prefix table + binary search (4-6 dependent loads per packet;
         * round 6: chain_36_symm x 8 consumers 40.0 -> see profiles/r6_projected_packets_hash_index_ab.txt).  16-32 bytes per state,
         * O(N / P); a table that cannot be built leaves the search in place.  LS_AMD_SCATTER_HASH=0: off (A/B). */
        char const *he = getenv("LS_AMD_SCATTER_HASH");
        if (!(he && atoi(he) == 0))
            for (int p = 0; p < pl->n_local; ++p) {
                part_state *ps = &pl->parts[p];
                if (ps->index.kind != LSK_INDEX_SEARCH || ps->index.dir || ps->count <= 0 || ps->count >= 0xffffffffLL) continue;
                if (ls_amd_internal_gtab_acquire(&ps->scatter_gt, op->basis->number_sites, ps->d_reps, ps->count, NULL, 1, stream) != 0) {
                    ps->scatter_gt = NULL; /* (the error text stays readable through ls_amd_last_error; the plan works without the table) */
                }
            }
    }
    if (pl->family == FAMILY_TILE && my_partition < 0 && pl->P > 1 && pl->key_bytes == 8) {
        /* the destinations' indexes as a device array: all segments a producer's round leaves are consumed by ONE launch */
        lsk_part_ctx *h = (lsk_part_ctx *)calloc((size_t)pl->P, sizeof(lsk_part_ctx));
        for (int d = 0; d < pl->P; ++d) {
            h[d].ix = pl->parts[d].index;
            h[d].norms = pl->dbs.k4_mode ? pl->parts[d].d_norms : NULL;
            if (pl->parts[d].scatter_gt) h[d].gt = pl->parts[d].scatter_gt->tab;
            if (!pl->parts[pl->part_ctx_any].index.dir && pl->parts[d].index.dir) pl->part_ctx_any = d;
        }
        void *pc = NULL;
        int const bad = lsk_malloc(&pc, sizeof(lsk_part_ctx) * (size_t)pl->P) != 0 || lsk_h2d(pc, h, sizeof(lsk_part_ctx) * (size_t)pl->P) != 0;
        free(h);
        if (bad) { if (pc) lsk_free(pc); ls_amd_plan_destroy(pl); return dev_error(); }
        pl->d_part_ctx = (lsk_part_ctx *)pc;
    }
    if (pl->family == FAMILY_TILE_PULL && pl->idx_mode) {
        /* static index table + room for x * norm(rep), acquired here: a table that cannot be built (no memory, a key that
         * finds no place within 255 buckets) puts the plan on the value-table path instead of failing the first matvec */
        part_state *ps = &pl->parts[0];
        int ok = ls_amd_internal_gtab_acquire(&pl->gtab, op->basis->number_sites, ps->d_reps, ps->count, NULL, 1, stream) == 0;
        if (ok && pl->dbs.k4_mode != 0 &&
            lsk_malloc(&pl->d_xs, (size_t)(pl->cplx ? 16 : 8) * (size_t)(ps->count > 0 ? ps->count : 1)) != 0) {
            ls_amd_internal_gtab_release(pl->gtab);
            pl->gtab = NULL;
            pl->d_xs = NULL;
            ok = 0;
        }
        if (!ok) pl->idx_mode = 0;
        else {
            pl->pull_halo = pull_halo_setting(1);
            char const *e = getenv("LS_AMD_PULL_SPLIT"); /* bytes of packet buffer; measurement of the two-kernel form */
            if (e && atoll(e) > 0) ls_amd_internal_plan_split_enable(pl, atoll(e));
            /* LS_AMD_SLOT_CACHE = bytes: ls_amd_plan_cache_slots for callers that cannot reach the plan -- the host-pointer entry
             * points under Diagonalize / PRIMME (ls_chpl_matrix_vector_product, ls_chpl_primme_matvec) keep one plan per
             * operator, which then resolves its packet streams on the first matvec only */
            e = getenv("LS_AMD_SLOT_CACHE");
            if (e && atoll(e) > 0 && ls_amd_plan_cache_slots(pl, atoll(e)) < 0) { ls_amd_plan_destroy(pl); return -1; }
            /* Value table (round 6; k_pull.hip, lsk_vtab_*): f64 vectors of one partition, matrix-free plans --
             * the far partners' values travel with their table bucket (one fabric request instead of two dependent ones) and
             * are refreshed once per matvec in table order.  32 bytes per bucket next to the shared index table's 16; taken while
             * it fits a third of the free HBM.  LS_AMD_PULL_VALUES=0 keeps the index table + x[slot] gather (always the path of
             * c128 vectors, of the replicated-x exchange -- no rank may do O(N) work per matvec there -- and of the slot cache). */
            e = getenv("LS_AMD_PULL_VALUES");
            int const want_values = e ? atoi(e) != 0 : LS_AMD_PULL_VALUES_DEFAULT;
            if (want_values && !pl->cplx && !pl->slot_cache && pl->split_rows == 0) {
                size_t fr = 0, tot = 0;
                size_t const need = (size_t)32 << pl->gtab->tab.bbits;
                void *vt = NULL;
                if (lsk_mem_info(&fr, &tot) == 0 && need <= fr / 3 && lsk_malloc(&vt, need) == 0) {
                    if (lsk_vtab_build(pl->gtab->tab, (uint64_t *)vt, stream) != 0) { lsk_free(vt); ls_amd_plan_destroy(pl); return dev_error(); }
                    pl->d_vtab = (uint64_t *)vt;
                }
            }
        }
    }
    if (lsk_sync(stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    *out = pl;
    ls_amd_internal_clear_error(); /* (internal fallbacks -- a table that did not fit, a layout that does not apply -- leave no message) */
    return 0;
}

void ls_amd_plan_destroy(ls_amd_plan *pl) {
    if (!pl) return;
    if (pl->d_vtab) lsk_free(pl->d_vtab);
    if (pl->d_gdir) lsk_free(pl->d_gdir);
    if (pl->d_part_ctx) lsk_free(pl->d_part_ctx);
    if (pl->d_send_parts) {
        for (int p = 0; p < pl->P; ++p) if (pl->d_send_parts[p]) lsk_free(pl->d_send_parts[p]);
        free(pl->d_send_parts);
    }
    if (pl->d_wsrcs) lsk_free(pl->d_wsrcs);
    if (pl->parts) {
        for (int i = 0; i < pl->n_local; ++i) {
            part_state *ps = &pl->parts[i];
            if (ps->d_table) lsk_free(ps->d_table);
            if (ps->d_dir) lsk_free(ps->d_dir);
            if (ps->d_norms) lsk_free(ps->d_norms);
            if (ps->d_layouts) lsk_free(ps->d_layouts);
            if (ps->d_wtab) lsk_free(ps->d_wtab);
            free(ps->wtab_first);
            if (ps->d_ttab) lsk_free(ps->d_ttab);
            if (ps->d_soff) lsk_free(ps->d_soff);
            if (ps->scatter_gt) ls_amd_internal_gtab_release(ps->scatter_gt);
            free(ps->h_soff);
            free(ps->ttab_first);
            free(ps->send_counts); free(ps->h_beta_off); free(ps->h_val_off);
        }
        free(pl->parts);
    }
    if (pl->d_gtable) lsk_free(pl->d_gtable);
    if (pl->d_row_gidx) lsk_free(pl->d_row_gidx);
    if (pl->d_norms_global) lsk_free(pl->d_norms_global);
    if (pl->d_tilemap) lsk_free(pl->d_tilemap);
    if (pl->d_chain_cache) lsk_free(pl->d_chain_cache);
    if (pl->d_chain_rec) lsk_free(pl->d_chain_rec);
    if (pl->d_pair_recs) lsk_free(pl->d_pair_recs);
    if (pl->d_pair_rows) lsk_free(pl->d_pair_rows);
    if (pl->d_pair_sites) lsk_free(pl->d_pair_sites);
    if (pl->d_rank_low) lsk_free(pl->d_rank_low);
    if (pl->d_pair_binom) lsk_free(pl->d_pair_binom);
    if (pl->d_states32) lsk_free(pl->d_states32);
    if (pl->d_htab) lsk_free(pl->d_htab);
    if (pl->d_slot_of) lsk_free(pl->d_slot_of);
    if (pl->d_xs) lsk_free(pl->d_xs);
    if (pl->gtab) ls_amd_internal_gtab_release(pl->gtab);
    split_free(pl);
    if (pl->d_send) lsk_free(pl->d_send);
    if (pl->d_cursors) lsk_free(pl->d_cursors);
    if (pl->d_counts) lsk_free(pl->d_counts);
    if (pl->d_err) lsk_free(pl->d_err);
    ls_amd_plan_enable_timing(pl, 0);
    ls_amd_plan_enable_stage_timing(pl, 0);
    free(pl);
}

int ls_amd_plan_num_rounds(ls_amd_plan const *pl) { return pl->family == FAMILY_TILE ? pl->parts[0].rounds : 1; }

/* -------------------------------------------------------------------------------------------- */
/* replicated-x plans                                                                            */
/* -------------------------------------------------------------------------------------------- */
static int plan_create_replicated_impl(ls_amd_plan **out, ls_hs_operator const *op, ls_amd_dtype dtype,
                                       int num_partitions, int my_partition, uint64_t const *d_reps_local,
                                       int64_t count_local, uint64_t const *d_reps_global, int64_t count_global,
                                       ls_amd_gtab *gt, void *stream);
int ls_amd_plan_create_replicated(ls_amd_plan **out, ls_hs_operator const *op, ls_amd_dtype dtype,
                                  int num_partitions, int my_partition, uint64_t const *d_reps_local,
                                  int64_t count_local, uint64_t const *d_reps_global, int64_t count_global,
                                  void *stream) {
    return plan_create_replicated_impl(out, op, dtype, num_partitions, my_partition, d_reps_local, count_local, d_reps_global,
                                       count_global, NULL, stream);
}
/* the replicated-x plan of the indexed exchange (dist.c): x arrives as the owners' blocks (slot order of `gt`), the rows are
 * the contiguous global rows d_reps_local = d_reps_global + row0.  Projected bases only. */
int ls_amd_internal_plan_create_replicated_indexed(ls_amd_plan **out, ls_hs_operator const *op, ls_amd_dtype dtype,
                                                   int num_partitions, int my_partition, uint64_t const *d_reps_local,
                                                   int64_t count_local, uint64_t const *d_reps_global, int64_t count_global,
                                                   ls_amd_gtab *gt, void *stream) {
    return plan_create_replicated_impl(out, op, dtype, num_partitions, my_partition, d_reps_local, count_local, d_reps_global,
                                       count_global, gt, stream);
}
static int plan_create_replicated_impl(ls_amd_plan **out, ls_hs_operator const *op, ls_amd_dtype dtype,
                                       int num_partitions, int my_partition, uint64_t const *d_reps_local,
                                       int64_t count_local, uint64_t const *d_reps_global, int64_t count_global,
                                       ls_amd_gtab *gt, void *stream) {
    *out = NULL;
    if (!op || !op->basis) return set_error("null operator");
    if (ls_hs_basis_number_words(op->basis) != 1) return set_error("bases with more than 64 bits are not yet implemented");
    if (!OEXT(op)->is_hermitian) return set_error("replicated-x (pull) plans need a Hermitian operator");
    if (num_partitions < 1 || num_partitions > LSK_MAX_PARTS || my_partition < 0 || my_partition >= num_partitions)
        return set_error("bad partition arguments");
    if (dtype == LS_AMD_F64 && !OEXT(op)->is_real) return set_error("an operator with complex coefficients needs dtype c128");
    ls_amd_plan *pl = (ls_amd_plan *)calloc(1, sizeof(*pl));
    pl->op = op;
    pl->cplx = dtype == LS_AMD_C128;
    pl->P = num_partitions;
    pl->me = my_partition;
    pl->n_local = 1;
    if (operator_device(op, &pl->dop) != 0 || basis_device(op->basis, &pl->dbs) != 0) { free(pl); return -1; }
    if (dtype == LS_AMD_F64)
        for (int g = 0; g < BEXT(op->basis)->order; ++g)
            if (BEXT(op->basis)->elems[g].ch_im != 0.0) { free(pl); return set_error("complex characters need dtype c128"); }
    pl->family = pl->dbs.proj == LSK_PROJ_FULL ? FAMILY_REPL_TILE : FAMILY_REPL_DIRECT;
    void *p;
    if (lsk_malloc(&p, sizeof(int)) != 0) { free(pl); return dev_error(); }
    pl->d_err = (int *)p;
    int zero = 0;
    lsk_h2d(pl->d_err, &zero, sizeof(int));
    pl->parts = (part_state *)calloc(1, sizeof(part_state));
    part_state *ps = &pl->parts[0];
    ps->count = count_local;
    ps->d_reps = d_reps_local;
    ps->rounds = 1;
    if (gt) {
        /* indexed mode: the static table replaces every other index of the global basis */
        if (pl->family != FAMILY_REPL_TILE) { ls_amd_plan_destroy(pl); return set_error("indexed replicated-x plans are for projected bases"); }
        if (!(d_reps_local >= d_reps_global && d_reps_local + count_local <= d_reps_global + count_global)) {
            ls_amd_plan_destroy(pl);
            return set_error("indexed replicated-x plans need a contiguous block of the global rows");
        }
        pthread_mutex_lock(&g_gtab_lock);
        ++gt->refs;
        pthread_mutex_unlock(&g_gtab_lock);
        pl->gtab = gt;
        pl->idx_mode = 1;
        pl->row_g0 = (int64_t)(d_reps_local - d_reps_global);
        pl->gindex.kind = LSK_INDEX_SEARCH;
        pl->gindex.count = count_global;
        pl->gindex.reps = d_reps_global;
        pl->pull_halo = pull_halo_setting(1);
        if (lsk_malloc(&p, 8 * (size_t)(count_local > 0 ? count_local : 1)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        ps->d_norms = (double *)p;
        if (lsk_norms(pl->dbs, count_local, d_reps_local, ps->d_norms, stream) != 0 || lsk_sync(stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        *out = pl;
        return 0;
    }
    /* index of the GLOBAL basis: reuse the single-partition logic on a temporary part */
    {
        ls_amd_plan tmp = *pl;
        part_state gps;
        memset(&gps, 0, sizeof(gps));
        gps.count = count_global;
        gps.d_reps = d_reps_global;
        tmp.P = 1;
        tmp.family = FAMILY_DIRECT_PULL; /* no tile-side tables */
        if (pl->dbs.proj == LSK_PROJ_FULL) {
            uint64_t const *d_binom;
            if (device_binom(&d_binom) != 0) { ls_amd_plan_destroy(pl); return -1; }
            gps.index.count = count_global; gps.index.reps = d_reps_global; gps.index.binom = d_binom;
            if (count_global > 0 && build_search_index(&gps, op->basis->number_sites, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
        } else if (plan_setup_part(&tmp, &gps, 0, 1, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
        pl->gindex = gps.index;
        pl->d_gtable = gps.d_table;
    }
    if (pl->family == FAMILY_REPL_DIRECT) {
        /* rows that are a contiguous block [row0, row0 + count_local) of the global basis (what ReplicatedOperator
         * hands over: a slice of the global array) can take the staged kernel */
        int const contiguous = d_reps_local >= d_reps_global && d_reps_local + count_local <= d_reps_global + count_global;
        if (contiguous && pl->gindex.kind == LSK_INDEX_COMBINADIC && chain_eligible(pl)) {
            pl->chain_row0 = (int64_t)(d_reps_local - d_reps_global);
            if (setup_chain(pl, pl->gindex, count_local, d_reps_local, stream) != 0) { ls_amd_plan_destroy(pl); return -1; }
        }
        if (!pl->has_chain && build_tilemap(pl, count_local, 256) != 0) { ls_amd_plan_destroy(pl); return -1; }
    }
    if (pl->gindex.kind == LSK_INDEX_SEARCH && count_local > 0) {
        if (lsk_malloc(&p, 8 * (size_t)count_local) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        pl->d_row_gidx = (int64_t *)p;
        if (lsk_state_index(pl->gindex, count_local, d_reps_local, pl->d_row_gidx, stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    }
    if (pl->family == FAMILY_REPL_TILE) {
        if (lsk_malloc(&p, 8 * (size_t)(count_local > 0 ? count_local : 1)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        ps->d_norms = (double *)p;
        if (lsk_norms(pl->dbs, count_local, d_reps_local, ps->d_norms, stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        if (pl->dbs.k4_mode != 0) {
            if (lsk_malloc(&p, 8 * (size_t)(count_global > 0 ? count_global : 1)) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
            pl->d_norms_global = (double *)p;
            if (lsk_norms(pl->dbs, count_global, d_reps_global, pl->d_norms_global, stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
        }
    }
    if (lsk_sync(stream) != 0) { ls_amd_plan_destroy(pl); return dev_error(); }
    *out = pl;
    return 0;
}

int ls_amd_internal_basis_is_projected(ls_hs_basis const *b) { return BEXT(b)->order > 1; }
/* does the plan's K4 mode read x * norm(rep)?  (then the owners prescale their blocks in the indexed exchange) */
int ls_amd_internal_plan_prescales(ls_amd_plan const *pl) { return pl->dbs.k4_mode != 0; }
/* norm(rep) of the `count` representatives a rank owns, in its own order (freshly allocated; the caller lsk_free's it):
 * the rows are pulled out of the global array through the slot permutation, then K4's stabiliser sum */
int ls_amd_internal_owner_norms(ls_hs_operator const *op, ls_amd_gtab const *gt, int me, double **d_norms, void *stream) {
    lsk_basis dbs;
    *d_norms = NULL;
    if (basis_device(op->basis, &dbs) != 0) return -1;
    int64_t const cnt = gt->counts[me];
    void *reps = NULL, *nrm = NULL;
    if (lsk_malloc(&reps, 8 * (size_t)(cnt > 0 ? cnt : 1)) != 0) return dev_error();
    if (lsk_malloc(&nrm, 8 * (size_t)(cnt > 0 ? cnt : 1)) != 0) { lsk_free(reps); return dev_error(); }
    int const rc = lsk_scatter_owned(gt->n, gt->d_perm, (int64_t)me * gt->max_count, cnt, gt->reps, (uint64_t *)reps, stream) ||
                   lsk_norms(dbs, cnt, (uint64_t const *)reps, (double *)nrm, stream) || lsk_sync(stream);
    lsk_free(reps);
    if (rc) { lsk_free(nrm); return dev_error(); }
    *d_norms = (double *)nrm;
    return 0;
}

int ls_amd_matvec_replicated(ls_amd_plan *pl, void const *d_x_global, void *d_y_local, void *stream) {
    part_state *ps = &pl->parts[0];
    int slot;
    ls_amd_internal_count_matvec(pl);
    if (pl->family == FAMILY_REPL_DIRECT) {
        int const st = stage_begin(pl, ST_ROWS, stream);
        slot = timing_begin(pl, stream);
        if (pl->has_chain)
            DEV(lsk_chain(pl->dop, pl->dbs, pl->gindex, pl->cplx, pl->chain_wide, pl->d_chain_rec != NULL, pl->tilemap, ps->count,
                          pl->d_chain_rec ? pl->d_chain_rec : ps->d_reps, pl->chain_row0, pl->gindex.count, d_x_global, d_y_local,
                          pl->chain_cached, pl->d_chain_cache, pl->chain_v[0], pl->chain_v[1], stream));
        else
            DEV(lsk_direct_gx(pl->dop, pl->dbs, pl->gindex, pl->cplx, pl->tilemap, ps->d_reps, pl->d_row_gidx, d_x_global,
                          d_y_local, pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    if (pl->family == FAMILY_REPL_TILE && pl->idx_mode) {
        /* d_x_global = the owners' blocks in slot order, already multiplied by norm(rep) where the K4 mode prescales */
        lsk_pullidx ix;
        ix.tab = pl->gtab->tab;
        ix.perm = pl->gtab->d_perm;
        ix.row_g0 = pl->row_g0;
        ix.vtab = NULL;
        int const st = stage_begin(pl, ST_ROWS, stream);
        slot = timing_begin(pl, stream);
        DEV(lsk_tile_pull_idx(pl->dop, pl->dbs, pl->cplx, 0, ps->count, ps->d_reps, ps->d_norms, ix, pl->gindex.reps,
                              pl->gindex.count, d_x_global, pl->pull_halo, d_y_local, pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    if (pl->family == FAMILY_REPL_TILE) {
        void const *xg;
        if (prescaled_x(pl, pl->gindex.count, pl->gindex.reps, d_x_global, pl->d_norms_global, &xg, stream) != 0) return -1;
        int const st = stage_begin(pl, ST_ROWS, stream);
        slot = timing_begin(pl, stream);
        DEV(lsk_tile_pull(pl->dop, pl->dbs, pl->gindex, pl->cplx, 0, ps->count, ps->d_reps, ps->d_norms,
                          pl->d_norms_global, pl->d_row_gidx, xg, pl->htab_bits, pl->gindex.reps, pl->gindex.count,
                          pl->d_xs ? pl->d_xs : d_x_global, pl->pull_halo, d_x_global, d_y_local, pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    return set_error("ls_amd_matvec_replicated: not a replicated-x plan");
}
/* Plan time (dist.c, replicated-x exchange of an unprojected basis): the blocks of 2^shift rows of the global vector that this
 * plan's rows read -- every off-diagonal partner, plus the rows themselves with `halo` rows either side (the LDS windows of the
 * staged kernels; row0 = global index of the plan's first row).  h_bitmap[nwords] is overwritten.  Returns 0, 1 when the question does not apply (projected bases read x
 * through the index table; inversion sectors canonicalise the partner first: they keep the whole vector), -1 on a device error. */
int ls_amd_internal_plan_reach(ls_amd_plan *pl, int shift, int64_t row0, int64_t halo, int64_t nwords, uint32_t *h_bitmap, void *stream) {
    if (pl->family != FAMILY_REPL_DIRECT || pl->dbs.proj != LSK_PROJ_NONE) return 1;
    part_state *ps = &pl->parts[0];
    int64_t const n_global = pl->gindex.count;
    if (((n_global - 1) >> shift) / 32 >= nwords) return set_error("internal error: reach bitmap too small");
    memset(h_bitmap, 0, 4 * (size_t)nwords);
    if (ps->count <= 0) return 0;
    void *d = NULL;
    DEV(lsk_malloc(&d, 4 * (size_t)nwords));
    if (lsk_memset_async(d, 0, 4 * (size_t)nwords, stream) != 0 ||
        lsk_reach_blocks(pl->dop, pl->gindex, ps->count, ps->d_reps, shift, (uint32_t *)d, stream) != 0 || lsk_sync(stream) != 0 ||
        lsk_d2h(h_bitmap, d, 4 * (size_t)nwords) != 0) { lsk_free(d); return dev_error(); }
    lsk_free(d);
    /* own rows [row0, row0 + count) +- halo */
    int64_t g0 = row0, g1 = g0 + ps->count;
    g0 = g0 > halo ? g0 - halo : 0;
    g1 = g1 + halo < n_global ? g1 + halo : n_global;
    for (int64_t b = g0 >> shift; b <= (g1 - 1) >> shift; ++b) h_bitmap[b >> 5] |= 1u << (b & 31);
    return 0;
}

/* The indexed replicated-x matvec in two steps (dist.c): BEGIN resolves the packets of the first split_rows rows -- stage A,
 * K4 and every slot look-up, nothing of x -- and is enqueued while the blocks of x travel; FINISH gathers them from the
 * received x and runs the fused kernel on whatever rows the packet buffer did not cover. */
int ls_amd_internal_repl_split_begin(ls_amd_plan *pl, void *stream) {
    if (pl->family != FAMILY_REPL_TILE || !pl->idx_mode || pl->split_rows <= 0) return 0;
    if (pl->slot_cache && pl->slot_cache_valid) return 0; /* the streams of an earlier matvec are still good */
    part_state *ps = &pl->parts[0];
    lsk_pullidx ix;
    ix.tab = pl->gtab->tab;
    ix.perm = pl->gtab->d_perm;
    ix.row_g0 = pl->row_g0;
    ix.vtab = NULL;
    int const st = stage_begin(pl, ST_GENERATE, stream);
    int const slot = timing_begin(pl, stream);
    DEV(lsk_tile_pull_resolve(pl->dop, pl->dbs, 0, split_rows_now(pl), ps->d_reps, ps->d_norms, ix, pl->gindex.reps, pl->gindex.count,
                              pl->pull_halo, pl->pbuf, pl->d_err, stream));
    pl->slot_cache_valid = pl->slot_cache; /* only once the resolve launch went through: a failed one leaves nothing to reuse */
    timing_end(pl, slot, stream);
    stage_end(pl, st, stream);
    return 0;
}
/* rows [row0, row1) of the split matvec (row0 a multiple of 256): what was resolved ahead is gathered, the rest takes the fused
 * kernel.  The chunked return of the replicated-x driver calls it once per chunk of rows; count = 1: this call opens the matvec. */
int ls_amd_internal_repl_split_rows(ls_amd_plan *pl, void const *d_x_global, void *d_y_local, int64_t row0, int64_t row1, int count, void *stream) {
    if (pl->family != FAMILY_REPL_TILE || !pl->idx_mode || pl->split_rows <= 0) return set_error("internal error: not a split plan");
    part_state *ps = &pl->parts[0];
    if (row1 > ps->count) row1 = ps->count;
    if (row0 < 0 || (row0 & 255) != 0 || row0 > row1) return set_error("internal error: bad row range of the split matvec");
    if (count) ls_amd_internal_count_matvec(pl);
    lsk_pullidx ix;
    ix.tab = pl->gtab->tab;
    ix.perm = pl->gtab->d_perm;
    ix.row_g0 = pl->row_g0;
    ix.vtab = NULL;
    int64_t const S = split_rows_now(pl);
    int const st = stage_begin(pl, ST_ROWS, stream);
    if (row0 < S) {
        int const slot = pl->slot_cache ? timing_begin(pl, stream) : -1; /* cached: the gather kernel is the dominant (only) one */
        DEV(lsk_tile_pull_gather(pl->dop, pl->dbs, pl->cplx, row0, row1 < S ? row1 : S, ps->d_reps, ps->d_norms, ix, d_x_global, pl->pbuf, d_y_local,
                                 stream));
        if (slot >= 0) timing_end(pl, slot, stream);
    }
    if (row1 > S)
        DEV(lsk_tile_pull_idx(pl->dop, pl->dbs, pl->cplx, row0 > S ? row0 : S, row1, ps->d_reps, ps->d_norms, ix, pl->gindex.reps,
                              pl->gindex.count, d_x_global, pl->pull_halo, d_y_local, pl->d_err, stream));
    stage_end(pl, st, stream);
    return 0;
}
int ls_amd_internal_repl_split_finish(ls_amd_plan *pl, void const *d_x_global, void *d_y_local, void *stream) {
    if (pl->family != FAMILY_REPL_TILE || !pl->idx_mode || pl->split_rows <= 0) return ls_amd_matvec_replicated(pl, d_x_global, d_y_local, stream);
    return ls_amd_internal_repl_split_rows(pl, d_x_global, d_y_local, 0, pl->parts[0].count, 1, stream);
}
char const *ls_amd_plan_kernel_name(ls_amd_plan const *pl) {
    switch (pl->family) {
    case FAMILY_DIRECT_PUSH: return pl->has_push_staged ? "direct-push+staged" : "direct-push";
    case FAMILY_DIRECT_PULL:
        return pl->has_chain ? "direct-pull+staged" : pl->has_pairs ? (pl->pairs.sites ? "direct-pull+pairsites" : pl->pairs.rows ? "direct-pull+pairrows" : "direct-pull+pairs") : "direct-pull";
    case FAMILY_TILE_PULL: return pl->idx_mode ? (pl->slot_cache ? "tile-pull+indexed+cached" : (pl->d_vtab ? "tile-pull+values" : "tile-pull+indexed")) : "tile-pull";
    case FAMILY_REPL_DIRECT:
        return pl->has_chain ? "replicated-direct-pull+staged" : "replicated-direct-pull";
    case FAMILY_REPL_TILE: return pl->idx_mode ? (pl->slot_cache ? "replicated-tile-pull+indexed+cached" : "replicated-tile-pull+indexed") : "replicated-tile-pull";
    default: return pl->streams ? "tile+streams" : (pl->has_push_staged ? "tile(one rank: push+staged)" : "tile");
    }
}
/* nominal bytes per packet (a segment of c packets takes ls_amd_plan_segment_bytes(c): pre-indexed keys are padded to 8 bytes) */
int ls_amd_plan_packet_bytes(ls_amd_plan const *pl) { return pl->key_bytes + (pl->cplx ? 16 : 8); }
/* bytes of per-row plan / basis data the plan's dominant kernel streams from HBM next to x and y (the roofline's
 * compulsory traffic is rows * (this + 2 w)): the staged row kernel reads the fused 8-byte record, or the low state word(s)
 * plus one cached partner rank per cached pair; the generic row kernels read the 8-byte state; the projected pull kernel
 * the state and norm(alpha) */
int ls_amd_plan_row_bytes(ls_amd_plan const *pl) {
    if (pl->has_pairs) return pl->pairs.wide ? 8 : 4; /* the plan's 4-byte copy of the states, or (33..64 sites) the 8-byte representatives */
    if (pl->has_chain) {
        if (pl->d_chain_rec) return 8;
        int const narrow = pl->op->basis->number_sites <= 32 && !pl->chain_wide;
        return (narrow ? 4 : 8) + pl->chain_cached * (pl->chain_wide ? 8 : 4);
    }
    if (pl->family == FAMILY_TILE_PULL || pl->family == FAMILY_REPL_TILE) return 16;
    return 8;
}
int64_t ls_amd_plan_nnz(ls_amd_plan const *pl) { return pl->nnz; }
int ls_amd_plan_send_counts(ls_amd_plan const *pl, int round, int64_t *counts) {
    if (pl->family != FAMILY_TILE) { for (int d = 0; d < pl->P; ++d) counts[d] = 0; return 0; }
    part_state const *ps = &pl->parts[0];
    if (round < 0 || round >= ps->rounds) return set_error("round out of range");
    memcpy(counts, ps->send_counts + (size_t)round * pl->P, sizeof(int64_t) * pl->P);
    return 0;
}

/* hash table over `n` (global) representatives: built on first use, values refreshed per matvec */
static int prescaled_x(ls_amd_plan *pl, int64_t n, uint64_t const *d_reps, void const *d_x, double const *d_norms,
                       void const **out, void *stream) {
    if (!pl->d_htab) {
        int bits = 4;
        while (((int64_t)1 << bits) < 2 * n) ++bits; /* load factor <= 0.5 */
        if (bits > 32) return set_error("hash index: more than 2^31 representatives per table are not supported");
        pl->htab_bits = bits;
        DEV(lsk_malloc(&pl->d_htab, (size_t)(pl->cplx ? 32 : 16) << bits));
        void *p;
        DEV(lsk_malloc(&p, 4 * (size_t)(n > 0 ? n : 1)));
        pl->d_slot_of = (uint32_t *)p;
        DEV(lsk_hash_build(pl->cplx, n, d_reps, bits, pl->d_htab, pl->d_slot_of, stream));
        /* near window (LS_AMD_PULL_HALO entries either side of a tile, default 512, 0 = every partner through the table) */
        pl->pull_halo = pull_halo_setting(0);
        if (pl->pull_halo > 0 && pl->dbs.k4_mode != 0) DEV(lsk_malloc(&pl->d_xs, (size_t)(pl->cplx ? 16 : 8) * (size_t)(n > 0 ? n : 1)));
    }
    int const st = stage_begin(pl, ST_REFRESH, stream);
    DEV(lsk_hash_fill(pl->cplx, n, pl->d_slot_of, d_x, pl->dbs.k4_mode != 0 ? d_norms : NULL, pl->d_htab, pl->d_xs, stream));
    stage_end(pl, st, stream);
    *out = pl->d_htab;
    return 0;
}

int ls_amd_diag(ls_amd_plan *pl, void const *d_x, void *d_y, void *stream) {
    part_state *ps = &pl->parts[0];
    int const st = stage_begin(pl, ST_DIAG, stream);
    if (pl->has_push_staged) { /* the staged push kernel adds the diagonal part itself: y is cleared (no diagonal terms: accumulated into) */
        if (pl->dop.n_diag > 0) DEV(lsk_memset_async(d_y, 0, (size_t)ps->count * (pl->cplx ? 16 : 8), stream));
    } else
    DEV(lsk_diag(pl->dop, pl->cplx, ps->count, ps->d_reps, d_x, d_y, stream));
    stage_end(pl, st, stream);
    return 0;
}

static int generate_round(ls_amd_plan *pl, part_state *ps, int pid, int round, void const *d_x, void *d_y,
                          void *d_send, void *stream) {
    int64_t row0 = ps->count * round / ps->rounds, row1 = ps->count * (round + 1) / ps->rounds;
    if (pl->P > 1 && !ps->d_wtab) DEV(lsk_memset_async(pl->d_cursors, 0, 8 * (size_t)pl->P, stream));
    int const st = stage_begin(pl, ST_GENERATE, stream);
    int slot = timing_begin(pl, stream);
    if (pl->has_push_staged) { /* (FAMILY_TILE, one rank, one round: every packet is this partition's own) */
        lsk_index cix = ps->index;
        cix.kind = LSK_INDEX_COMBINADIC;
        DEV(lsk_push_staged(pl->dop, pl->dbs, cix, pl->cplx, pl->tilemap, ps->count, ps->d_reps, d_x, d_y, pl->d_err, stream));
    } else if (pl->streams) {
        uint64_t const *d_binom;
        if (device_binom(&d_binom) != 0) return -1;
        DEV(lsk_tile_st(pl->dop, pl->gd, d_binom, pl->cplx, 0, pl->P, pl->st_S, pl->st_tile_rows, row0, row1, ps->d_reps, d_x,
                        ps->d_ttab + (size_t)ps->ttab_first[round] * (size_t)(pl->P * pl->st_S), ps->d_layouts + round, d_send, pl->d_err, stream));
    } else if (ps->d_wtab)
        DEV(lsk_tile_wv(pl->dop, pl->dbs, ps->index, pl->gd, pl->cplx, 0, pl->P, pid, row0, row1, ps->d_reps, ps->d_norms, d_x, d_y,
                        ps->d_wtab + (size_t)ps->wtab_first[round] * pl->P, ps->d_layouts + round, d_send, pl->d_err,
                        ps->scatter_gt ? ps->scatter_gt->tab : no_gtab(), stream));
    else
        DEV(lsk_tile(pl->dop, pl->dbs, ps->index, pl->cplx, 0, pl->P, pid, row0, row1, ps->d_reps, ps->d_norms, d_x,
                     d_y, pl->d_cursors, ps->d_layouts + round, d_send, pl->d_counts, pl->d_err, stream));
    timing_end(pl, slot, stream);
    stage_end(pl, st, stream);
    return 0;
}

int ls_amd_generate(ls_amd_plan *pl, int round, void const *d_x, void *d_y, void *d_send, void *stream) {
    if (pl->family != FAMILY_TILE) return set_error("ls_amd_generate: the plan uses a direct kernel (P == 1)");
    if (pl->me < 0) return set_error("ls_amd_generate: plan owns all partitions; use ls_amd_matvec");
    part_state *ps = &pl->parts[0];
    if (round < 0 || round >= ps->rounds) return set_error("round out of range");
    return generate_round(pl, ps, pl->me, round, d_x, d_y, d_send, stream);
}

int ls_amd_scatter(ls_amd_plan *pl, int64_t n, uint64_t const *d_betas, void const *d_values, void *d_y,
                   void *stream) {
    part_state *ps = &pl->parts[0];
    if (pl->streams) return set_error("ls_amd_scatter: the plan writes sorted streams (consumed by windows, dist.c)");
    if (ps->index.kind == LSK_INDEX_COMBINADIC) return set_error("ls_amd_scatter: plan has no search index");
    int const st = stage_begin(pl, ST_SCATTER, stream);
    if (pl->key_bytes == 4) { /* pre-indexed packets: d_betas is the segment's u32 index array */
        lsk_segs sg;
        memset(&sg, 0, sizeof(sg));
        sg.n = 1; sg.start[0] = 0; sg.start[1] = n; sg.key_off[0] = 0;
        sg.val_off[0] = (int64_t)((char const *)d_values - (char const *)d_betas);
        sg.y[0] = d_y;
        DEV(lsk_scatter_idx(pl->cplx, &sg, d_betas, stream));
    } else
        DEV(lsk_scatter(ps->index, pl->cplx, n, d_betas, d_values, d_y, pl->dbs.k4_mode ? ps->d_norms : NULL, pl->d_err,
                        stream));
    stage_end(pl, st, stream);
    return 0;
}
/* All segments of one round's receive buffer in ONE launch (consumer side of a rank): segment s holds counts[s] packets at
 * d_recv + offsets[s], laid out as ls_amd_plan_segment_bytes describes.  (The per-segment form above cost 56 launches per
 * matvec on chain_28 x 8.) */
int ls_amd_scatter_round(ls_amd_plan *pl, int num_segments, int64_t const *counts, int64_t const *offsets, void const *d_recv,
                         void *d_y, void *stream) {
    part_state *ps = &pl->parts[0];
    if (pl->streams) return set_error("ls_amd_scatter_round: the plan writes sorted streams (consumed by windows, dist.c)");
    if (ps->index.kind == LSK_INDEX_COMBINADIC) return set_error("ls_amd_scatter_round: plan has no search index");
    int const st = stage_begin(pl, ST_SCATTER, stream);
    int s = 0;
    while (s < num_segments) {
        lsk_segs sg;
        memset(&sg, 0, sizeof(sg));
        int64_t total = 0;
        for (; s < num_segments && sg.n < LSK_MAX_SEGS; ++s) {
            if (counts[s] <= 0) continue;
            sg.start[sg.n] = total;
            sg.key_off[sg.n] = offsets[s];
            sg.val_off[sg.n] = offsets[s] + segment_key_bytes(pl, counts[s]);
            sg.y[sg.n] = d_y;
            total += counts[s];
            ++sg.n;
        }
        if (sg.n == 0) break;
        sg.start[sg.n] = total;
        if (pl->key_bytes == 4) DEV(lsk_scatter_idx(pl->cplx, &sg, d_recv, stream));
        else {
            DEV(lsk_scatter_segs(ps->index, ps->scatter_gt ? ps->scatter_gt->tab : no_gtab(), pl->cplx, &sg, d_recv, pl->dbs.k4_mode ? ps->d_norms : NULL, pl->d_err, stream));
        }
    }
    stage_end(pl, st, stream);
    return 0;
}
/* sorted streams of a plan that owns one partition (dist.c): S streams per segment; offsets[rounds][P][S + 1] = where the streams
 * of the segment for destination d start in round r (packets); the consumer of one round over n_src source segments */
int ls_amd_internal_plan_streams(ls_amd_plan const *pl) { return pl->streams ? pl->st_S : 0; }
uint32_t const *ls_amd_internal_plan_stream_offsets(ls_amd_plan const *pl) { return pl->streams ? pl->parts[0].h_soff : NULL; }
int ls_amd_internal_window_round(ls_amd_plan *pl, lsk_wsrc const *d_srcs, int n_src, void *d_y, void *stream) {
    if (!pl->streams || pl->me < 0) return set_error("ls_amd_internal_window_round: not a streams plan of one partition");
    lsk_wdests wd;
    memset(&wd, 0, sizeof(wd));
    int const W = lsk_window_rows(pl->cplx);
    int64_t const windows = (pl->parts[0].count + W - 1) / W;
    wd.n = 1;
    wd.count[0] = pl->parts[0].count;
    wd.y[0] = d_y;
    wd.first_block[1] = (windows + pl->st_wpb - 1) / pl->st_wpb;
    if (pl->stream_gkeys) {
        uint64_t const *d_binom;
        if (device_binom(&d_binom) != 0) return -1;
        wd.dir[0] = pl->parts[0].index.dir; wd.reps[0] = pl->parts[0].d_reps;
        wd.binom = d_binom; wd.n_ranks = pl->gd.n_ranks; wd.weight = pl->gd.weight; wd.err = pl->d_err;
    }
    int const st = stage_begin(pl, ST_SCATTER, stream);
    DEV(lsk_window(pl->cplx, &wd, d_srcs, n_src, pl->st_S, pl->st_wpb, stream));
    stage_end(pl, st, stream);
    return 0;
}
int ls_amd_plan_key_bytes(ls_amd_plan const *pl) { return pl->key_bytes; }
int64_t ls_amd_plan_segment_bytes(ls_amd_plan const *pl, int64_t count) { return segment_key_bytes(pl, count) + (pl->cplx ? 16 : 8) * count; }
int64_t ls_amd_plan_segment_value_offset(ls_amd_plan const *pl, int64_t count) { return segment_key_bytes(pl, count); }
int64_t ls_amd_plan_packet_index_bytes(ls_amd_plan const *pl) {
    return pl->d_gdir ? (int64_t)sizeof(lsk_rankdir) * ((pl->gd.n_ranks + 63) / 64) * pl->gd.P : 0;
}

int ls_amd_matvec(ls_amd_plan *pl, void const *const *d_x, void *const *d_y, void *stream) {
    if (pl->me >= 0) return set_error("ls_amd_matvec: plan owns one partition; drive it with generate/scatter");
    for (int p = 0; p < pl->n_local; ++p) if (ls_amd_internal_check_y(pl, d_y[p]) != 0) return -1;
    ls_amd_internal_count_matvec(pl);
    if (pl->family == FAMILY_TILE_PULL && pl->idx_mode) {
        /* indexed mode on one device: slot = index; the only per-matvec preparation is x * norm(rep), a streaming pass
         * (the value table of the other mode costs N random 16-byte writes) */
        part_state *ps = &pl->parts[0];
        void const *xs = d_x[0];
        if (pl->dbs.k4_mode != 0) {
            int const sr = stage_begin(pl, ST_REFRESH, stream);
            DEV(lsk_scale(pl->cplx, ps->count, d_x[0], ps->d_norms, pl->d_xs, stream));
            stage_end(pl, sr, stream);
            xs = pl->d_xs;
        }
        lsk_pullidx ix;
        ix.tab = pl->gtab->tab;
        ix.perm = NULL;
        ix.row_g0 = 0;
        ix.vtab = NULL;
        if (pl->d_vtab && (pl->slot_cache || pl->split_rows > 0)) { lsk_free(pl->d_vtab); pl->d_vtab = NULL; } /* a cache enabled later takes over */
        if (pl->split_rows > 0 && pl->slot_cache) {
            /* slot cache: the streams of rows [0, split_rows) are resolved by the first matvec and kept; later matvecs gather.
             * Rows the cache has no room for take the fused kernel */
            lsk_pullbuf const pb = pl->pbuf;
            int st, slot;
            if (!pl->slot_cache_valid) {
                st = stage_begin(pl, ST_GENERATE, stream);
                slot = timing_begin(pl, stream);
                DEV(lsk_tile_pull_resolve(pl->dop, pl->dbs, 0, pl->split_rows, ps->d_reps, ps->d_norms, ix, ps->d_reps, ps->count, pl->pull_halo, pb,
                                          pl->d_err, stream));
                timing_end(pl, slot, stream);
                stage_end(pl, st, stream);
                pl->slot_cache_valid = 1;
            }
            st = stage_begin(pl, ST_ROWS, stream);
            slot = timing_begin(pl, stream); /* the gather kernel is the dominant one of a cached plan */
            DEV(lsk_tile_pull_gather(pl->dop, pl->dbs, pl->cplx, 0, pl->split_rows, ps->d_reps, ps->d_norms, ix, xs, pb, d_y[0], stream));
            timing_end(pl, slot, stream);
            if (pl->split_rows < ps->count) {
                slot = timing_begin(pl, stream);
                DEV(lsk_tile_pull_idx(pl->dop, pl->dbs, pl->cplx, pl->split_rows, ps->count, ps->d_reps, ps->d_norms, ix, ps->d_reps, ps->count, xs,
                                      pl->pull_halo, d_y[0], pl->d_err, stream));
                timing_end(pl, slot, stream);
            }
            stage_end(pl, st, stream);
            return 0;
        }
        if (pl->split_rows > 0) {
            /* the split form on one device (LS_AMD_PULL_SPLIT: measurement of what the replicated-x exchange runs, dist.c):
             * rounds of resolve | gather over the rows the packet buffer holds */
            for (int64_t r0 = 0; r0 < ps->count; r0 += pl->split_rows) {
                int64_t const r1 = r0 + pl->split_rows < ps->count ? r0 + pl->split_rows : ps->count;
                lsk_pullbuf pb = pl->pbuf;
                pb.row0 = r0;
                int st = stage_begin(pl, ST_GENERATE, stream);
                int const slot = timing_begin(pl, stream);
                DEV(lsk_tile_pull_resolve(pl->dop, pl->dbs, r0, r1, ps->d_reps, ps->d_norms, ix, ps->d_reps, ps->count, pl->pull_halo, pb,
                                          pl->d_err, stream));
                timing_end(pl, slot, stream);
                stage_end(pl, st, stream);
                st = stage_begin(pl, ST_ROWS, stream);
                DEV(lsk_tile_pull_gather(pl->dop, pl->dbs, pl->cplx, r0, r1, ps->d_reps, ps->d_norms, ix, xs, pb, d_y[0], stream));
                stage_end(pl, st, stream);
            }
            return 0;
        }
        if (pl->d_vtab) { /* the values of this matvec's x (x n(rep) in the prescaling K4 modes) into their buckets: table order */
            int const sr = stage_begin(pl, ST_REFRESH, stream);
            DEV(lsk_vtab_refresh(pl->gtab->tab, pl->d_vtab, xs, stream));
            stage_end(pl, sr, stream);
            ix.vtab = pl->d_vtab;
        }
        int const st = stage_begin(pl, ST_ROWS, stream);
        int slot = timing_begin(pl, stream);
        DEV(lsk_tile_pull_idx(pl->dop, pl->dbs, pl->cplx, 0, ps->count, ps->d_reps, ps->d_norms, ix, ps->d_reps, ps->count, xs,
                              pl->pull_halo, d_y[0], pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    if (pl->family == FAMILY_TILE_PULL) {
        part_state *ps = &pl->parts[0];
        void const *xg;
        if (prescaled_x(pl, ps->count, ps->d_reps, d_x[0], ps->d_norms, &xg, stream) != 0) return -1;
        int const st = stage_begin(pl, ST_ROWS, stream);
        int slot = timing_begin(pl, stream);
        DEV(lsk_tile_pull(pl->dop, pl->dbs, ps->index, pl->cplx, 0, ps->count, ps->d_reps, ps->d_norms, ps->d_norms,
                          NULL, xg, pl->htab_bits, ps->d_reps, ps->count, pl->d_xs ? pl->d_xs : d_x[0], pl->pull_halo, d_x[0],
                          d_y[0], pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    /* localDiagonal on every partition first: y is assigned (DMV:1062-1063) */
    if (pl->has_push_staged) { /* the staged push kernel adds the diagonal part itself: y is cleared (n_diag == 0: accumulated into, DMV:1062-1063) */
        if (pl->dop.n_diag > 0) {
            int const st = stage_begin(pl, ST_DIAG, stream);
            DEV(lsk_memset_async(d_y[0], 0, (size_t)pl->parts[0].count * (pl->cplx ? 16 : 8), stream));
            stage_end(pl, st, stream);
        }
    } else if (pl->family != FAMILY_DIRECT_PULL) {
        int const st = stage_begin(pl, ST_DIAG, stream);
        for (int p = 0; p < pl->n_local; ++p)
            DEV(lsk_diag(pl->dop, pl->cplx, pl->parts[p].count, pl->parts[p].d_reps, d_x[p], d_y[p], stream));
        stage_end(pl, st, stream);
    }
    if (pl->dop.n_groups == 0 && pl->family != FAMILY_DIRECT_PULL) return 0;
    if (pl->family != FAMILY_TILE) {
        part_state *ps = &pl->parts[0];
        int const st = stage_begin(pl, ST_ROWS, stream);
        int slot = timing_begin(pl, stream);
        if (pl->has_chain)
            DEV(lsk_chain(pl->dop, pl->dbs, ps->index, pl->cplx, pl->chain_wide, pl->d_chain_rec != NULL, pl->tilemap, ps->count,
                          pl->d_chain_rec ? pl->d_chain_rec : ps->d_reps, 0, ps->count, d_x[0], d_y[0], pl->chain_cached,
                          pl->d_chain_cache, pl->chain_v[0], pl->chain_v[1], stream));
        else if (pl->has_pairs)
            DEV(lsk_pairs(pl->pairs, pl->dbs.hamming_weight, pl->cplx, pl->tilemap, ps->count, d_x[0], d_y[0], stream));
        else if (pl->has_push_staged)
            DEV(lsk_push_staged(pl->dop, pl->dbs, ps->index, pl->cplx, pl->tilemap, ps->count, ps->d_reps, d_x[0], d_y[0], pl->d_err, stream));
        else
            DEV(lsk_direct(pl->dop, pl->dbs, ps->index, pl->cplx, pl->family == FAMILY_DIRECT_PULL ? (OEXT(pl->op)->is_hermitian ? 1 : 2) : 0, pl->tilemap, ps->d_reps,
                           d_x[0], d_y[0], pl->d_err, stream));
        timing_end(pl, slot, stream);
        stage_end(pl, st, stream);
        return 0;
    }
    int const P = pl->P;
    if (pl->streams) {
        /* sorted streams: every source writes its round into its own buffer, then ONE launch adds windows of every y in LDS */
        uint64_t const *d_binom;
        if (device_binom(&d_binom) != 0) return -1;
        int const W = lsk_window_rows(pl->cplx);
        lsk_wdests wd;
        memset(&wd, 0, sizeof(wd));
        wd.n = P;
        for (int d = 0; d < P; ++d) {
            int64_t const windows = (pl->parts[d].count + W - 1) / W;
            wd.count[d] = pl->parts[d].count;
            wd.y[d] = d_y[d];
            wd.first_block[d + 1] = wd.first_block[d] + (windows + pl->st_wpb - 1) / pl->st_wpb;
            if (pl->stream_gkeys) { wd.dir[d] = pl->parts[d].index.dir; wd.reps[d] = pl->parts[d].d_reps; }
        }
        if (pl->stream_gkeys) { wd.binom = d_binom; wd.n_ranks = pl->gd.n_ranks; wd.weight = pl->gd.weight; wd.err = pl->d_err; }
        for (int r = 0; r < pl->st_rounds; ++r) {
            int st = stage_begin(pl, ST_GENERATE, stream);
            for (int p = 0; p < P; ++p) {
                part_state *ps = &pl->parts[p];
                int64_t row0 = ps->count * r / ps->rounds, row1 = ps->count * (r + 1) / ps->rounds;
                int const slot = timing_begin(pl, stream);
                DEV(lsk_tile_st(pl->dop, pl->gd, d_binom, pl->cplx, 0, P, pl->st_S, pl->st_tile_rows, row0, row1, ps->d_reps, d_x[p],
                                ps->d_ttab + (size_t)ps->ttab_first[r] * (size_t)(P * pl->st_S), ps->d_layouts + r, pl->d_send_parts[p],
                                pl->d_err, stream));
                timing_end(pl, slot, stream);
            }
            stage_end(pl, st, stream);
            st = stage_begin(pl, ST_SCATTER, stream);
            DEV(lsk_window(pl->cplx, &wd, pl->d_wsrcs + (size_t)r * P * P, P, pl->st_S, pl->st_wpb, stream));
            stage_end(pl, st, stream);
        }
        return 0;
    }
    for (int p = 0; p < P; ++p) {
        part_state *ps = &pl->parts[p];
        for (int r = 0; r < ps->rounds; ++r) {
            if (generate_round(pl, ps, p, r, d_x[p], d_y[p], pl->d_send, stream) != 0) return -1;
            /* the "exchange": every destination consumes its segment straight from the send buffer */
            if (pl->key_bytes == 4) { /* pre-indexed packets: all destinations in one launch, each segment with its own y */
                for (int d0 = 0; d0 < P; d0 += LSK_MAX_SEGS) {
                    lsk_segs sg;
                    memset(&sg, 0, sizeof(sg));
                    int64_t total = 0;
                    for (int d = d0; d < P && d < d0 + LSK_MAX_SEGS; ++d) {
                        int64_t const c = ps->send_counts[(size_t)r * P + d];
                        if (d == p || c == 0) continue;
                        sg.start[sg.n] = total;
                        sg.key_off[sg.n] = ps->h_beta_off[(size_t)r * P + d];
                        sg.val_off[sg.n] = ps->h_val_off[(size_t)r * P + d];
                        sg.y[sg.n] = d_y[d];
                        total += c;
                        ++sg.n;
                    }
                    if (sg.n == 0) continue;
                    sg.start[sg.n] = total;
                    int const st = stage_begin(pl, ST_SCATTER, stream);
                    DEV(lsk_scatter_idx(pl->cplx, &sg, pl->d_send, stream));
                    stage_end(pl, st, stream);
                }
                continue;
            }
            if (pl->d_part_ctx) { /* state-carrying packets: one launch too, index and norms of a segment's destination from the context array */
                /* (a partition whose rank directory is missing while others have one keeps the per-segment form below) */
                int mixed = 0;
                for (int d = 0; d < P; ++d) if ((pl->parts[d].index.dir != NULL) != (pl->parts[pl->part_ctx_any].index.dir != NULL) && pl->parts[d].count > 0) mixed = 1;
                if (!mixed) {
                    for (int d0 = 0; d0 < P; d0 += LSK_MAX_SEGS) {
                        lsk_segs sg;
                        memset(&sg, 0, sizeof(sg));
                        int64_t total = 0;
                        for (int d = d0; d < P && d < d0 + LSK_MAX_SEGS; ++d) {
                            int64_t const c = ps->send_counts[(size_t)r * P + d];
                            if (d == p || c == 0) continue;
                            sg.start[sg.n] = total;
                            sg.key_off[sg.n] = ps->h_beta_off[(size_t)r * P + d];
                            sg.val_off[sg.n] = ps->h_val_off[(size_t)r * P + d];
                            sg.y[sg.n] = d_y[d];
                            sg.part[sg.n] = (uint8_t)d;
                            total += c;
                            ++sg.n;
                        }
                        if (sg.n == 0) continue;
                        sg.start[sg.n] = total;
                        int const st = stage_begin(pl, ST_SCATTER, stream);
                        DEV(lsk_scatter_parts(pl->d_part_ctx, pl->parts[pl->part_ctx_any].index, pl->cplx, &sg, pl->d_send, pl->d_err, stream));
                        stage_end(pl, st, stream);
                    }
                    continue;
                }
            }
            for (int d = 0; d < P; ++d) {
                int64_t c = ps->send_counts[(size_t)r * P + d];
                if (d == p || c == 0) continue;
                uint64_t const *betas = (uint64_t const *)((char *)pl->d_send + ps->h_beta_off[(size_t)r * P + d]);
                void const *vals = (char *)pl->d_send + ps->h_val_off[(size_t)r * P + d];
                int const st = stage_begin(pl, ST_SCATTER, stream);
                DEV(lsk_scatter(pl->parts[d].index, pl->cplx, c, betas, vals, d_y[d],
                                pl->dbs.k4_mode ? pl->parts[d].d_norms : NULL, pl->d_err, stream));
                stage_end(pl, st, stream);
            }
        }
    }
    return 0;
}

int ls_amd_plan_check(ls_amd_plan *pl, void *stream) {
    int flag = 0, zero = 0;
    DEV(lsk_sync(stream));
    DEV(lsk_d2h(&flag, pl->d_err, sizeof(int)));
    if (flag || pl->leaves_basis) {
        DEV(lsk_h2d(pl->d_err, &zero, sizeof(int)));
        return set_error("invalid index: the operator generated a state outside the basis "
                         "(it does not respect the basis symmetries)"); /* DMV:115-118 */
    }
    return 0;
}

/* ============================================================================================ */
/* enumeration and layout converters                                                            */
/* ============================================================================================ */
static int64_t candidate_count(ls_hs_basis const *b) {
    int const Leff = b->number_sites - (b->spin_inversion != 0 ? 1 : 0);
    int const h = BEXT(b)->hamming_weight;
    if (h >= 0) return (int64_t)binom(Leff, h);
    if (Leff >= 62) return -1;
    return (int64_t)1 << Leff;
}

int ls_amd_enumerate_states(ls_hs_basis const *basis, int num_locales, uint64_t **d_states, uint8_t **d_masks,
                            int64_t *count, void *stream) {
    lsk_basis dbs;
    uint64_t const *d_binom;
    if (basis_device(basis, &dbs) != 0 || device_binom(&d_binom) != 0) return -1;
    int64_t ncand = candidate_count(basis);
    if (ncand < 0) return set_error("basis too large to enumerate");
    DEV(lsk_enumerate(dbs, d_binom, ncand, d_states, count, stream));
    if (d_masks) {
        void *p;
        DEV(lsk_malloc(&p, (size_t)(*count > 0 ? *count : 1)));
        *d_masks = (uint8_t *)p;
        DEV(lsk_masks(*count, *d_states, num_locales < 1 ? 1 : num_locales, *d_masks, stream));
        DEV(lsk_sync(stream));
    }
    return 0;
}
int ls_amd_gather(int64_t n, void const *d_perm, int perm_is_64, int elt_size, void const *d_src, void *d_out, void *stream) {
    DEV(lsk_gather_perm(n, d_perm, perm_is_64, elt_size, d_src, d_out, stream));
    return 0;
}
int ls_amd_mask_counts(int64_t n, uint8_t const *d_masks, int num_locales, int64_t *counts, void *stream) {
    DEV(lsk_mask_counts(n, d_masks, num_locales, counts, stream));
    return 0;
}
int ls_amd_block_to_hashed(int64_t n, uint8_t const *d_masks, int num_locales, int elt_size, void const *d_src,
                           void *const *d_dest, void *stream) {
    DEV(lsk_block_to_hashed(n, d_masks, num_locales, elt_size, d_src, d_dest, stream));
    return 0;
}
int ls_amd_hashed_to_block(int64_t n, uint8_t const *d_masks, int num_locales, int elt_size,
                           void const *const *d_src, void *d_dest, void *stream) {
    DEV(lsk_hashed_to_block(n, d_masks, num_locales, elt_size, d_src, d_dest, stream));
    return 0;
}

/* ============================================================================================ */
/* ls_chpl_* exports (host pointers; stage through HBM)                                          */
/* ============================================================================================ */
void ls_chpl_enumerate_representatives(ls_hs_basis *basisPtr, uint64_t lower, uint64_t upper,
                                       chpl_external_array *dest) {
    (void)lower; (void)upper; /* ignored exactly as in the reference (StatesEnumeration.chpl:596) */
    uint64_t *d_states = NULL;
    int64_t count = 0;
    if (ls_amd_enumerate_states(basisPtr, 1, &d_states, NULL, &count, NULL) != 0) {
        halt_with("ls_chpl_enumerate_representatives: %s", g_last_error);
        return;
    }
    uint64_t *h = (uint64_t *)malloc(8 * (size_t)(count > 0 ? count : 1));
    if (lsk_d2h(h, d_states, 8 * (size_t)count) != 0) { free(h); lsk_free(d_states); halt_with("%s", lsk_last_error()); return; }
    lsk_free(d_states);
    dest->elts = h;
    dest->num_elts = (uint64_t)count;
    dest->freer = (void *)free;
}

static int ensure_device_reps(ls_hs_basis *b) {
    struct ls_amd_basis_ext *e = BEXT(b);
    if (!b->representatives.elts) return set_error("basis is not built"); /* ForeignTypes.chpl:113-114 */
    if (e->d_reps_cache && e->d_reps_count == b->representatives.num_elts) return 0;
    basis_drop_device_caches(b);
    void *p;
    DEV(lsk_malloc(&p, 8 * (size_t)b->representatives.num_elts));
    DEV(lsk_h2d(p, b->representatives.elts, 8 * (size_t)b->representatives.num_elts));
    e->d_reps_cache = (uint64_t *)p;
    e->d_reps_count = b->representatives.num_elts;
    return 0;
}

/* ---- the host-pointer boundary (DMV:1095-1110, Diagonalize.chpl:134-162) -------------------------------------------------
 * The reference's callers hand over `double *`.  What that costs here is decided by where the memory lives (stage.cpp):
 *   device memory (hipMalloc, torch)      used in place, zero copies
 *   pinned / registered host memory       one DMA per direction (ls_amd_host_register: once per workspace, PRIMME reuses its)
 *   pageable host memory                  double-buffered pinned bounce chunks, upload and download at the same time
 * x and y have persistent device copies in the cached plan slot (no hipMalloc / hipFree per call); y is uploaded only when the
 * operator has no diagonal terms (else it is assigned, DMV:1062-1063); the columns of a block go through one pipeline.
 * LS_AMD_STAGE=0 restores plain synchronous hipMemcpy (A/B in bench.py's boundary_host_ptr). */
static lsk_stager *g_stager = NULL;
static pthread_mutex_t g_stager_lock = PTHREAD_MUTEX_INITIALIZER;
static struct ls_amd_boundary_stats g_bstats;
void ls_amd_boundary_stats_get(struct ls_amd_boundary_stats *out, int reset) {
    *out = g_bstats;
    if (reset) memset(&g_bstats, 0, sizeof(g_bstats));
}
int ls_amd_host_register(void *p, size_t bytes) {
    if (lsk_host_register(p, bytes) != 0) return set_error("%s", lsk_stage_last_error());
    return 0;
}
int ls_amd_host_unregister(void *p) {
    if (lsk_host_unregister(p) != 0) return set_error("%s", lsk_stage_last_error());
    return 0;
}
int ls_amd_pointer_kind(void const *p) { return lsk_pointer_kind(p); }
/* LS_AMD_STAGE: 1 (default) = pinned bounce pipeline (pageable memory; registered memory: one DMA per direction), 0 = the runtime's own
 * copies one after the other (what a single vector costs either way: 179 vs 182 ms on chain_32; a block of four columns: 716 vs 555 ms) */
static int staging_mode(void) { char const *e = getenv("LS_AMD_STAGE"); return e ? atoi(e) : 1; }
/* (caller holds g_stager_lock -- and keeps it while it uses the stager: a change of the chunk knob destroys and recreates it) */
static lsk_stager *stager_locked(void) {
    char const *e = getenv("LS_AMD_STAGE_CHUNK_KB"), *t = getenv("LS_AMD_STAGE_THREADS");
    size_t const chunk = (e && atoi(e) > 0 ? (size_t)atoi(e) : (size_t)32768) << 10;
    if (g_stager && lsk_stager_chunk(g_stager) != (chunk < 4096 ? 4096 : chunk & ~(size_t)4095)) { lsk_stager_destroy(g_stager); g_stager = NULL; }
    if (!g_stager && lsk_stager_create(&g_stager, chunk, t ? atoi(t) : 0) != 0) { g_stager = NULL; set_error("%s", lsk_stage_last_error()); }
    return g_stager;
}
/* up: host -> device, down: device -> host, at the same time; kinds from lsk_pointer_kind (never LSK_PTR_DEVICE here) */
static int transfer(void *d_up, void const *h_up, size_t up_bytes, int up_kind, void *h_down, void const *d_down, size_t down_bytes,
                    int down_kind) {
    if (!up_bytes && !down_bytes) return 0;
    __atomic_fetch_add(&g_bstats.bytes_h2d, (int64_t)up_bytes, __ATOMIC_RELAXED); /* loop-back ranks call this from several threads */
    __atomic_fetch_add(&g_bstats.bytes_d2h, (int64_t)down_bytes, __ATOMIC_RELAXED);
    int const mode = staging_mode();
    if (mode == 0 && up_kind != LSK_PTR_MANAGED && down_kind != LSK_PTR_MANAGED) {
        if (up_bytes) DEV(lsk_h2d(d_up, h_up, up_bytes));
        if (down_bytes) DEV(lsk_d2h(h_down, d_down, down_bytes));
        return 0;
    }

    pthread_mutex_lock(&g_stager_lock);
    lsk_stager *st = stager_locked();
    int rc = st ? 0 : -1;
    if (st && lsk_stage_run(st, d_up, h_up, up_bytes, up_kind, h_down, d_down, down_bytes, down_kind) != 0) rc = set_error("%s", lsk_stage_last_error());
    pthread_mutex_unlock(&g_stager_lock);
    return rc;
}

/* localMatrixVector on `ncols` host (or device) vectors: column k at x + k ldx, y + k ldy.  The memory kind is looked up ONCE,
 * at x and at y: all columns of a block must live in one kind of memory (one PRIMME workspace does). */
static int host_matvec_block(ls_hs_operator *op, int64_t n, int ncols, double const *x, int64_t ldx, double *y, int64_t ldy,
                             ls_amd_comm *cm) {
    ls_hs_basis *b = op->basis;
    struct ls_amd_basis_ext *e = BEXT(b);
    if (ensure_device_reps(b) != 0) return -1;
    if ((uint64_t)n != e->d_reps_count) return set_error("vector length does not match the number of representatives");
    if (ncols < 1) return 0;
    if (!cm) cm = ls_amd_default_comm();
    if (cm && ls_amd_comm_size(cm) <= 1) cm = NULL;
    /* the cached plan of THIS operator on THIS communicator (least recently used slot is recycled) */
    struct host_plan_slot *sl = NULL, *victim = &e->host_plans[0];
    for (int i = 0; i < HOST_PLAN_SLOTS; ++i) {
        struct host_plan_slot *c = &e->host_plans[i];
        if (c->op == op && c->comm == (void *)cm) { sl = c; break; }
        if (!c->op) { if (victim->op) victim = c; }
        else if (victim->op && c->stamp < victim->stamp) victim = c;
    }
    if (!sl) {
        host_plan_slot_drop(victim);
        if (cm) {
            /* one locale per process: `representatives` is this locale's block of the hashed basis and x, y are the
             * matching blocks (Diagonalize.chpl:134-162 runs the callback on every locale in lock-step) */
            ls_amd_dist *dd;
            if (ls_amd_dist_create(&dd, cm, op, LS_AMD_F64, e->d_reps_cache, n, 0, NULL) != 0) return -1;
            victim->dist = dd;
        } else {
            ls_amd_plan *pl;
            uint64_t const *reps[1] = {e->d_reps_cache};
            int64_t counts[1] = {n};
            if (ls_amd_plan_create(&pl, op, LS_AMD_F64, 1, -1, reps, counts, 0, LS_AMD_MODE_AUTO, NULL) != 0) return -1;
            victim->plan = pl;
        }
        victim->op = op;
        victim->comm = cm;
        sl = victim;
    }
    sl->stamp = ++e->host_plan_clock;
    ls_amd_plan *plan = cm ? ls_amd_dist_plan((ls_amd_dist *)sl->dist) : (ls_amd_plan *)sl->plan;
    size_t const bytes = 8 * (size_t)n;
    int const xk = lsk_pointer_kind(x), yk = lsk_pointer_kind(y);
    int const upload_y = !op->diag_terms || op->diag_terms->number_terms == 0; /* y += H x: the caller's y matters */
    int const nbuf = ncols > 1 ? 2 : 1;
    for (int i = 0; i < nbuf; ++i) {
        if (xk != LSK_PTR_DEVICE && !sl->d_x[i]) DEV(lsk_malloc(&sl->d_x[i], bytes));
        if (yk != LSK_PTR_DEVICE && !sl->d_y[i]) DEV(lsk_malloc(&sl->d_y[i], bytes));
    }
    sl->stage_n = n;
    __atomic_fetch_add(&g_bstats.calls, 1, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_bstats.columns, ncols, __ATOMIC_RELAXED);
    if (xk == LSK_PTR_DEVICE) __atomic_fetch_add(&g_bstats.device_x, ncols, __ATOMIC_RELAXED);
    if (yk == LSK_PTR_DEVICE) __atomic_fetch_add(&g_bstats.device_y, ncols, __ATOMIC_RELAXED);
    /* pipeline over the columns: [upload x_0 (+ y_0)] ; for k: launch matvec_k | upload x_{k+1} (+ y_{k+1}) and download y_{k-1}
     * while it runs | check ; [download y_last] */
#define XDEV(k) (xk == LSK_PTR_DEVICE ? (void *)(x + (int64_t)(k) * ldx) : sl->d_x[(k) & (nbuf - 1)])
#define YDEV(k) (yk == LSK_PTR_DEVICE ? (void *)(y + (int64_t)(k) * ldy) : sl->d_y[(k) & (nbuf - 1)])
    if (xk != LSK_PTR_DEVICE && transfer(XDEV(0), x, bytes, xk, NULL, NULL, 0, 0) != 0) return -1;
    if (yk != LSK_PTR_DEVICE && upload_y && transfer(YDEV(0), y, bytes, yk, NULL, NULL, 0, 0) != 0) return -1;
    for (int k = 0; k < ncols; ++k) {
        int rc;
        if (cm) rc = ls_amd_dist_matvec((ls_amd_dist *)sl->dist, XDEV(k), YDEV(k), NULL);
        else {
            void const *xs[1] = {XDEV(k)};
            void *ys[1] = {YDEV(k)};
            rc = ls_amd_matvec(plan, xs, ys, NULL);
        }
        /* (error exits after a launch wait for the device: the slot's persistent d_x / d_y must not be overwritten by the next call
         * while a kernel of this one still reads them -- ADVICE r5) */
#define FAIL_AFTER_LAUNCH do { (void)lsk_device_sync(); return -1; } while (0)
        if (rc != 0) FAIL_AFTER_LAUNCH;
        /* while the kernel runs: the next column goes up, the previous result comes down (different buffers) */
        int const up = k + 1 < ncols && xk != LSK_PTR_DEVICE, down = k >= 1 && yk != LSK_PTR_DEVICE;
        if ((up || down) && transfer(up ? XDEV(k + 1) : NULL, up ? x + (int64_t)(k + 1) * ldx : NULL, up ? bytes : 0, xk,
                                     down ? y + (int64_t)(k - 1) * ldy : NULL, down ? YDEV(k - 1) : NULL, down ? bytes : 0, yk) != 0) FAIL_AFTER_LAUNCH;
        /* (y += H x: the next column's y goes up only now -- its buffer held the result that has just come down) */
        if (k + 1 < ncols && yk != LSK_PTR_DEVICE && upload_y &&
            transfer(YDEV(k + 1), y + (int64_t)(k + 1) * ldy, bytes, yk, NULL, NULL, 0, 0) != 0) FAIL_AFTER_LAUNCH;
        if (ls_amd_plan_check(plan, NULL) != 0) FAIL_AFTER_LAUNCH; /* synchronises the launch stream; halts on an invalid index */
#undef FAIL_AFTER_LAUNCH
    }
    if (yk != LSK_PTR_DEVICE &&
        transfer(NULL, NULL, 0, 0, y + (int64_t)(ncols - 1) * ldy, YDEV(ncols - 1), bytes, yk) != 0) return -1;
#undef XDEV
#undef YDEV
    return 0;
}

void ls_chpl_matrix_vector_product(ls_hs_operator *matrixPtr, int numVectors, double *xPtr, double *yPtr) {
    if (ls_hs_basis_number_words(matrixPtr->basis) != 1) { halt_with("bases with more than 64 bits are not yet implemented"); return; }
    if (numVectors != 1) { halt_with("applying the Operator to more than 1 vector is not yet implemented"); return; }
    if (!matrixPtr->basis->representatives.elts) { halt_with("basis is not built"); return; }
    int64_t n = (int64_t)matrixPtr->basis->representatives.num_elts;
    if (host_matvec_block(matrixPtr, n, 1, xPtr, n, yPtr, n, NULL) != 0) halt_with("%s", g_last_error);
}

void ls_chpl_primme_matvec(void *x, int64_t *ldx, void *y, int64_t *ldy, int *blockSize, void *primme,
                           int *ierr) {
    ls_primme_params_view *pp = (ls_primme_params_view *)primme;
    ls_hs_operator *op = (ls_hs_operator *)pp->matrix;
    int64_t n = pp->nLocal;
    *ierr = 0;
    if (*ldx < n || *ldy < n) { *ierr = -1; return; }
    /* all blockSize columns through ONE pipeline (x_{k+1} up and y_{k-1} down while column k computes) */
    if (host_matvec_block(op, n, *blockSize, (double const *)x, *ldx, (double *)y, *ldy, (ls_amd_comm *)pp->commInfo) != 0) {
        halt_with("%s", g_last_error);
        *ierr = -1;
    }
}
/* primmeGlobalSumReal / primmeBroadcastReal: dist.c (collective over the communicator) */

static void free_array(void *p) { free(p); }

void ls_chpl_operator_apply_diag(ls_hs_operator *matrixPtr, int64_t count, uint64_t *alphas,
                                 chpl_external_array *coeffs, int64_t numTasks) {
    (void)numTasks;
    if (ls_hs_basis_number_words(matrixPtr->basis) != 1) { halt_with("bases with more than 64 bits are not yet implemented"); return; }
    if (matrixPtr->basis->requires_projection) { halt_with("bases that require projection are not yet supported"); return; }
    lsk_operator dop;
    if (operator_device(matrixPtr, &dop) != 0) { halt_with("%s", g_last_error); return; }
    double *h = (double *)malloc(8 * (size_t)(count > 0 ? count : 1));
    void *da = NULL, *dy = NULL;
    int rc = lsk_malloc(&da, 8 * (size_t)count) || lsk_malloc(&dy, 8 * (size_t)count) ||
             lsk_h2d(da, alphas, 8 * (size_t)count) ||
             lsk_diag_coeffs(dop, count, (uint64_t const *)da, (double *)dy, NULL, NULL) || lsk_sync(NULL) ||
             lsk_d2h(h, dy, 8 * (size_t)count);
    lsk_free(da); lsk_free(dy);
    if (rc) { free(h); halt_with("%s", lsk_last_error()); return; }
    coeffs->elts = h;
    coeffs->num_elts = (uint64_t)count;
    coeffs->freer = (void *)free_array;
}

void ls_chpl_operator_apply_off_diag(ls_hs_operator *matrixPtr, int64_t count, uint64_t *alphas,
                                     chpl_external_array *betas, chpl_external_array *coeffs,
                                     chpl_external_array *offsets, int64_t numTasks) {
    (void)numTasks;
    if (ls_hs_basis_number_words(matrixPtr->basis) != 1) { halt_with("bases with more than 64 bits are not yet implemented"); return; }
    if (matrixPtr->basis->requires_projection) { halt_with("bases that require projection are not yet supported"); return; }
    int const T = OEXT(matrixPtr)->n_groups;
    int64_t *h_off = (int64_t *)calloc((size_t)count + 1, sizeof(int64_t));
    if (T == 0) {
        betas->elts = NULL; betas->num_elts = 0; betas->freer = NULL;
        coeffs->elts = NULL; coeffs->num_elts = 0; coeffs->freer = NULL;
        offsets->elts = h_off; offsets->num_elts = (uint64_t)count + 1; offsets->freer = (void *)free_array;
        return;
    }
    lsk_operator dop;
    if (operator_device(matrixPtr, &dop) != 0) { free(h_off); halt_with("%s", g_last_error); return; }
    size_t cap = (size_t)count * (size_t)T;
    uint64_t *h_b = (uint64_t *)calloc(cap ? cap : 1, 8);
    double *h_c = (double *)calloc(cap ? cap : 1, 16);
    void *da = NULL, *dcnt = NULL, *doff = NULL, *db = NULL, *dc = NULL;
    int rc = lsk_malloc(&da, 8 * (size_t)count) || lsk_malloc(&dcnt, 8 * ((size_t)count + 1)) ||
             lsk_malloc(&doff, 8 * ((size_t)count + 1)) || lsk_malloc(&db, 8 * cap) || lsk_malloc(&dc, 16 * cap) ||
             lsk_h2d(da, alphas, 8 * (size_t)count) || lsk_memset_async(dcnt, 0, 8 * ((size_t)count + 1), NULL) ||
             lsk_offdiag_counts(dop, count, (uint64_t const *)da, (int64_t *)dcnt, NULL) ||
             lsk_exclusive_scan_i64(count + 1, (int64_t const *)dcnt, (int64_t *)doff, NULL) ||
             lsk_offdiag_fill(dop, count, (uint64_t const *)da, (int64_t const *)doff, (uint64_t *)db, (double *)dc, NULL, NULL) ||
             lsk_sync(NULL) || lsk_d2h(h_off, doff, 8 * ((size_t)count + 1));
    if (!rc) {
        size_t total = (size_t)h_off[count];
        rc = lsk_d2h(h_b, db, 8 * total) || lsk_d2h(h_c, dc, 16 * total);
    }
    lsk_free(da); lsk_free(dcnt); lsk_free(doff); lsk_free(db); lsk_free(dc);
    if (rc) { free(h_off); free(h_b); free(h_c); halt_with("%s", lsk_last_error()); return; }
    betas->elts = h_b; betas->num_elts = cap; betas->freer = (void *)free_array;
    coeffs->elts = h_c; coeffs->num_elts = cap; coeffs->freer = (void *)free_array;
    offsets->elts = h_off; offsets->num_elts = (uint64_t)count + 1; offsets->freer = (void *)free_array;
}

/* ============================================================================================ */
/* batched externs with the reference's signatures (host arrays, device kernels)                 */
/* ============================================================================================ */
static uint64_t *gather_u64(uint64_t const *src, ptrdiff_t n, ptrdiff_t stride) {
    uint64_t *t = (uint64_t *)malloc(8 * (size_t)(n > 0 ? n : 1));
    for (ptrdiff_t i = 0; i < n; ++i) t[i] = src[i * stride];
    return t;
}

void ls_hs_state_index(ls_hs_basis const *basis_c, ptrdiff_t n, uint64_t const *spins, ptrdiff_t sstride,
                       ptrdiff_t *indices, ptrdiff_t istride) {
    ls_hs_basis *b = (ls_hs_basis *)basis_c;
    struct ls_amd_basis_ext *e = BEXT(b);
    if (n <= 0) return;
    if (ensure_device_reps(b) != 0) { halt_with("%s", g_last_error); return; }
    uint64_t const *d_binom;
    if (device_binom(&d_binom) != 0) { halt_with("%s", g_last_error); return; }
    lsk_index ix;
    memset(&ix, 0, sizeof(ix));
    ix.count = (int64_t)e->d_reps_count;
    ix.reps = e->d_reps_cache;
    ix.binom = d_binom;
    if (b->state_index_is_identity) ix.kind = LSK_INDEX_IDENTITY;
    else {
        if (!e->d_index_table) {
            part_state ps;
            memset(&ps, 0, sizeof(ps));
            ps.count = ix.count;
            ps.d_reps = ix.reps;
            if (build_search_index(&ps, b->number_sites, NULL) != 0 || lsk_sync(NULL) != 0) { halt_with("%s", g_last_error); return; }
            e->d_index_table = ps.d_table;
            e->index_shift = ps.index.shift;
        }
        ix.kind = LSK_INDEX_SEARCH;
        ix.table = e->d_index_table;
        ix.shift = e->index_shift;
    }
    uint64_t *h = gather_u64(spins, n, sstride);
    int64_t *hi = (int64_t *)malloc(8 * (size_t)n);
    void *ds = NULL, *di = NULL;
    int rc = lsk_malloc(&ds, 8 * (size_t)n) || lsk_malloc(&di, 8 * (size_t)n) || lsk_h2d(ds, h, 8 * (size_t)n) ||
             lsk_state_index(ix, n, (uint64_t const *)ds, (int64_t *)di, NULL) || lsk_sync(NULL) ||
             lsk_d2h(hi, di, 8 * (size_t)n);
    lsk_free(ds); lsk_free(di);
    if (!rc) for (ptrdiff_t i = 0; i < n; ++i) indices[i * istride] = (ptrdiff_t)hi[i];
    free(h); free(hi);
    if (rc) halt_with("%s", lsk_last_error());
}

static int state_info_host(ls_hs_basis const *b, ptrdiff_t n, uint64_t const *alphas, ptrdiff_t astride,
                           uint64_t *betas, double *chars, double *norms) {
    lsk_basis dbs;
    if (basis_device(b, &dbs) != 0) return -1;
    uint64_t *h = gather_u64(alphas, n, astride);
    void *da = NULL, *db = NULL, *dc = NULL, *dn = NULL;
    int rc = lsk_malloc(&da, 8 * (size_t)n) || lsk_malloc(&db, 8 * (size_t)n) || lsk_malloc(&dc, 16 * (size_t)n) ||
             lsk_malloc(&dn, 8 * (size_t)n) || lsk_h2d(da, h, 8 * (size_t)n) ||
             lsk_state_info(dbs, n, (uint64_t const *)da, (uint64_t *)db, (double *)dc, (double *)dn, NULL) ||
             lsk_sync(NULL) || lsk_d2h(betas, db, 8 * (size_t)n) || lsk_d2h(chars, dc, 16 * (size_t)n) ||
             lsk_d2h(norms, dn, 8 * (size_t)n);
    lsk_free(da); lsk_free(db); lsk_free(dc); lsk_free(dn);
    free(h);
    return rc ? dev_error() : 0;
}

void ls_hs_state_info(ls_hs_basis const *basis, ptrdiff_t n, uint64_t const *alphas, ptrdiff_t astride,
                      uint64_t *betas, ptrdiff_t bstride, ls_hs_scalar *characters, double *norms) {
    if (n <= 0) return;
    uint64_t *hb = (uint64_t *)malloc(8 * (size_t)n);
    if (state_info_host(basis, n, alphas, astride, hb, (double *)characters, norms) != 0) { free(hb); halt_with("%s", g_last_error); return; }
    for (ptrdiff_t i = 0; i < n; ++i) betas[i * bstride] = hb[i];
    free(hb);
}

void ls_hs_is_representative(ls_hs_basis const *basis, ptrdiff_t n, uint64_t const *alphas, ptrdiff_t astride,
                             uint8_t *are_representatives, double *norms) {
    if (n <= 0) return;
    uint64_t *hb = (uint64_t *)malloc(8 * (size_t)n);
    double *hc = (double *)malloc(16 * (size_t)n);
    if (state_info_host(basis, n, alphas, astride, hb, hc, norms) != 0) { free(hb); free(hc); halt_with("%s", g_last_error); return; }
    for (ptrdiff_t i = 0; i < n; ++i) are_representatives[i] = hb[i] == alphas[i * astride];
    free(hb); free(hc);
}

void ls_internal_operator_apply_diag_x1(ls_hs_operator const *op, ptrdiff_t n, uint64_t const *alphas, double *ys,
                                        double const *xs) {
    if (n <= 0) return;
    lsk_operator dop;
    if (operator_device(op, &dop) != 0) { halt_with("%s", g_last_error); return; }
    void *da = NULL, *dy = NULL, *dx = NULL;
    int rc = lsk_malloc(&da, 8 * (size_t)n) || lsk_malloc(&dy, 8 * (size_t)n) || lsk_h2d(da, alphas, 8 * (size_t)n);
    if (!rc && xs) rc = lsk_malloc(&dx, 8 * (size_t)n) || lsk_h2d(dx, xs, 8 * (size_t)n);
    if (!rc) rc = lsk_diag_coeffs(dop, n, (uint64_t const *)da, (double *)dy, (double const *)dx, NULL) || lsk_sync(NULL) ||
                  lsk_d2h(ys, dy, 8 * (size_t)n);
    lsk_free(da); lsk_free(dy); lsk_free(dx);
    if (rc) halt_with("%s", lsk_last_error());
}

void ls_internal_operator_apply_off_diag_x1(ls_hs_operator const *op, ptrdiff_t n, uint64_t const *alphas,
                                            uint64_t *betas, ls_hs_scalar *coeffs, ptrdiff_t *offsets,
                                            double const *xs) {
    offsets[0] = 0;
    if (n <= 0) return;
    int const T = OEXT(op)->n_groups;
    if (T == 0) { for (ptrdiff_t i = 0; i <= n; ++i) offsets[i] = 0; return; }
    lsk_operator dop;
    if (operator_device(op, &dop) != 0) { halt_with("%s", g_last_error); return; }
    size_t cap = (size_t)n * (size_t)T;
    int64_t *h_off = (int64_t *)malloc(8 * ((size_t)n + 1));
    void *da = NULL, *dcnt = NULL, *doff = NULL, *db = NULL, *dc = NULL, *dx = NULL;
    int rc = lsk_malloc(&da, 8 * (size_t)n) || lsk_malloc(&dcnt, 8 * ((size_t)n + 1)) ||
             lsk_malloc(&doff, 8 * ((size_t)n + 1)) || lsk_malloc(&db, 8 * cap) || lsk_malloc(&dc, 16 * cap) ||
             lsk_h2d(da, alphas, 8 * (size_t)n) || lsk_memset_async(dcnt, 0, 8 * ((size_t)n + 1), NULL);
    if (!rc && xs) rc = lsk_malloc(&dx, 8 * (size_t)n) || lsk_h2d(dx, xs, 8 * (size_t)n);
    if (!rc) rc = lsk_offdiag_counts(dop, n, (uint64_t const *)da, (int64_t *)dcnt, NULL) ||
                  lsk_exclusive_scan_i64(n + 1, (int64_t const *)dcnt, (int64_t *)doff, NULL) ||
                  lsk_offdiag_fill(dop, n, (uint64_t const *)da, (int64_t const *)doff, (uint64_t *)db, (double *)dc,
                                   (double const *)dx, NULL) ||
                  lsk_sync(NULL) || lsk_d2h(h_off, doff, 8 * ((size_t)n + 1));
    if (!rc) {
        size_t total = (size_t)h_off[n];
        rc = lsk_d2h(betas, db, 8 * total) || lsk_d2h(coeffs, dc, 16 * total);
        for (ptrdiff_t i = 0; i <= n; ++i) offsets[i] = (ptrdiff_t)h_off[i];
    }
    lsk_free(da); lsk_free(dcnt); lsk_free(doff); lsk_free(db); lsk_free(dc); lsk_free(dx);
    free(h_off);
    if (rc) halt_with("%s", lsk_last_error());
}
