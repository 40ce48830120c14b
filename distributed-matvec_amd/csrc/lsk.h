/* lsk.h -- internal interface between the C host side (host.c) and the HIP side (k_*.hip: runtime, rows, packets, pull, plan; shared device code in lsk_dev.hpp).
 * "Thin extern-C shim": host.c never includes a HIP header; everything device-related goes through
 * the lsk_* functions declared here.  All structs are plain C PODs passed by value to kernels.
 */
#ifndef LSK_H
#define LSK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSK_BENES_STAGES 11 /* distances 32,16,8,4,2,1,2,4,8,16,32 */
#define LSK_BINOM_K 34      /* binomial table is [64][LSK_BINOM_K]: C(n, k), n < 64, k < 34 */
#define LSK_MAX_PARTS 256   /* DMV:664 */

typedef struct lsk_term {
    double v_re, v_im;
    uint64_t m, r, s;
} lsk_term;

/* how the coefficient of a flip-mask group is evaluated */
enum { LSK_GROUP_GENERIC = 0, LSK_GROUP_EXCHANGE = 1,
       /* round 6: a DIRECTED pair, e.g. sigma^+_i sigma^-_j -- coefficient v iff exactly ONE site of the pair is set in alpha and it is
        * the pair's lower (HOP_LO) / upper (HOP_HI) site.  Non-Hermitian hopping then takes the same fast paths as an exchange. */
       LSK_GROUP_HOP_LO = 2, LSK_GROUP_HOP_HI = 3 };

/* one off-diagonal flip-mask group: beta = alpha ^ x, coefficient = sum over terms [begin, end) */
typedef struct lsk_group {
    uint64_t x;
    double v_re, v_im; /* EXCHANGE: coefficient when popcount(alpha & x) == 1, else 0; HOP_*: when alpha & x is the source site alone */
    int32_t begin, end;
    int32_t adj;       /* lo if x == 3 << lo (adjacent pair) else -1 */
    int32_t fast;      /* LSK_GROUP_* */
} lsk_group;

/* Structure the host recognises so that the row kernels can run tight, branch-free inner loops:
 *  - exchange runs: groups [0, n_run_groups) are EXCHANGE groups on adjacent pairs (lo, lo + 1) with
 *    consecutive lo and one common amplitude v  (e.g. the 31 open bonds of the Heisenberg ring);
 *  - zz runs: diagonal terms [0, n_zz_terms) are v * (-1)^popcount(alpha & (3 << lo)) with
 *    consecutive lo and a common real v, whose sum is v * (cnt - 2 * #anti-aligned pairs). */
#define LSK_MAX_RUNS 4
typedef struct lsk_runs {
    int n_runs, n_run_groups;
    int lo0[LSK_MAX_RUNS], cnt[LSK_MAX_RUNS]; /* cnt: pairs in the run | direction << 16 (0: exchange, 1: HOP_LO, 2: HOP_HI groups; k_direct
                                               * decodes it -- the staged chain kernel only ever sees Hermitian operators, i.e. direction 0) */
    double v_re[LSK_MAX_RUNS], v_im[LSK_MAX_RUNS];
    int n_zz, n_zz_terms;
    int zz_lo0[LSK_MAX_RUNS], zz_cnt[LSK_MAX_RUNS];
    double zz_v[LSK_MAX_RUNS];
} lsk_runs;

typedef struct lsk_operator {
    int n_diag, n_off, n_groups, is_real;
    int uni;      /* every off-diagonal group is an exchange pair (LSK_GROUP_EXCHANGE) with ONE real amplitude uni_v: the projected */
    double uni_v; /* pull kernels then keep no coefficient per packet */
    lsk_term const *diag;    /* device [n_diag]; zz-run terms first */
    lsk_term const *off;     /* device [n_off], sorted by group */
    lsk_group const *groups; /* device [n_groups]; exchange-run groups first */
    lsk_runs runs;
} lsk_operator;

enum { LSK_ELEM_BENES = 0, LSK_ELEM_ROT = 1, LSK_ELEM_REVROT = 2 };

typedef struct lsk_group_elem {
    int32_t kind;
    int32_t k; /* rotate-right amount within number_sites bits (ROT / REVROT) */
    double ch_re, ch_im;
    uint64_t masks[LSK_BENES_STAGES];
} lsk_group_elem;

enum { LSK_PROJ_NONE = 0, LSK_PROJ_INVERSION = 1, LSK_PROJ_FULL = 2 };

typedef struct lsk_basis {
    int number_sites, hamming_weight, spin_inversion, n_elems, proj;
    int chars_pm1; /* every character is +1 or -1 (integer stabiliser accumulation) */
    /* K4 evaluation mode of the staged kernel:
     *  0 general: orbit minimum + character + stabiliser norm per packet;
     *  1 trivial sector (every character, incl. inversion, is +1): only the orbit minimum is needed,
     *    norm(rep) is read from the owner's per-row norms when the index is looked up;
     *  2 as 1, and the permutation group is the full cyclic (rotation) group of the ring, with
     *    (reflect = 1) or without its reflections: rotations are generated incrementally;
     *  3 as 2, but only the rotations that start at a longest run of zeros are visited;
     *  4 as 1, for a group that contains every translation of a tw x (L / tw) torus (site = y tw + x): G = union of the
     *    right cosets T g_r, so the orbit is { t(g_r(a)) }: n_cosets compiled networks (the point group) and |G| cheap
     *    translation steps (rotate the rows by one site / the word by one row) instead of |G| networks.
     *  5 as 4, when those cosets are (modulo translations) the point group D2 = {1, r, o, r o} of the torus -- r reverses every
     *    row, o the order of the rows -- or, on a square torus, D4 = D2 x {1, transpose}, or a subgroup of it: d4_mask says
     *    which images belong to the group (bits 0-3: 1, r, o, r o of the word; bits 4-7: of its transpose), cosets[0] is the
     *    transpose network -- the only compiled network left -- and trow2 the row table with the fields of the reversed row
     *    in its high half (torus_min_d2, lsk_dev.hpp). */
    int k4_mode, reflect;
    int d4_mask;                    /* mode 5 */
    uint64_t const *trow2;          /* mode 5: device [2^tw] */
    int tw, n_cosets;               /* mode 4 / 5 */
    uint64_t tcol0;                 /* mode 4: the bits of column x = 0 */
    lsk_group_elem const *cosets;   /* mode 4: device [n_cosets] */
    uint32_t const *trow;           /* mode 4, tw <= 8: device [2^tw] row table of torus_min (lsk_torus_rowtab), else NULL */
    int debug_ablate; /* LS_AMD_ABLATE bitmask (profiling only): 1 skip stage B, 2 skip lookup+accumulate, 4 skip K4 */
    uint64_t site_mask;
    double inv_order; /* 1 / |G| including the inversion doubling */
    lsk_group_elem const *elems; /* device [n_elems] */
} lsk_basis;

enum { LSK_INDEX_IDENTITY = 0, LSK_INDEX_COMBINADIC = 1, LSK_INDEX_SEARCH = 2 };

/* Rank directory of a hash partition of an unprojected fixed-weight basis (optional part of a SEARCH index): the global
 * (colex) rank g of a state is closed-form, and the local index is the number of this partition's states below it --
 *     entry = dir[g >> 6];  local = entry.prefix + popcount(entry.bits & below(g));  member <=> bit g of entry.bits
 * one 16-byte load instead of the prefix-table look-up and the 3-4 dependent probes of the binary search. */
typedef struct lsk_rankdir {
    uint64_t bits;   /* bit j: global rank 64 w + j belongs to this partition */
    uint32_t prefix; /* states of this partition with global rank < 64 w */
    uint32_t pad;
} lsk_rankdir;

typedef struct lsk_index {
    int kind;
    int shift;             /* SEARCH: bucket = state >> shift */
    int64_t count;
    uint64_t const *reps;  /* device, ascending */
    uint32_t const *table; /* device [(max_state >> shift) + 2] lower bounds */
    uint64_t const *binom; /* device [64 * LSK_BINOM_K] */
    lsk_rankdir const *dir; /* device [ceil(C(dir_sites, dir_weight) / 64)] or NULL */
    int dir_sites, dir_weight;
} lsk_index;
/* dir[] for the n ascending states `reps` (all of weight `weight` on `sites` sites), entries = ceil(C(sites, weight) / 64); *d_flag is
 * raised if a state has another weight or the directory does not give reps[i] -> i back */
int lsk_rankdir_build(int64_t n, uint64_t const *reps, int sites, int weight, uint64_t const *d_binom, int64_t entries, lsk_rankdir *dir,
                      int *d_flag, void *stream);

/* All-destinations rank directory of the hash partition of an unprojected fixed-weight basis (pre-indexed packets): entry
 * [w * P + d] describes what partition d owns of the global (colex) ranks [64 w, 64 w + 64).  The owner of a state is a hash
 * of the state, so every rank builds the whole table by itself (unrank -> owner, one pass over the basis at plan time).  The
 * PRODUCER of a packet then knows the index of beta inside its destination's block:
 *     idx = e.prefix + popcount(e.bits & below(g)),  e = entries[(g >> 6) * P + owner(beta)],  member <=> bit g & 63 of e.bits
 * and sends (u32 idx, value): 12 instead of 16 bytes per packet, and the consumer is search-free.  P / 4 bytes per basis state. */
typedef struct lsk_gdir {
    lsk_rankdir const *entries; /* device [words * P]; NULL = no directory (packets carry the state) */
    int P, sites, weight;
    int64_t n_ranks;            /* global ranks [0, n_ranks) are basis states (C(sites, weight), or its lower half under inversion) */
} lsk_gdir;
/* entries: ceil(n_ranks / 64) * P, allocated by the caller; synchronises the stream */
int lsk_gdir_build(lsk_gdir gd, lsk_rankdir *entries, uint64_t const *d_binom, void *stream);
/* *d_flag is raised unless the directory gives reps[i] -> (part, i) back for the n ascending states of partition `part` */
int lsk_gdir_check(lsk_gdir gd, int part, int64_t n, uint64_t const *reps, uint64_t const *d_binom, int *d_flag, void *stream);

/* the segments of one fused consumer launch (one round's receive buffer, or the send buffer of a logical partition) */
#define LSK_MAX_SEGS 64
typedef struct lsk_segs {
    int n;
    int64_t start[LSK_MAX_SEGS + 1]; /* exclusive prefix of the packet counts */
    int64_t key_off[LSK_MAX_SEGS];   /* byte offset of the segment's u64 states / u32 indices */
    int64_t val_off[LSK_MAX_SEGS];   /* ... of its values */
    void *y[LSK_MAX_SEGS];           /* the vector the segment accumulates into (all the same with one partition per process) */
    uint8_t part[LSK_MAX_SEGS];      /* lsk_scatter_parts: the partition (entry of the context array) the segment belongs to */
} lsk_segs;
/* what a consumer needs to know of a destination partition: all partitions of one process (lsk_scatter_parts) */
/* static index table {state -> payload} (k_pull.hip; look-up: lsk_dev.hpp) */
typedef struct lsk_gtab {
    uint64_t const *entries; /* device [2 << bbits] */
    int L, bbits, tbits;
} lsk_gtab;
typedef struct lsk_part_ctx {
    lsk_index ix;
    double const *norms; /* per-row norms multiplied in (K4 modes that prescale), else NULL */
    lsk_gtab gt;         /* entries != NULL: {representative -> index} of the partition -- one 16-byte probe instead of the prefix table
                          * and the binary search (round 6; searched indexes, i.e. projected bases) */
} lsk_part_ctx;
/* every packet of every segment: y[seg][idx] += value; pre-indexed packets (u32 idx) */
int lsk_scatter_idx(int cplx, lsk_segs const *segs, void const *base, void *stream);
/* state-carrying packets whose segments belong to DIFFERENT partitions of this process (logical partitions on one device): the
 * index / norms of segment s are d_parts[segs->part[s]] (device array); all partitions share sites / weight of the rank directory */
int lsk_scatter_parts(lsk_part_ctx const *d_parts, lsk_index any, int cplx, lsk_segs const *segs, void const *base, int *d_err, void *stream);
/* the same for packets that carry the state: ONE index (all segments belong to the same destination partition) */
int lsk_scatter_segs(lsk_index ix, lsk_gtab gt, int cplx, lsk_segs const *segs, void const *base, double const *norms, int *d_err, void *stream);

/* per-round send layout: byte offsets of the beta / value arrays of every destination segment */
typedef struct lsk_round_layout {
    int64_t beta_off[LSK_MAX_PARTS];
    int64_t val_off[LSK_MAX_PARTS];
} lsk_round_layout;

/* Tile map of the row kernels: entry = first row | number of rows (<= 256) << 48.  The map holds 8 lists of
 * slots_per_xcd entries, one per XCD (block b of the launch runs on XCD b % 8 and walks list b % 8);
 * empty slots have a zero row count.  Every row appears in exactly one tile. */
typedef struct lsk_tilemap {
    uint64_t const *entries; /* device [8 * slots_per_xcd] */
    int64_t slots_per_xcd;
} lsk_tilemap;

/* runtime ---------------------------------------------------------------------------------- */
char const *lsk_last_error(void);
int lsk_ablate_mask(void); /* profiling builds (make ablate): the LS_AMD_ABLATE stage switches; the shipped library returns 0 */
int lsk_device_count(void);
int lsk_set_device(int device);
int lsk_malloc(void **p, size_t bytes);
int lsk_free(void *p);
int lsk_mem_info(size_t *free_bytes, size_t *total_bytes);
int lsk_h2d(void *dst, void const *src, size_t bytes);
int lsk_d2h(void *dst, void const *src, size_t bytes);
int lsk_d2d_async(void *dst, void const *src, size_t bytes, void *stream);
int lsk_memset_async(void *p, int value, size_t bytes, void *stream);
int lsk_sync(void *stream);
int lsk_device_sync(void);

/* host <-> HBM staging of the host-pointer entry points (stage.cpp) ---------------------------- */
enum { LSK_PTR_PAGEABLE = 0, LSK_PTR_PINNED = 1, LSK_PTR_DEVICE = 2,
       LSK_PTR_MANAGED = 3 /* hipMallocManaged: fine-grained unless advised otherwise -- never used in place (the push kernels'
                            * global_atomic_add_f64 is specified for coarse-grained memory only); staged by one DMA like pinned memory */ };
char const *lsk_stage_last_error(void);
int lsk_pointer_kind(void const *p); /* hipPointerGetAttributes: device, managed, pinned or registered host, anything else */
int lsk_host_register(void *p, size_t bytes);
int lsk_host_unregister(void *p);
typedef struct lsk_stager lsk_stager;
int lsk_stager_create(lsk_stager **out, size_t chunk_bytes, int threads /* <= 0: min(16, cores / 4) */);
void lsk_stager_destroy(lsk_stager *st);
int lsk_stager_threads(lsk_stager const *st);
size_t lsk_stager_chunk(lsk_stager const *st);
/* one upload and one download at the same time (either may have 0 bytes); *_kind = LSK_PTR_PAGEABLE (double-buffered pinned
 * bounce chunks, host copies by the stager's thread pool) or LSK_PTR_PINNED (one DMA); returns when both are complete */
int lsk_stage_run(lsk_stager *st, void *d_up, void const *h_up, size_t up_bytes, int up_kind, void *h_down, void const *d_down,
                  size_t down_bytes, int down_kind);

/* events (kernel timing on the launch stream) */
int lsk_event_create(void **ev);
int lsk_event_destroy(void *ev);
int lsk_event_record(void *ev, void *stream);
int lsk_event_elapsed_ms(void *start, void *stop, float *ms);

/* deterministic test / bench vectors keyed by the basis state: x_i = u(hash(state_i, seed)) - 0.5 */
int lsk_fill_random(int64_t n, uint64_t const *states, uint64_t seed, int cplx, void *out, void *stream);

/* hot path --------------------------------------------------------------------------------- */
/* y[i] = d(reps[i]) * x[i] */
int lsk_diag(lsk_operator op, int cplx, int64_t n, uint64_t const *reps, void const *x, void *y,
             void *stream);
/* fused single-partition kernel without permutation symmetries (row per lane).
 * pull == 0: y[idx(beta)] += c x[i] (atomics; y must already hold the diagonal part)
 * pull == 1: y[i] = d x[i] + sum conj(c) x[idx(beta)] */
/* plan-time check of a pull plan over a non-Hermitian operator: raises *d_err when the row expansion of some basis state leaves the basis */
int lsk_direct_validate(lsk_operator op, lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *reps, int *d_err, void *stream);
int lsk_direct(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int pull, lsk_tilemap tm,
               uint64_t const *reps, void const *x, void *y, int *d_err, void *stream);
/* staged row kernel (k_chain_t): pull, real Hermitian operator, the full fixed-Hamming-weight basis without symmetries
 * (<= 64 sites; f64 or c128 vectors).  The tile map must have been built with lsk_chain_tile_rows(cplx)-row tiles.  The n
 * rows (reps, cache, y) are the global rows [row0, row0 + n) of the basis; x is the whole vector of n_x elements.
 * wide_ranks != 0: ranks (cache entries, indices into x) are 64-bit (bases with >= 2^32 states). */
int lsk_chain_tile_rows(int cplx);
int lsk_chain(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int wide_ranks, int fused_records, lsk_tilemap tm,
              int64_t n, uint64_t const *reps, int64_t row0, int64_t n_x, void const *x, void *y, int n_cached,
              void const *cache, double cv0, double cv1, void *stream);
/* ---- staged row kernel for arbitrary exchange pairs (k_pairs_t, k_rows.hip): Heisenberg / XXZ on any lattice -------------
 * pairs are sorted by class: [0, n_near) both sites below bit 11, [n_near, n_near + n_str) i < 11 <= j, then both >= 11 */
#define LSK_MAX_PAIRS 128
#define LSK_PAIR_KC 20 /* columns of the binomial table: weight + 2 <= 20, i.e. hamming weight <= 18 (32 sites: 16) */
typedef struct lsk_pair {
    uint8_t i, j, pad[6]; /* i < j */
    double v;             /* exchange amplitude: coefficient of |..0_i..1_j..><..1_i..0_j..| + h.c. */
    double vz;            /* diagonal: vz (-1)^{[bits i, j differ]} */
} lsk_pair;
/* the same pair as the one-row-per-lane variant reads it (k_pairs_row): the masks its rank shift needs, ready-made */
typedef struct lsk_pair_row {
    uint64_t between; /* sites strictly between i and j */
    double v, vz;
    int i, j;         /* i < j */
} lsk_pair_row;
/* ... and as the one-row-per-lane variant that walks the PARTICLES of a row reads it (k_pairs_site): one table of 32-bit words,
 *   [n_sites][degree / 4]  the neighbours of every site, one byte each (slots past the degree of a site hold the site itself:
 *                          never active, it is occupied)
 *   [n_sites][degree / 4]  the amplitude class of every slot, one byte each
 *   [n_classes] x {double v, vz}   (starts on a 16-byte boundary)
 * n_classes == 1 -- one J for all bonds -- keeps v and vz in scalar registers and never reads the class words. */
#define LSK_PAIR_SITE_MAX_DEGREE 8
#define LSK_PAIR_SITE_MAX_CLASSES 64
typedef struct lsk_pairplan {
    int n_near, n_str, n_high;
    lsk_pair const *pairs;     /* device [n_near + n_str + n_high] */
    uint16_t const *rank_low;  /* device [2048]: rank of an 11-bit word among the words of its weight */
    uint32_t const *binom;     /* device [32 | 64][LSK_PAIR_KC] */
    void const *states;        /* device [n]: u32 low words of the representatives (the plan's 4-byte copy), or, wide, the u64 representatives */
    int wide;                  /* 33..64 sites: 8-byte states (ranks stay 32-bit) */
    double dsum;               /* sum of vz over all pairs */
    lsk_pair_row const *rows;  /* device [n_near + n_str + n_high], or NULL: the one-row-per-lane variant (far from half filling) */
    uint32_t const *sites;     /* device: the neighbour table above, or NULL: ... walking particles instead of pairs (weight x degree < pairs) */
    int n_sites, degree;       /* degree: slots per site, 4 or 8 */
    int n_classes, site_words; /* amplitude classes; 32-bit words of the whole table */
} lsk_pairplan;
/* staged push (k_push_t): LDS window of y per tile for near targets + the diagonal part; y cleared by the caller when n_diag > 0 */
int lsk_push_tile_rows(int cplx);
int lsk_push_staged(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, lsk_tilemap tm, int64_t n, uint64_t const *reps,
                    void const *x, void *y, int *d_err, void *stream);
int lsk_pairs_tile_rows(int cplx);
int lsk_pairs(lsk_pairplan pp, int hamming_weight, int cplx, lsk_tilemap tm, int64_t n, void const *x, void *y, void *stream);
int lsk_narrow_states(int64_t n, uint64_t const *reps, uint32_t *out, void *stream);

/* fused_records != 0 (32-bit states and ranks): `reps` is out[] of lsk_chain_pack -- state | partner rank of the first
 * cached pair << 32 (cache == NULL: no cached pair) -- and `cache` only holds a second cached pair at cache + n */
int lsk_chain_pack(int64_t n, uint64_t const *reps, void const *cache, uint64_t *out, void *stream);
/* partner ranks of a non-adjacent exchange pair for every row (u32, or u64 when wide_ranks); *d_flag is raised if a
 * partner leaves the basis */
int lsk_chain_cache(lsk_basis bs, lsk_index ix, int64_t n, uint64_t const *reps, uint64_t xmask, void *out, int wide_ranks,
                    int *d_flag, void *stream);
/* staged kernel (LDS term lists): rows [row0, row1) of partition `me`.
 * count_only != 0: adds the number of packets per destination to d_counts[P] and does nothing else.
 * otherwise: local packets -> index + atomic add into y; remote packets -> d_send according to
 * *d_layout using the per-destination cursors d_cursors[P] (must be zero on entry). */
int lsk_tile(lsk_operator op, lsk_basis bs, lsk_index ix, int cplx, int count_only, int P, int me,
             int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x,
             void *y, unsigned long long *d_cursors, lsk_round_layout const *d_layout, void *d_send,
             unsigned long long *d_counts, int *d_err, void *stream);
/* the same producer with per-wave packet rings and a send layout fixed by the plan (P <= lsk_tile_wv_max_parts()):
 * count_only: d_wtab[wave][P] <- packets of every (wave of 64 rows counted from row0, destination); otherwise d_wtab holds
 * the exclusive offsets of every (wave, destination) inside the round's segments of *d_layout.  No cursors, no atomics on
 * the send side; the packet order does not depend on the block schedule. */
int lsk_tile_wv_max_parts(void);
/* gd.entries != NULL: pre-indexed packets -- every packet's local index at its destination comes out of the all-destinations
 * directory (own-partition packets included: ix is not used), and the send segments hold u32 indices instead of u64 states */
int lsk_tile_wv(lsk_operator op, lsk_basis bs, lsk_index ix, lsk_gdir gd, int cplx, int count_only, int P, int me,
                int64_t row0, int64_t row1, uint64_t const *reps, double const *norms, void const *x, void *y,
                uint32_t *d_wtab, lsk_round_layout const *d_layout, void *d_send, int *d_err, lsk_gtab own_gt, void *stream);
/* Packets in SORTED STREAMS (k_packets.hip, k_tile_st / k_window): unprojected fixed-weight bases, operators whose off-diagonal
 * groups are all exchange pairs.  stream = 2 * group + (bit of alpha at the pair's lower site); along a stream beta - alpha is a
 * constant, so the packets of one (destination, stream) -- written in row order -- carry ASCENDING destination indices, and the
 * consumer adds a window of y at a time in LDS instead of one fabric atomic per packet.
 * lsk_tile_st: rows [row0, row1) of one partition; tile t = rows [row0 + t tile_rows, ...), walked by one wave.  count_only:
 * d_ttab[tile][P * S] <- packets of every (tile, class = destination * S + stream); otherwise d_ttab holds the position of the
 * tile's first packet of every class inside the destination's segment of *d_layout (u32 keys, then values), and every packet --
 * the own partition's too -- goes to d_send as (u32 index at the destination, value). */
int lsk_tile_st_max_classes(void);
int lsk_tile_st(lsk_operator op, lsk_gdir gd, uint64_t const *d_binom, int cplx, int count_only, int P, int S, int tile_rows,
                int64_t row0, int64_t row1, uint64_t const *reps, void const *x, uint32_t *d_ttab,
                lsk_round_layout const *d_layout, void *d_send, int *d_err, void *stream);
/* one (source -> destination) segment as the consumer sees it */
typedef struct lsk_wsrc {
    uint32_t const *keys; /* device: indices at the destination, ascending inside every stream */
    double const *vals;   /* device: the values (re, im interleaved for c128) */
    uint32_t const *soff; /* device [S + 1]: stream s = packets [soff[s], soff[s + 1]) of the segment */
} lsk_wsrc;
/* the destinations of one consumer launch (passed by value): y[d] has count[d] rows and is served by the blocks
 * [first_block[d], first_block[d + 1]), wpb windows of lsk_window_rows(cplx) rows per block */
typedef struct lsk_wdests {
    int n;
    int64_t count[LSK_MAX_SEGS];
    int64_t first_block[LSK_MAX_SEGS + 1];
    void *y[LSK_MAX_SEGS];
    /* GLOBAL-RANK keys (round 6; dir[d] != NULL): a key is the colex rank of beta among ALL states of the weight, not its index at
     * the destination -- the producer needs no all-destinations directory (P / 4 bytes per global state on every rank), the
     * consumer turns a key into a row of y[d] with destination d's OWN rank directory (lsk_rankdir: 1 / 4 byte per global state,
     * whatever P) and finds its windows by the ranks of their first rows */
    lsk_rankdir const *dir[LSK_MAX_SEGS];
    uint64_t const *reps[LSK_MAX_SEGS]; /* the destination's ascending representatives (window bounds) */
    uint64_t const *binom;              /* device [64 * LSK_BINOM_K] */
    int64_t n_ranks;
    int weight;
    int *err;                           /* raised when a key is not a state of its destination (DMV:115-118) */
} lsk_wdests;
int lsk_window_rows(int cplx);
/* y[d][key] += value over all packets of the n_src source segments of every destination d (d_srcs: device [dests->n][n_src]) */
int lsk_window(int cplx, lsk_wdests const *dests, lsk_wsrc const *d_srcs, int n_src, int S, int wpb, void *stream);
/* replicated-x pull (Hermitian operators): rows of ONE partition against the whole vector in global
 * ascending order.  ix_global indexes the global basis; row_gidx[i] = global index of local row i
 * (may be NULL for lsk_direct_gx with closed-form indices, and for lsk_tile_pull when local == global). */
int lsk_direct_gx(lsk_operator op, lsk_basis bs, lsk_index ix_global, int cplx, lsk_tilemap tm,
                  uint64_t const *reps, int64_t const *row_gidx, void const *x_global, void *y, int *d_err,
                  void *stream);
/* hash table {rep -> x * norm(rep)} of the staged pull kernel: 2^bits entries of 16 (f64) / 32 (c128)
 * bytes; build once (keys + slot_of[i]), fill the values every matvec (norms NULL: unscaled) */
int lsk_hash_build(int cplx, int64_t n, uint64_t const *reps, int bits, void *tab, uint32_t *slot_of, void *stream);
int lsk_hash_fill(int cplx, int64_t n, uint32_t const *slot_of, void const *x, double const *norms, void *tab,
                  void *xs /* NULL, or the same scaled values in index order (near window of lsk_tile_pull) */, void *stream);
/* halo > 0: partners within `halo` (<= 512) entries of the tile in the sorted global representatives are resolved in an
 * LDS window and read from xs_global (= what the table holds, in index order); halo == 0: every partner through the table */
int lsk_tile_pull(lsk_operator op, lsk_basis bs, lsk_index ix_global, int cplx, int64_t row0, int64_t row1,
                  uint64_t const *reps, double const *norms_local, double const *norms_global,
                  int64_t const *row_gidx, void const *tab, int tab_bits, uint64_t const *reps_global,
                  int64_t n_global, void const *xs_global, int halo, void const *x_global, void *y,
                  int *d_err, void *stream);
/* Static index table {representative -> 32-bit payload} of the staged pull kernel's INDEXED mode (lsk_tile_pull_idx): built
 * once per basis, never refreshed.  Open addressing over 16-byte buckets of two 8-byte entries (one 16-byte request per
 * probe).  The key itself is not stored: h = an L-bit bijection of the state, bucket = its top `bbits` bits, and the entry
 * keeps the other L - bbits bits (tag), its displacement from the home bucket (8 bits) and the payload:
 *     entry = tag << 40 | displacement << 32 | payload,   empty = ~0
 * so 2^(bbits + 1) entries of 8 bytes serve n <= 2^bbits keys at load <= 0.5 (half the bytes of a {key, value} table, and
 * nothing to rewrite per matvec).  Needs tbits = L - bbits <= 24 and payloads < 2^32 - 1. */
/* choose bbits for n keys of L bits (returns -1 when no admissible size exists below max_bytes) */
int lsk_gtab_bits(int L, int64_t n, int64_t max_bytes);
/* fills entries (device, 16 << bbits bytes, allocated by the caller) with reps[i] -> payload[i] (payload NULL: i);
 * *d_flag is raised when a key cannot be placed within 255 buckets of its home */
int lsk_gtab_build(lsk_gtab t, uint64_t *entries, int64_t n, uint64_t const *reps, uint32_t const *payload, int *d_flag,
                   void *stream);
/* host-side mirror of the device lookup on a host copy of the entries (tests): payload or -1 */
int64_t lsk_test_gtab_find(lsk_gtab t, uint64_t const *h_entries, uint64_t key);
int lsk_test_gtab_build_host(lsk_gtab t, uint64_t *h_entries, int64_t n, uint64_t const *reps, uint32_t const *payload);
/* out[i] = lookup(keys[i]) or 0xffffffff (tests / plan-time checks) */
int lsk_gtab_lookup(lsk_gtab t, int64_t n, uint64_t const *keys, uint32_t *out, void *stream);

/* INDEXED mode of the staged pull kernel: the table yields a SLOT and the value is xsrc[slot] -- xsrc is x (times norm(rep) in
 * the K4 modes that prescale) in whatever order the caller holds it: index order on one device (perm == NULL, slot = global
 * index), or the received blocks of the replicated-x exchange as they arrive, owner-major (perm[g] = slot of global row g).
 * Rows are the contiguous global rows [row_g0 + row0, row_g0 + row1). */
typedef struct lsk_pullidx {
    lsk_gtab tab;
    uint32_t const *perm; /* device [n_global] or NULL */
    int64_t row_g0;
    uint64_t const *vtab; /* NULL, or the VALUE table of the same shape (k_pull.hip, lsk_vtab_*): 32-byte buckets {entry0, entry1, x[slot0],
                           * x[slot1]} -- one fabric request per far partner; f64 vectors of one partition, refreshed per matvec */
} lsk_pullidx;
/* value table: 32 << bbits bytes (allocated by the caller) copied from the index table `t`; refresh = x[slot] of every entry, in table order */
int lsk_vtab_build(lsk_gtab t, uint64_t *vtab, void *stream);
int lsk_vtab_refresh(lsk_gtab t, uint64_t *vtab, void const *xsrc, void *stream);
int lsk_tile_pull_idx(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                      double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global,
                      void const *xsrc, int halo, void *y, int *d_err, void *stream);
int lsk_pull_max_halo(void); /* largest near window (entries either side of a tile) the LDS hash set of the kernels takes */
/* The same matvec in TWO kernels, for the replicated-x exchange (dist.c): RESOLVE = stage A + K4 + slot look-ups, everything
 * that does not need x, writes one 4-byte slot per packet (+ its row inside the wave, + its coefficient unless the operator
 * has one real amplitude) to a per-wave packet stream; GATHER streams that, reads xsrc[slot] and writes y.  RESOLVE runs
 * while the blocks of x are on the wire.  Stream w belongs to rows [row0 + 64 w, row0 + 64 w + 64): `cap` packets of room
 * each (lsk_pullbuf_cap: 64 x number of flip-mask groups, the most 64 rows can generate), counts[w] of them valid. */
typedef struct lsk_pullbuf {
    uint32_t *slots;  /* device [streams * cap]; 0xffffffff = dead packet (zero-norm orbit) */
    uint8_t *rows;    /* device [streams * cap] */
    double *coefs;    /* device [streams * cap * lsk_pullbuf_coef_doubles] or NULL */
    uint32_t *counts; /* device [streams] */
    int64_t cap, row0;
    int64_t const *offs; /* NULL: stream w starts at w * cap; else device [streams + 1] exact offsets (lsk_tile_pull_stream_offsets) */
} lsk_pullbuf;
int lsk_tile_pull_stream_offsets(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps, int64_t *out,
                                 void *stream);
int64_t lsk_pullbuf_cap(lsk_operator op);
int lsk_pullbuf_coef_doubles(lsk_operator op, lsk_basis bs);
int lsk_tile_pull_resolve(lsk_operator op, lsk_basis bs, int64_t row0, int64_t row1, uint64_t const *reps,
                          double const *norms_local, lsk_pullidx ix, uint64_t const *reps_global, int64_t n_global, int halo,
                          lsk_pullbuf buf, int *d_err, void *stream);
int lsk_tile_pull_gather(lsk_operator op, lsk_basis bs, int cplx, int64_t row0, int64_t row1, uint64_t const *reps,
                         double const *norms_local, lsk_pullidx ix, void const *xsrc, lsk_pullbuf buf, void *y, void *stream);
/* out[i] = x[i] * norms[i] (f64 / c128): the owner-side prescaling of the indexed mode */
int lsk_scale(int cplx, int64_t n, void const *x, double const *norms, void *out, void *stream);
/* dst[perm[g] - base] = src[g] for every g with base <= perm[g] < base + count (8-byte elements): the rows one owner holds,
 * out of the global array, in its own (ascending) order */
int lsk_scatter_owned(int64_t n, uint32_t const *perm, int64_t base, int64_t count, uint64_t const *src, uint64_t *dst,
                      void *stream);

/* host-side check of the window search of the value-table kernel (tests, no device): position of `key` among the ascending
 * reps[0, n), n <= 1280, or -1 */
int lsk_test_window_find(uint64_t const *reps, int n, uint64_t key);
/* ... and of the LDS hash set of the indexed kernels (n <= 1024): position, or -1 when the window does not answer (absent,
 * or dropped from a full set -- such a partner goes through the static index table) */
int lsk_test_nw_find(uint64_t const *reps, int n, uint64_t key);
/* block -> tile of the pull kernels and of k_scatter (C consecutive tiles per XCD): host mirror for the tests */
int64_t lsk_test_pull_tile_of_block(int64_t b, int64_t n_tiles, int C);
int lsk_test_chain_near_table(int elem, int ldsp, int16_t *out);
uint64_t lsk_test_rep_trivial_dihedral(uint64_t a, int L, int inv, int reflect);
/* K4 mode 4: the row table of torus_min (out[2^tw], tw <= 8) and the minimum of v (and, with inv, of its complement) over the
 * translations of the tw x (L / tw) torus, starting from `best` (host mirrors of the device code) */
int lsk_torus_rowtab(int tw, uint32_t *out);
int lsk_torus_rowtab2(int tw, uint64_t *out); /* mode 5: rowtab[r] | rowtab[reversed r] << 32 */
uint64_t lsk_test_torus_min_d2(uint64_t v, int L, int tw, int inv, int present, uint64_t const *rowtab2, uint64_t best);
uint64_t lsk_test_torus_min(uint64_t v, int L, int tw, int inv, uint32_t const *rowtab, uint64_t best);
int lsk_bench_k4(int L, int inv, int reflect, int variant, int64_t n, uint64_t const *reps, uint64_t *out, void *stream);
/* n packets -> y[idx(beta)] += value */
int lsk_scatter(lsk_index ix, int cplx, int64_t n, uint64_t const *betas, void const *vals, void *y,
                double const *norms /* NULL, or per-row norms multiplied in (K4 modes 1, 2) */, int *d_err,
                void *stream);

/* plan-time helpers -------------------------------------------------------------------------- */
/* norms[i] = sqrt(stab(reps[i]) / |G|) */
int lsk_norms(lsk_basis bs, int64_t n, uint64_t const *reps, double *norms, void *stream);
/* *d_flag = 1 unless reps[i] == unrank(i) for all i (full fixed-Hamming prefix) */
int lsk_check_combinadic(lsk_index ix, int hamming_weight, int64_t n, uint64_t const *reps,
                         int *d_flag, void *stream);
/* SEARCH prefix table: table[b] = lower_bound(reps, b << shift), b in [0, nbuckets] */
int lsk_build_table(int64_t n, uint64_t const *reps, int shift, int64_t nbuckets, uint32_t *table,
                    void *stream);

/* batched externs (device pointers) ---------------------------------------------------------- */
int lsk_state_info(lsk_basis bs, int64_t n, uint64_t const *alphas, uint64_t *betas,
                   double *characters /* (re, im) */, double *norms, void *stream);
int lsk_state_index(lsk_index ix, int64_t n, uint64_t const *spins, int64_t *indices, void *stream);
/* counts[i] = number of non-zero off-diagonal groups of alphas[i] */
int lsk_offdiag_counts(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t *counts,
                       void *stream);
/* writes (beta, c [* xs[i]]) of row i at offsets[i]...; xs may be NULL */
int lsk_offdiag_fill(lsk_operator op, int64_t n, uint64_t const *alphas, int64_t const *offsets,
                     uint64_t *betas, double *coeffs /* (re, im) */, double const *xs, void *stream);
/* ys[i] = d(alphas[i]) [* xs[i]] (real part) */
int lsk_diag_coeffs(lsk_operator op, int64_t n, uint64_t const *alphas, double *ys, double const *xs,
                    void *stream);
int lsk_exclusive_scan_i64(int64_t n, int64_t const *in, int64_t *out, void *stream);

/* enumeration + layout ------------------------------------------------------------------------ */
/* number of candidate states (fixed-Hamming rank space or 2^L), given the top-bit cut */
int lsk_enumerate(lsk_basis bs, uint64_t const *d_binom, int64_t n_candidates, uint64_t **d_states,
                  int64_t *count, void *stream);
int lsk_masks(int64_t n, uint64_t const *states, int P, uint8_t *masks, void *stream);
int lsk_mask_counts(int64_t n, uint8_t const *masks, int P, int64_t *h_counts, void *stream);
int lsk_block_to_hashed(int64_t n, uint8_t const *masks, int P, int elt_size, void const *src,
                        void *const *h_dest, void *stream);
int lsk_hashed_to_block(int64_t n, uint8_t const *masks, int P, int elt_size,
                        void const *const *h_src, void *dest, void *stream);

/* out[i] = src[perm[i]], elements of 8 or 16 bytes, perm int32 or int64 */
int lsk_gather_perm(int64_t n, void const *perm, int perm_is_64, int elt_size, void const *src, void *out, void *stream);
/* repl.hip: the same over up to 64 index ranges [lo_k, hi_k) of perm / out in ONE launch (the rows of one chunk of y, grouped by
 * owner: one range per owner), and the event helpers of the chunked return / the adaptive split */
typedef struct lsk_ranges { int n; int64_t lo[64], hi[64]; } lsk_ranges;
int lsk_gather_perm_ranges(lsk_ranges const *ranges, void const *perm, int perm_is_64, int elt_size, void const *src, void *out, void *stream);
int lsk_event_query(void *ev);                      /* 0 complete, 1 not yet, -1 error */
int lsk_stream_wait_event(void *stream, void *ev);  /* work queued on `stream` after this call waits for `ev` */
char *lsk_error_buffer(size_t *capacity);           /* k_runtime.hip: the thread's error message buffer (lsk_last_error) */
/* bitmap bit (index of partner >> shift) <- 1 for every off-diagonal partner of the rows alphas[0, n) (u32 words, device): the
 * blocks of 2^shift rows of the global vector those rows read */
int lsk_reach_blocks(lsk_operator op, lsk_index ix_global, int64_t n, uint64_t const *alphas, int shift, uint32_t *bitmap, void *stream);

int lsk_iota_i64(int64_t n, int64_t base, int64_t *out, void *stream);
int lsk_narrow_i32(int64_t n, int64_t const *in, int32_t *out, void *stream);
int lsk_add_into(int cplx, int64_t n, void const *a, void *y, void *stream); /* y += a */

/* RCCL (comm.cpp) ------------------------------------------------------------------------------ */
typedef struct lsk_comm lsk_comm;
char const *lsk_comm_last_error(void);
int lsk_comm_available(void);
int lsk_comm_unique_id(void *id128);
int lsk_comm_create(lsk_comm **out, int size, int rank, void const *id128);
int lsk_comm_create_local(lsk_comm **out /* [size] */, int size); /* loop-back group: one process, one device, one thread per rank */
void lsk_comm_destroy(lsk_comm *c);
int lsk_comm_size(lsk_comm const *c);
int lsk_comm_rank(lsk_comm const *c);
int lsk_comm_rccl_count(lsk_comm const *c); /* ncclCommCount; 0 = loop-back group, -1 = error */
int lsk_comm_allreduce(lsk_comm *c, void *d_buf, int64_t count, int dtype /* 0 f64, 1 f32, 2 i64 */, int op /* 0 sum, 1 max */,
                       void *stream);
int lsk_comm_broadcast(lsk_comm *c, void *d_buf, int64_t bytes, int root, void *stream);
int lsk_comm_allgather(lsk_comm *c, void const *d_send, void *d_recv, int64_t bytes_per_rank, void *stream);
/* double-buffered all-to-all-v on the exchange stream (see comm.cpp) */
int lsk_comm_exchange_begin(lsk_comm *c, int slot, void *compute_stream);
int lsk_comm_alltoallv(lsk_comm *c, void const *d_send, int64_t const *send_off, int64_t const *send_bytes, void *d_recv,
                       int64_t const *recv_off, int64_t const *recv_bytes);
int lsk_comm_exchange_end(lsk_comm *c, int slot);
/* the same grouped send/recv on a stream of the caller's (no second stream, no events) */
/* K segments per peer in one grouped exchange: entry [k * size + p] = segment k for / from peer p */
int lsk_comm_alltoallv_multi_on(lsk_comm *c, void *stream, int K, void const *d_send, int64_t const *send_off,
                                int64_t const *send_bytes, void *d_recv, int64_t const *recv_off, int64_t const *recv_bytes);
int lsk_comm_alltoallv_on(lsk_comm *c, void *stream, void const *d_send, int64_t const *send_off, int64_t const *send_bytes,
                          void *d_recv, int64_t const *recv_off, int64_t const *recv_bytes);
int lsk_comm_exchange_wait(lsk_comm *c, int slot, void *compute_stream);
/* never hang (comm.cpp): what the next exchange is, for the watchdog's message; a deadline wait on `stream` + the exchange stream
 * (rc -1 with a message instead of a hung hipStreamSynchronize); the collective set-up cross-check of an exchange layout
 * ([K][size] byte counts: what s sends to d == what d expects from s, on every rank alike); test hook: stall the exchange stream */
int lsk_comm_exchange_ms(lsk_comm *c, int slot, float *ms); /* duration of the slot's last completed exchange; 1 = none yet; never blocks */
void lsk_comm_set_tag(lsk_comm *c, char const *what);
int lsk_comm_wait(lsk_comm *c, void *stream, double timeout_s);
int lsk_comm_check_counts(lsk_comm *c, int K, int64_t const *send_bytes, int64_t const *recv_bytes, char const *what, void *stream);
int lsk_comm_test_stall(lsk_comm *c, double seconds);

#ifdef __cplusplus
}
#endif
#endif /* LSK_H */
