/* l2sim.c -- offline model of one XCD's L2 under the staged row kernel (k_chain) on heisenberg_chain_L:
 * predicts L2 misses per row for a given tile order (default: contiguous eighth; transposed: sets closed under
 * the top flips, see build_tilemap in host.c).  Model: 4 MiB, 128-byte lines, 16-way LRU; `BLOCKS` resident blocks
 * walk the XCD's tile list round-robin, interleaved per 256-row sub-tile.  Accesses per 1024-row tile: the staged
 * x window, states (8 B/row), cached ring partners (4 B/row), y (8 B/row), per wave the 64-row x slices of its
 * anti-aligned far pairs (lo >= 12) and the ring-bond partners.
 *   usage: l2sim L transposed top_bits set_rows max_tiles [blocks=224] [bypass_streams=0]
 *   bypass_streams=1: states / cached partners / y do not allocate in the L2 (what-if)
 * Exploration tool (not part of the product); cc -O2 -o l2sim l2sim.c */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t C[65][65];
static void binom_init(void) {
    for (int n = 0; n <= 64; ++n) {
        C[n][0] = 1;
        for (int k = 1; k <= 64; ++k) C[n][k] = n == 0 ? 0 : C[n - 1][k - 1] + C[n - 1][k];
    }
}
static uint64_t rank_of(uint64_t s) {
    uint64_t r = 0;
    int j = 1;
    while (s) { int p = __builtin_ctzll(s); r += C[p][j]; ++j; s &= s - 1; }
    return r;
}
static uint64_t unrank(uint64_t r, int L, int k) {
    uint64_t s = 0;
    for (int p = L - 1; p >= 0 && k > 0; --p)
        if (r >= C[p][k]) { s |= 1ULL << p; r -= C[p][k]; --k; }
    return s;
}
static uint64_t next_state(uint64_t v) {
    uint64_t t = v | (v - 1);
    return (t + 1) | (((~t & -~t) - 1) >> (__builtin_ctzll(v) + 1));
}

/* ---- cache ---- */
enum { WAYS = 16, LINE = 128 };
static int64_t n_sets;
static uint64_t *tags; /* [n_sets][WAYS], most recent first; 0 = empty (tags stored +1) */
static uint64_t hits, misses;
static uint64_t pair_hits[64], pair_misses[64];
static int cur_pair = 63;
static void touch(uint64_t byte_addr) {
    uint64_t line = byte_addr / LINE;
    uint64_t set = (line * 0x9E3779B97F4A7C15ULL >> 20) % (uint64_t)n_sets;
    uint64_t *w = tags + set * WAYS, tag = line + 1;
    int i;
    for (i = 0; i < WAYS; ++i) if (w[i] == tag) break;
    if (i < WAYS) { ++hits; ++pair_hits[cur_pair]; } else { ++misses; ++pair_misses[cur_pair]; i = WAYS - 1; }
    memmove(w + 1, w, sizeof(uint64_t) * (size_t)i);
    w[0] = tag;
}
static void touch_range(uint64_t base, uint64_t first_elem, uint64_t count, int elt) {
    uint64_t a0 = base + first_elem * elt, a1 = base + (first_elem + count) * elt - 1;
    for (uint64_t l = a0 / LINE; l <= a1 / LINE; ++l) touch(l * LINE);
}

typedef struct { uint64_t row; int cnt; } tile;

int main(int argc, char **argv) {
    int L = argc > 1 ? atoi(argv[1]) : 32, transposed = argc > 2 ? atoi(argv[2]) : 0, t = argc > 3 ? atoi(argv[3]) : 8;
    int64_t set_rows = argc > 4 ? atoll(argv[4]) : 65536, max_tiles = argc > 5 ? atoll(argv[5]) : 4000;
    int const BLOCKS = argc > 6 ? atoi(argv[6]) : 224, bypass = argc > 7 ? atoi(argv[7]) : 0;
    int const hw = L / 2, TILE = 1024, HALO = 512;
    binom_init();
    uint64_t const n = C[L][hw];
    /* ---- tile list of XCD 0 (same construction as tilemap_host) ---- */
    tile *list = NULL;
    int64_t nl = 0, cap = 0;
#define PUSH(r, c) do { if (nl == cap) { cap = cap ? 2 * cap : 4096; list = realloc(list, sizeof(tile) * cap); } list[nl].row = (r); list[nl].cnt = (c); ++nl; } while (0)
    if (!transposed) {
        int64_t tiles = (n + TILE - 1) / TILE;
        for (int64_t q = 0; q < tiles / 8; ++q) PUSH((uint64_t)q * TILE, TILE);
    } else {
        int Lr = L - t, nT = 1 << t;
        int64_t *base = malloc(sizeof(int64_t) * (nT + 1)), acc = 0, rows_of[8] = {0};
        for (int T = 0; T < nT; ++T) { base[T] = acc; int j = __builtin_popcount(T); acc += (hw - j >= 0 && hw - j <= Lr) ? (int64_t)C[Lr][hw - j] : 0; }
        int *segs = malloc(sizeof(int) * nT);
        for (int j = 0; j <= t; ++j) {
            if (hw - j < 0 || hw - j > Lr) continue;
            int64_t len = (int64_t)C[Lr][hw - j];
            if (!len) continue;
            int nseg = 0;
            for (int T = 0; T < nT; ++T) if (__builtin_popcount(T) == j) segs[nseg++] = T;
            int64_t W = TILE;
            while (W * 2 * nseg <= set_rows) W *= 2;
            for (int64_t w0 = 0; w0 < len; w0 += W) {
                int k = 0;
                for (int q = 1; q < 8; ++q) if (rows_of[q] < rows_of[k]) k = q;
                for (int64_t off = w0; off < w0 + W && off < len; off += TILE)
                    for (int s = 0; s < nseg; ++s) {
                        int64_t c = len - off < TILE ? len - off : TILE;
                        if (k == 0) PUSH((uint64_t)(base[segs[s]] + off), (int)c);
                        rows_of[k] += c;
                    }
            }
        }
    }
    /* simulate a window from the middle of the list (the ends are atypical) */
    int64_t start = nl / 2;
    if (start + max_tiles > nl) start = nl > max_tiles ? nl - max_tiles : 0;
    int64_t ntiles = nl - start < max_tiles ? nl - start : max_tiles;
    n_sets = (4 << 20) / LINE / WAYS;
    tags = calloc((size_t)n_sets * WAYS, sizeof(uint64_t));
    uint64_t const X = 0, REPS = (uint64_t)1 << 40, CACHE = (uint64_t)2 << 40, Y = (uint64_t)3 << 40;
    uint64_t rows_done = 0, warm_rows = 0, warm_misses = 0, stream_lines = 0;
    /* per block: current tile and sub-tile; states are generated on the fly */
    int64_t rounds = (ntiles + BLOCKS - 1) / BLOCKS;
    uint64_t *st = malloc(sizeof(uint64_t) * TILE);
    for (int64_t rd = 0; rd < rounds; ++rd) {
        if (rd == rounds / 3) { warm_rows = rows_done; warm_misses = misses; } /* skip the cold start */
        for (int sub = -1; sub < 4; ++sub) {
            for (int b = 0; b < BLOCKS; ++b) {
                int64_t ti = start + rd * BLOCKS + b;
                if (ti >= start + ntiles) continue;
                tile T = list[ti];
                if (sub < 0) { /* staging */
                    int64_t w0 = (int64_t)T.row - HALO;
                    if (w0 < 0) w0 = 0;
                    uint64_t w1 = T.row + T.cnt + HALO;
                    if (w1 > n) w1 = n;
                    touch_range(X, (uint64_t)w0, w1 - (uint64_t)w0, 8);
                    continue;
                }
                int r0 = sub * 256, r1 = r0 + 256;
                if (r0 >= T.cnt) continue;
                if (r1 > T.cnt) r1 = T.cnt;
                if (!bypass) {
                    touch_range(REPS, T.row + r0, r1 - r0, 8);
                    touch_range(CACHE, T.row + r0, r1 - r0, 4);
                    touch_range(Y, T.row + r0, r1 - r0, 8);
                } else stream_lines += ((uint64_t)(r1 - r0) * 20 + LINE - 1) / LINE;
                uint64_t s = unrank(T.row + r0, L, hw);
                for (int r = r0; r < r1; ++r) { st[r] = s; s = next_state(s); }
                for (int w = r0; w < r1; w += 64) {
                    int we = w + 64 < r1 ? w + 64 : r1;
                    uint64_t a0 = st[w];
                    int uni = 1;
                    for (int r = w; r < we; ++r) if ((st[r] ^ a0) >> 12) { uni = 0; break; }
                    for (int p = 12; p < L - 1; ++p) {
                        cur_pair = p;
                        if (uni) {
                            if ((((a0 >> p) ^ (a0 >> (p + 1))) & 1) == 0) continue;
                            int kk = hw - __builtin_popcountll(a0 >> p);
                            uint64_t d = C[p][kk], i = T.row + w;
                            uint64_t idx = ((a0 >> p) & 1) ? i + d : i - d;
                            touch_range(X, idx, (uint64_t)(we - w), 8);
                        } else
                            for (int r = w; r < we; ++r) {
                                uint64_t a = st[r];
                                if ((((a >> p) ^ (a >> (p + 1))) & 1) == 0) continue;
                                int kk = __builtin_popcountll(a & ((1ULL << p) - 1));
                                uint64_t d = C[p][kk], i = T.row + r;
                                touch(X + 8 * (((a >> p) & 1) ? i + d : i - d));
                            }
                    }
                    cur_pair = 62;
                    for (int r = w; r < we; ++r) { /* ring bond */
                        uint64_t a = st[r];
                        if ((((a >> (L - 1)) ^ a) & 1) == 0) continue;
                        touch(X + 8 * rank_of(a ^ ((1ULL << (L - 1)) | 1ULL)));
                    }
                }
                cur_pair = 63;
                rows_done += (uint64_t)(r1 - r0);
            }
        }
    }
    misses += stream_lines; /* bypassed streams still cross the fabric once */
    printf("L=%d %s t=%d set_rows=%lld blocks=%d bypass=%d: %lld tiles, rows %llu, fabric lines/row %.3f, hit rate %.3f\n", L,
           transposed ? "transposed" : "default", t, (long long)set_rows, BLOCKS, bypass, (long long)ntiles,
           (unsigned long long)rows_done, (double)misses / rows_done, (double)hits / (double)(hits + misses));
    if (getenv("L2SIM_PAIRS")) {
        for (int p = 12; p < L - 1; ++p)
            printf("  pair %2d: lines/row %.4f  hit rate %.3f\n", p, (double)(pair_hits[p] + pair_misses[p]) / rows_done,
                   (double)pair_hits[p] / (double)(pair_hits[p] + pair_misses[p] + 1));
        printf("  ring   : lines/row %.4f  hit rate %.3f\n", (double)(pair_hits[62] + pair_misses[62]) / rows_done,
               (double)pair_hits[62] / (double)(pair_hits[62] + pair_misses[62] + 1));
        printf("  streams: lines/row %.4f  hit rate %.3f\n", (double)(pair_hits[63] + pair_misses[63]) / rows_done,
               (double)pair_hits[63] / (double)(pair_hits[63] + pair_misses[63] + 1));
    }
    return 0;
}
