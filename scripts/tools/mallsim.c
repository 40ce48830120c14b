/* mallsim.c -- offline model of the whole memory hierarchy seen by the staged row kernel (k_chain) on
 * heisenberg_chain_L: 8 XCDs with a 4 MiB L2 each (128-byte lines, 16-way LRU) in front of one shared 256 MiB
 * memory-side cache (Infinity Cache; modelled 16-way LRU, 128-byte lines).  Reports L2 misses (what FETCH_SIZE
 * counts) and Infinity-Cache misses (HBM reads) per row for two ways of dealing tiles to XCDs:
 *   chunk = 0: XCD k gets the k-th contiguous eighth of the tiles (the default tile map);
 *   chunk = G: tiles are dealt round-robin in chunks of G consecutive tiles, so all XCDs advance through the
 *              same region of the vector together.
 *   top_bits = t > 0: the tiles are first put into the global set order (all segments of a popcount class of
 *              the top t bits x a window of set_rows / #segments rows, see build_tilemap in host.c), then dealt
 *              in chunks as above: the whole chip works on one set at a time.
 * Same access model as l2sim.c.  usage: mallsim L chunk tiles_per_xcd [blocks=224] [top_bits=0] [set_rows=0]
 * Exploration tool (not part of the product); cc -O2 -o mallsim mallsim.c */
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
enum { WAYS = 16, LINE = 128 };
typedef struct { int64_t sets; uint64_t *tags; uint64_t hits, misses; } cache_t;
static void cache_init(cache_t *c, int64_t bytes) { c->sets = bytes / LINE / WAYS; c->tags = calloc((size_t)c->sets * WAYS, 8); c->hits = c->misses = 0; }
static int cache_touch(cache_t *c, uint64_t line) {
    uint64_t set = (line * 0x9E3779B97F4A7C15ULL >> 20) % (uint64_t)c->sets, *w = c->tags + set * WAYS, tag = line + 1;
    int i;
    for (i = 0; i < WAYS; ++i) if (w[i] == tag) break;
    int hit = i < WAYS;
    if (hit) ++c->hits; else { ++c->misses; i = WAYS - 1; }
    memmove(w + 1, w, 8 * (size_t)i);
    w[0] = tag;
    return hit;
}
static cache_t l2[8], mall;
static int cur;
static void touch(uint64_t byte_addr) {
    uint64_t line = byte_addr / LINE;
    if (!cache_touch(&l2[cur], line)) cache_touch(&mall, line);
}
static void touch_range(uint64_t base, uint64_t first, uint64_t count, int elt) {
    uint64_t a0 = base + first * elt, a1 = base + (first + count) * elt - 1;
    for (uint64_t l = a0 / LINE; l <= a1 / LINE; ++l) touch(l * LINE);
}

int main(int argc, char **argv) {
    int L = argc > 1 ? atoi(argv[1]) : 32;
    int64_t chunk = argc > 2 ? atoll(argv[2]) : 0, per_xcd = argc > 3 ? atoll(argv[3]) : 20000;
    int const BLOCKS = argc > 4 ? atoi(argv[4]) : 224, hw = L / 2, TILE = 1024, HALO = 512;
    binom_init();
    uint64_t const n = C[L][hw];
    int64_t const tiles = (int64_t)((n + TILE - 1) / TILE);
    for (int k = 0; k < 8; ++k) cache_init(&l2[k], 4 << 20);
    cache_init(&mall, 256 << 20);
    uint64_t const X = 0, REPS = (uint64_t)1 << 40, CACHE = (uint64_t)2 << 40, Y = (uint64_t)3 << 40;
    /* slot s of XCD k -> tile index; the simulated window starts in the middle of every XCD's list */
    int const t = argc > 5 ? atoi(argv[5]) : 0;
    int64_t const set_rows = argc > 6 ? atoll(argv[6]) : 0;
    uint64_t *order = NULL; /* order[q] = first row of the q-th tile in the global order; cnts[q] its rows */
    int *cnts = NULL;
    int64_t n_order = 0;
    if (t > 0) {
        int Lr = L - t, nT = 1 << t;
        int64_t *base = malloc(8 * (nT + 1)), acc = 0, cap = 0;
        for (int T = 0; T < nT; ++T) { base[T] = acc; int j = __builtin_popcount(T); acc += (hw - j >= 0 && hw - j <= Lr) ? (int64_t)C[Lr][hw - j] : 0; }
        int *segs = malloc(4 * nT);
        for (int j = 0; j <= t; ++j) {
            if (hw - j < 0 || hw - j > Lr) continue;
            int64_t len = (int64_t)C[Lr][hw - j];
            if (!len) continue;
            int nseg = 0;
            for (int T = 0; T < nT; ++T) if (__builtin_popcount(T) == j) segs[nseg++] = T;
            int64_t W = TILE;
            while (W * 2 * nseg <= set_rows) W *= 2;
            for (int64_t w0 = 0; w0 < len; w0 += W)
                for (int sg = 0; sg < nseg; ++sg) /* segment-major inside a set: runs of consecutive tiles */
                    for (int64_t off = w0; off < w0 + W && off < len; off += TILE) {
                        if (n_order == cap) { cap = cap ? 2 * cap : 1 << 16; order = realloc(order, 8 * cap); cnts = realloc(cnts, 4 * cap); }
                        order[n_order] = (uint64_t)(base[segs[sg]] + off);
                        cnts[n_order] = (int)(len - off < TILE ? len - off : TILE);
                        ++n_order;
                    }
        }
    }
    int64_t const total_tiles = t > 0 ? n_order : tiles;
    int64_t const list_len = total_tiles / 8, s0 = list_len / 2;
    uint64_t rows_done = 0, warm_rows = 0, warm_l2 = 0, warm_mall = 0;
    uint64_t *st = malloc(8 * TILE);
    int64_t rounds = (per_xcd + BLOCKS - 1) / BLOCKS;
    for (int64_t rd = 0; rd < rounds; ++rd) {
        if (rd == rounds / 2) {
            warm_rows = rows_done; warm_mall = mall.misses; warm_l2 = 0;
            for (int k = 0; k < 8; ++k) warm_l2 += l2[k].misses;
        }
        for (int sub = -1; sub < 4; ++sub)
            for (int b = 0; b < BLOCKS; ++b)
                for (cur = 0; cur < 8; ++cur) {
                    int64_t s = s0 + rd * BLOCKS + b;
                    if (s >= s0 + per_xcd || s >= list_len) continue;
                    int64_t q = chunk == 0 ? cur * list_len + s : ((s / chunk) * 8 + cur) * chunk + s % chunk;
                    if (q >= total_tiles) continue;
                    uint64_t row = t > 0 ? order[q] : (uint64_t)q * TILE;
                    int cnt = t > 0 ? cnts[q] : (n - row < (uint64_t)TILE ? (int)(n - row) : TILE);
                    if (sub < 0) {
                        int64_t w0 = (int64_t)row - HALO;
                        if (w0 < 0) w0 = 0;
                        uint64_t w1 = row + cnt + HALO;
                        if (w1 > n) w1 = n;
                        touch_range(X, (uint64_t)w0, w1 - (uint64_t)w0, 8);
                        continue;
                    }
                    int r0 = sub * 256, r1 = r0 + 256;
                    if (r0 >= cnt) continue;
                    if (r1 > cnt) r1 = cnt;
                    touch_range(REPS, row + r0, r1 - r0, 8);
                    touch_range(CACHE, row + r0, r1 - r0, 4);
                    touch_range(Y, row + r0, r1 - r0, 8);
                    uint64_t s_ = unrank(row + r0, L, hw);
                    for (int r = r0; r < r1; ++r) { st[r] = s_; s_ = next_state(s_); }
                    for (int w = r0; w < r1; w += 64) {
                        int we = w + 64 < r1 ? w + 64 : r1;
                        uint64_t a0 = st[w];
                        int uni = 1;
                        for (int r = w; r < we; ++r) if ((st[r] ^ a0) >> 12) { uni = 0; break; }
                        for (int p = 12; p < L - 1; ++p) {
                            if (uni) {
                                if ((((a0 >> p) ^ (a0 >> (p + 1))) & 1) == 0) continue;
                                int kk = hw - __builtin_popcountll(a0 >> p);
                                uint64_t d = C[p][kk], i = row + w;
                                touch_range(X, ((a0 >> p) & 1) ? i + d : i - d, (uint64_t)(we - w), 8);
                            } else
                                for (int r = w; r < we; ++r) {
                                    uint64_t a = st[r];
                                    if ((((a >> p) ^ (a >> (p + 1))) & 1) == 0) continue;
                                    int kk = __builtin_popcountll(a & ((1ULL << p) - 1));
                                    uint64_t d = C[p][kk], i = row + r;
                                    touch(X + 8 * (((a >> p) & 1) ? i + d : i - d));
                                }
                        }
                        for (int r = w; r < we; ++r) {
                            uint64_t a = st[r];
                            if ((((a >> (L - 1)) ^ a) & 1) == 0) continue;
                            touch(X + 8 * rank_of(a ^ ((1ULL << (L - 1)) | 1ULL)));
                        }
                    }
                    rows_done += (uint64_t)(r1 - r0);
                }
    }
    uint64_t l2m = 0;
    for (int k = 0; k < 8; ++k) l2m += l2[k].misses;
    double rows = (double)(rows_done - warm_rows);
    printf("L=%d chunk=%lld t=%d set_rows=%lld tiles/xcd=%lld: rows %llu; second half: L2 misses/row %.3f (%.1f B/row), Infinity-Cache misses/row %.3f (%.1f B/row)\n",
           L, (long long)chunk, t, (long long)set_rows, (long long)per_xcd, (unsigned long long)rows_done, (double)(l2m - warm_l2) / rows,
           128.0 * (double)(l2m - warm_l2) / rows, (double)(mall.misses - warm_mall) / rows, 128.0 * (double)(mall.misses - warm_mall) / rows);
    return 0;
}
