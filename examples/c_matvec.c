/* examples/c_matvec.c -- a plain C caller of libls_amd.so: no Python, no torch, no HIP headers.
 *
 * What the reference's hosts do through the plug-in table (/root/reference/src/library.c:19-34,
 * /root/reference/src/DistributedMatrixVector.chpl:1095-1110), spelled out:
 *   ls_chpl_init -> a basis and an operator, either loaded from a YAML file of the reference's schema by the library itself
 *   (c_matvec model.yaml: ls_hs_load_yaml_config) or built from flat term arrays (c_matvec L) -> ls_hs_basis_build (the
 *   registered enumerate_states kernel)
 *   -> kernels->matrix_vector_product(op, 1, x, y) on host arrays,
 * then the same product through the device-level API with 3 hash partitions (ls_amd_matvec) and, when librccl is
 * there, through the one-locale-per-process path with a single rank (ls_amd_dist_matvec) -- all three must agree.
 *
 *   gcc -std=c11 -Iinclude examples/c_matvec.c -Ldistributed-matvec_amd -lls_amd -Wl,-rpath,$PWD/distributed-matvec_amd -lm -o c_matvec
 *
 * Model: periodic Heisenberg ring of L sites, sigma.sigma per bond, half filling (data/heisenberg_chain_L.yaml), written as
 * non-branching terms: zz part (diagonal): +1 * (-1)^{popcount(alpha & pair)};  xx+yy part: 2 on anti-aligned pairs. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ls_amd.h"
#include "ls_chpl.h"
#include "ls_hs.h"

#define CHECK(expr) do { if ((expr) != 0) { fprintf(stderr, "%s failed: %s\n", #expr, ls_amd_last_error()); return 1; } } while (0)

int main(int argc, char **argv) {
    char const *arg = argc > 1 ? argv[1] : "16";
    size_t const alen = strlen(arg);
    int const from_yaml = alen > 5 && strcmp(arg + alen - 5, ".yaml") == 0;
    int L = from_yaml ? 0 : atoi(arg);
    ls_chpl_init();
    ls_hs_basis *basis;
    ls_hs_operator *op;
    if (from_yaml) {
        /* loadConfigFromYaml (/root/reference/src/ForeignTypes.chpl:261-288): load, clone what is wanted, destroy the config */
        ls_hs_yaml_config *conf = ls_hs_load_yaml_config(arg);
        if (!conf) { fprintf(stderr, "failed to load Config from '%s': %s\n", arg, ls_amd_last_error()); return 1; }
        if (!conf->hamiltonian) { fprintf(stderr, "'%s' does not contain a Hamiltonian\n", arg); return 1; }
        op = ls_hs_clone_operator(conf->hamiltonian);
        ls_hs_destroy_yaml_config(conf);
        if (!op) { fprintf(stderr, "%s\n", ls_amd_last_error()); return 1; }
        basis = op->basis; /* the operator's own basis (what Operator.basis is in ForeignTypes.chpl): built below, released with op */
        L = basis->number_sites;
    } else {
        basis = ls_hs_create_spin_basis(L, L / 2, 0, 0, NULL, NULL);
        if (!basis) { fprintf(stderr, "%s\n", ls_amd_last_error()); return 1; }
        /* terms: per bond one diagonal term and two off-diagonal ones (01 -> 10 and 10 -> 01) */
        int const nt = 3 * L;
        double *v = calloc(2 * (size_t)nt, sizeof(double));
        uint64_t *m = calloc(nt, 8), *r = calloc(nt, 8), *x = calloc(nt, 8), *s = calloc(nt, 8);
        for (int b = 0; b < L; ++b) {
            uint64_t const i = 1ULL << b, j = 1ULL << ((b + 1) % L), pair = i | j;
            v[2 * (3 * b)] = 1.0; m[3 * b] = 0; r[3 * b] = 0; x[3 * b] = 0; s[3 * b] = pair;       /* zz */
            v[2 * (3 * b + 1)] = 2.0; m[3 * b + 1] = pair; r[3 * b + 1] = i; x[3 * b + 1] = pair;   /* |..1..0..> -> |..0..1..> */
            v[2 * (3 * b + 2)] = 2.0; m[3 * b + 2] = pair; r[3 * b + 2] = j; x[3 * b + 2] = pair;
        }
        op = ls_hs_create_operator_from_terms(basis, nt, v, m, r, x, s);
        free(v); free(m); free(r); free(x); free(s);
    }
    if (!op) { fprintf(stderr, "%s\n", ls_amd_last_error()); return 1; }
    if (!ls_hs_operator_is_hermitian(op) || !ls_hs_operator_is_real(op)) { fprintf(stderr, "operator flags\n"); return 1; }

    ls_hs_basis_build(basis); /* -> ls_chpl_enumerate_representatives -> HIP enumeration */
    int64_t const n = (int64_t)basis->representatives.num_elts;
    uint64_t const *reps = (uint64_t const *)basis->representatives.elts;
    double *hx = malloc(8 * (size_t)n), *hy = malloc(8 * (size_t)n), *hy2 = malloc(8 * (size_t)n), *hy3 = malloc(8 * (size_t)n);
    for (int64_t k = 0; k < n; ++k) { hx[k] = sin(0.37 * (double)k) + 0.1; hy[k] = 7.0; }

    /* 1. the plug-in entry, host arrays */
    ls_chpl_kernels const *kt = ls_hs_internal_get_chpl_kernels();
    ((void (*)(ls_hs_operator *, int, double *, double *))kt->matrix_vector_product)(op, 1, hx, hy);

    /* 2. device-level API, 3 hash partitions in this process */
    enum { P = 3 };
    int64_t counts[P] = {0, 0, 0};
    for (int64_t k = 0; k < n; ++k) counts[ls_amd_locale_idx_of(reps[k], P)]++;
    uint64_t *hr[P]; double *hxp[P];
    void *dr[P], *dx[P], *dy[P];
    int64_t fill[P] = {0, 0, 0};
    for (int p = 0; p < P; ++p) { hr[p] = malloc(8 * (size_t)(counts[p] + 1)); hxp[p] = malloc(8 * (size_t)(counts[p] + 1)); }
    for (int64_t k = 0; k < n; ++k) { int p = ls_amd_locale_idx_of(reps[k], P); hr[p][fill[p]] = reps[k]; hxp[p][fill[p]++] = hx[k]; }
    for (int p = 0; p < P; ++p) {
        CHECK(ls_amd_malloc(&dr[p], 8 * (size_t)(counts[p] + 1))); CHECK(ls_amd_malloc(&dx[p], 8 * (size_t)(counts[p] + 1)));
        CHECK(ls_amd_malloc(&dy[p], 8 * (size_t)(counts[p] + 1)));
        CHECK(ls_amd_memcpy_h2d(dr[p], hr[p], 8 * (size_t)counts[p])); CHECK(ls_amd_memcpy_h2d(dx[p], hxp[p], 8 * (size_t)counts[p]));
    }
    ls_amd_plan *plan;
    CHECK(ls_amd_plan_create(&plan, op, LS_AMD_F64, P, -1, (uint64_t const *const *)dr, counts, 0, LS_AMD_MODE_AUTO, NULL));
    CHECK(ls_amd_matvec(plan, (void const *const *)dx, dy, NULL));
    CHECK(ls_amd_plan_check(plan, NULL));
    memset(fill, 0, sizeof(fill));
    for (int p = 0; p < P; ++p) CHECK(ls_amd_memcpy_d2h(hxp[p], dy[p], 8 * (size_t)counts[p])); /* reuse hxp as y parts */
    for (int64_t k = 0; k < n; ++k) { int p = ls_amd_locale_idx_of(reps[k], P); hy2[k] = hxp[p][fill[p]++]; }
    printf("kernel (3 partitions): %s, nnz = %lld\n", ls_amd_plan_kernel_name(plan), (long long)ls_amd_plan_nnz(plan));
    ls_amd_plan_destroy(plan);

    /* 3. one locale per process, exchange inside the library (a communicator of one rank here) */
    int have_rccl = ls_amd_comm_available();
    if (have_rccl) {
        unsigned char id[LS_AMD_UNIQUE_ID_BYTES];
        ls_amd_comm *comm; ls_amd_dist *dist;
        void *d_reps, *d_x, *d_y;
        CHECK(ls_amd_comm_unique_id(id));
        CHECK(ls_amd_comm_create(&comm, 1, 0, id));
        CHECK(ls_amd_malloc(&d_reps, 8 * (size_t)n)); CHECK(ls_amd_malloc(&d_x, 8 * (size_t)n)); CHECK(ls_amd_malloc(&d_y, 8 * (size_t)n));
        CHECK(ls_amd_memcpy_h2d(d_reps, reps, 8 * (size_t)n)); CHECK(ls_amd_memcpy_h2d(d_x, hx, 8 * (size_t)n));
        CHECK(ls_amd_dist_create(&dist, comm, op, LS_AMD_F64, (uint64_t const *)d_reps, n, 3, NULL));
        CHECK(ls_amd_dist_matvec(dist, d_x, d_y, NULL));
        CHECK(ls_amd_plan_check(ls_amd_dist_plan(dist), NULL));
        CHECK(ls_amd_memcpy_d2h(hy3, d_y, 8 * (size_t)n));
        printf("one-rank RCCL path: %d rounds, kernel %s\n", ls_amd_dist_num_rounds(dist), ls_amd_plan_kernel_name(ls_amd_dist_plan(dist)));
        ls_amd_dist_destroy(dist); ls_amd_comm_destroy(comm);
        ls_amd_free(d_reps); ls_amd_free(d_x); ls_amd_free(d_y);
    }
    double scale = 0, e2 = 0, e3 = 0, dot = 0, nrm = 0;
    for (int64_t k = 0; k < n; ++k) {
        if (fabs(hy[k]) > scale) scale = fabs(hy[k]);
        if (fabs(hy[k] - hy2[k]) > e2) e2 = fabs(hy[k] - hy2[k]);
        if (have_rccl && fabs(hy[k] - hy3[k]) > e3) e3 = fabs(hy[k] - hy3[k]);
        dot += hx[k] * hy[k]; nrm += hx[k] * hx[k];
    }
    printf("L = %d, N = %lld, <x|H|x>/<x|x> = %.12f, max|y| = %.6f, |plug-in - partitions| = %.2e, |plug-in - rccl| = %.2e\n",
           L, (long long)n, dot / nrm, scale, e2, e3);
    if (argc > 2) {
        /* dump for an external checker (tests/test_gpu_c_example.py compares every element with the oracle): n, then the
         * representatives, x, and y of the three paths (plug-in | partitions | one-rank RCCL, the last zero-filled without RCCL) */
        FILE *f = fopen(argv[2], "wb");
        if (!f) { fprintf(stderr, "cannot write %s\n", argv[2]); return 1; }
        if (!have_rccl) memset(hy3, 0, 8 * (size_t)n);
        int64_t hdr[2] = {n, have_rccl};
        if (fwrite(hdr, 8, 2, f) != 2 || fwrite(reps, 8, (size_t)n, f) != (size_t)n || fwrite(hx, 8, (size_t)n, f) != (size_t)n ||
            fwrite(hy, 8, (size_t)n, f) != (size_t)n || fwrite(hy2, 8, (size_t)n, f) != (size_t)n || fwrite(hy3, 8, (size_t)n, f) != (size_t)n) {
            fprintf(stderr, "short write to %s\n", argv[2]); fclose(f); return 1;
        }
        fclose(f);
    }
    ls_hs_destroy_operator(op);
    if (!from_yaml) ls_hs_destroy_basis(basis); /* the creator's reference of the hand-built basis */
    ls_chpl_finalize();
    if (e2 > 1e-12 * scale || e3 > 1e-12 * scale) { fprintf(stderr, "MISMATCH\n"); return 2; }
    printf("OK\n");
    return 0;
}
