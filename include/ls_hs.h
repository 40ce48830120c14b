/* ls_hs.h -- the subset of the lattice-symmetries-haskell C ABI ("ls_hs_*") that the reference's hot
 * path consumes, re-provided natively.
 *
 * The reference declares these as `extern` in /root/reference/src/FFI.chpl:85-239 and gets them from
 * liblattice_symmetries_haskell (pinned 14e7319, /root/reference/.github/workflows/ci.yml:6), which is
 * not vendored.  Struct prefixes below are layout-compatible with what the reference looks inside
 * (FFI.chpl:90-126); everything after the "other stuff" marker is ours.  Field lists of
 * ls_hs_nonbranching_terms beyond number_terms/number_bits are [upstream-memory].
 *
 * Host side is plain C11; no HIP or torch types appear in any signature.
 */
#ifndef LS_HS_H
#define LS_HS_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Chapel runtime's external array descriptor (FFI.chpl:36-55 builds them with
 * chpl_make_external_array_ptr); `freer` is a void(*)(void*) or NULL for borrowed storage. */
typedef struct chpl_external_array {
    void *elts;
    uint64_t num_elts;
    void *freer;
} chpl_external_array;

typedef int ls_hs_particle_type; /* FFI.chpl:85-88 */
enum { LS_HS_SPIN = 0, LS_HS_SPINFUL_FERMION = 1, LS_HS_SPINLESS_FERMION = 2 };

typedef struct ls_hs_scalar { double re, im; } ls_hs_scalar; /* == double _Complex */

/* FFI.chpl:90-93 */
typedef struct ls_hs_basis_kernels {
    void *state_index_kernel;
    void *state_index_data;
} ls_hs_basis_kernels;


/* FFI.chpl:94-105 */
typedef struct ls_hs_basis {
    int number_sites;
    int number_particles;
    int number_up;
    ls_hs_particle_type particle_type;
    int spin_inversion;
    bool state_index_is_identity;
    bool requires_projection;
    ls_hs_basis_kernels *kernels;
    chpl_external_array representatives;
    /* ... other stuff ... */
} ls_hs_basis;

/* FFI.chpl:109-113.  Term t acts on a basis state alpha as
 *   active  iff (alpha & m[t]) == r[t]
 *   beta    =  alpha ^ x[t]                      (l[t] = r[t] ^ (x[t] & m[t]) is the image pattern)
 *   coeff   =  v[t] * (-1)^popcount(alpha & s[t])
 * Diagonal terms have x == 0.  number_words == 1 only (DMV:1099-1100). */
typedef struct ls_hs_nonbranching_terms {
    int number_terms;
    int number_bits;
    ls_hs_scalar const *v;
    uint64_t const *m;
    uint64_t const *l;
    uint64_t const *r;
    uint64_t const *x;
    uint64_t const *s;
} ls_hs_nonbranching_terms;


/* FFI.chpl:114-119 */
typedef struct ls_hs_operator {
    ls_hs_basis *basis;
    ls_hs_nonbranching_terms *off_diag_terms;
    ls_hs_nonbranching_terms *diag_terms;
    /* ... other stuff ... */
} ls_hs_operator;

/* FFI.chpl:233-239 */
typedef struct ls_chpl_kernels {
    void *enumerate_states;
    void *operator_apply_off_diag;
    void *operator_apply_diag;
    void *matrix_vector_product;
} ls_chpl_kernels;

void ls_hs_init(void);  /* FFI.chpl:128 */
void ls_hs_exit(void);  /* FFI.chpl:129 */

/* Constructors.  Upstream builds these from JSON / expressions (ls_hs_basis_from_json FFI.chpl:156,
 * ls_hs_create_operator FFI.chpl:191-192); here the host mirror parses the YAML
 * (distributed-matvec_amd/config.py) and hands over flat arrays.
 *   hamming_weight < 0  : unrestricted;  spin_inversion in {0, +1, -1};
 *   permutations[g*number_sites + i] = p_i of generator g ("output bit i = input bit p_i");
 *   sectors[g] : character of generator g is exp(-2 pi i sector / order(g)).
 * The group the generators span is closed here (at most 65536 elements: generators of a larger group -- usually a typo in one
 * permutation -- are an error, reported in milliseconds); sectors that are no one-dimensional representation of it are an error.
 * Returns NULL on error (message via ls_amd_last_error()). */
ls_hs_basis *ls_hs_create_spin_basis(int number_sites, int hamming_weight, int spin_inversion,
                                     int number_generators, int const *permutations,
                                     int const *sectors);
ls_hs_basis *ls_hs_clone_basis(ls_hs_basis const *basis);       /* FFI.chpl:141 */
void ls_hs_destroy_basis(ls_hs_basis *basis);                   /* FFI.chpl:142 */

/* v: interleaved (re, im) pairs.  Terms are split into diagonal / off-diagonal, merged and the
 * off-diagonal ones grouped by flip mask. */
ls_hs_operator *ls_hs_create_operator_from_terms(ls_hs_basis const *basis, int number_terms,
                                                 double const *v, uint64_t const *m,
                                                 uint64_t const *r, uint64_t const *x,
                                                 uint64_t const *s);
ls_hs_operator *ls_hs_clone_operator(ls_hs_operator const *op); /* FFI.chpl:193 */
void ls_hs_destroy_operator(ls_hs_operator *op);                /* FFI.chpl:196 */

/* /root/reference/src/FFI.chpl:121-126: what ls_hs_load_yaml_config hands back; loadConfigFromYaml
 * (/root/reference/src/ForeignTypes.chpl:261-288) clones basis / hamiltonian / observables out of it and destroys it */
typedef struct ls_hs_yaml_config {
    ls_hs_basis *basis;
    ls_hs_operator *hamiltonian; /* NULL when the file has no `hamiltonian` section */
    int number_observables;
    ls_hs_operator **observables;
} ls_hs_yaml_config;
/* /root/reference/src/FFI.chpl:208-209.  The YAML subset of the YAML files under /root/reference/data (basis: number_spins, hamming_weight,
 * spin_inversion, symmetries; hamiltonian / observables: terms of `expression` + `sites`; anchors and aliases, block and
 * flow collections).  NULL on failure, with the reason in ls_amd_last_error(). */
ls_hs_yaml_config *ls_hs_load_yaml_config(char const *filename);
void ls_hs_destroy_yaml_config(ls_hs_yaml_config *config);
/* the same from a NUL-terminated YAML text in memory (not in the reference's ABI) */
ls_hs_yaml_config *ls_amd_load_yaml_config_from_string(char const *text);

uint64_t ls_hs_min_state_estimate(ls_hs_basis const *basis);            /* FFI.chpl:143 */
uint64_t ls_hs_max_state_estimate(ls_hs_basis const *basis);            /* FFI.chpl:144 */
int ls_hs_basis_number_bits(ls_hs_basis const *basis);                  /* FFI.chpl:145 */
int ls_hs_basis_number_words(ls_hs_basis const *basis);                 /* FFI.chpl:146 */
bool ls_hs_basis_has_fixed_hamming_weight(ls_hs_basis const *basis);    /* FFI.chpl:147 */
bool ls_hs_basis_has_spin_inversion_symmetry(ls_hs_basis const *basis); /* FFI.chpl:148 */
bool ls_hs_basis_has_permutation_symmetries(ls_hs_basis const *basis);  /* FFI.chpl:149 */
bool ls_hs_basis_requires_projection(ls_hs_basis const *basis);         /* FFI.chpl:150 */

ptrdiff_t ls_hs_fixed_hamming_state_to_index(uint64_t basis_state);                  /* FFI.chpl:165 */
uint64_t ls_hs_fixed_hamming_index_to_state(ptrdiff_t state_index, int hamming_weight); /* FFI.chpl:166 */

/* Builds basis->representatives through the registered enumerate_states kernel (FFI.chpl:168). */
void ls_hs_basis_build(ls_hs_basis *basis);
/* Borrows `states` (host memory) as the basis' representatives (FFI.chpl:170-171). */
void ls_hs_unchecked_set_representatives(ls_hs_basis *basis, chpl_external_array const *states);

int ls_hs_operator_max_number_off_diag(ls_hs_operator const *op); /* FFI.chpl:200 */
bool ls_hs_operator_is_hermitian(ls_hs_operator const *op);       /* FFI.chpl:201 */
bool ls_hs_operator_is_real(ls_hs_operator const *op);            /* FFI.chpl:202 */

/* Batched per-state queries with the reference's signatures (FFI.chpl:173-184, 219-225).  Upstream
 * these are CPU loops of lattice-symmetries-haskell; here each call stages its HOST arrays through HBM
 * and runs the corresponding HIP kernel (the matvec itself never calls them -- it fuses them).
 *   ls_hs_state_index        indices[k] = position of spins[k] in basis->representatives, < 0 if absent
 *   ls_hs_is_representative  flag = alpha is its orbit minimum; norm = sqrt(stabiliser sum / |G|)
 *   ls_hs_state_info         beta = orbit minimum, character = conj(chi(g0)), norm as above
 *   ls_internal_operator_apply_diag_x1      ys[i] = d(alphas[i]) * xs[i]   (xs == NULL: ys[i] = d)
 *   ls_internal_operator_apply_off_diag_x1  CSR expansion, coefficients times xs[i] when xs != NULL;
 *                                           betas/coeffs need batch_size * max_number_off_diag slots
 * Strides are in elements. */
void ls_hs_state_index(ls_hs_basis const *basis, ptrdiff_t batch_size, uint64_t const *spins,
                       ptrdiff_t spins_stride, ptrdiff_t *indices, ptrdiff_t indices_stride);
void ls_hs_is_representative(ls_hs_basis const *basis, ptrdiff_t batch_size, uint64_t const *alphas,
                             ptrdiff_t alphas_stride, uint8_t *are_representatives, double *norms);
void ls_hs_state_info(ls_hs_basis const *basis, ptrdiff_t batch_size, uint64_t const *alphas,
                      ptrdiff_t alphas_stride, uint64_t *betas, ptrdiff_t betas_stride,
                      ls_hs_scalar *characters, double *norms);
void ls_internal_operator_apply_diag_x1(ls_hs_operator const *op, ptrdiff_t batch_size,
                                        uint64_t const *alphas, double *ys, double const *xs);
void ls_internal_operator_apply_off_diag_x1(ls_hs_operator const *op, ptrdiff_t batch_size,
                                            uint64_t const *alphas, uint64_t *betas,
                                            ls_hs_scalar *coeffs, ptrdiff_t *offsets, double const *xs);

void ls_hs_internal_set_chpl_kernels(ls_chpl_kernels const *kernels); /* FFI.chpl:239 */
ls_chpl_kernels const *ls_hs_internal_get_chpl_kernels(void);

#ifdef __cplusplus
}
#endif
#endif /* LS_HS_H */
