#include <pthread.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include "ls_hs.h"
#include "ls_amd.h"
static char **files; static int nfiles;
static void *worker(void *arg) {
    long id = (long)arg;
    for (int it = 0; it < 40; ++it) {
        ls_hs_yaml_config *c = ls_hs_load_yaml_config(files[(id + it) % nfiles]);
        if (!c) { fprintf(stderr, "load failed: %s\n", ls_amd_last_error()); continue; }
        ls_hs_operator *o = c->hamiltonian ? ls_hs_clone_operator(c->hamiltonian) : NULL;
        ls_hs_basis *b = ls_hs_clone_basis(c->basis);
        (void)ls_hs_basis_requires_projection(b);
        (void)ls_hs_max_state_estimate(b); (void)ls_hs_min_state_estimate(b);
        (void)ls_hs_fixed_hamming_state_to_index(0x0f0fULL + (uint64_t)id); (void)ls_hs_fixed_hamming_index_to_state(it + 1, 6);
        if (o) { (void)ls_hs_operator_max_number_off_diag(o); (void)ls_hs_operator_is_hermitian(o); }
        ls_hs_destroy_yaml_config(c);
        if (o) ls_hs_destroy_operator(o);
        ls_hs_destroy_basis(b);
    }
    return NULL;
}
int main(int argc, char **argv) {
    files = argv + 1; nfiles = argc - 1;
    pthread_t t[8];
    for (long i = 0; i < 8; ++i) pthread_create(&t[i], NULL, worker, (void *)i);
    for (int i = 0; i < 8; ++i) pthread_join(t[i], NULL);
    puts("done");
    return 0;
}
