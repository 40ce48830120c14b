#include <stdio.h>
#include <stdlib.h>
#include "ls_hs.h"
#include "ls_amd.h"
int main(int argc, char **argv) {
    ls_hs_init();
    for (int i = 1; i < argc; ++i) {
        ls_hs_yaml_config *c = ls_hs_load_yaml_config(argv[i]);
        if (!c) { printf("%s: %s\n", argv[i], ls_amd_last_error()); continue; }
        if (c->hamiltonian) { ls_hs_operator *o = ls_hs_clone_operator(c->hamiltonian); if (o) ls_hs_destroy_operator(o); }
        ls_hs_basis *b = ls_hs_clone_basis(c->basis); if (b) ls_hs_destroy_basis(b);
        ls_hs_destroy_yaml_config(c);
    }
    /* error paths */
    ls_hs_yaml_config *bad = ls_amd_load_yaml_config_from_string("basis:\n  number_spins: 4\n  symmetries:\n    - permutation: [1,0,3]\n      sector: 0\n");
    if (bad) ls_hs_destroy_yaml_config(bad);
    bad = ls_amd_load_yaml_config_from_string("basis:\n  number_spins: 4\nhamiltonian:\n  terms:\n    - expression: \"\xcf\x83\xe1\xb6\xbb\xe2\x82\x80 \xcf\x83\xe1\xb6\xbb\xe2\x82\x81\"\n      sites: [[0,9]]\n");
    if (bad) ls_hs_destroy_yaml_config(bad);
    ls_hs_exit();
    return 0;
}
