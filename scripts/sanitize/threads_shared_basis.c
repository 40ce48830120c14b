#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include "ls_hs.h"
#include "ls_amd.h"
static ls_hs_yaml_config *conf;
static void *worker(void *arg) {
    for (int it = 0; it < 200; ++it) {
        ls_hs_operator *o = ls_hs_clone_operator(conf->hamiltonian);
        ls_hs_basis *b = ls_hs_clone_basis(conf->basis);
        ls_hs_operator *o2 = ls_hs_clone_operator(o);
        ls_hs_destroy_operator(o);
        ls_hs_destroy_basis(b);
        ls_hs_destroy_operator(o2);
    }
    return NULL;
}
int main(int argc, char **argv) {
    conf = ls_hs_load_yaml_config(argv[1]);
    if (!conf) return 1;
    pthread_t t[8];
    for (long i = 0; i < 8; ++i) pthread_create(&t[i], NULL, worker, (void *)i);
    for (int i = 0; i < 8; ++i) pthread_join(t[i], NULL);
    ls_hs_destroy_yaml_config(conf);
    puts("done");
    return 0;
}
