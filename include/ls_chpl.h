/* ls_chpl.h -- the symbols the reference's Chapel shared library exports ("ls_chpl_*"), provided by
 * the MI355X-native library instead.  Same names, argument meaning and error behaviour; each entry
 * cites the reference definition it replaces.
 *
 * Error behaviour: the reference `halt`s (aborts the process) on every precondition failure
 * (e.g. DMV:1099-1102, DMV:115-118).  These entry points call the installed error handler
 * (ls_amd_set_error_handler in ls_amd.h), whose default prints to stderr and abort()s.
 */
#ifndef LS_CHPL_H
#define LS_CHPL_H

#include "ls_hs.h"

#ifdef __cplusplus
extern "C" {
#endif

/* /root/reference/src/library.c:19-34.  Boots the runtime: selects the HIP device, creates the
 * default stream, registers the kernel table (the last module initialiser does that in the
 * reference, /root/reference/src/LatticeSymmetries.chpl:31).  Not re-entrant. */
void ls_chpl_init(void);
void ls_chpl_finalize(void);

/* /root/reference/src/LatticeSymmetries.chpl:16-29: fills an ls_chpl_kernels table with the four
 * entry points below and passes it to ls_hs_internal_set_chpl_kernels. */
void ls_chpl_init_kernels(void);

/* /root/reference/src/DistributedMatrixVector.chpl:1095-1110.  y <- H x on HOST pointers of length
 * op->basis->representatives.num_elts (f64).  y is overwritten by the diagonal pass when the
 * operator has diagonal terms and then accumulated into (DMV:1062-1069); halts unless
 * numVectors == 1, number_words == 1 and the basis is built.  Stages x, y and the
 * representatives through HBM; the device-resident entry point is ls_amd_matvec (ls_amd.h). */
void ls_chpl_matrix_vector_product(ls_hs_operator *matrixPtr, int numVectors, double *xPtr,
                                   double *yPtr);

/* /root/reference/src/BatchedOperator.chpl:217-234: coeffs[i] = d(alphas[i]) as f64; halts if the
 * basis requires projection.  The returned array is malloc'ed; free through coeffs->freer. */
void ls_chpl_operator_apply_diag(ls_hs_operator *matrixPtr, int64_t count, uint64_t *alphas,
                                 chpl_external_array *coeffs, int64_t numTasks);

/* /root/reference/src/BatchedOperator.chpl:236-275: CSR-style expansion of `count` rows.
 * betas/coeffs have count * numberOffDiagTerms slots of which offsets[count] are meaningful;
 * coeffs are complex128. */
void ls_chpl_operator_apply_off_diag(ls_hs_operator *matrixPtr, int64_t count, uint64_t *alphas,
                                     chpl_external_array *betas, chpl_external_array *coeffs,
                                     chpl_external_array *offsets, int64_t numTasks);

/* /root/reference/src/StatesEnumeration.chpl:588-603: all representatives of `basis` in ascending
 * order (lower/upper are ignored exactly as in the reference, :596). */
void ls_chpl_enumerate_representatives(ls_hs_basis *basisPtr, uint64_t lower, uint64_t upper,
                                       chpl_external_array *dest);

/* ------------------------------------------------------------------------------------------
 * PRIMME callbacks (/root/reference/src/Diagonalize.chpl:134-162, /root/reference/src/PRIMME.chpl:
 * 313-322, 363-373).  `primme` points at a PRIMME 3.1 primme_params; only the fields named in
 * ls_primme_params_view are read, at the offsets of the public PRIMME 3.1 ABI with 64-bit
 * PRIMME_INT (/root/reference/primme_headers/primme_eigs.h:166-260).
 * ------------------------------------------------------------------------------------------ */
typedef struct ls_primme_params_view {
    int64_t n;
    void (*matrixMatvec)(void *, int64_t *, void *, int64_t *, int *, void *, int *);
    int matrixMatvec_type;
    void (*applyPreconditioner)(void *, int64_t *, void *, int64_t *, int *, void *, int *);
    int applyPreconditioner_type;
    void (*massMatrixMatvec)(void *, int64_t *, void *, int64_t *, int *, void *, int *);
    int massMatrixMatvec_type;
    int numProcs;
    int procID;
    int64_t nLocal;
    void *commInfo;
    void (*globalSumReal)(void *, void *, int *, void *, int *);
    int globalSumReal_type;
    void (*broadcastReal)(void *, int *, void *, int *);
    int broadcastReal_type;
    int numEvals;
    int target;
    int numTargetShifts;
    double *targetShifts;
    int dynamicMethodSwitch;
    int locking;
    int initSize;
    int numOrthoConst;
    int maxBasisSize;
    int minRestartSize;
    int maxBlockSize;
    int64_t maxMatvecs;
    int64_t maxOuterIterations;
    int64_t iseed[4];
    double aNorm;
    double BNorm;
    double invBNorm;
    double eps;
    int orth;
    int internalPrecision;
    int printLevel;
    void *outputFile;
    void *matrix; /* ls_hs_operator*  (Diagonalize.chpl:129-132,195) */
} ls_primme_params_view;

/* matrixMatvec: Y_k <- H X_k for k < *blockSize, columns at x + ldx*k, f64, n = primme->nLocal. */
void ls_chpl_primme_matvec(void *x, int64_t *ldx, void *y, int64_t *ldy, int *blockSize,
                           void *primme, int *ierr);
/* globalSumReal / broadcastReal (/root/reference/src/PRIMME.chpl:267-373): collective over the communicator in
 * primme->commInfo (an ls_amd_comm*) or, when that is NULL, the one installed with ls_amd_set_default_comm
 * (include/ls_amd.h): host buffers are staged through the GPU and reduced with ncclAllReduce / ncclBroadcast over xGMI.
 * No communicator, or one rank: the sum is a copy (sendBuf may alias recvBuf), the broadcast a no-op.  f32 and f64 sums,
 * f64 broadcast, as the reference. */
void primmeGlobalSumReal(void *sendBuf, void *recvBuf, int *count, void *primme, int *ierr);
void primmeBroadcastReal(void *buffer, int *count, void *primme, int *ierr);

#ifdef __cplusplus
}
#endif
#endif /* LS_CHPL_H */
