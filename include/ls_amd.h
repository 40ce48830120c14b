/* ls_amd.h -- device-level C ABI of the MI355X-native matvec: plain pointers and sizes only.
 *
 * This is the surface the reference's Chapel-level entry matrixVectorProduct(H, x, y,
 * representatives) (/root/reference/src/DistributedMatrixVector.chpl:1072-1093) maps onto:
 * x, y and representatives are hash-partitioned, per-partition ascending arrays
 * (partition of sigma = hash64_01(sigma) % P, /root/reference/src/StatesEnumeration.chpl:122-136)
 * that live in HBM.  Pointers named d_* are DEVICE pointers (hipMalloc / torch allocations);
 * `stream` is a hipStream_t passed as void* (NULL = the library's default stream).
 *
 * Two ways to run the off-diagonal exchange (DMV:856-1053 replaced):
 *   - all P partitions in this process on one device  -> ls_amd_matvec (logical partitions; the
 *     "exchange" is a pointer hand-off), and
 *   - one partition per process / GPU               -> ls_amd_dist_matvec (generate + RCCL all-to-all-v
 *     + scatter inside the library), or the three stages separately (ls_amd_generate, an exchange of
 *     the caller's, ls_amd_scatter).
 * All functions return 0 on success or a negative error code; ls_amd_last_error() explains.
 */
#ifndef LS_AMD_H
#define LS_AMD_H

#include "ls_hs.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { LS_AMD_F64 = 0, LS_AMD_C128 = 1 } ls_amd_dtype;

/* how y is produced when P == 1 and the operator is Hermitian */
typedef enum {
    LS_AMD_MODE_AUTO = 0, /* pull when the operator is Hermitian, else push (LS_AMD_MODE env: "push" | "pull") */
    LS_AMD_MODE_PUSH = 1, /* y[idx(beta)] += c x[i]   -- atomic scatter (ConcurrentAccessor) */
    LS_AMD_MODE_PULL = 2  /* y[i] = d x[i] + sum conj(c) x[idx(beta)]  -- gather, no atomics  */
} ls_amd_mode;

typedef struct ls_amd_plan ls_amd_plan;

/* error plumbing ------------------------------------------------------------------------- */
char const *ls_amd_last_error(void);
typedef void (*ls_amd_error_handler)(char const *message);
/* handler used by the halting ls_chpl_* / ls_hs_* entry points; NULL restores print+abort */
void ls_amd_set_error_handler(ls_amd_error_handler handler);

/* device helpers (thin wrappers so that a C / ctypes caller needs no HIP headers) ---------- */
int ls_amd_device_count(void);
int ls_amd_set_device(int device);
int ls_amd_malloc(void **d_ptr, size_t bytes);
int ls_amd_free(void *d_ptr);
int ls_amd_memcpy_h2d(void *d_dst, void const *h_src, size_t bytes);
int ls_amd_memcpy_d2h(void *h_dst, void const *d_src, size_t bytes);
int ls_amd_memcpy_d2d(void *d_dst, void const *d_src, size_t bytes, void *stream); /* async */
int ls_amd_memset(void *d_ptr, int value, size_t bytes, void *stream);
int ls_amd_synchronize(void *stream);

/* a plain streaming copy (16 bytes per lane): the device copy kernel the attainable HBM rate of the box is measured with
 * (SURVEY.md 8(d)); bytes is rounded down to a multiple of 16 */
int ls_amd_stream_copy(void *d_dst, void const *d_src, int64_t bytes, void *stream);
/* the read-only counterpart: every thread reads `per_thread` 16-byte elements and keeps an xor of them (d_sink: 4 bytes of device
 * memory, practically never written): the attainable READ rate of the box, the line the pull kernels' traffic (> 90 % reads)
 * is held against */
int ls_amd_stream_read(void const *d_src, int64_t bytes, int per_thread, void *d_sink, void *stream);

/* Eigensolver callers (Diagonalize.chpl:174-225 hands H to PRIMME; its Lanczos / Davidson steps orthogonalise one vector against
 * a block of basis vectors).  One sweep over the block V (m <= ls_amd_orth_max_rows() rows of n f64, row stride ldv) and w:
 *     if d_h_in:  w <- w - sum_k h_in[k] V[k];    out[k] = <V[k], w> (k < m, over the updated w);    out[m] = <w, w>
 * d_out: device [m + 1].  Pass 1 (h_in NULL) -> coefficients and norm; pass 2 (h_in = them) applies them and returns the remaining
 * overlaps in the same sweep, so classical Gram-Schmidt "twice" reads V two times instead of four when they are at rounding
 * level.
 * Requirements (not validated beyond the sizes): V and w are f64, each row of V and w contiguous, w does not alias any row of
 * V, d_out / d_h_in do not alias V or w.  Failures return -1 with a message in ls_amd_last_error(). */
int ls_amd_orth_max_rows(void);
int ls_amd_orth_pass(int m, int64_t n, double const *d_V, int64_t ldv, double *d_w, double const *d_h_in, double *d_out, void *stream);
/* thick restart of such a solver: V[:m_out] <- S^T V[:m_in] in place (d_S: m_in x m_out, row-major; m_out <= m_in <= max rows) */
int ls_amd_basis_rotate(int m_in, int m_out, int64_t n, double *d_V, int64_t ldv, double const *d_S, void *stream);

/* ------------------------------------------------------------------------------------------
 * The host-pointer boundary: ls_chpl_matrix_vector_product (DMV:1095-1110) and ls_chpl_primme_matvec (Diagonalize.chpl:134-162)
 * take `double *` as the reference's do.  What a call costs is decided by where that memory lives:
 *   LS_AMD_PTR_DEVICE    device memory (hipMalloc, a torch CUDA tensor): used in place, ZERO copies -- a GPU-resident
 *                        eigensolver can call the reference's entry points as they are
 *   LS_AMD_PTR_MANAGED   hipMallocManaged memory: fine-grained unless advised otherwise, and the push kernels' hardware f64 atomics
 *                        are specified for coarse-grained memory only -- never used in place: staged through the plan's own
 *                        hipMalloc vectors by one DMA per direction.  (The device-pointer entries -- ls_amd_matvec,
 *                        ls_amd_dist_matvec -- REFUSE a managed y for plans that accumulate with atomics.)
 *   LS_AMD_PTR_PINNED    hipHostMalloc'ed memory, or memory registered with ls_amd_host_register (once per workspace: PRIMME
 *                        reuses its vectors): one DMA per direction at the PCIe rate
 *   LS_AMD_PTR_PAGEABLE  anything else: double-buffered pinned bounce chunks filled by a small pool of host threads, upload
 *                        and download at the same time
 * The device copies of x / y persist with the cached plan (no hipMalloc / hipFree per call), y is uploaded only when the
 * operator has no diagonal terms (else it is assigned, DMV:1062-1063), and the columns of a PRIMME block share one pipeline
 * (column k + 1 goes up and column k - 1 comes down while column k computes).  Knobs: LS_AMD_STAGE=0 (plain synchronous
 * hipMemcpy), LS_AMD_STAGE_CHUNK_KB (32768), LS_AMD_STAGE_THREADS (min(16, cores / 4)). */
enum { LS_AMD_PTR_PAGEABLE = 0, LS_AMD_PTR_PINNED = 1, LS_AMD_PTR_DEVICE = 2, LS_AMD_PTR_MANAGED = 3 };
int ls_amd_pointer_kind(void const *p);
/* (Register long-lived, page-aligned workspaces.  On ROCm 7.x without XNACK, registering small malloc'ed heap blocks, unregistering
 * and freeing them, and letting the allocator reuse those addresses for pageable buffers of later hipMemcpy calls ended in GPU
 * memory faults of those later copies -- tests/test_gpu_matvec.py::test_host_pointer_boundary_memory_kinds, round 6.) */
int ls_amd_host_register(void *p, size_t bytes);   /* hipHostRegister: the caller keeps the memory alive until ... */
int ls_amd_host_unregister(void *p);               /* ... this */
struct ls_amd_boundary_stats {
    int64_t calls, columns;        /* entries into the boundary, vectors applied */
    int64_t bytes_h2d, bytes_d2h;  /* bytes that crossed PCIe */
    int64_t device_x, device_y;    /* columns whose x / y was device memory (used in place) */
};
void ls_amd_boundary_stats_get(struct ls_amd_boundary_stats *out, int reset);

/* hash64_01 / localeIdxOf on the host (StatesEnumeration.chpl:122-136) */
uint64_t ls_amd_hash64_01(uint64_t x);
int ls_amd_locale_idx_of(uint64_t basis_state, int num_locales);

/* ------------------------------------------------------------------------------------------
 * Objects built by somebody else (the real lattice-symmetries-haskell, /root/reference/src/FFI.chpl:94-119).
 * This library keeps everything it knows beyond the reference's struct prefix in a side table keyed by the object's
 * address, never in the struct: a foreign ls_hs_basis / ls_hs_operator only has to be registered once.
 *   ls_amd_adopt_basis     reads number_sites, number_up (Hamming weight or -1), spin_inversion, requires_projection and
 *                          `representatives` from the prefix; the symmetry generators are not part of the prefix and are
 *                          passed in the convention of ls_hs_create_spin_basis (0 generators for an unsymmetrised basis)
 *   ls_amd_adopt_operator  rebuilds the term / flip-mask-group tables from off_diag_terms / diag_terms
 *                          (ls_hs_nonbranching_terms with number_bits <= 64); its basis must be known
 *   ls_amd_release         forgets an adopted object and frees its tables (the foreign struct is left alone)
 * After that every entry point (ls_chpl_matrix_vector_product, plans, ls_chpl_primme_matvec, ...) accepts the foreign
 * pointers.  An unknown pointer halts with a message naming these functions.
 */
int ls_amd_adopt_basis(ls_hs_basis const *basis, int number_generators, int const *permutations, int const *sectors);
int ls_amd_adopt_operator(ls_hs_operator const *op);
void ls_amd_release(void const *object);

/* ------------------------------------------------------------------------------------------
 * One locale per process / GPU: the inter-GPU exchange lives in the C host (RCCL over xGMI).
 *
 * Replaces /root/reference/src/DistributedMatrixVector.chpl:313-853 (GlobalPtrStore, _LocalBuffer /
 * _RemoteBuffer mailboxes with one-sided PUTs and remote flag stores, Producer / Consumer tasks,
 * three barriers per matvec) by bulk-synchronous rounds:
 *     generate(r)  ->  ONE grouped ncclSend / ncclRecv with the plan's exact per-round byte counts
 *                      (all-to-all-v of (sigma_j, c_j x_i) packets; every GPU pair uses its own xGMI link)
 *                  ->  scatter(r)
 * double-buffered over two HIP streams: the exchange of round r overlaps generate(r + 1) and scatter(r - 1).
 * librccl is loaded at run time (dlopen); the bootstrap of the 128-byte unique id is the caller's
 * (MPI_Bcast, a file, torch.distributed's store, ...), exactly like ncclGetUniqueId / ncclCommInitRank.
 * rank == locale index == hash64_01(sigma) % size (StatesEnumeration.chpl:133-136).
 */
typedef struct ls_amd_comm ls_amd_comm;
#define LS_AMD_UNIQUE_ID_BYTES 128
int ls_amd_comm_available(void);                 /* 1 when librccl can be loaded */
int ls_amd_comm_unique_id(void *id /* [LS_AMD_UNIQUE_ID_BYTES], written on the calling rank */);
int ls_amd_comm_create(ls_amd_comm **comm, int size, int rank, void const *id);
void ls_amd_comm_destroy(ls_amd_comm *comm);
/* Loop-back group (test infrastructure): `size` communicators inside this process on the current device, one per host
 * thread; every ls_amd_comm_* / ls_amd_dist_* / ls_amd_repl_* call works on them unchanged, the transport being
 * device-to-device copies behind a barrier.  The way to run more than one rank on a one-GPU box (RCCL refuses that). */
int ls_amd_comm_create_local(ls_amd_comm **comms /* [size] */, int size);
int ls_amd_comm_size(ls_amd_comm const *comm);
int ls_amd_comm_rank(ls_amd_comm const *comm);
/* the communicator size as RCCL itself reports it (ncclCommCount): what `bench.py --gpus N` prints as `rccl.comm_count` so that
 * a multi-GPU number carries evidence of the transport it ran on; 0 = loop-back group (test transport), -1 = error */
int ls_amd_comm_rccl_count(ls_amd_comm const *comm);
/* in-place collectives on device buffers (ordered on `stream`) */
int ls_amd_comm_allreduce_sum_f64(ls_amd_comm *comm, double *d_buf, int64_t count, void *stream);
int ls_amd_comm_allreduce_max_i64(ls_amd_comm *comm, int64_t *d_buf, int64_t count, void *stream);
int ls_amd_comm_broadcast(ls_amd_comm *comm, void *d_buf, int64_t bytes, int root, void *stream);
/* the communicator primmeGlobalSumReal / primmeBroadcastReal / ls_chpl_primme_matvec use when
 * primme->commInfo is NULL (the reference's PRIMME callbacks are collective over all locales,
 * /root/reference/src/PRIMME.chpl:267-373); NULL = single process */
void ls_amd_set_default_comm(ls_amd_comm *comm);
ls_amd_comm *ls_amd_default_comm(void);

/* matrixVectorProduct for this rank's block of the hashed vectors (DMV:1072-1093, one locale per process).
 * d_reps_local: this rank's representatives (ascending, device, borrowed for the lifetime of the object).
 * num_rounds <= 0: agreed on collectively (max local count / LS_AMD_ROWS_PER_ROUND, default 2^24 rows). */
typedef struct ls_amd_dist ls_amd_dist;
int ls_amd_dist_create(ls_amd_dist **dist, ls_amd_comm *comm, ls_hs_operator const *op, ls_amd_dtype dtype,
                       uint64_t const *d_reps_local, int64_t count_local, int num_rounds, void *stream);
void ls_amd_dist_destroy(ls_amd_dist *dist);
/* y <- H x (y is assigned by the diagonal pass, then accumulated into; untouched first when the operator has
 * no diagonal terms, DMV:1062-1063).  Collective: every rank of the communicator must call it. */
int ls_amd_dist_matvec(ls_amd_dist *dist, void const *d_x, void *d_y, void *stream);
ls_amd_plan *ls_amd_dist_plan(ls_amd_dist *dist);          /* timing, check, nnz, kernel name */
int64_t ls_amd_dist_exchange_bytes(ls_amd_dist const *dist); /* bytes this rank sends per matvec */
int ls_amd_dist_num_rounds(ls_amd_dist const *dist);

/* The cheaper exchange for Hermitian operators: replicate x instead of sending packets (N w bytes instead of
 * nnz (8 + w): ~30x fewer on the chains).  x, y and the representatives stay hash-partitioned at the interface; per matvec
 * every rank's block of x goes to every peer (one grouped send/recv), one permutation pass built from `masks` puts the
 * blocks into global ascending order (arrFromHashedToBlock, HashedToBlock.chpl:67-153), the pull kernels compute the
 * contiguous global rows [N r / P, N (r + 1) / P) with no atomics, and the results return to their owners with one
 * all-to-all-v (arrFromBlockToHashed restricted to the range).  Needs N w (+ the global representatives, and for projected
 * bases the hash table) of HBM per rank.
 * d_reps_global: the whole basis in ascending order; d_masks[i] = owner of state i (ls_amd_enumerate_states); both borrowed. */
typedef struct ls_amd_repl ls_amd_repl;
int ls_amd_repl_create(ls_amd_repl **repl, ls_amd_comm *comm, ls_hs_operator const *op, ls_amd_dtype dtype,
                       uint64_t const *d_reps_global, uint8_t const *d_masks, int64_t count_global, void *stream);
void ls_amd_repl_destroy(ls_amd_repl *repl);
/* y_local <- (H x)_local for this rank's blocks of the hashed vectors; collective */
int ls_amd_repl_matvec(ls_amd_repl *repl, void const *d_x_local, void *d_y_local, void *stream);
ls_amd_plan *ls_amd_repl_plan(ls_amd_repl *repl);
int64_t ls_amd_repl_exchange_bytes(ls_amd_repl const *repl);
/* bytes of x this rank receives per matvec.  Projected bases: every peer's block.  Unprojected bases: the contiguous rows of a
 * rank read only their own neighbourhood and the partner blocks of the top bonds, kept as <= 16 intervals of global rows; an
 * owner's elements are ascending in global rank, so each interval is one contiguous piece of every owner's block and is sent
 * as it lies (chain_32 at 8 ranks: 44 % of the vector; LS_AMD_REPL_REACH=0: the whole vector). */
int64_t ls_amd_repl_x_in_bytes(ls_amd_repl const *repl);
/* Fault injection (test hooks of the self-verification of `bench.py --gpus N`, tests/test_gpu_loopback.py): shift ONE segment
 * offset of the exchange layout by one element, memory-safely (the segment shrinks by the same element), so that the exchange
 * still completes but delivers misplaced data.  Return 1 when a segment was corrupted, 0 when the layout has none to corrupt
 * (e.g. a packet plan with a single rank: nothing leaves the GPU). */
int ls_amd_test_corrupt_dist(ls_amd_dist *dist);
/* test hook (thread-local): plans created afterwards consume their sorted packet streams with n windows of y per block (n > 1:
 * the run of a stream in one window starts where its run in the previous window ended); 0 = the plan decides */
void ls_amd_test_set_stream_windows_per_block(int n);
/* test hook (thread-local): the per-source stream buffers of ls_amd_matvec plans "do not fit" -- such a plan must come back with one
 * shared buffer and the atomic consumers instead of failing */
void ls_amd_test_fail_stream_buffers(int on);
/* ... and the set-up of the sorted streams fails on the calling rank of ls_amd_dist_create: ALL ranks must come back with the atomic
 * consumers and the default rows per round (the verdict of every set-up step is collective) */
void ls_amd_test_fail_dist_streams(int on);
int ls_amd_test_corrupt_repl(ls_amd_repl *repl);
/* Never hang (csrc/comm.cpp).  Every exchange layout is cross-checked between all ranks at set-up (what s sends to d == what d
 * expects from s; ls_amd_dist_create / ls_amd_repl_create fail on EVERY rank with the segment, the two ranks and both byte counts).
 * At run time a watchdog thread per RCCL communicator ends the process (exit code 86, message on stderr: rank, what was in
 * flight, for how long) when a collective or an exchange has not completed LS_AMD_COMM_WATCHDOG_S seconds (default 300, 0 = off)
 * after it was issued -- a dead peer or mismatched counts would otherwise leave every rank in a stream synchronisation for ever;
 * loop-back groups meet at a rendezvous with the same deadline.  ls_amd_comm_wait is the polite form: it waits until `stream` and
 * the communicator's exchange stream have drained, or returns -1 (message in ls_amd_last_error) after timeout_s (<= 0: the
 * watchdog's deadline). */
int ls_amd_comm_wait(ls_amd_comm *comm, void *stream, double timeout_s);
/* test hooks: rank `rank` sends delta_bytes more (less) to its right neighbour than that one expects -- late == 0: in the layout the
 * set-up check sees, late == 1: after it (a run-time fault); rank < 0 switches it off.  ls_amd_comm_test_stall: the exchange
 * stream of an RCCL communicator stalls for `seconds` behind an armed watchdog item. */
void ls_amd_test_skew_exchange(int rank, int64_t delta_bytes, int late);
int ls_amd_comm_test_stall(ls_amd_comm *comm, double seconds);
/* ... and the one-GPU counterpart (`bench.py --inject-fault` at N = 1): the row kernel of a plan whose partitions all live in this
 * process skips one row of the first tile (memory-safe); 1 when a tile was shortened, 0 when the plan has no tile map. */
int ls_amd_test_corrupt_plan(ls_amd_plan *plan);

/* ------------------------------------------------------------------------------------------
 * Plans.  A plan binds an operator to the partition layout and owns every per-basis device table:
 * term tables, symmetry-group networks, per-row norms, index prefix tables, per-round send counts.
 *
 *   num_partitions   P >= 1 (<= 256, DMV:664)
 *   my_partition     -1: all P partitions live in this process (counts[P], d_reps[P]);
 *                    p >= 0: this process owns only partition p (counts[0], d_reps[0] describe it)
 *   d_reps           device pointers to the ascending representatives of each owned partition;
 *                    borrowed -- must outlive the plan
 *   num_rounds       rows of a partition are processed in this many bulk-synchronous rounds
 *                    (0 = choose from LS_AMD_ROWS_PER_ROUND, default 2^24 rows)
 * ------------------------------------------------------------------------------------------ */
int ls_amd_plan_create(ls_amd_plan **plan, ls_hs_operator const *op, ls_amd_dtype dtype,
                       int num_partitions, int my_partition, uint64_t const *const *d_reps,
                       int64_t const *counts, int num_rounds, ls_amd_mode mode, void *stream);
void ls_amd_plan_destroy(ls_amd_plan *plan);

int ls_amd_plan_num_rounds(ls_amd_plan const *plan);
/* which kernel family the plan selected: "direct-push", "direct-pull", "tile", "tile-pull",
 * "replicated-direct-pull", "replicated-tile-pull" */
char const *ls_amd_plan_kernel_name(ls_amd_plan const *plan);
/* number of (beta, value) packets partition `my_partition` sends to every destination in `round`
 * (counts[P]; the own slot is 0 because local contributions are scattered in place) */
int ls_amd_plan_send_counts(ls_amd_plan const *plan, int round, int64_t *counts);
/* bytes of one packet segment entry: 8 (beta) + 8 or 16 (value) */
int ls_amd_plan_packet_bytes(ls_amd_plan const *plan);
/* bytes of per-row plan / basis data the dominant kernel streams next to x and y (8-byte fused record of the staged row
 * kernel, state words + cached partner ranks, state + norm for projected bases): compulsory traffic = rows (this + 2 w) */
int ls_amd_plan_row_bytes(ls_amd_plan const *plan);
/* total off-diagonal non-zeros generated per matvec by the owned partitions (from the count pass;
 * 0 when the plan runs a direct kernel and never counted) */
int64_t ls_amd_plan_nnz(ls_amd_plan const *plan);

/* Slot cache of the projected-basis pull path (opt-in; NOT matrix-free).  The indexed pull kernel spends ~80 % of a matvec
 * on work that does not depend on x: term expansion (BatchedOperator.chpl:163-213), the orbit minimum of every generated state
 * and the look-up of its representative (ls_hs_state_index, DMV:102) -- per non-zero one 4-byte slot + one row byte (+ the
 * coefficient unless every packet has the same real amplitude).  A plan that is applied many times (Diagonalize / PRIMME)
 * can keep those streams in HBM: the first matvec resolves them, every later one only gathers x[slot] and accumulates.
 * max_bytes = ceiling for the streams (<= 0: whatever the device can allocate).  Returns the number of rows whose streams
 * are kept -- every row, or the longest prefix of whole 256-row tiles that fits (the rows behind it keep the fused kernel);
 * 0 = nothing cached, the plan stays matrix-free (unprojected bases, value-table mode, no room). */
int64_t ls_amd_plan_cache_slots(ls_amd_plan *plan, int64_t max_bytes);
/* rows covered by the slot cache and the HBM it holds (0 / 0 when off) */
int ls_amd_plan_slot_cache_rows(ls_amd_plan const *plan, int64_t *rows, int64_t *bytes);

/* matrixVectorProduct(H, x, y, representatives), all partitions in this process
 * (DMV:1072-1093).  d_x[p], d_y[p]: device arrays of counts[p] elements of the plan's dtype.
 * Asynchronous on `stream`; call ls_amd_plan_check to synchronise and collect the
 * "invalid index" condition the reference halts on (DMV:115-118). */
int ls_amd_matvec(ls_amd_plan *plan, void const *const *d_x, void *const *d_y, void *stream);
int ls_amd_plan_check(ls_amd_plan *plan, void *stream);

/* ------------------------------------------------------------------------------------------
 * Replicated-x plans (Hermitian operators, one partition per process).  Instead of exchanging
 * (sigma_j, c_j x_i) packets, every rank holds the whole x in global ascending ("block") order --
 * the caller all-gathers it, nnz / N ~ 16 times fewer bytes than the packets -- and computes y for
 * its own rows by the pull formulation  y_r = d x_r + sum conj(c) x[idx_global(beta)].
 * d_reps_global: all representatives, ascending (borrowed, like d_reps_local).
 * ------------------------------------------------------------------------------------------ */
int ls_amd_plan_create_replicated(ls_amd_plan **plan, ls_hs_operator const *op, ls_amd_dtype dtype,
                                  int num_partitions, int my_partition, uint64_t const *d_reps_local,
                                  int64_t count_local, uint64_t const *d_reps_global,
                                  int64_t count_global, void *stream);
int ls_amd_matvec_replicated(ls_amd_plan *plan, void const *d_x_global, void *d_y_local, void *stream);

/* Kernel timing with HIP events recorded on the launch stream, around every launch of the plan's
 * dominant kernel (direct-push/direct-pull: the fused row kernel; tile: the staged kernel).
 * enable with max_samples > 0 (ring of that many event pairs; 0 disables); read back after
 * ls_amd_plan_check / a stream sync.  *count receives the number of samples written (<= capacity)
 * and the ring is reset. */
int ls_amd_plan_enable_timing(ls_amd_plan *plan, int max_samples);
/* Stage timers: the reference's --kDisplayTimings tree (DMV:1028-1052), HIP events around every stage launch on the
 * launch stream.  stages: 0 localDiagonal, 1 preparation of x (value-table refresh / x n(rep) / hashed -> block permutation),
 * 2 row kernel (fused paths), 3 producers (k_tile), 4 exchange (wait on the compute stream; replicated-x: the all-to-all of x),
 * 5 consumers (k_scatter), 6 replicated-x: y rows grouped by owner and returned.  max_events = capacity of the event pool
 * between two reads (0 disables). */
#define LS_AMD_NUM_STAGES 7
int ls_amd_plan_enable_stage_timing(ls_amd_plan *plan, int max_events);
int ls_amd_plan_stage_times(ls_amd_plan *plan, double *ms /* [LS_AMD_NUM_STAGES] totals */, int64_t *calls /* [LS_AMD_NUM_STAGES] */, int64_t *matvecs);
int ls_amd_plan_timing_report(ls_amd_plan *plan, char *buf, size_t capacity); /* the tree as text, per matvec */
int ls_amd_plan_kernel_times(ls_amd_plan *plan, float *ms, int capacity, int *count);

/* x[i] = u(hash(states[i], seed)) - 0.5 (re and im for c128): deterministic vectors keyed by the
 * basis state, identical for every partitioning (tests / bench input) */
int ls_amd_fill_random(int64_t n, uint64_t const *d_states, uint64_t seed, ls_amd_dtype dtype,
                       void *d_out, void *stream);

/* one-partition-per-process building blocks -------------------------------------------------
 * ls_amd_diag      y = d(sigma) x           (localDiagonal, DMV:59-71; no-op without diag terms)
 * ls_amd_generate  rows of `round` -> term expansion, projection, hash bucketing; packets for
 *                  remote owners are packed into d_send as P segments in destination order,
 *                  segment d = [beta x count_d][value x count_d] with count_d from
 *                  ls_amd_plan_send_counts; packets owned by this partition are indexed and
 *                  atomically added into d_y directly   (Producer.run, DMV:663-734)
 * ls_amd_scatter   n received packets (beta at d_betas, values at d_values) -> local index ->
 *                  atomic y[idx] += value                 (Consumer.run / localProcess, DMV:73-127)
 * ------------------------------------------------------------------------------------------ */
int ls_amd_diag(ls_amd_plan *plan, void const *d_x, void *d_y, void *stream);
int ls_amd_generate(ls_amd_plan *plan, int round, void const *d_x, void *d_y, void *d_send,
                    void *stream);
int ls_amd_scatter(ls_amd_plan *plan, int64_t n, uint64_t const *d_betas, void const *d_values,
                   void *d_y, void *stream);
/* Packet layout.  A segment of c packets is [key x c (padded to 8 bytes)][value x c]:
 *   ls_amd_plan_key_bytes == 8   the key is the state beta (u64) and the consumer ranks / searches it (every projected basis)
 *   ls_amd_plan_key_bytes == 4   PRE-INDEXED packets (hash partitions of unprojected fixed-weight bases): the key is the u32
 *                                index of beta inside the destination's block -- the producer reads it off an
 *                                all-destinations rank directory every rank derives alone (the owner of a state is a hash of
 *                                the state): 12 instead of 16 bytes per f64 packet on the wire and a search-free consumer, for
 *                                P / 4 bytes of HBM per basis state (ls_amd_plan_packet_index_bytes); taken while that fits
 *                                LS_AMD_PACKET_INDEX_MAX (default: a quarter of the free HBM), LS_AMD_PACKET_INDEX=0: never
 * ls_amd_plan_segment_bytes(c) = bytes of such a segment (what the all-to-all-v moves per (round, peer));
 * ls_amd_plan_segment_value_offset(c) = where its values start; ls_amd_plan_packet_bytes = key + value bytes (nominal).
 * ls_amd_scatter takes either kind (d_betas = the segment's key array).  ls_amd_scatter_round consumes ALL segments of a
 * round's receive buffer in one launch: segment s = counts[s] packets at d_recv + offsets[s].
 *
 * SORTED STREAMS (round 5; drivers that own both ends of the exchange: ls_amd_matvec over the partitions of one process and
 * ls_amd_dist_matvec with >= 2 ranks; csrc/k_packets.hip, k_tile_st / k_window).  When every off-diagonal term of the operator is
 * an exchange pair and the basis is an unprojected fixed-weight one, the pre-indexed packets of a segment are written as
 * 2 x (number of pairs) STREAMS, one per (pair, pattern of alpha on the pair): along a stream beta = alpha + constant, so --
 * the producer's rows and the destination's states both ascending -- the keys of a stream ASCEND.  The consumer then owns a
 * window of 2048 rows of y, finds every stream's sub-run for the window by binary search, adds those packets into an LDS copy of
 * the window and writes y once: no atomic, no fabric request per packet (chain_28 over 8 partitions 21.7 -> 10.8 ms, c128 42.8 ->
 * 14.7 ms; profiles/r5_packets_streams_ab.txt).  The own partition's packets take the same way (a segment of the send buffer).
 * Layout on the wire is unchanged (12-byte packets, keys then values); the stream starts of every segment are exchanged once at
 * set-up.  LS_AMD_PACKET_STREAMS=0: the atomic consumers above (A/B).  A plan driven through ls_amd_generate / ls_amd_scatter by a
 * foreign exchange never writes streams. */
int ls_amd_plan_key_bytes(ls_amd_plan const *plan);
int64_t ls_amd_plan_segment_bytes(ls_amd_plan const *plan, int64_t count);
int64_t ls_amd_plan_segment_value_offset(ls_amd_plan const *plan, int64_t count);
int64_t ls_amd_plan_packet_index_bytes(ls_amd_plan const *plan);
int ls_amd_scatter_round(ls_amd_plan *plan, int num_segments, int64_t const *counts, int64_t const *offsets,
                         void const *d_recv, void *d_y, void *stream);

/* ------------------------------------------------------------------------------------------
 * Basis construction on the device (enumerateStates, StatesEnumeration.chpl:516-585) and the
 * block <-> hashed layout converters (BlockToHashed.chpl:87-208, HashedToBlock.chpl:67-153).
 * ------------------------------------------------------------------------------------------ */
/* Enumerates all representatives in ascending order into a freshly allocated device array
 * (*d_states, ls_amd_free it) and, when d_masks != NULL, the owner hash64_01 % num_locales of each
 * state in the same (global ascending, "block") order. */
int ls_amd_enumerate_states(ls_hs_basis const *basis, int num_locales, uint64_t **d_states,
                            uint8_t **d_masks, int64_t *count, void *stream);
/* counts[p] = number of masks equal to p */
/* d_out[i] = d_src[d_perm[i]] (elements of 8 or 16 bytes; perm int32 or int64): the permutation pass of the replicated-x
 * exchange -- the received blocks (hashed order) into global ascending order, i.e. arrFromHashedToBlock
 * (HashedToBlock.chpl:67-153) with the permutation precomputed from `masks` */
int ls_amd_gather(int64_t n, void const *d_perm, int perm_is_64, int elt_size, void const *d_src, void *d_out, void *stream);
int ls_amd_mask_counts(int64_t n, uint8_t const *d_masks, int num_locales, int64_t *counts,
                       void *stream);
/* stable partition of a block-order array (elt_size 8 or 16 bytes) into P hashed parts */
int ls_amd_block_to_hashed(int64_t n, uint8_t const *d_masks, int num_locales, int elt_size,
                           void const *d_src, void *const *d_dest, void *stream);
/* inverse: k-way unmerge by masks */
int ls_amd_hashed_to_block(int64_t n, uint8_t const *d_masks, int num_locales, int elt_size,
                           void const *const *d_src, void *d_dest, void *stream);

/* test hooks: evaluate compiled host-side tables on the CPU (no device work) ------------------ */
int ls_amd_basis_group_order(ls_hs_basis const *basis);
uint64_t ls_amd_basis_apply_group_element(ls_hs_basis const *basis, int element, uint64_t state);
int ls_amd_basis_group_character(ls_hs_basis const *basis, int element, double *re, double *im);

/* Host-only test hook (no device needed): the tile map of the row kernels (distributed-matvec_amd/csrc/lsk.h: lsk_tilemap)
 * for n rows in tiles of tile_rows, dealt to the 8 XCD lists contiguously (chunk == 0) or in round-robin chunks.
 * Returns the number of entries per list; *entries (malloc'ed, release with ls_amd_test_free) holds the 8 lists
 * (first row | rows << 48; 0 = empty slot); < 0 on error. */
int64_t ls_amd_test_tilemap(int64_t n, int tile_rows, int64_t chunk, uint64_t **entries);
void ls_amd_test_free(void *p);
/* Host-only test hook: the near-window search of the staged pull kernel for projected bases (k_tile_pull): position of
 * `key` among the ascending reps[0, n) (n <= 1280, stored as saturating 32-bit offsets from reps[0] exactly as a tile
 * stores them in LDS), -1 if it is not there or not representable (then the kernel takes the hash table), -2 on bad n */
int ls_amd_test_window_find(uint64_t const *reps, int n, uint64_t key);
/* ... and the near window of the INDEXED kernels (k_pull_t), a two-way hash set in LDS over the same offsets (n <= 1024): the
 * position of `key`, or -1 when the window does not answer -- the key is absent, or its set was already full when it was
 * staged (the kernel then takes the static index table, which holds every representative); never a wrong position */
int ls_amd_test_nw_find(uint64_t const *reps, int n, uint64_t key);
/* Host-only test hook: the near-pair table of the staged row kernel (distributed-matvec_amd/csrc/k_rows.hip: chain_lds_image) for
 * vectors of `elem` bytes per entry and `ldsp` pairs served from the LDS window: 480 entries of four int16 into `out`. */
int ls_amd_test_chain_near_table(int elem, int ldsp, int16_t *out);
/* Host-only test hook: the orbit minimum of `a` under the translations of a ring of L sites (and its reflections / the global
 * spin flip when asked), computed by the candidate-pruning routine the projected-basis kernels use (K4, trivial sector). */
uint64_t ls_amd_test_rep_trivial_dihedral(uint64_t a, int L, int inv, int reflect);
/* Profiling entry: K4 alone over the packets of a ring (every state of d_reps with each adjacent pair flipped), see scripts/k4_rate.py */
int ls_amd_bench_k4(int L, int inv, int reflect, int variant, int64_t n, uint64_t const *d_reps, uint64_t *d_out, void *stream);
/* Host-only test hooks of the lattice-group form of K4 (trivial sectors of groups that contain every translation of a
 * tw x (L / tw) torus: the translations are walked with bit operations, only one element per coset -- the point group -- goes
 * through a compiled network): the width found (-1: the group has no such subgroup) and the number of cosets; the orbit
 * minimum of `state` computed that way (with the global spin flip folded in when the basis has one; ~0 when not applicable) */
int ls_amd_test_translation_cosets(ls_hs_basis const *basis, int *n_cosets);
uint64_t ls_amd_test_rep_by_cosets(ls_hs_basis const *basis, uint64_t state);
/* ... and of its factorised form (K4 mode 5): when the cosets are the point group of the torus itself -- D2 = {1, r, o, r o}
 * (rows reversed, row order reversed, both) or on a square torus D4 = D2 x {1, transpose}, or a subgroup -- the mask of the
 * images that belong to the group (bits 0-3 of the word, 4-7 of its transpose; 0: the cosets are not of that form).  Then only
 * the transpose is a compiled network and ls_amd_test_rep_by_cosets mirrors THAT routine (LS_AMD_K4=cosets: mode 4). */
int ls_amd_test_d4_mask(ls_hs_basis const *basis);
/* Host-only test hooks of the static index table {representative -> 32-bit payload} of the indexed pull mode
 * (distributed-matvec_amd/csrc/lsk.h: lsk_gtab): bucket bits for n keys of L bits (-1: no admissible shape), a sequential
 * build with the device kernel's placement rule into a malloc'ed array of 2 << bbits entries (release with
 * ls_amd_test_free; payload NULL: the key's position), and the lookup (payload, or -1 when absent) */
int ls_amd_test_gtab_bits(int L, int64_t n);
int ls_amd_test_gtab_build(int L, int bbits, int64_t n, uint64_t const *reps, uint32_t const *payload, uint64_t **entries);
int64_t ls_amd_test_gtab_find(int L, int bbits, uint64_t const *entries, uint64_t key);
/* byte offsets of commInfo / globalSumReal_type inside primme_params as the PRIMME callbacks read them (ls_chpl.h) */
int ls_amd_test_primme_comminfo_offset(void);
int ls_amd_test_primme_sumtype_offset(void);
int ls_amd_test_primme_nlocal_offset(void);
int ls_amd_test_primme_matrix_offset(void);

#ifdef __cplusplus
}
#endif
#endif /* LS_AMD_H */
