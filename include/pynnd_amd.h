/*
 * pynnd_amd.h -- C ABI of the MI355X (gfx950) NN-Descent index builder.
 *
 * This is the drop-in boundary for the BUILD path of lmcinnes/pynndescent 0.6.0.
 * The reference has no FFI of its own (it is Python + numba); the boundary is the
 * two call sites inside NNDescent.__init__ (reference pynndescent/pynndescent_.py):
 *
 *     make_forest(...) + rptree_leaf_array(...)      pynndescent_.py:1118-1130
 *     nn_descent(data, n_neighbors, rng_state, max_candidates, dist, n_iters,
 *                delta, init_graph, rp_tree_init, leaf_array, ...)
 *                                                    pynndescent_.py:1247-1260
 *
 * Each entry point below names the reference function(s) it replaces.  All
 * pointers are plain host or device pointers, all sizes are explicit, no C++ or
 * torch types cross the boundary.  INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - Every function returns 0 on success, non-zero on failure; the message is
 *     available from nnd_last_error(handle) (or nnd_last_global_error() when no
 *     handle exists yet).  There is NO CPU fallback: without a gfx950 device
 *     nnd_create fails.
 *   - "alt space": distances are squared-euclidean (metric 0) or
 *     log2(|x||y|/<x,y>) (metric 1), exactly what the reference keeps in
 *     NNDescent._neighbor_graph (distances.py:63-91, 583-630); the caller applies
 *     sqrt / 1-2^-d itself (distances.py:2170-2173), as NNDescent.neighbor_graph does.
 *   - Output rows are ascending in distance; missing entries are (-1, +inf) at
 *     the row tail (utils.py:130-158, 189-218).
 *   - One handle = one GPU = one HIP stream.  Calls on one handle must be
 *     serialised by the caller; distinct handles are independent.
 */
#ifndef PYNND_AMD_H
#define PYNND_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NND_METRIC_SQEUCLIDEAN 0 /* reference distances.py:63  squared_euclidean  */
#define NND_METRIC_ALT_COSINE 1  /* reference distances.py:583 alternative_cosine */

#define NND_ABI_VERSION 6 /* 6 (round 6): nnd_stats grew join_substeps[] and nnd_shard_info grew gather_bytes[] / gather_section[] at their ends; nnd_host_alloc / nnd_host_free; 5: nnd_search_graph / nnd_search_graph_fetch */

typedef struct nnd_handle_s *nnd_handle_t;

/* Build parameters; mirrors the NNDescent ctor kwargs that reach the build path
 * (pynndescent_.py:976-1007) after the ctor's defaulting (pynndescent_.py:1009-1012,
 * 1135-1138; rp_trees.py:2845-2846).  The host code does the defaulting. */
typedef struct nnd_params {
    int64_t n;              /* points */
    int32_t dim;            /* features */
    int32_t metric;         /* NND_METRIC_* */
    int32_t n_neighbors;    /* k, 1..256 (rows above 64 entries are merged through LDS: correct, not tuned) */
    int32_t n_trees;        /* 0 = no RP-forest initialisation */
    int32_t leaf_size;      /* > 0 */
    int32_t max_depth;      /* max_rptree_depth (pynndescent_.py:1000) */
    int32_t max_candidates; /* 1..128 (above 64: five passes of the 64-slot join over blocks of the candidate lists) */
    int32_t n_iters;
    float delta;            /* stop when c <= delta*k*n (pynndescent_.py:317) */
    int64_t rng_state[3];   /* NNDescent.rng_state (pynndescent_.py:1105-1107) */
    int64_t tree_rng[3];    /* first row of make_forest's per-tree draw (rp_trees.py:2850) */
    int32_t device;         /* HIP device ordinal */
    int32_t join_blocks;    /* descent sub-steps per iteration; reference blocks by 16384 vertices (pynndescent_.py:279).  0 = chosen by the
                             * library: 1 up to 64 neighbours, ceil(k / 32) above (a row takes at most 64 updates per sub-step) */
    int32_t flags;          /* NND_FLAG_*; 0 for a build handle */
    int32_t reserved[5];
} nnd_params;

/* Auxiliary handles (the pruning pass and the hub search tree of NNDescent.prepare() run on a handle of their own):
 * NND_FLAG_NO_GRAPH skips the k-lists, candidate / proposal / reverse-offer tables and the routing-forest tables -- a
 * handle for nnd_diversify_*_host / nnd_degree_prune_host / nnd_hub_tree_*; every build entry point fails on it.
 * NND_FLAG_NO_PREP additionally skips the prepared (padded, centred / normalised) copy of the rows: hub tree only. */
#define NND_FLAG_NO_GRAPH 1
#define NND_FLAG_NO_PREP 2
/* test hook: candidate selection by the one-wave-per-vertex kernel where the two-vertices-per-wave kernel would run
 * (tests/test_gpu_kernels.py proves the two produce identical lists) */
#define NND_FLAG_TEST_SELECT_WAVE 4
/* test hook (row-sharded build): proposal regions of ONE record per destination row instead of 32, so that the deferral
 * path (records that do not fit stay in the sender's table and travel with the next iteration's) is exercised at test
 * sizes (tests/test_gpu_sharded.py) */
#define NND_FLAG_TEST_SMALL_REGIONS 8
/* test hook: the rp forest's routing pass as ONE walk per (tree, point) through global memory (the round-2 kernel)
 * instead of the two coherent passes; the two assign every point to the same cell (tests/test_gpu_kernels.py) */
#define NND_FLAG_TEST_ROUTE_PLAIN 16
/* test hooks of the row-sharded build (tests/test_gpu_sharded.py): the forest split by tree (the round-3 scheme) where it
 * would be sharded by cell; a rank that FAILS at the start of its second iteration (error return: the other ranks must
 * return an error too, promptly); a rank that VANISHES there (returns without telling anybody, as a killed process
 * would: the other ranks must give up after the communicator's timeout) */
#define NND_FLAG_TEST_FOREST_BY_TREE 32
#define NND_FLAG_TEST_FAIL 64
#define NND_FLAG_TEST_VANISH 128
/* test hook: the reverse offers of the candidate sampling through the round-1..4 kernel (one device-scope atomicMin per
 * edge into 32 hashed slots per bank) instead of the bucketed transposition (sample.hip): comparison / timing */
#define NND_FLAG_TEST_SAMPLE_ATOMIC 256
/* test hook (row-sharded build): a rank pretends that the forest sharded by cell cannot be built on this data -- at the
 * tree tops (512), at the owners' shares (1024) or at the over-long cells (512 | 1024); the ranks must agree and build
 * the forest split by tree instead (tests/test_gpu_sharded.py) */
#define NND_FLAG_TEST_FOREST_FALLBACK_TOPS 512
#define NND_FLAG_TEST_FOREST_FALLBACK_SHARE 1024
#define NND_TEST_FALLBACK_AT(flags) ((((flags) & 512) ? 1 : 0) + (((flags) & 1024) ? 2 : 0))
/* test hook: the candidate sampling behaves as if the record regions of the bucketed transposition could not be allocated
 * (wide rows on a full device): the handle must switch to the hashed slots and build, not fail (tests/test_gpu_kernels.py) */
#define NND_FLAG_TEST_SAMPLE_NOMEM 2048
/* test hook: the fused candidate selection with 32 lanes per vertex where rows and candidate lists of at most 16 entries would
 * take 16 (four vertices per wave): the two forms must write identical lists (tests/test_gpu_kernels.py) */
#define NND_FLAG_TEST_SELECT_HALF 4096
/* test hook (row-sharded build): the per-iteration threshold / neighbour-id all-gather on the BUILD's channel, in front of the offer
 * exchange (rounds 3-5), where it would run on the second channel beside the sampling: same graph either way (tests/test_gpu_sharded.py) */
#define NND_FLAG_TEST_GATHER_INLINE 8192
/* test hook: the local join of 17..32 candidates per class (k_local_join_w) reads the neighbour lists of its membership tests from
 * global memory (rounds 3-5) where it would stage them in LDS (round 6, rows of <= 32 neighbours on one GPU): the two forms must
 * build the same graph, entry for entry (tests/test_gpu_kernels.py) */
#define NND_FLAG_TEST_JOIN_UNSTAGED 16384

/* Run-time statistics for measurement (bench.py roofline). */
typedef struct nnd_stats {
    int64_t n_iters_run;
    int64_t n_leaves;
    int64_t tree_levels;
    int64_t leaf_pairs;          /* pair distances evaluated during leaf seeding */
    int64_t leaf_rows;           /* point rows loaded by the leaf kernel */
    int64_t join_pairs[64];      /* P_i: pair distances evaluated, per iteration */
    int64_t join_rows[64];       /* C_i: candidate rows gathered, per iteration */
    int64_t join_active[64];     /* vertices with >=1 new candidate, per iteration */
    int64_t proposals[64];       /* proposals emitted by the join, per iteration */
    int64_t updates[64];         /* c: accepted k-list insertions, per iteration */
    float ms_prep, ms_forest, ms_leaf_init, ms_random_init, ms_descent, ms_finalize;
    float ms_sample[64], ms_join[64], ms_merge[64];
    int64_t join_mfma[64];       /* v_mfma_f32_16x16x4_f32 instructions issued by the join (2048 flop each), per iteration */
    int64_t leaf_mfma;           /* the same for the leaf-seeding kernel, whole stage */
    int64_t n_cells;             /* rp forest: cells of the routing pass (0 = whole-set level-synchronous build) */
    int64_t join_substeps[64];   /* ABI 6: join + merge sub-steps of every iteration (join_blocks, or the library's schedule when join_blocks = 0) */
} nnd_stats;

int32_t nnd_abi_version(void);
const char *nnd_last_global_error(void);
const char *nnd_last_error(nnd_handle_t h);

/* Create a builder on params->device and allocate its HBM state (k-lists,
 * candidate lists, proposal buffers).  Fails if the device is not gfx950. */
int32_t nnd_create(nnd_handle_t *out, const nnd_params *params);
/* nnd_destroy PARKS one plain handle instead of releasing it (every hipFree synchronises the device: ~9 ms for the state
 * of a 1 M-point build, and as much again for the next hipMalloc): an nnd_create of the same geometry re-arms it.  A
 * create of another geometry, an allocation that fails, or nnd_release_pending() releases the parked handle's HBM. */
int32_t nnd_destroy(nnd_handle_t h);
int32_t nnd_release_pending(void);

/* Host-side helpers for the result arrays (NNDescent.neighbor_graph returns a COPY of the ids and the corrected
 * distances, pynndescent_.py:2145-2158): a fresh 60 MB destination costs ~15 k first-touch page faults, which a
 * single-threaded numpy copy / sqrt pays one after the other (18 + 14 ms at 1 M x 15).  These split the range over a few
 * host threads.  nnd_host_sqrt_f32 is IEEE sqrtf per element: bit-identical to numpy.sqrt on float32. */
int32_t nnd_host_copy(void *dst, const void *src, int64_t bytes);
int32_t nnd_host_sqrt_f32(float *dst, const float *src, int64_t count);
/* Pinned host memory for result arrays (round 6).  The reference returns freshly allocated numpy arrays
 * (pynndescent_.py:2145-2158, 1247-1260); at 1 M x 15 their first touch cost more than the device-to-host copy itself.  The host
 * side of the drop-in class keeps a small pool of these buffers (pynndescent_amd/_capi.py HostPool): a buffer goes back to the
 * pool when the last numpy view of it dies.  nnd_host_alloc returns NULL without a device or without memory (callers fall back
 * to ordinary allocations); the finalize / build entry points recognise a pinned destination and copy into it with one DMA. */
void *nnd_host_alloc(int64_t bytes);
int32_t nnd_host_free(void *p);

/* Point set, float32 C-contiguous (n, dim) -- NNDescent._raw_data (pynndescent_.py:1054-1057).
 * Host variant copies H2D; device variant BORROWS the pointer (it must outlive the handle's
 * build calls).  Both then run the prep kernel (pad to 32 floats, centre / L2-normalise, norms)
 * on the handle's stream, immediately: the device buffer must be COMPLETE when the call is made --
 * synchronise the stream that produced it first, or make the handle run on that stream
 * (nnd_set_stream).  The prepared copy is made once per call, not once per build. */
int32_t nnd_set_data_host(nnd_handle_t h, const float *x);
int32_t nnd_set_data_device(nnd_handle_t h, const float *x_dev);
/* *out = 1 when the point set held a NaN or an infinity (seen by the prep kernel while it read the rows).  The reference
 * rejects such input in check_array (pynndescent_.py:1054) with a scan of its own; the host mirror raises the same
 * error from this flag instead of scanning 488 MB on one core (24 ms at 1 M x 128). */
int32_t nnd_data_nonfinite(nnd_handle_t h, int32_t *out);

/* make_forest (rp_trees.py:2815-2888): builds all n_trees trees level-synchronously on device. */
int32_t nnd_make_forest(nnd_handle_t h);
/* rptree_leaf_array (rp_trees.py:2891-2922): shape query then copy, int32 (n_leaves, max_leaf_size), -1 padded. */
int32_t nnd_leaf_array_shape(nnd_handle_t h, int64_t *n_leaves, int32_t *max_leaf_size);
int32_t nnd_get_leaf_array(nnd_handle_t h, int32_t *out_host);

/* make_heap (utils.py:130-158): reset the k-lists to (-1, +inf, 0). */
int32_t nnd_reset_graph(nnd_handle_t h);
/* init_rp_tree + generate_leaf_updates (pynndescent_.py:73-185): all-pairs inside every leaf. */
int32_t nnd_init_from_leaves(nnd_handle_t h);
/* The same on a CALLER-PROVIDED leaf array: the `leaf_array` argument of the reference's nn_descent
 * (pynndescent_.py:324-337) -- host int32 (n_leaves, max_leaf_size), -1 padded, as rptree_leaf_array returns it
 * (rp_trees.py:2891-2922); a row ends at its first negative entry (pynndescent_.py:88-92).  A caller that keeps the
 * reference's make_forest hands its leaves in here; the handle needs no forest of its own (n_trees may be 0). */
int32_t nnd_init_from_leaf_array(nnd_handle_t h, const int32_t *leaf_array, int64_t n_leaves, int32_t max_leaf_size);
/* init_random (pynndescent_.py:188-203): top up rows that are not full with random points. */
int32_t nnd_init_random(nnd_handle_t h);
/* initalize_heap_from_graph_indices[_and_distances] (utils.py:836-860), used for init_graph /
 * init_dist (pynndescent_.py:1225-1242).  init_dist may be NULL. Host pointers, (n, width), width <= 256. */
int32_t nnd_init_from_graph(nnd_handle_t h, const int32_t *init_idx, const float *init_dist, int32_t width);
/* init_from_neighbor_graph (pynndescent_.py:206-214), the warm start of NNDescent.update
 * (pynndescent_.py:2512-2517): the entries of an existing graph -- host (n, width) indices (-1 = none) and
 * alt-space distances (required) -- are inserted with flag 0 ("old").  Call after nnd_reset_graph. */
int32_t nnd_init_from_neighbor_graph(nnd_handle_t h, const int32_t *init_idx, const float *init_dist, int32_t width);

/* One iteration of nn_descent_internal (pynndescent_.py:296-320):
 * new_build_candidates (utils.py:221-320) + generate_graph_update_array (utils.py:536-658)
 * + apply_graph_update_array (utils.py:661-733).  *c_out = number of k-list insertions. */
int32_t nnd_descent_iter(nnd_handle_t h, int64_t *c_out);
/* The whole loop with the reference's stop rule (pynndescent_.py:317). */
int32_t nnd_descent(nnd_handle_t h);

/* deheap_sort (utils.py:189-218) + exact (f64-accumulated) recomputation of the n*k distances.
 * Writes int32 (n,k) indices and float32 (n,k) alt-space distances. */
int32_t nnd_finalize_host(nnd_handle_t h, int32_t *out_idx, float *out_dist);
int32_t nnd_finalize_device(nnd_handle_t h, int32_t *out_idx_dev, float *out_dist_dev);

/* Everything between "data is resident" and "graph is resident":
 * forest -> leaf init -> random init -> descent -> finalize (device outputs). */
int32_t nnd_build_device(nnd_handle_t h, int32_t *out_idx_dev, float *out_dist_dev);

/* One-shot host-buffer build == make_forest + rptree_leaf_array + nn_descent of the reference
 * (pynndescent_.py:1118-1130, 1247-1260).  init_idx/init_dist nullable (init_graph path). */
int32_t nnd_build(const nnd_params *params, const float *x, const int32_t *init_idx, const float *init_dist,
                  int32_t init_width, int32_t *out_idx, float *out_dist, nnd_stats *stats, char *err,
                  int32_t errlen);

int32_t nnd_get_stats(nnd_handle_t h, nnd_stats *out);
int32_t nnd_synchronize(nnd_handle_t h);

/* ---- introspection used by the parity tests (device state -> host) ---- */
/* Current k-lists, unsorted-by-contract but kept ascending: int32 idx (-1 empty), f32 dist, u8 new-flag. */
int32_t nnd_get_graph(nnd_handle_t h, int32_t *idx, float *dist, uint8_t *flags);
/* Candidate lists of the last nnd_descent_iter: int32 (n, max_candidates) each, -1 padded. */
int32_t nnd_get_candidates(nnd_handle_t h, int32_t *new_idx, int32_t *old_idx);
/* Sampling only (no join): fills the candidate lists and clears sampled new-flags. */
int32_t nnd_sample_candidates(nnd_handle_t h);
/* Pairwise alt-space distances between listed rows, computed by the same MFMA Gram tile code
 * the join uses: out (na, nb) float32 host. */
int32_t nnd_pairwise_gram(nnd_handle_t h, const int32_t *rows_a, int32_t na, const int32_t *rows_b, int32_t nb,
                          float *out);

/* ---- row-sharded multi-GPU build ----
 * The reference is single-process; its sharding idea is the owner-computes rule of apply_graph_update_array /
 * new_build_candidates / init_rp_tree (utils.py:709-731, 259-306; pynndescent_.py:154-185: each thread owns a contiguous
 * vertex range) and its knob is n_jobs (pynndescent_.py:1141-1143).  Here the same rule crosses GPUs: rank r owns rows
 * [lo_r, hi_r) of the k-lists; the point set is replicated once (all-gather over xGMI, on a second channel behind the
 * forest's first steps: candidate vectors never travel again); the forest is sharded BY CELL -- rank r builds the top of
 * its share of the trees on the global sample, the packed tops are all-gathered, every rank routes ITS rows through ALL
 * trees, the (cell, row) pairs go to the rank that owns the cell (1 / G of every tree's cells), which finishes the cells
 * and seeds the k-lists from their leaves; partial k-list rows go to their owners and are merged there (small point sets:
 * split by tree).  The forest is the single-GPU forest whatever the number of ranks.  Per NN-descent iteration
 *   (1) all-gather of the thresholds, 4 bytes per row, and -- while the lists still change much -- of the neighbour ids;
 *   (2) reverse-offer all-to-all-v of 8-byte records (the cross-process form of the ownership test utils.py:266-273);
 *   (3) local sampling + join of the owned vertices; (4) proposal all-to-all-v of 12-byte records, owner-side merge
 *   (utils.py:721-731); the update counts of all ranks (stop rule, pynndescent_.py:317) ride on the record-count exchange.
 * The whole per-rank build runs inside the library (csrc/shard.hip) on one HIP stream; the exchanges are issued from
 * the C side on that stream -- ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd (RCCL) -- so kernels and collectives
 * need no host synchronisation between them; the host waits twice per iteration, for the record counts.
 *
 * Communicators.  One per rank.  RCCL: rank 0 calls nnd_comm_unique_id, the launcher hands the 128 bytes to the other
 * ranks (any side channel), every rank calls nnd_comm_create_rccl.  LOCAL: world ranks as threads of ONE process that
 * may share a GPU (tests, nnd_build_multi with repeated device ids).  HOST: device buffers staged through pinned host
 * memory and exchanged by a caller-supplied callback (debugging transport; gloo in the tests). */
typedef struct nnd_comm_s *nnd_comm_t;
#define NND_COMM_ID_BYTES 128
int32_t nnd_comm_unique_id(void *id_out /* NND_COMM_ID_BYTES */);
int32_t nnd_comm_create_rccl(nnd_comm_t *out, const void *id, int32_t world, int32_t rank, int32_t device);
int32_t nnd_comm_create_local(nnd_comm_t *out /* [world] */, int32_t world, const int32_t *devices /* [world], NULL: all 0 */);
/* all-to-all-v on HOST memory: byte segment [send_off[d], + send_bytes[d]) of `send` goes to rank d; what rank s sends
 * arrives at recv + recv_off[s] (recv_bytes[s] bytes, known to the receiver).  Returns 0 on success. */
typedef int32_t (*nnd_host_exchange_fn)(void *user, const void *send, const int64_t *send_off, const int64_t *send_bytes,
                                        void *recv, const int64_t *recv_off, const int64_t *recv_bytes);
int32_t nnd_comm_create_host(nnd_comm_t *out, int32_t world, int32_t rank, int32_t device, nnd_host_exchange_fn fn, void *user);
/* Second channel of an RCCL communicator (its own ncclComm_t from a second unique id, its own stream): the point-set
 * all-gather runs on it while the build's stream works.  Collective: every rank calls it (or none).  LOCAL communicators
 * are created with theirs; HOST has none (the transfer then runs on the build's channel). */
int32_t nnd_comm_add_channel_rccl(nnd_comm_t c, const void *id /* a second id from nnd_comm_unique_id */);
/* Failure handling.  Every host wait of a build polls the communicator: a rank whose peer failed (ranks of one process
 * share a flag: LOCAL, nnd_build_multi) or that has waited longer than the timeout (default 120 s; the only signal
 * between processes) gives up -- ncclCommAbort on its communicators, error return -- instead of blocking in a collective
 * for ever.  nnd_comm_abort makes THIS rank give up (and tells the ranks of its process). */
int32_t nnd_comm_set_timeout(nnd_comm_t c, int64_t timeout_ms);
int32_t nnd_comm_destroy(nnd_comm_t c);
int32_t nnd_comm_abort(nnd_comm_t c);
/* out[0] transport (1 RCCL, 2 LOCAL, 3 HOST), out[1] ranks, out[2] ncclGetVersion() (0 unless RCCL), out[3] second channel present */
int32_t nnd_comm_info(nnd_comm_t c, int32_t *out /* [4] */);
/* Collective (every rank of the world calls it): each rank sends its rank number to every other rank through the transport's
 * ordinary exchange and checks what arrived -- the exchange nnd_comm_create_rccl / nnd_comm_add_channel_rccl end with.
 * RCCL and LOCAL transports; 0 = every pair of ranks moved its bytes (both channels). */
int32_t nnd_comm_self_test(nnd_comm_t c);
/* LOCAL only: compute sections of the ranks run one at a time (per-rank timings on a shared GPU; tools/rank_critical_path.py) */
int32_t nnd_comm_local_set_serial(nnd_comm_t c, int32_t on);
const char *nnd_comm_last_error(nnd_comm_t c /* NULL: the error of a failed create */);

/* One rank of a sharded build.  params: n = the GLOBAL point count, n_trees = the GLOBAL tree count, device = this
 * rank's GPU, everything else as for nnd_create (the reference's derived defaults are taken on the global n by the
 * host).  shard_sizes[world]: rows per rank, in rank order (rank r owns the rows after those of ranks < r). */
typedef struct nnd_shard_s *nnd_shard_t;
typedef struct nnd_shard_info {
    int64_t n_total, own_lo, own_hi;
    int32_t world, rank, local_trees, iters;
    int64_t c[64];                /* GLOBAL update count per iteration (the stop rule's c, pynndescent_.py:317) */
    int64_t offer_records[64];    /* reverse-offer records this rank sent to other ranks, per iteration */
    int64_t proposal_records[64]; /* proposal records this rank sent to other ranks, per iteration */
    int64_t deferred[64];         /* proposal records that did not fit their destination's region and were kept for the next iteration */
    int64_t dropped_offers;       /* must stay 0: the offer regions are sized for every owned edge */
    int64_t bytes_sent;           /* payload bytes this rank sent to other ranks during the build */
    float ms_total;               /* this rank's wall time of the last build, exchanges included */
    float ms_allgather, ms_klist_exchange; /* stream time of the two bulk exchanges */
    int32_t n_sections;           /* compute sections of the last build (timeline below) */
    float section_ms[256];        /* LOCAL serial mode: GPU time of each compute section of this rank, in program order */
    int64_t section_bytes[256];   /* payload bytes this rank sent in the exchange that FOLLOWS the section */
    int32_t forest_by_cell;       /* 1: forest sharded by cell, 0: split by tree */
    int32_t n_sections_overlap;   /* the first sections of the build that need only this rank's rows: the point-set all-gather runs beside them */
    int64_t forest_positions;     /* point-trees this rank finished and seeded (forest by cell: ~ n_trees * n / ranks) */
    /* ABI 6: the threshold / neighbour-id all-gather of every iteration runs on the SECOND channel beside the offer exchange and the
     * second half of the sampling (only the join needs what it brings).  gather_bytes[i]: payload this rank sent in it;
     * gather_section[i]: index of the compute section it runs beside (LOCAL serial mode times that section apart; -1: the gather ran on
     * the build's channel, in front of the offer exchange, as it did through round 5) */
    int64_t gather_bytes[64];
    int32_t gather_section[64];
} nnd_shard_info;
int32_t nnd_shard_create(nnd_shard_t *out, const nnd_params *params, nnd_comm_t comm, const int64_t *shard_sizes);
/* x_local_dev: this rank's rows, float32 (n_local, dim) on its GPU, complete when the call is made (or produced on
 * x_stream, which the build then waits for).  Outputs: device buffers (n_local, k): GLOBAL neighbour ids, alt-space
 * distances, rows ascending.  Blocks until this rank's rows are final. */
int32_t nnd_shard_build(nnd_shard_t s, const float *x_local_dev, void *x_stream, int32_t *out_idx_dev, float *out_dist_dev);
/* ... from the rank's OWN rows of an init graph (device (n_own, init_width), global ids; init_dist_dev nullable); the shard must have been
 * created with n_trees = 0 */
int32_t nnd_shard_build_from_graph(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *init_idx_dev,
                                   const float *init_dist_dev, int32_t init_width, int32_t *out_idx_dev, float *out_dist_dev);
/* ... NNDescent.update() (pynndescent_.py:2498-2535): the rank's rows of the previous graph enter as OLD entries (alt-space distances
 * required) before the fresh forest's leaves are joined; no random fill */
int32_t nnd_shard_build_update(nnd_shard_t s, const float *x_local_dev, void *x_stream, const int32_t *old_idx_dev, const float *old_dist_dev,
                               int32_t width, int32_t *out_idx_dev, float *out_dist_dev);
int32_t nnd_shard_get_info(nnd_shard_t s, nnd_shard_info *out);
int32_t nnd_shard_get_stats(nnd_shard_t s, nnd_stats *out); /* this rank's kernels */
/* the rank's builder handle (tests: nnd_leaf_array_shape / nnd_get_leaf_array give the leaves this rank seeded from) */
nnd_handle_t nnd_shard_handle(nnd_shard_t s);
int32_t nnd_shard_destroy(nnd_shard_t s);
const char *nnd_shard_last_error(nnd_shard_t s /* NULL: the error of a failed create */);

/* nnd_build over n_devices GPUs of this node: host buffers in, host buffers out, one host thread per GPU inside the
 * library (the reference's analogue is n_jobs, pynndescent_.py:1141-1143).  devices: HIP ordinals, NULL = 0..n_devices-1.
 * Distinct ordinals talk over RCCL; a list that repeats an ordinal (several ranks on one GPU: tests on a one-GPU box)
 * uses the LOCAL transport.  Results depend on n_devices as the reference's depend on its thread count.  stats: rank 0's. */
int32_t nnd_build_multi(const nnd_params *params, const float *x, int32_t n_devices, const int32_t *devices, int32_t *out_idx,
                        float *out_dist, nnd_stats *stats, nnd_shard_info *info_rank0 /* nullable */, char *err, int32_t errlen);
/* ... from an init graph (ABI 5): the warm start of NNDescent(init_graph=, init_dist=), pynndescent_.py:1225-1242 /
 * initalize_heap_from_graph_indices[_and_distances], utils.py:836-860 -- init_idx host (n, init_width) GLOBAL ids (-1: empty),
 * init_dist nullable (alt-space distances; computed when NULL).  No forest (an init graph disables it, pynndescent_.py:1059-1062:
 * params->n_trees is taken as 0), no random fill.  Every rank seeds ITS rows from its rows of the init graph. */
int32_t nnd_build_multi_from_graph(const nnd_params *params, const float *x, int32_t n_devices, const int32_t *devices,
                                   const int32_t *init_idx, const float *init_dist, int32_t init_width, int32_t *out_idx, float *out_dist,
                                   nnd_stats *stats, nnd_shard_info *info_rank0 /* nullable */, char *err, int32_t errlen);
/* NNDescent.update() (pynndescent_.py:2381-2553, the rebuild at 2498-2535) over n_devices GPUs: a fresh forest of params->n_trees trees
 * (n_trees_after_update) over the changed point set, the previous graph's surviving entries as OLD entries (init_from_neighbor_graph,
 * pynndescent_.py:206-214; old_idx host (n, width) global ids, -1 where invalidated, old_dist their alt-space distances), no random fill */
int32_t nnd_build_multi_update(const nnd_params *params, const float *x, int32_t n_devices, const int32_t *devices, const int32_t *old_idx,
                               const float *old_dist, int32_t width, int32_t *out_idx, float *out_dist, nnd_stats *stats,
                               nnd_shard_info *info_rank0 /* nullable */, char *err, int32_t errlen);

/* Stream the handle runs on: the caller's HIP stream (e.g. the one that produces the point set) instead of the handle's
 * own, so the device-pointer entry points need no synchronisation with it.  NULL: the handle's own stream again. */
int32_t nnd_set_stream(nnd_handle_t h, void *hip_stream);
/* one descent iteration in two steps (the state between them is what nnd_reset_graph has to clear: tests) */
int32_t nnd_descent_sample(nnd_handle_t h);
int32_t nnd_descent_join(nnd_handle_t h);

/* ---- search-graph pruning pass (BASELINE config 5; reference NNDescent._init_search_graph, pynndescent_.py:1451-1611) ----
 * Host arrays in / out: like the reference, the conversions between these kernels (COO->CSR, transpose, maximum,
 * binarise) are scipy calls on the host. */
typedef struct nnd_prune_opts {
    float prune_probability; /* diversify_prob: an eligible edge is pruned with this probability (pynndescent_.py:388, 582) */
    int32_t degree_aware;    /* 0: diversify / diversify_csr; 1: diversify_degree_aware / diversify_csr_degree_aware */
    int32_t max_degree;      /* degree-aware: the degree above which the threshold is relaxed (pynndescent_.py:1478, 1567) */
    float aggressiveness;    /* degree-aware: degree_prune_aggressiveness */
    float alpha;             /* degree-aware forward pass only (pynndescent_.py:435, 528) */
    uint32_t seed;           /* coin hash seed (the reference draws from NNDescent.rng_state) */
    int32_t reserved[2];
} nnd_prune_opts;
/* diversify (pynndescent_.py:369-403) / diversify_degree_aware (433-546): (n,k) rows ascending in alt space; pruned
 * slots -> (-1, +inf).  opts NULL = the reference defaults (standard method, probability 1).  degree: host int32 (n)
 * undirected degrees (compute_degrees, pynndescent_.py:406-418), required when opts->degree_aware. */
int32_t nnd_diversify_host(nnd_handle_t h, int32_t *idx, float *dist, const nnd_prune_opts *opts, const int32_t *degree);
/* diversify_csr (pynndescent_.py:549-588) / diversify_csr_degree_aware (625-726) on CSR rows of <= 64 entries; pruned
 * entries get weight 0.  degree: host int32 (n) (compute_degrees_csr, pynndescent_.py:591-622) when degree_aware. */
int32_t nnd_diversify_csr_host(nnd_handle_t h, const int32_t *indptr, const int32_t *indices, float *data, int64_t nnz,
                               const nnd_prune_opts *opts, const int32_t *degree);
/* degree_prune_internal (pynndescent_.py:728-738): rows longer than max_degree keep entries <= sorted(row)[max_degree] */
int32_t nnd_degree_prune_host(nnd_handle_t h, const int32_t *indptr, float *data, int64_t nnz, int32_t max_degree);

/* The whole pruning pass of NNDescent._init_search_graph (pynndescent_.py:1451-1611) on the device (csrc/searchgraph.hip):
 * forward diversify -> COO -> CSR -> "reverse" diversify_csr on the shared arrays -> union max(F', F'^T) -> diagonal and zeros
 * dropped -> degree_prune to round(pruning_degree_multiplier * n_neighbors) -> binarise; ONE device-to-host copy at the end.
 * Replaces, besides the three numba kernels above, the scipy glue between them (coo_matrix / tocsr 1527-1537, transpose 1549,
 * maximum 1599, setdiag / eliminate_zeros 1602-1604, `!= 0` 1611).
 * idx / dist: the (n, k) neighbour graph (k = the handle's n_neighbors; rows ascending in alt space, -1 / +inf padded), on the
 * host (on_device = 0) or on the handle's device (1); not modified.  The handle must hold the point set (nnd_set_data_*): an
 * NND_FLAG_NO_GRAPH handle, or the handle that built the graph (no second upload of the rows).  fwd_rows / fwd_dist: optional
 * host (n, k) arrays that receive the graph after the forward pass (the `stages` view of the tests), or NULL. */
typedef struct nnd_search_graph_stats {
    int64_t forward_nnz;   /* edges after the forward pass */
    int64_t reverse_nnz;   /* ... after the second pass (the reference's reverse_graph.nnz; = the forward matrix's, shared arrays) */
    int64_t union_nnz;     /* entries of max(F', F'^T) without the diagonal */
    int64_t final_nnz;     /* entries of the search graph */
    float min_distance;    /* NNDescent._min_distance (pynndescent_.py:1539) */
    int32_t max_degree_out;
    float ms_device;       /* stream time of the pass, copies of the graph in included */
    int32_t reserved[3];
} nnd_search_graph_stats;
int32_t nnd_search_graph(nnd_handle_t h, const int32_t *idx, const float *dist, int32_t on_device, int32_t n_neighbors,
                         float pruning_degree_multiplier, float diversify_prob, int32_t degree_aware, float degree_prune_aggressiveness,
                         uint32_t seed, int32_t *fwd_rows_host, float *fwd_dist_host, nnd_search_graph_stats *stats);
/* the finished search graph of the last nnd_search_graph call: CSR pattern, indptr (n + 1), indices (final_nnz) sorted by column */
int32_t nnd_search_graph_fetch(nnd_handle_t h, int32_t *indptr_host, int32_t *indices_host);

/* ---- hub search tree of NNDescent.prepare() (reference rp_trees.py:714-1312 make_hub_tree and its splits,
 * rp_trees.py:2926-3049 convert_tree_format) ----
 * Built level-synchronously on the device from the ORIGINAL rows (nnd_set_data_*; the handle must have been created
 * with n_trees >= 1: the builder borrows the forest's scan / scatter buffers) and the finished graph's in-degrees:
 * rank_order = the point ids sorted by (-in-degree, id) (compute_global_degrees, rp_trees.py:714-744, is a bincount of
 * the neighbour array; the order is host glue).  Angular splits when the handle's metric is NND_METRIC_ALT_COSINE.
 * Result: the reference's FlatTree in pre-order numbering: hyperplanes (n_nodes, dim), offsets (n_nodes),
 * children (n_nodes, 2) [internal: child node ids; leaf: (-leaf_start, -leaf_end) into indices], indices (n). */
int32_t nnd_hub_tree_build(nnd_handle_t h, const int32_t *rank_order_host, int32_t leaf_size, int32_t max_depth,
                           int64_t *n_nodes_out);
int32_t nnd_hub_tree_fetch(nnd_handle_t h, float *hyperplanes, float *offsets, int32_t *children, int32_t *indices,
                           int32_t *max_leaf_size);

/* ---- batched queries against a prepared index (reference NNDescent.query, pynndescent_.py:2275-2379: the search
 * closure of _init_search_function 1793-1883, select_side / search_flat_tree rp_trees.py:2662-2741, deheap_sort) ----
 * The searcher owns device copies of what the reference's closure captures: the (reordered) raw data, the CSR search
 * graph, the FlatTree of the search forest's first tree (n_nodes = 0: no tree, random starts only), min_distance and
 * n_neighbors.  All pointers are HOST pointers.  One wave per query; k <= 256 (the query's k, not the index's; above 64 the result list is two or four entries per lane).  Output rows ascending in the
 * alternative distance space, vertex numbers in the searcher's (reordered) numbering; unfilled slots (-1, +inf). */
typedef struct nnd_searcher_s *nnd_searcher_t;
int32_t nnd_searcher_create(nnd_searcher_t *out, int32_t device, int64_t n, int32_t dim, int32_t metric, const float *data,
                            const int32_t *indptr, const int32_t *indices, int64_t nnz, const float *hyperplanes,
                            const float *offsets, const int32_t *children, const int32_t *tree_indices, int64_t n_nodes,
                            float min_distance, int32_t n_neighbors, const int64_t *search_rng_state /* 3 */);
int32_t nnd_searcher_query(nnd_searcher_t s, const float *queries /* (nq, dim) */, int64_t nq, int32_t k, float epsilon,
                           int32_t *out_idx /* (nq, k) */, float *out_dist /* (nq, k) */);
/* The search runs in two tiers: per-query LDS structures (3400 visited vertices, 512 frontier entries) and, for exactly
 * the queries that would overflow them (large k / epsilon, oversized tree leaves), a global-memory tier with the
 * reference's own structures (visited bitset over all n points, utils.py:323-349; frontier of 65536 entries).  No answer
 * comes from a truncated search.  nnd_searcher_last_spilled: queries of the last call that ran on the second tier;
 * nnd_searcher_set_tier(s, 1) sends every query there (tests), 0 = automatic. */
int64_t nnd_searcher_last_spilled(nnd_searcher_t s);
int32_t nnd_searcher_set_tier(nnd_searcher_t s, int32_t tier);
int32_t nnd_searcher_destroy(nnd_searcher_t s);
const char *nnd_searcher_last_error(nnd_searcher_t s /* NULL: the error of a failed create */);

#ifdef __cplusplus
}
#endif
#endif /* PYNND_AMD_H */
