// state.h -- host-side state of one builder handle (one GPU, one HIP stream) and the
// launch entry points implemented by the kernel translation units.
#pragma once
#include <mutex>
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <vector>

#include "../../include/pynnd_amd.h"

// Experiment / debugging knobs (NND_* environment variables: table sizes, kernel variants, NND_POISON, NND_FOREST_DEBUG)
// exist only in a library built with `make KNOBS=1` (-DNND_EXPERIMENT_KNOBS): the product library reads no environment
// variable -- an environment leftover must not be able to change what a build computes.
#include <stdlib.h>
static inline const char *nnd_knob(const char *name) {
#ifdef NND_EXPERIMENT_KNOBS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// device-side counters (one int64 each), reset per phase
enum {
    CNT_ACCEPT = 0,   // k-list insertions (c of pynndescent_.py:317)
    CNT_PAIRS,        // pair distances evaluated
    CNT_ROWS,         // point rows gathered
    CNT_PROPOSALS,    // proposals emitted by the join
    CNT_ACTIVE,       // vertices with >=1 new candidate
    CNT_DEGENERATE,   // rp-forest: segments whose split left one side empty
    CNT_ACTIVE_SEGS,  // rp-forest: splittable segments for the next level
    CNT_LEAVES,
    CNT_SCRATCH,      // + 1 .. + 3: forest scratch words
    CNT_MFMA = 12,    // v_mfma_f32_16x16x4_f32 instructions issued (2048 flop each) by the join / leaf kernels
    CNT_COUNT = 16
};
// Hot kernels add to one of NND_CNT_STRIPES copies of the counter block (stripe = workgroup id), one
// atomic per workgroup per counter: a single shared word saturates at ~12 ns per atomic on MI355X and
// would serialise a million-wave launch.  nnd_read_counters sums the stripes.
#define NND_CNT_STRIPES 512

struct nnd_tlog { int ev; float *dst; bool add; };
struct nnd_hub_result;  // hubtree.hip: the last hub search tree built on this handle (FlatTree arrays, host side)  // a pending stage timer: events ev, ev+1 -> *dst

struct nnd_handle_s {
    nnd_params p{};
    nnd_stats stats{};
    char err[512] = {0};
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> tev;   // event pool of the deferred stage timers (capi.hip t_begin / t_end / t_flush)
    int tev_used = 0;
    std::vector<nnd_tlog> tlog;
    hipEvent_t ev_spin = nullptr;  // nnd_sync_spin: busy-polled completion of small read-backs (lower wake-up latency)
    // a shard's host waits go through its communicator (comm.h comm_wait: abort flag of the other ranks, timeout) -- a
    // collective whose peer died would otherwise keep every later wait on the stream blocked for ever
    int (*wait_hook)(void *user, hipStream_t stream) = nullptr;
    void *wait_user = nullptr;

    // geometry
    int64_t n = 0;
    int d = 0, dp = 0;      // dp = d rounded up to 32 floats (128-byte rows)
    int k = 0, ks = 0;      // ks = k rounded up to 16 (64-byte k-list rows)
    int mc = 0, mcp = 0;    // mcp = max_candidates rounded up to 16/32/64
    int rcap = 0, pcap = 0; // reverse-offer slots per (vertex,class); proposal slots per vertex
    int iter = 0;
    int64_t own_lo = 0, own_hi = 0; // rows this handle owns (row-sharded multi-GPU build); default [0, n)
    int n_ranks = 0;                // > 0 once nnd_set_shard_bounds was called: this handle is one shard of n_ranks
    int64_t *shard_bounds = nullptr;      // device (n_ranks + 1): first row of every rank, then n
    long long *shard_cursors = nullptr;   // device (66): per-destination record cursors, [64] dropped, [65] deferred
    bool stream_owned = true;             // false after nnd_set_stream: the caller's stream is borrowed
    // A shard created by nnd_create_impl with bounds allocates the per-OWNED-row tables (cand, rbuf, active) for its own
    // rows only; the pointers above are biased by -own_lo rows so that kernels keep indexing by global vertex id.
    bool lists_replicated = false;        // shard: the neighbour ids of ALL rows are refreshed before every join (shard.hip): remote targets are tested too
    const int32_t *own_order = nullptr;   // shard: the owned vertices in a spatially coherent order (shard.hip), or nullptr
    bool slim = false;
    void *slim_alloc[4] = {nullptr, nullptr, nullptr, nullptr};  // the allocations behind cand / rbuf / active / pbuf (always; biased or not)
    int64_t slim_rows() const { return slim ? own_hi - own_lo : n; }  // rows those three tables hold
    int64_t slim_row0() const { return slim ? own_lo : 0; }            // first of them
    uint32_t seed = 0, tree_seed = 0;

    // data
    const float *x_orig = nullptr; // (n,d) original rows (device); owned iff x_owned
    bool x_owned = false;
    bool x_valid = false;          // x_orig holds the rows of a nnd_set_data_* call made on THIS use of the handle
    float *xp = nullptr;   // (n,dp) prepared rows: centred (euclid) or L2-normalised (cosine), zero padded
    float *nrm = nullptr;  // (n) |x-mu|^2 (euclid) or 1/0 non-zero flag (cosine)
    float2 *nr2 = nullptr;                    // (n) (nrm, |x - bf16(x)|) per row: what the forest's margin kernels read (rpforest.hip rp_band)
    uint16_t *xh = nullptr; // (n,dp) bf16 copy of xp: screening pass of the rp-forest margins (half the bytes)
    float *mean = nullptr; // (dp + 4): column means | [dp] scale of the half-precision screening copies (prep.hip k_screen_scale) | [dp + 1] 1 / scale^2 | [dp + 2] sampled max |x|

    // k-lists, rows ascending by (dist, idx)
    uint32_t *knn_e = nullptr; // (n,ks) neighbour | NEW_BIT ; 0xFFFFFFFF = empty
    float *knn_d = nullptr;    // (n,ks) alt-space distance ; +inf = empty
    float *th = nullptr;       // (n) worst distance of every row (= knn_d[v][k-1]); compact so that threshold gathers stay in L2

    // candidates / proposals
    int32_t *cand = nullptr;  // (n, 2*mcp): [new | old], -1 padded
    uint32_t *rbuf = nullptr; // (n, 2, rcap) reverse offers, hashed slots: 32-bit invertible priority of the source (sample.hip)
    uint64_t *pbuf = nullptr; // (n, pcap) proposals (dist_bits<<32 | source), hashed slots; a shard holds its OWNED rows only (biased)
    uint64_t *pbuf_r = nullptr; // shard: (n, pcap_r) proposals for vertices owned elsewhere, exported every iteration (merge.hip)
    int pcap_r = 32;
    uint8_t *pdirty = nullptr; // (n) 1 when the row has pending proposals
    uint8_t *active = nullptr; // (n) 1 when the vertex will hold >= 1 new candidate this iteration

    // rp forest (all trees in one position space P = n_trees*n)
    int64_t P = 0;
    int32_t *perm[2] = {nullptr, nullptr};    // point id per position (ping-pong)
    int32_t *pos_seg[2] = {nullptr, nullptr}; // active segment index per position or -1
    int32_t *inv = nullptr;                   // (P) seg_pt: segment of point i in tree t, point-major [tree][point], -1 once final (top levels)
    uint8_t *side = nullptr;                  // (P) 0 left / 1 right
    uint8_t *side_pt = nullptr;               // (P) the same, point-major [tree][point] (written by the fused margin pass)
    uint8_t *leaf_flag = nullptr;             // (P) 1 at the first position of every final leaf
    int32_t *scan_out = nullptr;              // (P) exclusive scan scratch
    int32_t *scan_blk = nullptr;              // block sums
    int32_t *seg_start[2] = {nullptr, nullptr}, *seg_len[2] = {nullptr, nullptr};
    int32_t *seg_nleft = nullptr, *seg_child = nullptr; // per active segment scratch
    float *hyper = nullptr;                   // (max_segs, dp+?) hyperplane + offset
    uint16_t *hyper_h = nullptr;              // (max_segs, dp) bf16 copy of the normals (screening pass)
    int64_t max_segs = 0;
    int cur = 0; // which ping-pong half holds the finished permutation
    // routing pass (rpforest.hip forest_by_routing): sample, recorded top of the trees, cells
    int64_t s_m = 0, s_stride = 0;            // sample size per tree (0: routing disabled), sampling stride
    int cell_leaf = 0;                        // recorded trees stop at nodes of <= cell_leaf sample members
    int early_stop = 8;                       // ... or when fewer than 1 / early_stop of the sample positions are still splittable
    float *xs = nullptr;                      // (s_m, dp) compact copy of the sample rows
    uint16_t *xsh = nullptr;                  // (s_m, dp) bf16
    float2 *nr2s = nullptr;                   // (s_m) the sample's copy of nr2
    int64_t node_cap = 0, cell_cap = 0;
    float *node_hf = nullptr;                 // (node_cap, dp + 4) f32 hyperplane + offset + |h|
    uint16_t *node_hh = nullptr;              // (node_cap, dp) bf16 hyperplane
    int32_t *node_child = nullptr;            // (node_cap, 2) child node id, or -2 - first sample position of a cell
    unsigned char *node_pack = nullptr;       // (node_cap, 2 * dp + 16) packed records read by the routing passes (compacted ids)
    float *node_hfc = nullptr;                // (node_cap, dp + 4) node_hf compacted like node_pack (exact rechecks of the routing passes)
    int32_t *route_roots = nullptr;           // (4096) root node of every routed tree
    unsigned char *route_ws = nullptr;        // workspace of the coherent routing passes (rpforest.hip route_geometry), grow-only
    size_t route_ws_cap = 0;
    int32_t *s_leaf_depth = nullptr;          // (n_trees * s_m) depth of the cell that starts at a sample position
    int32_t *cell_count = nullptr, *cell_start = nullptr, *cell_depth = nullptr;  // (cell_cap)
    int32_t *small_list = nullptr;            // (3, cell_cap) start / len / depth of the cells finished one wave per cell
    int32_t *leaf_start = nullptr, *leaf_len = nullptr; // (n_leaves) after the forest is done; grow-only buffers
    double *colsum_partial = nullptr;         // prep scratch (column sums per row block), grow-only
    size_t colsum_cap = 0;
    int64_t leaf_cap = 0;                     // allocated entries of leaf_start / leaf_len
    std::vector<int32_t> h_leaf_start, h_leaf_len;  // host copies, fetched lazily (nnd_fetch_leaf_tables)
    bool h_leaf_valid = false;
    long long *tree_begin_dev = nullptr;      // (n_trees) first leaf index of every tree
    long long *h_tree_begin = nullptr;        // pinned host copy
    int32_t *wl_start = nullptr, *wl_len = nullptr; // leaf-seeding work list when leaves had to be cut (grow-only)
    int64_t wl_cap = 0;
    int64_t n_leaves = 0;
    int32_t max_leaf = 0;
    std::vector<int64_t> tree_leaf_begin; // per tree: first leaf index (host)
    bool forest_built = false;
    bool all_new = false;  // every edge of the graph still carries the "new" flag (true from reset until the first sampling pass)
    // candidate sampling by transposition (sample.hip): the reverse offers of an iteration are placed by the target's BUCKET (a run
    // of consecutive positions of the visiting order) and appended to the targets' slot banks in LDS
    int32_t *rv_pos = nullptr;                // (n) position of every vertex in the visiting order, grow-only
    int rv_pos_gen = -1;                      // forest_gen the table was built for
    int forest_gen = 0;                       // bumped whenever a forest (a visiting order) is finished
    uint32_t *rv_in_cursor = nullptr;         // (n_buckets, 8) cursors of the record sub-regions (+ the overflow count behind them)
    uint2 *rv_in_rec = nullptr;               // (n_buckets, 8, cap) records: (slot word = invertible priority of the source, target's index in its bucket | class << 15)
    uint2 *rv_ov = nullptr;                   // overflow list (hubs): (word, bucket << 9 | class << 8 | index)
    int64_t rv_cap_in = 0, rv_cap_ov = 0, rv_cap_pos = 0;
    int rv_in_cap = 0;
    const int32_t *rv_pos_of = nullptr;       // the order rv_pos inverts
    bool rv_off = false;                      // the record regions could not be allocated: this handle samples through the hashed atomicMin slots (rbuf) from now on
    bool jb_auto = false;                     // join_blocks was left to the library: sub-steps per iteration follow the update volume (capi.hip descent_iter)
    int jb_max = 8, jb_div = 2, jb_first = 8; // the schedule: sub-steps = pow2ceil(expected insertions per row / jb_div), at most jb_max; first iteration: as if jb_first per row
    int64_t last_updates = -1;                // k-list insertions of the previous iteration (-1: unknown): picks the late-iteration form of the pass
    bool pbuf_clean = false, rbuf_clean = false;  // every proposal / reverse-offer slot is EMPTY (their consumers re-arm what they read): nnd_launch_reset_graph then skips the 2 x 512 MB memsets

    int32_t *out_idx = nullptr;           // finished graph of the host-buffer entry points (grow-only; capi.hip out_buffers)
    float *out_dist = nullptr;
    size_t out_cap = 0;
    nnd_hub_result *hub = nullptr;
    struct nnd_sg_state *sg = nullptr;    // searchgraph.hip: workspace and result of the device pruning pass

    long long *counters = nullptr;      // device NND_CNT_STRIPES x CNT_COUNT (stripe 0 doubles as scratch for single-block kernels)
    long long h_counters[CNT_COUNT] = {0};
    long long *counters_sum = nullptr;        // (CNT_COUNT) device: stripes summed by k_counters_reduce
    long long *h_pin = nullptr;               // pinned host words for the small latency-critical read-backs
    long long *h_pin_dev = nullptr;           // the same words as the device sees them: a kernel can hand the host a few values itself
    long long flag_seq = 0;                   // sequence number of the last such hand-over (rpforest.hip forest_levels)

    void set_error(const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
    }
};
typedef nnd_handle_s nnd_ctx;

// capi.hip: nnd_create with the shard geometry known up front (bounds == nullptr: a plain handle)
int nnd_create_impl(nnd_handle_t *out, const nnd_params *p, const int64_t *bounds_host, int n_ranks, int rank);
// ---- implemented in the kernel translation units; each returns 0 / sets ctx->err ----
int nnd_launch_prep(nnd_ctx *ctx);
// the pieces of nnd_launch_prep (prep.hip), for the sharded build: a rank preps its own rows first, the others once they arrive
void nnd_prep_mean_geometry(int64_t n, int64_t *n_s, int64_t *stride);
int nnd_prep_partial_blocks(int64_t members);
double *nnd_prep_partial_buffer(nnd_ctx *ctx, size_t doubles);
int nnd_prep_mean_partial(nnd_ctx *ctx, const float *x_rows, int64_t row0, int64_t r_lo, int64_t r_hi, int64_t stride, double *partial);
int nnd_prep_mean_finish(nnd_ctx *ctx, const double *partial, int nblocks, int64_t n_s);
int nnd_prep_rows(nnd_ctx *ctx, const float *x_all, int64_t row_lo, int64_t row_hi, bool first);
int nnd_launch_reset_graph(nnd_ctx *ctx);
int nnd_launch_forest(nnd_ctx *ctx);
// the forest of the row-sharded build, sharded by cell (rpforest.hip; driven by shard.hip)
struct nnd_tops_info {
    int64_t n_packed = 0;        // packed node records of this rank's trees
    int64_t n_low = 0, high_lo = 0;
    int32_t n_cells = 0;         // cells of this rank's trees, numbered tree-major
    int32_t tree_cells[64] = {0};  // ... per local tree
    int levels = 0;
};
int nnd_forest_sample_gather(nnd_ctx *ctx, int64_t j_lo, int64_t j_hi);
int nnd_forest_tops(nnd_ctx *ctx, int T_loc, int tree_bias, nnd_tops_info *out);
int nnd_forest_tops_pack(nnd_ctx *ctx, const nnd_tops_info *ti, int64_t node_base, const int32_t *cell_gid_dev, unsigned char *pack_dst, float *hf_dst);
int nnd_forest_route_rows(nnd_ctx *ctx, const unsigned char *pack_all, const float *hf_all, const int32_t *roots_dev, int T_all, int64_t row_lo,
                          int64_t nrows, int64_t n_cells_all, int32_t *cell_count_all);
int nnd_forest_route_records(nnd_ctx *ctx, int T_all, int64_t row_lo, int64_t nrows, int32_t n_cells_all, int32_t *cell_count_all, int32_t *count_copy,
                             const int32_t *dest_cell_dev, int G, int32_t *rec_cell, int32_t *rec_row, long long *dest_off_dev);
int nnd_forest_finish_owned(nnd_ctx *ctx, const int32_t *rec_cell, const int32_t *rec_row, int64_t n_rec, int32_t cell_base, int32_t n_cells_own,
                            const int32_t *cnt_src, int G, const int32_t *cell_depth_all, const int32_t *depth_map_dev,
                            const int32_t *tree_first_cell_dev, int T_all);
int nnd_launch_leaf_array(nnd_ctx *ctx, int32_t *out_dev /* (n_leaves,max_leaf) */);
int nnd_fetch_leaf_tables(nnd_ctx *ctx);
int nnd_launch_leaf_init(nnd_ctx *ctx);
int nnd_launch_leaf_init_array(nnd_ctx *ctx, const int32_t *leaf_host, int64_t n_leaves, int32_t max_leaf_size);
int nnd_launch_random_init(nnd_ctx *ctx);
int nnd_launch_init_from_graph(nnd_ctx *ctx, const int32_t *idx_dev, const float *dist_dev, int width);
int nnd_launch_sample(nnd_ctx *ctx);
// One process-wide lock around the creation and the tear-down of handles (hipMalloc / hipFree storms, stream and event
// creation / destruction): ranks that live as threads of one process (LOCAL transport, nnd_build_multi) create and destroy
// their state at the same moment; none of it is on a timed path, so it is simply serialised.
std::recursive_mutex &nnd_lifecycle_mutex();

int nnd_launch_sample_begin(nnd_ctx *ctx, int64_t cap, int32_t *targets_dev, uint32_t *sources_dev, long long *counts_dev);
int nnd_launch_sample_finish(nnd_ctx *ctx, const int32_t *targets_dev, const uint32_t *sources_dev, int64_t count);
int nnd_launch_proposal_export_regions(nnd_ctx *ctx, int64_t cap, int32_t *targets_dev, uint64_t *keys_dev, long long *counts_dev);
int nnd_launch_join(nnd_ctx *ctx, int64_t v_begin, int64_t v_end);
int nnd_launch_merge(nnd_ctx *ctx);
int nnd_launch_finalize(nnd_ctx *ctx, int32_t *out_idx_dev, float *out_dist_dev);
int nnd_launch_pairwise(nnd_ctx *ctx, const int32_t *rows_a_dev, int na, const int32_t *rows_b_dev, int nb,
                        float *out_dev);
int nnd_launch_import_proposals(nnd_ctx *ctx, const uint64_t *keys, const int32_t *targets, int64_t count);
int nnd_launch_merge_graph_rows(nnd_ctx *ctx, int64_t lo, int64_t hi, const uint32_t *e_src, const float *d_src, int n_src, int64_t stride);
int nnd_launch_clear_new_flags(nnd_ctx *ctx);
int nnd_launch_diversify_rows(nnd_ctx *ctx, int32_t *idx_dev, float *dist_dev, const nnd_prune_opts *opts,
                              const int32_t *degree_dev);
int nnd_launch_diversify_csr(nnd_ctx *ctx, const int32_t *indptr_dev, const int32_t *indices_dev, float *data_dev,
                             int *too_long_dev, const nnd_prune_opts *opts, const int32_t *degree_dev);
int nnd_launch_degree_prune(nnd_ctx *ctx, const int32_t *indptr_dev, float *data_dev, int max_degree);
void nnd_forest_stable_partition(nnd_ctx *ctx, int64_t n, const int32_t *ord, const int32_t *pos, uint8_t *side, const int32_t *seg_start,
                                 const int32_t *seg_len, int n_segs, int32_t *nleft, const int32_t *seg_child, int32_t *ord_out,
                                 int32_t *pos_out);
int nnd_hub_tree_build_impl(nnd_ctx *ctx, const int32_t *rank_order_host, int leaf_size, int max_depth, int angular);
int nnd_hub_tree_fetch_impl(nnd_ctx *ctx, float *hyperplanes, float *offsets, int32_t *children, int32_t *indices, int32_t *max_leaf);
int64_t nnd_hub_tree_nodes(const nnd_ctx *ctx);
void nnd_hub_tree_free(nnd_ctx *ctx);
void nnd_search_graph_free(nnd_ctx *ctx);
int nnd_search_graph_impl(nnd_ctx *ctx, const int32_t *idx_src, const float *dist_src, bool src_on_device, int n_neighbors, float multiplier,
                          float diversify_prob, bool aware, float aggressiveness, uint32_t seed, int32_t *fwd_rows_host, float *fwd_dist_host,
                          nnd_search_graph_stats *st);
int nnd_search_graph_fetch_impl(nnd_ctx *ctx, int32_t *indptr_host, int32_t *indices_host);
int nnd_read_counters(nnd_ctx *ctx);  // device -> ctx->h_counters (synchronises the stream)
int nnd_zero_counters(nnd_ctx *ctx);

// Visiting order for the vertex-parallel kernels (join, finalize): the first tree's leaf order -- a permutation of
// [0, n) in which consecutive entries are close in space -- or nullptr (identity) when there is no forest or the
// handle owns only a slice of the rows.
static inline const int32_t *nnd_vertex_order(const nnd_ctx *ctx) {
    if (!ctx->forest_built || ctx->p.n_trees <= 0 || ctx->own_lo != 0 || ctx->own_hi != ctx->n) return nullptr;
    return ctx->perm[ctx->cur];
}

// Rows whose CURRENT neighbour lists this handle holds (the join's membership test reads them): every row on a plain
// handle and on a shard whose builder all-gathers the neighbour ids before every join (lists_replicated); only the owned
// slice otherwise (the owners then dedup at import / in the merge, utils.py:489-492).
static inline int64_t nnd_list_lo(const nnd_ctx *ctx) { return (ctx->n_ranks > 1 && !ctx->lists_replicated) ? ctx->own_lo : 0; }
static inline int64_t nnd_list_hi(const nnd_ctx *ctx) { return (ctx->n_ranks > 1 && !ctx->lists_replicated) ? ctx->own_hi : ctx->n; }

// Wait for everything queued on the handle's stream by polling an event: the per-level read-backs of the forest build
// are latency critical (the GPU idles until the host has the segment count), and a blocking wait wakes up late.
static inline hipError_t nnd_sync_spin(nnd_ctx *ctx) {
    if (ctx->wait_hook) return ctx->wait_hook(ctx->wait_user, ctx->stream) ? hipErrorUnknown : hipSuccess;
    if (!ctx->ev_spin) return hipStreamSynchronize(ctx->stream);
    hipError_t e = hipEventRecord(ctx->ev_spin, ctx->stream);
    if (e != hipSuccess) return e;
    while ((e = hipEventQuery(ctx->ev_spin)) == hipErrorNotReady) {
    }
    return e;
}

// Stage timers are DEFERRED: begin/end events are recorded on the stream and read back in one go (t_flush) where the
// host waits anyway, so timing a stage never drains the GPU pipeline between stages.
static inline int t_begin(nnd_ctx *ctx) {
    const int idx = ctx->tev_used;
    while ((int)ctx->tev.size() < idx + 2) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return -1;  // callers skip the timer
        ctx->tev.push_back(e);
    }
    (void)hipEventRecord(ctx->tev[idx], ctx->stream);
    ctx->tev_used += 2;
    return idx;
}
static inline void t_end(nnd_ctx *ctx, int idx, float *dst, bool add) {
    if (idx < 0) return;
    (void)hipEventRecord(ctx->tev[idx + 1], ctx->stream);
    ctx->tlog.push_back({idx, dst, add});
}
static inline void t_flush(nnd_ctx *ctx) {
    if (ctx->tlog.empty()) { ctx->tev_used = 0; return; }
    (void)nnd_sync_spin(ctx);
    for (const nnd_tlog &t : ctx->tlog) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ctx->tev[t.ev], ctx->tev[t.ev + 1]);
        if (t.add) *t.dst += ms; else *t.dst = ms;
    }
    ctx->tlog.clear();
    ctx->tev_used = 0;
}

