// query.hip -- batched k-NN queries against a prepared index: tree descent + best-first graph search.
//
// Replaces the search closure of NNDescent._init_search_function (reference pynndescent_.py:1793-1883), the tree
// descent select_side / search_flat_tree (rp_trees.py:2662-2741) and the final deheap_sort (utils.py:189-218) for
// dense float32 data with the euclidean / cosine metrics.  The reference walks one query at a time (optionally one
// numba thread per query); here ONE WAVE owns one query:
//   * result list: the k best (distance, vertex) pairs, sorted ascending, one entry per lane (k <= 64; two per lane up to k = 128) -- the
//     reference's max-heap of size k (simple_heap_push, utils.py:352-406): a candidate enters iff it beats the worst
//     entry, the worst leaves; insertion = one ballot (rank) + one lane shift;
//   * frontier (`seed_set`, a heapq in the reference): (distance, vertex) pairs in LDS, pop-min by a wave reduction.
//     Entries at or beyond the current distance bound can never be expanded (the bound only shrinks), so they are
//     dropped when the array fills up;
//   * visited set (a bitset over all n points in the reference, utils.py:323-349): a hash set in LDS (open addressing);
//   * distances: a quad (4 lanes) per candidate, 16 candidates of an adjacency row per step, rows gathered from HBM,
//     the query vector in LDS; float32 in the reference's formulas (distances.py:63-91, 583-630) on the RAW rows --
//     cosine queries are normalised first (pynndescent_.py:1808-1815), data rows are not;
//   * stop rule: the nearest unexpanded frontier vertex is farther than
//         bound = worst + epsilon * (worst - min_distance)                         (pynndescent_.py:1850-1853).
// Random choices (ties in the tree descent, random start vertices when the tree leaf holds fewer than
// min(k, n_neighbors) points) come from the counter hash, keyed by the query's number.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "../../include/pynnd_amd.h"

#define Q_FRONTIER 512   // frontier entries per query (LDS tier)
#define Q_VISITED 4096   // visited-set slots per query (power of two; LDS tier)
#define Q_VISITED_MAX 3400  // entries after which the LDS set counts as full
#define Q_BIG_FRONTIER 65536  // frontier entries per query of the global-memory tier
#define Q_BIG_BATCH 256       // queries per launch of the global-memory tier (scratch = batch * (n / 8 + 512 KB))
// Two tiers.  The LDS tier (visited = hash set of Q_VISITED_MAX vertices, frontier of Q_FRONTIER entries) covers what a
// search with the usual k / epsilon touches.  A query that would overflow either structure is NOT answered from a
// truncated search: the wave raises the query's flag and stops, and the host re-runs exactly those queries on the
// global-memory tier -- visited = a bitset over all n points (the reference's own structure, utils.py:323-349), frontier
// of Q_BIG_FRONTIER entries in HBM -- so no result ever comes from a search the reference would have continued.
#define Q_CHUNK 64       // candidates handled per step
#define Q_EMPTY 0xFFFFFFFFu

struct nnd_searcher_s {
    int device = 0;
    int64_t n = 0, nnz = 0, n_nodes = 0;
    int d = 0, dp = 0, metric = 0, n_neighbors = 0;
    float min_distance = 0.0f;
    uint32_t seed = 0;
    float *x = nullptr;        // (n, dp) rows padded to a multiple of 4 floats
    float *xn2 = nullptr;      // (n) squared norms (cosine)
    int32_t *indptr = nullptr, *indices = nullptr;
    float *hyper = nullptr, *offsets = nullptr;  // (n_nodes, dp), (n_nodes)
    int32_t *children = nullptr, *tree_idx = nullptr;
    hipStream_t stream = nullptr;
    int64_t last_spilled = 0;  // queries of the last call that ran on the global-memory tier
    bool force_big = false;    // nnd_searcher_set_tier(1): every query on the global-memory tier (tests)
    char err[512] = {0};
    void set_error(const char *fmt, ...) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(err, sizeof(err), fmt, ap);
        va_end(ap);
    }
};

static thread_local char g_serr[512] = {0};

// quad-cooperative alt-space distance between the query (LDS, dp floats, |q|^2 = qn2) and row `v`
__device__ __forceinline__ float q_quad_dist(const float *__restrict__ x, const float *__restrict__ xn2, int dp, int metric,
                                             const float *qs, float qn2, int64_t v, int sub) {
    const float4 *row = (const float4 *)(x + v * dp);
    const float4 *q4 = (const float4 *)qs;
    float acc = 0.0f;
    if (metric == 0) {
        for (int c = sub; c < (dp >> 2); c += 4) {
            const float4 a = row[c], b = q4[c];
            const float d0 = b.x - a.x, d1 = b.y - a.y, d2 = b.z - a.z, d3 = b.w - a.w;
            acc += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
        }
    } else {
        for (int c = sub; c < (dp >> 2); c += 4) {
            const float4 a = row[c], b = q4[c];
            acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    }
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0xB1, 0xF, 0xF, false));
    acc += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x4E, 0xF, 0xF, false));
    if (metric == 0) return acc;
    // alternative_cosine (distances.py:600-630)
    const float nx = xn2[v];
    if (qn2 == 0.0f && nx == 0.0f) return 0.0f;
    if (qn2 == 0.0f || nx == 0.0f || acc <= 0.0f) return NND_FLT_MAX;
    const float r = sqrtf(qn2 * nx) / acc;
    return r > 1.0f ? log2f(r) : 0.0f;
}

// KU: result entries per lane -- entry u of lane j is position 64 u + j of the list (k <= 64 KU: 1, 2 or 4)
template <bool BIG, int KU>
__global__ __launch_bounds__(256) void k_query(const float *__restrict__ x, const float *__restrict__ xn2, int dp, int d, int metric,
                                               int64_t n, const int32_t *__restrict__ indptr, const int32_t *__restrict__ indices,
                                               const float *__restrict__ hyper, const float *__restrict__ offsets,
                                               const int32_t *__restrict__ children, const int32_t *__restrict__ tree_idx,
                                               int64_t n_nodes, const float *__restrict__ queries, int64_t nq, int k, float epsilon,
                                               float min_distance, int n_neighbors, uint32_t seed, int32_t *__restrict__ out_idx,
                                               float *__restrict__ out_dist, uint8_t *__restrict__ overflow /* (nq) LDS tier: set when the query needs the big tier */,
                                               const int32_t *__restrict__ qlist /* BIG: the queries of this batch */,
                                               int n_list, unsigned char *__restrict__ scratch, size_t scratch_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char qsm[];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t slot = (int64_t)blockIdx.x * 4 + w;
    if (BIG ? slot >= n_list : slot >= nq) return;  // whole wave; no workgroup barrier below
    const int64_t qi = BIG ? (int64_t)qlist[slot] : slot;
    constexpr int FCAP = BIG ? Q_BIG_FRONTIER : Q_FRONTIER;
    const size_t per_wave = BIG ? (size_t)dp * 4 + Q_CHUNK * 8 : (size_t)dp * 4 + Q_FRONTIER * 8 + Q_VISITED * 4 + Q_CHUNK * 8;
    unsigned char *mine = qsm + (size_t)w * ((per_wave + 15) & ~(size_t)15);
    float *qs = (float *)mine;                       // dp
    float *fd;                                       // FCAP distances
    int32_t *fv;                                     // FCAP vertices
    uint32_t *vis;                                   // LDS tier: Q_VISITED hash slots; big tier: (n + 31) / 32 bit words
    int32_t *cl;                                     // Q_CHUNK candidate ids
    if (BIG) {
        unsigned char *gs = scratch + (size_t)slot * scratch_stride;
        fd = (float *)gs;
        fv = (int32_t *)(fd + FCAP);
        vis = (uint32_t *)(fv + FCAP);
        cl = (int32_t *)(qs + dp);
    } else {
        fd = qs + dp;
        fv = (int32_t *)(fd + FCAP);
        vis = (uint32_t *)(fv + FCAP);
        cl = (int32_t *)(vis + Q_VISITED);
    }
    float *cd = (float *)(cl + Q_CHUNK);             // Q_CHUNK candidate distances
    const int sub = lane & 3, grp = lane >> 2;
    const int64_t vis_words = BIG ? (n + 31) / 32 : Q_VISITED;
    bool spilled = false;  // LDS tier: a structure overflowed -- the query is handed to the big tier (wave-uniform)

    // ---- the query: cosine queries are normalised (pynndescent_.py:1808-1815); a zero cosine query returns nothing ----
    float part = 0.0f;
    for (int j = lane; j < dp; j += 64) {
        const float v = j < d ? queries[qi * d + j] : 0.0f;
        qs[j] = v;
        part += v * v;
    }
    float qn2 = nnd_wave_sum_f32(part);
    for (int64_t s = lane; s < vis_words; s += 64) vis[s] = BIG ? 0u : Q_EMPTY;
    bool dead = false;
    if (metric == 1) {
        const float nrm = sqrtf(qn2);
        if (nrm > 0.0f) {
            nnd_wave_lds_sync();
            for (int j = lane; j < dp; j += 64) qs[j] = qs[j] / nrm;
            nnd_wave_lds_sync();
            float p2 = 0.0f;
            for (int j = lane; j < dp; j += 64) p2 += qs[j] * qs[j];
            qn2 = nnd_wave_sum_f32(p2);
        } else {
            dead = true;
        }
    }
    nnd_wave_lds_sync();

    float rd[KU];    // result list, ascending: entry u of lane j holds the (64 u + j)-th best
    int32_t rv[KU];
#pragma unroll
    for (int u = 0; u < KU; u++) {
        rd[u] = INFINITY;
        rv[u] = -1;
    }
    int fn = 0;            // frontier size (wave-uniform)
    float bound = INFINITY;
    auto worst = [&]() -> float {
        float w = rd[0];
#pragma unroll
        for (int u = 1; u < KU; u++) w = ((k - 1) >> 6) == u ? rd[u] : w;  // (wave-uniform)
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(w), (k - 1) & 63));
    };
    auto update_bound = [&]() {
        const float wd = worst();
        bound = wd + epsilon * (wd - min_distance);  // inf while the list is not full
    };
    // simple_heap_push (utils.py:352-406) on the sorted list: enters iff it beats the worst entry
    auto result_push = [&](float dc, int32_t vc) {
        if (!(dc < worst())) return;
        int pos = 0;  // entries that stay in front of the new one
        float cd[KU];  // what falls from the end of every 64-entry segment when the entries behind `pos` move up by one
        int32_t cv[KU];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            pos += __popcll(__ballot(64 * u + lane < k && rd[u] <= dc));
            cd[u] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rd[u]), 63));
            cv[u] = __builtin_amdgcn_readlane(rv[u], 63);
        }
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int p = pos - 64 * u;  // where the new entry lands, seen from this segment (wave-uniform)
            if (p >= 64) continue;       // behind this segment: nothing moves here
            const float dn = __shfl_up(rd[u], 1, 64);
            const int32_t vn = __shfl_up(rv[u], 1, 64);
            if (p < 0) {  // in an earlier segment: the whole segment moves up, its first lane takes what fell from the one before
                if (64 * u + lane < k) {
                    rd[u] = lane == 0 ? cd[u > 0 ? u - 1 : 0] : dn;
                    rv[u] = lane == 0 ? cv[u > 0 ? u - 1 : 0] : vn;
                }
            } else {
                if (lane > p && 64 * u + lane < k) { rd[u] = dn; rv[u] = vn; }
                if (lane == p) { rd[u] = dc; rv[u] = vc; }
            }
        }
    };
    auto frontier_push = [&](float dc, int32_t vc) {
        if (fn == FCAP) {  // drop what can never be expanded any more (the bound only shrinks)
            int kept = 0;
            for (int s0 = 0; s0 < FCAP; s0 += 64) {
                const float e = fd[s0 + lane];
                const int32_t ev = fv[s0 + lane];
                const bool keep = e < bound;
                const unsigned long long m = __ballot(keep);
                nnd_wave_lds_sync();
                if (keep) {
                    fd[kept + nnd_prefix_popc(m)] = e;
                    fv[kept + nnd_prefix_popc(m)] = ev;
                }
                kept += __popcll(m);
                nnd_wave_lds_sync();
            }
            fn = kept;
            if (fn == FCAP) {  // still full: the reference's heap would have grown -- LDS tier: hand the query over
                if (!BIG) spilled = true;  // (big tier, 65536 live entries: the candidate stays in the result list only)
                return;
            }
        }
        if (lane == 0) {
            fd[fn] = dc;
            fv[fn] = vc;
        }
        fn++;
    };
    // check_and_mark_visited (utils.py:335-349): true if `u` had not been seen.  Big tier: the reference's bitset over
    // all n points.  LDS tier: a hash set; once it holds Q_VISITED_MAX vertices the query is handed to the big tier
    // (the callers test nvis after every batch).
    int nvis = 0;  // wave-uniform (updated with ballots by the callers)
    auto mark_fresh = [&](uint32_t u) -> bool {
        if (BIG) {
            const uint32_t bit = 1u << (u & 31u);
            return (atomicOr(&vis[u >> 5], bit) & bit) == 0u;
        }
        if (nvis >= Q_VISITED_MAX) return false;
        uint32_t h = nnd_mix32(u) & (Q_VISITED - 1);
        for (int probe = 0; probe < Q_VISITED; probe++) {
            const uint32_t old = atomicCAS(&vis[h], Q_EMPTY, u);
            if (old == Q_EMPTY) return true;
            if (old == u) return false;
            h = (h + 1) & (Q_VISITED - 1);
        }
        return false;
    };
    // distances of cl[0..nc) -> cd, a quad per candidate
    auto chunk_dists = [&](int nc) {
        for (int c0 = 0; c0 < nc; c0 += 16) {
            const int c = c0 + grp;
            const int64_t v = cl[c < nc ? c : 0];
            const float dv = q_quad_dist(x, xn2, dp, metric, qs, qn2, v, sub);
            if (sub == 0 && c < nc) cd[c] = dv;
        }
        nnd_wave_lds_sync();
    };
    auto in_result = [&](int32_t vc) -> bool {
        bool hit = false;
#pragma unroll
        for (int u = 0; u < KU; u++) hit = hit || (64 * u + lane < k && rv[u] == vc);
        return __ballot(hit) != 0ull;
    };

    if (!dead) {
        // ---- init from the tree (rp_trees.py:2732-2741): descend to a leaf ----
        int ls = 0, le = 0;
        if (n_nodes > 0) {
            int node = 0;
            int depth = 0;
            while (children[2 * node] > 0) {
                float m = 0.0f;
                const float *h = hyper + (int64_t)node * dp;
                for (int j = lane; j < dp; j += 64) m += h[j] * qs[j];
                m = nnd_wave_sum_f32(m) + offsets[node];
                int side;
                if (fabsf(m) < 1e-8f) side = (int)(nnd_hash3(seed, (uint32_t)qi, (uint32_t)depth) & 1u);  // rp_trees.py:2668-2673
                else side = m > 0.0f ? 0 : 1;
                node = children[2 * node + side];
                depth++;
            }
            ls = -children[2 * node];
            le = -children[2 * node + 1];
        }
        const int n_initial = le - ls;
        for (int c0 = 0; c0 < n_initial; c0 += Q_CHUNK) {
            const int nc = n_initial - c0 < Q_CHUNK ? n_initial - c0 : Q_CHUNK;
            bool fr = false;
            if (lane < nc) {
                const int32_t u = tree_idx[ls + c0 + lane];
                cl[lane] = u;
                fr = mark_fresh((uint32_t)u);
            }
            nvis += __popcll(__ballot(fr));
            if (!BIG && nvis >= Q_VISITED_MAX) spilled = true;
            nnd_wave_lds_sync();
            chunk_dists(nc);
            for (int j = 0; j < nc; j++) {  // pynndescent_.py:1826-1832
                result_push(cd[j], cl[j]);
                frontier_push(cd[j], cl[j]);
            }
            nnd_wave_lds_sync();
            if (spilled) break;
        }
        // ---- random start vertices if the leaf was small (pynndescent_.py:1834-1848) ----
        const int n_random = (k < n_neighbors ? k : n_neighbors) - n_initial;
        for (int j = 0; j < n_random && !spilled; j++) {
            const uint32_t u = nnd_hash3(seed ^ 0x3C6EF372u, (uint32_t)qi, (uint32_t)j) % (uint32_t)n;
            bool fresh = false;
            if (lane == 0) fresh = mark_fresh(u);
            fresh = __shfl((int)fresh, 0, 64);
            if (!fresh) continue;
            nvis++;
            if (lane == 0) cl[0] = (int32_t)u;
            nnd_wave_lds_sync();
            chunk_dists(1);
            result_push(cd[0], cl[0]);
            frontier_push(cd[0], cl[0]);
            nnd_wave_lds_sync();
        }
        update_bound();

        // ---- best-first search (pynndescent_.py:1850-1881) ----
        while (fn > 0 && !spilled) {
            nnd_wave_lds_sync();
            // pop the nearest frontier vertex
            float bd = INFINITY;
            int bp = -1;
            for (int s = lane; s < fn; s += 64) {
                const float e = fd[s];
                if (e < bd) { bd = e; bp = s; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float od = __shfl_xor(bd, o, 64);
                const int op = __shfl_xor(bp, o, 64);
                if (od < bd || (od == bd && op >= 0 && (bp < 0 || op < bp))) { bd = od; bp = op; }
            }
            if (bp < 0 || !(bd < bound)) break;  // pynndescent_.py:1857
            const int32_t vertex = fv[bp];
            nnd_wave_lds_sync();
            if (lane == 0) {
                fd[bp] = fd[fn - 1];
                fv[bp] = fv[fn - 1];
            }
            fn--;
            const int a = indptr[vertex], b = indptr[vertex + 1];
            for (int e0 = a; e0 < b; e0 += Q_CHUNK) {
                const int e = e0 + lane;
                bool fresh = false;
                int32_t u = -1;
                if (e < b) {
                    u = indices[e];
                    fresh = mark_fresh((uint32_t)u);
                }
                const unsigned long long m = __ballot(fresh);
                const int nc = __popcll(m);
                nvis += nc;
                if (!BIG && nvis >= Q_VISITED_MAX) spilled = true;  // this batch is still exact (the set has room for it)
                if (nc == 0) continue;
                nnd_wave_lds_sync();
                if (fresh) cl[nnd_prefix_popc(m)] = u;
                nnd_wave_lds_sync();
                chunk_dists(nc);
                for (int j = 0; j < nc; j++) {
                    const float dc = cd[j];
                    const int32_t vc = cl[j];
                    if (dc < bound && !in_result(vc)) {  // pynndescent_.py:1866-1874
                        result_push(dc, vc);
                        frontier_push(dc, vc);
                        update_bound();
                    }
                }
                nnd_wave_lds_sync();
                if (spilled) break;
            }
        }
    }
    if (!BIG && lane == 0) overflow[qi] = spilled ? 1 : 0;  // the host re-runs flagged queries on the big tier
#pragma unroll
    for (int u = 0; u < KU; u++)
        if (64 * u + lane < k) {  // ascending, like deheap_sort; unfilled slots (-1, inf)
            out_idx[qi * k + 64 * u + lane] = rv[u];
            out_dist[qi * k + 64 * u + lane] = rd[u];
        }
}

// squared norms of the padded rows (alternative_cosine recomputes them per call, distances.py:617-620)
__global__ void k_row_norm2(const float *__restrict__ x, int64_t n, int dp, float *__restrict__ out) {
    const int lane = nnd_lane();
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= n) return;
    float s = 0.0f;
    for (int j = lane; j < dp; j += 64) {
        const float v = x[r * dp + j];
        s += v * v;
    }
    s = nnd_wave_sum_f32(s);
    if (lane == 0) out[r] = s;
}

#define S_HIP(expr)                                                                                 \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            s->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__);  \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)

static int upload_padded(nnd_searcher_s *s, float **dst, const float *src, int64_t rows, int d, int dp) {
    S_HIP(hipMalloc((void **)dst, sizeof(float) * (size_t)(rows ? rows : 1) * dp));
    if (rows == 0) return 0;
    if (d == dp) {
        S_HIP(hipMemcpy(*dst, src, sizeof(float) * (size_t)rows * d, hipMemcpyHostToDevice));
    } else {
        S_HIP(hipMemset(*dst, 0, sizeof(float) * (size_t)rows * dp));
        S_HIP(hipMemcpy2D(*dst, sizeof(float) * dp, src, sizeof(float) * d, sizeof(float) * d, (size_t)rows, hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" const char *nnd_searcher_last_error(nnd_searcher_t s) { return s ? s->err : g_serr; }

extern "C" int32_t nnd_searcher_destroy(nnd_searcher_t s) {
    if (!s) return 0;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    void *ptrs[] = {s->x, s->xn2, s->indptr, s->indices, s->hyper, s->offsets, s->children, s->tree_idx};
    for (void *p : ptrs)
        if (p) (void)hipFree(p);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return 0;
}

static int searcher_fill(nnd_searcher_s *s, const float *data, const int32_t *indptr, const int32_t *indices, const float *hyperplanes,
                         const float *offsets, const int32_t *children, const int32_t *tree_indices) {
    S_HIP(hipSetDevice(s->device));
    S_HIP(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    if (upload_padded(s, &s->x, data, s->n, s->d, s->dp)) return 1;
    S_HIP(hipMalloc((void **)&s->xn2, sizeof(float) * (size_t)s->n));
    hipLaunchKernelGGL(k_row_norm2, dim3((unsigned)((s->n + 3) / 4)), dim3(256), 0, s->stream, s->x, s->n, s->dp, s->xn2);
    S_HIP(hipMalloc((void **)&s->indptr, sizeof(int32_t) * (size_t)(s->n + 1)));
    S_HIP(hipMemcpy(s->indptr, indptr, sizeof(int32_t) * (size_t)(s->n + 1), hipMemcpyHostToDevice));
    S_HIP(hipMalloc((void **)&s->indices, sizeof(int32_t) * (size_t)(s->nnz ? s->nnz : 1)));
    S_HIP(hipMemcpy(s->indices, indices, sizeof(int32_t) * (size_t)s->nnz, hipMemcpyHostToDevice));
    if (s->n_nodes > 0) {
        if (upload_padded(s, &s->hyper, hyperplanes, s->n_nodes, s->d, s->dp)) return 1;
        S_HIP(hipMalloc((void **)&s->offsets, sizeof(float) * (size_t)s->n_nodes));
        S_HIP(hipMemcpy(s->offsets, offsets, sizeof(float) * (size_t)s->n_nodes, hipMemcpyHostToDevice));
        S_HIP(hipMalloc((void **)&s->children, sizeof(int32_t) * 2 * (size_t)s->n_nodes));
        S_HIP(hipMemcpy(s->children, children, sizeof(int32_t) * 2 * (size_t)s->n_nodes, hipMemcpyHostToDevice));
        S_HIP(hipMalloc((void **)&s->tree_idx, sizeof(int32_t) * (size_t)s->n));
        S_HIP(hipMemcpy(s->tree_idx, tree_indices, sizeof(int32_t) * (size_t)s->n, hipMemcpyHostToDevice));
    }
    S_HIP(hipStreamSynchronize(s->stream));
    return 0;
}

extern "C" int32_t nnd_searcher_create(nnd_searcher_t *out, int32_t device, int64_t n, int32_t dim, int32_t metric, const float *data,
                                       const int32_t *indptr, const int32_t *indices, int64_t nnz, const float *hyperplanes,
                                       const float *offsets, const int32_t *children, const int32_t *tree_indices, int64_t n_nodes,
                                       float min_distance, int32_t n_neighbors, const int64_t *rng_state) {
    auto fail = [&](const char *msg) {
        snprintf(g_serr, sizeof(g_serr), "nnd_searcher_create: %s", msg);
        return 1;
    };
    if (!out || !data || !indptr || !indices || n < 1 || dim < 1) return fail("bad arguments");
    if (metric != NND_METRIC_SQEUCLIDEAN && metric != NND_METRIC_ALT_COSINE) return fail("unknown metric");
    if (n_nodes > 0 && (!hyperplanes || !offsets || !children || !tree_indices)) return fail("tree arrays missing");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail("no HIP device visible (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail("device out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess || strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail("this build targets gfx950 (MI355X) only");
    nnd_searcher_s *s = new nnd_searcher_s();
    s->device = device;
    s->n = n;
    s->nnz = nnz;
    s->n_nodes = n_nodes;
    s->d = dim;
    s->dp = (dim + 3) & ~3;
    s->metric = metric;
    s->n_neighbors = n_neighbors;
    s->min_distance = min_distance;
    s->seed = rng_state ? nnd_mix32((uint32_t)rng_state[0] ^ nnd_mix32((uint32_t)rng_state[1] + 0x9E3779B9u) ^ nnd_mix32((uint32_t)rng_state[2] + 0x7F4A7C15u)) : 1u;
    if (searcher_fill(s, data, indptr, indices, hyperplanes, offsets, children, tree_indices)) {
        snprintf(g_serr, sizeof(g_serr), "nnd_searcher_create: %s", s->err);
        nnd_searcher_destroy(s);
        return 1;
    }
    *out = s;
    return 0;
}

extern "C" int64_t nnd_searcher_last_spilled(nnd_searcher_t s) { return s ? s->last_spilled : -1; }
extern "C" int32_t nnd_searcher_set_tier(nnd_searcher_t s, int32_t tier) {
    if (!s) { snprintf(g_serr, sizeof(g_serr), "nnd_searcher_set_tier: null searcher"); return 1; }
    if (tier != 0 && tier != 1) { s->set_error("nnd_searcher_set_tier: tier must be 0 (automatic) or 1 (global-memory tier for every query)"); return 1; }
    s->force_big = tier == 1;
    return 0;
}

extern "C" int32_t nnd_searcher_query(nnd_searcher_t s, const float *queries, int64_t nq, int32_t k, float epsilon, int32_t *out_idx,
                                      float *out_dist) {
    if (!s) { snprintf(g_serr, sizeof(g_serr), "nnd_searcher_query: null searcher"); return 1; }
    if (k < 1 || k > 256) { s->set_error("nnd_searcher_query: k must be in 1..256 (got %d)", k); return 1; }
    // (k > 64, round 5: the result list as two or four entries per lane; the reference takes any k, pynndescent_.py:2275-2379)
    auto kq_lds = k > 128 ? k_query<false, 4> : (k > 64 ? k_query<false, 2> : k_query<false, 1>);
    auto kq_big = k > 128 ? k_query<true, 4> : (k > 64 ? k_query<true, 2> : k_query<true, 1>);
    if (nq <= 0) return 0;
    if (nq >= (int64_t)0x7FFFFFF0) { s->set_error("nnd_searcher_query: too many queries in one call"); return 1; }
    S_HIP(hipSetDevice(s->device));
    float *dq = nullptr, *dd = nullptr;
    int32_t *di = nullptr, *dlist = nullptr;
    uint8_t *dov = nullptr;
    unsigned char *scratch = nullptr;
    int rc = 0;
    const size_t per_wave = ((size_t)s->dp * 4 + Q_FRONTIER * 8 + Q_VISITED * 4 + Q_CHUNK * 8 + 15) & ~(size_t)15;
    const size_t smem = 4 * per_wave;
    const size_t per_wave_big = ((size_t)s->dp * 4 + Q_CHUNK * 8 + 15) & ~(size_t)15;
    s->last_spilled = 0;
    std::vector<uint8_t> hov((size_t)nq, 1);
    do {
        if (hipMalloc((void **)&dq, sizeof(float) * (size_t)nq * s->d) != hipSuccess || hipMalloc((void **)&di, sizeof(int32_t) * (size_t)nq * k) != hipSuccess ||
            hipMalloc((void **)&dd, sizeof(float) * (size_t)nq * k) != hipSuccess || hipMalloc((void **)&dov, (size_t)nq) != hipSuccess) { s->set_error("nnd_searcher_query: out of device memory"); rc = 1; break; }
        if (hipMemcpyAsync(dq, queries, sizeof(float) * (size_t)nq * s->d, hipMemcpyHostToDevice, s->stream) != hipSuccess) { s->set_error("H2D of the queries failed"); rc = 1; break; }
        if (!s->force_big) {
            if (smem > 64 * 1024 && hipFuncSetAttribute((const void *)kq_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
                s->set_error("nnd_searcher_query: rows of %d floats need %zu bytes of LDS per workgroup", s->d, smem); rc = 1; break;
            }
            hipLaunchKernelGGL(kq_lds, dim3((unsigned)((nq + 3) / 4)), dim3(256), smem, s->stream, s->x, s->xn2, s->dp, s->d, s->metric, s->n,
                               s->indptr, s->indices, s->hyper, s->offsets, s->children, s->tree_idx, s->n_nodes, dq, nq, k, epsilon,
                               s->min_distance, s->n_neighbors, s->seed, di, dd, dov, (const int32_t *)nullptr, 0, (unsigned char *)nullptr, (size_t)0);
            if (hipGetLastError() != hipSuccess) { s->set_error("k_query launch failed"); rc = 1; break; }
            if (hipMemcpyAsync(hov.data(), dov, (size_t)nq, hipMemcpyDeviceToHost, s->stream) != hipSuccess || hipStreamSynchronize(s->stream) != hipSuccess) {
                s->set_error("nnd_searcher_query: kernel or D2H failed: %s", hipGetErrorString(hipGetLastError())); rc = 1; break;
            }
        }
        // queries whose search outgrew the LDS structures (or all of them, nnd_searcher_set_tier(1)): global-memory tier
        std::vector<int32_t> again;
        for (int64_t i = 0; i < nq; i++)
            if (hov[(size_t)i]) again.push_back((int32_t)i);
        s->last_spilled = (int64_t)again.size();
        if (!again.empty()) {
            const size_t stride = ((size_t)Q_BIG_FRONTIER * 8 + (size_t)((s->n + 31) / 32) * 4 + 255) & ~(size_t)255;
            const size_t batch = again.size() < Q_BIG_BATCH ? again.size() : (size_t)Q_BIG_BATCH;
            if (hipMalloc((void **)&scratch, stride * batch) != hipSuccess || hipMalloc((void **)&dlist, sizeof(int32_t) * again.size()) != hipSuccess) {
                s->set_error("nnd_searcher_query: out of device memory for the global-memory tier (%zu bytes)", stride * batch); rc = 1; break;
            }
            if (hipMemcpyAsync(dlist, again.data(), sizeof(int32_t) * again.size(), hipMemcpyHostToDevice, s->stream) != hipSuccess) { s->set_error("H2D of the query list failed"); rc = 1; break; }
            for (size_t b0 = 0; b0 < again.size() && !rc; b0 += batch) {
                const int nb = (int)(again.size() - b0 < batch ? again.size() - b0 : batch);
                hipLaunchKernelGGL(kq_big, dim3((unsigned)((nb + 3) / 4)), dim3(256), 4 * per_wave_big, s->stream, s->x, s->xn2, s->dp, s->d, s->metric,
                                   s->n, s->indptr, s->indices, s->hyper, s->offsets, s->children, s->tree_idx, s->n_nodes, dq, nq, k, epsilon,
                                   s->min_distance, s->n_neighbors, s->seed, di, dd, (uint8_t *)nullptr, (const int32_t *)(dlist + b0), nb, scratch, stride);
                if (hipGetLastError() != hipSuccess) { s->set_error("k_query (global-memory tier) launch failed"); rc = 1; }
            }
            if (rc) break;
        }
        if (hipMemcpyAsync(out_idx, di, sizeof(int32_t) * (size_t)nq * k, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
            hipMemcpyAsync(out_dist, dd, sizeof(float) * (size_t)nq * k, hipMemcpyDeviceToHost, s->stream) != hipSuccess ||
            hipStreamSynchronize(s->stream) != hipSuccess) { s->set_error("nnd_searcher_query: kernel or D2H failed: %s", hipGetErrorString(hipGetLastError())); rc = 1; break; }
    } while (0);
    if (dq) (void)hipFree(dq);
    if (di) (void)hipFree(di);
    if (dd) (void)hipFree(dd);
    if (dov) (void)hipFree(dov);
    if (dlist) (void)hipFree(dlist);
    if (scratch) (void)hipFree(scratch);
    return rc;
}
