// merge.h -- wave-level insertion of a batch of candidates into one sorted k-list.
//
// Replaces checked_flagged_heap_push (reference utils.py:471-533) applied to one row: a candidate
// (q, d) is accepted iff d < worst(row) strictly (utils.py:484) and q is not already in the row
// (utils.py:489-492); the worst entry is evicted; accepted entries carry the "new" flag.
// The reference keeps a binary max-heap and pushes one candidate at a time; here the row is kept
// sorted ascending by (dist, idx) and a whole batch is merged by rank counting:
//     new position of a list entry  = old position + #{accepted candidates with a smaller key}
//     position of a candidate       = #{list entries with a smaller key} + #{candidates with a smaller key}
// which yields the k smallest keys of (row U batch) -- the same set sequential pushes produce,
// independent of arrival order (ties in distance are broken by index).
#pragma once
#include "common.h"

#define NND_MAX_K 64
#define NND_MERGE_CMAX 256

struct nnd_merge_scratch {
    uint64_t lkey[NND_MAX_K];        // sorted list keys (dist_bits<<32 | idx), EMPTY_KEY for empty slots
    uint32_t lraw[NND_MAX_K];        // raw neighbour words (idx | NEW_BIT)
    uint64_t ckey[NND_MERGE_CMAX];   // compacted accepted candidates
};

// All 64 lanes of the wave call this with the same arguments.
// cand(c, id, d) -> bool : candidate c of [0, ncand) (ids must be unique inside the batch).
// Returns the number of accepted candidates (same value on every lane).
template <typename CandFn>
__device__ __forceinline__ int nnd_merge_row(int64_t v, int k, int ks, uint32_t *__restrict__ knn_e,
                                             float *__restrict__ knn_d, nnd_merge_scratch &sc, int ncand,
                                             CandFn cand) {
    const int lane = nnd_lane();
    uint32_t *row_e = knn_e + v * ks;
    float *row_d = knn_d + v * ks;
    uint32_t e = NND_EMPTY_E;
    float d = INFINITY;
    if (lane < k) {
        e = row_e[lane];
        d = row_d[lane];
    }
    uint64_t mykey = (e == NND_EMPTY_E) ? NND_EMPTY_KEY : nnd_make_key(d, e);
    sc.lkey[lane] = mykey;
    sc.lraw[lane] = e;
    const float th = __shfl(d, k - 1, 64);  // worst distance; +inf while the row is not full
    nnd_wave_lds_sync();

    int nv = 0;
    for (int base = 0; base < ncand; base += 64) {
        int c = base + lane;
        uint32_t id = 0;
        float dc = 0.0f;
        bool ok = (c < ncand) && cand(c, id, dc);
        ok = ok && (dc < th);  // strict, utils.py:484
        if (ok) {
            for (int j = 0; j < k; j++) {  // utils.py:489-492 (LDS broadcast reads)
                uint64_t lk = sc.lkey[j];
                if (lk != NND_EMPTY_KEY && nnd_key_idx(lk) == id) {
                    ok = false;
                    break;
                }
            }
        }
        unsigned long long m = __ballot(ok);
        if (ok) sc.ckey[nv + nnd_prefix_popc(m)] = nnd_make_key(dc, id);
        nv += __popcll(m);
    }
    if (nv == 0) return 0;
    nnd_wave_lds_sync();

    // surviving list entries shift right by the number of smaller accepted candidates
    if (lane < k) {
        int shift = 0;
        for (int c = 0; c < nv; c++) shift += (sc.ckey[c] < mykey) ? 1 : 0;
        int np = lane + shift;
        if (shift > 0 && np < k) {
            row_e[np] = e;
            row_d[np] = d;
        }
    }
    int accepted = 0;
    for (int base = 0; base < nv; base += 64) {
        int c = base + lane;
        if (c < nv) {
            uint64_t key = sc.ckey[c];
            int r = 0;
            for (int j = 0; j < k; j++) r += (sc.lkey[j] < key) ? 1 : 0;
            if (r < k) {
                for (int c2 = 0; c2 < nv; c2++) r += (sc.ckey[c2] < key) ? 1 : 0;
                if (r < k) {
                    row_e[r] = nnd_key_idx(key) | NND_NEW_BIT;
                    row_d[r] = nnd_key_dist(key);
                    accepted++;
                }
            }
        }
    }
    accepted = nnd_wave_sum_i32(accepted);
    nnd_wave_lds_sync();  // scratch may be reused by the caller's next row
    return accepted;
}
