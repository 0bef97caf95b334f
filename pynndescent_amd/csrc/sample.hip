// sample.hip -- new/old candidate sampling for one NN-descent iteration.
//
// Replaces new_build_candidates (reference utils.py:221-320): every directed edge v->u of the
// current graph, with its new/old flag f, draws a uniform priority and is offered to the candidate
// list of v (forward) and of u (reverse) -- the "new" lists if f = 1, the "old" lists if f = 0;
// each list keeps the max_candidates smallest priorities with unique ids (utils.py:277-306).
// Afterwards every forward new edge whose target made it into new[v] has its flag cleared
// (utils.py:311-318).
//
// The reference makes every thread scan all n*k edges and push into the heaps of the vertices it owns.  Here, since round 5,
// the reverse offers are TRANSPOSED (second half of this file: k_rev_place / k_rev_import / k_rev_select / k_rev_fill) -- placed
// by target bucket and appended to the targets' slot banks in LDS: a bank that receives no more offers than it has slots keeps
// every one of them, which is what the reference's heaps do.  The priority of the offer v -> u is mix32(v ^ salt(u)): mix32 is a
// bijection, so a 4-byte slot word IS the source (v = unmix32(word) ^ salt(u)) and its order is a fresh pseudo-random order of the
// sources for every target and iteration.  Forward priorities are hash(seed, iteration, v, u); both kinds are uniform 32-bit words,
// ranked together by k_rev_select (k <= 32) or by k_sample_select / _h / _wide on banks k_rev_fill wrote to rbuf.
//
// The first half is the form of rounds 1-4, kept behind NND_FLAG_TEST_SAMPLE_ATOMIC (comparison, A/B timing):
//   k_sample_reverse : one thread per edge scatters the reverse offer into a bank of RCAP hashed slots per (target, class) with a
//                      32-bit atomicMin on the offer's priority -- order independent, but two offers that share a slot lose one;
//   k_sample_select  : one wave per vertex gathers its k forward offers and its reverse slots, drops reverse offers that duplicate
//                      a forward id, ranks by priority and writes the max_candidates smallest of each class, clears the flags of
//                      the sampled forward-new edges and re-arms the reverse slots.
#include "common.h"
#include "state.h"

// A reverse offer that repeats a forward edge of the same class (v and u are each other's neighbours) is dropped: the item keeps
// its forward draw.  The reference pushes both into the same heap with ONE shared priority per edge and its duplicate check
// rejects whichever push comes second while the first is still in the heap (utils.py:277-306, 427-430): a mutual neighbour gets
// a second draw only if the first was evicted before the second edge is scanned -- an order-dependent rule.  Measured (round 6,
// tests/test_gpu_kernels.py::test_candidate_lists_have_the_reference_algorithms_distribution): share of forward (own-row)
// entries in the new lists of a first pass 0.757 here, 0.779 in the reference, and 0.799 when the item keeps the SMALLER of the
// two draws (built and measured: recall@10 0.6099 -> 0.6090 on 200 000 iid Gaussian points x 32, 0.9944 -> 0.9938 at the 1 M
// bench set, +0.3 ms of sampling: dropped).  The reference sits between the two order-independent rules; the one kept is the one
// with the better graph.
// per-target, per-iteration salt of the reverse priorities
__device__ __forceinline__ uint32_t nnd_offer_salt(uint32_t it_seed, uint32_t u) { return nnd_hash2(it_seed ^ 0x3C6EF372u, u); }

// slot of source v in target u's bank.  It depends on the TARGET as well (through its salt): until the end of round 4 it was
// a function of (iteration, v) alone, so two sources that shared a slot shared it in EVERY bank of the iteration -- the
// pair could not meet as reverse candidates of any common neighbour, and two vertices with a common neighbour are exactly
// the pairs NN-descent exists to evaluate.  Measured at 1.2 M x 100 cosine (tools/recall_study.py, 20 000 rows, 4 seeds):
// see DESIGN.md section 2.
__device__ __forceinline__ uint32_t nnd_offer_slot(uint32_t salt_u, uint32_t v, int rcap) { return nnd_hash2(salt_u ^ 0x68E31DA4u, v) & (uint32_t)(rcap - 1); }
// word of u's bank pair [old class | new class] that the offer v -> u competes for.  WIDE: before the first sampling pass
// every edge is new (ctx->all_new), the old-class bank would stay empty -- the new class takes both, 2 * rcap slots.  That
// is the iteration with the most offers per bank (in-degree ~k, all of one class) and the one that moves the graph most:
// with 32 slots a fifth of its reverse offers were lost to collisions and the lists held too few reverse candidates (the
// forward ones are each other's leaf mates and mostly compared already): recall@10 0.9884 -> see DESIGN.md section 2
// (the CPU oracle: 0.9901; 64 slots per class everywhere: 0.9893 and 3 % more time).
__device__ __forceinline__ int64_t nnd_offer_addr(uint32_t u, uint32_t cls, uint32_t salt_u, uint32_t v, int rcap, int wide) {
    return wide ? (int64_t)u * 2 * rcap + nnd_offer_slot(salt_u, v, 2 * rcap) : ((int64_t)u * 2 + cls) * rcap + nnd_offer_slot(salt_u, v, rcap);
}
// slot word of vertex u's bank -> the 64-bit item key (priority << 32 | source) the selection ranks
__device__ __forceinline__ uint64_t nnd_offer_key(uint32_t slot_word, uint32_t salt_u) {
    return ((uint64_t)slot_word << 32) | (uint64_t)(nnd_unmix32(slot_word) ^ salt_u);
}

// Two passes over the edges.  pass 0: NEW edges -- reverse offer into the target's "new" slots, and both endpoints are
// marked active (they will hold at least one new candidate).  pass 1: OLD edges -- offered only to ACTIVE targets: a
// vertex without new candidates does no join (utils.py:611-613), so its old list is never read; late iterations, where
// almost every edge is old and almost every vertex inactive, then cost a scan instead of n*k atomics.
// Thread layout: blockDim = (KSP, 256 / KSP) with KSP = the row stride rounded up to a power of two, x = slot, y = row:
// no division per edge.  Rows are visited in `order` (spatially coherent, one contiguous eighth per XCD): the targets
// of a window of rows are each other's neighbours, so the offers of a window land in a few slot banks that stay in L2.
__global__ __launch_bounds__(256) void k_sample_reverse(const uint32_t *__restrict__ knn_e, int64_t row_lo, int64_t n, int k, int ks, uint32_t it_seed,
                                                        uint32_t *__restrict__ rbuf, int rcap, int64_t own_lo, int64_t own_hi, int pass,
                                                        uint8_t *__restrict__ active, const int32_t *__restrict__ order, int wide) {
    int64_t b = blockIdx.x;
    if ((gridDim.x & 7) == 0) b = (b & 7) * (gridDim.x >> 3) + (b >> 3);
    const int64_t g = row_lo + b * blockDim.y + threadIdx.y;  // rows [row_lo, n): all of them, or the owned slice (sharded)
    const int j = threadIdx.x;
    if (g >= n || j >= k) return;
    const int64_t v = order ? (int64_t)order[g] : g;
    uint32_t e = knn_e[v * ks + j];
    if (e == NND_EMPTY_E) return;
    uint32_t u = e & NND_IDX_MASK;
    uint32_t cls = e >> 31;  // 1 = new
    if (cls != (uint32_t)(pass == 0)) return;
    if (pass == 0 && v >= own_lo && v < own_hi) active[v] = 1;  // forward new edge
    if ((int64_t)u < own_lo || (int64_t)u >= own_hi) return;  // owner-computes: only targets this handle owns (utils.py:270-273)
    if (pass == 0) active[u] = 1;
    else if (!active[u]) return;
    const uint32_t salt = nnd_offer_salt(it_seed, u);
    atomicMin(&rbuf[nnd_offer_addr(u, cls, salt, (uint32_t)v, rcap, wide)], nnd_mix32((uint32_t)v ^ salt));  // (a priority equal to NND_EMPTY_SLOT, 1 in 2^32, is a lost offer)
}

#define SAMPLE_MAX_ITEMS 128  // k (<=64) forward + rcap (<=64) reverse offers per class

struct sample_scratch {
    uint64_t key[2][SAMPLE_MAX_ITEMS];  // [class][item] priority<<32 | id
};

__global__ __launch_bounds__(256) void k_sample_select(uint32_t *__restrict__ knn_e, int64_t n, int k, int ks, int mc,
                                                       int mcp, uint32_t it_seed, uint32_t *__restrict__ rbuf, int rcap,
                                                       int32_t *__restrict__ cand, int64_t own_lo, int64_t own_hi,
                                                       const uint8_t *__restrict__ active, int wide) {
    __shared__ sample_scratch scr[4];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t v = own_lo + (int64_t)blockIdx.x * 4 + w;
    if (v >= own_hi) return;
    sample_scratch &sc = scr[w];
    if (!active[v]) {  // no new candidate can reach v: empty new list, nothing else to do (no offers were stored for it)
        for (int j = lane; j < mcp; j += 64) cand[v * 2 * mcp + j] = -1;
        return;
    }

    uint32_t e = NND_EMPTY_E;
    if (lane < k) e = knn_e[v * ks + lane];
    const bool valid = e != NND_EMPTY_E;
    const uint32_t u = e & NND_IDX_MASK;
    const uint32_t cls = e >> 31;
    const uint64_t fkey = ((uint64_t)nnd_hash3(it_seed, (uint32_t)v, u) << 32) | u;
    int cnt[2];
#pragma unroll
    for (int c = 0; c < 2; c++) {
        bool mine = valid && cls == (uint32_t)c;
        unsigned long long m = __ballot(mine);
        if (mine) sc.key[c][nnd_prefix_popc(m)] = fkey;
        cnt[c] = __popcll(m);
    }
    nnd_wave_lds_sync();
    const int nfwd[2] = {cnt[0], cnt[1]};
    const uint32_t salt = nnd_offer_salt(it_seed, (uint32_t)v);
    // reverse offers: when both classes' slot banks fit one wave (2 * rcap <= 64) they are fetched and screened together
    if (2 * rcap <= 64) {
        const int c = (wide || lane >= rcap) ? 1 : 0;  // wide: both banks hold new-class offers (nnd_offer_addr)
        uint32_t *slots = rbuf + v * 2 * rcap;  // [class 0 | class 1] are adjacent
        uint32_t rw = NND_EMPTY_SLOT;
        if (lane < 2 * rcap) {
            rw = slots[lane];
            if (rw != NND_EMPTY_SLOT) slots[lane] = NND_EMPTY_SLOT;  // re-arm for the next iteration
        }
        bool ok = rw != NND_EMPTY_SLOT;
        const uint64_t rk = nnd_offer_key(rw, salt);
        if (ok) {  // utils.py:427-430: an id already in the list is not pushed again
            const uint32_t src = (uint32_t)rk;
            const int nf = c ? nfwd[1] : nfwd[0];
            for (int j = 0; j < nf; j++) ok &= ((uint32_t)sc.key[c][j] != src);
        }
#pragma unroll
        for (int cc = 0; cc < 2; cc++) {
            const bool mine = ok && c == cc;
            const unsigned long long m = __ballot(mine);
            if (mine) sc.key[cc][cnt[cc] + nnd_prefix_popc(m)] = rk;
            cnt[cc] += __popcll(m);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 2; c++) {  // (never wide: the launcher asks for it only with 32 slots per class)
            uint32_t *slots = rbuf + (v * 2 + c) * rcap;
            for (int s0 = 0; s0 < rcap; s0 += 64) {
                int s = s0 + lane;
                uint32_t rw = NND_EMPTY_SLOT;
                if (s < rcap) {
                    rw = slots[s];
                    if (rw != NND_EMPTY_SLOT) slots[s] = NND_EMPTY_SLOT;  // re-arm for the next iteration
                }
                bool ok = rw != NND_EMPTY_SLOT;
                const uint64_t rk = nnd_offer_key(rw, salt);
                if (ok) {  // utils.py:427-430: an id already in the list is not pushed again
                    uint32_t src = (uint32_t)rk;
                    for (int j = 0; j < nfwd[c]; j++) ok &= ((uint32_t)sc.key[c][j] != src);
                }
                unsigned long long m = __ballot(ok);
                if (ok) sc.key[c][cnt[c] + nnd_prefix_popc(m)] = rk;
                cnt[c] += __popcll(m);
            }
        }
    }
    nnd_wave_lds_sync();

    int32_t *out = cand + v * 2 * mcp;
    int my_rank = 1 << 30;  // rank of this lane's forward new edge among the new offers
    const int my_item = (valid && cls == 1u) ? nnd_prefix_popc(__ballot(valid && cls == 1u)) : -1;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int M = cnt[c];
        int32_t *dst = out + (c == 1 ? 0 : mcp);  // layout [new | old]
        for (int i0 = 0; i0 < M; i0 += 64) {
            int i = i0 + lane;
            int r = 1 << 30;
            uint64_t key = 0;
            if (i < M) {
                key = sc.key[c][i];
                r = 0;
                for (int j = 0; j < M; j++) r += (sc.key[c][j] < key) ? 1 : 0;
                if (r < mc) dst[r] = (int32_t)(uint32_t)key;
            }
            if (c == 1 && i0 == 0) {  // forward items sit at the front in lane order: hand each its rank (k <= 64)
                const int got = __shfl(r, my_item >= 0 ? my_item : 0, 64);
                if (my_item >= 0) my_rank = got;
            }
        }
        int filled = M < mc ? M : mc;
        for (int j = filled + lane; j < mcp; j += 64) dst[j] = -1;
    }
    // flag reset (utils.py:311-318): a forward new edge that was sampled becomes old
    if (valid && cls == 1u && my_rank < mc) knn_e[v * ks + lane] = u;
}

// 64 < k <= NND_WIDE_K (256): the same selection with up to NND_WIDE_U forward edges per lane (the reference has no bound on
// n_neighbors).
// Items of a class: <= k forward + rcap reverse offers; ranks by counting over the LDS copy; the rank of every forward
// new edge goes through LDS (rank_new) to the lane that holds the edge, which clears its flag when it was sampled.
__global__ __launch_bounds__(256) void k_sample_select_wide(uint32_t *__restrict__ knn_e, int64_t n, int k, int ks, int mc,
                                                            int mcp, uint32_t it_seed, uint32_t *__restrict__ rbuf, int rcap,
                                                            int32_t *__restrict__ cand, int64_t own_lo, int64_t own_hi,
                                                            const uint8_t *__restrict__ active) {
    constexpr int MAXI = NND_WIDE_K + 128;  // k forward edges + rcap <= 128 reverse offers per class
    __shared__ uint64_t skey[4][2][MAXI];
    __shared__ int srank[4][NND_WIDE_K];  // rank among the new offers of forward item i (class new)
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t v = own_lo + (int64_t)blockIdx.x * 4 + w;
    if (v >= own_hi) return;
    if (!active[v]) {
        for (int j = lane; j < mcp; j += 64) cand[v * 2 * mcp + j] = -1;
        return;
    }
    uint64_t(*key)[MAXI] = skey[w];
    uint32_t e[NND_WIDE_U];
    int item[NND_WIDE_U];  // index of my forward edge among its class's items, or -1
    int cnt[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < NND_WIDE_U; u++) {
        const int j = lane + 64 * u;
        e[u] = j < k ? knn_e[v * ks + j] : NND_EMPTY_E;
        item[u] = -1;
    }
#pragma unroll
    for (int c = 0; c < 2; c++) {
#pragma unroll
        for (int u = 0; u < NND_WIDE_U; u++) {
            const bool mine = e[u] != NND_EMPTY_E && (e[u] >> 31) == (uint32_t)c;
            const unsigned long long m = __ballot(mine);
            if (mine) {
                const uint32_t tgt = e[u] & NND_IDX_MASK;
                item[u] = cnt[c] + nnd_prefix_popc(m);
                key[c][item[u]] = ((uint64_t)nnd_hash3(it_seed, (uint32_t)v, tgt) << 32) | tgt;
            }
            cnt[c] += __popcll(m);
        }
    }
    nnd_wave_lds_sync();
    const int nfwd[2] = {cnt[0], cnt[1]};
    const uint32_t salt = nnd_offer_salt(it_seed, (uint32_t)v);
#pragma unroll
    for (int c = 0; c < 2; c++) {
        uint32_t *slots = rbuf + (v * 2 + c) * rcap;
        for (int s0 = 0; s0 < rcap; s0 += 64) {
            const int sidx = s0 + lane;
            uint32_t rw = NND_EMPTY_SLOT;
            if (sidx < rcap) {
                rw = slots[sidx];
                if (rw != NND_EMPTY_SLOT) slots[sidx] = NND_EMPTY_SLOT;  // re-arm for the next iteration
            }
            bool ok = rw != NND_EMPTY_SLOT;
            const uint64_t rk = nnd_offer_key(rw, salt);
            if (__ballot(ok)) {  // utils.py:427-430: an id already in the list is not pushed again
                const uint32_t src = (uint32_t)rk;
                for (int j = 0; j < nfwd[c]; j++) ok = ok && ((uint32_t)key[c][j] != src);
            }
            const unsigned long long m = __ballot(ok);
            if (ok) key[c][cnt[c] + nnd_prefix_popc(m)] = rk;
            cnt[c] += __popcll(m);
        }
    }
    nnd_wave_lds_sync();
    int32_t *out = cand + v * 2 * mcp;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int M = cnt[c];
        int32_t *dst = out + (c == 1 ? 0 : mcp);  // layout [new | old]
        for (int i0 = 0; i0 < M; i0 += 64) {
            const int i = i0 + lane;
            if (i < M) {
                const uint64_t kk = key[c][i];
                int r = 0;
                for (int j = 0; j < M; j++) r += (key[c][j] < kk) ? 1 : 0;
                if (r < mc) dst[r] = (int32_t)(uint32_t)kk;
                if (c == 1 && i < nfwd[1]) srank[w][i] = r;
            }
        }
        const int filled = M < mc ? M : mc;
        for (int j = filled + lane; j < mcp; j += 64) dst[j] = -1;
    }
    nnd_wave_lds_sync();
    // flag reset (utils.py:311-318): a forward new edge that was sampled becomes old
#pragma unroll
    for (int u = 0; u < NND_WIDE_U; u++) {
        const int j = lane + 64 * u;
        if (e[u] != NND_EMPTY_E && (e[u] >> 31) == 1u && item[u] >= 0 && srank[w][item[u]] < mc) knn_e[v * ks + j] = e[u] & NND_IDX_MASK;
    }
}

// k <= 32 with 32 reverse slots per class (max_candidates <= 32, the BASELINE regime): TWO vertices per wave, 32 lanes
// each.  k_sample_select keeps ~30 of 64 lanes busy and is bound by instruction issue; here the loops (duplicate screen
// over the forward ids, rank count over the items) run for both vertices at once.  Lane j of a half holds forward edge
// j, reverse slots j (old class) and 32 + j (new class), and items j and 32 + j of each class's list (<= k + 32 <= 64
// items).  Same keys, same duplicate rule, same ranks: the candidate lists are identical to k_sample_select's.
// WIDE (the first pass of a build: every edge new, both banks hold new-class offers, nnd_offer_addr): one list of up to
// k + 64 <= 96 items -- the two classes' LDS arrays of a half-wave are adjacent and are used as one, three items per lane.
// The selection for one vertex per half-wave (both halves of the wave at once): forward edge word `e` of lane j, reverse slot
// words rw0 (old-class bank, slot j) and rw1 (new-class bank) however the caller obtained them -- from rbuf
// (k_sample_select_h) or from the LDS banks of the bucketed reverse pass (k_rev_select).  `sk`: the half-wave's LDS lists.
// k <= 32 with 32 reverse slots per class (max_candidates <= 32, the BASELINE regime): TWO vertices per wave, 32 lanes
// each.  k_sample_select keeps ~30 of 64 lanes busy and is bound by instruction issue; here the loops (duplicate screen
// over the forward ids, rank count over the items) run for both vertices at once.  Lane j of a half holds forward edge
// j, reverse slots j (old class) and 32 + j (new class), and items j and 32 + j of each class's list (<= k + 32 <= 64
// items).  Same keys, same duplicate rule, same ranks: the candidate lists are identical to k_sample_select's.
// WIDE (the first pass of a build: every edge new, both banks hold new-class offers, nnd_offer_addr): one list of up to
// k + 64 <= 96 items -- the two classes' LDS arrays of a half-wave are adjacent and are used as one, three items per lane.
// The selection for one vertex per half-wave (both halves of the wave at once): forward edge word `e` of lane j, reverse slot
// words rw0 (old-class bank, slot j) and rw1 (new-class bank) however the caller obtained them -- from rbuf
// (k_sample_select_h) or from the LDS banks of the bucketed reverse pass (k_rev_select).  `sk`: the half-wave's LDS lists.
template <bool WIDE>
__device__ __forceinline__ void nnd_select_half(uint32_t *__restrict__ knn_e, int k, int ks, int mc, int mcp, uint32_t it_seed,
                                                int32_t *__restrict__ cand, int64_t vv, bool act, uint32_t e, uint32_t rw0, uint32_t rw1,
                                                uint64_t (*sk)[64], int j, int hb) {
    const uint32_t salt = nnd_offer_salt(it_seed, (uint32_t)vv);
    const uint64_t rk0 = nnd_offer_key(rw0, salt), rk1 = nnd_offer_key(rw1, salt);
    const bool valid = e != NND_EMPTY_E;
    const uint32_t u = e & NND_IDX_MASK;
    const uint32_t cls = e >> 31;
    const uint64_t fkey = ((uint64_t)nnd_hash3(it_seed, (uint32_t)vv, u) << 32) | u;
    const uint32_t below = (1u << j) - 1u;
    int cnt[2];
    uint32_t fmask1 = 0;  // forward new edges of my half
#pragma unroll
    for (int c = 0; c < 2; c++) {
        const bool mine = valid && cls == (uint32_t)c;
        const uint32_t hm = (uint32_t)(__ballot(mine) >> hb);
        if (mine) sk[c][__popc(hm & below)] = fkey;
        cnt[c] = __popc(hm);
        if (c == 1) fmask1 = hm;
    }
    nnd_wave_lds_sync();
    const int nf0 = cnt[0], nf1 = cnt[1];
    int32_t *out = cand + vv * 2 * mcp;
    int my_rank = 1 << 30;  // rank of this lane's forward new edge among the new offers
    const int my_item = (valid && cls == 1u) ? __popc(fmask1 & below) : -1;
    if constexpr (WIDE) {
        // every forward edge is new (nf0 == 0): the items live in ONE list fl[0 .. M), forward edges first.  fl aliases
        // sk[0] | sk[1]; the forward keys were written to sk[1] = fl + 64 above and move to the front here.
        uint64_t *fl = &sk[0][0];
        const uint64_t fk = j < nf1 ? sk[1][j] : NND_EMPTY_KEY;
        nnd_wave_lds_sync();
        if (j < nf1) fl[j] = fk;
        nnd_wave_lds_sync();
        bool ok0 = rw0 != NND_EMPTY_SLOT, ok1 = rw1 != NND_EMPTY_SLOT;
        {
            const int a0 = __builtin_amdgcn_readlane(nf1, 0), a1 = __builtin_amdgcn_readlane(nf1, 32);
            const int nfm = a0 > a1 ? a0 : a1;  // wave-uniform trip count
            const uint32_t s0 = (uint32_t)rk0, s1 = (uint32_t)rk1;
            for (int q = 0; q < nfm; q++) {
                const uint32_t f = (uint32_t)fl[q];
                ok0 = ok0 && !(q < nf1 && f == s0);
                ok1 = ok1 && !(q < nf1 && f == s1);
            }
        }
        int M = nf1;
        {   // bank order [old-class bank | new-class bank] = the order k_sample_select walks the 64 slots in
            const uint32_t hm0 = (uint32_t)(__ballot(ok0) >> hb), hm1 = (uint32_t)(__ballot(ok1) >> hb);
            if (ok0) fl[M + __popc(hm0 & below)] = rk0;
            M += __popc(hm0);
            if (ok1) fl[M + __popc(hm1 & below)] = rk1;
            M += __popc(hm1);
        }
        nnd_wave_lds_sync();
        const int m0 = __builtin_amdgcn_readlane(M, 0), m1 = __builtin_amdgcn_readlane(M, 32);
        const int mm = m0 > m1 ? m0 : m1;  // wave-uniform trip count
        const uint64_t key0 = j < M ? fl[j] : NND_EMPTY_KEY, key1 = 32 + j < M ? fl[32 + j] : NND_EMPTY_KEY, key2 = 64 + j < M ? fl[64 + j] : NND_EMPTY_KEY;
        int r0 = 0, r1 = 0, r2 = 0;
        for (int q = 0; q < mm; q++) {
            const uint64_t kq = fl[q];
            const bool in = q < M;
            r0 += (in && kq < key0) ? 1 : 0;
            r1 += (in && kq < key1) ? 1 : 0;
            r2 += (in && kq < key2) ? 1 : 0;
        }
        if (act) {
            if (j < M && r0 < mc) out[r0] = (int32_t)(uint32_t)key0;
            if (32 + j < M && r1 < mc) out[r1] = (int32_t)(uint32_t)key1;
            if (64 + j < M && r2 < mc) out[r2] = (int32_t)(uint32_t)key2;
            const int filled = M < mc ? M : mc;
            for (int q = filled + j; q < mcp; q += 32) out[q] = -1;
            for (int q = j; q < mcp; q += 32) out[mcp + q] = -1;  // no old candidates yet
        }
        // forward item i < k <= 32 is item 0 of lane i of my half
        const int got = __builtin_amdgcn_ds_bpermute((hb + (my_item >= 0 ? my_item : 0)) << 2, r0);
        if (my_item >= 0) my_rank = got;
        if (act && valid && cls == 1u && my_rank < mc) knn_e[vv * ks + j] = u;
        return;
    }
    // utils.py:427-430: an id already in the list is not pushed again (a reverse offer that repeats a forward edge)
    bool ok0 = rw0 != NND_EMPTY_SLOT, ok1 = rw1 != NND_EMPTY_SLOT;
    {
        const int a = nf0 > nf1 ? nf0 : nf1;
        const int a0 = __builtin_amdgcn_readlane(a, 0), a1 = __builtin_amdgcn_readlane(a, 32);
        const int nfm = a0 > a1 ? a0 : a1;  // wave-uniform trip count
        const uint32_t s0 = (uint32_t)rk0, s1 = (uint32_t)rk1;
        for (int q = 0; q < nfm; q++) {
            const uint32_t f0 = (uint32_t)sk[0][q], f1 = (uint32_t)sk[1][q];
            ok0 = ok0 && !(q < nf0 && f0 == s0);
            ok1 = ok1 && !(q < nf1 && f1 == s1);
        }
    }
    {
        const uint32_t hm0 = (uint32_t)(__ballot(ok0) >> hb), hm1 = (uint32_t)(__ballot(ok1) >> hb);
        if (ok0) sk[0][cnt[0] + __popc(hm0 & below)] = rk0;
        if (ok1) sk[1][cnt[1] + __popc(hm1 & below)] = rk1;
        cnt[0] += __popc(hm0);
        cnt[1] += __popc(hm1);
    }
    nnd_wave_lds_sync();

#pragma unroll
    for (int c = 0; c < 2; c++) {
        const int M = cnt[c];
        const int m0 = __builtin_amdgcn_readlane(M, 0), m1 = __builtin_amdgcn_readlane(M, 32);
        const int mm = m0 > m1 ? m0 : m1;  // wave-uniform trip count
        int32_t *dst = out + (c == 1 ? 0 : mcp);  // layout [new | old]
        const uint64_t key0 = j < M ? sk[c][j] : NND_EMPTY_KEY, key1 = 32 + j < M ? sk[c][32 + j] : NND_EMPTY_KEY;
        int r0 = 0, r1 = 0;
        for (int q = 0; q < mm; q++) {
            const uint64_t kq = sk[c][q];
            const bool in = q < M;
            r0 += (in && kq < key0) ? 1 : 0;
            r1 += (in && kq < key1) ? 1 : 0;
        }
        if (act) {
            if (j < M && r0 < mc) dst[r0] = (int32_t)(uint32_t)key0;
            if (32 + j < M && r1 < mc) dst[r1] = (int32_t)(uint32_t)key1;
            const int filled = M < mc ? M : mc;
            for (int q = filled + j; q < mcp; q += 32) dst[q] = -1;
        }
        if (c == 1) {  // forward items sit at the front in lane order: item i < k <= 32 is item 0 of lane i of my half
            const int got = __builtin_amdgcn_ds_bpermute((hb + (my_item >= 0 ? my_item : 0)) << 2, r0);
            if (my_item >= 0) my_rank = got;
        }
    }
    // flag reset (utils.py:311-318): a forward new edge that was sampled becomes old
    if (act && valid && cls == 1u && my_rank < mc) knn_e[vv * ks + j] = u;
}

// One class of one vertex per GROUP of LW lanes (round 6; used with LW = 16: FOUR vertices per wave -- rows of up to 16
// neighbours with up to 16 candidates per class, the BASELINE regime; with 32 lanes per vertex and lists of up to 32 the form
// measured 0.4 ms per build SLOWER than nnd_select_half above at k = 20 and k = 30, which therefore keeps those regimes).  Rounds 2-5 ranked EVERY item of a class against every
// other one -- k forward edges + up to 32 (first pass: 64) reverse offers, two or three items per lane of a half-wave, a loop of
// up to 47 (79) LDS broadcasts with a 64-bit compare per item: 400 of the 500 us of k_rev_select
// (profiles/r06_sample_select_floor.log).  Only the max_candidates smallest keys matter, and the keys' priority words are
// uniform 32-bit hashes: the items whose priority lies under a threshold tau -- chosen so that ~max_candidates + 10 pass -- are
// compacted (at most two per lane), the reverse offers among THEM are screened against the forward ids, and only they are
// ranked.  An item's rank among the survivors is its rank among all items (every smaller key passed too), so the lists are the
// ones the full ranking writes, entry for entry (tests/test_gpu_kernels.py: exact-sample test, half-wave = wave test).  tau
// moves (bisection) in the few per cent of the cases where fewer than max_candidates or more than 2 LW items pass.
//   F = list[0 .. LW): the class's forward keys (the ids the duplicate screen reads); C = list[LW .. 3 LW): the survivors.
// Returns the rank of this lane's forward item (1 << 30: not sampled).
template <int LW>
__device__ __forceinline__ int nnd_group_max(int v) {  // the largest v over the wave's groups (v is uniform inside a group)
    int m = __builtin_amdgcn_readlane(v, 0);
#pragma unroll
    for (int g = LW; g < 64; g += LW) {
        const int o = __builtin_amdgcn_readlane(v, g);
        m = o > m ? o : m;
    }
    return m;
}
template <int LW, int NR>
__device__ __forceinline__ int nnd_select_class(bool act, bool fvalid, uint64_t fkey, const bool (&rvalid)[NR], const uint64_t (&rkey)[NR], int mc,
                                                int mcp, int32_t *dst, uint64_t *list, int j, int gb) {
    constexpr uint32_t GM = LW == 32 ? 0xFFFFFFFFu : ((1u << (LW & 31)) - 1u);
    constexpr int CAP = 2 * LW;
    uint64_t *F = list, *C = list + LW;
    const uint32_t below = (1u << j) - 1u;
    const uint32_t fmask = (uint32_t)(__ballot(fvalid) >> gb) & GM;
    const int nf = __popc(fmask);
    if (fvalid) F[__popc(fmask & below)] = fkey;
    int M = nf;
#pragma unroll
    for (int i = 0; i < NR; i++) M += __popc((uint32_t)(__ballot(rvalid[i]) >> gb) & GM);
    const uint32_t fp = (uint32_t)(fkey >> 32);
    uint32_t rp[NR];
#pragma unroll
    for (int i = 0; i < NR; i++) rp[i] = (uint32_t)(rkey[i] >> 32);
    uint32_t tau = 0xFFFFFFFFu, lo = 0u, hi = 0xFFFFFFFFu;
    if (M > CAP || M > 2 * mc + 4) {  // (a list that is not much longer than what is kept is ranked whole: nothing to gain from a threshold)
        const float t = (float)(mc + 10) * 4294967296.0f / (float)M;
        tau = t >= 4294967040.0f ? 0xFFFFFFFFu : (uint32_t)t;
    }
    const int nfm = nnd_group_max<LW>(nf);  // wave-uniform trip count of the screen
    int count = 0, cv = 0, tpos = -1, cn = 0;
    uint64_t key0 = NND_EMPTY_KEY, key1 = NND_EMPTY_KEY;
    bool dup0 = false, dup1 = false;
#pragma unroll 1
    for (int iter = 0; iter < 30; iter++) {
        nnd_wave_lds_sync();  // F is written / the previous round's reads of C are done
        const bool fpass = fvalid && fp <= tau;
        const uint32_t pm = (uint32_t)(__ballot(fpass) >> gb) & GM;
        const int npf = __popc(pm);
        tpos = fpass ? __popc(pm & below) : -1;
        if (fpass) C[tpos] = fkey;  // (npf <= nf <= LW)
        int pos = npf;
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const bool rpass = rvalid[i] && rp[i] <= tau;
            const uint32_t m = (uint32_t)(__ballot(rpass) >> gb) & GM;
            const int p = pos + __popc(m & below);
            if (rpass && p < CAP) C[p] = rkey[i];
            pos += __popc(m);
        }
        count = pos;
        nnd_wave_lds_sync();
        cn = count < CAP ? count : CAP;
        const bool two = nnd_group_max<LW>(cn) > LW;  // wave-uniform: some group holds a second survivor per lane
        key0 = j < cn ? C[j] : NND_EMPTY_KEY;
        key1 = (two && LW + j < cn) ? C[LW + j] : NND_EMPTY_KEY;
        // utils.py:427-430: an id already in the list is not pushed again (a reverse offer that repeats a forward edge of the class)
        const bool rev0 = j >= npf && j < cn, rev1 = two && LW + j < cn;  // (survivors LW .. are reverse offers: npf <= LW)
        dup0 = dup1 = false;
        if (__ballot(rev0 || rev1)) {
            const uint32_t s0 = (uint32_t)key0, s1 = (uint32_t)key1;
            if (two) {
#pragma unroll 4
                for (int q = 0; q < nfm; q++) {
                    const uint32_t f = (uint32_t)F[q];
                    dup0 = dup0 || (rev0 && q < nf && f == s0);
                    dup1 = dup1 || (rev1 && q < nf && f == s1);
                }
            } else {
#pragma unroll 4
                for (int q = 0; q < nfm; q++) dup0 = dup0 || (rev0 && q < nf && (uint32_t)F[q] == s0);
            }
        }
        cv = cn - __popc((uint32_t)(__ballot(dup0) >> gb) & GM) - __popc((uint32_t)(__ballot(dup1) >> gb) & GM);
        const bool ok = count <= CAP && (cv >= mc || tau == 0xFFFFFFFFu);
        if (!__ballot(!ok)) break;  // every vertex of the wave is settled
        if (!ok) {
            if (count > CAP) {
                hi = tau;
                tau = lo + (hi - lo) / 2;
            } else {
                lo = tau;
                const uint64_t up = (uint64_t)tau + (tau >> 1) + 1u;
                tau = hi == 0xFFFFFFFFu ? (up >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)up) : lo + (hi - lo) / 2 + 1u;
            }
        }
    }
    if (dup0) key0 = NND_EMPTY_KEY;
    if (dup1) key1 = NND_EMPTY_KEY;
    const int cmax = nnd_group_max<LW>(cn);  // wave-uniform trip count
    nnd_wave_lds_sync();  // every lane holds its survivors: C is rewritten with the screened keys
    C[j] = key0;
    if (cmax > LW) C[LW + j] = key1;
    nnd_wave_lds_sync();
    int r0 = 0, r1 = 0;
    if (cmax > LW) {
#pragma unroll 4
        for (int q = 0; q < cmax; q++) {
            const uint64_t kq = C[q];
            const bool in = q < cn;
            r0 += (in && kq < key0) ? 1 : 0;
            r1 += (in && kq < key1) ? 1 : 0;
        }
    } else {
#pragma unroll 4
        for (int q = 0; q < cmax; q++) r0 += (q < cn && C[q] < key0) ? 1 : 0;
    }
    if (act) {
        if (key0 != NND_EMPTY_KEY && r0 < mc) dst[r0] = (int32_t)(uint32_t)key0;
        if (key1 != NND_EMPTY_KEY && r1 < mc) dst[r1] = (int32_t)(uint32_t)key1;
        const int filled = cv < mc ? cv : mc;
        for (int q = filled + j; q < mcp; q += LW) dst[q] = -1;
    }
    const int got = __builtin_amdgcn_ds_bpermute((gb + (tpos >= 0 ? tpos : 0)) << 2, r0);
    return tpos >= 0 ? got : (1 << 30);
}

// The selection for one vertex per group of LW lanes: forward edge word `e` of lane j of the group, the lane's NRC reverse slot
// words per class (rwo: old-class bank, rwn: new-class bank; slot i * LW + j) however the caller obtained them -- from rbuf
// (k_sample_select_h) or from the LDS banks of the bucketed reverse pass (k_rev_select).  `list`: 3 LW keys of LDS per group.
// WIDE (the first pass of a build: every edge new, both banks hold new-class offers, nnd_offer_addr): one class.
template <bool WIDE, int LW, int NRC>
__device__ __forceinline__ void nnd_select_group(uint32_t *__restrict__ knn_e, int k, int ks, int mc, int mcp, uint32_t it_seed,
                                                 int32_t *__restrict__ cand, int64_t vv, bool act, uint32_t e, const uint32_t (&rwo)[NRC],
                                                 const uint32_t (&rwn)[NRC], uint64_t *list, int j, int gb) {
    const uint32_t salt = nnd_offer_salt(it_seed, (uint32_t)vv);
    const bool valid = e != NND_EMPTY_E;
    const uint32_t u = e & NND_IDX_MASK;
    const uint32_t cls = e >> 31;
    const uint64_t fkey = ((uint64_t)nnd_hash3(it_seed, (uint32_t)vv, u) << 32) | u;
    int32_t *out = cand + vv * 2 * mcp;  // layout [new | old]
    int my_rank;  // rank of this lane's forward new edge among the new offers
    if constexpr (WIDE) {
        bool rv[2 * NRC];
        uint64_t rk[2 * NRC];
#pragma unroll
        for (int i = 0; i < NRC; i++) {
            rv[i] = rwo[i] != NND_EMPTY_SLOT;
            rk[i] = nnd_offer_key(rwo[i], salt);
            rv[NRC + i] = rwn[i] != NND_EMPTY_SLOT;
            rk[NRC + i] = nnd_offer_key(rwn[i], salt);
        }
        my_rank = nnd_select_class<LW, 2 * NRC>(act, valid, fkey, rv, rk, mc, mcp, out, list, j, gb);
        if (act)
            for (int q = j; q < mcp; q += LW) out[mcp + q] = -1;  // no old candidates yet
    } else {
        bool rv0[NRC], rv1[NRC];
        uint64_t rk0[NRC], rk1[NRC];
#pragma unroll
        for (int i = 0; i < NRC; i++) {
            rv0[i] = rwo[i] != NND_EMPTY_SLOT;
            rk0[i] = nnd_offer_key(rwo[i], salt);
            rv1[i] = rwn[i] != NND_EMPTY_SLOT;
            rk1[i] = nnd_offer_key(rwn[i], salt);
        }
        (void)nnd_select_class<LW, NRC>(act, valid && cls == 0u, fkey, rv0, rk0, mc, mcp, out + mcp, list, j, gb);
        my_rank = nnd_select_class<LW, NRC>(act, valid && cls == 1u, fkey, rv1, rk1, mc, mcp, out, list, j, gb);
    }
    // flag reset (utils.py:311-318): a forward new edge that was sampled becomes old
    if (act && valid && cls == 1u && my_rank < mc) knn_e[vv * ks + j] = u;
}

template <bool WIDE>
__global__ __launch_bounds__(256) void k_sample_select_h(uint32_t *__restrict__ knn_e, int64_t n, int k, int ks, int mc,
                                                         int mcp, uint32_t it_seed, uint32_t *__restrict__ rbuf,
                                                         int32_t *__restrict__ cand, int64_t own_lo, int64_t own_hi,
                                                         const uint8_t *__restrict__ active) {
    constexpr int RCAP = 32;
    __shared__ uint64_t skey[8][2][64];  // [half-wave of the workgroup][class][item] priority<<32 | id
    const int lane = nnd_lane(), w = threadIdx.x >> 6, h = lane >> 5, j = lane & 31, hb = lane & 32;
    const int64_t v = own_lo + ((int64_t)blockIdx.x * 4 + w) * 2 + h;
    const bool von = v < own_hi;
    const bool act = von && active[von ? v : own_lo] != 0;
    if (von && !act)  // no new candidate can reach v: empty new list, nothing else to do (no offers were stored for it)
        for (int q = j; q < mcp; q += 32) cand[v * 2 * mcp + q] = -1;
    if (!__ballot(act)) return;  // wave-uniform
    uint64_t(*sk)[64] = skey[w * 2 + h];
    const int64_t vv = act ? v : own_lo;

    uint32_t e = NND_EMPTY_E;
    if (act && j < k) e = knn_e[vv * ks + j];
    uint32_t rw0 = NND_EMPTY_SLOT, rw1 = NND_EMPTY_SLOT;  // reverse offers: old class, new class
    uint32_t *slots = rbuf + vv * 2 * RCAP;                // [class 0 | class 1] are adjacent: one 256-byte bank pair per vertex
    if (act) {
        rw0 = slots[j];
        rw1 = slots[RCAP + j];
        if (rw0 != NND_EMPTY_SLOT) slots[j] = NND_EMPTY_SLOT;  // re-arm for the next iteration
        if (rw1 != NND_EMPTY_SLOT) slots[RCAP + j] = NND_EMPTY_SLOT;
    }
    nnd_select_half<WIDE>(knn_e, k, ks, mc, mcp, it_seed, cand, vv, act, e, rw0, rw1, sk, j, hb);
}

// ------------------------------------------------------------------------------------------------
// Bucketed reverse pass (round 5).  k_sample_reverse above is 15 M random device-scope atomicMin per launch:
// on this chip EVERY global atomic is executed at the memory side (per-XCD L2s are not coherent; workgroup- and agent-scope
// atomics are the same instruction), 0.28 ms per launch whatever the traffic, and 32 hashed slots lose a fifth of the
// offers of a bank to collisions -- the reference's heaps lose an offer only by priority (utils.py:277-306), and that loss
// was the measured recall gap to the reference algorithm.  Here the reverse offers are TRANSPOSED
// instead of scattered one atomic at a time.  A BUCKET is 128 (256 in one regime) consecutive positions of the visiting
// order; every bucket owns 8 fixed-capacity SUB-REGIONS of records (word, target's index in the bucket | class) and their
// 8 cursors; what does not fit a sub-region (hubs) goes to ONE overflow list that only the buckets with a full sub-region
// read -- nothing is dropped, and the set of records a bucket sees is the set of offers made to its vertices:
//   k_rev_mark     (late iterations, shards) the endpoints of new edges are marked active first, so that the old-class offers
//                  nobody will read are dropped before they cost anything;
//   k_rev_place    one thread per edge: looks the target's POSITION in the visiting order up (the only random read per edge),
//                  forms the record, and places it: the offers of a 16-lane row that share a bucket are grouped by DPP
//                  rotations (a row's neighbours sit in a handful of buckets), the groups of a workgroup meet in an LDS hash
//                  table, ONE returning global atomic per (workgroup, bucket) reserves their run in the sub-region the
//                  workgroup's number picks, the records are written from registers -- the graph is walked ONCE (a first
//                  version counted, scanned and scattered: two walks + 128 MB of staged records per iteration);
//   k_rev_import   a shard's received offers, one atomic each (their targets are spread over all of the rank's buckets);
//   k_rev_select   (k <= 32) one workgroup per bucket: its targets' slot banks live in LDS, the records are APPENDED (an LDS
//                  counter per bank: no collision, no lost offer while a bank receives <= its slots; a bank that overflows is
//                  redone with hashed atomicMin slots -- order independent, hubs only), then the selection of k_sample_select_h
//                  runs on the banks where they are;  k_rev_fill: the other regimes -- the banks go to rbuf and
//                  k_sample_select* read them exactly as before (they never depended on WHICH slot an offer sits in).
// The result is a function of the graph and the seed alone: the set of appended words is the set of offers, the hashed
// fallback is a min.
#define RV_TAB 1024  // entries of the per-workgroup hash table (bucket -> count); what does not fit goes straight to global
#ifndef RV_RPT
#define RV_RPT 8     // row groups (of 256 / ksp rows) per workgroup = records per thread
#endif
#define RV_NOHASH 0xFFFFu
#ifndef RV_SEL_THREADS
#define RV_SEL_THREADS 512  // threads of the fused fill + select kernel (16 half-waves, 8 of the bucket's 128 targets each; 1024: 0.52 instead of 0.41 ms)
#endif

__device__ __forceinline__ uint32_t rv_tab_hash(uint32_t b) { return (b * 2654435761u) >> 22; }  // 10 bits

// slot of bucket b in the workgroup's table (claimed on first use), or -1 when 24 probes found no room
__device__ __forceinline__ int rv_tab_find(volatile uint32_t *hkey, uint32_t b) {
    uint32_t h = rv_tab_hash(b);
#pragma unroll 1
    for (int p = 0; p < 24; p++) {
        uint32_t cur = hkey[h];  // (a plain read first: after its first offer a bucket's slot is found without an atomic)
        if (cur == 0xFFFFFFFFu) cur = atomicCAS((uint32_t *)&hkey[h], 0xFFFFFFFFu, b);
        if (cur == 0xFFFFFFFFu || cur == b) return (int)h;
        h = (h + 1) & (RV_TAB - 1);
    }
    return -1;
}
#define RV_NOKEY 0xFFFFFFFFu
// Lanes of an aligned group of 16 (one k-list row when ks = 16) that hold the same key form a group: a row's neighbours
// sit in a handful of buckets, and an LDS atomic per LANE on those few addresses is serialised lane by lane -- 15 M of them
// made the first version of these kernels as slow as the global atomics they replace.  Fifteen DPP row rotations of
// (key, lane): rank = members below this lane, size, leader = lowest member.  Every lane of the wave must be active.
__device__ __forceinline__ void rv_row_groups(uint32_t key, int &rank, int &size, int &leader) {
    // (the lane of the WAVE: these kernels run (ksp, rows)-shaped workgroups, threadIdx.x is the slot of a row there)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    rank = 0;
    size = 1;
    leader = lane;
#define RV_ROT(r)                                                                         \
    {                                                                                     \
        const uint32_t ok = (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(r)>((int)key);           \
        const int ol = nnd_dpp_i32<NND_DPP_ROW_ROR(r)>(lane);                              \
        const bool eq = ok == key;                                                        \
        size += eq ? 1 : 0;                                                               \
        rank += (eq && ol < lane) ? 1 : 0;                                                \
        leader = (eq && ol < leader) ? ol : leader;                                       \
    }
    RV_ROT(1) RV_ROT(2) RV_ROT(3) RV_ROT(4) RV_ROT(5) RV_ROT(6) RV_ROT(7) RV_ROT(8)
    RV_ROT(9) RV_ROT(10) RV_ROT(11) RV_ROT(12) RV_ROT(13) RV_ROT(14) RV_ROT(15)
#undef RV_ROT
}
// word a bank stores for the offer v -> u: as k_sample_reverse (the slot word IS the source)
__device__ __forceinline__ uint32_t rv_offer_word(uint32_t it_seed, uint32_t u, uint32_t v) { return nnd_mix32(v ^ nnd_offer_salt(it_seed, u)); }

// late iterations (few new edges) and shards: the active flags first, so that k_rev_count can drop the offers nobody will read.
// Rows [row0, row0 + n_rows) are walked; only targets inside that range are marked (a shard's other targets travel as records).
__global__ __launch_bounds__(256) void k_rev_mark(const uint32_t *__restrict__ knn_e, int64_t row0, int64_t n_rows, int ks, uint8_t *__restrict__ active) {
    const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n_rows * ks) return;
    const u32x4 e = *(const u32x4 *)(knn_e + row0 * ks + i);  // (rows are ks = 16 * m words: a vector never straddles two)
    const int64_t v = row0 + i / ks;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        const uint32_t w = e[q];
        if (w != NND_EMPTY_E && (w >> 31)) {
            active[v] = 1;
            const uint32_t t = w & NND_IDX_MASK;
            if ((uint32_t)(t - (uint32_t)row0) < (uint32_t)n_rows) active[t] = 1;
        }
    }
}
// ... and the new-class offers a shard has received for its rows
__global__ void k_rev_mark_records(const int32_t *__restrict__ targets, int64_t count, int64_t row0, int64_t n_rows, uint8_t *__restrict__ active) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t t = (uint32_t)targets[i];
    const uint32_t u = t & NND_IDX_MASK;
    if ((t >> 31) && (uint32_t)(u - (uint32_t)row0) < (uint32_t)n_rows) active[u] = 1;
}

// The records of every bucket: 8 sub-regions of `cap` records + their cursors (a cursor may run past cap: the rest of its
// records is on the overflow list).
struct rv_inbox {
    const uint32_t *cursor = nullptr;  // (n_buckets, 8) records offered to every sub-region (beyond cap: on the overflow list)
    const uint2 *rec = nullptr;        // (n_buckets, 8, cap): (word, target's index in the bucket | class << 15) -- one 8-byte store per record
    int cap = 0;
    const uint2 *ov = nullptr;         // overflow list: (word, bucket << 9 | class << 8 | target's index in the bucket)
    const uint32_t *ov_count = nullptr;
};
// the local edges: see the header of this section.  Rows walked: order[0 .. n) (or row0 .. row0 + n); targets outside
// [row0, row0 + n) are not this handle's (a shard: they have travelled as records, k_offer_export); positions are relative to row0.
__global__ __launch_bounds__(256) void k_rev_place(const uint32_t *__restrict__ knn_e, int64_t row0, int64_t n, int k, int ks, uint32_t it_seed,
                                                   const int32_t *__restrict__ order, const int32_t *__restrict__ pos, int logB,
                                                   uint8_t *__restrict__ active, int mark, uint32_t *__restrict__ in_cursor,
                                                   uint2 *__restrict__ in_rec, int cap, uint2 *__restrict__ ov, uint32_t *__restrict__ ov_count) {
    __shared__ uint32_t hkey[RV_TAB], hcnt[RV_TAB], hbase[RV_TAB];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    for (int i = tid; i < RV_TAB; i += 256) { hkey[i] = 0xFFFFFFFFu; hcnt[i] = 0; }
    __syncthreads();
    const int j = threadIdx.x;
    const uint32_t sub = blockIdx.x & 7u, bmask = (1u << logB) - 1u;
    uint32_t ev[RV_RPT], vv[RV_RPT];
#pragma unroll
    for (int it = 0; it < RV_RPT; it++) {  // all loads of the workgroup's rows first
        const int64_t g = ((int64_t)blockIdx.x * RV_RPT + it) * blockDim.y + threadIdx.y;
        ev[it] = NND_EMPTY_E;
        vv[it] = 0;
        if (g < n && j < k) {
            vv[it] = (uint32_t)(order ? order[g] : (int32_t)(row0 + g));
            ev[it] = knn_e[(int64_t)vv[it] * ks + j];
        }
    }
    uint32_t bb[RV_RPT], w[RV_RPT], at[RV_RPT];  // bucket; slot word; rank inside the (workgroup, bucket) run, or the final place
    uint16_t m[RV_RPT], hh[RV_RPT];              // target's index in the bucket | class << 15; table slot, RV_NOHASH: `at` is final
#pragma unroll
    for (int it = 0; it < RV_RPT; it++) {
        bb[it] = RV_NOKEY;
        w[it] = at[it] = 0;
        m[it] = 0;
        hh[it] = RV_NOHASH;
        if (ev[it] != NND_EMPTY_E && (uint32_t)((ev[it] & NND_IDX_MASK) - (uint32_t)row0) >= (uint32_t)n) {
            if ((ev[it] >> 31) && mark == 1) active[vv[it]] = 1;
            ev[it] = NND_EMPTY_E;  // the target is another rank's
        }
        if (ev[it] != NND_EMPTY_E) {
            const uint32_t u = ev[it] & NND_IDX_MASK, cls = ev[it] >> 31;
            if (cls && mark == 1) { active[vv[it]] = 1; active[u] = 1; }  // a new edge: both endpoints will hold a new candidate
            // mark == 2 (k_rev_mark has run): an old-class offer to a vertex that will not join is dropped here, before it
            // costs a group, a record and a slot (utils.py:611-613: its old list is never read)
            if (mark == 2 && !cls && !active[u]) ev[it] = NND_EMPTY_E;
        }
        if (ev[it] != NND_EMPTY_E) {
            const uint32_t u = ev[it] & NND_IDX_MASK, cls = ev[it] >> 31;
            const uint32_t ul = u - (uint32_t)row0;
            const uint32_t p = (uint32_t)(pos ? pos[ul] : (int32_t)ul);
            bb[it] = p >> logB;
            w[it] = rv_offer_word(it_seed, u, vv[it]);
            m[it] = (uint16_t)((p & bmask) | (cls << 15));
        }
    }
#pragma unroll
    for (int it = 0; it < RV_RPT; it++) {
        int rank, size, leader;
        rv_row_groups(bb[it], rank, size, leader);
        int h = -1;
        uint32_t base = 0;
        if (bb[it] != RV_NOKEY && rank == 0) {  // one LDS (or, table full, global) atomic per group
            h = rv_tab_find(hkey, bb[it]);
            base = h >= 0 ? atomicAdd(&hcnt[h], (uint32_t)size) : atomicAdd(&in_cursor[(size_t)bb[it] * 8 + sub], (uint32_t)size);
        }
        h = __shfl(h, leader, 64);
        base = (uint32_t)__shfl((int)base, leader, 64);
        if (bb[it] != RV_NOKEY) {
            hh[it] = h >= 0 ? (uint16_t)h : RV_NOHASH;
            at[it] = base + (uint32_t)rank;
        }
    }
    __syncthreads();
    for (int i = tid; i < RV_TAB; i += 256)
        if (hcnt[i]) hbase[i] = atomicAdd(&in_cursor[(size_t)hkey[i] * 8 + sub], hcnt[i]);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < RV_RPT; it++) {
        if (bb[it] == RV_NOKEY) continue;
        const uint32_t at_ = hh[it] == RV_NOHASH ? at[it] : hbase[hh[it]] + at[it];
        if (at_ < (uint32_t)cap)
            in_rec[((size_t)bb[it] * 8 + sub) * cap + at_] = make_uint2(w[it], (uint32_t)m[it]);
        else
            ov[atomicAdd(ov_count, 1u)] = make_uint2(w[it], (bb[it] << 9) | (((uint32_t)m[it] >> 15) << 8) | ((uint32_t)m[it] & 0xFFu));
    }
}
// A shard's RECEIVED offers (target | class << 31, source; k_offer_export on the sender).  They arrive in the senders' row
// order: their targets are spread over ALL of this rank's buckets (rows are owned by number, 7 of 8 of a vertex's neighbours
// live on other ranks), so there is nothing for row groups or an LDS table to aggregate (measured: the same time with and
// without, 0.64 ms per 16 M records at 8 x 1.25 M rows, and as much again when they were counted first and placed second):
// one returning atomic and one 8-byte store per record; the thread number picks the sub-region.
__global__ __launch_bounds__(256) void k_rev_import(const int32_t *__restrict__ targets, const uint32_t *__restrict__ sources, int64_t count,
                                                    uint32_t it_seed, int64_t row0, int64_t n, const int32_t *__restrict__ pos, int logB,
                                                    const uint8_t *__restrict__ active, int filter, uint32_t *__restrict__ in_cursor,
                                                    uint2 *__restrict__ in_rec, int cap, uint2 *__restrict__ ov, uint32_t *__restrict__ ov_count) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint32_t t = (uint32_t)targets[i];
    const uint32_t cls = t >> 31, u = t & NND_IDX_MASK, ul = u - (uint32_t)row0;
    if (ul >= (uint32_t)n || (filter && !cls && !active[u])) return;  // (old-class offer to a vertex that will not join: never read)
    const uint32_t p = (uint32_t)(pos ? pos[ul] : (int32_t)ul);
    const uint32_t b = p >> logB, tl = p & ((1u << logB) - 1u);
    const uint32_t w = rv_offer_word(it_seed, u, sources[i]);
    const size_t sr = (size_t)b * 8 + (threadIdx.x & 7);
    const uint32_t at = atomicAdd(&in_cursor[sr], 1u);
    if (at < (uint32_t)cap) {
        in_rec[sr * cap + at] = make_uint2(w, tl | (cls << 15));
    } else {
        ov[atomicAdd(ov_count, 1u)] = make_uint2(w, (b << 9) | (cls << 8) | tl);
    }
}
// f(word, meta) for every record of bucket b.  The 8 cursors are read together and the sub-regions walked as ONE index space
// (eight loops, each behind its own cursor load, were eight dependent round trips per workgroup).
template <typename F>
__device__ __forceinline__ void rv_each_record(const rv_inbox &ib, int64_t b, int tid, int nthr, F f) {
    uint32_t c[8], pre[9];
    bool over = false;
#pragma unroll
    for (int sr = 0; sr < 8; sr++) c[sr] = ib.cursor[b * 8 + sr];
    pre[0] = 0;
#pragma unroll
    for (int sr = 0; sr < 8; sr++) {
        over = over || c[sr] > (uint32_t)ib.cap;
        pre[sr + 1] = pre[sr] + (c[sr] < (uint32_t)ib.cap ? c[sr] : (uint32_t)ib.cap);
    }
    const uint2 *base = ib.rec + (size_t)b * 8 * ib.cap;
    for (uint32_t i = tid; i < pre[8]; i += nthr) {
        int sr = 0;
#pragma unroll
        for (int q = 1; q < 8; q++) sr += i >= pre[q] ? 1 : 0;
        uint32_t lo = pre[0];
#pragma unroll
        for (int q = 1; q < 8; q++) lo = sr >= q ? pre[q] : lo;
        const uint2 r = base[(size_t)sr * ib.cap + (i - lo)];
        f(r.x, r.y);
    }
    if (over) {  // (workgroup-uniform) this bucket has records on the overflow list
        const uint32_t M = *ib.ov_count;
        for (uint32_t i = tid; i < M; i += nthr) {
            const uint2 r = ib.ov[i];
            if ((int64_t)(r.y >> 9) == b) f(r.x, (r.y & 0xFFu) | (((r.y >> 8) & 1u) << 15));
        }
    }
}

// slot of a word in an overflowing bank: any fixed function of the word (itself a mixed value) will do
__device__ __forceinline__ uint32_t rv_ovf_slot(uint32_t word, uint32_t cap) { return nnd_mix32(word ^ 0x68E31DA4u) & (cap - 1u); }

// One workgroup per bucket.  RCAP slots per (target, class); WIDE: the first pass of a build, every offer is new-class and a
// target's two banks form ONE bank of 2 * RCAP slots (nnd_offer_addr).  LDS: NB * 2 * RCAP words = 64 KB, two workgroups per CU.
template <int RCAP, int NB, bool WIDE>
__global__ __launch_bounds__(1024) void k_rev_fill(rv_inbox ib, int64_t row0, int64_t n, const int32_t *__restrict__ order,
                                                   const uint8_t *__restrict__ active, uint32_t *__restrict__ rbuf) {
    constexpr int ROW = 2 * RCAP;                 // words per target
    constexpr int CAP = WIDE ? 2 * RCAP : RCAP;   // slots per bank
    constexpr int NBANK = WIDE ? NB : 2 * NB;
    __shared__ uint32_t bank[NB * ROW];
    __shared__ uint32_t cnt[NBANK];
    __shared__ int any_ovf;
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    for (int i = tid; i < NB * ROW; i += 1024) bank[i] = NND_EMPTY_SLOT;
    for (int i = tid; i < NBANK; i += 1024) cnt[i] = 0;
    if (tid == 0) any_ovf = 0;
    __syncthreads();
    rv_each_record(ib, b, tid, 1024, [&](uint32_t w, uint32_t m) {
        const uint32_t bk = WIDE ? (m & 0x7FFFu) : ((m & 0x7FFFu) * 2 + (m >> 15));  // [old | new], as rbuf
        const uint32_t s = atomicAdd(&cnt[bk], 1u);
        if (s < (uint32_t)CAP) bank[bk * CAP + s] = w;
    });
    __syncthreads();
    for (int i = tid; i < NBANK; i += 1024)
        if (cnt[i] > (uint32_t)CAP) any_ovf = 1;
    __syncthreads();
    if (any_ovf) {  // (workgroup-uniform) a hub: more offers than slots -- those banks keep the smallest word per hashed slot
        for (int i = tid; i < NBANK * CAP; i += 1024)
            if (cnt[i / CAP] > (uint32_t)CAP) bank[i] = NND_EMPTY_SLOT;
        __syncthreads();
        rv_each_record(ib, b, tid, 1024, [&](uint32_t w, uint32_t m) {
            const uint32_t bk = WIDE ? (m & 0x7FFFu) : ((m & 0x7FFFu) * 2 + (m >> 15));
            if (cnt[bk] > (uint32_t)CAP) atomicMin(&bank[bk * CAP + rv_ovf_slot(w, CAP)], w);
        });
        __syncthreads();
    }
    // banks of the ACTIVE targets -> rbuf, whole rows (the selection reads no other; an inactive target's row stays EMPTY)
    const int tl0 = tid / ROW, c = tid % ROW;
    for (int tl = tl0; tl < NB; tl += 1024 / ROW) {
        const int64_t p = b * NB + tl;
        if (p >= n) break;
        const int64_t v = order ? (int64_t)order[p] : row0 + p;
        if (!active[v]) continue;
        rbuf[v * ROW + c] = bank[tl * ROW + c];
    }
}

// The BASELINE regime (k <= 32, 32 slots per bank, max_candidates <= 32): k_rev_fill and k_sample_select_h in ONE kernel --
// the banks never leave LDS (rbuf is not touched: 256 B written, read and re-armed per vertex and iteration otherwise).
// RV_SEL_THREADS threads per bucket of 128 targets: the records are appended to the banks as in k_rev_fill, then every half-wave
// runs the selection (nnd_select_half) for 8 of the bucket's targets in turn, the next one's k-list row requested while the
// current one is ranked.  LDS: 32 KB banks + 16 KB selection lists: three workgroups = 24 waves per CU (measured against 1024
// threads -- 4 targets per half-wave, 32 KB of lists, two workgroups = 32 waves per CU: 0.52 instead of 0.41 ms per launch).
template <bool WIDE, int LW>
__global__ __launch_bounds__(RV_SEL_THREADS) void k_rev_select(rv_inbox ib, int64_t row0, int64_t n, const int32_t *__restrict__ order,
                                                    const uint8_t *__restrict__ active,
                                                    uint32_t *__restrict__ knn_e, int k, int ks, int mc, int mcp, uint32_t it_seed,
                                                    int32_t *__restrict__ cand) {
    // LW lanes per target: 32 (k <= 32), or 16 -- four targets per wave -- when rows and candidate lists hold at most 16 entries
    constexpr int RCAP = 32, NB = 128, ROW = 2 * RCAP, NT = RV_SEL_THREADS, NG = NT / LW, PER = NB / NG, NRC = RCAP / LW;
    constexpr int CAP = WIDE ? 2 * RCAP : RCAP;
    constexpr int NBANK = WIDE ? NB : 2 * NB;
    __shared__ uint32_t bank[NB * ROW];
    __shared__ uint32_t cnt[NBANK];
    __shared__ uint64_t skey[NG][LW == 32 ? 128 : 3 * LW];  // LW = 32: the two 64-item class lists of nnd_select_half
    __shared__ int32_t vtx[NB];
    __shared__ int any_ovf;
    const int tid = threadIdx.x, lane = tid & 63, grp = tid / LW, j = lane & (LW - 1), gb = lane & (64 - LW);
    const int64_t b = blockIdx.x;
    // the bucket's vertices: v >= 0 active, -2 - v inactive, -1 beyond the last vertex
    if (tid < NB) {
        const int64_t p = b * NB + tid;
        int32_t v = -1;
        if (p < n) {
            v = order ? order[p] : (int32_t)(row0 + p);
            if (!active[v]) v = -2 - v;
        }
        vtx[tid] = v;
    }
    for (int i = tid; i < NB * ROW; i += NT) bank[i] = NND_EMPTY_SLOT;
    for (int i = tid; i < NBANK; i += NT) cnt[i] = 0;
    if (tid == 0) any_ovf = 0;
    __syncthreads();
    // the k-list row of this group's first target is requested now and lands during the fill phase; the next target's
    // row is requested while the current one is ranked
    uint32_t pre_e = NND_EMPTY_E;
    {
        const int32_t v0 = vtx[grp];
        if (v0 >= 0 && j < k) pre_e = knn_e[(int64_t)v0 * ks + j];
    }
    rv_each_record(ib, b, tid, NT, [&](uint32_t w, uint32_t m) {
        const uint32_t bk = WIDE ? (m & 0x7FFFu) : ((m & 0x7FFFu) * 2 + (m >> 15));  // [old | new], as rbuf
        const uint32_t sl = atomicAdd(&cnt[bk], 1u);
        if (sl < (uint32_t)CAP) bank[bk * CAP + sl] = w;
    });
    __syncthreads();
    for (int i = tid; i < NBANK; i += NT)
        if (cnt[i] > (uint32_t)CAP) any_ovf = 1;
    __syncthreads();
    if (any_ovf) {  // (workgroup-uniform) hubs: see k_rev_fill
        for (int i = tid; i < NBANK * CAP; i += NT)
            if (cnt[i / CAP] > (uint32_t)CAP) bank[i] = NND_EMPTY_SLOT;
        __syncthreads();
        rv_each_record(ib, b, tid, NT, [&](uint32_t w, uint32_t m) {
            const uint32_t bk = WIDE ? (m & 0x7FFFu) : ((m & 0x7FFFu) * 2 + (m >> 15));
            if (cnt[bk] > (uint32_t)CAP) atomicMin(&bank[bk * CAP + rv_ovf_slot(w, CAP)], w);
        });
        __syncthreads();
    }
    uint64_t *sk = skey[grp];
#pragma unroll 1
    for (int r = 0; r < PER; r++) {
        const int tl = r * NG + grp;
        const int32_t v = vtx[tl];
        const bool act = v >= 0;
        const uint32_t e = pre_e;
        pre_e = NND_EMPTY_E;
        if (r + 1 < PER) {
            const int32_t vn = vtx[tl + NG];
            if (vn >= 0 && j < k) pre_e = knn_e[(int64_t)vn * ks + j];
        }
        if (v <= -2)  // no new candidate can reach the vertex: empty new list (its old list is never read)
            for (int q = j; q < mcp; q += LW) cand[(int64_t)(-2 - v) * 2 * mcp + q] = -1;
        if (!__ballot(act)) continue;  // wave-uniform
        uint32_t rwo[NRC], rwn[NRC];
#pragma unroll
        for (int i = 0; i < NRC; i++) {
            rwo[i] = act ? bank[tl * ROW + i * LW + j] : NND_EMPTY_SLOT;
            rwn[i] = act ? bank[tl * ROW + RCAP + i * LW + j] : NND_EMPTY_SLOT;
        }
        nnd_wave_lds_sync();  // the previous target's lists are done with
#ifdef NND_RV_NOSELECT  // timing experiments only: the kernel without the per-target selection
        if (e == 0x12345u && rwo[0] == 7u && rwn[0] == 9u) cand[0] = 1;
        continue;
#endif
        if constexpr (LW == 32)
            nnd_select_half<WIDE>(knn_e, k, ks, mc, mcp, it_seed, cand, act ? (int64_t)v : 0, act, e, rwo[0], rwn[0], (uint64_t(*)[64])sk, j, gb);
        else
            nnd_select_group<WIDE, LW, NRC>(knn_e, k, ks, mc, mcp, it_seed, cand, act ? (int64_t)v : 0, act, e, rwo, rwn, sk, j, gb);
    }
}

__global__ void k_rev_invert(const int32_t *__restrict__ order, int64_t row0, int64_t n, int32_t *__restrict__ pos) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) pos[order[g] - row0] = (int32_t)g;
}

// rows scanned for reverse offers: every row on a plain handle; the owned slice when shard bounds are set (row-sharded
// build: offers to targets owned elsewhere travel as records, see k_offer_export)
// both banks of a vertex for the new class while no old edge exists (nnd_offer_addr): 32 slots per class (the
// max_candidates <= 32 regime) and k <= 64, so that k forward + 64 reverse items fit the selection's 128-item lists
static bool sample_wide(const nnd_ctx *ctx) { return ctx->all_new && ctx->rcap == 32 && ctx->k <= 64; }

static void launch_reverse_pass(nnd_ctx *ctx, int pass, uint32_t it_seed) {
    const bool shard = ctx->n_ranks > 1;
    const int64_t row_lo = shard ? ctx->own_lo : 0, row_hi = shard ? ctx->own_hi : ctx->n;
    int ksp = 16;
    while (ksp < ctx->ks) ksp <<= 1;
    const int rows = 256 / ksp;
    unsigned grid = (unsigned)((row_hi - row_lo + rows - 1) / rows);
    grid = (grid + 7u) & ~7u;  // whole multiples of the XCD count
    // the spatial order is a permutation of ALL rows: usable whenever there is a forest and all rows are scanned
    const int32_t *order = (!shard && ctx->forest_built && ctx->p.n_trees > 0) ? ctx->perm[ctx->cur] : nullptr;
    ctx->rbuf_clean = false;
    hipLaunchKernelGGL(k_sample_reverse, dim3(grid), dim3(ksp, rows), 0, ctx->stream, ctx->knn_e, row_lo, row_hi, ctx->k, ctx->ks,
                       it_seed, ctx->rbuf, ctx->rcap, ctx->own_lo, ctx->own_hi, pass, ctx->active, order, sample_wide(ctx) ? 1 : 0);
}

static uint32_t sample_seed(const nnd_ctx *ctx) { return nnd_hash2(ctx->seed ^ 0x9E3779B9u, (uint32_t)ctx->iter + 1u); }

static void launch_select(nnd_ctx *ctx, uint32_t it_seed, bool wide) {
    if (ctx->k > 64 || ctx->mc > 64) {  // rows, or candidate lists, of more than one entry per lane
        hipLaunchKernelGGL(k_sample_select_wide, dim3((unsigned)((ctx->own_hi - ctx->own_lo + 3) / 4)), dim3(256), 0, ctx->stream,
                           ctx->knn_e, ctx->n, ctx->k, ctx->ks, ctx->mc, ctx->mcp, it_seed, ctx->rbuf, ctx->rcap, ctx->cand, ctx->own_lo,
                           ctx->own_hi, ctx->active);
        if (ctx->n_ranks <= 1) ctx->rbuf_clean = true;
        return;
    }
    const bool force_old = (ctx->p.flags & NND_FLAG_TEST_SELECT_WAVE) != 0;  // parity test: the one-wave-per-vertex kernel
    if (ctx->k <= 32 && ctx->rcap == 32 && ctx->mc <= 32 && !force_old) {
        auto kern = wide ? k_sample_select_h<true> : k_sample_select_h<false>;
        hipLaunchKernelGGL(kern, dim3((unsigned)((ctx->own_hi - ctx->own_lo + 7) / 8)), dim3(256), 0, ctx->stream,
                           ctx->knn_e, ctx->n, ctx->k, ctx->ks, ctx->mc, ctx->mcp, it_seed, ctx->rbuf, ctx->cand, ctx->own_lo,
                           ctx->own_hi, ctx->active);
        if (ctx->n_ranks <= 1) ctx->rbuf_clean = true;
        return;
    }
    hipLaunchKernelGGL(k_sample_select, dim3((unsigned)((ctx->own_hi - ctx->own_lo + 3) / 4)), dim3(256), 0, ctx->stream,
                       ctx->knn_e, ctx->n, ctx->k, ctx->ks, ctx->mc, ctx->mcp, it_seed, ctx->rbuf, ctx->rcap, ctx->cand, ctx->own_lo,
                       ctx->own_hi, ctx->active, wide ? 1 : 0);
    if (ctx->n_ranks <= 1) ctx->rbuf_clean = true;  // every bank that received an offer belongs to an active vertex and was re-armed
}

// grow-only tables of the bucketed reverse pass; the inverse of the visiting order once per forest.  An allocation that fails is
// NOT a build error (returns false, nothing left behind): the caller falls back to the hashed slots, which need 8 * rcap bytes a row
template <typename T>
static bool rv_grow(T **p, size_t count) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    if (hipMalloc((void **)p, sizeof(T) * count) != hipSuccess) {
        (void)hipGetLastError();  // (the error is handled here: do not let a later hipGetLastError() report it)
        *p = nullptr;
        return false;
    }
    return true;
}
// rows [row0, row0 + n_rows) are walked (all rows; a shard: the owned slice), `extra` records arrive from elsewhere.  The
// record regions: nb buckets x 8 sub-regions x cap records of 8 bytes + the overflow list, sized for every offer.  cap = 4 x the
// mean load of a sub-region when every offer of a bucket's vertices is counted (1024 at k = 15: 0.5 GB per million rows) for
// k <= 32 -- the regime that is tuned: a sub-region that overflows sends its bucket to the overflow list -- and 2 x beyond (wide
// rows: the regions would be 4 .. 16 GB per million rows at k = 64 .. 256 with the factor 4 and a power-of-two capacity; round-5
// advisor item).  Returns 0, 1 (error) or 2: the tables could not be allocated -- the caller samples through the hashed slots.
static int rv_prepare(nnd_ctx *ctx, int logB, const int32_t *order, int64_t row0, int64_t n_rows, int64_t extra, int *cap_out) {
    const int64_t nb = ((n_rows - 1) >> logB) + 1, nov = n_rows * ctx->k + extra;
    const int64_t mean = n_rows * ctx->k / (nb * 8) + 1;
    int64_t cap64 = (ctx->k <= 32 ? 4 : 2) * mean;
    cap64 = cap64 < 64 ? 64 : ((cap64 + 63) / 64) * 64;
    const int cap = (int)cap64;
    *cap_out = cap;
    const int64_t need = nb * 8 * cap;
    bool ok = !(ctx->p.flags & NND_FLAG_TEST_SAMPLE_NOMEM);  // test hook: behave as if the first allocation had failed
    if (ok && (need > ctx->rv_cap_in || cap != ctx->rv_in_cap)) {
        ctx->rv_cap_in = 0;  // (a failed allocation leaves no stale capacity behind)
        ok = rv_grow(&ctx->rv_in_cursor, (size_t)nb * 8 + 8) && rv_grow(&ctx->rv_in_rec, (size_t)need);
        if (ok) {
            ctx->rv_cap_in = need;
            ctx->rv_in_cap = cap;
        }
    }
    if (ok && nov > ctx->rv_cap_ov) {
        const int64_t c = extra > 0 ? nov + nov / 4 : nov;  // (a shard's inbox varies from iteration to iteration: head room)
        ctx->rv_cap_ov = 0;
        ok = rv_grow(&ctx->rv_ov, (size_t)c);
        if (ok) ctx->rv_cap_ov = c;
    }
    if (ok && order && (!ctx->rv_pos || ctx->rv_pos_gen != ctx->forest_gen || ctx->rv_pos_of != order)) {
        if (n_rows > ctx->rv_cap_pos) {
            ctx->rv_cap_pos = 0;
            ok = rv_grow(&ctx->rv_pos, (size_t)n_rows);
            if (ok) ctx->rv_cap_pos = n_rows;
        }
        if (ok) {
            hipLaunchKernelGGL(k_rev_invert, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, ctx->stream, order, row0, n_rows, ctx->rv_pos);
            ctx->rv_pos_gen = ctx->forest_gen;
            ctx->rv_pos_of = order;
        }
    }
    if (!ok) {
        // give the memory back and switch this handle to the hashed-slot form of the pass for good: the banks (rbuf) exist on
        // every handle; they are re-armed here because the bucketed fill (k_rev_fill) leaves them in its own state
        if (ctx->rv_in_cursor) { (void)hipFree(ctx->rv_in_cursor); ctx->rv_in_cursor = nullptr; }
        if (ctx->rv_in_rec) { (void)hipFree(ctx->rv_in_rec); ctx->rv_in_rec = nullptr; }
        if (ctx->rv_ov) { (void)hipFree(ctx->rv_ov); ctx->rv_ov = nullptr; }
        ctx->rv_cap_in = ctx->rv_cap_ov = 0;
        ctx->rv_in_cap = 0;
        ctx->rv_off = true;
        NND_HIP_CHECK(hipMemsetAsync(ctx->rbuf + (size_t)ctx->slim_row0() * 2 * ctx->rcap, 0xFF, sizeof(uint32_t) * (size_t)ctx->slim_rows() * 2 * ctx->rcap, ctx->stream));
        ctx->rbuf_clean = true;
        return 2;
    }
    return 0;
}

// the BASELINE regime: the fused kernel (k_rev_select) -- the conditions of k_sample_select_h
static bool rv_fused(const nnd_ctx *ctx) {
    return ctx->k <= 32 && ctx->rcap == 32 && ctx->mc <= 32 && !(ctx->p.flags & NND_FLAG_TEST_SELECT_WAVE);
}
static bool rv_bucketed(const nnd_ctx *ctx) { return (ctx->rcap == 32 || ctx->rcap == 64) && !ctx->rv_off && !(ctx->p.flags & NND_FLAG_TEST_SAMPLE_ATOMIC); }
// The reverse offers of this iteration by transposition (see above), then the selection; active[] is set on the way.  A shard
// (n_ranks > 1) walks its owned rows and adds the `n_in` offers it has received (in_targets: target | class << 31, in_sources).
static int launch_sample_bucketed(nnd_ctx *ctx, uint32_t it_seed, bool wide, const int32_t *in_targets, const uint32_t *in_sources, int64_t n_in) {
    const bool fused = rv_fused(ctx), shard = ctx->n_ranks > 1;
    const int logB = (fused || ctx->rcap != 32) ? 7 : 8;
    const int64_t row0 = shard ? ctx->own_lo : 0, n_rows = shard ? ctx->own_hi - ctx->own_lo : ctx->n;
    if (n_rows <= 0) return 0;
    const int32_t *order = shard ? ctx->own_order : ((ctx->forest_built && ctx->p.n_trees > 0) ? ctx->perm[ctx->cur] : nullptr);
    int cap = 0;
    if (const int prc = rv_prepare(ctx, logB, order, row0, n_rows, n_in, &cap)) return prc;  // 2: no memory for the regions (rv_off is set)
    const int32_t *pos = order ? ctx->rv_pos : nullptr;
    const int64_t nb = ((n_rows - 1) >> logB) + 1;
    if (nb >= ((int64_t)1 << 23)) { ctx->set_error("candidate sampling: %lld buckets exceed the 2^23 bucket ids of an overflow record", (long long)nb); return 1; }
    int ksp = 16;
    while (ksp < ctx->ks) ksp <<= 1;
    const int rows = 256 / ksp;
    const unsigned grid = (unsigned)((n_rows + (int64_t)rows * RV_RPT - 1) / ((int64_t)rows * RV_RPT));
    hipStream_t st = ctx->stream;
    uint32_t *ov_count = ctx->rv_in_cursor + (size_t)nb * 8;
    NND_HIP_CHECK(hipMemsetAsync(ctx->rv_in_cursor, 0, sizeof(uint32_t) * ((size_t)nb * 8 + 8), st));  // (+ the overflow count behind the cursors)
    // the first pass of a build: every edge is new, every vertex with an edge is active -- one memset instead of n * k random byte stores
    // (a vertex without any edge then "joins" with an empty new list, which is what an inactive one does)
    NND_HIP_CHECK(hipMemsetAsync(ctx->active + row0, ctx->all_new ? 1 : 0, (size_t)n_rows, st));
    // how the active flags come about: 0 = the memset above, 1 = k_rev_place marks them while it walks, 2 = k_rev_mark first and
    // k_rev_place filters by them -- once the previous iteration inserted into fewer than an eighth of the slots, and always on
    // a shard (the received new-class offers mark too, before anything is filtered)
    int mark = ctx->all_new ? 0 : 1;
    if (mark == 1 && (shard || (ctx->last_updates >= 0 && ctx->last_updates * 8 < ctx->n * ctx->k))) {
        hipLaunchKernelGGL(k_rev_mark, dim3((unsigned)((n_rows * ctx->ks / 4 + 255) / 256)), dim3(256), 0, st, ctx->knn_e, row0, n_rows, ctx->ks, ctx->active);
        if (n_in > 0)
            hipLaunchKernelGGL(k_rev_mark_records, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, in_targets, n_in, row0, n_rows, ctx->active);
        mark = 2;
    }
    hipLaunchKernelGGL(k_rev_place, dim3(grid), dim3(ksp, rows), 0, st, ctx->knn_e, row0, n_rows, ctx->k, ctx->ks, it_seed, order, pos, logB, ctx->active,
                       mark, ctx->rv_in_cursor, ctx->rv_in_rec, cap, ctx->rv_ov, ov_count);
    if (n_in > 0)
        hipLaunchKernelGGL(k_rev_import, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, in_targets, in_sources, n_in, it_seed, row0, n_rows, pos, logB,
                           ctx->active, mark == 2 ? 1 : 0, ctx->rv_in_cursor, ctx->rv_in_rec, cap, ctx->rv_ov, ov_count);
    rv_inbox ib;
    ib.cursor = ctx->rv_in_cursor;
    ib.rec = ctx->rv_in_rec;
    ib.cap = cap;
    ib.ov = ctx->rv_ov;
    ib.ov_count = ov_count;
    if (fused) {
        const bool q16 = ctx->ks == 16 && ctx->mc <= 16 && !(ctx->p.flags & NND_FLAG_TEST_SELECT_HALF);  // four targets per wave
        auto kern = q16 ? (wide ? k_rev_select<true, 16> : k_rev_select<false, 16>) : (wide ? k_rev_select<true, 32> : k_rev_select<false, 32>);
        hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(RV_SEL_THREADS), 0, st, ib, row0, n_rows, order, ctx->active, ctx->knn_e, ctx->k, ctx->ks, ctx->mc, ctx->mcp,
                           it_seed, ctx->cand);
        return 0;
    }
    ctx->rbuf_clean = false;
    auto fill = ctx->rcap == 32 ? (wide ? k_rev_fill<32, 256, true> : k_rev_fill<32, 256, false>) : k_rev_fill<64, 128, false>;
    hipLaunchKernelGGL(fill, dim3((unsigned)nb), dim3(1024), 0, st, ib, row0, n_rows, order, ctx->active, ctx->rbuf);
    launch_select(ctx, it_seed, wide);
    return 0;
}

int nnd_launch_sample(nnd_ctx *ctx) {
    if (ctx->n_ranks > 1) {
        ctx->set_error("nnd_launch_sample: this handle is one shard of a row-sharded build; use nnd_sample_begin / nnd_sample_finish");
        return 1;
    }
    const uint32_t it_seed = sample_seed(ctx);
    if (rv_bucketed(ctx)) {
        const bool wide_b = sample_wide(ctx);
        const int brc = launch_sample_bucketed(ctx, it_seed, wide_b, nullptr, nullptr, 0);
        if (brc == 1) return 1;
        if (brc == 0) {
            ctx->all_new = false;
            NND_HIP_CHECK(hipGetLastError());
            return 0;
        }
        // brc == 2: the record regions do not fit the device -- this and every later pass of the handle: hashed slots (below)
    }
    NND_HIP_CHECK(hipMemsetAsync(ctx->active, 0, (size_t)ctx->n, ctx->stream));
    // every edge still carries the "new" flag before the first sampling pass: there are no old edges to offer
    const int n_pass = ctx->all_new ? 1 : 2;
    const bool wide = sample_wide(ctx);
    for (int pass = 0; pass < n_pass; pass++) launch_reverse_pass(ctx, pass, it_seed);
    ctx->all_new = false;
    launch_select(ctx, it_seed, wide);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
// Row-sharded build (exchange X2): a reverse offer (v -> u) whose target u is owned by another
// rank is not applied here -- it becomes an 8-byte record (u | class << 31, v) in the region of
// u's owner, the host ships the regions (all-to-all-v over RCCL), and the owner folds what it receives into its slot
// banks with the same atomicMin as a local offer.  This is the cross-process form of the ownership test of
// new_build_candidates (utils.py:266-273): every rank scans only ITS rows instead of all n * k edges.
// Thread layout of k_sample_reverse (x = slot, y = row), OFFER_IT row groups per workgroup: a thread keeps its OFFER_IT
// edges in registers, records are counted per destination inside the workgroup (one LDS atomic per wave, pass and
// destination), ONE global atomic per workgroup and destination reserves their slots in the destination's region -- the
// cursors are <= 64 hot addresses: a reservation per 16 rows (the first version) spent the launch serialising on them
// (0.95 ms per launch at 1.25 M owned rows) -- and the records are written from the registers.
#define OFFER_IT 16
__global__ __launch_bounds__(256) void k_offer_export(const uint32_t *__restrict__ knn_e, int64_t row_lo, int64_t row_hi, int k, int ks,
                                                      uint32_t it_seed, const int64_t *__restrict__ bounds, int n_ranks,
                                                      int64_t own_lo, int64_t own_hi, int64_t cap, long long *__restrict__ cursors,
                                                      int32_t *__restrict__ targets, uint32_t *__restrict__ sources,
                                                      long long *__restrict__ dropped) {
    __shared__ int cnt[64];
    __shared__ long long base[64];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if (tid < 64) cnt[tid] = 0;
    __syncthreads();
    const int j = threadIdx.x;
    uint32_t ev[OFFER_IT];
    int dest[OFFER_IT], my[OFFER_IT];
#pragma unroll
    for (int it = 0; it < OFFER_IT; it++) {
        const int64_t g = row_lo + ((int64_t)blockIdx.x * OFFER_IT + it) * blockDim.y + threadIdx.y;
        ev[it] = NND_EMPTY_E;
        if (g < row_hi && j < k) ev[it] = knn_e[g * ks + j];
    }
#pragma unroll
    for (int it = 0; it < OFFER_IT; it++) {
        dest[it] = -1;
        my[it] = 0;
        if (ev[it] != NND_EMPTY_E) {
            const int64_t u = (int64_t)(ev[it] & NND_IDX_MASK);
            if (u < own_lo || u >= own_hi) dest[it] = nnd_owner_of(bounds, n_ranks, u);
        }
        unsigned long long todo = __ballot(dest[it] >= 0);
        while (todo) {  // wave-uniform: one LDS atomic per destination present in the wave
            const int lead = __builtin_ctzll(todo);
            const int dd = __builtin_amdgcn_readlane(dest[it], lead);
            const unsigned long long md = __ballot(dest[it] == dd);
            int b = 0;
            if ((tid & 63) == lead) b = atomicAdd(&cnt[dd], __popcll(md));
            b = __builtin_amdgcn_readlane(b, lead);
            if (dest[it] == dd) my[it] = b + nnd_prefix_popc(md);
            todo &= ~md;
        }
    }
    __syncthreads();
    if (tid < n_ranks && cnt[tid] > 0) base[tid] = (long long)atomicAdd((unsigned long long *)&cursors[tid], (unsigned long long)cnt[tid]);
    __syncthreads();
    int n_drop = 0;
#pragma unroll
    for (int it = 0; it < OFFER_IT; it++) {
        if (dest[it] < 0) continue;
        const int64_t g = row_lo + ((int64_t)blockIdx.x * OFFER_IT + it) * blockDim.y + threadIdx.y;
        const uint32_t u = ev[it] & NND_IDX_MASK, cls = ev[it] >> 31;
        const long long at = base[dest[it]] + my[it];
        if (at < cap) {
            const int64_t idx = (int64_t)dest[it] * cap + at;
            targets[idx] = (int32_t)(u | (cls << 31));
            sources[idx] = (uint32_t)g;
        } else {
            n_drop++;  // cannot happen with cap = owned rows * k
        }
    }
    if (n_drop) atomicAdd((unsigned long long *)dropped, (unsigned long long)n_drop);
}

__global__ void k_offer_import(const int32_t *__restrict__ targets, const uint32_t *__restrict__ sources, int64_t count, uint32_t want_cls,
                               uint32_t it_seed, uint32_t *__restrict__ rbuf, int rcap, uint8_t *__restrict__ active,
                               int64_t own_lo, int64_t own_hi, int wide) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t t = (uint32_t)targets[i];
    const uint32_t cls = t >> 31, u = t & NND_IDX_MASK;
    if (cls != want_cls || (int64_t)u < own_lo || (int64_t)u >= own_hi) return;
    if (cls == 1u) active[u] = 1;
    else if (!active[u]) return;  // no new candidate reaches u: its old list is never read
    const uint32_t v = sources[i];  // the priority is a function of (source, target): it does not travel
    const uint32_t salt = nnd_offer_salt(it_seed, u);
    atomicMin(&rbuf[nnd_offer_addr(u, cls, salt, v, rcap, wide)], nnd_mix32(v ^ salt));
}

// first half of a sharded sampling pass: local new edges, and the records for targets owned elsewhere (both classes)
int nnd_launch_sample_begin(nnd_ctx *ctx, int64_t cap, int32_t *targets_dev, uint32_t *sources_dev, long long *counts_dev) {
    if (ctx->n_ranks < 1 || !ctx->shard_bounds) { ctx->set_error("nnd_sample_begin: call nnd_set_shard_bounds first"); return 1; }
    const uint32_t it_seed = sample_seed(ctx);
    NND_HIP_CHECK(hipMemsetAsync(ctx->shard_cursors, 0, sizeof(long long) * 66, ctx->stream));
    if (!rv_bucketed(ctx)) {  // (the bucketed pass handles local and received offers together, in nnd_launch_sample_finish)
        NND_HIP_CHECK(hipMemsetAsync(ctx->active + ctx->slim_row0(), 0, (size_t)ctx->slim_rows(), ctx->stream));
        launch_reverse_pass(ctx, 0, it_seed);
    }
    if (ctx->n_ranks > 1) {
        int ksp = 16;
        while (ksp < ctx->ks) ksp <<= 1;
        const int rows = 256 / ksp;
        const unsigned grid = (unsigned)((ctx->own_hi - ctx->own_lo + (int64_t)rows * OFFER_IT - 1) / ((int64_t)rows * OFFER_IT));
        if (grid > 0)
            hipLaunchKernelGGL(k_offer_export, dim3(grid), dim3(ksp, rows), 0, ctx->stream, ctx->knn_e, ctx->own_lo, ctx->own_hi, ctx->k,
                               ctx->ks, it_seed, ctx->shard_bounds, ctx->n_ranks, ctx->own_lo, ctx->own_hi, cap, ctx->shard_cursors,
                               targets_dev, sources_dev, ctx->shard_cursors + 64);
    }
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(counts_dev, ctx->shard_cursors, sizeof(long long) * (size_t)ctx->n_ranks, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// second half: received records (new class first: they decide which vertices are active), local old edges, received
// old-class records, then the per-vertex selection
int nnd_launch_sample_finish(nnd_ctx *ctx, const int32_t *targets_dev, const uint32_t *sources_dev, int64_t count) {
    const uint32_t it_seed = sample_seed(ctx);
    const unsigned grid = (unsigned)((count + 255) / 256);
    const bool wide = sample_wide(ctx);  // (all_new is still what it was when nnd_launch_sample_begin ran)
    if (rv_bucketed(ctx)) {  // round 5: local edges and received records are transposed together (no hashed slots, no lost offers)
        const int brc = launch_sample_bucketed(ctx, it_seed, wide, targets_dev, sources_dev, count);
        if (brc == 1) return 1;
        if (brc == 0) {
            ctx->all_new = false;
            NND_HIP_CHECK(hipGetLastError());
            return 0;
        }
        // brc == 2: no memory for the record regions.  nnd_launch_sample_begin left the local new-class pass to the bucketed form:
        // run it now, then go on with the hashed slots (every rank decides for itself: the exchange does not depend on the form)
        NND_HIP_CHECK(hipMemsetAsync(ctx->active + ctx->slim_row0(), 0, (size_t)ctx->slim_rows(), ctx->stream));
        launch_reverse_pass(ctx, 0, it_seed);
    }
    if (count > 0) {
        ctx->rbuf_clean = false;
        hipLaunchKernelGGL(k_offer_import, dim3(grid), dim3(256), 0, ctx->stream, targets_dev, sources_dev, count, 1u, it_seed, ctx->rbuf,
                           ctx->rcap, ctx->active, ctx->own_lo, ctx->own_hi, wide ? 1 : 0);
    }
    if (!ctx->all_new) {
        launch_reverse_pass(ctx, 1, it_seed);
        if (count > 0)
            hipLaunchKernelGGL(k_offer_import, dim3(grid), dim3(256), 0, ctx->stream, targets_dev, sources_dev, count, 0u, it_seed, ctx->rbuf,
                               ctx->rcap, ctx->active, ctx->own_lo, ctx->own_hi, 0);
    }
    ctx->all_new = false;
    launch_select(ctx, it_seed, wide);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
