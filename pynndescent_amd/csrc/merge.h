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
//
// Everything lives in registers: lane j < k holds list entry j, lane l holds candidate(s) l, l+64, ...
// and values are broadcast with v_readlane (the broadcast lane is wave-uniform), so the inner loops
// run at a few cycles per step instead of an LDS round trip per step.
#pragma once
#include "common.h"

#define NND_MAX_K 64

__device__ __forceinline__ uint64_t nnd_readlane_u64(uint64_t v, int src_lane) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src_lane);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}

// All 64 lanes of the wave call this with the same arguments.
// cand(c, id, d) -> bool : candidate c of [0, ncand), ncand <= 64 * NCHUNK (ids unique inside the batch).
// Returns (same value on every lane) the number of candidates that beat the row's worst distance as it was when the
// merge started and were not in the row yet -- the pushes that succeed on the snapshot.  This is what feeds the stop
// rule c <= delta*k*n: the reference counts every successful push of its SEQUENTIAL apply (utils.py:717-729), also
// those evicted later in the same pass, so the net number of new entries would under-count and stop descents that
// converge slowly one or two iterations before the reference does.
// Core: lane j < k already holds list entry j in (e, d) (lanes >= k hold EMPTY / +inf; filled entries are packed
// at the front of the row).
//
// The work is VALU instruction count (this routine is what bounds the leaf-seeding kernel), so every loop is sized
// by what is actually there: the number of filled list entries, the number of candidates that survive the threshold;
// and a batch much larger than k (first tree: every leaf-mate beats an empty row) is cut down to ~k + a few by a
// sampled EXACT bound before the rank pass: a candidate with >= k keys at or below it among (batch U list) is an
// upper bound of the final worst key, so everything above it can be dropped without changing the result.
template <int NCHUNK, typename CandFn>
__device__ __forceinline__ int nnd_merge_row_regs(uint32_t *__restrict__ row_e, float *__restrict__ row_d,
                                                  float *__restrict__ th_slot, uint32_t e, float d, int k, int ncand,
                                                  CandFn cand) {
    const int lane = nnd_lane();
    const uint64_t mykey = (e == NND_EMPTY_E) ? NND_EMPTY_KEY : nnd_make_key(d, e);
    const uint32_t myidx = e & NND_IDX_MASK;  // 0x7FFFFFFF for empty slots: never a valid id
    const float th = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(d), k - 1));  // worst distance (+inf while not full)
    const int nlist = __popcll(__ballot(e != NND_EMPTY_E));

    uint32_t cid[NCHUNK];
    float cd[NCHUNK];
    unsigned long long cmask[NCHUNK];
    int nv = 0;
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) {
        const int c = ch * 64 + lane;
        uint32_t id = 0;
        float dc = 0.0f;
        bool ok = false;
        if (ch * 64 < ncand) {  // wave-uniform
            ok = (c < ncand) && cand(c, id, dc);
            ok = ok && (dc < th);  // strict, utils.py:484
        }
        cid[ch] = id;
        cd[ch] = dc;
        cmask[ch] = __ballot(ok);
        nv += __popcll(cmask[ch]);
    }
    if (nv == 0) return 0;

    // utils.py:489-492: drop candidates already in the row -- walk whichever side is shorter
    if (nlist > 0) {
        if (nv < nlist) {
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ch++) {
                unsigned long long m = cmask[ch];
                while (m) {
                    const int src = __builtin_ctzll(m);
                    m &= m - 1;
                    const uint32_t ids = (uint32_t)__builtin_amdgcn_readlane((int)cid[ch], src);
                    if (__ballot(myidx == ids)) cmask[ch] &= ~(1ull << src);
                }
            }
        } else {
            bool ok[NCHUNK];
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ch++) ok[ch] = (cmask[ch] >> lane) & 1ull;
            for (int j = 0; j < nlist; j++) {
                const uint32_t idj = (uint32_t)__builtin_amdgcn_readlane((int)myidx, j);
#pragma unroll
                for (int ch = 0; ch < NCHUNK; ch++) ok[ch] = ok[ch] && (cid[ch] != idj);
            }
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ch++) cmask[ch] = __ballot(ok[ch]);
        }
        nv = 0;
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ch++) nv += __popcll(cmask[ch]);
        if (nv == 0) return 0;
    }
    const int pushed = nv;

    uint64_t ckey[NCHUNK];
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) ckey[ch] = ((cmask[ch] >> lane) & 1ull) ? nnd_make_key(cd[ch], cid[ch]) : NND_EMPTY_KEY;

    // batch much larger than k: tighten with an exact sampled bound (see above)
    if (nv > k + 12) {
        uint64_t bound = NND_EMPTY_KEY;
        unsigned long long m = cmask[0];
        for (int t = 0; t < 16 && m; t++) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t kc = nnd_readlane_u64(ckey[0], src);
            int cnt = __popcll(__ballot(mykey <= kc));
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ch++) cnt += __popcll(__ballot(ckey[ch] <= kc));
            if (cnt >= k && kc < bound) bound = kc;
        }
        if (bound != NND_EMPTY_KEY) {
            nv = 0;
#pragma unroll
            for (int ch = 0; ch < NCHUNK; ch++) {
                if (ckey[ch] > bound) ckey[ch] = NND_EMPTY_KEY;
                cmask[ch] = __ballot(ckey[ch] != NND_EMPTY_KEY);
                nv += __popcll(cmask[ch]);
            }
        }
    }

    // rank pass over the surviving candidates: list entries count how many precede them (shift), candidates count
    // how many candidates precede them; a short batch also picks up its list rank here (ballot), a long one in a
    // second pass over the filled list entries
    int shift = 0;
    int rank[NCHUNK];
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) rank[ch] = 0;
    const bool fused = nv <= nlist;
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) {
        unsigned long long m = cmask[ch];
        while (m) {
            const int src = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t kc = nnd_readlane_u64(ckey[ch], src);
            const bool before = kc < mykey;
            shift += before ? 1 : 0;
#pragma unroll
            for (int c2 = 0; c2 < NCHUNK; c2++) rank[c2] += (kc < ckey[c2]) ? 1 : 0;
            if (fused) {
                // list keys are distinct from candidate keys (dedupe above), so #list < kc = nlist - #(kc < list)
                const int below = nlist - __popcll(__ballot(before && lane < nlist));
                if (lane == src) rank[ch] += below;
            }
        }
    }
    if (!fused) {
        for (int j = 0; j < nlist; j++) {
            const uint64_t lj = nnd_readlane_u64(mykey, j);
#pragma unroll
            for (int c2 = 0; c2 < NCHUNK; c2++) rank[c2] += (lj < ckey[c2]) ? 1 : 0;
        }
    }
    // every lane has read its own entry already; positions are a bijection, so plain stores suffice
    if (lane < nlist && shift > 0 && lane + shift < k) {
        row_e[lane + shift] = e;
        row_d[lane + shift] = d;
        if (lane + shift == k - 1) *th_slot = d;  // new worst distance of the row
    }
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) {
        if (ckey[ch] != NND_EMPTY_KEY && rank[ch] < k) {
            row_e[rank[ch]] = nnd_key_idx(ckey[ch]) | NND_NEW_BIT;
            row_d[rank[ch]] = nnd_key_dist(ckey[ch]);
            if (rank[ch] == k - 1) *th_slot = nnd_key_dist(ckey[ch]);
        }
    }
    return pushed;
}

// Loader wrapper: fetches row v of the k-lists from global memory, then merges.
template <int NCHUNK, typename CandFn>
__device__ __forceinline__ int nnd_merge_row(int64_t v, int k, int ks, uint32_t *__restrict__ knn_e,
                                             float *__restrict__ knn_d, float *__restrict__ th, int ncand, CandFn cand) {
    const int lane = nnd_lane();
    uint32_t *row_e = knn_e + v * ks;
    float *row_d = knn_d + v * ks;
    uint32_t e = NND_EMPTY_E;
    float d = INFINITY;
    if (lane < k) {
        e = row_e[lane];
        d = row_d[lane];
    }
    return nnd_merge_row_regs<NCHUNK>(row_e, row_d, th + v, e, d, k, ncand, cand);
}

// ------------------------------------------------------------------------------------------------------------------
// Quarter-wave variant for k <= 16: FOUR rows per wave, 16 lanes each.  Lane j of a row's 16-lane group holds list
// entry j in (e, d) (lanes >= k: EMPTY / +inf) and, per block of 16 candidates, candidate blk*16 + j of ITS row.
// Same result as nnd_merge_row_regs (the k smallest keys of row U {candidates that beat the row's worst distance as it
// was at the start and are not in the row}), same return value (summed over the wave's rows), a quarter of the
// instructions: the row-wide steps are DPP row operations (rotate / shift by one inside a 16-lane row) instead of
// v_readlane + ballot over the whole wave.
//   filter : one compare per lane; duplicates by rotating the row's 16 neighbour ids past the 16 candidates;
//   insert : candidates are taken in lane order; the candidate is broadcast inside every row with ds_bpermute (the
//            next broadcast is in flight while the current one is inserted) and every list lane decides locally:
//            key > candidate: take the left neighbour's entry if that one moves too, else the candidate.
// row_on is false for the groups of a wave that have no row (they still execute, nothing is stored).
__device__ __forceinline__ int nnd_dpp_row_shr1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true); }
__device__ __forceinline__ int nnd_dpp_row_ror1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x121, 0xF, 0xF, false); }

// core: the row's entries (e, d) are updated IN the registers, nothing is stored (chained merges: k_merge_graph_rows_q)
template <int NBLK, typename CandFn>
__device__ __forceinline__ int nnd_merge_rows_q16_regs(bool row_on, uint32_t &e, float &d, int k, int ncand, CandFn cand) {
    const int lane = nnd_lane(), j = lane & 15, gbase = lane & 48;
    uint32_t klo = e & NND_IDX_MASK, khi = __float_as_uint(d);  // key = khi:klo; empty slots: 0x7FFFFFFF / +inf bits
    if (e == NND_EMPTY_E) { klo = 0xFFFFFFFFu; khi = 0xFFFFFFFFu; }
    uint32_t flag = e == NND_EMPTY_E ? 0u : (e & NND_NEW_BIT);
    const uint32_t myidx = e & NND_IDX_MASK;  // 0x7FFFFFFF for empty slots: never a valid id
    // worst distance of my row as it is now (+inf while the row is not full)
    const float th = __int_as_float(__builtin_amdgcn_ds_bpermute((gbase + k - 1) << 2, __float_as_int(d)));
    const bool any_list = __ballot(e != NND_EMPTY_E) != 0;  // first tree: every row of the wave is still empty
    int pushed = 0;
#pragma unroll
    for (int blk = 0; blk < NBLK; blk++) {
        if (blk * 16 >= ncand) break;  // wave-uniform
        const int c = blk * 16 + j;
        uint32_t cid = 0;
        float dc = 0.0f;
        bool ok = row_on && c < ncand && cand(c, cid, dc);
        ok = ok && (dc < th);  // strict, utils.py:484
        if (!__ballot(ok)) continue;
        if (any_list) {  // utils.py:489-492: drop candidates already in the row
            uint32_t rid = myidx;
            bool dup = false;
#pragma unroll
            for (int t = 0; t < 16; t++) {
                dup |= rid == cid;
                rid = (uint32_t)nnd_dpp_row_ror1((int)rid);
            }
            ok = ok && !dup;
        }
        const unsigned long long mask = __ballot(ok);
        if (!mask) continue;
        pushed += __popcll(mask);
        const uint32_t clo = ok ? cid : 0xFFFFFFFFu, chi = ok ? __float_as_uint(dc) : 0xFFFFFFFFu;
        uint32_t u = (uint32_t)(mask | (mask >> 16) | (mask >> 32) | (mask >> 48)) & 0xFFFFu;  // positions live in some row
        int pnext = __builtin_ctz(u);
        uint32_t nlo = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + pnext) << 2, (int)clo);
        uint32_t nhi = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + pnext) << 2, (int)chi);
        while (u) {
            const uint32_t blo = nlo, bhi = nhi;  // this step's candidate of my row (all-ones: nothing to insert)
            u &= u - 1;
            if (u) {
                pnext = __builtin_ctz(u);
                nlo = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + pnext) << 2, (int)clo);
                nhi = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + pnext) << 2, (int)chi);
            }
            const uint64_t mykey = ((uint64_t)khi << 32) | klo, ck = ((uint64_t)bhi << 32) | blo;
            const bool gt = mykey > ck;  // my entry moves one slot to the right (an all-ones candidate moves nothing)
            const int gtl = nnd_dpp_row_shr1(gt ? 1 : 0);  // does my left neighbour move too?  (lane 0 of the row: no)
            const uint32_t llo = (uint32_t)nnd_dpp_row_shr1((int)klo), lhi = (uint32_t)nnd_dpp_row_shr1((int)khi);
            const uint32_t lfl = (uint32_t)nnd_dpp_row_shr1((int)flag);
            if (gt) {
                klo = gtl ? llo : blo;
                khi = gtl ? lhi : bhi;
                flag = gtl ? lfl : NND_NEW_BIT;
            }
        }
    }
    if (pushed == 0) return 0;
    const bool empty = (klo & khi) == 0xFFFFFFFFu;
    e = empty ? NND_EMPTY_E : (klo | flag);
    d = empty ? INFINITY : __uint_as_float(khi);
    return pushed;
}

template <int NBLK, typename CandFn>
__device__ __forceinline__ int nnd_merge_rows_q16(bool row_on, uint32_t *__restrict__ row_e, float *__restrict__ row_d,
                                                  float *__restrict__ th_slot, uint32_t e, float d, int k, int ncand,
                                                  CandFn cand) {
    const int j = nnd_lane() & 15;
    const uint32_t e_in = e;
    const float d_in = d;
    const int pushed = nnd_merge_rows_q16_regs<NBLK>(row_on, e, d, k, ncand, cand);
    if (pushed == 0) return 0;
    if (row_on && j < k && (e != e_in || d != d_in)) {
        row_e[j] = e;
        row_d[j] = d;
        if (j == k - 1) *th_slot = d;  // new worst distance of the row
    }
    return pushed;
}

// ------------------------------------------------------------------------------------------------------------------
// Quarter-wave merge of MANY candidates per row (the leaf kernel: 40..80 leaf-mates against a row of k <= 16), round 6.
// nnd_merge_rows_q16 above inserts one candidate per step and a step serves the UNION of the candidate positions that
// are live in any of the wave's four rows: ~20 steps of ~15 VALU instructions per block-of-16 pass on a later tree, 41 on
// the first (profiles/r05_pmc_sq.txt: 200 M VALU wave-instructions per launch against 5.5 M MFMA).  Here the candidates
// that beat the row's CURRENT worst distance are first COMPACTED, row by row, into a wave-private LDS queue (32 entries
// per row), and a queue is folded into its row by a sorting network over the row's 16 lanes, all four rows of the wave
// at once, whatever the number of survivors: bitonic sort of the <= 16 queued keys (10 compare-exchange stages, partners
// by DPP quad_perm / row_half_mirror / row_mirror), elementwise min against the mirrored row (the 16 smallest of the
// union, a bitonic sequence), four half-cleaner stages.  Duplicates (utils.py:489-492) are screened against the row's ids
// after the compaction (16 row rotations per batch instead of 16 per block of candidates).  The threshold tightens after
// every batch -- the reference's sequential pushes compare with the heap's current root too (utils.py:484) -- so a row
// that starts empty keeps ~k + 12 of 41 leaf-mates instead of inserting all of them.
// Result: the k smallest keys of row U candidates, rows sorted by (dist, idx) -- what nnd_merge_rows_q16 leaves.
// Return value: candidates that entered a batch (beat the current threshold, not in the row); statistics only.
#define NND_Q16B_CAP 32  // queue entries per row: a batch is flushed before a block of 16 could overflow it
#define NND_DPP_QUAD_XOR1 0xB1         // quad_perm [1,0,3,2]
#define NND_DPP_QUAD_XOR2 0x4E         // quad_perm [2,3,0,1]
#define NND_DPP_QUAD_MIRROR 0x1B       // quad_perm [3,2,1,0]
#define NND_DPP_ROW_MIRROR 0x140
#define NND_DPP_ROW_HALF_MIRROR 0x141
#define NND_DPP_ROW_SHL(n) (0x100 + (n))

// partner value of a compare-exchange stage.  Every lane of a row has a valid source lane in these patterns, so the DPP
// moves are issued with bound_ctrl (no `old` operand to initialise, and the compiler may fold the move into its consumer).
// CTRL < 0: lane j ^ 4 of the row = half-row mirror of the quad mirror (no single DPP pattern)
template <int CTRL>
__device__ __forceinline__ uint32_t nnd_q16_partner(uint32_t v) {
    if constexpr (CTRL >= 0) {
        return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
    } else {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, NND_DPP_QUAD_MIRROR, 0xF, 0xF, true);  // j ^ 3
        return (uint32_t)__builtin_amdgcn_update_dpp(0, t, NND_DPP_ROW_HALF_MIRROR, 0xF, 0xF, true);  // (j ^ 3) ^ 7
    }
}
// one compare-exchange stage on 64-bit keys hi:lo; keep_max lanes end up with the larger key of the pair
template <int CTRL>
__device__ __forceinline__ void nnd_q16_cx(uint32_t &lo, uint32_t &hi, bool keep_max) {
    const uint32_t plo = nnd_q16_partner<CTRL>(lo), phi = nnd_q16_partner<CTRL>(hi);
    const bool less = (((uint64_t)phi << 32) | plo) < (((uint64_t)hi << 32) | lo);
    const bool take = less != keep_max;
    lo = take ? plo : lo;
    hi = take ? phi : hi;
}
template <int CTRL>
__device__ __forceinline__ void nnd_q16_cx(uint32_t &lo, uint32_t &hi, uint32_t &fl, bool keep_max) {
    const uint32_t plo = nnd_q16_partner<CTRL>(lo), phi = nnd_q16_partner<CTRL>(hi), pfl = nnd_q16_partner<CTRL>(fl);
    const bool less = (((uint64_t)phi << 32) | plo) < (((uint64_t)hi << 32) | lo);
    const bool take = less != keep_max;
    lo = take ? plo : lo;
    hi = take ? phi : hi;
    fl = take ? pfl : fl;
}
template <int T>
__device__ __forceinline__ bool nnd_q16_any_eq(uint32_t ids, uint32_t x) {  // does x equal any of the 16 ids of my row?
    if constexpr (T == 0) return ids == x;
    else return ((uint32_t)__builtin_amdgcn_update_dpp(0, (int)ids, NND_DPP_ROW_ROR(T), 0xF, 0xF, true) == x) | nnd_q16_any_eq<T - 1>(ids, x);
}

template <int NBLK, typename CandFn>
__device__ __forceinline__ int nnd_merge_rows_q16b(bool row_on, uint32_t *__restrict__ row_e, float *__restrict__ row_d,
                                                   float *__restrict__ th_slot, uint32_t e, float d, int k, int ncand,
                                                   CandFn cand, uint2 *wave_scr) {
    const int lane = nnd_lane(), j = lane & 15, gbase = lane & 48;
    uint2 *my = wave_scr + (lane >> 4) * NND_Q16B_CAP;
    const uint32_t e_in = e;
    const float d_in = d;
    uint32_t klo = e & NND_IDX_MASK, khi = __float_as_uint(d), flag = e & NND_NEW_BIT;  // key = khi:klo
    if (e == NND_EMPTY_E) { klo = 0xFFFFFFFFu; khi = 0xFFFFFFFFu; flag = 0u; }
    const bool any_list = __ballot(e != NND_EMPTY_E) != 0;  // first tree: every row of the wave is still empty
    // worst distance of my row as it is now (+inf while the row is not full: d is +inf in empty slots)
    float th_cur = __int_as_float(__builtin_amdgcn_ds_bpermute((gbase + k - 1) << 2, __float_as_int(d)));
    const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0, b2 = (j & 4) != 0, b3 = (j & 8) != 0;
    int fill = 0, pushed = 0;

    // ONE copy of the batch code per call site (the loops are kept rolled: the leaf kernel inlines this routine once per
    // pass over its rows, and a sorting network is ~150 instructions)
    const int nblk = (ncand + 15) >> 4;
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk++) {
        const int c = blk * 16 + j;
        uint32_t cid = 0;
        float dc = 0.0f;
        bool ok = row_on && c < ncand && cand(c, cid, dc);
        ok = ok && (dc < th_cur);  // strict, utils.py:484
        const unsigned long long cmask = __ballot(ok);
        if (cmask) {
            const uint32_t m16 = (uint32_t)(cmask >> gbase) & 0xFFFFu;
            if (ok) my[fill + __popc(m16 & ((1u << j) - 1u))] = make_uint2(cid, __float_as_uint(dc));
            fill += __popc(m16);
        }
        // fold the queues into the rows when a queue could overflow with the next block, at the last block, and while a
        // row is not full (it accepts everything: fold what it has so that the next block meets a threshold)
        const bool last = blk + 1 >= nblk;
        if (!__ballot(fill > 16 || (fill > 0 && (last || th_cur == INFINITY)))) continue;
        nnd_wave_lds_sync();
#pragma unroll 1
        for (int b = 0; b < NND_Q16B_CAP / 16; b++) {
            if (!__ballot(fill > 16 * b)) break;  // wave-uniform
            const uint2 s = my[16 * b + j];
            bool sok = 16 * b + j < fill && __uint_as_float(s.y) < th_cur;  // queued against an older threshold: test again
            if (any_list && __ballot(sok)) {  // utils.py:489-492 (ids of this leaf are unique)
                // every lane evaluates the rotations: a DPP source lane that is masked off reads as zero
                const bool dup = nnd_q16_any_eq<15>(klo, s.x);
                sok = sok && !dup;
            }
            const unsigned long long mask = __ballot(sok);
            if (!mask) continue;
            pushed += __popcll(mask);
            uint32_t slo = sok ? s.x : 0xFFFFFFFFu, shi = sok ? s.y : 0xFFFFFFFFu;
            // bitonic sort of the batch, ascending over the row's lanes
            nnd_q16_cx<NND_DPP_QUAD_XOR1>(slo, shi, b0);
            nnd_q16_cx<NND_DPP_QUAD_MIRROR>(slo, shi, b1);
            nnd_q16_cx<NND_DPP_QUAD_XOR1>(slo, shi, b0);
            nnd_q16_cx<NND_DPP_ROW_HALF_MIRROR>(slo, shi, b2);
            nnd_q16_cx<NND_DPP_QUAD_XOR2>(slo, shi, b1);
            nnd_q16_cx<NND_DPP_QUAD_XOR1>(slo, shi, b0);
            nnd_q16_cx<NND_DPP_ROW_MIRROR>(slo, shi, b3);
            nnd_q16_cx<-1>(slo, shi, b2);
            nnd_q16_cx<NND_DPP_QUAD_XOR2>(slo, shi, b1);
            nnd_q16_cx<NND_DPP_QUAD_XOR1>(slo, shi, b0);
            // the 16 smallest of row U batch: min(row[j], batch[15 - j]) is bitonic; four half-cleaners sort it
            {
                const uint32_t tlo = nnd_q16_partner<NND_DPP_ROW_MIRROR>(slo), thi = nnd_q16_partner<NND_DPP_ROW_MIRROR>(shi);
                const bool take = (((uint64_t)thi << 32) | tlo) < (((uint64_t)khi << 32) | klo);
                klo = take ? tlo : klo;
                khi = take ? thi : khi;
                flag = take ? NND_NEW_BIT : flag;
            }
            nnd_q16_cx<NND_DPP_ROW_ROR(8)>(klo, khi, flag, b3);
            nnd_q16_cx<-1>(klo, khi, flag, b2);
            nnd_q16_cx<NND_DPP_QUAD_XOR2>(klo, khi, flag, b1);
            nnd_q16_cx<NND_DPP_QUAD_XOR1>(klo, khi, flag, b0);
            if (j >= k) { klo = 0xFFFFFFFFu; khi = 0xFFFFFFFFu; flag = 0u; }  // k < 16: the row ends at slot k - 1
            const uint32_t wk = (uint32_t)__builtin_amdgcn_ds_bpermute((gbase + k - 1) << 2, (int)khi);
            th_cur = wk == 0xFFFFFFFFu ? INFINITY : __uint_as_float(wk);
        }
        fill = 0;
        nnd_wave_lds_sync();  // the queue is read: the next block may overwrite it
    }
    if (pushed == 0) return 0;
    const bool empty = (klo & khi) == 0xFFFFFFFFu;
    e = empty ? NND_EMPTY_E : (klo | flag);
    d = empty ? INFINITY : __uint_as_float(khi);
    if (row_on && j < k && (e != e_in || d != d_in)) {
        row_e[j] = e;
        row_d[j] = d;
        if (j == k - 1) *th_slot = d;  // new worst distance of the row
    }
    return pushed;
}

// ------------------------------------------------------------------------------------------------------------------
// The same merge for rows of up to 32 neighbours (k = 30, the reference's default): TWO rows per wave, 32 lanes each (round 6).
// Round 4 built two-rows-per-wave merges twice (insertion, rank counting) and measured no gain; what differs here is what
// differed at k <= 16: the candidates are compacted first (a later tree leaves ~25 of ~87 leaf-mates under the threshold) and a
// queue of up to 32 is folded by a sorting network over the row's 32 lanes -- 15 compare-exchange stages to sort the batch, the
// elementwise min against the mirrored row, 5 half-cleaners.  Partners inside a 16-lane DPP row come by DPP, the three stages
// that cross rows (lane ^ 31 twice, lane ^ 16) by ds_bpermute.
#define NND_Q32B_CAP 64  // queue entries per row
template <int PX>
__device__ __forceinline__ uint32_t nnd_q32_partner(uint32_t v, int lane) {  // value of lane ^ PX (inside the 32-lane half)
    if constexpr (PX == 1) return nnd_q16_partner<NND_DPP_QUAD_XOR1>(v);
    else if constexpr (PX == 2) return nnd_q16_partner<NND_DPP_QUAD_XOR2>(v);
    else if constexpr (PX == 3) return nnd_q16_partner<NND_DPP_QUAD_MIRROR>(v);
    else if constexpr (PX == 4) return nnd_q16_partner<-1>(v);
    else if constexpr (PX == 7) return nnd_q16_partner<NND_DPP_ROW_HALF_MIRROR>(v);
    else if constexpr (PX == 8) return nnd_q16_partner<NND_DPP_ROW_ROR(8)>(v);
    else if constexpr (PX == 15) return nnd_q16_partner<NND_DPP_ROW_MIRROR>(v);
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ PX) << 2, (int)v);  // 16, 31: across the DPP rows
}
template <int PX>
__device__ __forceinline__ void nnd_q32_cx(uint32_t &lo, uint32_t &hi, bool keep_max, int lane) {
    const uint32_t plo = nnd_q32_partner<PX>(lo, lane), phi = nnd_q32_partner<PX>(hi, lane);
    const bool take = ((((uint64_t)phi << 32) | plo) < (((uint64_t)hi << 32) | lo)) != keep_max;
    lo = take ? plo : lo;
    hi = take ? phi : hi;
}
template <int PX>
__device__ __forceinline__ void nnd_q32_cx(uint32_t &lo, uint32_t &hi, uint32_t &fl, bool keep_max, int lane) {
    const uint32_t plo = nnd_q32_partner<PX>(lo, lane), phi = nnd_q32_partner<PX>(hi, lane), pfl = nnd_q32_partner<PX>(fl, lane);
    const bool take = ((((uint64_t)phi << 32) | plo) < (((uint64_t)hi << 32) | lo)) != keep_max;
    lo = take ? plo : lo;
    hi = take ? phi : hi;
    fl = take ? pfl : fl;
}

template <typename CandFn>
__device__ __forceinline__ int nnd_merge_rows_q32b(bool row_on, uint32_t *__restrict__ row_e, float *__restrict__ row_d,
                                                   float *__restrict__ th_slot, uint32_t e, float d, int k, int ncand,
                                                   CandFn cand, uint2 *wave_scr) {
    const int lane = nnd_lane(), j = lane & 31, gb = lane & 32;
    uint2 *my = wave_scr + (lane >> 5) * NND_Q32B_CAP;
    const uint32_t e_in = e;
    const float d_in = d;
    uint32_t klo = e & NND_IDX_MASK, khi = __float_as_uint(d), flag = e & NND_NEW_BIT;  // key = khi:klo
    if (e == NND_EMPTY_E) { klo = 0xFFFFFFFFu; khi = 0xFFFFFFFFu; flag = 0u; }
    const bool any_list = __ballot(e != NND_EMPTY_E) != 0;
    float th_cur = __int_as_float(__builtin_amdgcn_ds_bpermute((gb + k - 1) << 2, __float_as_int(d)));  // (+inf while the row is not full)
    const bool b0 = (j & 1) != 0, b1 = (j & 2) != 0, b2 = (j & 4) != 0, b3 = (j & 8) != 0, b4 = (j & 16) != 0;
    const uint32_t below = (1u << j) - 1u;
    int fill = 0, pushed = 0;
    const int nblk = (ncand + 31) >> 5;
#pragma unroll 1
    for (int blk = 0; blk < nblk; blk++) {
        const int c = blk * 32 + j;
        uint32_t cid = 0;
        float dc = 0.0f;
        bool ok = row_on && c < ncand && cand(c, cid, dc);
        ok = ok && (dc < th_cur);  // strict, utils.py:484
        const unsigned long long cmask = __ballot(ok);
        if (cmask) {
            const uint32_t m32 = (uint32_t)(cmask >> gb);
            if (ok) my[fill + __popc(m32 & below)] = make_uint2(cid, __float_as_uint(dc));
            fill += __popc(m32);
        }
        const bool last = blk + 1 >= nblk;
        if (!__ballot(fill > 32 || (fill > 0 && (last || th_cur == INFINITY)))) continue;
        nnd_wave_lds_sync();
#pragma unroll 1
        for (int b = 0; b < NND_Q32B_CAP / 32; b++) {
            if (!__ballot(fill > 32 * b)) break;  // wave-uniform
            const uint2 s = my[32 * b + j];
            bool sok = 32 * b + j < fill && __uint_as_float(s.y) < th_cur;  // queued against an older threshold: test again
            if (any_list && __ballot(sok)) {  // utils.py:489-492 (every lane evaluates the rotations: see nnd_merge_rows_q16b)
                const uint32_t other = (uint32_t)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)klo);
                const bool dup0 = nnd_q16_any_eq<15>(klo, s.x), dup1 = nnd_q16_any_eq<15>(other, s.x);
                const bool dup = dup0 || dup1;
                sok = sok && !dup;
            }
            const unsigned long long mask = __ballot(sok);
            if (!mask) continue;
            pushed += __popcll(mask);
            uint32_t slo = sok ? s.x : 0xFFFFFFFFu, shi = sok ? s.y : 0xFFFFFFFFu;
            // bitonic sort of the batch, ascending over the row's 32 lanes
            nnd_q32_cx<1>(slo, shi, b0, lane);
            nnd_q32_cx<3>(slo, shi, b1, lane);
            nnd_q32_cx<1>(slo, shi, b0, lane);
            nnd_q32_cx<7>(slo, shi, b2, lane);
            nnd_q32_cx<2>(slo, shi, b1, lane);
            nnd_q32_cx<1>(slo, shi, b0, lane);
            nnd_q32_cx<15>(slo, shi, b3, lane);
            nnd_q32_cx<4>(slo, shi, b2, lane);
            nnd_q32_cx<2>(slo, shi, b1, lane);
            nnd_q32_cx<1>(slo, shi, b0, lane);
            nnd_q32_cx<31>(slo, shi, b4, lane);
            nnd_q32_cx<8>(slo, shi, b3, lane);
            nnd_q32_cx<4>(slo, shi, b2, lane);
            nnd_q32_cx<2>(slo, shi, b1, lane);
            nnd_q32_cx<1>(slo, shi, b0, lane);
            {   // the 32 smallest of row U batch: min(row[j], batch[31 - j]) is bitonic; five half-cleaners sort it
                const uint32_t tlo = nnd_q32_partner<31>(slo, lane), thi = nnd_q32_partner<31>(shi, lane);
                const bool take = (((uint64_t)thi << 32) | tlo) < (((uint64_t)khi << 32) | klo);
                klo = take ? tlo : klo;
                khi = take ? thi : khi;
                flag = take ? NND_NEW_BIT : flag;
            }
            nnd_q32_cx<16>(klo, khi, flag, b4, lane);
            nnd_q32_cx<8>(klo, khi, flag, b3, lane);
            nnd_q32_cx<4>(klo, khi, flag, b2, lane);
            nnd_q32_cx<2>(klo, khi, flag, b1, lane);
            nnd_q32_cx<1>(klo, khi, flag, b0, lane);
            if (j >= k) { klo = 0xFFFFFFFFu; khi = 0xFFFFFFFFu; flag = 0u; }  // k < 32: the row ends at slot k - 1
            const uint32_t wk = (uint32_t)__builtin_amdgcn_ds_bpermute((gb + k - 1) << 2, (int)khi);
            th_cur = wk == 0xFFFFFFFFu ? INFINITY : __uint_as_float(wk);
        }
        fill = 0;
        nnd_wave_lds_sync();  // the queue is read: the next block may overwrite it
    }
    if (pushed == 0) return 0;
    const bool empty = (klo & khi) == 0xFFFFFFFFu;
    e = empty ? NND_EMPTY_E : (klo | flag);
    d = empty ? INFINITY : __uint_as_float(khi);
    if (row_on && j < k && (e != e_in || d != d_in)) {
        row_e[j] = e;
        row_d[j] = d;
        if (j == k - 1) *th_slot = d;  // new worst distance of the row
    }
    return pushed;
}

// ------------------------------------------------------------------------------------------------------------------
// Wide rows, 64 < k <= NND_WIDE_K = 256 (the reference has no bound on n_neighbors, utils.py:130-158; 256 is also the cap of its
// default leaf size, rp_trees.py:2845): the row does not fit one entry per lane -- a lane holds entries lane, 64 + lane, ... --
// so it is merged through LDS -- same result as nnd_merge_row_regs (the k smallest keys of row U {candidates
// that beat the row's worst distance as it was at the start and are not in the row}), same return value, by the same
// rank counting; the loops read LDS broadcasts instead of v_readlane.  `scr`: NND_WIDE_SCRATCH_WORDS 64-bit words of
// LDS private to this wave.  The row is read from and written to global memory here (every row is merged at most once
// per launch by its callers).  Not tuned: k > 64 builds instead of raising; the k <= 64 paths are the fast ones.
#define NND_WIDE_MAXC 256
#define NND_WIDE_SCRATCH_WORDS (NND_WIDE_K + NND_WIDE_MAXC)
template <int NCHUNK, typename CandFn>
__device__ __forceinline__ int nnd_merge_row_lds(uint64_t *scr, uint32_t *__restrict__ row_e, float *__restrict__ row_d,
                                                 float *__restrict__ th_slot, int k, int ncand, CandFn cand) {
    static_assert(NCHUNK * 64 <= NND_WIDE_MAXC, "candidate scratch");
    const int lane = nnd_lane();
    uint64_t *rkey = scr;               // [k] the row's keys (dist_bits << 32 | idx | flag kept apart below)
    uint64_t *ckey = scr + NND_WIDE_K;  // [nv] surviving candidates' keys, compacted
    uint32_t me[NND_WIDE_U];
    float md[NND_WIDE_U];
    int nlist = 0;
#pragma unroll
    for (int u = 0; u < NND_WIDE_U; u++) {
        const int j = lane + 64 * u;
        me[u] = j < k ? row_e[j] : NND_EMPTY_E;
        md[u] = j < k ? row_d[j] : INFINITY;
        if (j < k) rkey[j] = me[u] == NND_EMPTY_E ? NND_EMPTY_KEY : nnd_make_key(md[u], me[u]);
        nlist += __popcll(__ballot(me[u] != NND_EMPTY_E));
    }
    nnd_wave_lds_sync();
    const float th = nlist >= k ? nnd_key_dist(rkey[k - 1]) : INFINITY;  // strict, utils.py:484
    // filter + dedupe (utils.py:489-492), then compact the survivors' keys
    uint64_t mykey[NCHUNK];
    int nv = 0;
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) {
        const int c = ch * 64 + lane;
        uint32_t id = 0;
        float dc = 0.0f;
        bool ok = false;
        if (ch * 64 < ncand) {  // wave-uniform
            ok = (c < ncand) && cand(c, id, dc) && (dc < th);
            if (__ballot(ok))
                for (int j = 0; j < nlist; j++) ok = ok && (nnd_key_idx(rkey[j]) != id);
        }
        mykey[ch] = ok ? nnd_make_key(dc, id) : NND_EMPTY_KEY;
        const unsigned long long m = __ballot(ok);
        if (ok) ckey[nv + nnd_prefix_popc(m)] = mykey[ch];
        nv += __popcll(m);
    }
    if (nv == 0) return 0;
    nnd_wave_lds_sync();
    // list entries: new position = old position + #{candidates before it}
    int shift[NND_WIDE_U];
#pragma unroll
    for (int u = 0; u < NND_WIDE_U; u++) shift[u] = 0;
    for (int c = 0; c < nv; c++) {
        const uint64_t kc = ckey[c];
#pragma unroll
        for (int u = 0; u < NND_WIDE_U; u++) {
            const uint64_t rk = me[u] == NND_EMPTY_E ? NND_EMPTY_KEY : nnd_make_key(md[u], me[u]);
            shift[u] += kc < rk ? 1 : 0;
        }
    }
    // candidates: rank = #{list entries before it} + #{candidates before it}
    int rank[NCHUNK];
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) rank[ch] = 0;
    for (int j = 0; j < nlist; j++) {
        const uint64_t lj = rkey[j];
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ch++) rank[ch] += lj < mykey[ch] ? 1 : 0;
    }
    for (int c = 0; c < nv; c++) {
        const uint64_t kc = ckey[c];
#pragma unroll
        for (int ch = 0; ch < NCHUNK; ch++) rank[ch] += kc < mykey[ch] ? 1 : 0;
    }
    nnd_wave_lds_sync();  // every lane has read what it needs from the LDS copies (the caller may reuse them)
#pragma unroll
    for (int u = 0; u < NND_WIDE_U; u++) {
        const int j = lane + 64 * u;
        if (j < nlist && shift[u] > 0 && j + shift[u] < k) {
            row_e[j + shift[u]] = me[u];
            row_d[j + shift[u]] = md[u];
            if (j + shift[u] == k - 1) *th_slot = md[u];
        }
    }
#pragma unroll
    for (int ch = 0; ch < NCHUNK; ch++) {
        if (mykey[ch] != NND_EMPTY_KEY && rank[ch] < k) {
            row_e[rank[ch]] = nnd_key_idx(mykey[ch]) | NND_NEW_BIT;
            row_d[rank[ch]] = nnd_key_dist(mykey[ch]);
            if (rank[ch] == k - 1) *th_slot = nnd_key_dist(mykey[ch]);
        }
    }
    return nv;
}
