// join.hip -- the NN-descent local join (dominant kernel of the build).
//
// Replaces generate_graph_update_array (reference utils.py:536-658): for every vertex v, every
// pair (p, q) with p in new[v], q in new[v] from p's own slot on (the p == q self pair included,
// utils.py:619) or q in old[v] (utils.py:640): d = dist(x_p, x_q); keep if d <= max(th_p, th_q).
// The reference writes every kept (p,q,d) to a slab and apply_graph_update_array (utils.py:661-733)
// later attempts BOTH pushes; a push succeeds iff d < th_target and the source is not already in the
// target's list (utils.py:484-492).  Measured on the reference: ~half of the evaluated pairs pass
// the max() filter but only 1-5 % of pushes succeed, almost all failures being "already present".
//
// MI355X design (one kernel family, k_local_join16 for max_candidates <= 16 and k_local_join_w<32|64> above that)
//   * ONE WAVE PER VERTEX, no workgroup barrier, no LDS row tile: a wave computes its vertex's whole
//     [new] x [new | old] block, nothing is shared between waves, so every lane loads its MFMA operands straight
//     from global memory in MFMA layout (lane (r16, g) holds 16-byte chunks 4t+g of candidate row r16 of a tile);
//     the gather of candidate rows is the HBM term that prices the kernel (C_i * row bytes);
//   * the new x (new U old) distance block is a Gram contraction on the f32 MFMA pipe (16x16x4; tiles below the
//     diagonal of new x new skipped; only filled slots are fetched);
//   * software pipeline per wave: right after the last MFMA has consumed the operand registers, the gather of the
//     wave's next vertex is issued into them and flies during the epilogue; candidate ids are prefetched two
//     vertices ahead;
//   * epilogue: each endpoint is tested on its OWN threshold (d < th_p for p, d < th_q for q); survivors go to a
//     wave-private LDS queue that is drained on full waves: membership test against the target's current
//     neighbour ids (a proposal leaves the wave only if the push would succeed on the snapshot), then
//   * surviving proposals go to a per-target bank of PCAP hashed slots with a 64-bit atomicMin on
//     (dist_bits << 32 | source): order independent (deterministic), duplicate proposals of the
//     same source collapse into one slot, and a slot collision keeps the nearer source;
//   * vertices are visited in the first tree's leaf order, one contiguous eighth per XCD.
#include "common.h"
#include "gram.h"
#include "state.h"


// is `id` among the neighbour ids of LDS row kl?  All KS16*4 16-byte reads are issued before the first compare.
template <int KS16>
__device__ __forceinline__ bool klist_has(const uint32_t *kl, uint32_t id) {
    uint4 w[KS16 * 4];
#pragma unroll
    for (int c = 0; c < KS16 * 4; c++) w[c] = *(const uint4 *)(kl + 4 * c);
    bool present = false;
#pragma unroll
    for (int c = 0; c < KS16 * 4; c++) present |= (w[c].x == id) | (w[c].y == id) | (w[c].z == id) | (w[c].w == id);
    return present;
}

// ------------------------------------------------------------------------------------------------
// max_candidates <= 16 (k <= 16: the BASELINE / UMAP regime): one WAVE per vertex, no workgroup barrier and no
// LDS row tile at all.  One wave computes the whole [new] x [new | old] block, so nothing is shared between waves
// and staging rows in LDS would be pure overhead: every lane loads its MFMA operands STRAIGHT from global memory
// in MFMA layout -- lane (r16, g) holds 16-byte chunks 4t+g of candidate row r16 (new) and 16+r16 (old); the
// new x new tile uses the same registers for both operands.  The register budget (<= 128) lets 4 waves share a
// SIMD; LDS per wave is ~7.5 KB (ids, norms, thresholds, neighbour lists, the pair queue).  Right after the last
// MFMA has consumed the row registers the gather of the wave's next vertex is issued into them, so it is in
// flight during the whole epilogue.
#ifndef NND_J16_WAVES
#define NND_J16_WAVES 3
#endif
template <int DC, int KS16, bool SHARD>
__global__ __launch_bounds__(256, NND_J16_WAVES) void k_local_join16(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                         int metric, const int32_t *__restrict__ cand,
                                                         const int32_t *__restrict__ order, int64_t v_begin,
                                                         int64_t v_end, int k, int ks, const uint32_t *__restrict__ knn_e,
                                                         const float *__restrict__ th, uint64_t *__restrict__ pbuf,
                                                         uint8_t *__restrict__ pdirty, int pcap, uint32_t slot_seed,
                                                         long long *__restrict__ counters, int64_t own_lo, int64_t own_hi,
                                                         uint64_t *__restrict__ pbuf_r, int pcap_r, int64_t rt_lo, int64_t rt_hi) {
    constexpr int MCP = 16, RV = 32;          // rows per vertex: [new(16) | old(16)]
    constexpr int NT = DC / 16;               // 16-byte chunks per lane, row and K block
    constexpr int KQ = KS16 * 4;              // uint4 chunks per neighbour-list row
    constexpr int KQL = RV * KQ / 64;         // of which per lane
    constexpr int RPK = 64 / KQ;              // rows per neighbour-list load
    constexpr int kls = KS16 * 16 + 4;        // padded neighbour-list row stride (words)
#ifndef NND_J16_QCAP
#define NND_J16_QCAP (8 * 64)
#endif
    constexpr int QCAP = NND_J16_QCAP;        // pair queue: 8 (tile, row) combinations x 64 lanes; 4 x 64: drained after every tile
    constexpr int WAVE_BYTES = QCAP * 8 + 2 * RV * 4 + 4 * 4 + 5 * RV * 4 + RV * kls * 4 + 8;
    // klist rows are read / written as 16-byte vectors: everything in front of them is a multiple of 16 bytes
    static_assert((QCAP * 8 + 2 * RV * 4 + 4 * 4 + 5 * RV * 4) % 16 == 0 && (kls * 4) % 16 == 0, "klist must be 16-byte aligned");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    unsigned char *mine = smem + (size_t)w * ((WAVE_BYTES + 15) & ~15);
    uint2 *queue = (uint2 *)mine;                       // QCAP
    int32_t *cidbuf = (int32_t *)(queue + QCAP);        // 2 * RV
    int32_t *nnewbuf = cidbuf + 2 * RV;                 // 2 used, padded to 4 (16 bytes)
    int32_t *cid = nnewbuf + 4;                         // RV
    float *cnrm = (float *)(cid + RV);                  // RV
    float *cth = cnrm + RV;                             // RV
    uint32_t *cslot = (uint32_t *)(cth + RV);           // RV: proposal slot of each candidate id (hashed once per vertex)
    uint32_t *cflag = cslot + RV;                       // RV: 1 when a proposal was stored for the row (-> pdirty)
    uint32_t *klist = cflag + RV;                       // RV * kls
    const int kq = ks >> 2;
    // [own_lo, own_hi): the rows whose CURRENT neighbour lists this handle holds (membership tests); [rt_lo, rt_hi): the
    // rows it owns.  Proposal slot of target t: a shard keeps proposals for vertices owned ELSEWHERE in a narrow table
    // (pcap_r slots per row: a rank sends a remote row a few proposals per iteration, and the export streams that table)
    auto prop_slot = [&](int t, uint32_t slot) __attribute__((always_inline)) -> unsigned long long * {
        if (SHARD && ((int64_t)t < rt_lo || (int64_t)t >= rt_hi))  // (SHARD = false: the plain build carries none of this)
            return (unsigned long long *)&pbuf_r[(int64_t)t * pcap_r + (slot & (uint32_t)(pcap_r - 1))];
        return (unsigned long long *)&pbuf[(int64_t)t * pcap + slot];
    };
    const int r16 = lane & 15, gq = lane >> 4;
    // Vertices are visited in `order` (the first tree's leaf order when there is a forest): vertices that are close
    // in space run at the same time, and their candidate sets overlap heavily, so most row / neighbour-list gathers
    // and proposal atomics of a window hit L2 instead of HBM.  Each XCD (own L2; workgroups are dealt round-robin to
    // the 8 XCDs) walks its own contiguous eighth of the order.
    int64_t n_v = v_end - v_begin;
    int64_t g, stride;
    if ((gridDim.x & 7) == 0) {
        const int64_t per = (((n_v + 7) >> 3) + 3) & ~(int64_t)3;
        const int64_t g0 = (int64_t)(blockIdx.x & 7) * per;
        n_v = n_v < g0 + per ? n_v : g0 + per;  // this XCD's range is [g0, n_v)
        g = g0 + (int64_t)(blockIdx.x >> 3) * 4 + w;
        stride = (int64_t)(gridDim.x >> 3) * 4;
    } else {
        g = (int64_t)blockIdx.x * 4 + w;
        stride = (int64_t)gridDim.x * 4;
    }

    auto load_cand = [&](int64_t g) __attribute__((always_inline)) -> int {
        const int row = lane < RV ? lane : 0;
        const bool ok = lane < RV && g < n_v;
        const int64_t gg = ok ? g : 0;  // idle lanes read the launch's first vertex: on a shard the table holds the owned rows only
        const int64_t v = order ? (int64_t)order[v_begin + gg] : v_begin + gg;
        const int c = cand[v * RV + row];
        return ok ? c : -1;
    };
    auto store_cand = [&](int buf, int c) __attribute__((always_inline)) {
        const unsigned long long m = __ballot(c >= 0 && lane < RV);  // both lists are filled from the front
        if (lane < RV) cidbuf[buf * RV + lane] = c;
        if (lane == 0) nnewbuf[buf] = __popcll(m & 0xFFFFull) | (__popcll(m >> MCP) << 8);  // n_new | n_old << 8
    };
    f32x4 ra[NT], rb[NT];  // this lane's chunks of its new row / its old row (one K block)
    u32x4 klv[KQL];
    float nx_nrm = 0.0f, nx_th = 0.0f;
    int nx_id = -1;
    // rows of K block [c0, c0 + cw) of the vertex whose ids are in cb[]; empty slots read row 0 (cache hit, masked later)
    auto load_rows = [&](const int32_t *cb, int c0, int cw, bool with_old) __attribute__((always_inline)) {
        const int ida = cb[r16], idb = cb[MCP + r16];
        const float *pa = xp + (int64_t)(ida >= 0 ? ida : 0) * dp + c0 + 4 * gq;
        const float *pb = xp + (int64_t)(idb >= 0 ? idb : 0) * dp + c0 + 4 * gq;
#pragma clang loop unroll(full)
        for (int t = 0; t < NT; t++)
            if (16 * t < cw) ra[t] = *(const f32x4 *)(pa + 16 * t);
        if (with_old) {
#pragma clang loop unroll(full)
            for (int t = 0; t < NT; t++)
                if (16 * t < cw) rb[t] = *(const f32x4 *)(pb + 16 * t);
        }
    };
    auto issue_gather = [&](int buf) __attribute__((always_inline)) {
        const int32_t *cb = cidbuf + buf * RV;
        const int cnt = nnewbuf[buf], nn = cnt & 255, no = cnt >> 8;
        if (nn == 0) return;  // wave-uniform: the vertex has no new candidate, no join (utils.py:611-613)
        load_rows(cb, 0, dp < DC ? dp : DC, no > 0);
        {
            const int row = lane < RV ? lane : 0;
            nx_id = lane < RV ? cb[row] : -1;
            const int64_t ide = nx_id >= 0 ? nx_id : 0;
            nx_nrm = nrm[ide];
            nx_th = th[ide];  // compact per-row worst distance (L2 resident), not a 128-byte line per candidate
#ifdef NND_JOIN_EXTRA_LOAD  // perturbation experiment: what does one more random 4-byte load per candidate cost?
            nx_nrm += nrm[(uint32_t)(((uint64_t)(uint32_t)ide * 2654435761ull) % (uint64_t)own_hi)] * 0.0f;
#endif
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < KQL; i++) {
            const int r0 = i * RPK;  // neighbour lists of the filled slots only (wave-uniform trip counts)
            if (!(r0 < MCP ? r0 < nn : r0 - MCP < no)) continue;
            const int idx = lane + i * 64;
            const int r = idx / KQ, c = idx % KQ;
            const int id = cb[r];
            const bool ok = c < kq && id >= 0;
            klv[i] = *(const u32x4 *)(knn_e + (ok ? (int64_t)id * ks + 4 * c : 0));  // raw; masked when it lands
        }
    };
    auto land_gather = [&](int buf) __attribute__((always_inline)) {
        const int32_t *cb = cidbuf + buf * RV;
        const int cnt = nnewbuf[buf], nn = cnt & 255, no = cnt >> 8;
        if (lane < RV) {
            cid[lane] = nx_id;
            cnrm[lane] = nx_nrm;
            cth[lane] = nx_th;
            cslot[lane] = nnd_hash2(slot_seed, (uint32_t)nx_id) & (uint32_t)(pcap - 1);
            cflag[lane] = 0;
        }
#pragma clang loop unroll(full)
        for (int i = 0; i < KQL; i++) {
            const int r0 = i * RPK;
            if (!(r0 < MCP ? r0 < nn : r0 - MCP < no)) continue;
            const int idx = lane + i * 64;
            const int r = idx / KQ, c = idx % KQ;
            // padding beyond k is EMPTY already; a candidate owned by ANOTHER rank (row-sharded build) has no current
            // neighbour list here: no membership test for it, its owner dedups in the merge (utils.py:489-492)
            const bool ok = c < kq && cb[r] >= 0 && (int64_t)cb[r] >= own_lo && (int64_t)cb[r] < own_hi;
            const u32x4 empty = {NND_IDX_MASK, NND_IDX_MASK, NND_IDX_MASK, NND_IDX_MASK};
            *(u32x4 *)(klist + r * kls + 4 * c) = ok ? (klv[i] & NND_IDX_MASK) : empty;
        }
    };

    // The wave walks only the vertices of its sequence g, g + stride, ... that HAVE a new candidate (the lists are filled
    // from the front: slot 0 tells).  64 of them are tested at a time, one per lane, so a run of idle vertices costs
    // one pair of loads instead of one trip through the pipeline each -- in the last iterations, where a few percent
    // of the vertices still join, that trip (~1.5 us of dependent latency) was most of the kernel's time.
    int64_t scan_g = g;            // first vertex of the next block of 64 to test
    unsigned long long scan_m = 0; // vertices of the current block still to visit
    int64_t scan_base = g;
    auto next_active = [&]() __attribute__((always_inline)) -> int64_t {
        while (scan_m == 0) {
            if (scan_g >= n_v) return n_v;  // wave-uniform
            const int64_t gl = scan_g + (int64_t)lane * stride;
            int c0 = -1;
            if (gl < n_v) {
                const int64_t v = order ? (int64_t)order[v_begin + gl] : v_begin + gl;
                c0 = cand[v * RV];
            }
            scan_m = __ballot(c0 >= 0);
            scan_base = scan_g;
            scan_g += 64 * stride;
        }
        const int bit = __builtin_ctzll(scan_m);
        scan_m &= scan_m - 1;
        return scan_base + (int64_t)bit * stride;
    };
    g = next_active();
    int64_t g1 = next_active(), g2 = n_v;
    store_cand(0, load_cand(g));
    store_cand(1, load_cand(g1));
    nnd_wave_lds_sync();
    issue_gather(0);

    int tot_pairs = 0, tot_prop = 0, tot_rows = 0, tot_act = 0, tot_tiles = 0;
    for (int it = 0; g < n_v; g = g1, g1 = g2, it++) {
        g2 = next_active();
        const int cur = it & 1;
        const int my_cnt = nnewbuf[cur], my_new = my_cnt & 255;
        const bool has_old = (my_cnt >> 8) > 0;
        if (my_new > 0) tot_tiles += has_old ? 2 : 1;  // 16x16 Gram tiles of this vertex (wave-uniform)
        const int c2 = load_cand(g2);
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        if (my_new > 0) {
            land_gather(cur);
            nnd_wave_lds_sync();
            for (int c0 = 0; c0 < dp; c0 += DC) {
                const int cw = (dp - c0) < DC ? (dp - c0) : DC;
                if (c0 > 0) load_rows(cid, c0, cw, has_old);  // rows wider than one K block: the rest is fetched in turn
#ifndef NND_JOIN_NOGRAM
#pragma clang loop unroll(full)
                for (int t = 0; t < NT; t++) {
                    if (16 * t >= cw) continue;
                    if (has_old) {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].x, ra[t].x, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].x, rb[t].x, acc[1], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].y, ra[t].y, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].y, rb[t].y, acc[1], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].z, ra[t].z, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].z, rb[t].z, acc[1], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].w, ra[t].w, acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].w, rb[t].w, acc[1], 0, 0, 0);
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].x, ra[t].x, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].y, ra[t].y, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].z, ra[t].z, acc[0], 0, 0, 0);
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra[t].w, ra[t].w, acc[0], 0, 0, 0);
                    }
                }
#endif
            }
        }
        // the row registers are free: the gather of this wave's next vertex flies during the epilogue
        if (g1 < n_v) issue_gather(cur ^ 1);
        if (my_new > 0) {
            // Epilogue in two steps.  (a) every lane screens its 8 pairs against the two thresholds and pushes the few
            // that pass into a wave-private queue; (b) the queue is drained 64 entries at a time, so the neighbour-list
            // membership tests and the atomics run on full waves instead of once per (tile, row) combination with a
            // handful of live lanes -- past the first iteration only a few percent of the pairs get this far.
            int pid4[4];
            float pn4[4], pth4[4];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                pid4[r] = cid[4 * gq + r];
                pn4[r] = cnrm[4 * gq + r];
                pth4[r] = cth[4 * gq + r];
            }
            int qn = 0;
            auto drain = [&]() __attribute__((always_inline)) {
                nnd_wave_lds_sync();
                for (int base = 0; base < qn; base += 64) {
                    const int t = base + lane;
                    if (t < qn) {
                        const uint2 en = queue[t];
                        const int a = en.x & 31, b = (en.x >> 5) & 31;
                        const float d = __uint_as_float(en.y);
                        const int pid = cid[a], qid = cid[b];
                        if ((en.x & 1024u) && !klist_has<KS16>(klist + a * kls, (uint32_t)qid)) {  // p <- q
#ifndef NND_JOIN_NOATOMIC  // timing experiments only
                            {
                                unsigned long long *slot = prop_slot(pid, cslot[b]);
                                const unsigned long long key = (unsigned long long)nnd_make_key(d, (uint32_t)qid);
                                atomicMin(slot, key);
                            }
#endif
                            cflag[a] = 1;
                            tot_prop++;
                        }
                        if ((en.x & 2048u) && !klist_has<KS16>(klist + b * kls, (uint32_t)pid)) {  // q <- p
#ifndef NND_JOIN_NOATOMIC
                            {
                                unsigned long long *slot = prop_slot(qid, cslot[a]);
                                const unsigned long long key = (unsigned long long)nnd_make_key(d, (uint32_t)pid);
                                atomicMin(slot, key);
                            }
#endif
                            cflag[b] = 1;
                            tot_prop++;
                        }
                    }
                }
                qn = 0;
            };
#pragma unroll
            for (int J = 0; J < 2; J++) {
                if (J == 1 && !has_old) break;  // first iteration: there are no old candidates at all
                if (QCAP < 8 * 64 && J == 1) {  // small queue: the first tile's pairs leave before the second tile's arrive
                    drain();
                    nnd_wave_lds_sync();
                }
                const int jj = J * 16 + r16;  // index inside [new | old]
                const int qid = cid[jj];
                const float qn_ = cnrm[jj], qth = cth[jj];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int i = 4 * gq + r;  // index inside new
                    const int pid = pid4[r];
                    const bool valid = pid >= 0 && qid >= 0 && (jj >= MCP || jj >= i);
                    tot_pairs += valid ? 1 : 0;
                    const bool self = (pid == qid);
                    const float d = self ? 0.0f : nnd_gram_to_dist(metric, acc[J][r], pn4[r], qn_);
#ifdef NND_JOIN_NOEPI
                    const bool need_p = valid && d == -12345.0f, need_q = false;
#else
                    const bool need_p = valid && d < pth4[r], need_q = valid && !self && d < qth;
#endif
                    const unsigned long long pm = __ballot(need_p | need_q);
                    if (pm) {  // wave-uniform
                        if (need_p | need_q) {
                            const int off = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                            queue[off] = make_uint2((uint32_t)i | ((uint32_t)jj << 5) | (need_p ? 1024u : 0u) | (need_q ? 2048u : 0u),
                                                    __float_as_uint(d));
                        }
                        qn += __popcll(pm);
                    }
                }
            }
            drain();
            nnd_wave_lds_sync();
            if (lane < RV && cflag[lane]) pdirty[cid[lane]] = 1;
            if (lane < RV) tot_rows += cid[lane] >= 0;
            if (lane == 0) tot_act += 1;
        }
        nnd_wave_lds_sync();
        store_cand(cur, c2);
    }
    // ---- statistics: one atomic per workgroup per counter (striped) ----
    tot_pairs = nnd_wave_sum_i32(tot_pairs);
    tot_prop = nnd_wave_sum_i32(tot_prop);
    tot_rows = nnd_wave_sum_i32(tot_rows);
    tot_act = nnd_wave_sum_i32(tot_act);
    __syncthreads();
    int *red = (int *)smem;  // every wave is done with its region
    if (lane == 0) {
        red[w * 5 + 0] = tot_pairs; red[w * 5 + 1] = tot_prop; red[w * 5 + 2] = tot_rows; red[w * 5 + 3] = tot_act;
        red[w * 5 + 4] = tot_tiles;
    }
    __syncthreads();
    if (tid < 5) {
        long long sum = (long long)red[tid] + red[5 + tid] + red[10 + tid] + red[15 + tid];
        if (tid == 4) sum *= (dp >> 2);  // tiles -> v_mfma_f32_16x16x4 instructions (4 k-values each)
        const int which = tid == 0 ? CNT_PAIRS : (tid == 1 ? CNT_PROPOSALS : (tid == 2 ? CNT_ROWS : (tid == 3 ? CNT_ACTIVE : CNT_MFMA)));
        nnd_count(counters, which, sum);
    }
}

// visiting order of a launch over [v_begin, v_end): the whole-set spatial order, or -- on a shard -- the owned vertices in
// the order shard.hip derived from the rank's first tree (positions [v_begin - own_lo, v_end - own_lo) of that list)
static const int32_t *join_order(const nnd_ctx *ctx, int64_t &v_begin, int64_t &v_end) {
    const int32_t *order = nnd_vertex_order(ctx);
    if (!order && ctx->own_order && v_begin >= ctx->own_lo && v_end <= ctx->own_hi) {
        order = ctx->own_order;
        v_begin -= ctx->own_lo;
        v_end -= ctx->own_lo;
    }
    return order;
}

template <int DC, int KS16, bool SHARD>
static int launch_join16_t(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    constexpr int RV = 32, kls = KS16 * 16 + 4;
    constexpr int WAVE_BYTES = NND_J16_QCAP * 8 + 2 * RV * 4 + 4 * 4 + 5 * RV * 4 + RV * kls * 4 + 8;  // = the kernel's
    size_t smem = 4 * (size_t)((WAVE_BYTES + 15) & ~15);
    auto kern = k_local_join16<DC, KS16, SHARD>;
    // function attributes and occupancy are per DEVICE: cached per device ordinal, not per process
    static int wg_per_cu_dev[64] = {0}, n_cu_dev[64] = {0};
    int &wg_per_cu = wg_per_cu_dev[ctx->p.device & 63], &n_cu = n_cu_dev[ctx->p.device & 63];
    if (wg_per_cu == 0) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipDeviceProp_t prop;
        NND_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->p.device));
        n_cu = prop.multiProcessorCount;
        int occ = 0;
        NND_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, smem));
        wg_per_cu = occ < 1 ? 1 : occ;
        if (const char *cap = nnd_knob("NND_J16_WGCAP")) {  // experiments: fewer resident workgroups per CU
            const int c = atoi(cap);
            if (c >= 1 && c < wg_per_cu) wg_per_cu = c;
        }
    }
    int64_t nv = v_end - v_begin;
    int64_t groups = (nv + 3) / 4;
    int64_t resident = (int64_t)n_cu * wg_per_cu;
    unsigned grid = (unsigned)(groups < resident ? groups : resident);
    if (grid > 8) grid &= ~7u;  // whole multiples of the XCD count: every XCD walks its own slice of the order
    uint32_t slot_seed = nnd_hash2(ctx->seed ^ 0x2545F491u, (uint32_t)ctx->iter);
    ctx->pbuf_clean = false;
    const int32_t *order = join_order(ctx, v_begin, v_end);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->cand,
                       order, v_begin, v_end, ctx->k, ctx->ks, ctx->knn_e, ctx->th, ctx->pbuf, ctx->pdirty, ctx->pcap, slot_seed,
                       ctx->counters, nnd_list_lo(ctx), nnd_list_hi(ctx), ctx->pbuf_r, ctx->pcap_r, ctx->own_lo, ctx->own_hi);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int DC>
static int launch_join16_ks(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    if (ctx->pbuf_r) {  // a shard of a row-sharded build: proposals for rows owned elsewhere go to the narrow table
        if (ctx->ks <= 16) return launch_join16_t<DC, 1, true>(ctx, v_begin, v_end);
        if (ctx->ks <= 32) return launch_join16_t<DC, 2, true>(ctx, v_begin, v_end);
        return launch_join16_t<DC, 4, true>(ctx, v_begin, v_end);
    }
    if (ctx->ks <= 16) return launch_join16_t<DC, 1, false>(ctx, v_begin, v_end);
    if (ctx->ks <= 32) return launch_join16_t<DC, 2, false>(ctx, v_begin, v_end);
    return launch_join16_t<DC, 4, false>(ctx, v_begin, v_end);
}

// ------------------------------------------------------------------------------------------------
// max_candidates 17..64 (k = 30, the reference's default, lands here with mcp = 32): the same wave-per-vertex,
// operands-from-global design as k_local_join16, generalised to NA = mcp/16 tile rows of new candidates against
// NB = 2 NA tiles of [new | old].  The neighbour lists are not staged (64-128 rows x k ids would cost the occupancy):
// the membership test of a queued pair reads the target's list straight from global memory (L2) when the queue is
// drained; the queue holds 512 entries and is drained whenever half full.
#ifndef NND_JW_WAVES
#define NND_JW_WAVES 3
#endif
#ifndef NND_JW_DCW
#define NND_JW_DCW 16  // floats per K block, rows of >= 128 floats (round 6: 32 -> 16: no spilled register at three waves per SIMD; k = 30 join 18.4 -> 16.5 ms)
#endif
#ifndef NND_JW_DC
#define NND_JW_DC 16   // ... narrower rows (32 -> 16: 14.7 -> 12.7 ms at d = 64, 16.6 -> 14.6 at d = 96, k = 30)
#endif
#ifndef NND_JW64_DC
#define NND_JW64_DC 32 // ... the 64-slot kernel (max_candidates 33..64 and the blocked passes; 16: the same time)
#endif
// BLOCKED (max_candidates 65..128, candidate lists of [new(128) | old(128)] slots): the MCP = 64 kernel run over 64-slot blocks of
// the lists -- `new` rows from slots [new_off, new_off + 64), `old` rows from [old_off, old_off + 64) of a vertex's cstride slots;
// SKIP_TRI leaves the new x new triangle out (the passes that meet a `new` block a second time, and the pass whose "old" block is
// the second block of new candidates: every pair of the reference's join exactly once, pynndescent_.py:228-258).
// ASEL >= 0 (round 6): the launch computes ONE tile row of new candidates (rows [16 ASEL, 16 ASEL + 16)) against its column tiles
// and is followed by a launch for the other row(s): 4 + 3 tiles instead of 7 per wave -- 16 accumulator and 32 operand registers
// a lane instead of 32 + 64 -- so that more waves are resident; a vertex with no candidate in the tile row is passed over.
#ifndef NND_JW_WAVES_SPLIT
#define NND_JW_WAVES_SPLIT 4
#endif
// KL (round 6; MCP = 32, rows of <= 32 neighbours, one GPU): the neighbour lists of the vertex's 64 candidates ARE staged -- 8 KB a
// wave, written by LDS-DMA (global_load_lds: no register on the way) when the vertex's ids land, i.e. before its MFMA phase, and
// read by the membership tests of the drains after it.  The queue shrinks to 384 entries (a drain still takes full waves) so
// that three workgroups keep fitting a CU: 12.8 KB a wave (the launcher checks the occupancy and falls back to the unstaged form).  Without it a drain step waits for two dependent list reads from L2.
#ifndef NND_JW_KL_QCAP
#define NND_JW_KL_QCAP 384  // 320: 18.9 ms, 384: 18.4, 416: 18.3 (k = 30 join at 1 M points); 448 no longer fits three workgroups a CU: 30 ms
#endif
template <bool KL>
struct join_w_lds {
    static constexpr int QCAP = KL ? NND_JW_KL_QCAP : 512;
    static constexpr int KL_BYTES = KL ? 64 * 128 : 0;
    static constexpr int wave_bytes(int RV) { return KL_BYTES + QCAP * 8 + 2 * RV * 4 + 2 * 4 + 5 * RV * 4 + 8; }
};
__device__ __forceinline__ void nnd_glds16(const void *src, void *lds_dst_wave_uniform) {  // 16 bytes per lane: LDS[dst + 16 lane] <- *src
    const uint32_t off = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)lds_dst_wave_uniform);
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(uintptr_t)src,
                                     (__attribute__((address_space(3))) void *)off, 16, 0, 0);
}
template <int MCP, int DC, bool SHARD, bool BLOCKED = false, bool SKIP_TRI = false, int ASEL = -1, bool KL = false>
__global__ __launch_bounds__(256, MCP == 32 ? (ASEL >= 0 ? NND_JW_WAVES_SPLIT : NND_JW_WAVES) : 2) void k_local_join_w(const float *__restrict__ xp, int dp,
                                                                       const float *__restrict__ nrm, int metric,
                                                                       const int32_t *__restrict__ cand,
                                                                       const int32_t *__restrict__ order, int64_t v_begin,
                                                                       int64_t v_end, int k, int ks,
                                                                       const uint32_t *__restrict__ knn_e,
                                                                       const float *__restrict__ th, uint64_t *__restrict__ pbuf,
                                                                       uint8_t *__restrict__ pdirty, int pcap, uint32_t slot_seed,
                                                                       long long *__restrict__ counters, int64_t own_lo, int64_t own_hi,
                                                                       uint64_t *__restrict__ pbuf_r, int pcap_r, int64_t rt_lo, int64_t rt_hi,
                                                                       int cstride, int new_off, int old_off) {
    static_assert(!BLOCKED || MCP == 64, "the blocked passes run the 64-slot kernel");
    constexpr int NA = MCP / 16, NB = 2 * NA, RV = 2 * MCP;
    constexpr int NT = DC / 16;                       // 16-byte chunks per lane, row and K block
    constexpr int RPL = RV / 64;                      // candidate slots per lane (1 or 2)
    static_assert(!KL || (MCP == 32 && !SHARD && !BLOCKED && ASEL < 0), "staged neighbour lists: the plain 32-slot kernel only");
    constexpr int QCAP = join_w_lds<KL>::QCAP;
    constexpr int WAVE_BYTES = join_w_lds<KL>::wave_bytes(RV);

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    unsigned char *mine = smem + (size_t)w * ((WAVE_BYTES + 15) & ~15);
    uint32_t *klist = (uint32_t *)mine;                 // KL: 64 rows x 32 words (row r at r * 128 bytes; rows of 16 words use the first half)
    uint2 *queue = (uint2 *)(mine + join_w_lds<KL>::KL_BYTES);  // QCAP
    int32_t *cidbuf = (int32_t *)(queue + QCAP);        // 2 * RV
    int32_t *nnewbuf = cidbuf + 2 * RV;                 // 2
    int32_t *cid = nnewbuf + 2;                         // RV
    float *cnrm = (float *)(cid + RV);                  // RV
    float *cth = cnrm + RV;                             // RV
    uint32_t *cslot = (uint32_t *)(cth + RV);           // RV
    uint32_t *cflag = cslot + RV;                       // RV
    const int kq = ks >> 2;
    // [own_lo, own_hi): the rows whose CURRENT neighbour lists this handle holds (membership tests); [rt_lo, rt_hi): the
    // rows it owns.  Proposal slot of target t: a shard keeps proposals for vertices owned ELSEWHERE in a narrow table
    // (pcap_r slots per row: a rank sends a remote row a few proposals per iteration, and the export streams that table)
    auto prop_slot = [&](int t, uint32_t slot) __attribute__((always_inline)) -> unsigned long long * {
        if (SHARD && ((int64_t)t < rt_lo || (int64_t)t >= rt_hi))  // (SHARD = false: the plain build carries none of this)
            return (unsigned long long *)&pbuf_r[(int64_t)t * pcap_r + (slot & (uint32_t)(pcap_r - 1))];
        return (unsigned long long *)&pbuf[(int64_t)t * pcap + slot];
    };
    const int r16 = lane & 15, gq = lane >> 4;
    int64_t n_v = v_end - v_begin;
    int64_t g, stride;
    if ((gridDim.x & 7) == 0) {  // one contiguous eighth of the visiting order per XCD (see k_local_join16)
        const int64_t per = (((n_v + 7) >> 3) + 3) & ~(int64_t)3;
        const int64_t g0 = (int64_t)(blockIdx.x & 7) * per;
        n_v = n_v < g0 + per ? n_v : g0 + per;
        g = g0 + (int64_t)(blockIdx.x >> 3) * 4 + w;
        stride = (int64_t)(gridDim.x >> 3) * 4;
    } else {
        g = (int64_t)blockIdx.x * 4 + w;
        stride = (int64_t)gridDim.x * 4;
    }

    // candidate ids of vertex g: slot lane + 64*u of [new(MCP) | old(MCP)]
    auto load_cand = [&](int64_t g, int (&c)[RPL]) __attribute__((always_inline)) {
        const bool ok = g < n_v;
        const int64_t gg = ok ? g : 0;  // idle lanes read the launch's first vertex: on a shard the table holds the owned rows only
        const int64_t v = order ? (int64_t)order[v_begin + gg] : v_begin + gg;
#pragma unroll
        for (int u = 0; u < RPL; u++) {
            const int cc = BLOCKED ? cand[v * cstride + (u == 0 ? new_off : old_off) + lane] : cand[v * RV + lane + 64 * u];
            c[u] = ok ? cc : -1;
        }
    };
    auto store_cand = [&](int buf, const int (&c)[RPL]) __attribute__((always_inline)) {
        int nn = 0, no = 0;
#pragma unroll
        for (int u = 0; u < RPL; u++) {
            const unsigned long long m = __ballot(c[u] >= 0);
            cidbuf[buf * RV + lane + 64 * u] = c[u];
            if (RPL == 1) {  // RV == 64: lanes [0,32) new, [32,64) old
                nn += __popcll(m & 0xFFFFFFFFull);
                no += __popcll(m >> 32);
            } else {  // RV == 128: u == 0 new, u == 1 old
                if (u == 0) nn += __popcll(m); else no += __popcll(m);
            }
        }
        if (lane == 0) nnewbuf[buf] = nn | (no << 8);
    };
    f32x4 rt[NB][NT];   // this lane's chunks of its row in every tile (one K block)
    f32x4 rt2[NB][NT];  // ... of the NEXT K block (round 6: requested before the MFMAs of the current one, not after them)
    float nx_nrm[RPL], nx_th[RPL];
    int nx_id[RPL];
    auto tile_live = [&](int t, int nn, int no) __attribute__((always_inline)) -> bool {
        return t < NA ? 16 * t < nn : 16 * (t - NA) < no;
    };
    constexpr int A_LO = ASEL >= 0 ? ASEL : 0;        // first tile row of new candidates this launch computes
    constexpr int NEW_MIN = 16 * A_LO;                // a vertex joins in this launch when it has more new candidates than this
    auto tile_needed = [&](int t) __attribute__((always_inline)) -> bool {  // as an operand of this launch
        return ASEL < 0 || t == ASEL || t >= (SKIP_TRI ? NA : ASEL);
    };
    // rows of K block [c0, c0 + cw) of the vertex whose ids are in cb[]; empty slots read row 0 (cache hit, masked later)
    auto load_rows_into = [&](f32x4 (&dst)[NB][NT], const int32_t *cb, int c0, int cw, int nn, int no) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NB; t++) {
            if (!tile_needed(t) || !tile_live(t, nn, no)) continue;
            const int id = cb[16 * t + r16];
            const float *pr = xp + (int64_t)(id >= 0 ? id : 0) * dp + c0 + 4 * gq;
#pragma unroll
            for (int j = 0; j < NT; j++)
                if (16 * j < cw) dst[t][j] = *(const f32x4 *)(pr + 16 * j);
        }
    };
    auto load_rows = [&](const int32_t *cb, int c0, int cw, int nn, int no) __attribute__((always_inline)) { load_rows_into(rt, cb, c0, cw, nn, no); };
    auto issue_gather = [&](int buf) __attribute__((always_inline)) {
        const int32_t *cb = cidbuf + buf * RV;
        const int cnt = nnewbuf[buf], nn = cnt & 255, no = cnt >> 8;
        if (nn <= NEW_MIN) return;  // wave-uniform: no new candidate (in this launch's tile row), no join (utils.py:611-613)
        load_rows(cb, 0, dp < DC ? dp : DC, nn, no);
#pragma unroll
        for (int u = 0; u < RPL; u++) {
            nx_id[u] = cb[lane + 64 * u];
            const int64_t ide = nx_id[u] >= 0 ? nx_id[u] : 0;
            nx_nrm[u] = nrm[ide];
            nx_th[u] = th[ide];
        }
    };
    auto land_gather = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < RPL; u++) {
            const int r = lane + 64 * u;
            cid[r] = nx_id[u];
            cnrm[r] = nx_nrm[u];
            cth[r] = nx_th[u];
            cslot[r] = nnd_hash2(slot_seed, (uint32_t)nx_id[u]) & (uint32_t)(pcap - 1);
            cflag[r] = 0;
        }
    };
    // is `id` among the neighbour ids of vertex `row` (global memory; the list sits in L2 more often than not)
    auto list_has = [&](int row, uint32_t id) __attribute__((always_inline)) -> bool {
        if ((int64_t)row < own_lo || (int64_t)row >= own_hi) return false;  // owned elsewhere: its owner dedups (row-sharded build)
        const u32x4 *kl = (const u32x4 *)(knn_e + (int64_t)row * ks);
        bool present = false;
        for (int c = 0; c < kq; c += 4) {
            u32x4 wv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wv[j] = kl[c + j < kq ? c + j : c];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32x4 m = wv[j] & NND_IDX_MASK;
                present |= (c + j < kq) && ((m.x == id) | (m.y == id) | (m.z == id) | (m.w == id));
            }
        }
        return present;
    };
    // KL: the lists of the candidates in slots [0, nn) and [MCP, MCP + no), requested when the ids have landed: instruction i
    // brings rows 8 i .. 8 i + 7 (lane = 8 * row + chunk), 1 KB of LDS each; lanes of empty slots / chunks beyond the row stay off
    auto issue_lists = [&](int nn, int no) __attribute__((always_inline)) {
#pragma unroll 1
        for (int i = 0; i < 8; i++) {  // (rolled: eight address pairs at once would be spilled -- the operand registers are live here)
            if (!(8 * i < MCP ? 8 * i < nn : 8 * i - MCP < no)) continue;  // wave-uniform
            const int row = 8 * i + (lane >> 3), c = lane & 7;
            const int id = cid[row];
            if (id >= 0 && c < kq) nnd_glds16(knn_e + (int64_t)id * ks + 4 * c, klist + i * 256);
        }
    };
    auto list_has_lds = [&](int row, uint32_t id) __attribute__((always_inline)) -> bool {
        const u32x4 *kl = (const u32x4 *)(klist + row * 32);
        bool present = false;
        for (int h = 0; h < kq; h += 4) {  // four chunks at a time (registers); rows start on different chunks (bank conflicts)
            u32x4 wv[4];
#pragma unroll
            for (int j = 0; j < 4; j++) wv[j] = kl[h + ((j + row) & 3)];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32x4 m = wv[j] & NND_IDX_MASK;
                present |= (m.x == id) | (m.y == id) | (m.z == id) | (m.w == id);
            }
        }
        return present;
    };
    int tot_pairs = 0, tot_prop = 0, tot_rows = 0, tot_act = 0, tot_tiles = 0;
    int qn = 0;
    auto drain = [&]() __attribute__((always_inline)) {
        nnd_wave_lds_sync();
        for (int base = 0; base < qn; base += 64) {
            const int t = base + lane;
            if (t < qn) {
                const uint2 en = queue[t];
                const int a = en.x & 63, b = (en.x >> 6) & 127;
                const float d = __uint_as_float(en.y);
                const int pid = cid[a], qid = cid[b];
                if ((en.x & (1u << 13)) && !(KL ? list_has_lds(a, (uint32_t)qid) : list_has(pid, (uint32_t)qid))) {  // p <- q
                    atomicMin(prop_slot(pid, cslot[b]), (unsigned long long)nnd_make_key(d, (uint32_t)qid));
                    cflag[a] = 1;
                    tot_prop++;
                }
                if ((en.x & (1u << 14)) && !(KL ? list_has_lds(b, (uint32_t)pid) : list_has(qid, (uint32_t)pid))) {  // q <- p
                    atomicMin(prop_slot(qid, cslot[a]), (unsigned long long)nnd_make_key(d, (uint32_t)pid));
                    cflag[b] = 1;
                    tot_prop++;
                }
            }
        }
        nnd_wave_lds_sync();
        qn = 0;
    };

    {
        int c0[RPL], c1[RPL];
        load_cand(g, c0);
        load_cand(g + stride, c1);
        store_cand(0, c0);
        store_cand(1, c1);
    }
    nnd_wave_lds_sync();
    issue_gather(0);

    for (int it = 0; g < n_v; g += stride, it++) {
        const int cur = it & 1;
        const int my_cnt = nnewbuf[cur], nn = my_cnt & 255, no = my_cnt >> 8;
        int c2[RPL];
        load_cand(g + 2 * stride, c2);
        f32x4 acc[NA][NB];
#pragma unroll
        for (int a = 0; a < NA; a++)
#pragma unroll
            for (int b = 0; b < NB; b++) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (nn > NEW_MIN) {
            land_gather();
            nnd_wave_lds_sync();
            if constexpr (KL) issue_lists(nn, no);  // in flight during the MFMA phase
            // K blocks, double-buffered in registers: the rows of block c + 1 are requested BEFORE the MFMAs of block c (rounds 1-5
            // fetched them "in turn": three exposed gather latencies per vertex at d = 128 with 32-float blocks, and a wave is
            // one of twelve on its CU)
            auto gram_block = [&](const f32x4 (&r)[NB][NT], int c0, int cw) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < NA; a++) {
                    if ((ASEL >= 0 && a != ASEL) || !tile_live(a, nn, no)) continue;
#pragma unroll
                    for (int b = SKIP_TRI ? NA : a; b < NB; b++) {  // new x new from the diagonal tile up, then new x old
                        if (!tile_live(b, nn, no)) continue;
                        if (c0 == 0) tot_tiles++;  // wave-uniform
#pragma unroll
                        for (int j = 0; j < NT; j++) {
                            if (16 * j >= cw) continue;
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[a][j].x, r[b][j].x, acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[a][j].y, r[b][j].y, acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[a][j].z, r[b][j].z, acc[a][b], 0, 0, 0);
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(r[a][j].w, r[b][j].w, acc[a][b], 0, 0, 0);
                        }
                    }
                }
            };
            for (int c0 = 0; c0 < dp; c0 += 2 * DC) {
                const int cw = (dp - c0) < DC ? (dp - c0) : DC;
                const int c1 = c0 + DC, cw1 = (dp - c1) < DC ? (dp - c1) : DC;
#ifndef NND_JW_NO_PIPELINE
                if constexpr (ASEL >= 0) {  // one register set: the point of the split launches is the register count
                    if (c0 > 0) load_rows_into(rt, cid, c0, cw, nn, no);
                    gram_block(rt, c0, cw);
                    if (c1 < dp) {
                        load_rows_into(rt, cid, c1, cw1, nn, no);
                        gram_block(rt, c1, cw1);
                    }
                    continue;
                }
                if (c1 < dp) load_rows_into(rt2, cid, c1, cw1, nn, no);
                gram_block(rt, c0, cw);
                if (c1 < dp) {
                    if (c1 + DC < dp) load_rows_into(rt, cid, c1 + DC, (dp - c1 - DC) < DC ? (dp - c1 - DC) : DC, nn, no);
                    gram_block(rt2, c1, cw1);
                }
#else
                gram_block(rt, c0, cw);
                if (c1 < dp) {
                    load_rows_into(rt2, cid, c1, cw1, nn, no);
                    gram_block(rt2, c1, cw1);
                    if (c1 + DC < dp) load_rows_into(rt, cid, c1 + DC, (dp - c1 - DC) < DC ? (dp - c1 - DC) : DC, nn, no);
                }
#endif
            }
        }
        // the row registers are free: the gather of this wave's next vertex flies during the epilogue
        if constexpr (KL) {
            // the lists were requested before the MFMA phase: they have landed (or nearly); waiting HERE, before the next vertex's
            // gather is issued, keeps that gather out of the wait
            if (nn > NEW_MIN) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (g + stride < n_v) issue_gather(cur ^ 1);
        if (nn > NEW_MIN) {
#pragma unroll
            for (int a = 0; a < NA; a++) {
                if ((ASEL >= 0 && a != ASEL) || !tile_live(a, nn, no)) continue;
                int pid4[4];
                float pn4[4], pth4[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    pid4[r] = cid[16 * a + 4 * gq + r];
                    pn4[r] = cnrm[16 * a + 4 * gq + r];
                    pth4[r] = cth[16 * a + 4 * gq + r];
                }
#pragma unroll
                for (int b = SKIP_TRI ? NA : a; b < NB; b++) {
                    if (!tile_live(b, nn, no)) continue;
                    const int jj = 16 * b + r16;  // index inside [new | old]
                    const int qid = cid[jj];
                    const float qn_ = cnrm[jj], qth = cth[jj];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int i = 16 * a + 4 * gq + r;  // index inside new
                        const int pid = pid4[r];
                        const bool valid = pid >= 0 && qid >= 0 && (jj >= MCP || jj >= i);
                        tot_pairs += valid ? 1 : 0;
                        const bool self = (pid == qid);
                        const float d = self ? 0.0f : nnd_gram_to_dist(metric, acc[a][b][r], pn4[r], qn_);
                        const bool need_p = valid && d < pth4[r], need_q = valid && !self && d < qth;
                        const unsigned long long pm = __ballot(need_p | need_q);
                        if (pm) {  // wave-uniform
                            if (need_p | need_q) {
                                const int off = qn + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
                                queue[off] = make_uint2((uint32_t)i | ((uint32_t)jj << 6) | (need_p ? (1u << 13) : 0u) | (need_q ? (1u << 14) : 0u),
                                                        __float_as_uint(d));
                            }
                            qn += __popcll(pm);
                        }
                    }
                    if (qn > QCAP - 256) drain();  // room for one more tile (4 x 64 entries)
                }
            }
            if (qn > 0) drain();
            nnd_wave_lds_sync();
#pragma unroll
            for (int u = 0; u < RPL; u++) {
                const int r = lane + 64 * u;
                if (cflag[r]) pdirty[cid[r]] = 1;
                if (ASEL <= 0) tot_rows += cid[r] >= 0;  // (a vertex's rows and the vertex itself are counted by its first launch)
            }
            if (ASEL <= 0 && lane == 0) tot_act += 1;
        }
        nnd_wave_lds_sync();
        store_cand(cur, c2);
    }
    tot_pairs = nnd_wave_sum_i32(tot_pairs);
    tot_prop = nnd_wave_sum_i32(tot_prop);
    tot_rows = nnd_wave_sum_i32(tot_rows);
    tot_act = nnd_wave_sum_i32(tot_act);
    __syncthreads();
    int *red = (int *)smem;  // every wave is done with its region
    if (lane == 0) {
        red[w * 5 + 0] = tot_pairs; red[w * 5 + 1] = tot_prop; red[w * 5 + 2] = tot_rows; red[w * 5 + 3] = tot_act;
        red[w * 5 + 4] = tot_tiles;
    }
    __syncthreads();
    if (tid < 5) {
        long long sum = (long long)red[tid] + red[5 + tid] + red[10 + tid] + red[15 + tid];
        if (tid == 4) sum *= (dp >> 2);  // tiles -> v_mfma_f32_16x16x4 instructions
        const int which = tid == 0 ? CNT_PAIRS : (tid == 1 ? CNT_PROPOSALS : (tid == 2 ? CNT_ROWS : (tid == 3 ? CNT_ACTIVE : CNT_MFMA)));
        nnd_count(counters, which, sum);
    }
}

template <int MCP, int DC, bool SHARD, bool BLOCKED = false, bool SKIP_TRI = false, int ASEL = -1, bool KL = false>
static int launch_join_w_t(nnd_ctx *ctx, int64_t v_begin, int64_t v_end, int cstride = 0, int new_off = 0, int old_off = 0) {
    constexpr int RV = 2 * MCP;
    constexpr int WAVE_BYTES = join_w_lds<KL>::wave_bytes(RV);
    size_t smem = 4 * (size_t)((WAVE_BYTES + 15) & ~15);
    auto kern = k_local_join_w<MCP, DC, SHARD, BLOCKED, SKIP_TRI, ASEL, KL>;
    // function attributes and occupancy are per DEVICE: cached per device ordinal, not per process
    static int wg_per_cu_dev[64] = {0}, n_cu_dev[64] = {0};
    int &wg_per_cu = wg_per_cu_dev[ctx->p.device & 63], &n_cu = n_cu_dev[ctx->p.device & 63];
    if (wg_per_cu == 0) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipDeviceProp_t prop;
        NND_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->p.device));
        n_cu = prop.multiProcessorCount;
        int occ = 0;
        NND_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, smem));
        wg_per_cu = occ < 1 ? 1 : occ;
    }
    int64_t nv = v_end - v_begin;
    int64_t groups = (nv + 3) / 4;
    int64_t resident = (int64_t)n_cu * wg_per_cu;
    unsigned grid = (unsigned)(groups < resident ? groups : resident);
    if (grid > 8) grid &= ~7u;
    uint32_t slot_seed = nnd_hash2(ctx->seed ^ 0x2545F491u, (uint32_t)ctx->iter);
    ctx->pbuf_clean = false;
    const int32_t *order = join_order(ctx, v_begin, v_end);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->cand,
                       order, v_begin, v_end, ctx->k, ctx->ks, ctx->knn_e, ctx->th, ctx->pbuf, ctx->pdirty,
                       ctx->pcap, slot_seed, ctx->counters, nnd_list_lo(ctx), nnd_list_hi(ctx), ctx->pbuf_r, ctx->pcap_r, ctx->own_lo, ctx->own_hi,
                       cstride, new_off, old_off);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int MCP, int DC>
static int launch_join_w(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
#ifdef NND_JW_SPLIT  // two launches, one tile row of new candidates each (k_local_join_w, ASEL)
    if (MCP == 32) {
        if (ctx->pbuf_r) {
            if (launch_join_w_t<MCP, DC, true, false, false, 0>(ctx, v_begin, v_end)) return 1;
            return launch_join_w_t<MCP, DC, true, false, false, 1>(ctx, v_begin, v_end);
        }
        if (launch_join_w_t<MCP, DC, false, false, false, 0>(ctx, v_begin, v_end)) return 1;
        return launch_join_w_t<MCP, DC, false, false, false, 1>(ctx, v_begin, v_end);
    }
#endif
#ifndef NND_JW_NO_LDS_LISTS
    if constexpr (MCP == 32) {  // rows of <= 32 neighbours on one GPU: the candidates' neighbour lists are staged in LDS (k_local_join_w, KL)
        if (!ctx->pbuf_r && ctx->ks <= 32 && nnd_list_lo(ctx) <= 0 && nnd_list_hi(ctx) >= ctx->n && !(ctx->p.flags & NND_FLAG_TEST_JOIN_UNSTAGED)) {
            // ... as long as three workgroups of it fit a CU (LDS): with two the kernel is 60 % slower than the unstaged form
            static int kl_ok_dev[64] = {0};  // 0 unknown, 1 yes, -1 no (per device ordinal)
            int &kl_ok = kl_ok_dev[ctx->p.device & 63];
            if (kl_ok == 0) {
                constexpr size_t smem = 4 * (size_t)((join_w_lds<true>::wave_bytes(2 * MCP) + 15) & ~15);
                auto kern = k_local_join_w<MCP, DC, false, false, false, -1, true>;
                int occ = 0;
                if (hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
                    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)kern, 256, smem) != hipSuccess) {
                    (void)hipGetLastError();
                    occ = 0;
                }
                kl_ok = occ >= 3 ? 1 : -1;
            }
            if (kl_ok > 0) return launch_join_w_t<MCP, DC, false, false, false, -1, true>(ctx, v_begin, v_end);
        }
    }
#endif
    return ctx->pbuf_r ? launch_join_w_t<MCP, DC, true>(ctx, v_begin, v_end) : launch_join_w_t<MCP, DC, false>(ctx, v_begin, v_end);
}

// max_candidates 65..128: candidate lists [newA(64) newB(64) | oldA(64) oldB(64)] (filled from the front: the B blocks are empty
// unless a class has more than 64 candidates, and a pass whose `new` block is empty does nothing).  Five passes of the 64-slot
// kernel cover every pair once: A x A + A x oldA, A x oldB, B x B + B x oldA, B x oldB, A x B.
template <bool SHARD>
static int launch_join_blocked(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    if (launch_join_w_t<64, 32, SHARD, true, false>(ctx, v_begin, v_end, 256, 0, 128)) return 1;
    if (launch_join_w_t<64, 32, SHARD, true, true>(ctx, v_begin, v_end, 256, 0, 192)) return 1;
    if (launch_join_w_t<64, 32, SHARD, true, false>(ctx, v_begin, v_end, 256, 64, 128)) return 1;
    if (launch_join_w_t<64, 32, SHARD, true, true>(ctx, v_begin, v_end, 256, 64, 192)) return 1;
    return launch_join_w_t<64, 32, SHARD, true, true>(ctx, v_begin, v_end, 256, 0, 64);
}

int nnd_launch_join(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    if (v_end <= v_begin) return 0;
    const bool wide = ctx->dp >= 128;
    switch (ctx->mcp) {
#ifndef NND_J16_DCW
#define NND_J16_DCW 64
#endif
        case 16: return wide ? launch_join16_ks<NND_J16_DCW>(ctx, v_begin, v_end) : launch_join16_ks<32>(ctx, v_begin, v_end);
        case 32: return wide ? launch_join_w<32, NND_JW_DCW>(ctx, v_begin, v_end) : launch_join_w<32, NND_JW_DC>(ctx, v_begin, v_end);
        case 64: return launch_join_w<64, NND_JW64_DC>(ctx, v_begin, v_end);
        case 128: return ctx->pbuf_r ? launch_join_blocked<true>(ctx, v_begin, v_end) : launch_join_blocked<false>(ctx, v_begin, v_end);
    }
    ctx->set_error("unsupported padded max_candidates %d", ctx->mcp);
    return 1;
}
