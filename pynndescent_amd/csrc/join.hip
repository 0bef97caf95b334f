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
// MI355X design
//   * a workgroup of 4 waves joins 64/MCP vertices (MCP = max_candidates padded to 16/32/64):
//     128 candidate rows are gathered per workgroup -- [new | old] per vertex -- into swizzled LDS
//     with 16-byte-per-lane coalesced loads of whole 128-byte lines; this gather is the HBM term
//     that bounds the kernel (C_i * row bytes).  All global loads of a
//     workgroup (rows, neighbour lists, norms, thresholds) are issued back to back before the first
//     LDS write, so one memory round trip covers the whole gather;
//   * the new x (new U old) distance block is a Gram contraction on the f32 MFMA pipe
//     (16x16x4, one A tile row per wave, tiles below the diagonal of new x new skipped);
//   * each endpoint is tested on its OWN threshold (d < th_p for p, d < th_q for q) and against the
//     target's current neighbour ids, which are gathered next to the vectors (k*4 bytes per
//     candidate): a proposal leaves the workgroup only if the push would succeed on the snapshot;
//   * surviving proposals go to a per-target bank of PCAP hashed slots with a 64-bit atomicMin on
//     (dist_bits << 32 | source): order independent (deterministic), duplicate proposals of the
//     same source collapse into one slot, and a slot collision keeps the nearer source.
#include "common.h"
#include "gram.h"
#include "state.h"

#define JOIN_ROWS 128  // candidate rows per workgroup

__device__ __forceinline__ bool klist_has(const uint32_t *kl, int ks, uint32_t id) {
    bool present = false;
    for (int c = 0; c < ks; c += 4) {
        const uint4 w = *(const uint4 *)(kl + c);
        present |= (w.x == id) | (w.y == id) | (w.z == id) | (w.w == id);
    }
    return present;
}

template <int MCP, int DC>
__global__ __launch_bounds__(256) void k_local_join(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                    int metric, const int32_t *__restrict__ cand, int64_t v_begin,
                                                    int64_t v_end, int k, int ks, const uint32_t *__restrict__ knn_e,
                                                    const float *__restrict__ knn_d, uint64_t *__restrict__ pbuf,
                                                    uint8_t *__restrict__ pdirty, int pcap, uint32_t slot_seed,
                                                    long long *__restrict__ counters) {
    constexpr int NA = MCP / 16;      // A tile rows per vertex == waves per vertex
    constexpr int NB = 2 * MCP / 16;  // B tiles per vertex: [new | old]
    constexpr int RV = 2 * MCP;       // candidate rows per vertex
    constexpr int VPW = 64 / MCP;     // vertices per workgroup
    constexpr int ROWS = JOIN_ROWS;   // VPW * RV
    constexpr int NCH = DC / 4;       // 16-byte chunks per staged row
    constexpr int NLD = ROWS * NCH / 256;  // row chunks per thread

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *Xs = (float *)smem;                   // ROWS * DC floats
    int32_t *cid = (int32_t *)(Xs + ROWS * DC);  // ROWS
    float *cnrm = (float *)(cid + ROWS);         // ROWS
    float *cth = cnrm + ROWS;                    // ROWS
    int32_t *nnew = (int32_t *)(cth + ROWS);     // 4 (per vertex slot)
    uint32_t *klist = (uint32_t *)(nnew + 4);    // ROWS * kls neighbour ids of every candidate
    const int kls = ks + 4;                      // padded row stride: conflict-free 16-byte reads across rows

    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    const int64_t vbase = v_begin + (int64_t)blockIdx.x * VPW;

    // ---- phase A: candidate ids; new-candidate count per vertex slot ----
    if (tid < ROWS) {
        const int slot = tid / RV, within = tid - slot * RV;
        const int64_t v = vbase + slot;
        const int c = v < v_end ? cand[v * RV + within] : -1;
        cid[tid] = c;
        // candidate lists are filled from the front (sample.hip): count valid new ones per slot
        const unsigned long long m = __ballot(c >= 0 && within < MCP);
        if (RV >= 64) {
            if (within == 0) nnew[slot] = __popcll(m);
        } else {  // RV == 32: two slots per wave
            if (within == 0) nnew[slot] = __popcll(lane < 32 ? (m & 0xFFFFFFFFull) : (m >> 32));
        }
    }
    __syncthreads();
    bool any_active = false;
#pragma unroll
    for (int s = 0; s < VPW; s++) any_active |= nnew[s] > 0;
    if (!any_active) return;  // uniform across the workgroup

    // ---- phase B: every global load of the gather is issued before the first LDS write ----
    float4 rowv[NLD];
    const int cw0 = dp < DC ? dp : DC;
    {
        const int nch0 = cw0 >> 2;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * 256;
            int r, ch;
            if (cw0 == DC) { r = idx / NCH; ch = idx % NCH; } else { r = idx / nch0; ch = idx - r * nch0; }
            const int id = (r < ROWS && nnew[r / RV] > 0) ? cid[r] : -1;
            rowv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (id >= 0) rowv[i] = *(const float4 *)(xp + (int64_t)id * dp + 4 * ch);
        }
    }
    float my_nrm = 0.0f, my_th = 0.0f;
    int my_id = -1;
    if (tid < ROWS) {
        my_id = nnew[tid / RV] > 0 ? cid[tid] : -1;
        if (my_id >= 0) {
            my_nrm = nrm[my_id];
            my_th = knn_d[(int64_t)my_id * ks + (k - 1)];
        }
    }
    const int kq = ks >> 2;  // uint4 chunks per neighbour-list row
    for (int idx = tid; idx < ROWS * kq; idx += 256) {
        const int r = idx / kq, c = idx - r * kq;
        const int id = nnew[r / RV] > 0 ? cid[r] : -1;
        uint4 wv = make_uint4(NND_IDX_MASK, NND_IDX_MASK, NND_IDX_MASK, NND_IDX_MASK);
        if (id >= 0) {
            wv = *(const uint4 *)(knn_e + (int64_t)id * ks + 4 * c);
            wv.x &= NND_IDX_MASK; wv.y &= NND_IDX_MASK; wv.z &= NND_IDX_MASK; wv.w &= NND_IDX_MASK;
            // entries beyond k (row padding) are EMPTY already
        }
        *(uint4 *)(klist + r * kls + 4 * c) = wv;
    }
    if (tid < ROWS) {
        cnrm[tid] = my_nrm;
        cth[tid] = my_th;
    }
    {
        const int nch0 = cw0 >> 2;
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * 256;
            int r, ch;
            if (cw0 == DC) { r = idx / NCH; ch = idx % NCH; } else { r = idx / nch0; ch = idx - r * nch0; }
            if (r < ROWS) *(float4 *)&Xs[nnd_swz<DC>(r, ch)] = rowv[i];
        }
    }
    __syncthreads();
    if (tid < ROWS && my_id < 0) cid[tid] = -1;  // rows of inactive vertices are empty from here on

    const int slot = w / NA, ar = w % NA;
    const int base = slot * RV;
    const int my_new = nnew[slot];
    const int nb_new = (my_new + 15) >> 4;  // occupied new tiles
    f32x4 acc[NB];
#pragma unroll
    for (int J = 0; J < NB; J++) acc[J] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool wave_on = ar < nb_new;  // this wave's A tile row holds at least one new candidate

    for (int c0 = 0; c0 < dp; c0 += DC) {
        const int cw = (dp - c0) < DC ? (dp - c0) : DC;
        if (c0 > 0) {
            nnd_stage_rows<DC>(xp, dp, cid, ROWS, c0, cw, Xs, tid, 256);
            __syncthreads();
        }
        if (wave_on)
            nnd_gram_chunk<DC, NB>(Xs, base + ar * 16, base, cw, acc,
                                   [ar](int J) { return J >= NA || J >= ar; });  // new x new: diagonal and above
        __syncthreads();
    }

    // ---- epilogue: thresholds, dedup against the targets' neighbour lists, proposals ----
    int n_pairs = 0, n_prop = 0;
    if (wave_on) {
        const int r16 = lane & 15, g = lane >> 4;
#pragma unroll
        for (int J = 0; J < NB; J++) {
            if (J < NA && J < ar) continue;
            const int jj = J * 16 + r16;  // index inside [new | old]
            const int qrow = base + jj;
            const int qid = cid[qrow];
            const float qn = cnrm[qrow], qth = cth[qrow];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = ar * 16 + 4 * g + r;  // index inside new
                const int prow = base + i;
                const int pid = cid[prow];
                const bool valid = pid >= 0 && qid >= 0 && (jj >= MCP || jj >= i);
                if (!valid) continue;
                n_pairs++;
                const bool self = (pid == qid);
                const float d = self ? 0.0f : nnd_gram_to_dist(metric, acc[J][r], cnrm[prow], qn);
                if (d < cth[prow] && !klist_has(klist + prow * kls, ks, (uint32_t)qid)) {  // p <- q
                    const uint32_t s = nnd_hash2(slot_seed, (uint32_t)qid) & (uint32_t)(pcap - 1);
                    atomicMin((unsigned long long *)&pbuf[(int64_t)pid * pcap + s],
                              (unsigned long long)nnd_make_key(d, (uint32_t)qid));
                    pdirty[pid] = 1;
                    n_prop++;
                }
                if (!self && d < qth && !klist_has(klist + qrow * kls, ks, (uint32_t)pid)) {  // q <- p
                    const uint32_t s = nnd_hash2(slot_seed, (uint32_t)pid) & (uint32_t)(pcap - 1);
                    atomicMin((unsigned long long *)&pbuf[(int64_t)qid * pcap + s],
                              (unsigned long long)nnd_make_key(d, (uint32_t)pid));
                    pdirty[qid] = 1;
                    n_prop++;
                }
            }
        }
    }
    // statistics: one atomic per workgroup per counter (striped), never one per wave
    n_pairs = nnd_wave_sum_i32(n_pairs);
    n_prop = nnd_wave_sum_i32(n_prop);
    __syncthreads();  // every wave is past its last read of nnew / cid
    int *wstat = (int *)cnrm;  // reuse: cnrm is dead after the epilogue
    if (lane == 0) {
        wstat[2 * w] = n_pairs;
        wstat[2 * w + 1] = n_prop;
    }
    int rows = 0;
    if (w == 0) {
        rows = (lane * 2 < ROWS ? (cid[lane * 2] >= 0) + (cid[lane * 2 + 1] >= 0) : 0);
        rows = nnd_wave_sum_i32(rows);
    }
    __syncthreads();
    if (tid == 0) {
        int act = 0;
        for (int s = 0; s < VPW; s++) act += nnew[s] > 0;
        nnd_count(counters, CNT_PAIRS, (long long)wstat[0] + wstat[2] + wstat[4] + wstat[6]);
        nnd_count(counters, CNT_PROPOSALS, (long long)wstat[1] + wstat[3] + wstat[5] + wstat[7]);
        nnd_count(counters, CNT_ROWS, rows);
        nnd_count(counters, CNT_ACTIVE, act);
    }
}

template <int MCP, int DC>
static int launch_join_t(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    constexpr int VPW = 64 / MCP;
    size_t smem = sizeof(float) * JOIN_ROWS * DC + sizeof(int32_t) * JOIN_ROWS * 3 + 16 +
                  sizeof(uint32_t) * JOIN_ROWS * (size_t)(ctx->ks + 4);
    auto kern = k_local_join<MCP, DC>;
    static size_t configured = 0;
    if (smem > configured) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        configured = smem;
    }
    int64_t nv = v_end - v_begin;
    unsigned grid = (unsigned)((nv + VPW - 1) / VPW);
    uint32_t slot_seed = nnd_hash2(ctx->seed ^ 0x2545F491u, (uint32_t)ctx->iter);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), smem, ctx->stream, ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, ctx->cand,
                       v_begin, v_end, ctx->k, ctx->ks, ctx->knn_e, ctx->knn_d, ctx->pbuf, ctx->pdirty, ctx->pcap,
                       slot_seed, ctx->counters);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

int nnd_launch_join(nnd_ctx *ctx, int64_t v_begin, int64_t v_end) {
    if (v_end <= v_begin) return 0;
    const bool wide = ctx->dp >= 128;
    switch (ctx->mcp) {
        case 16: return wide ? launch_join_t<16, 128>(ctx, v_begin, v_end) : launch_join_t<16, 32>(ctx, v_begin, v_end);
        case 32: return wide ? launch_join_t<32, 128>(ctx, v_begin, v_end) : launch_join_t<32, 32>(ctx, v_begin, v_end);
        case 64: return wide ? launch_join_t<64, 128>(ctx, v_begin, v_end) : launch_join_t<64, 32>(ctx, v_begin, v_end);
    }
    ctx->set_error("unsupported padded max_candidates %d", ctx->mcp);
    return 1;
}
