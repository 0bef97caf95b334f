// leaf_join.hip -- RP-tree leaf seeding: all pairs inside every leaf, merged straight into the
// k-lists of the leaf's own points.
//
// Replaces generate_leaf_updates + init_rp_tree (reference pynndescent_.py:73-185): for every leaf,
// every pair p != q: d = dist(x_p, x_q); keep if it beats either endpoint's current worst distance;
// push (q,d) into p's heap and (p,d) into q's heap with the "new" flag.
//
// MI355X design: one workgroup per leaf.  The leaf's rows are gathered once into LDS (coalesced
// 128-byte-line loads, swizzled), the |leaf| x |leaf| distance block is a Gram contraction on the
// f32 MFMA pipe, and -- because inside ONE tree every point belongs to exactly one leaf -- the
// workgroup OWNS the k-lists of its points for the duration of the launch: each row of the block
// is merged into its point's sorted k-list by one wave with no atomics and no update buffer
// (the reference round-trips 12-byte (p,q,d) triples through memory and scans them once per
// thread, pynndescent_.py:154-185).  Trees are processed one launch after another, so thresholds
// tighten between trees exactly like the reference's leaf blocks tighten them (pynndescent_.py:137-152).
#include <type_traits>

#include "common.h"
#include "gram.h"
#include "merge.h"
#include "state.h"

#ifndef NND_LEAF_QW_OCC
#define NND_LEAF_QW_OCC 6  // waves per SIMD of the 4-wave class: 80 VGPRs, no spills (7: 72 VGPRs with 7 spilled -- 0.1 GB of scratch traffic per tree and 2-5 % slower)
#endif
template <int NT, int NW, int DC, bool QW = false>
struct leaf_cfg {
    static constexpr int MP = NT * 16;                      // max leaf rows
    static constexpr int TR = (NT + NW - 1) / NW;           // tile rows per wave
    static constexpr int DSTRIDE = MP + 1;                  // row stride of the distance block
    static constexpr int XS_FLOATS = MP * DC;
    // NT <= 6: the whole leaf x leaf distance block lives in LDS and its rows are dealt round-robin to ALL waves
    // (balanced merges); larger leaves: each wave keeps only the 16 rows of the tile row it is working on.
    static constexpr bool FULLD = NT <= 6;
    // FULLD: the leaf x leaf block is symmetric -- only the tile pairs I <= J are computed (dealt round-robin to the
    // waves) and every off-diagonal tile is written twice, once transposed
    static constexpr int TPW = (NT * (NT + 1) / 2 + NW - 1) / NW;
    static constexpr int DB_FLOATS = FULLD ? MP * DSTRIDE : NW * 16 * DSTRIDE;
    static constexpr int QPASS = (MP + NW * 4 - 1) / (NW * 4);  // k <= 16: four rows per wave and pass (merge.h, quarter-wave merge)
    static constexpr bool PREFETCH = FULLD && !QW;           // k-list prefetch buffers sit next to the distance block
    static constexpr int PRE_FLOATS = PREFETCH ? MP * 16 * 2 : 0;  // budgeted for k <= 16; larger k uses what Xs leaves free
    static constexpr int EPI_FLOATS = DB_FLOATS + PRE_FLOATS;
    static constexpr int BIG_FLOATS = XS_FLOATS > EPI_FLOATS ? XS_FLOATS : EPI_FLOATS;  // Xs aliases the epilogue buffers
};

// QW (host: k <= 16 and the whole distance block in LDS): quarter-wave merges, four rows per wave (merge.h)
template <int NT, int NW, int DC, bool QW = false>
__global__ __launch_bounds__(NW * 64, (NT <= 5 && NW == 8) ? 8 : (QW && NT <= 5 && NW == 4) ? NND_LEAF_QW_OCC : 1) void k_leaf_join(const float *__restrict__ xp, int dp, const float *__restrict__ nrm,
                                                       int metric, const int32_t *__restrict__ perm,
                                                       const int32_t *__restrict__ wl_start,
                                                       const int32_t *__restrict__ wl_len, int64_t leaf0,
                                                       int64_t n_leaves, int k, int ks, uint32_t *__restrict__ knn_e,
                                                       float *__restrict__ knn_d, float *__restrict__ th,
                                                       long long *__restrict__ counters, int m_lo) {
    using C = leaf_cfg<NT, NW, DC, QW>;
    static_assert(!QW || C::FULLD, "quarter-wave merges read the whole distance block from LDS");
    __shared__ __attribute__((aligned(16))) float big[C::BIG_FLOATS];
    __shared__ int32_t ids[C::MP];
    __shared__ float nrs[C::MP];
    __shared__ uint2 qscr[QW ? NW * 4 * NND_Q16B_CAP : 1];  // QW: per wave, the four rows' queues of surviving candidates (merge.h)
#ifdef NND_LEAF_PAD_LDS  // occupancy experiment: fewer workgroups per CU
    __shared__ float lds_pad[NND_LEAF_PAD_LDS];
    if (dp < 0) lds_pad[threadIdx.x] = 1.0f, counters[0] = (long long)lds_pad[(threadIdx.x + 1) & 63];
#endif

    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    const int64_t leaf = leaf0 + blockIdx.x;
    if (leaf >= n_leaves) return;
    const int start = wl_start[leaf];
    const int m = wl_len[leaf];
    if (m < 2) return;  // no pairs
    if (m <= m_lo || m > C::MP) return;  // a launch takes the leaves of ITS size class (run_leaf_rounds)
    const int nt = (m + 15) >> 4;
    const int mp = nt << 4;

    for (int r = tid; r < C::MP; r += NW * 64) ids[r] = r < m ? perm[start + r] : -1;
    __syncthreads();

    // One burst of global loads once the ids are known: the rows of the first K block (registers, then LDS), the
    // norms, and -- when they fit the register budget -- the k-lists of all m points, which are only needed after the
    // Gram.  A leaf then pays one exposed memory latency instead of three.
    constexpr int NTHR = NW * 64;
    constexpr int NLD = (C::MP * (DC / 4) + NTHR - 1) / NTHR;  // 16-byte row chunks per thread
    constexpr int NKL = (C::MP * 16 + NTHR - 1) / NTHR;        // k-list words per thread (row stride <= 16)
    constexpr bool quarter = QW;  // four rows per wave, k-lists prefetched straight into the lanes that merge them
    const bool use_pre = !quarter && C::PREFETCH && ks <= 16 && (m * ks * 2 <= C::BIG_FLOATS - C::DB_FLOATS);
    uint32_t qe[C::QPASS];
    float qd[C::QPASS];
    float *Xs = big;
    uint32_t pe[NKL];
    float pd[NKL];
    f32x4 acc[C::TR][NT];   // !FULLD: tile row I = w + tr*NW against every J
    f32x4 acct[C::TPW];     // FULLD: tile pairs t = w, w + NW, ... of the upper triangle
    int tI[C::TPW], tJ[C::TPW];
    bool tOn[C::TPW];
#pragma unroll
    for (int tr = 0; tr < C::TR; tr++)
#pragma unroll
        for (int J = 0; J < NT; J++) acc[tr][J] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < C::TPW; q++) {
        acct[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        int t = w + q * NW, I = 0;
        while (I < nt && t >= nt - I) {  // wave-uniform: row-major walk of the upper triangle
            t -= nt - I;
            I++;
        }
        tOn[q] = I < nt;  // idle slots recompute tile (0, 0) and drop the result: no conditional register arrays
        tI[q] = I < nt ? I : 0;
        tJ[q] = I < nt ? I + t : 0;
    }

    for (int c0 = 0; c0 < dp; c0 += DC) {
        const int cw = (dp - c0) < DC ? (dp - c0) : DC;
        {
            static_assert(DC == 64, "cw is 32 or 64 (dp is a multiple of 32): chunk index by shift");
            const int nch = cw >> 2, total = mp * nch, nsh = cw == 64 ? 4 : 3;
            f32x4 rv[NLD];
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                // unconditional (clamped) loads: a conditionally assigned register array costs whole-array copies
                const int idx = tid + q * NTHR, idc = idx < total ? idx : 0;
                const int r = idc >> nsh, ch = idc & (nch - 1);
                const int id = ids[r];
                rv[q] = *(const f32x4 *)(xp + (int64_t)(id >= 0 ? id : 0) * dp + c0 + 4 * ch);  // rows >= m: row 0, never used
            }
            float my_nrm = 0.0f;
            if (c0 == 0) {
                if (!QW && tid < C::MP) {  // QW: the squared norms are the diagonal of the Gram block (see below)
                    const int id = ids[tid];
                    my_nrm = nrm[id >= 0 ? id : 0];
                }
                if constexpr (quarter) {
#pragma unroll
                    for (int q = 0; q < C::QPASS; q++) {
                        const int i = q * NW * 4 + (tid >> 4), j = lane & 15;
                        const bool on = i < m && j < k;
                        const int64_t v = ids[i < m ? i : 0];
                        const uint32_t ev = knn_e[v * ks + (on ? j : 0)];
                        const float dv = knn_d[v * ks + (on ? j : 0)];
                        qe[q] = on ? ev : NND_EMPTY_E;
                        qd[q] = on ? dv : INFINITY;
                    }
                }
                if (use_pre) {
#pragma unroll
                    for (int q = 0; q < NKL; q++) {
                        const int idx = tid + q * NTHR, idc = idx < m * ks ? idx : 0;
                        const int i = idc / ks, j = idc - i * ks;
                        pe[q] = knn_e[(int64_t)ids[i] * ks + j];
                        pd[q] = knn_d[(int64_t)ids[i] * ks + j];
                    }
                }
            }
            if (c0 > 0) __syncthreads();  // the previous block's operand reads are done
#pragma unroll
            for (int q = 0; q < NLD; q++) {
                const int idx = tid + q * NTHR;
                if (idx < total) {
                    const int r = idx >> nsh, ch = idx & (nch - 1);
                    *(f32x4 *)&Xs[nnd_swz<DC>(r, ch)] = rv[q];
                }
            }
            if (!QW && c0 == 0 && tid < C::MP) nrs[tid] = my_nrm;
        }
        __syncthreads();
        if constexpr (C::FULLD) {
            const int lr = lane & 15, lg = lane >> 4;
            // only the slots that hold a tile issue LDS reads and MFMAs (they are a prefix: t = w + q * NW grows with q;
            // the count is wave-uniform): a 41-point leaf has 6 tiles for 8 waves x TPW slots
            auto run = [&](auto nq_tag) {
                constexpr int NQ = decltype(nq_tag)::value;
#ifdef NND_LEAF_NOGRAM  // timing experiments only
                for (int t = 0; t < 1; t++) {
#else
                for (int t = 0; t < (cw >> 4); t++) {
#endif
                    const int c = 4 * t + lg;
                    float4 a[NQ], b[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; q++) {
                        a[q] = *(const float4 *)&Xs[nnd_swz<DC>(tI[q] * 16 + lr, c)];
                        b[q] = *(const float4 *)&Xs[nnd_swz<DC>(tJ[q] * 16 + lr, c)];
                    }
#pragma unroll
                    for (int q = 0; q < NQ; q++) acct[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acct[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NQ; q++) acct[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acct[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NQ; q++) acct[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acct[q], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < NQ; q++) acct[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acct[q], 0, 0, 0);
                }
            };
            int nq = 0;
#pragma unroll
            for (int q = 0; q < C::TPW; q++) nq += tOn[q] ? 1 : 0;
            static_assert(C::TPW <= 4, "slot dispatch below");
            if (nq == 1) run(std::integral_constant<int, 1>{});
            else if (nq == 2) run(std::integral_constant<int, C::TPW >= 2 ? 2 : 1>{});
            else if (nq == 3) run(std::integral_constant<int, C::TPW >= 3 ? 3 : 1>{});
            else if (nq == 4) run(std::integral_constant<int, C::TPW >= 4 ? 4 : 1>{});
        } else {
#pragma unroll
            for (int tr = 0; tr < C::TR; tr++) {
                const int I = w + tr * NW;
                if (I < nt) nnd_gram_chunk<DC, NT>(Xs, I * 16, 0, cw, acc[tr], [nt](int J) { return J < nt; });
            }
        }
    }
    if constexpr (QW) {
        // |x_i|^2 = G[i][i]: the diagonal tiles hold the squared norms of this leaf's rows, in the SAME arithmetic as the
        // off-diagonal products (d(i, i) is exactly 0), and the leaf needs no gather of n random 4-byte norms -- each of
        // which costs a 128-byte line: 128 MB of the ~1 GB a launch moves at 1 M points.
        const int r16d = lane & 15, gd = lane >> 4;
#pragma unroll
        for (int q = 0; q < C::TPW; q++) {
            if (!tOn[q] || tI[q] != tJ[q]) continue;  // wave-uniform
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (4 * gd + r == r16d) nrs[tI[q] * 16 + r16d] = acct[q][r];
        }
    }
    __syncthreads();  // Xs is overwritten by the distance blocks below

    // distances -> LDS, then every row is merged into its point's k-list by one wave
    const int r16 = lane & 15, g = lane >> 4;
    int accepted = 0;
    if constexpr (C::FULLD) {
        float *Dm = big;  // MP x DSTRIDE; Xs is dead (barrier at the end of the K loop)
#pragma unroll
        for (int q = 0; q < C::TPW; q++) {
            if (!tOn[q]) continue;
            const int I = tI[q], J = tJ[q];
            const int j = J * 16 + r16;
            const float nj = nrs[j];
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int il = I * 16 + 4 * g + r;
                const float dv = nnd_gram_to_dist(metric, acct[q][r], nrs[il], nj);
                Dm[il * C::DSTRIDE + j] = dv;
                if (I != J) Dm[j * C::DSTRIDE + il] = dv;  // the mirrored tile
            }
        }
        // the k-lists fetched with the rows (this workgroup owns them for this tree) land next to the distance block
        uint32_t *pre_e = (uint32_t *)(big + C::DB_FLOATS);
        float *pre_d = big + C::DB_FLOATS + m * ks;
        if (use_pre) {
#pragma unroll
            for (int q = 0; q < NKL; q++) {
                const int idx = tid + q * NTHR;
                if (idx < m * ks) {
                    pre_e[idx] = pe[q];
                    pre_d[idx] = pd[q];
                }
            }
        }
        __syncthreads();
        if constexpr (quarter) {
#pragma unroll
            for (int q = 0; q < C::QPASS; q++) {
                if (q * NW * 4 + w * 4 >= m) break;  // wave-uniform: no row left for this wave
                const int i = q * NW * 4 + (tid >> 4);
                const bool on = i < m;
                const float *Drow = Dm + (on ? i : 0) * C::DSTRIDE;
                const int64_t v = ids[on ? i : 0];
#ifdef NND_LEAF_NOMERGE  // timing experiments only
                if (qe[q] == 12345u && qd[q] == 3.0f && Drow[lane] == 7.0f) accepted++;
                if (false)
#endif
#ifdef NND_LEAF_OLD_MERGE  // A/B: the one-candidate-per-step insertion of rounds 2-5
                accepted += nnd_merge_rows_q16<NT>(on, knn_e + v * ks, knn_d + v * ks, th + v, qe[q], qd[q], k, m,
                                                   [&](int c, uint32_t &id, float &dc) {
                                                       id = (uint32_t)ids[c];
                                                       dc = Drow[c];
                                                       return c != i;  // pynndescent_.py:97: p != q
                                                   });
#else
                accepted += nnd_merge_rows_q16b<NT>(on, knn_e + v * ks, knn_d + v * ks, th + v, qe[q], qd[q], k, m,
                                                    [&](int c, uint32_t &id, float &dc) {
                                                        id = (uint32_t)ids[c];
                                                        dc = Drow[c];
                                                        return c != i;  // pynndescent_.py:97: p != q
                                                    }, qscr + w * 4 * NND_Q16B_CAP);
#endif
            }
        } else
        for (int i = w; i < m; i += NW) {  // rows dealt round-robin: every wave gets ~m/NW merges
            const float *Drow = Dm + i * C::DSTRIDE;
            const int64_t v = ids[i];
            uint32_t e0 = NND_EMPTY_E;
            float d0 = INFINITY;
            if (lane < k) {
                if (use_pre) {
                    e0 = pre_e[i * ks + lane];
                    d0 = pre_d[i * ks + lane];
                } else {
                    e0 = knn_e[v * ks + lane];
                    d0 = knn_d[v * ks + lane];
                }
            }
            auto cf = [&](int c, uint32_t &id, float &dc) {
                id = (uint32_t)ids[c];
                dc = Drow[c];
                return c != i;  // pynndescent_.py:97: p != q
            };
#ifdef NND_LEAF_NOMERGE
            if (e0 == 12345u && d0 == 3.0f && Drow[lane] == 7.0f) accepted++;
            if (false)
#endif
            if (C::MP > 64 && m <= 64)  // wave-uniform: one candidate per lane is enough for this leaf
                accepted += nnd_merge_row_regs<1>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m, cf);
            else
                accepted += nnd_merge_row_regs<(C::MP + 63) / 64>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m, cf);
        }
    } else {
        float *Dw = big + w * 16 * C::DSTRIDE;
#pragma unroll
        for (int tr = 0; tr < C::TR; tr++) {
            const int I = w + tr * NW;
            if (I >= nt) continue;
#pragma unroll
            for (int J = 0; J < NT; J++) {
                if (J < nt) {
                    const int j = J * 16 + r16;
                    const float nj = nrs[j];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int il = 4 * g + r;
                        Dw[il * C::DSTRIDE + j] = nnd_gram_to_dist(metric, acc[tr][J][r], nrs[I * 16 + il], nj);
                    }
                }
            }
            nnd_wave_lds_sync();
            // rolling prefetch: the k-list row of point il+1 is in flight while point il is merged
            const int lk = lane < k ? lane : 0;
            uint32_t e_nx = knn_e[(int64_t)ids[I * 16] * ks + lk];
            float d_nx = knn_d[(int64_t)ids[I * 16] * ks + lk];
            for (int il = 0; il < 16; il++) {
                const int i = I * 16 + il;
                if (i >= m) break;
                const float *Drow = Dw + il * C::DSTRIDE;
                const int64_t v = ids[i];
                const uint32_t e0 = lane < k ? e_nx : NND_EMPTY_E;
                const float d0 = lane < k ? d_nx : INFINITY;
                if (il + 1 < 16 && i + 1 < m) {
                    const int64_t vn = ids[i + 1];
                    e_nx = knn_e[vn * ks + lk];
                    d_nx = knn_d[vn * ks + lk];
                }
                accepted += nnd_merge_row_regs<(C::MP + 63) / 64>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m,
                                                                 [&](int c, uint32_t &id, float &dc) {
                                                                     id = (uint32_t)ids[c];
                                                                     dc = Drow[c];
                                                                     return c != i;  // pynndescent_.py:97: p != q
                                                                 });
            }
            nnd_wave_lds_sync();
        }
    }
    __syncthreads();
    int *wacc = (int *)nrs;  // nrs is dead
    if (lane == 0) wacc[w] = accepted;
    __syncthreads();
    if (tid == 0) {
        long long a = 0;
        for (int i = 0; i < NW; i++) a += wacc[i];
        nnd_count(counters, CNT_ACCEPT, a);
        nnd_count(counters, CNT_PAIRS, (long long)m * (m - 1) / 2);
        nnd_count(counters, CNT_ROWS, m);
        // MFMA instructions: tiles computed (upper triangle when the whole block lives in LDS) x dp / 4 k-steps
        nnd_count(counters, CNT_MFMA, (long long)(C::FULLD ? nt * (nt + 1) / 2 : nt * nt) * (dp >> 2));
    }
}

// Leaves of more than 96 points (k = 30, the reference's default, gives leaf_size 150).  The one-workgroup-per-leaf kernel
// above then keeps 16 rows of the distance block per wave, needs 82 KB of LDS (ONE workgroup of 8 waves per CU) and deals
// whole tile rows to its waves (10 tile rows over 8 waves: two waves merge 32 rows while six merge 16): 3.4 ms per tree at
// 1 M points, 7 x the k = 15 cost.  Here a workgroup of NW waves takes a BLOCK OF 8 * NW ROWS of a leaf: all m rows of
// the leaf are staged (they are the block's candidates), the block x m distance rows are computed (tiles dealt
// round-robin) and each wave merges 8 rows.  41 KB of LDS at 160 points: three workgroups per CU, every wave with the
// same share of the merges; the leaf's Gram block is computed twice (no symmetry across workgroups) and its rows are
// staged once per block (the reason for 64-row blocks: 32-row blocks staged every leaf five times).  Rows of one leaf
// are owned by exactly one workgroup: no atomics.
template <int NT, int NW = 8, bool WIDE = false>
__global__ __launch_bounds__(NW * 64, 2) void k_leaf_join_rb(const float *__restrict__ xp, int dp, const float *__restrict__ nrm, int metric,
                                                         const int32_t *__restrict__ perm, const int32_t *__restrict__ wl_start,
                                                         const int32_t *__restrict__ wl_len, int64_t leaf0, int64_t n_leaves, int k, int ks,
                                                         uint32_t *__restrict__ knn_e, float *__restrict__ knn_d, float *__restrict__ th,
                                                         long long *__restrict__ counters) {
    constexpr int DC = 64, MP = NT * 16, RB = NW * 8, TRB = RB / 16, DSTRIDE = MP + 1;  // 8 rows merged per wave
    constexpr int TPW = (TRB * NT + NW - 1) / NW;                     // tiles per wave: TRB tile rows x NT tile columns
    constexpr int NLD = (MP * (DC / 4) + NW * 64 - 1) / (NW * 64);    // 16-byte row chunks per thread and K block
    constexpr int BIG = MP * DC > RB * DSTRIDE ? MP * DC : RB * DSTRIDE;
    __shared__ __attribute__((aligned(16))) float big[BIG];           // row tile, then the 32 x m distance block
    __shared__ int32_t ids[MP];
    __shared__ float nrs[MP];
    __shared__ uint64_t wide_scr[WIDE ? NW : 1][WIDE ? NND_WIDE_SCRATCH_WORDS : 1];  // 64 < k <= NND_WIDE_K: rows merged through LDS
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    const int64_t leaf = leaf0 + blockIdx.x;
    if (leaf >= n_leaves) return;
    const int start = wl_start[leaf], m = wl_len[leaf];
    const int r0 = blockIdx.y * RB;  // first row of this workgroup's block
    if (m < 2 || r0 >= m) return;
    const int nt = (m + 15) >> 4, mp = nt << 4;
    for (int r = tid; r < MP; r += NW * 64) {
        const int id = r < m ? perm[start + r] : -1;
        ids[r] = id;
        nrs[r] = nrm[id >= 0 ? id : 0];
    }
    __syncthreads();
    f32x4 acc[TPW];
    int tI[TPW], tJ[TPW];
    bool tOn[TPW];
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int t = w + q * NW;                // row-major over (TRB, nt)
        const int I = t / nt, J = t - I * nt;    // wave-uniform
        tOn[q] = t < TRB * nt && (r0 >> 4) + I < nt;
        tI[q] = tOn[q] ? (r0 >> 4) + I : 0;
        tJ[q] = tOn[q] ? J : 0;
    }
    float *Xs = big;
    const int r16 = lane & 15, g = lane >> 4;
    for (int c0 = 0; c0 < dp; c0 += DC) {
        const int cw = (dp - c0) < DC ? (dp - c0) : DC;
        const int nch = cw >> 2, total = mp * nch, nsh = cw == 64 ? 4 : 3;
        f32x4 rv[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * NW * 64, idc = idx < total ? idx : 0;
            const int r = idc >> nsh, ch = idc & (nch - 1);
            const int id = ids[r];
            rv[q] = *(const f32x4 *)(xp + (int64_t)(id >= 0 ? id : 0) * dp + c0 + 4 * ch);
        }
        if (c0 > 0) __syncthreads();  // the previous block's operand reads are done
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * NW * 64;
            if (idx < total) {
                const int r = idx >> nsh, ch = idx & (nch - 1);
                *(f32x4 *)&Xs[nnd_swz<DC>(r, ch)] = rv[q];
            }
        }
        __syncthreads();
        for (int t = 0; t < (cw >> 4); t++) {
            const int c = 4 * t + g;
            float4 a[TPW], b[TPW];
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                a[q] = *(const float4 *)&Xs[nnd_swz<DC>(tI[q] * 16 + r16, c)];
                b[q] = *(const float4 *)&Xs[nnd_swz<DC>(tJ[q] * 16 + r16, c)];
            }
#pragma unroll
            for (int q = 0; q < TPW; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].x, b[q].x, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TPW; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].y, b[q].y, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TPW; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].z, b[q].z, acc[q], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < TPW; q++) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q].w, b[q].w, acc[q], 0, 0, 0);
        }
    }
    __syncthreads();  // Xs is overwritten by the distance block
    float *Dm = big;  // RB x DSTRIDE: row il = global leaf row r0 + il
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        if (!tOn[q]) continue;
        const int j = tJ[q] * 16 + r16;
        const float nj = nrs[j];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int il = tI[q] * 16 - r0 + 4 * g + r;
            Dm[il * DSTRIDE + j] = nnd_gram_to_dist(metric, acc[q][r], nrs[r0 + il], nj);
        }
    }
    __syncthreads();
    int accepted = 0;
    const int lk = lane < k ? lane : 0;
    const int row_first = r0 + w * (RB / NW), row_end = (row_first + RB / NW) < m ? (row_first + RB / NW) : m;
    uint32_t e_nx = NND_EMPTY_E;
    float d_nx = INFINITY;
    if (row_first < row_end) {
        e_nx = knn_e[(int64_t)ids[row_first] * ks + lk];
        d_nx = knn_d[(int64_t)ids[row_first] * ks + lk];
    }
    if constexpr (WIDE) {
        for (int i = row_first; i < row_end; i++) {
            const float *Drow = Dm + (i - r0) * DSTRIDE;
            const int64_t v = ids[i];
            accepted += nnd_merge_row_lds<(MP + 63) / 64>(wide_scr[w], knn_e + v * ks, knn_d + v * ks, th + v, k, m,
                                                         [&](int c, uint32_t &id, float &dc) {
                                                             id = (uint32_t)ids[c];
                                                             dc = Drow[c];
                                                             return c != i;  // pynndescent_.py:97: p != q
                                                         });
            nnd_wave_lds_sync();
        }
    } else
    for (int i = row_first; i < row_end; i++) {  // rolling prefetch: the next row's k-list is in flight during this merge
        const float *Drow = Dm + (i - r0) * DSTRIDE;
        const int64_t v = ids[i];
        const uint32_t e0 = lane < k ? e_nx : NND_EMPTY_E;
        const float d0 = lane < k ? d_nx : INFINITY;
        if (i + 1 < row_end) {
            const int64_t vn = ids[i + 1];
            e_nx = knn_e[vn * ks + lk];
            d_nx = knn_d[vn * ks + lk];
        }
        auto cf = [&](int c, uint32_t &id, float &dc) {
            id = (uint32_t)ids[c];
            dc = Drow[c];
            return c != i;  // pynndescent_.py:97: p != q
        };
        if (m <= 64) accepted += nnd_merge_row_regs<1>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m, cf);
        else if (MP > 128 && m <= 128) accepted += nnd_merge_row_regs<2>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m, cf);
        else accepted += nnd_merge_row_regs<(MP + 63) / 64>(knn_e + v * ks, knn_d + v * ks, th + v, e0, d0, k, m, cf);
    }
    __syncthreads();
    int *wacc = (int *)nrs;  // nrs is dead
    if (lane == 0) wacc[w] = accepted;
    __syncthreads();
    if (tid == 0) {
        long long a = 0;
        for (int q = 0; q < NW; q++) a += wacc[q];
        nnd_count(counters, CNT_ACCEPT, a);
        if (blockIdx.y == 0) {  // per-leaf statistics once
            nnd_count(counters, CNT_PAIRS, (long long)m * (m - 1) / 2);
            nnd_count(counters, CNT_ROWS, m);
        }
        const int trows = (r0 >> 4) + TRB <= nt ? TRB : nt - (r0 >> 4);
        nnd_count(counters, CNT_MFMA, (long long)trows * nt * (dp >> 2));
    }
}

// Rows of 17 .. 32 neighbours (k = 30, the reference's default, with leaves of up to 150 points), round 6.  k_leaf_join_rb above
// gives every 64-row block of a leaf its own workgroup: the leaf's rows are staged once per block (three times at 150 points),
// the Gram block is computed without symmetry (120 tiles for 55), and each row is merged by a whole wave (rank counting over
// v_readlane broadcasts): 1.33 ms per tree at 1 M points even when nothing passes a threshold, 15.6 ms for eight trees
// (profiles/r05_k30_per_tree.log).  Here ONE workgroup takes the whole leaf: rows staged once, the upper triangle of the Gram
// block in registers (<= 7 tiles per wave), and the distance rows go to LDS 64 at a time -- a pass writes the rows of its block
// from the tiles that hold them, directly or transposed, over the LDS the row tile occupied -- where two rows per wave are
// merged by nnd_merge_rows_q32b (candidates compacted, then a 32-lane sorting network).  Rows of one leaf are owned by one
// workgroup: no atomics.
#ifndef NND_SYM_OCC_S
#define NND_SYM_OCC_S 4
#endif
#ifndef NND_SYM_OCC_L
#define NND_SYM_OCC_L 3
#endif
template <int NT, int NW = 8>
__global__ __launch_bounds__(NW * 64, NT <= 6 ? NND_SYM_OCC_S : NND_SYM_OCC_L) void k_leaf_join_sym(const float *__restrict__ xp, int dp, const float *__restrict__ nrm, int metric,
                                                          const int32_t *__restrict__ perm, const int32_t *__restrict__ wl_start,
                                                          const int32_t *__restrict__ wl_len, int64_t leaf0, int64_t n_leaves, int k, int ks,
                                                          uint32_t *__restrict__ knn_e, float *__restrict__ knn_d, float *__restrict__ th,
                                                          long long *__restrict__ counters, int m_lo) {
    constexpr int DC = 64, MP = NT * 16, RB = 64, DSTRIDE = MP + 1, NTHR = NW * 64;
    constexpr int NTILES = NT * (NT + 1) / 2, TPW = (NTILES + NW - 1) / NW;
    constexpr int NLD = (MP * (DC / 4) + NTHR - 1) / NTHR;  // 16-byte row chunks per thread and K block
    constexpr int BIG = MP * DC > RB * DSTRIDE ? MP * DC : RB * DSTRIDE;
    constexpr int SPW = RB / (2 * NW);                      // merge steps (two rows each) per wave and pass
    __shared__ __attribute__((aligned(16))) float big[BIG];  // row tile, then 64 distance rows at a time
    __shared__ int32_t ids[MP];
    __shared__ float nrs[MP];
    __shared__ uint2 qscr[NW * 2 * NND_Q32B_CAP];
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    const int64_t leaf = leaf0 + blockIdx.x;
    if (leaf >= n_leaves) return;
    const int start = wl_start[leaf], m = wl_len[leaf];
    if (m < 2) return;                    // no pairs
    if (m <= m_lo || m > MP) return;      // a launch takes the leaves of ITS size class (run_leaf_rounds)
    const int nt = (m + 15) >> 4, mp = nt << 4;
    for (int r = tid; r < MP; r += NTHR) {
        const int id = r < m ? perm[start + r] : -1;
        ids[r] = id;
        nrs[r] = nrm[id >= 0 ? id : 0];
    }
    __syncthreads();
    f32x4 acc[TPW];
    int tI[TPW], tJ[TPW];
    bool tOn[TPW];
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        int t = w + q * NW, I = 0;
        while (I < nt && t >= nt - I) {  // wave-uniform: row-major walk of the upper triangle
            t -= nt - I;
            I++;
        }
        tOn[q] = I < nt;
        tI[q] = I < nt ? I : 0;
        tJ[q] = I < nt ? I + t : 0;
    }
    float *Xs = big;
    const int r16 = lane & 15, g = lane >> 4;
    for (int c0 = 0; c0 < dp; c0 += DC) {
        const int cw = (dp - c0) < DC ? (dp - c0) : DC;
        const int nch = cw >> 2, total = mp * nch, nsh = cw == 64 ? 4 : 3;
        f32x4 rv[NLD];
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * NTHR, idc = idx < total ? idx : 0;
            const int r = idc >> nsh, ch = idc & (nch - 1);
            const int id = ids[r];
            rv[q] = *(const f32x4 *)(xp + (int64_t)(id >= 0 ? id : 0) * dp + c0 + 4 * ch);
        }
        if (c0 > 0) __syncthreads();  // the previous block's operand reads are done
#pragma unroll
        for (int q = 0; q < NLD; q++) {
            const int idx = tid + q * NTHR;
            if (idx < total) {
                const int r = idx >> nsh, ch = idx & (nch - 1);
                *(f32x4 *)&Xs[nnd_swz<DC>(r, ch)] = rv[q];
            }
        }
        __syncthreads();
        for (int t = 0; t < (cw >> 4); t++) {
            const int c = 4 * t + g;
#pragma unroll
            for (int q = 0; q < TPW; q++) {
                if (!tOn[q]) continue;  // wave-uniform
                const float4 a = *(const float4 *)&Xs[nnd_swz<DC>(tI[q] * 16 + r16, c)];
                const float4 b = *(const float4 *)&Xs[nnd_swz<DC>(tJ[q] * 16 + r16, c)];
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[q], 0, 0, 0);
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[q], 0, 0, 0);
            }
        }
    }
    // Gram values -> distances, once (registers)
#pragma unroll
    for (int q = 0; q < TPW; q++) {
        if (!tOn[q]) continue;
        const float nj = nrs[tJ[q] * 16 + r16];
#pragma unroll
        for (int r = 0; r < 4; r++) acc[q][r] = nnd_gram_to_dist(metric, acc[q][r], nrs[tI[q] * 16 + 4 * g + r], nj);
    }
    float *Dm = big;  // RB x DSTRIDE: row il = leaf row p0 + il
    int accepted = 0;
    const int j32 = lane & 31, h = lane >> 5;
    for (int p0 = 0; p0 < m; p0 += RB) {
        // this wave's k-list rows of the pass: in flight while the distance rows are written
        uint32_t pe[SPW];
        float pd[SPW];
#pragma unroll
        for (int st = 0; st < SPW; st++) {
            const int i = p0 + w * (2 * SPW) + 2 * st + h;
            const bool on = i < m && j32 < k;
            const int64_t v = ids[i < m ? i : 0];
            const uint32_t ev = knn_e[v * ks + (on ? j32 : 0)];
            const float dv = knn_d[v * ks + (on ? j32 : 0)];
            pe[st] = on ? ev : NND_EMPTY_E;
            pd[st] = on ? dv : INFINITY;
        }
        __syncthreads();  // Xs (first pass) / the previous pass's rows are done with
#pragma unroll
        for (int q = 0; q < TPW; q++) {
            if (!tOn[q]) continue;
            const int I = tI[q], J = tJ[q];
            if (I * 16 >= p0 && I * 16 < p0 + RB) {  // the tile's rows belong to this pass
#pragma unroll
                for (int r = 0; r < 4; r++) Dm[(I * 16 + 4 * g + r - p0) * DSTRIDE + J * 16 + r16] = acc[q][r];
            }
            if (I != J && J * 16 >= p0 && J * 16 < p0 + RB) {  // ... and, transposed, its columns
#pragma unroll
                for (int r = 0; r < 4; r++) Dm[(J * 16 + r16 - p0) * DSTRIDE + I * 16 + 4 * g + r] = acc[q][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int st = 0; st < SPW; st++) {
            if (p0 + w * (2 * SPW) + 2 * st >= m) break;  // wave-uniform: no row left for this wave
            const int i = p0 + w * (2 * SPW) + 2 * st + h;
            const bool on = i < m;
            const float *Drow = Dm + (on ? i - p0 : 0) * DSTRIDE;
            const int64_t v = ids[on ? i : 0];
            accepted += nnd_merge_rows_q32b(on, knn_e + v * ks, knn_d + v * ks, th + v, pe[st], pd[st], k, m,
                                            [&](int c, uint32_t &id, float &dc) {
                                                id = (uint32_t)ids[c];
                                                dc = Drow[c];
                                                return c != i;  // pynndescent_.py:97: p != q
                                            }, qscr + w * 2 * NND_Q32B_CAP);
        }
    }
    __syncthreads();
    int *wacc = (int *)nrs;  // nrs is dead
    if (lane == 0) wacc[w] = accepted;
    __syncthreads();
    if (tid == 0) {
        long long a = 0;
        for (int q = 0; q < NW; q++) a += wacc[q];
        nnd_count(counters, CNT_ACCEPT, a);
        nnd_count(counters, CNT_PAIRS, (long long)m * (m - 1) / 2);
        nnd_count(counters, CNT_ROWS, m);
        nnd_count(counters, CNT_MFMA, (long long)(nt * (nt + 1) / 2) * (dp >> 2));
    }
}

// Work list: leaves longer than 256 points (possible only when max_depth cuts the recursion short,
// rp_trees.py:2188) are cut into runs of <= 256 consecutive positions for seeding purposes.
static constexpr int LEAF_MAX = 256;

// Seed the k-lists from leaves given as a work list: members of leaf i are members[wl_start[i] .. + wl_len[i]); the leaves
// of round r are [tb[r], tb[r + 1]).  Inside a round no point may belong to two leaves -- the workgroup of a leaf owns
// the k-lists of its points -- so rounds run launch after launch (the library's own forest: one round per tree).
static int run_leaf_rounds(nnd_ctx *ctx, const int32_t *perm, const int32_t *d_ws, const int32_t *d_wl,
                           const std::vector<int64_t> &tb, int maxlen) {
    if (nnd_zero_counters(ctx)) return 1;
    const int T = (int)tb.size() - 1;
    for (int t = 0; t < T; t++) {
        int64_t cnt = tb[t + 1] - tb[t];
        if (cnt <= 0) continue;
        dim3 grid((unsigned)cnt);
#define LEAF_ARGS ctx->xp, ctx->dp, ctx->nrm, ctx->p.metric, perm, d_ws, d_wl, tb[t], tb[t + 1], ctx->k, ctx->ks, \
                  ctx->knn_e, ctx->knn_d, ctx->th, ctx->counters
        const bool qw = ctx->k <= 16;
        if (ctx->k > NND_MAX_K) {  // wide rows: the row-block kernel with LDS merges, whatever the leaf size
            if (maxlen <= 128)
                hipLaunchKernelGGL((k_leaf_join_rb<8, 8, true>), dim3((unsigned)cnt, 2), dim3(512), 0, ctx->stream, LEAF_ARGS);
            else if (maxlen <= 160)
                hipLaunchKernelGGL((k_leaf_join_rb<10, 8, true>), dim3((unsigned)cnt, 3), dim3(512), 0, ctx->stream, LEAF_ARGS);
            else
                hipLaunchKernelGGL((k_leaf_join_rb<16, 8, true>), dim3((unsigned)cnt, 4), dim3(512), 0, ctx->stream, LEAF_ARGS);
        }
#ifndef NND_LEAF_OLD_K32
        else if (!qw && ctx->k <= 32 && maxlen <= 160) {  // k = 17 .. 32: one workgroup per leaf, two rows per wave (k_leaf_join_sym)
            hipLaunchKernelGGL((k_leaf_join_sym<6, 8>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
            if (maxlen > 96) hipLaunchKernelGGL((k_leaf_join_sym<10, 8>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 96);
        }
#endif
        else if (maxlen <= 64 && qw)
            hipLaunchKernelGGL((k_leaf_join<4, 8, 64, true>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
        else if (maxlen <= 64)
            hipLaunchKernelGGL((k_leaf_join<4, 8, 64>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
        else if (maxlen <= 80 && qw) {  // the default leaf_size (<= 75 points) with k <= 16
            // Two size classes, two launches over the tree's leaf table (a workgroup whose leaf belongs to the other class
            // leaves at once): leaves of <= 64 points -- 85 % of the leaves, 3/4 of the points -- get FOUR waves and 21 KB of LDS,
            // so that 6 of them are in flight per CU instead of 4.  The kernel is bound by exposed latency, not by a pipe
            // (profiles/r06_leaf_occupancy.log: 375 / 425 / 535 us per tree with 4 / 3 / 2 workgroups per CU).
#ifdef NND_LEAF_ONE_CLASS
            hipLaunchKernelGGL((k_leaf_join<5, 8, 64, true>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
#else
            hipLaunchKernelGGL((k_leaf_join<4, 4, 64, true>), grid, dim3(256), 0, ctx->stream, LEAF_ARGS, 0);
#ifdef NND_LEAF_EMPTY_L  // timing experiment: what do the workgroups that leave at once cost?
            hipLaunchKernelGGL((k_leaf_join<5, 8, 64, true>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 1000);
#elif defined(NND_LEAF_L_NW4)  // experiment: the large class with four waves too
            hipLaunchKernelGGL((k_leaf_join<5, 4, 64, true>), grid, dim3(256), 0, ctx->stream, LEAF_ARGS, 64);
#else
            hipLaunchKernelGGL((k_leaf_join<5, 8, 64, true>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 64);
#endif
#endif
        }
        else if (maxlen <= 80)
            hipLaunchKernelGGL((k_leaf_join<5, 8, 64>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
        else if (maxlen <= 96 && qw)
            hipLaunchKernelGGL((k_leaf_join<6, 8, 64, true>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
        else if (maxlen <= 96)
            hipLaunchKernelGGL((k_leaf_join<6, 8, 64>), grid, dim3(512), 0, ctx->stream, LEAF_ARGS, 0);
        else if (maxlen <= 128)  // larger leaves: a workgroup per block of 32 rows (k_leaf_join_rb)
            hipLaunchKernelGGL((k_leaf_join_rb<8, 8>), dim3((unsigned)cnt, 2), dim3(512), 0, ctx->stream, LEAF_ARGS);
        else if (maxlen <= 160)  // k = 30 (leaf_size 150)
            hipLaunchKernelGGL((k_leaf_join_rb<10, 8>), dim3((unsigned)cnt, 3), dim3(512), 0, ctx->stream, LEAF_ARGS);
        else
            hipLaunchKernelGGL((k_leaf_join_rb<16, 8>), dim3((unsigned)cnt, 4), dim3(512), 0, ctx->stream, LEAF_ARGS);
#undef LEAF_ARGS
    }
    NND_HIP_CHECK(hipGetLastError());
    if (nnd_read_counters(ctx)) return 1;
    ctx->stats.leaf_pairs = ctx->h_counters[CNT_PAIRS];
    ctx->stats.leaf_rows = ctx->h_counters[CNT_ROWS];
    ctx->stats.leaf_mfma = ctx->h_counters[CNT_MFMA];
    return 0;
}

int nnd_launch_leaf_init(nnd_ctx *ctx) {
    if (!ctx->forest_built || ctx->n_leaves == 0) return 0;
    const int T = (int)ctx->tree_leaf_begin.size() - 1;  // rounds: the handle's trees, or every tree of a sharded build
    // Work list = the leaf table itself (device resident, per-tree offsets known from the forest build) unless some
    // leaf is longer than LEAF_MAX: only then are the pieces listed on the host and uploaded.
    const int32_t *d_ws = ctx->leaf_start, *d_wl = ctx->leaf_len;
    std::vector<int64_t> tb(ctx->tree_leaf_begin.begin(), ctx->tree_leaf_begin.end());
    int maxlen = ctx->max_leaf;
    if (ctx->max_leaf > LEAF_MAX) {
        if (nnd_fetch_leaf_tables(ctx)) return 1;
        const std::vector<int32_t> &hs = ctx->h_leaf_start, &hl = ctx->h_leaf_len;
        std::vector<int32_t> ws, wl;
        ws.reserve(ctx->n_leaves);
        wl.reserve(ctx->n_leaves);
        maxlen = 0;
        for (int t = 0; t < T; t++) {
            tb[t] = (int64_t)ws.size();
            for (int64_t i = ctx->tree_leaf_begin[t]; i < ctx->tree_leaf_begin[t + 1]; i++) {
                int32_t a = hs[i], len = hl[i];
                while (len > 0) {
                    int32_t piece = len > LEAF_MAX ? LEAF_MAX : len;
                    ws.push_back(a);
                    wl.push_back(piece);
                    if (piece > maxlen) maxlen = piece;
                    a += piece;
                    len -= piece;
                }
            }
        }
        tb[T] = (int64_t)ws.size();
        const int64_t nw = (int64_t)ws.size();
        if (nw == 0) return 0;
        if (nw > ctx->wl_cap) {
            if (ctx->wl_start) { NND_HIP_CHECK(hipFree(ctx->wl_start)); ctx->wl_start = nullptr; }
            if (ctx->wl_len) { NND_HIP_CHECK(hipFree(ctx->wl_len)); ctx->wl_len = nullptr; }
            ctx->wl_cap = nw + nw / 4;
            NND_HIP_CHECK(hipMalloc((void **)&ctx->wl_start, sizeof(int32_t) * (size_t)ctx->wl_cap));
            NND_HIP_CHECK(hipMalloc((void **)&ctx->wl_len, sizeof(int32_t) * (size_t)ctx->wl_cap));
        }
        NND_HIP_CHECK(hipMemcpyAsync(ctx->wl_start, ws.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(ctx->wl_len, wl.data(), sizeof(int32_t) * nw, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // ws / wl are about to go out of scope
        d_ws = ctx->wl_start;
        d_wl = ctx->wl_len;
    }
    return run_leaf_rounds(ctx, ctx->perm[ctx->cur], d_ws, d_wl, tb, maxlen);
}


// init_rp_tree on a CALLER-PROVIDED leaf array -- the `leaf_array` argument of the reference's nn_descent
// (pynndescent_.py:324-337; rptree_leaf_array's int32 (n_leaves, max_leaf_size) table, -1 padded; a row ends at its first
// negative entry, pynndescent_.py:88-92).  A caller that keeps the reference's make_forest hands its leaves in here.
// The leaf kernel lets a workgroup own the k-lists of its leaf's points, so the leaves are dealt into ROUNDS in which no
// point occurs twice (round of a leaf = the first round after the last one that used any of its points; a forest's
// leaf array -- trees concatenated, each a partition of the points -- falls into one round per tree) and the rounds run
// launch after launch.  The seeded k-lists are the k nearest leaf-mates of every point over all leaves, as the
// reference's pushes leave them (ties in distance aside).  Leaves longer than 256 points are cut into runs of 256.
int nnd_launch_leaf_init_array(nnd_ctx *ctx, const int32_t *leaf_host, int64_t n_leaves, int32_t max_leaf_size) {
    if (n_leaves <= 0 || max_leaf_size <= 0) return 0;
    const int64_t n = ctx->n;
    std::vector<int32_t> seen((size_t)n, 0);
    struct piece { int64_t off; int32_t len; int32_t round; };
    std::vector<piece> pieces;
    pieces.reserve((size_t)n_leaves);
    int n_rounds = 0, maxlen = 0;
    for (int64_t l = 0; l < n_leaves; l++) {
        const int32_t *row = leaf_host + l * max_leaf_size;
        int32_t len = 0;
        while (len < max_leaf_size && row[len] >= 0) {
            if ((int64_t)row[len] >= n) { ctx->set_error("nnd_init_from_leaf_array: leaf %lld holds point %d but n = %lld", (long long)l, row[len], (long long)n); return 1; }
            len++;
        }
        if (len < 2) continue;  // no pairs
        int32_t r = 0;
        for (int32_t i = 0; i < len; i++) r = seen[row[i]] > r ? seen[row[i]] : r;
        for (int32_t i = 0; i < len; i++) {
            if (seen[row[i]] == r + 1) { ctx->set_error("nnd_init_from_leaf_array: point %d occurs twice in leaf %lld", row[i], (long long)l); return 1; }
            seen[row[i]] = r + 1;
        }
        if (r + 1 > n_rounds) n_rounds = r + 1;
        for (int32_t a = 0; a < len; a += LEAF_MAX) {  // pieces of one leaf share no point: same round
            const int32_t pl = len - a > LEAF_MAX ? LEAF_MAX : len - a;
            if (pl < 2) continue;
            pieces.push_back({l * max_leaf_size + a, pl, r});
            if (pl > maxlen) maxlen = pl;
        }
    }
    if (pieces.empty()) return 0;
    // counting sort by round; members stay where they are in the uploaded table (wl_start indexes into it)
    std::vector<int64_t> tb((size_t)n_rounds + 1, 0);
    for (const piece &pc : pieces) tb[(size_t)pc.round + 1]++;
    for (int r = 0; r < n_rounds; r++) tb[(size_t)r + 1] += tb[(size_t)r];
    std::vector<int64_t> cur(tb.begin(), tb.end() - 1);
    if ((int64_t)n_leaves * max_leaf_size >= (int64_t)0x7FFFFFF0) { ctx->set_error("nnd_init_from_leaf_array: leaf table too large for int32 offsets"); return 1; }
    std::vector<int32_t> ws(pieces.size()), wl(pieces.size());
    for (const piece &pc : pieces) {
        const int64_t at = cur[(size_t)pc.round]++;
        ws[(size_t)at] = (int32_t)pc.off;
        wl[(size_t)at] = pc.len;
    }
    int32_t *d_tab = nullptr, *d_ws = nullptr, *d_wl = nullptr;
    const size_t tab = (size_t)n_leaves * max_leaf_size, np = pieces.size();
    int rc = 0;
    if (hipMalloc((void **)&d_tab, sizeof(int32_t) * tab) != hipSuccess || hipMalloc((void **)&d_ws, sizeof(int32_t) * np) != hipSuccess ||
        hipMalloc((void **)&d_wl, sizeof(int32_t) * np) != hipSuccess) {
        ctx->set_error("nnd_init_from_leaf_array: out of device memory");
        rc = 1;
    }
    if (!rc && (hipMemcpyAsync(d_tab, leaf_host, sizeof(int32_t) * tab, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(d_ws, ws.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                hipMemcpyAsync(d_wl, wl.data(), sizeof(int32_t) * np, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)) {
        ctx->set_error("nnd_init_from_leaf_array: H2D copy failed");
        rc = 1;
    }
    if (!rc) rc = run_leaf_rounds(ctx, d_tab, d_ws, d_wl, tb, maxlen);  // ends with a counter read-back: the stream has drained
    (void)hipStreamSynchronize(ctx->stream);
    if (d_tab) (void)hipFree(d_tab);
    if (d_ws) (void)hipFree(d_ws);
    if (d_wl) (void)hipFree(d_wl);
    if (!rc) ctx->stats.n_leaves = n_leaves;
    return rc;
}
