// rpforest.hip -- random-projection forest, built level-synchronously for ALL trees at once.
//
// Replaces make_forest / make_dense_tree / make_euclidean_tree / make_angular_tree and the
// *_random_projection_split functions (reference rp_trees.py:41-171, 304-420, 2173-2302,
// 2515-2554, 2815-2888) and rptree_leaf_array (rp_trees.py:2891-2922).
//
// The reference recurses per tree (one joblib thread per tree).  Here every tree lives in one
// position space of P = n_trees * n slots: perm[g] is the point at position g, each tree node is a
// contiguous segment of positions, and one level of ALL nodes of ALL trees is processed by a fixed
// sequence of launches:
//   k_hyperplane : one wave per splittable segment: two random members -> hyperplane (+offset)
//   k_margin     : 16 lanes per position: margin = h.x + off -> side bit (coin flip if |m| < 1e-8).
//                  While the hyperplane table of a level fits in L2 (top ~11 levels) the point-major
//                  variant k_margin_fused streams every point row ONCE for all trees of the level
//                  (rows from HBM in order, hyperplanes from L2); deeper levels gather rows in
//                  position order, where neighbours share a hyperplane.
//   scan         : exclusive scan of "goes left" over all positions (3 launches)
//   k_seg_count  : per segment n_left; a one-sided split is replaced by an even split of the
//                  segment's (arbitrarily ordered) members -- the reference re-draws every member by
//                  a fair coin (rp_trees.py:393-403); both cut the node in two near-equal random halves
//   k_children   : child segments, which of them split again, compacted ids, final-leaf marks
//   k_scatter    : stable partition of every segment (left block, then right block)
// Positions stay in depth-first left-to-right order, so the finished permutation IS the leaf
// array: leaves are the maximal runs between leaf marks.  Only the leaves are consumed by the build
// (pynndescent_.py:1130); hyperplanes are discarded level by level.
//
// Random choices come from the counter hash (common.h), not from a sequential Tausworthe stream:
// the forest is statistically, not bitwise, the reference's.
#include "common.h"
#include "state.h"

#define RP_EPS 1e-8f  // rp_trees.py:23

static constexpr int SCAN_ITEMS = 8;
static constexpr int SCAN_BLOCK = 256;
static constexpr int SCAN_TILE = SCAN_ITEMS * SCAN_BLOCK;

// ------------------------------------------------------------------ init --
__global__ void k_forest_init(int32_t *__restrict__ perm, int32_t *__restrict__ pos_seg, uint8_t *__restrict__ leaf_flag,
                              int32_t *__restrict__ inv, int64_t n, int64_t P, int splittable) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    int64_t t = g / n;
    int64_t i = g - t * n;
    perm[g] = (int32_t)i;
    inv[g] = splittable ? (int32_t)t : -1;  // point-major: segment of point i in tree t (here: the root)
    pos_seg[g] = splittable ? (int32_t)t : -1;
    leaf_flag[g] = (!splittable && i == 0) ? 1 : 0;
}
__global__ void k_forest_init_segs(int32_t *__restrict__ seg_start, int32_t *__restrict__ seg_len, int n_trees,
                                   int64_t n) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_trees) {
        seg_start[t] = (int32_t)(t * n);
        seg_len[t] = (int32_t)n;
    }
}

// ------------------------------------------------------------ hyperplane --
// One wave per segment.  euclid (rp_trees.py:350-367): h = x_l - x_r, off = -h.(x_l+x_r)/2.
// angular (rp_trees.py:87-118): h = x_l/|x_l| - x_r/|x_r| normalised, offset 0; xp rows are already
// L2-normalised (zero rows are zero, matching the reference's "norm := 1" for them).
__global__ __launch_bounds__(256) void k_hyperplane(const float *__restrict__ xp, int dp, const int32_t *__restrict__ perm,
                                                    const int32_t *__restrict__ seg_start,
                                                    const int32_t *__restrict__ seg_len, int n_segs, int angular,
                                                    uint32_t seed, int depth, float *__restrict__ hyper, int hs,
                                                    uint16_t *__restrict__ hyper_h) {
    int lane = nnd_lane();
    int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_segs) return;
    int a = seg_start[s], len = seg_len[s];
    uint32_t li = nnd_hash3(seed, (uint32_t)a, (uint32_t)(2 * depth)) % (uint32_t)len;
    uint32_t ri = nnd_hash3(seed, (uint32_t)a, (uint32_t)(2 * depth + 1)) % (uint32_t)len;
    if (ri == li) ri = (ri + 1) % (uint32_t)len;  // rp_trees.py:353-354
    const float *xl = xp + (int64_t)perm[a + li] * dp;
    const float *xr = xp + (int64_t)perm[a + ri] * dp;
    float *h = hyper + (int64_t)s * hs;
    uint16_t *hb = hyper_h + (int64_t)s * dp;  // bf16 copy read by the screening pass of the margin kernels
    float acc = 0.0f, sq = 0.0f;
    for (int j = lane; j < dp; j += 64) {
        float l = xl[j], r = xr[j];
        float v = l - r;
        h[j] = v;
        if (!angular) hb[j] = nnd_f32_to_bf16(v);
        acc += angular ? v * v : v * (l + r);
        sq += v * v;
    }
    acc = nnd_wave_sum_f32(acc);
    sq = nnd_wave_sum_f32(sq);
    if (angular) {
        float nh = sqrtf(acc);
        float inv = nh < RP_EPS ? 1.0f : 1.0f / nh;  // rp_trees.py:113-118
        for (int j = lane; j < dp; j += 64) {
            const float v = h[j] * inv;
            h[j] = v;
            hb[j] = nnd_f32_to_bf16(v);
        }
        if (lane == 0) {
            h[dp] = 0.0f;
            h[dp + 1] = nh * inv;  // |h| after normalisation (1, or |h| itself when degenerate)
        }
    } else if (lane == 0) {
        h[dp] = -0.5f * acc;
        h[dp + 1] = sqrtf(sq);
    }
}

// ---------------------------------------------------------------- margin --
// margin = h . x + off for one point, computed by the 4 lanes of a quad (lane `sub` takes the 16-byte chunks
// sub, sub+4, ... of the row, i.e. 64 contiguous bytes per 4-chunk step).
// Screening pass: row AND hyperplane are read from their bf16 copies (half the bytes) and multiplied with the packed
// v_dot2_f32_bf16 (two products per instruction, no conversions).  bf16 carries 8 significant bits: round-to-nearest
// moves each operand by at most 2^-8 relative, each product by at most (2 * 2^-8 + 2^-16), so
//     |margin_bf16 - margin_f32| <= (2 * 2^-8 + 2^-16) * sum|h_i x_i|  <=  RP_BAND * |h| |x|      (Cauchy-Schwarz)
// with RP_BAND also covering the f32 accumulation-order difference (<= d * 2^-24 relative).  If the screened margin is
// further from zero than that, its SIGN is already the f32 sign and the f32 row is never touched.  The points inside
// the band (a few percent at the top of a tree, more in small dense nodes) are recomputed from the f32 row and the
// f32 hyperplane, so the split is exactly the f32 split.
#define RP_BAND 0.00786f
typedef __attribute__((ext_vector_type(2))) __bf16 rp_bf16x2;
__device__ __forceinline__ float rp_dot8(uint4 q, uint4 p, float acc) {
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2, q.x), __builtin_bit_cast(rp_bf16x2, p.x), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2, q.y), __builtin_bit_cast(rp_bf16x2, p.y), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2, q.z), __builtin_bit_cast(rp_bf16x2, p.z), acc, false);
    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(rp_bf16x2, q.w), __builtin_bit_cast(rp_bf16x2, p.w), acc, false);
    return acc;
}
// sum over the 4 lanes of an aligned quad (DPP quad permutes; every lane ends with the same value)
__device__ __forceinline__ float rp_quad_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
    return v;
}
// exact f32 margin (without the offset) of one point by its quad
__device__ __forceinline__ float rp_exact_quad(const float *__restrict__ xf_row, const float *h, int dp, int sub) {
    float acc = 0.0f;
    const float4 *x4 = (const float4 *)xf_row;
    const float4 *h4 = (const float4 *)h;
    for (int c = sub; c < (dp >> 2); c += 4) {
        const float4 a = x4[c], b = h4[c];
        acc += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    }
    return rp_quad_sum(acc);
}
// side of the split from a screened margin; `key` feeds the coin flip of rp_trees.py:380-385
__device__ __forceinline__ uint8_t rp_side(float m, float band, const float *__restrict__ xf_row, const float *h, float off,
                                           int dp, int sub, uint32_t seed, uint32_t key, int depth) {
    if (!(fabsf(m) > band)) m = rp_exact_quad(xf_row, h, dp, sub) + off;  // uniform inside the quad
    if (fabsf(m) < RP_EPS) return (uint8_t)(nnd_hash3(seed ^ 0x5bd1e995u, key, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
    return m > 0.0f ? 0 : 1;                                                                              // rp_trees.py:386-391
}

// position-major: neighbouring positions share a hyperplane; rows are gathered through perm
__global__ __launch_bounds__(256) void k_margin(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                const float *__restrict__ nrm, int metric, int dp,
                                                const int32_t *__restrict__ perm, const int32_t *__restrict__ pos_seg,
                                                int64_t P, const float *__restrict__ hyper, int hs,
                                                const uint16_t *__restrict__ hyper_h, uint32_t seed, int depth,
                                                uint8_t *__restrict__ side) {
    const int sub = threadIdx.x & 3;
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const int s = g < P ? pos_seg[g] : -1;
    if (s < 0) return;  // whole quad
    const int64_t pt = perm[g];
    const uint4 *x8 = (const uint4 *)(xh + pt * dp);
    const uint4 *h8 = (const uint4 *)(hyper_h + (int64_t)s * dp);
    const float *h = hyper + (int64_t)s * hs;
    const float xn = nrm[pt], off = h[dp], hnorm = h[dp + 1];
    float acc = 0.0f;
    for (int c = sub; c < (dp >> 3); c += 16) {  // 4 chunks per lane and step: the 8 loads are issued together
        uint4 q[4], p[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int cc = c + 4 * j < (dp >> 3) ? c + 4 * j : c;
            q[j] = x8[cc];
            p[j] = h8[cc];
        }
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (c + 4 * j < (dp >> 3)) acc = rp_dot8(q[j], p[j], acc);
    }
    const float m = rp_quad_sum(acc) + off;
    const float band = RP_BAND * hnorm * (metric == 0 ? sqrtf(xn) : xn) + 1e-30f;
    const uint8_t sd = rp_side(m, band, xp + pt * dp, h, off, dp, sub, seed, (uint32_t)g, depth);
    if (sub == 0) side[g] = sd;
}

// point-major variant: one pass over the points serves every tree (rows read once per level).  Everything it touches
// is point-major too -- seg_pt[t*n + i] = the point's segment in tree t (-1 once its segment is final), the side goes
// to side_pt[t*n + i] -- so apart from the hyperplane look-ups (a table that sits in L2) all its traffic is sequential.
// The scan that follows brings the sides into position order (k_scan_reduce mode 2).
__global__ __launch_bounds__(256) void k_margin_fused(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                      const float *__restrict__ nrm, int metric, int dp, int64_t n,
                                                      int n_trees, const int32_t *__restrict__ seg_pt,
                                                      const float *__restrict__ hyper, int hs,
                                                      const uint16_t *__restrict__ hyper_h, uint32_t seed, int depth,
                                                      uint8_t *__restrict__ side_pt) {
    const int sub = threadIdx.x & 3;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    if (i >= n) return;  // whole quad
    const uint4 *x8 = (const uint4 *)(xh + i * dp);
    const float xn = nrm[i];
    const float xnorm = metric == 0 ? sqrtf(xn) : xn;
    const int nch = dp >> 3;
    // trees in batches of 4: segment ids, then hyperplane chunks of the whole batch, are independent loads issued together
    for (int t0 = 0; t0 < n_trees; t0 += 4) {
        int sg[4];
        float acc[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            sg[u] = t0 + u < n_trees ? seg_pt[(int64_t)(t0 + u) * n + i] : -1;
            acc[u] = 0.0f;
        }
        if (sg[0] < 0 && sg[1] < 0 && sg[2] < 0 && sg[3] < 0) continue;  // whole quad
        for (int c = sub; c < nch; c += 16) {
            uint4 q[4];
#pragma unroll
            for (int j = 0; j < 4; j++) q[j] = x8[c + 4 * j < nch ? c + 4 * j : c];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint4 *h8 = (const uint4 *)(hyper_h + (int64_t)(sg[u] >= 0 ? sg[u] : 0) * dp);
                uint4 p[4];
#pragma unroll
                for (int j = 0; j < 4; j++) p[j] = h8[c + 4 * j < nch ? c + 4 * j : c];
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (c + 4 * j < nch) acc[u] = rp_dot8(q[j], p[j], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (sg[u] < 0) continue;  // whole quad
            const float *h = hyper + (int64_t)sg[u] * hs;
            const float off = h[dp], hnorm = h[dp + 1];
            const float m = rp_quad_sum(acc[u]) + off;
            const float band = RP_BAND * hnorm * xnorm + 1e-30f;
            const int64_t slot = (int64_t)(t0 + u) * n + i;
            const uint8_t sd = rp_side(m, band, xp + i * dp, h, off, dp, sub, seed, (uint32_t)slot, depth);
            if (sub == 0) side_pt[slot] = sd;
        }
    }
}

// ------------------------------------------------------------------ scan --
// exclusive scan over flag(g) = (pos_seg[g] >= 0 && side[g] == 0)  [mode 0]  or  leaf_flag[g] [mode 1]
__device__ __forceinline__ int scan_flag(int mode, const int32_t *pos_seg, const uint8_t *bytes, int64_t g, int64_t P) {
    if (g >= P) return 0;
    if (mode == 0) {  // both loads are issued (no short circuit): the unrolled callers then have 2 * SCAN_ITEMS loads in flight
        const int sg = pos_seg[g];
        const uint8_t b = bytes[g];
        return (sg >= 0) & (b == 0);
    }
    return bytes[g] ? 1 : 0;
}

// mode 2 = mode 0 after a point-major margin pass: the side of position g is side_pt[tree(g)*n + perm[g]]; it is
// gathered here once and stored to bytes[g] (position order), which k_scan_apply and k_scatter then read as in mode 0.
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_reduce(int mode, const int32_t *__restrict__ pos_seg,
                                                            uint8_t *__restrict__ bytes, int64_t P,
                                                            int32_t *__restrict__ blk, const int32_t *__restrict__ perm,
                                                            const uint8_t *__restrict__ side_pt, int64_t n) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
    if (mode == 2) {
        // three rounds of independent loads (segment ids, point ids, sides) instead of a dependent chain per item
        const int64_t tb0 = base < P ? (base / n) * n : 0;
        int sg[SCAN_ITEMS], pt[SCAN_ITEMS];
        uint8_t sd[SCAN_ITEMS];
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const int64_t g = base + i < P ? base + i : P - 1;
            sg[i] = pos_seg[g];
            pt[i] = perm[g];
        }
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const int64_t g = base + i;
            int64_t tb = tb0;
            while (g >= tb + n && tb + n < P) tb += n;  // a run of SCAN_ITEMS positions rarely crosses a tree boundary
            sd[i] = side_pt[tb + pt[i]];
        }
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) {
            const int64_t g = base + i;
            if (g < P && sg[i] >= 0) {
                bytes[g] = sd[i];
                s += sd[i] == 0;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_ITEMS; i++) s += scan_flag(mode, pos_seg, bytes, base + i, P);
    }
    s = nnd_wave_sum_i32(s);
    if (nnd_lane() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) t += wsum[w];
        blk[blockIdx.x] = t;
    }
}

// single block: exclusive scan of blk[0..nb) in place; total -> total_out[0]
__global__ __launch_bounds__(256) void k_scan_blocks(int32_t *__restrict__ blk, int nb, int32_t *__restrict__ total_out) {
    __shared__ int part[256];
    int chunk = (nb + 255) / 256;
    int b0 = threadIdx.x * chunk, b1 = b0 + chunk < nb ? b0 + chunk : nb;
    int s = 0;
    for (int b = b0; b < b1; b++) s += blk[b];
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int i = 0; i < 256; i++) {
            int v = part[i];
            part[i] = run;
            run += v;
        }
        total_out[0] = run;
    }
    __syncthreads();
    int run = part[threadIdx.x];
    for (int b = b0; b < b1; b++) {
        int v = blk[b];
        blk[b] = run;
        run += v;
    }
}

__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(int mode, const int32_t *__restrict__ pos_seg,
                                                           const uint8_t *__restrict__ bytes, int64_t P,
                                                           const int32_t *__restrict__ blk, int32_t *__restrict__ out) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    int lane = nnd_lane(), w = threadIdx.x >> 6;
    int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS];
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        f[i] = scan_flag(mode, pos_seg, bytes, base + i, P);
        s += f[i];
    }
    // inclusive scan of per-thread sums inside the wave
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int i = 0; i < w; i++) woff += wsum[i];
    int run = blk[blockIdx.x] + woff + incl - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < P) out[base + i] = run;
        run += f[i];
    }
}

// ------------------------------------------------------------- per segment --
// n_left from the scan.  A one-sided split (rp_trees.py:393-403) is encoded as nleft = -(ceil(len/2)) - 1:
// k_children / k_scatter then send the members at even offsets left and those at odd offsets right.
__global__ void k_seg_count(const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_len, int n_segs,
                            const int32_t *__restrict__ scan, const int32_t *__restrict__ scan_total, int64_t P,
                            int32_t *__restrict__ nleft) {
    int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_segs) return;
    int a = seg_start[s], len = seg_len[s];
    int64_t e = (int64_t)a + len;
    int hi = e < P ? scan[e] : scan_total[0];
    int nl = hi - scan[a];
    if (nl == 0 || nl == len) nl = -((len + 1) / 2) - 1;
    nleft[s] = nl;
}

// single block: children of every segment -> next level's segment list (compacted), child ids, leaf marks
__global__ __launch_bounds__(256) void k_children(const int32_t *__restrict__ seg_start, const int32_t *__restrict__ seg_len,
                                                  const int32_t *__restrict__ nleft, int n_segs, int leaf_size,
                                                  int child_can_split, int fin_max, int child_depth,
                                                  int32_t *__restrict__ next_start, int32_t *__restrict__ next_len,
                                                  int32_t *__restrict__ seg_child, uint8_t *__restrict__ leaf_flag,
                                                  int32_t *__restrict__ fin_start, int32_t *__restrict__ fin_len,
                                                  int32_t *__restrict__ fin_depth, long long *__restrict__ counters) {
    // a child that splits again either stays in the level-synchronous passes (len > fin_max) or is handed to
    // k_finish_subtrees (len <= fin_max: its whole subtree fits in one workgroup's LDS)
    __shared__ int part[256], partf[256];
    int chunk = (n_segs + 255) / 256;
    int s0 = threadIdx.x * chunk, s1 = s0 + chunk < n_segs ? s0 + chunk : n_segs;
    int cnt = 0, cntf = 0;
    for (int s = s0; s < s1; s++) {
        int len = seg_len[s], nl = nleft[s];
        if (nl < 0) nl = -nl - 1;
        int lens[2] = {nl, len - nl};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (child_can_split && lens[c] > leaf_size) {
                if (lens[c] > fin_max) cnt++; else cntf++;
            }
        }
    }
    part[threadIdx.x] = cnt;
    partf[threadIdx.x] = cntf;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0, runf = (int)counters[CNT_SCRATCH + 1];  // finisher list grows across levels
        for (int i = 0; i < 256; i++) {
            int v = part[i]; part[i] = run; run += v;
            int vf = partf[i]; partf[i] = runf; runf += vf;
        }
        counters[CNT_ACTIVE_SEGS] = run;
        counters[CNT_SCRATCH + 1] = runf;
    }
    __syncthreads();
    int run = part[threadIdx.x], runf = partf[threadIdx.x];
    long long active_pos = 0;
    int max_stay = 0;
    for (int s = s0; s < s1; s++) {
        int a = seg_start[s], len = seg_len[s], nl = nleft[s];
        if (nl < 0) nl = -nl - 1;
        int lens[2] = {nl, len - nl};
        int starts[2] = {a, a + nl};
#pragma unroll
        for (int c = 0; c < 2; c++) {
            if (child_can_split && lens[c] > leaf_size) {  // rp_trees.py:2188
                if (lens[c] > fin_max) {
                    next_start[run] = starts[c];
                    next_len[run] = lens[c];
                    seg_child[2 * s + c] = run++;
                    active_pos += lens[c];
                    if (lens[c] > max_stay) max_stay = lens[c];
                } else {
                    fin_start[runf] = starts[c];
                    fin_len[runf] = lens[c];
                    fin_depth[runf] = child_depth;
                    runf++;
                    seg_child[2 * s + c] = -1;  // leaves the level-synchronous passes
                }
            } else {
                seg_child[2 * s + c] = -1;
                if (lens[c] > 0) leaf_flag[starts[c]] = 1;  // rp_trees.py:2229-2232
            }
        }
    }
    if (active_pos) atomicAdd((unsigned long long *)&counters[CNT_LEAVES], (unsigned long long)active_pos);  // positions still in the passes
    if (max_stay) atomicMax((unsigned long long *)&counters[CNT_SCRATCH + 3], (unsigned long long)max_stay);  // longest of them
}

// stable partition (rp_trees.py:405-418): lefts keep their order at the front, rights behind them
__global__ void k_scatter(const int32_t *__restrict__ perm, const int32_t *__restrict__ pos_seg,
                          const uint8_t *__restrict__ side, const int32_t *__restrict__ scan,
                          const int32_t *__restrict__ seg_start, const int32_t *__restrict__ nleft,
                          const int32_t *__restrict__ seg_child, int64_t P, int64_t n, int32_t *__restrict__ perm_out,
                          int32_t *__restrict__ pos_seg_out, int32_t *__restrict__ inv) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= P) return;
    int s = pos_seg[g];
    if (s < 0) {
        perm_out[g] = perm[g];
        pos_seg_out[g] = -1;
        return;
    }
    int a = seg_start[s];
    int nl = nleft[s];
    int right;
    int64_t dest;
    if (nl < 0) {  // one-sided split: even offsets left, odd offsets right
        nl = -nl - 1;
        int off = (int)(g - a);
        right = off & 1;
        dest = right ? (int64_t)a + nl + (off >> 1) : (int64_t)a + (off >> 1);
    } else {
        int L = scan[g] - scan[a];
        right = side[g];
        dest = right ? (int64_t)a + nl + ((int)(g - a) - L) : (int64_t)a + L;
    }
    int32_t p = perm[g];
    perm_out[dest] = p;
    pos_seg_out[dest] = seg_child[2 * s + right];
    if (inv) inv[(g / n) * n + p] = seg_child[2 * s + right];  // point-major segment table of the next level
}

// ------------------------------------------------------------ subtree finisher --
// Once every splittable segment fits in LDS (<= FIN_MAX points) the level-synchronous passes stop and one
// workgroup per segment finishes its whole subtree on its own: explicit stack of sub-segments, hyperplane and
// member ids in LDS, margins by 16-lane groups, stable partition by a block-wide scan.  No global
// synchronisation, no per-level launches; deep unbalanced branches only cost their own workgroup.
// Same split rule, same hashes (position- and depth-keyed) as the level-synchronous kernels above.
#ifndef NND_FIN_MAX
#define NND_FIN_MAX 2048
#endif
static constexpr int FIN_MAX = NND_FIN_MAX;     // points per finisher segment
#ifndef NND_BIG_MAX
#define NND_BIG_MAX 8192
#endif
static constexpr int BIG_MAX = NND_BIG_MAX;     // longest segment the global-memory finisher variant takes (one workgroup each)
static constexpr int FIN_STACK = 512;    // sub-segments pending (depth budget is 200: a DFS needs <= depth+1 entries)

// BIG = true: the same node loop for the FEW segments between FIN_MAX and BIG_MAX points that are left when the
// level-synchronous passes stop paying (most positions already handed over): member ids, partition scratch and side
// bits live in global memory (perm itself, the other perm buffer, side[]), and a node that has shrunk to <= FIN_MAX
// points is appended to the LDS finisher's work list instead of being split here.
template <bool BIG>
__global__ __launch_bounds__(256) void k_finish_subtrees(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                         const float *__restrict__ nrm, int metric, int dp,
                                                         int32_t *__restrict__ perm,
                                                         const int32_t *__restrict__ seg_start,
                                                         const int32_t *__restrict__ seg_len,
                                                         const int32_t *__restrict__ seg_depth, int depth0, int n_segs,
                                                         int angular, uint32_t seed, int max_depth, int leaf_size,
                                                         uint8_t *__restrict__ leaf_flag, int32_t *__restrict__ tmp_g,
                                                         uint8_t *__restrict__ side_g, int fin_max,
                                                         int32_t *__restrict__ fin_start, int32_t *__restrict__ fin_len,
                                                         int32_t *__restrict__ fin_depth, long long *__restrict__ fin_count) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int s = blockIdx.x;
    if (s >= n_segs) return;
    const int a = seg_start[s], len = seg_len[s];
    constexpr int NLDS = BIG ? 0 : FIN_MAX;
    int32_t *ids = BIG ? perm + a : (int32_t *)fsm;               // member ids of the segment
    int32_t *tmp = BIG ? tmp_g + a : (int32_t *)fsm + NLDS;       // partition scratch
    uint8_t *sd = BIG ? side_g + a : (uint8_t *)((int32_t *)fsm + 2 * NLDS);  // side bits
    float *h = (float *)(fsm + (size_t)NLDS * 9);  // dp + 4 hyperplane + offset
    uint16_t *hb = (uint16_t *)(h + dp + 4);       // dp: bf16 copy of the normal (dp is a multiple of 32)
    int32_t *stk = (int32_t *)(hb + dp);           // FIN_STACK * 3: (start, len, depth)
    int32_t *wsum = stk + FIN_STACK * 3;           // 8: per-wave partial sums / scalars
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    if (!BIG)
        for (int i = tid; i < len; i += 256) ids[i] = perm[a + i];
    if (tid == 0) {
        stk[0] = 0; stk[1] = len; stk[2] = seg_depth ? seg_depth[s] : depth0;
        wsum[7] = 1;  // stack size
    }
    __syncthreads();
    while (true) {
        const int sp = wsum[7];
        if (sp == 0) break;
        const int ss = stk[(sp - 1) * 3], l = stk[(sp - 1) * 3 + 1], dep = stk[(sp - 1) * 3 + 2];
        __syncthreads();
        if (tid == 0) wsum[7] = sp - 1;
        if (!(l > leaf_size && (max_depth - dep) > 0)) {  // rp_trees.py:2188: this node is a leaf
            if (tid == 0 && l > 0) leaf_flag[a + ss] = 1;
            __syncthreads();
            continue;
        }
        if (BIG && l <= fin_max) {  // small enough for the LDS finisher: hand it over
            if (tid == 0) {
                const int idx = (int)atomicAdd((unsigned long long *)fin_count, 1ull);
                fin_start[idx] = a + ss;
                fin_len[idx] = l;
                fin_depth[idx] = dep;
            }
            __syncthreads();
            continue;
        }
        // two random members -> hyperplane (rp_trees.py:350-367 / 87-118); hashes keyed like k_hyperplane
        const uint32_t gpos = (uint32_t)(a + ss);
        uint32_t li = nnd_hash3(seed, gpos, (uint32_t)(2 * dep)) % (uint32_t)l;
        uint32_t ri = nnd_hash3(seed, gpos, (uint32_t)(2 * dep + 1)) % (uint32_t)l;
        if (ri == li) ri = (ri + 1) % (uint32_t)l;
        const float *xl = xp + (int64_t)ids[ss + li] * dp;
        const float *xr = xp + (int64_t)ids[ss + ri] * dp;
        float part = 0.0f, psq = 0.0f;
        for (int j = tid; j < dp; j += 256) {
            const float lv = xl[j], rv = xr[j];
            const float v = lv - rv;
            h[j] = v;
            part += angular ? v * v : v * (lv + rv);
            psq += v * v;
        }
        part = nnd_wave_sum_f32(part);
        psq = nnd_wave_sum_f32(psq);
        if (lane == 0) {
            ((float *)wsum)[w] = part;
        }
        __syncthreads();
        const float tot = ((float *)wsum)[0] + ((float *)wsum)[1] + ((float *)wsum)[2] + ((float *)wsum)[3];
        __syncthreads();
        if (lane == 0) ((float *)wsum)[w] = psq;
        __syncthreads();
        const float totsq = ((float *)wsum)[0] + ((float *)wsum)[1] + ((float *)wsum)[2] + ((float *)wsum)[3];
        __syncthreads();
        if (angular) {  // normalise in place (rp_trees.py:113-118); offset 0
            const float nh = sqrtf(tot);
            const float inv = nh < RP_EPS ? 1.0f : 1.0f / nh;
            for (int j = tid; j < dp; j += 256) h[j] *= inv;
            if (tid == 0) { h[dp] = 0.0f; h[dp + 1] = nh * inv; }
        } else if (tid == 0) {
            h[dp] = -0.5f * tot;
            h[dp + 1] = sqrtf(totsq);
        }
        __syncthreads();
        // margins: one quad per member (bf16-screened like k_margin), two members per quad and step so that 8 row
        // fetches are in flight per lane; the bf16 hyperplane comes from LDS
        for (int j = tid; j < dp; j += 256) hb[j] = nnd_f32_to_bf16(h[j]);
        __syncthreads();
        const int sub = tid & 3, grp = tid >> 2;
        const int nch = dp >> 3;
        const float off = h[dp], hnorm = h[dp + 1];
        const uint4 *h8 = (const uint4 *)hb;
        for (int i0 = 0; i0 < l; i0 += 128) {
            int64_t pt[2];
            float acc[2], xn[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * 64 + grp;
                pt[u] = ids[ss + (i < l ? i : 0)];
                acc[u] = 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 2; u++) xn[u] = nrm[pt[u]];
            for (int c = sub; c < nch; c += 16) {
                uint4 q[2][4], p[4];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++) q[u][j] = ((const uint4 *)(xh + pt[u] * dp))[c + 4 * j < nch ? c + 4 * j : c];
#pragma unroll
                for (int j = 0; j < 4; j++) p[j] = h8[c + 4 * j < nch ? c + 4 * j : c];
#pragma unroll
                for (int u = 0; u < 2; u++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (c + 4 * j < nch) acc[u] = rp_dot8(q[u][j], p[j], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * 64 + grp;
                if (i >= l) continue;  // whole quad
                const float m = rp_quad_sum(acc[u]) + off;
                const float band = RP_BAND * hnorm * (metric == 0 ? sqrtf(xn[u]) : xn[u]) + 1e-30f;
                const uint8_t side = rp_side(m, band, xp + pt[u] * dp, h, off, dp, sub, seed, gpos + (uint32_t)i, dep);
                if (sub == 0) sd[i] = side;
            }
        }
        __syncthreads();
        // stable partition: block-wide exclusive scan of "left" over l <= FIN_MAX members (16 per thread)
        const int per = (l + 255) / 256;
        const int b0 = tid * per, b1 = b0 + per < l ? b0 + per : l;
        int cntl = 0;
        for (int i = b0; i < b1; i++) cntl += sd[i] == 0;
        int incl = cntl;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) wsum[w] = incl;
        __syncthreads();
        int woff = 0;
        for (int i = 0; i < w; i++) woff += wsum[i];
        int nl = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        const bool one_sided = (nl == 0 || nl == l);  // rp_trees.py:393-403 -> even split by offset parity
        if (one_sided) nl = (l + 1) / 2;
        int run = woff + incl - cntl;
        for (int i = b0; i < b1; i++) {
            int dest;
            if (one_sided) dest = (i & 1) ? nl + (i >> 1) : (i >> 1);
            else if (sd[i] == 0) dest = run++;
            else dest = nl + (i - run);
            tmp[dest] = ids[ss + i];
        }
        __syncthreads();
        for (int i = tid; i < l; i += 256) ids[ss + i] = tmp[i];
        if (tid == 0) {  // right child first so that the left one is processed next (order is immaterial)
            int top = wsum[7];
            stk[top * 3] = ss + nl; stk[top * 3 + 1] = l - nl; stk[top * 3 + 2] = dep + 1;
            top++;
            stk[top * 3] = ss; stk[top * 3 + 1] = nl; stk[top * 3 + 2] = dep + 1;
            wsum[7] = top + 1;
        }
        __syncthreads();
    }
    if (!BIG)
        for (int i = tid; i < len; i += 256) perm[a + i] = ids[i];
}

// ------------------------------------------------------------ leaf tables --
__global__ void k_leaf_starts(const uint8_t *__restrict__ leaf_flag, const int32_t *__restrict__ scan, int64_t P,
                              int32_t *__restrict__ leaf_start) {
    int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < P && leaf_flag[g]) leaf_start[scan[g]] = (int32_t)g;
}
__global__ void k_leaf_lens(const int32_t *__restrict__ leaf_start, int64_t n_leaves, int64_t n, int64_t P,
                            int32_t *__restrict__ leaf_len, int32_t *__restrict__ max_len) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int len = 0;
    if (i < n_leaves) {
        int64_t a = leaf_start[i];
        int64_t e = i + 1 < n_leaves ? leaf_start[i + 1] : P;
        int64_t tree_end = (a / n + 1) * n;  // leaves never cross a tree boundary
        if (e > tree_end) e = tree_end;
        len = (int)(e - a);
        leaf_len[i] = len;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(len, o, 64);
        len = other > len ? other : len;
    }
    if (nnd_lane() == 0 && len > 0) atomicMax(max_len, len);  // one atomic per wave
}
// first leaf index of every tree = the exclusive leaf-flag scan at the tree's first position
__global__ void k_tree_leaf_begin(const int32_t *__restrict__ scan, int n_trees, int64_t n, long long *__restrict__ out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n_trees) out[t] = scan[(int64_t)t * n];
}
__global__ void k_fill_leaf_array(const int32_t *__restrict__ perm, const int32_t *__restrict__ leaf_start,
                                  const int32_t *__restrict__ leaf_len, int64_t n_leaves, int max_leaf,
                                  int32_t *__restrict__ out) {
    int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_leaves * max_leaf) return;
    int64_t i = t / max_leaf;
    int j = (int)(t - i * max_leaf);
    out[t] = j < leaf_len[i] ? perm[leaf_start[i] + j] : -1;
}

// -------------------------------------------------------------- host side --
static int run_scan(nnd_ctx *ctx, int mode, const int32_t *pos_seg, uint8_t *bytes, int32_t *total_dev,
                    const int32_t *perm = nullptr) {
    int64_t P = ctx->P;
    int nb = (int)((P + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, mode, pos_seg, bytes, P, ctx->scan_blk, perm,
                       ctx->side_pt, ctx->n);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, ctx->scan_blk, nb, total_dev);
    hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, mode == 2 ? 0 : mode, pos_seg, bytes, P,
                       ctx->scan_blk, ctx->scan_out);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

int nnd_launch_forest(nnd_ctx *ctx) {
    const int64_t n = ctx->n, P = ctx->P;
    const int T = ctx->p.n_trees, dp = ctx->dp, leaf_size = ctx->p.leaf_size, max_depth = ctx->p.max_depth;
    const int angular = ctx->p.metric == NND_METRIC_ALT_COSINE;
    const int hs = dp + 4;
    ctx->forest_built = false;
    ctx->n_leaves = 0;
    ctx->max_leaf = leaf_size;
    ctx->tree_leaf_begin.clear();
    if (T <= 0) return 0;
    if (P >= (int64_t)0x7FFFFFF0) {
        ctx->set_error("n_trees * n = %lld exceeds the int32 position space", (long long)P);
        return 1;
    }
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);  // device scratch word(s)
    int splittable = (n > leaf_size && max_depth > 0) ? 1 : 0;
    int cur = 0;
    unsigned gridP = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(k_forest_init, dim3(gridP), dim3(256), 0, ctx->stream, ctx->perm[0], ctx->pos_seg[0],
                       ctx->leaf_flag, ctx->inv, n, P, splittable);
    hipLaunchKernelGGL(k_forest_init_segs, dim3((T + 63) / 64), dim3(64), 0, ctx->stream, ctx->seg_start[0],
                       ctx->seg_len[0], T, n);
    int64_t S = splittable ? T : 0;
    int depth = 0;
    bool inv_live = true;  // inv[] is maintained while the point-major margin kernel is in use
    long long active_pos = P;
    const int fin_max = FIN_MAX;
    const size_t fin_smem = sizeof(int32_t) * 2 * FIN_MAX + FIN_MAX + sizeof(float) * (dp + 4) + sizeof(uint16_t) * dp +
                            sizeof(int32_t) * (FIN_STACK * 3 + 8);
    const size_t fin_smem_big = sizeof(float) * (dp + 4) + sizeof(uint16_t) * dp + sizeof(int32_t) * (FIN_STACK * 3 + 8);
    int32_t *fin_start = ctx->seg_child + 2 * ctx->max_segs;  // finisher work list lives behind seg_child
    int32_t *fin_len = fin_start + ctx->max_segs;
    int32_t *fin_depth = fin_len + ctx->max_segs;
    NND_HIP_CHECK(hipMemsetAsync(ctx->counters + CNT_SCRATCH + 1, 0, sizeof(long long), ctx->stream));
    if (S > 0 && n <= fin_max) {  // small point sets: the roots go straight to the finisher
        std::vector<int32_t> hs(T), hl(T), hd(T, 0);
        for (int t = 0; t < T; t++) { hs[t] = (int32_t)(t * n); hl[t] = (int32_t)n; }
        NND_HIP_CHECK(hipMemcpyAsync(fin_start, hs.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(fin_len, hl.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(fin_depth, hd.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        long long cntf = T;
        NND_HIP_CHECK(hipMemcpyAsync(ctx->counters + CNT_SCRATCH + 1, &cntf, sizeof(long long), hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        S = 0;
    }
    while (S > 0) {
        if (S > ctx->max_segs) {
            ctx->set_error("rp-forest: %lld segments exceed the allocation of %lld", (long long)S, (long long)ctx->max_segs);
            return 1;
        }
        hipLaunchKernelGGL(k_hyperplane, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, ctx->stream, ctx->xp, dp,
                           ctx->perm[cur], ctx->seg_start[cur], ctx->seg_len[cur], (int)S, angular, ctx->tree_seed, depth,
                           ctx->hyper, hs, ctx->hyper_h);
        // point-major pass: hyperplane table fits in L2 AND enough positions are still active to amortise
        // streaming every row once (it costs n rows regardless of how many positions are active)
        const bool fused = inv_live && (S * (int64_t)dp * 2 <= (int64_t)6 << 20) && (active_pos * 2 >= 3 * n);
        if (fused) {
            hipLaunchKernelGGL(k_margin_fused, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ctx->stream, ctx->xp, ctx->xh, ctx->nrm,
                               ctx->p.metric, dp, n, T, ctx->inv, ctx->hyper, hs, ctx->hyper_h, ctx->tree_seed, depth, ctx->side_pt);
        } else {
            inv_live = false;
            hipLaunchKernelGGL(k_margin, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, ctx->stream, ctx->xp, ctx->xh, ctx->nrm,
                               ctx->p.metric, dp, ctx->perm[cur], ctx->pos_seg[cur], P, ctx->hyper, hs, ctx->hyper_h, ctx->tree_seed, depth,
                               ctx->side);
        }
        if (run_scan(ctx, fused ? 2 : 0, ctx->pos_seg[cur], ctx->side, scan_total, ctx->perm[cur])) return 1;
        hipLaunchKernelGGL(k_seg_count, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, ctx->seg_start[cur],
                           ctx->seg_len[cur], (int)S, ctx->scan_out, scan_total, P, ctx->seg_nleft);
        int child_can_split = (max_depth - (depth + 1)) > 0 ? 1 : 0;
        NND_HIP_CHECK(hipMemsetAsync(ctx->counters + CNT_LEAVES, 0, sizeof(long long), ctx->stream));
        NND_HIP_CHECK(hipMemsetAsync(ctx->counters + CNT_SCRATCH + 3, 0, sizeof(long long), ctx->stream));
        hipLaunchKernelGGL(k_children, dim3(1), dim3(256), 0, ctx->stream, ctx->seg_start[cur], ctx->seg_len[cur],
                           ctx->seg_nleft, (int)S, leaf_size, child_can_split, fin_max, depth + 1, ctx->seg_start[1 - cur],
                           ctx->seg_len[1 - cur], ctx->seg_child, ctx->leaf_flag, fin_start, fin_len, fin_depth, ctx->counters);
        hipLaunchKernelGGL(k_scatter, dim3(gridP), dim3(256), 0, ctx->stream, ctx->perm[cur], ctx->pos_seg[cur], ctx->side,
                           ctx->scan_out, ctx->seg_start[cur], ctx->seg_nleft, ctx->seg_child, P, n, ctx->perm[1 - cur],
                           ctx->pos_seg[1 - cur], inv_live ? ctx->inv : (int32_t *)nullptr);
        NND_HIP_CHECK(hipGetLastError());
        // one small read-back per level: the number of segments that stay in the level-synchronous passes
        long long *next = ctx->h_pin + 32;  // CNT_ACTIVE_SEGS, CNT_LEAVES, CNT_SCRATCH.. are adjacent; pinned words
        static_assert(CNT_LEAVES == CNT_ACTIVE_SEGS + 1 && CNT_SCRATCH == CNT_LEAVES + 1, "counter layout");
        NND_HIP_CHECK(hipMemcpyAsync(next, ctx->counters + CNT_ACTIVE_SEGS, 6 * sizeof(long long), hipMemcpyDeviceToHost,
                                     ctx->stream));
        NND_HIP_CHECK(nnd_sync_spin(ctx));
        S = next[0];
        active_pos = next[1];
        const long long max_stay = next[5];  // CNT_SCRATCH + 3
        cur = 1 - cur;
        depth++;
        // Tail of the level loop: once most positions have been handed over, a level-synchronous pass still costs P
        // positions per kernel for a few hundred segments.  If what is left fits the global-memory variant of the
        // finisher (every segment <= BIG_MAX), one launch finishes the tail: its nodes are split in place until they fit
        // the LDS finisher, whose work list they join.
        if (S > 0 && (active_pos * 2 < 3 * n) && max_stay <= BIG_MAX) {
            hipLaunchKernelGGL(k_finish_subtrees<true>, dim3((unsigned)S), dim3(256), fin_smem_big, ctx->stream, ctx->xp, ctx->xh,
                               ctx->nrm, ctx->p.metric, dp, ctx->perm[cur], ctx->seg_start[cur], ctx->seg_len[cur],
                               (const int32_t *)nullptr, depth, (int)S, angular, ctx->tree_seed, max_depth, leaf_size, ctx->leaf_flag,
                               ctx->perm[1 - cur], ctx->side, fin_max, fin_start, fin_len, fin_depth, ctx->counters + CNT_SCRATCH + 1);
            NND_HIP_CHECK(hipGetLastError());
            S = 0;
        }
    }
    {  // subtrees that fit in LDS: one workgroup each, no more global passes
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 34, ctx->counters + CNT_SCRATCH + 1, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        NND_HIP_CHECK(nnd_sync_spin(ctx));
        const long long nfin = ctx->h_pin[34];
        if (nfin > ctx->max_segs) {
            ctx->set_error("rp-forest: %lld finisher segments exceed the allocation of %lld", nfin, (long long)ctx->max_segs);
            return 1;
        }
        if (nfin > 0) {
            static bool configured = false;
            if (!configured) {
                NND_HIP_CHECK(hipFuncSetAttribute((const void *)k_finish_subtrees<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  (int)fin_smem));
                configured = true;
            }
            hipLaunchKernelGGL(k_finish_subtrees<false>, dim3((unsigned)nfin), dim3(256), fin_smem, ctx->stream, ctx->xp, ctx->xh, ctx->nrm,
                               ctx->p.metric, dp, ctx->perm[cur], fin_start, fin_len, fin_depth, 0, (int)nfin, angular, ctx->tree_seed,
                               max_depth, leaf_size, ctx->leaf_flag, (int32_t *)nullptr, (uint8_t *)nullptr, fin_max, fin_start, fin_len,
                               fin_depth, ctx->counters + CNT_SCRATCH + 1);
            NND_HIP_CHECK(hipGetLastError());
        }
        ctx->stats.n_leaves = nfin;  // overwritten below; kept for debugging
    }
    ctx->cur = cur;
    ctx->stats.tree_levels = depth;
    // leaf tables
    if (run_scan(ctx, 1, nullptr, ctx->leaf_flag, scan_total)) return 1;
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 35, scan_total, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const int32_t nl = *(const int32_t *)(ctx->h_pin + 35);
    ctx->n_leaves = nl;
    if (nl + 1 > ctx->leaf_cap) {  // grow-only: repeated builds on one handle do not pay hipFree / hipMalloc (both synchronise)
        if (ctx->leaf_start) { NND_HIP_CHECK(hipFree(ctx->leaf_start)); ctx->leaf_start = nullptr; }
        if (ctx->leaf_len) { NND_HIP_CHECK(hipFree(ctx->leaf_len)); ctx->leaf_len = nullptr; }
        ctx->leaf_cap = (int64_t)(nl + 1) + (nl + 1) / 4;
        NND_HIP_CHECK(hipMalloc((void **)&ctx->leaf_start, sizeof(int32_t) * (size_t)ctx->leaf_cap));
        NND_HIP_CHECK(hipMalloc((void **)&ctx->leaf_len, sizeof(int32_t) * (size_t)ctx->leaf_cap));
    }
    hipLaunchKernelGGL(k_leaf_starts, dim3(gridP), dim3(256), 0, ctx->stream, ctx->leaf_flag, ctx->scan_out, P,
                       ctx->leaf_start);
    // leaf lengths, the longest leaf and the per-tree leaf offsets stay on the device; the host reads T + 1 words
    // (the full tables are fetched lazily, only when a leaf has to be cut or the caller asks for the leaf array)
    int32_t *max_len_dev = (int32_t *)(ctx->counters + CNT_SCRATCH + 2);
    NND_HIP_CHECK(hipMemsetAsync(max_len_dev, 0, sizeof(long long), ctx->stream));
    hipLaunchKernelGGL(k_leaf_lens, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, ctx->stream, ctx->leaf_start,
                       (int64_t)nl, n, P, ctx->leaf_len, max_len_dev);
    hipLaunchKernelGGL(k_tree_leaf_begin, dim3((T + 63) / 64), dim3(64), 0, ctx->stream, ctx->scan_out, T, n, ctx->tree_begin_dev);
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_tree_begin, ctx->tree_begin_dev, sizeof(long long) * T, hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 36, max_len_dev, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    ctx->h_leaf_valid = false;
    int32_t mx = leaf_size;  // rp_trees.py:2548
    const int32_t longest = *(const int32_t *)(ctx->h_pin + 36);
    if (longest > mx) mx = longest;
    ctx->tree_leaf_begin.assign(T + 1, nl);
    for (int t = 0; t < T; t++) ctx->tree_leaf_begin[t] = ctx->h_tree_begin[t];
    ctx->max_leaf = mx;
    ctx->stats.n_leaves = nl;
    ctx->forest_built = true;
    return 0;
}

// host copies of the leaf table, fetched on demand (work-list cutting for over-long leaves)
int nnd_fetch_leaf_tables(nnd_ctx *ctx) {
    if (ctx->h_leaf_valid) return 0;
    const int64_t nl = ctx->n_leaves;
    ctx->h_leaf_start.resize(nl);
    ctx->h_leaf_len.resize(nl);
    if (nl > 0) {
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_leaf_start.data(), ctx->leaf_start, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_leaf_len.data(), ctx->leaf_len, sizeof(int32_t) * nl, hipMemcpyDeviceToHost, ctx->stream));
        NND_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }
    ctx->h_leaf_valid = true;
    return 0;
}

int nnd_launch_leaf_array(nnd_ctx *ctx, int32_t *out_dev) {
    int64_t total = ctx->n_leaves * ctx->max_leaf;
    if (total == 0) return 0;
    hipLaunchKernelGGL(k_fill_leaf_array, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, ctx->perm[ctx->cur],
                       ctx->leaf_start, ctx->leaf_len, ctx->n_leaves, ctx->max_leaf, out_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}
