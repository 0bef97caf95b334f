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
#include <atomic>
#include <chrono>

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
                                                    uint16_t *__restrict__ hyper_h, uint32_t pos_bias, const float *__restrict__ scal) {
    // pos_bias: the sharded build splits the tree tops by tree over the ranks; a rank's trees sit at positions
    // [0, T_local * n) here but draw what they would draw at their GLOBAL positions -- the forest does not depend on
    // the number of ranks
    int lane = nnd_lane();
    int s = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (s >= n_segs) return;
    int a = seg_start[s], len = seg_len[s];
    uint32_t li = nnd_hash3(seed, (uint32_t)a + pos_bias, (uint32_t)(2 * depth)) % (uint32_t)len;
    uint32_t ri = nnd_hash3(seed, (uint32_t)a + pos_bias, (uint32_t)(2 * depth + 1)) % (uint32_t)len;
    if (ri == li) ri = (ri + 1) % (uint32_t)len;  // rp_trees.py:353-354
    const float *xl = xp + (int64_t)perm[a + li] * dp;
    const float *xr = xp + (int64_t)perm[a + ri] * dp;
    float *h = hyper + (int64_t)s * hs;
    uint16_t *hb = hyper_h + (int64_t)s * dp;  // half-precision copy read by the screening pass of the margin kernels
    const float hsc = scal[0], hinv = 1.0f / hsc;
    float acc = 0.0f, sq = 0.0f, res = 0.0f;
    for (int j = lane; j < dp; j += 64) {
        float l = xl[j], r = xr[j];
        float v = l - r;
        h[j] = v;
        if (!angular) {
            const uint16_t b = nnd_f32_to_h16(v, hsc);
            hb[j] = b;
            const float e = v - nnd_h16_to_f32(b, hinv);
            res += e * e;
        }
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
            const uint16_t b = nnd_f32_to_h16(v, hsc);
            hb[j] = b;
            const float e = v - nnd_h16_to_f32(b, hinv);
            res += e * e;
        }
        res = nnd_wave_sum_f32(res);
        if (lane == 0) {
            h[dp] = 0.0f;
            h[dp + 1] = nh * inv;  // |h| after normalisation (1, or |h| itself when degenerate)
            h[dp + 2] = sqrtf(res) * 1.000001f;  // |h - bf16(h)|
        }
    } else {
        res = nnd_wave_sum_f32(res);
        if (lane == 0) {
            h[dp] = -0.5f * acc;
            h[dp + 1] = sqrtf(sq);
            h[dp + 2] = sqrtf(res) * 1.000001f;  // |h - bf16(h)|
        }
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
// The bound actually used is tighter: with r_x = |x - bf16(x)| (stored per point by the prep kernel) and
// r_h = |h - bf16(h)| (stored per hyperplane),
//     x.h - bf(x).bf(h) = (x - bf(x)).h + bf(x).(h - bf(h))   =>   |error| <= r_x |h| + (|x| + r_x) r_h
// by Cauchy-Schwarz on the two residual vectors -- rounding errors do not line up with the other operand the way the
// elementwise worst case assumes, and r is ~0.4 * 2^-8 of the norm on average, so the band is ~2.6x narrower than
// RP_BAND |h||x| and as rigorous.  RP_ACC covers the f32 accumulation-order differences of both sums.
#define RP_ACC 3e-5f
#ifdef NND_RP_ALWAYS_EXACT  // debugging: every margin takes the exact f32 path
#define RP_IN_BAND(m, band) ((band) >= 0.0f || (m) == (m))
#else
#define RP_IN_BAND(m, band) (!(fabsf(m) > (band)))
#endif
__device__ __forceinline__ float rp_band(float xnorm, float rx, float hnorm, float rh) {
    return rx * hnorm + (xnorm + rx) * rh + RP_ACC * hnorm * xnorm + RP_EPS;  // + RP_EPS: outside the band the exact margin is no coin flip either
}
// non-negative f32 -> bf16 bits, rounded UP (packed bounds stay bounds)
__device__ __forceinline__ uint32_t rp_bf16_up(float v) { return (__float_as_uint(v) + 0xFFFFu) >> 16; }
typedef _Float16 rp_h16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float rp_dot8(uint4 q, uint4 p, float acc) {  // 8 half x half products, f32 accumulation (v_dot2_f32_f16)
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rp_h16x2, q.x), __builtin_bit_cast(rp_h16x2, p.x), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rp_h16x2, q.y), __builtin_bit_cast(rp_h16x2, p.y), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rp_h16x2, q.z), __builtin_bit_cast(rp_h16x2, p.z), acc, false);
    acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(rp_h16x2, q.w), __builtin_bit_cast(rp_h16x2, p.w), acc, false);
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
#ifdef NND_RP_NORECHECK  // timing experiments only
    if (band < 0.0f) m = rp_exact_quad(xf_row, h, dp, sub) + off;
#else
    if (RP_IN_BAND(m, band)) m = rp_exact_quad(xf_row, h, dp, sub) + off;  // uniform inside the quad
#endif
    if (fabsf(m) < RP_EPS) return (uint8_t)(nnd_hash3(seed ^ 0x5bd1e995u, key, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
    return m > 0.0f ? 0 : 1;                                                                              // rp_trees.py:386-391
}

// position-major: neighbouring positions share a hyperplane; rows are gathered through perm
__global__ __launch_bounds__(256) void k_margin(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                const float2 *__restrict__ nr, int metric, int dp,
                                                const int32_t *__restrict__ perm, const int32_t *__restrict__ pos_seg,
                                                int64_t P, const float *__restrict__ hyper, int hs,
                                                const uint16_t *__restrict__ hyper_h, uint32_t seed, int depth,
                                                uint8_t *__restrict__ side, uint32_t pos_bias, const float *__restrict__ scal) {
    const int sub = threadIdx.x & 3;
    const int64_t g = (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    const int s = g < P ? pos_seg[g] : -1;
    if (s < 0) return;  // whole quad
    const int64_t pt = perm[g];
    const uint4 *x8 = (const uint4 *)(xh + pt * dp);
    const uint4 *h8 = (const uint4 *)(hyper_h + (int64_t)s * dp);
    const float *h = hyper + (int64_t)s * hs;
    const float2 nrv = nr[pt];  // (|x|^2 or the unit-norm flag, |x - bf16(x)|)
    const float xn = nrv.x, rx = nrv.y, off = h[dp], hnorm = h[dp + 1], rh = h[dp + 2];
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
    const float m = rp_quad_sum(acc) * scal[1] + off;
    const float band = rp_band(metric == 0 ? sqrtf(xn) : xn, rx, hnorm, rh);
    const uint8_t sd = rp_side(m, band, xp + pt * dp, h, off, dp, sub, seed, (uint32_t)g + pos_bias, depth);
    if (sub == 0) side[g] = sd;
}

// point-major variant: one pass over the points serves every tree (rows read once per level).  Everything it touches
// is point-major too -- seg_pt[t*n + i] = the point's segment in tree t (-1 once its segment is final), the side goes
// to side_pt[t*n + i] -- so apart from the hyperplane look-ups (a table that sits in L2) all its traffic is sequential.
// The scan that follows brings the sides into position order (k_scan_reduce mode 2).
__global__ __launch_bounds__(256) void k_margin_fused(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                      const float2 *__restrict__ nr, int metric, int dp, int64_t n,
                                                      int n_trees, const int32_t *__restrict__ seg_pt,
                                                      const float *__restrict__ hyper, int hs,
                                                      const uint16_t *__restrict__ hyper_h, uint32_t seed, int depth,
                                                      uint8_t *__restrict__ side_pt, uint32_t pos_bias, const float *__restrict__ scal) {
    const int sub = threadIdx.x & 3;
    const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 2) + (threadIdx.x >> 2);
    if (i >= n) return;  // whole quad
    const uint4 *x8 = (const uint4 *)(xh + i * dp);
    const float2 nrv = nr[i];
    const float xn = nrv.x, rx = nrv.y;
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
            const float off = h[dp], hnorm = h[dp + 1], rh = h[dp + 2];
            const float m = rp_quad_sum(acc[u]) * scal[1] + off;
            const float band = rp_band(xnorm, rx, hnorm, rh);
            const int64_t slot = (int64_t)(t0 + u) * n + i;
            const uint8_t sd = rp_side(m, band, xp + i * dp, h, off, dp, sub, seed, (uint32_t)slot + pos_bias, depth);
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

// RAW: blk[] holds the tiles' sums, not their exclusive scan -- every workgroup adds up the sums in front of it itself
// (a few hundred L2-resident words) and the last one writes the grand total: the single-workgroup k_scan_blocks launch
// between reduce and apply (7-8 us, once per level of the forest) is gone.
template <bool RAW>
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_apply(int mode, const int32_t *__restrict__ pos_seg,
                                                           const uint8_t *__restrict__ bytes, int64_t P,
                                                           const int32_t *__restrict__ blk, int32_t *__restrict__ out,
                                                           int32_t *__restrict__ total_out) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    int lane = nnd_lane(), w = threadIdx.x >> 6;
    int blk_prefix = 0;
    if (RAW) {
        __shared__ int psum[SCAN_BLOCK / 64];
        int part = 0;
        for (int b = threadIdx.x; b < (int)blockIdx.x; b += SCAN_BLOCK) part += blk[b];
        part = nnd_wave_sum_i32(part);
        if (lane == 0) psum[w] = part;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < SCAN_BLOCK / 64; i++) blk_prefix += psum[i];
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) total_out[0] = blk_prefix + blk[blockIdx.x];
    } else {
        blk_prefix = blk[blockIdx.x];
    }
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
    int run = blk_prefix + woff + incl - s;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < P) out[base + i] = run;
        run += f[i];
    }
}

// In-place exclusive scan of an int32 array of any length (cell counts: 37 k entries at 1 M points, 560 k in a sharded
// 10 M build -- k_scan_blocks alone walks its input with ONE workgroup: 0.47 ms there): tile sums, scan of the tile sums,
// tile-local scan + offset.
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_i32_reduce(const int32_t *__restrict__ data, int64_t n, int32_t *__restrict__ blk) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) s += base + i < n ? data[base + i] : 0;
    s = nnd_wave_sum_i32(s);
    if (nnd_lane() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < SCAN_BLOCK / 64; w++) t += wsum[w];
        blk[blockIdx.x] = t;
    }
}
__global__ __launch_bounds__(SCAN_BLOCK) void k_scan_i32_apply(int32_t *__restrict__ data, int64_t n, const int32_t *__restrict__ blk) {
    __shared__ int wsum[SCAN_BLOCK / 64];
    const int lane = nnd_lane(), w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE + (int64_t)threadIdx.x * SCAN_ITEMS;
    int f[SCAN_ITEMS], s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        f[i] = base + i < n ? data[base + i] : 0;
        s += f[i];
    }
    const int incl = nnd_wave_incl_scan_i32(s);
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int run = blk[blockIdx.x] + incl - s;
    for (int i = 0; i < w; i++) run += wsum[i];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        if (base + i < n) data[base + i] = run;
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
                                                  int32_t *__restrict__ fin_depth, long long *__restrict__ counters,
                                                  int32_t *__restrict__ node_child, int node_base, int next_base,
                                                  int32_t *__restrict__ leaf_depth, int node_top, long long *__restrict__ host_words,
                                                  long long seq) {
    // host_words (optional): pinned HOST memory.  The host needs this level's segment count before it can launch the next
    // level; instead of a device-to-host copy behind the scatter kernel and a stream synchronisation (the GPU then idles
    // for a launch latency per level), this workgroup writes the six words itself and raises a sequence number: the host,
    // spinning on it, queues the next level WHILE the scatter kernel runs.
    // node_child != nullptr (sample forest, see nnd_launch_forest): the tree itself is recorded -- node (node_base + s)
    // gets its two children: >= 0 the child's node id (next_base + its index in the next level; node_top - its index in
    // the finisher's work list when its subtree is recorded by k_finish_subtrees<.., RECORD>), <= -2 a final leaf
    // ("cell") encoded as -2 - first position; leaf_depth[first position] = its depth.
    // a child that splits again either stays in the level-synchronous passes (len > fin_max) or is handed to
    // k_finish_subtrees (len <= fin_max: its whole subtree fits in one workgroup's LDS)
    __shared__ int part[256], partf[256];
    if (threadIdx.x == 0) {  // this launch's accumulators (single workgroup: ordered by the barrier below)
        atomicExch((unsigned long long *)&counters[CNT_LEAVES], 0ull);  // atomics: ordered with the atomicAdd / atomicMax below at L2
        atomicExch((unsigned long long *)&counters[CNT_SCRATCH + 3], 0ull);
    }
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
                    if (node_child) node_child[2 * (node_base + s) + c] = next_base + run;
                    seg_child[2 * s + c] = run++;
                    active_pos += lens[c];
                    if (lens[c] > max_stay) max_stay = lens[c];
                } else {
                    fin_start[runf] = starts[c];
                    fin_len[runf] = lens[c];
                    fin_depth[runf] = child_depth;
                    if (node_child) node_child[2 * (node_base + s) + c] = node_top - runf;
                    runf++;
                    seg_child[2 * s + c] = -1;  // leaves the level-synchronous passes
                }
            } else {
                seg_child[2 * s + c] = -1;
                if (lens[c] > 0) leaf_flag[starts[c]] = 1;  // rp_trees.py:2229-2232
                if (node_child) {
                    node_child[2 * (node_base + s) + c] = -2 - starts[c];
                    if (lens[c] > 0) leaf_depth[starts[c]] = child_depth;
                }
            }
        }
    }
    if (active_pos) atomicAdd((unsigned long long *)&counters[CNT_LEAVES], (unsigned long long)active_pos);  // positions still in the passes
    if (max_stay) atomicMax((unsigned long long *)&counters[CNT_SCRATCH + 3], (unsigned long long)max_stay);  // longest of them
    if (host_words) {
        __syncthreads();  // (with its release / acquire fences: every thread's atomics above have been performed)
        if (threadIdx.x == 0) {
#pragma unroll
            for (int q = 0; q < 6; q++)
                host_words[q] = __hip_atomic_load(&counters[CNT_ACTIVE_SEGS + q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(&host_words[6], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
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
// Once a segment fits in LDS (<= FIN_MAX points) no global pass touches it again: one workgroup finishes its whole
// subtree on its own: explicit stack of sub-segments, hyperplane and member ids in LDS, margins by quads, stable
// partition by a block-wide scan.  No global synchronisation, no per-level launches; deep unbalanced branches only
// cost their own workgroup.
//
// Everything random here is keyed by POINT ID (and tree, depth), never by position, and every final leaf is written
// in ascending id order: the result does not depend on the order in which the members of the segment arrive.  That is
// what lets the routing pass (k_route below) place points into their cells with one atomicAdd each instead of a
// stable sort, and still leaves the forest bit-reproducible for a seed.
//   pivots      : the two members with the smallest hash(tree, id, depth) -- a uniformly random pair (rp_trees.py:351-356)
//   |margin|<eps: coin = hash bit of (tree, id, depth)                                  (rp_trees.py:380-385)
//   one-sided   : every member re-assigned by its hash bit -- the reference's rule       (rp_trees.py:393-403)
#ifndef NND_FIN_MAX
#define NND_FIN_MAX 2048
#endif
static constexpr int FIN_MAX = NND_FIN_MAX;     // points per finisher segment
#ifndef NND_BIG_MAX
#define NND_BIG_MAX 8192
#endif
static constexpr int BIG_MAX = NND_BIG_MAX;     // whole-set passes: longest segment handed to the global-memory variant early
static constexpr int FIN_STACK = 40;            // sub-segments pending: <= log2(2^31) + 1 because the smaller child is split first
static constexpr int FIN_SMALL = 512;           // cells of <= FIN_SMALL points: one wave per cell
static constexpr int FIN_WS = 40;               // int32 scratch words behind the stack

__device__ __forceinline__ void rp_top2_push(uint64_t &a, uint64_t &b, uint64_t k) {
    if (k < a) {
        b = a;
        a = k;
    } else if (k < b) {
        b = k;
    }
}
__device__ __forceinline__ uint64_t rp_shfl_xor_u64(uint64_t v, int o) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, o, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), o, 64);
    return ((uint64_t)hi << 32) | lo;
}

// BIG = true: the same node loop for segments of any length: member ids, partition scratch and side bits live in
// global memory (perm itself, the other perm buffer, side[]), and a node that has shrunk to <= fin_max points is
// appended to the LDS finisher's work list instead of being split here.
// RECORD = true (the sample forest of the routing pass): the subtree is RECORDED instead of written out as leaves --
// every split stores its hyperplane and children in the node tables (node ids: the segment's own id comes from
// k_children, deeper nodes take ids downwards from `down_base` through an atomic counter; ids are table slots, nothing
// depends on their order), a child that stops splitting becomes a cell: child = -2 - its first position, leaf mark and
// depth at that position.
struct rp_record {
    float *hf;            // (node_cap, hs) f32 hyperplane + offset + |h|
    uint16_t *hh;         // (node_cap, dp) bf16 hyperplane
    int32_t *child;       // (node_cap, 2)
    int32_t *leaf_depth;  // per sample position
    int hs, node_top, down_base, lvl_end;
    int *counter, *overflow;
    int min_len, max_len;  // this launch takes the segments with min_len <= len <= max_len (two launches share one work list)
};
// Which tree a segment belongs to (the per-tree salt of every hash): position / n plus tree_bias when every tree holds
// n positions; the sharded build finishes cells of ALL trees in one position space of its own, tree t at positions
// [tree_begin[t], tree_begin[t + 1]).
struct rp_tree_map {
    int tree_bias = 0;
    const int32_t *tree_begin = nullptr;
    int n_tree_begin = 0;
};
__device__ __forceinline__ uint32_t rp_tree_of(const rp_tree_map &tm, int64_t a, int64_t n) {
    if (!tm.tree_begin) return (uint32_t)(a / n) + (uint32_t)tm.tree_bias;
    int t = 0;
    for (int q = 1; q < tm.n_tree_begin; q++) t = a >= tm.tree_begin[q] ? q : t;
    return (uint32_t)t;
}

// NTHR threads per workgroup, CAP = most points of a segment whose ids live in LDS.  Small cells run with ONE WAVE per
// cell (NTHR = 64, CAP = 512: ~6 KB of LDS, the barriers are single-wave): a node of a hundred points is a chain of
// dependent latencies (pivot rows, member rows), so what pays is many independent cells per CU, not many lanes per cell.
template <bool BIG, int NTHR, int CAP, bool RECORD = false>
__global__ __launch_bounds__(NTHR, NTHR == 64 ? 5 : 1) void k_finish_subtrees(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                         const float2 *__restrict__ nr, int metric, int dp, int64_t n,
                                                         int32_t *__restrict__ perm,
                                                         const int32_t *__restrict__ seg_start,
                                                         const int32_t *__restrict__ seg_len,
                                                         const int32_t *__restrict__ seg_depth, int depth0, int n_segs,
                                                         int angular, uint32_t seed, int max_depth, int leaf_size,
                                                         uint8_t *__restrict__ leaf_flag, int32_t *__restrict__ tmp_g,
                                                         uint8_t *__restrict__ side_g, int fin_max,
                                                         int32_t *__restrict__ fin_start, int32_t *__restrict__ fin_len,
                                                         int32_t *__restrict__ fin_depth, long long *__restrict__ fin_count,
                                                         rp_tree_map tm, const float *__restrict__ scal, rp_record rec = rp_record{}) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int s = blockIdx.x;
    if (s >= n_segs) return;
    const int a = seg_start[s], len = seg_len[s];
    if (len <= 0) return;
    if (RECORD && (len < rec.min_len || len > rec.max_len)) return;
    constexpr int NLDS = BIG ? 0 : CAP;
    constexpr int NW = NTHR / 64;
    int32_t *ids = BIG ? perm + a : (int32_t *)fsm;               // member ids of the segment
    int32_t *tmp = BIG ? tmp_g + a : (int32_t *)fsm + NLDS;       // partition scratch
    uint8_t *sd = BIG ? side_g + a : (uint8_t *)((int32_t *)fsm + 2 * NLDS);  // side bits
    float *h = (float *)(fsm + (size_t)NLDS * 9);  // dp + 4 hyperplane + offset
    uint16_t *hb = (uint16_t *)(h + dp + 4);       // dp: bf16 copy of the normal (dp is a multiple of 32)
    int32_t *stk = (int32_t *)(hb + dp);           // FIN_STACK * 4: (start, len, depth, node id when recording)
    int32_t *wsum = stk + FIN_STACK * 4;           // FIN_WS: per-wave partial sums / scalars (FIN_STACK entries: see the push below)
    uint64_t *wkeys = (uint64_t *)(wsum + 8);      // 8 keys (16 words): per-wave top-2 of the pivot draw
    const int tid = threadIdx.x, lane = nnd_lane(), w = tid >> 6;
    const uint32_t seedt = seed ^ (rp_tree_of(tm, a, n) * 0x9E3779B9u);  // per tree
    const float hsc = scal[0], hinv = 1.0f / hsc, inv_s2 = scal[1];
    // The screening band of a member is |h| (r_x + ACC |x|) + |h - half(h)| (|x| + r_x) + eps (rp_band regrouped).  The LDS
    // variants take the LARGEST (r_x + ACC |x|) and (|x| + r_x) of the cell for all its members: they are read ONCE per
    // member, while the ids are loaded, instead of once per member and level (a random 8-byte gather that costs a whole
    // line: a quarter of this kernel's fetches); members of a cell lie close together, so the common band is hardly wider.
    float cellA = 0.0f, cellB = 0.0f;
    if (!BIG) {
        for (int i = tid; i < len; i += NTHR) {
            const int32_t id = perm[a + i];
            ids[i] = id;
            const float2 nrv = nr[id];
            const float xnm = metric == 0 ? sqrtf(nrv.x) : nrv.x;
            cellA = fmaxf(cellA, nrv.y + RP_ACC * xnm);
            cellB = fmaxf(cellB, xnm + nrv.y);
        }
        cellA = nnd_wave_max_f32_u(cellA);
        cellB = nnd_wave_max_f32_u(cellB);
        if (NW > 1) {
            if (lane == 0) {
                ((float *)wsum)[w] = cellA;
                ((float *)wsum)[8 + w] = cellB;
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NW; q++) {
                cellA = fmaxf(cellA, ((float *)wsum)[q]);
                cellB = fmaxf(cellB, ((float *)wsum)[8 + q]);
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        stk[0] = 0; stk[1] = len; stk[2] = seg_depth ? seg_depth[s] : depth0;
        stk[3] = RECORD ? rec.node_top - s : 0;  // k_children numbered the finisher segments downwards from node_top
        wsum[7] = 1;  // stack size
    }
    __syncthreads();
    while (true) {
        const int sp = wsum[7];
        if (sp == 0) break;
        const int ss = stk[(sp - 1) * 4], l = stk[(sp - 1) * 4 + 1], dep = stk[(sp - 1) * 4 + 2], me = stk[(sp - 1) * 4 + 3];
        __syncthreads();
        if (tid == 0) wsum[7] = sp - 1;
        if (!(l > leaf_size && (max_depth - dep) > 0)) {  // rp_trees.py:2188: this node is a leaf
            if (tid == 0 && l > 0) leaf_flag[a + ss] = 1;
            if (!RECORD && l > 1) {  // canonical order: ascending ids (rank by counting; leaves are small)
                for (int i = tid; i < l; i += NTHR) {
                    const int32_t id = ids[ss + i];
                    int r = 0;
                    for (int j = 0; j < l; j++) r += ids[ss + j] < id ? 1 : 0;
                    tmp[r] = id;
                }
                __syncthreads();
                for (int i = tid; i < l; i += NTHR) ids[ss + i] = tmp[i];
            }
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
        // two random members -> hyperplane (rp_trees.py:350-367 / 87-118): the two smallest hash(tree, id, depth)
        uint64_t k1 = ~0ull, k2 = ~0ull;
        for (int i = tid; i < l; i += NTHR) {
            const uint32_t id = (uint32_t)ids[ss + i];
            rp_top2_push(k1, k2, ((uint64_t)nnd_hash3(seedt, id, (uint32_t)(2 * dep)) << 32) | id);
        }
        {  // the wave's two smallest keys: the smallest, then the smallest of what is left (keys are distinct: they end in the id)
            const uint64_t g1 = nnd_wave_min_u64_u(k1);
            const uint64_t g2 = nnd_wave_min_u64_u(k1 == g1 ? k2 : k1);
            k1 = g1;
            k2 = g2;
        }
        if (lane == 0) {
            wkeys[2 * w] = k1;
            wkeys[2 * w + 1] = k2;
        }
        __syncthreads();
        k1 = k2 = ~0ull;
#pragma unroll
        for (int q = 0; q < 2 * NW; q++) rp_top2_push(k1, k2, wkeys[q]);
        const int64_t idl = (int64_t)(uint32_t)k1, idr = (int64_t)(uint32_t)k2;
        const float *xl = xp + idl * dp;
        const float *xr = xp + idr * dp;
        float part = 0.0f, psq = 0.0f;
        for (int j = tid; j < dp; j += NTHR) {
            const float lv = xl[j], rv = xr[j];
            const float v = lv - rv;
            h[j] = v;
            part += angular ? v * v : v * (lv + rv);
            psq += v * v;
        }
        part = nnd_wave_sum_f32_u(part);
        psq = nnd_wave_sum_f32_u(psq);
        if (lane == 0) {
            ((float *)wsum)[w] = part;
        }
        __syncthreads();
        float tot = 0.0f;
#pragma unroll
        for (int q = 0; q < NW; q++) tot += ((float *)wsum)[q];
        __syncthreads();
        if (lane == 0) ((float *)wsum)[w] = psq;
        __syncthreads();
        float totsq = 0.0f;
#pragma unroll
        for (int q = 0; q < NW; q++) totsq += ((float *)wsum)[q];
        __syncthreads();
        if (angular) {  // normalise in place (rp_trees.py:113-118); offset 0
            const float nh = sqrtf(tot);
            const float inv = nh < RP_EPS ? 1.0f : 1.0f / nh;
            for (int j = tid; j < dp; j += NTHR) h[j] *= inv;
            if (tid == 0) { h[dp] = 0.0f; h[dp + 1] = nh * inv; }
        } else if (tid == 0) {
            h[dp] = -0.5f * tot;
            h[dp + 1] = sqrtf(totsq);
        }
        __syncthreads();
        // margins: one quad per member (bf16-screened like k_margin), two members per quad and step so that 8 row
        // fetches are in flight per lane; the bf16 hyperplane comes from LDS
        float pres = 0.0f;  // |h - bf16(h)|^2: the hyperplane's share of the screening band (rp_band)
        for (int j = tid; j < dp; j += NTHR) {
            const float v = h[j];
            const uint16_t b = nnd_f32_to_h16(v, hsc);
            hb[j] = b;
            const float e = v - nnd_h16_to_f32(b, hinv);
            pres += e * e;
        }
        pres = nnd_wave_sum_f32_u(pres);
        if (lane == 0) ((float *)wsum)[w] = pres;
        __syncthreads();
        float rhv = 0.0f;
#pragma unroll
        for (int q = 0; q < NW; q++) rhv += ((float *)wsum)[q];
        rhv = sqrtf(rhv) * 1.000001f;
        if (RECORD) {
            float *rh = rec.hf + (int64_t)me * rec.hs;
            uint16_t *rb = rec.hh + (int64_t)me * dp;
            for (int j = tid; j < dp; j += NTHR) {
                rh[j] = h[j];
                rb[j] = hb[j];
            }
            if (tid == 0) {
                rh[dp] = h[dp];
                rh[dp + 1] = h[dp + 1];
                rh[dp + 2] = rhv;
            }
        }
        const int sub = tid & 3, grp = tid >> 2;
        const int nch = dp >> 3;
        const float off = h[dp], hnorm = h[dp + 1];
        const uint4 *h8 = (const uint4 *)hb;
        constexpr int GQ = NTHR / 4;  // quads per workgroup
        for (int i0 = 0; i0 < l; i0 += 2 * GQ) {
            int64_t pt[2];
            float acc[2], xn[2], rxv[2];
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const int i = i0 + u * GQ + grp;
                pt[u] = ids[ss + (i < l ? i : 0)];
                acc[u] = 0.0f;
            }
            if (BIG) {
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const float2 nrv = nr[pt[u]];
                    xn[u] = nrv.x;
                    rxv[u] = nrv.y;
                }
            }
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
                const int i = i0 + u * GQ + grp;
                if (i >= l) continue;  // whole quad
                const float m = rp_quad_sum(acc[u]) * inv_s2 + off;
                const float band = BIG ? rp_band(metric == 0 ? sqrtf(xn[u]) : xn[u], rxv[u], hnorm, rhv) : hnorm * cellA + rhv * cellB + RP_EPS;
                const uint8_t side = rp_side(m, band, xp + pt[u] * dp, h, off, dp, sub, seedt, (uint32_t)pt[u], dep);
                if (sub == 0) sd[i] = side;
            }
        }
        __syncthreads();
        // stable partition: block-wide exclusive scan of "left" over the l members (l / 256 per thread).  A one-sided
        // split (rp_trees.py:393-403) re-draws every member's side from its hash bit first (and, should those agree
        // too, sends the first pivot left on its own), then scans again.
        const int per = (l + NTHR - 1) / NTHR;
        const int b0 = tid * per < l ? tid * per : l, b1 = b0 + per < l ? b0 + per : l;
        int cntl, incl, nl, woff;
        for (int attempt = 0;; attempt++) {
            cntl = 0;
            for (int i = b0; i < b1; i++) cntl += sd[i] == 0;
            incl = nnd_wave_incl_scan_i32(cntl);
            if (lane == 63) wsum[w] = incl;
            __syncthreads();
            woff = 0;
            for (int i = 0; i < w; i++) woff += wsum[i];
            nl = 0;
#pragma unroll
            for (int q = 0; q < NW; q++) nl += wsum[q];
            if (nl != 0 && nl != l) break;  // block-uniform
            __syncthreads();                // everyone has read wsum
            for (int i = b0; i < b1; i++) {
                const uint32_t id = (uint32_t)ids[ss + i];
                sd[i] = attempt == 0 ? (uint8_t)(nnd_hash3(seedt ^ 0x5bd1e995u, id, (uint32_t)(2 * dep + 1)) & 1u)
                                     : (uint8_t)((int64_t)id == idl ? 0 : 1);
            }
            __syncthreads();
        }
        int run = woff + incl - cntl;
        for (int i = b0; i < b1; i++) {
            int dest;
            if (sd[i] == 0) dest = run++;
            else dest = nl + (i - run);
            tmp[dest] = ids[ss + i];
        }
        __syncthreads();
        for (int i = tid; i < l; i += NTHR) ids[ss + i] = tmp[i];
        if (tid == 0) {  // the LARGER child is pushed first, the smaller one is processed next: the pending stack never
                         // holds more than log2(len) + 1 entries (the order in which nodes are split is immaterial)
            int top = wsum[7];
            const bool left_big = nl >= l - nl;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                const bool left = (q == 0) == left_big;
                const int cs = left ? ss : ss + nl, cl = left ? nl : l - nl;
                int cid = 0;
                if (RECORD) {
                    bool split = cl > leaf_size && (max_depth - (dep + 1)) > 0;
                    if (split) {
                        cid = rec.down_base - atomicAdd(rec.counter, 1);
                        if (cid <= rec.lvl_end) {  // node tables exhausted: the host falls back to the whole-set passes
                            *rec.overflow = 1;
                            split = false;
                        }
                    }
                    if (!split) {  // a cell
                        rec.child[2 * me + (left ? 0 : 1)] = -2 - (a + cs);
                        leaf_flag[a + cs] = 1;
                        rec.leaf_depth[a + cs] = dep + 1;
                        continue;
                    }
                    rec.child[2 * me + (left ? 0 : 1)] = cid;
                }
                stk[top * 4] = cs; stk[top * 4 + 1] = cl; stk[top * 4 + 2] = dep + 1; stk[top * 4 + 3] = cid;
                top++;
            }
            wsum[7] = top;
        }
        __syncthreads();
    }
    if (!BIG)
        for (int i = tid; i < len; i += NTHR) perm[a + i] = ids[i];
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
__global__ void k_tree_leaf_begin(const int32_t *__restrict__ scan, int n_trees, int64_t n, const int32_t *__restrict__ tree_begin,
                                  int64_t P, int32_t n_leaves, long long *__restrict__ out) {
    // tree t starts at position t * n, or at tree_begin[t] (sharded build: a rank's own cells of every tree); a tree that
    // holds no position here starts where the next one does
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_trees) return;
    const int64_t g = tree_begin ? tree_begin[t] : (int64_t)t * n;
    out[t] = g < P ? scan[g] : n_leaves;
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

// ------------------------------------------------------------ routing pass --
// Large point sets do not run the level-synchronous passes over all P = n_trees * n positions.  The TOP of every tree is
// built from a SAMPLE (every SAMPLE_STRIDE-th point, jittered): the level-synchronous machinery above runs on the
// compact copy of the sample rows -- 1/8 of the positions -- and records the tree (hyperplanes, children) down to nodes
// of <= cell_leaf sample members ("cells": a few hundred points each).  A node's two pivots are uniformly random
// members of the sample inside the node, i.e. uniformly random members of the node (rp_trees.py:351-356); the top
// nodes all hold far more than leaf_size points, so the reference's stop rule (rp_trees.py:2188) never fires there.
// Then ONE pass routes every point through the recorded trees (k_route): the f32 row stays in registers for all
// trees and levels, only hyperplanes are fetched (bf16 screen from L2, exact f32 recheck inside the error band, same
// coin flips for |margin| < eps), the point's cell is counted with one atomicAdd whose return value is its slot in
// the cell, and k_place writes the permutation.  Cells are finished by k_finish_subtrees, which is order independent.
__global__ void k_gather_sample(const float *__restrict__ xp, const uint16_t *__restrict__ xh, const float2 *__restrict__ nr,
                                int dp, int64_t j_lo, int64_t j_hi, int64_t stride, uint32_t seed, float *__restrict__ xs,
                                uint16_t *__restrict__ xsh, float2 *__restrict__ nrs) {
    // sample members [j_lo, j_hi) (the sharded build gathers the members among a rank's own rows, then all-gathers)
    const int sub = threadIdx.x & 15;
    const int64_t j = j_lo + (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    if (j >= j_hi) return;
    const int64_t i = j * stride + (int64_t)(nnd_hash2(seed ^ 0x7F4A7C15u, (uint32_t)j) % (uint32_t)stride);
    for (int c = sub; c < (dp >> 2); c += 16) ((float4 *)(xs + j * dp))[c] = ((const float4 *)(xp + i * dp))[c];
    for (int c = sub; c < (dp >> 3); c += 16) ((uint4 *)(xsh + j * dp))[c] = ((const uint4 *)(xh + i * dp))[c];
    if (sub == 0) nrs[j] = nr[i];
}

__device__ __forceinline__ uint32_t rp_pack_h16(float a, float b, float scale) {
    return (uint32_t)nnd_f32_to_h16(a, scale) | ((uint32_t)nnd_f32_to_h16(b, scale) << 16);
}
// explicit fused multiply-adds, in this order: both routing forms must round an exact margin the same way (left to the
// compiler's contraction one kernel got v_pk_mul + v_add, the other v_pk_fma: one decision in 30 million differed)
__device__ __forceinline__ float rp_dot4f(float4 a, float4 b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    return __builtin_fmaf(a.w, b.w, acc);
}

// Recorded nodes are packed for the walk: [dp bf16 hyperplane | f32 offset | bf16 |h| : bf16 |h - bf16(h)| (both rounded
// up) | child 0 | child 1], one record of 2 * dp + 16 bytes per node (a walk step touches ONE contiguous record instead
// of three tables).  The table is COMPACTED on the way: the level-synchronous passes number their nodes upwards from 0
// ([0, n_low)), the recording finisher downwards from the end of the table ([high_lo, node_cap)); record v' of the packed
// table is node v' (v' < n_low) or node high_lo + (v' - n_low).  Children: a node -> its packed id + node_base (the
// sharded build concatenates the tables of all ranks); a cell -> -2 - its GLOBAL cell number, looked up through
// cell_gid (sharded build: cells are renumbered owner-major, rpforest.hip nnd_forest_tops_pack) or cell_base + its
// number in position order.  hf_out (optional): the f32 hyperplanes (exact rechecks) compacted the same way.
struct rp_pack_map {
    int64_t n_low, high_lo;     // compaction (see above)
    int node_base, cell_base;   // rebasing
    const int32_t *cell_gid;    // (n_cells) local cell number -> global cell number, or nullptr
};
__global__ void k_pack_nodes(const uint16_t *__restrict__ node_hh, const float *__restrict__ node_hf, int hs,
                             const int32_t *__restrict__ node_child, const int32_t *__restrict__ leafscan, int dp, int64_t n_packed,
                             rp_pack_map mp, unsigned char *__restrict__ pack, float *__restrict__ hf_out) {
    const int sub = threadIdx.x & 15;
    const int64_t vp = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    if (vp >= n_packed) return;
    const int64_t v = vp < mp.n_low ? vp : mp.high_lo + (vp - mp.n_low);
    const int rec = 2 * dp + 16;
    uint4 *dst = (uint4 *)(pack + vp * rec);
    const uint4 *src = (const uint4 *)(node_hh + v * dp);
    for (int c = sub; c < (dp >> 3); c += 16) dst[c] = src[c];
    const float *h = node_hf + v * hs;
    if (hf_out)
        for (int c = sub; c < (hs >> 2); c += 16) ((float4 *)(hf_out + vp * hs))[c] = ((const float4 *)h)[c];
    if (sub == 0) {
        int ch[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const int c = node_child[2 * v + q];
            if (c >= 0) {
                ch[q] = (int)((int64_t)c < mp.n_low ? c : c - (mp.high_lo - mp.n_low)) + mp.node_base;
            } else {  // <= -2: a cell, encoded as -2 - its first sample position
                const int lc = leafscan[-2 - c];
                ch[q] = -2 - (mp.cell_gid ? mp.cell_gid[lc] : mp.cell_base + lc);
            }
        }
        dst[dp >> 3] = make_uint4(__float_as_uint(h[dp]), (rp_bf16_up(h[dp + 1]) << 16) | rp_bf16_up(h[dp + 2]), (uint32_t)ch[0], (uint32_t)ch[1]);
    }
}

// One quad per point; lane `sub` of the quad holds the 8-float chunks sub, sub+4, ... of the row (NC = dp / 32 of them)
// as f32 (exact recheck) and as packed bf16 (screening operand of v_dot2_f32_bf16).  TB trees walk down in lock step
// from their roots, so step d of every walk is at depth d: the records of the first l_top levels (node ids
// [0, n_top), the level-synchronous build numbers nodes level by level) are served from an LDS copy, deeper ones from
// L2.  Persistent workgroups (the LDS copy is loaded once per workgroup).
template <int NC, int TB>
__global__ __launch_bounds__(512) void k_route(const float *__restrict__ xp, const float2 *__restrict__ nr, int metric, int dp,
                                               int64_t n, int n_trees, const unsigned char *__restrict__ node_pack,
                                               const float *__restrict__ node_hf, int hs, uint32_t seed,
                                               int32_t *__restrict__ cell_count, int32_t *__restrict__ cell_of,
                                               int32_t *__restrict__ rank_of, int n_top, int l_top, const float *__restrict__ scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char top_tab[];
    const int rec = 2 * dp + 16;
    const float hsc = scal[0], inv_s2 = scal[1];
    {
        const uint4 *src = (const uint4 *)node_pack;
        uint4 *dst = (uint4 *)top_tab;
        const int total = n_top * (rec >> 4);
        for (int q = threadIdx.x; q < total; q += blockDim.x) dst[q] = src[q];
    }
    __syncthreads();
    const int sub = threadIdx.x & 3;
    const int qpb = blockDim.x >> 2;
    for (int64_t i0 = (int64_t)blockIdx.x * qpb; i0 < n; i0 += (int64_t)gridDim.x * qpb) {
        const int64_t i = i0 + (threadIdx.x >> 2);
        if (i >= n) continue;  // whole quad; no workgroup barrier below
        float4 xa[NC], xb[NC];
        uint4 xq[NC];
        {
            const float4 *row = (const float4 *)(xp + i * dp);
#pragma unroll
            for (int q = 0; q < NC; q++) {
                const int c = sub + 4 * q;
                xa[q] = row[2 * c];
                xb[q] = row[2 * c + 1];
            }
#pragma unroll
            for (int q = 0; q < NC; q++)
                xq[q] = make_uint4(rp_pack_h16(xa[q].x, xa[q].y, hsc), rp_pack_h16(xa[q].z, xa[q].w, hsc), rp_pack_h16(xb[q].x, xb[q].y, hsc),
                                   rp_pack_h16(xb[q].z, xb[q].w, hsc));
        }
        const float2 nrv = nr[i];
        const float xn = nrv.x, rx = nrv.y;
        const float xnorm = metric == 0 ? sqrtf(xn) : xn;
        for (int t0 = 0; t0 < n_trees; t0 += TB) {
            int node[TB];  // >= 0: current node (the root of tree t is node t); -1: this walk is over
            int rk[TB];    // the point's slot in its cell: the atomic's return value is not touched before the walks of
                           // this batch are over, so its round trip overlaps the other walks' record fetches
#pragma unroll
            for (int u = 0; u < TB; u++) {
                node[u] = t0 + u < n_trees ? t0 + u : -1;
                rk[u] = 0;
            }
            for (int depth = 0;; depth++) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < TB; u++) any |= node[u] >= 0;
                if (!__ballot(any)) break;  // wave-uniform
                uint4 p[TB][NC], meta[TB];
                bool high = false;  // a subtree recorded by the finisher has its nodes at the far end of the table
#pragma unroll
                for (int u = 0; u < TB; u++) high |= node[u] >= n_top;
                if (depth < l_top && !__ballot(high)) {  // wave-uniform: every live walk is at a node of the LDS copy
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const uint4 *r8 = (const uint4 *)(top_tab + (size_t)(node[u] >= 0 ? node[u] : 0) * rec);
#pragma unroll
                        for (int q = 0; q < NC; q++) p[u][q] = r8[sub + 4 * q];
                        meta[u] = r8[dp >> 3];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < TB; u++) {
                        const uint4 *r8 = (const uint4 *)(node_pack + (int64_t)(node[u] >= 0 ? node[u] : 0) * rec);
#pragma unroll
                        for (int q = 0; q < NC; q++) p[u][q] = r8[sub + 4 * q];
                        meta[u] = r8[dp >> 3];
                    }
                }
                float acc[TB];
#pragma unroll
                for (int u = 0; u < TB; u++) {
                    acc[u] = 0.0f;
#pragma unroll
                    for (int q = 0; q < NC; q++) acc[u] = rp_dot8(xq[q], p[u][q], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < TB; u++) {
                    if (node[u] < 0) continue;  // whole quad
                    const float off = __uint_as_float(meta[u].x), hnorm = __uint_as_float(meta[u].y & 0xFFFF0000u),
                                rh = __uint_as_float(meta[u].y << 16);
                    float m = rp_quad_sum(acc[u]) * inv_s2 + off;
                    const float band = rp_band(xnorm, rx, hnorm, rh);
                    if (RP_IN_BAND(m, band)) {  // inside the screening error band: the exact f32 margin decides (quad-uniform)
                        const float4 *h4 = (const float4 *)(node_hf + (int64_t)node[u] * hs);
                        float e = 0.0f;
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            const int c = sub + 4 * q;
                            e = rp_dot4f(xa[q], h4[2 * c], e);
                            e = rp_dot4f(xb[q], h4[2 * c + 1], e);
                        }
                        m = rp_quad_sum(e) + off;
                    }
                    const int64_t slot = (int64_t)(t0 + u) * n + i;
                    int side;
                    if (fabsf(m) < RP_EPS) side = (int)(nnd_hash3(seed ^ 0x5bd1e995u, (uint32_t)slot, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
                    else side = m > 0.0f ? 0 : 1;                                                                             // rp_trees.py:386-391
                    const int nxt = (int)(side ? meta[u].w : meta[u].z);
                    if (nxt <= -2) {  // reached a cell (its number is baked into the record, k_pack_nodes)
                        if (sub == 0) {
                            const int cell = -2 - nxt;
                            cell_of[slot] = cell;
                            rk[u] = atomicAdd(&cell_count[cell], 1);
                        }
                        node[u] = -1;
                    } else {
                        node[u] = nxt;
                    }
                }
            }
            if (sub == 0) {
#pragma unroll
                for (int u = 0; u < TB; u++)
                    if (t0 + u < n_trees) rank_of[(int64_t)(t0 + u) * n + i] = rk[u];
            }
        }
    }
}

// ------------------------------------------------------------ coherent routing --
// k_route above walks a point through a whole recorded tree in one go: beyond the first few levels every step is a
// dependent fetch of a random 2 * dp + 16-byte record that misses the 4 MB L2 of its XCD (9.3 GB of fetches for 0.5 GB
// of rows at 1 M points x 8 trees; a 10 M-point tree has 11 MB of records).  The coherent form makes every record
// fetch an LDS read, in two passes:
//   pass 1  k_route_top     every point walks the first L1 levels of every tree from an LDS copy of those levels
//                           (heap order, k_top_heap) and lands in a BUCKET = one of the 2^L1 subtrees below them;
//           k_bucket_prefix / k_bucket_scatter   counting sort of the (tree, point) pairs by bucket;
//   pass 2  k_route_bucket  a workgroup takes a run of one bucket's points, stages that subtree's records in LDS ONCE
//                           (breadth-first, k_bucket_tables) and walks the run through them, rows read as bf16 (the
//                           screening operand; the f32 row is touched only inside the error band).
// Same arithmetic, same coins, same cells as k_route (tests/test_gpu_kernels.py compares the two): a point's cell does
// not depend on the order in which points arrive.  The walk works on any row range [row_lo, row_lo + nrows) and any
// set of packed trees -- the sharded build routes a rank's own rows through the gathered tops of ALL trees.
#define RP_GLOBAL_TAG 0x40000000  // child code in an LDS record: >= TAG: node id + TAG, fetched from global memory

// heap-ordered copy of the first L1 levels of every tree: slot 1 = root, children of slot s at 2s and 2s + 1
__global__ __launch_bounds__(256) void k_top_heap(const unsigned char *__restrict__ pack, int rec, int dp, const int32_t *__restrict__ roots,
                                                  int L1, unsigned char *__restrict__ top_rec, int32_t *__restrict__ top_node,
                                                  int32_t *__restrict__ bucket_root) {
    __shared__ int32_t hn[256];  // 2^(L1 + 1) <= 256
    const int t = blockIdx.x, nslots = 1 << L1, r16 = rec >> 4;
    if (threadIdx.x == 0) {
        hn[0] = -1;
        hn[1] = roots[t];
    }
    __syncthreads();
    for (int l = 0; l < L1; l++) {
        for (int sl = (1 << l) + threadIdx.x; sl < (2 << l); sl += blockDim.x) {
            const int nd = hn[sl];
            int c0 = -1, c1 = -1;
            if (nd >= 0) {
                const uint4 meta = *(const uint4 *)(pack + (int64_t)nd * rec + 2 * dp);
                c0 = (int)meta.z;
                c1 = (int)meta.w;
            }
            hn[2 * sl] = c0 >= 0 ? c0 : -1;  // cells (<= -2) have no record
            hn[2 * sl + 1] = c1 >= 0 ? c1 : -1;
        }
        __syncthreads();
    }
    for (int q = threadIdx.x; q < nslots * r16; q += blockDim.x) {
        const int sl = q / r16, wd = q - sl * r16;
        const int nd = hn[sl];
        if (nd >= 0) ((uint4 *)(top_rec + (size_t)t * nslots * rec))[q] = ((const uint4 *)(pack + (int64_t)nd * rec))[wd];
    }
    for (int sl = threadIdx.x; sl < nslots; sl += blockDim.x) {
        top_node[t * nslots + sl] = hn[sl];
        bucket_root[t * nslots + sl] = hn[nslots + sl];
    }
}

// Both passes give a point to a PAIR of lanes (lane `sub` holds the 16-byte chunks sub, sub + 2, ... of the half-precision
// row: NC2 = dp / 16 of them): the walk is bound by VALU issue (SQ counters: the four waves of a SIMD are issuing all the
// time), and per point a pair spends ~40 % fewer wave instructions than a quad -- the per-step overhead (address, band,
// side, child) is paid once per lane, the dot products are the same.
__device__ __forceinline__ float rp_pair_sum(float v) {
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
}
// exact f32 margin (without the offset) by a pair of lanes, in the summation order of the quad kernels (rp_exact_quad,
// k_route): lane `sub` computes the partial sums the quad's lanes sub and sub + 2 would, and the four are added as
// (e0 + e1) + (e2 + e3) -- bitwise the quad's result, so both routing forms make the same decision on every margin
template <int NC>
__device__ __forceinline__ float rp_exact_pair(const float *__restrict__ xrow, const float *__restrict__ h, int sub) {
    const float4 *x4 = (const float4 *)xrow;
    const float4 *h4 = (const float4 *)h;
    float ea = 0.0f, eb = 0.0f;
#pragma unroll
    for (int q = 0; q < NC; q++) {
        const int c = sub + 4 * q, c2 = sub + 2 + 4 * q;
        ea = rp_dot4f(x4[2 * c], h4[2 * c], ea);
        ea = rp_dot4f(x4[2 * c + 1], h4[2 * c + 1], ea);
        eb = rp_dot4f(x4[2 * c2], h4[2 * c2], eb);
        eb = rp_dot4f(x4[2 * c2 + 1], h4[2 * c2 + 1], eb);
    }
    ea = rp_pair_sum(ea);
    eb = rp_pair_sum(eb);
    return ea + eb;
}

// pass 1.  The trees of the group [tg0, tg0 + tgn) one after the other; the half-precision row stays in registers for all
// of them.  code[t * nrows + r] = the point's bucket (t * 2^L1 + b), or -1 when the walk ended in a cell inside the top
// levels (counted and placed here).  Bucket populations are counted in LDS and flushed once per workgroup.
template <int NC>
__global__ __launch_bounds__(1024) void k_route_top(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                    const float2 *__restrict__ nr, int metric, int dp, int64_t n, int64_t row_lo,
                                                    int64_t nrows, int tg0, int tgn, int L1, const unsigned char *__restrict__ top_rec,
                                                    const int32_t *__restrict__ top_node, const float *__restrict__ node_hf, int hs,
                                                    uint32_t seed, int32_t *__restrict__ cell_count, int32_t *__restrict__ cell_of,
                                                    int32_t *__restrict__ rank_of, int32_t *__restrict__ code,
                                                    int32_t *__restrict__ bucket_count, const float *__restrict__ scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char top_tab[];
    constexpr int NC2 = 2 * NC;
    const int rec = 2 * dp + 16, nslots = 1 << L1;
    const float inv_s2 = scal[1];
    int32_t *hist = (int32_t *)(top_tab + (size_t)tgn * nslots * rec);
    int32_t *tnode = hist + tgn * nslots;
    {
        const uint4 *src = (const uint4 *)(top_rec + (size_t)tg0 * nslots * rec);
        uint4 *dst = (uint4 *)top_tab;
        const int total = tgn * nslots * (rec >> 4);
        for (int q = threadIdx.x; q < total; q += blockDim.x) dst[q] = src[q];
        for (int q = threadIdx.x; q < tgn * nslots; q += blockDim.x) {
            hist[q] = 0;
            tnode[q] = top_node[tg0 * nslots + q];
        }
    }
    __syncthreads();
    const int sub = threadIdx.x & 1;
    const int ppb = blockDim.x >> 1;
    for (int64_t r0 = (int64_t)blockIdx.x * ppb; r0 < nrows; r0 += (int64_t)gridDim.x * ppb) {
        const int64_t r = r0 + (threadIdx.x >> 1);
        const bool on = r < nrows;  // (idle pairs of the last round walk row 0 and write nothing: no divergent loop bounds)
        const int64_t i = row_lo + (on ? r : 0);
        uint4 xq[NC2];
        {
            const uint4 *row = (const uint4 *)(xh + i * dp);
#pragma unroll
            for (int q = 0; q < NC2; q++) xq[q] = row[sub + 2 * q];
        }
        const float2 nrv = nr[i];
        const float xnorm = metric == 0 ? sqrtf(nrv.x) : nrv.x;
        const float bA = nrv.y + RP_ACC * xnorm, bB = xnorm + nrv.y;  // band = |h| bA + |h - half(h)| bB + eps (rp_band regrouped)
        for (int tl = 0; tl < tgn; tl++) {
            const int t = tg0 + tl;
            const int64_t slot = (int64_t)t * n + i;
            int hidx = 1, rk = 0;
            bool fin = false;
            for (int depth = 0; depth < L1; depth++) {
                if (!__ballot(!fin)) break;  // wave-uniform
                const uint4 *r8 = (const uint4 *)(top_tab + ((size_t)tl * nslots + (fin ? 1 : hidx)) * rec);
                uint4 p[NC2];
#pragma unroll
                for (int q = 0; q < NC2; q++) p[q] = r8[sub + 2 * q];
                const uint4 meta = r8[dp >> 3];
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < NC2; q++) acc = rp_dot8(xq[q], p[q], acc);
                if (fin) continue;  // whole pair
                const float off = __uint_as_float(meta.x), hnorm = __uint_as_float(meta.y & 0xFFFF0000u), rh = __uint_as_float(meta.y << 16);
                float m = rp_pair_sum(acc) * inv_s2 + off;
                const float band = hnorm * bA + rh * bB + RP_EPS;
                if (RP_IN_BAND(m, band))  // inside the screening error band: the exact f32 margin decides (pair-uniform)
                    m = rp_exact_pair<NC>(xp + i * dp, node_hf + (int64_t)tnode[tl * nslots + hidx] * hs, sub) + off;
                int side;
                if (fabsf(m) < RP_EPS) side = (int)(nnd_hash3(seed ^ 0x5bd1e995u, (uint32_t)slot, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
                else side = m > 0.0f ? 0 : 1;                                                                             // rp_trees.py:386-391
                const int nxt = (int)(side ? meta.w : meta.z);
                if (nxt <= -2) {  // reached a cell inside the top levels
                    if (sub == 0 && on) {
                        const int cell = -2 - nxt;
                        cell_of[(int64_t)t * nrows + r] = cell;
                        rk = atomicAdd(&cell_count[cell], 1);
                    }
                    fin = true;
                } else {
                    hidx = 2 * hidx + side;
                }
            }
            if (sub == 0 && on) {
                const int64_t o = (int64_t)t * nrows + r;
                if (fin) {
                    rank_of[o] = rk;
                    code[o] = -1;
                } else {
                    const int b = hidx - nslots;  // L1 steps taken: the walk stands at a root of the bucket level
                    code[o] = t * nslots + b;
                    atomicAdd(&hist[tl * nslots + b], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int q = threadIdx.x; q < tgn * nslots; q += blockDim.x)
        if (hist[q]) atomicAdd(&bucket_count[tg0 * nslots + q], hist[q]);
}

// single workgroup: bucket offsets and the work items of pass 2 (a bucket's points in runs of `chunk`)
__global__ __launch_bounds__(256) void k_bucket_prefix(const int32_t *__restrict__ bucket_count, int nb, int chunk,
                                                       int32_t *__restrict__ bucket_start, int2 *__restrict__ items,
                                                       int32_t *__restrict__ n_items, int max_items) {
    __shared__ int pa[256], pi[256];
    const int per = (nb + 255) / 256;
    const int b0 = threadIdx.x * per < nb ? threadIdx.x * per : nb, b1 = b0 + per < nb ? b0 + per : nb;
    int sa = 0, si = 0;
    for (int b = b0; b < b1; b++) {
        const int c = bucket_count[b];
        sa += c;
        si += (c + chunk - 1) / chunk;
    }
    pa[threadIdx.x] = sa;
    pi[threadIdx.x] = si;
    __syncthreads();
    if (threadIdx.x == 0) {
        int ra = 0, ri = 0;
        for (int q = 0; q < 256; q++) {
            const int va = pa[q], vi = pi[q];
            pa[q] = ra;
            pi[q] = ri;
            ra += va;
            ri += vi;
        }
        bucket_start[nb] = ra;
        n_items[0] = ri < max_items ? ri : max_items;
    }
    __syncthreads();
    int ra = pa[threadIdx.x], ri = pi[threadIdx.x];
    for (int b = b0; b < b1; b++) {
        const int c = bucket_count[b];
        bucket_start[b] = ra;
        ra += c;
        const int nc = (c + chunk - 1) / chunk;
        for (int j = 0; j < nc; j++)
            if (ri + j < max_items) items[ri + j] = make_int2(b, j);
        ri += nc;
    }
}

// counting sort of the (tree, point) pairs by bucket: grid (runs of 4096 points, trees); a workgroup counts its run per
// bucket in LDS, reserves its share of every bucket with ONE global atomic, and places the points
__global__ __launch_bounds__(256) void k_bucket_scatter(const int32_t *__restrict__ code, int64_t nrows, int64_t row_lo, int L1,
                                                        const int32_t *__restrict__ bucket_start, int32_t *__restrict__ bucket_cursor,
                                                        int32_t *__restrict__ bucket_rows, const float2 *__restrict__ nr, int metric,
                                                        uint32_t *__restrict__ bucket_ab) {
    // bucket_ab: the point's band constants (r_x + ACC |x|, |x| + r_x), each rounded UP to 16 bits (bf16), next to its id:
    // read here in row order (coalesced); pass 2 would gather them one 128-byte line per 8 bytes
    __shared__ int cnt[256], base[256];
    const int t = blockIdx.y, nslots = 1 << L1;
    for (int q = threadIdx.x; q < nslots; q += 256) cnt[q] = 0;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * 4096;
    int myc[16], myr[16];
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int64_t r = r0 + it * 256 + threadIdx.x;
        const int c = r < nrows ? code[(int64_t)t * nrows + r] : -1;
        myc[it] = c;
        myr[it] = c >= 0 ? atomicAdd(&cnt[c - t * nslots], 1) : 0;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < nslots; q += 256) base[q] = cnt[q] ? atomicAdd(&bucket_cursor[t * nslots + q], cnt[q]) : 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 16; it++) {
        const int c = myc[it];
        if (c < 0) continue;
        const int64_t r = r0 + it * 256 + threadIdx.x;
        const int at = bucket_start[c] + base[c - t * nslots] + myr[it];
        bucket_rows[at] = (int32_t)(row_lo + r);
        const float2 nrv = nr[row_lo + r];
        const float xnm = metric == 0 ? sqrtf(nrv.x) : nrv.x;
        bucket_ab[at] = (rp_bf16_up(nrv.y + RP_ACC * xnm) << 16) | rp_bf16_up(xnm + nrv.y);
    }
}

// one wave per bucket: the subtree below the bucket's root, breadth first, up to R nodes: bt_node[slot] = node id,
// bt_child[slot][c] = the child's code: < TAG a slot of this table, >= TAG a node beyond it (id + TAG), <= -2 a cell
// bt_rec: the records themselves, in slot order, children replaced by their codes: what pass 2 copies into LDS in one
// coalesced sweep (gathering them per work item through bt_node cost two dependent fetches per 16 bytes, with nothing
// else running on the CU)
__global__ __launch_bounds__(64) void k_bucket_tables(const unsigned char *__restrict__ pack, int rec, int dp,
                                                      const int32_t *__restrict__ bucket_root, int R, int32_t *__restrict__ bt_cnt,
                                                      int32_t *__restrict__ bt_node, int32_t *__restrict__ bt_child,
                                                      unsigned char *__restrict__ bt_rec, int32_t *__restrict__ bt_cell) {
    // Cells below the staged nodes get LOCAL numbers (child code -2 - local number; bt_cell[local] = the cell, at most
    // R + 1 of them): pass 2 counts a run's points per cell in LDS and reserves their slots with one global atomic per
    // cell and run -- the points of a bucket all fall into its few dozen cells, and hundreds of same-address atomics from
    // every workgroup working on the bucket are serialised at the L2.
    const int b = blockIdx.x, lane = nnd_lane();
    int32_t *nodes = bt_node + (size_t)b * R;
    int32_t *child = bt_child + (size_t)b * R * 2;
    int32_t *cells = bt_cell + (size_t)b * (R + 8);
    const int root = bucket_root[b];
    if (root < 0) {
        if (lane == 0) bt_cnt[b] = 0;
        return;
    }
    int ccount = 0;
    __shared__ int32_t q_nodes[1024];  // R <= 1024
    if (lane == 0) q_nodes[0] = root;
    int count = 1, lb = 0, le = 1;
    nnd_wave_lds_sync();
    while (lb < le) {
        for (int i0 = lb; i0 < le; i0 += 64) {  // wave-uniform bounds
            const int idx = i0 + lane;
            int c0 = -1, c1 = -1;
            if (idx < le) {
                const uint4 meta = *(const uint4 *)(pack + (int64_t)q_nodes[idx] * rec + 2 * dp);
                c0 = (int)meta.z;
                c1 = (int)meta.w;
            }
            const unsigned long long m0 = __ballot(idx < le && c0 >= 0), m1 = __ballot(idx < le && c1 >= 0);
            const int p0 = count + nnd_prefix_popc(m0), p1 = count + __popcll(m0) + nnd_prefix_popc(m1);
            const unsigned long long z0 = __ballot(idx < le && c0 <= -2), z1 = __ballot(idx < le && c1 <= -2);
            const int l0 = ccount + nnd_prefix_popc(z0), l1 = ccount + __popcll(z0) + nnd_prefix_popc(z1);
            ccount += __popcll(z0) + __popcll(z1);
            if (idx < le) {
                int k0 = c0, k1 = c1;
                if (c0 <= -2) { cells[l0] = -2 - c0; k0 = -2 - l0; }
                if (c1 <= -2) { cells[l1] = -2 - c1; k1 = -2 - l1; }
                if (c0 >= 0) {
                    if (p0 < R) { q_nodes[p0] = c0; k0 = p0; } else k0 = c0 + RP_GLOBAL_TAG;
                }
                if (c1 >= 0) {
                    if (p1 < R) { q_nodes[p1] = c1; k1 = p1; } else k1 = c1 + RP_GLOBAL_TAG;
                }
                child[2 * idx] = k0;
                child[2 * idx + 1] = k1;
            }
            count += __popcll(m0) + __popcll(m1);
            if (count > R) count = R;
            nnd_wave_lds_sync();
        }
        lb = le;
        le = count;
    }
    for (int i = lane; i < count; i += 64) nodes[i] = q_nodes[i];
    if (lane == 0) bt_cnt[b] = count;
    __threadfence_block();  // this wave's child codes (global stores above) are read back below
    const int r16 = rec >> 4;
    uint4 *dst = (uint4 *)(bt_rec + (size_t)b * R * rec);
    for (int q = lane; q < count * r16; q += 64) {
        const int sl = q / r16, wd = q - sl * r16;
        uint4 v = ((const uint4 *)(pack + (int64_t)q_nodes[sl] * rec))[wd];
        if (wd == r16 - 1) {
            v.z = (uint32_t)child[2 * sl];
            v.w = (uint32_t)child[2 * sl + 1];
        }
        dst[q] = v;
    }
}

// pass 2.  Persistent workgroups over the work items (bucket, run); a pair of lanes per point.
template <int NC>
__global__ __launch_bounds__(1024) void k_route_bucket(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                      const float2 *__restrict__ nr, int metric, int dp, int64_t n, int64_t row_lo,
                                                      int64_t nrows, int L1, int chunk, const unsigned char *__restrict__ pack,
                                                      const float *__restrict__ node_hf, int hs, const int2 *__restrict__ items,
                                                      const int32_t *__restrict__ n_items, const int32_t *__restrict__ bucket_start,
                                                      const int32_t *__restrict__ bucket_rows, const uint32_t *__restrict__ bucket_ab, int R,
                                                      const int32_t *__restrict__ bt_cnt, const int32_t *__restrict__ bt_node,
                                                      const unsigned char *__restrict__ bt_rec, const int32_t *__restrict__ bt_cell, uint32_t seed, int32_t *__restrict__ cell_count,
                                                      int32_t *__restrict__ cell_of, int32_t *__restrict__ rank_of,
                                                      const float *__restrict__ scal) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sub_tab[];
    constexpr int NC2 = 2 * NC;
    const int rec = 2 * dp + 16, r16 = rec >> 4;
    const float inv_s2 = scal[1];
    int32_t *lcnt = (int32_t *)(sub_tab + (size_t)R * rec);  // (R + 8) points of this run per local cell
    int32_t *lbase = lcnt + (R + 8);                          // (R + 8) first slot reserved for them in the cell
    int32_t *lcell = lbase + (R + 8);                         // (R + 8) local -> global cell number
    uint32_t *rowbuf = (uint32_t *)(lcell + (R + 8));         // (chunk) per point of the run: local cell << 16 | rank inside the run
    const int sub = threadIdx.x & 1, ppb = blockDim.x >> 1, pair = threadIdx.x >> 1;
    const int ni = n_items[0];
#ifndef NND_ROUTE_CONTIG
    const int item0 = (int)blockIdx.x, item1 = ni, istep = (int)gridDim.x;
#else
    const int per = (ni + (int)gridDim.x - 1) / (int)gridDim.x;
    const int item0 = (int)blockIdx.x * per, item1 = item0 + per < ni ? item0 + per : ni, istep = 1;
#endif
    int cur_b = -1;
    for (int item = item0; item < item1; item += istep) {
        const int2 it = items[item];
        const int b = it.x;
        if (b != cur_b) {
            __syncthreads();  // the previous table is no longer read
            const int cnt = bt_cnt[b];
            const uint4 *src = (const uint4 *)(bt_rec + (size_t)b * R * rec);
            for (int q = threadIdx.x; q < cnt * r16; q += blockDim.x) ((uint4 *)sub_tab)[q] = src[q];
            for (int q = threadIdx.x; q < R + 8; q += blockDim.x) {
                lcnt[q] = 0;
                lcell[q] = q <= cnt ? bt_cell[(size_t)b * (R + 8) + q] : 0;  // a subtree of cnt staged nodes has <= cnt + 1 cells below them
            }
            cur_b = b;
            __syncthreads();
        }
        const int t = b >> L1;
        const int base = bucket_start[b] + it.y * chunk;
        int cnt_rows = bucket_start[b + 1] - base;
        if (cnt_rows > chunk) cnt_rows = chunk;
        for (int r0 = 0; r0 < cnt_rows; r0 += ppb) {
            const int idx = r0 + pair;
            const bool on = idx < cnt_rows;
            const int64_t pi = bucket_rows[base + (on ? idx : 0)];
            const uint32_t ab = bucket_ab[base + (on ? idx : 0)];
            uint4 xq[NC2];
            {
                const uint4 *row = (const uint4 *)(xh + pi * dp);
#pragma unroll
                for (int q = 0; q < NC2; q++) xq[q] = row[sub + 2 * q];
            }
            const float bA = __uint_as_float(ab & 0xFFFF0000u), bB = __uint_as_float(ab << 16);
            const int64_t slot = (int64_t)t * n + pi;
            int cur = on ? 0 : -1;  // >= 0: code of the node the walk stands at; -1: over
            int rk = -1;            // local cell << 16 | rank: the LDS atomic's return value is not touched before the walk is over
            for (int depth = L1;; depth++) {
                if (!__ballot(cur >= 0)) break;  // wave-uniform
                uint4 p[NC2], meta;
                const bool glob = cur >= RP_GLOBAL_TAG;  // the record comes from global memory: its cell children are GLOBAL cell numbers
                if (glob) {  // beyond the staged subtree (pair-uniform)
                    const uint4 *r8 = (const uint4 *)(pack + (int64_t)(cur - RP_GLOBAL_TAG) * rec);
#pragma unroll
                    for (int q = 0; q < NC2; q++) p[q] = r8[sub + 2 * q];
                    meta = r8[dp >> 3];
                    if ((int)meta.z >= 0) meta.z += RP_GLOBAL_TAG;
                    if ((int)meta.w >= 0) meta.w += RP_GLOBAL_TAG;
                } else {
                    const uint4 *r8 = (const uint4 *)(sub_tab + (size_t)(cur >= 0 ? cur : 0) * rec);
#pragma unroll
                    for (int q = 0; q < NC2; q++) p[q] = r8[sub + 2 * q];
                    meta = r8[dp >> 3];
                }
                float acc = 0.0f;
#pragma unroll
                for (int q = 0; q < NC2; q++) acc = rp_dot8(xq[q], p[q], acc);
                if (cur < 0) continue;  // whole pair
                const float off = __uint_as_float(meta.x), hnorm = __uint_as_float(meta.y & 0xFFFF0000u), rh = __uint_as_float(meta.y << 16);
                float m = rp_pair_sum(acc) * inv_s2 + off;
                const float band = hnorm * bA + rh * bB + RP_EPS;
                if (RP_IN_BAND(m, band)) {  // inside the screening error band: the exact f32 margin decides (pair-uniform)
                    const int nd = glob ? cur - RP_GLOBAL_TAG : bt_node[(size_t)b * R + cur];
                    m = rp_exact_pair<NC>(xp + pi * dp, node_hf + (int64_t)nd * hs, sub) + off;
                }
                int side;
                if (fabsf(m) < RP_EPS) side = (int)(nnd_hash3(seed ^ 0x5bd1e995u, (uint32_t)slot, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
                else side = m > 0.0f ? 0 : 1;                                                                             // rp_trees.py:386-391
                const int nxt = (int)(side ? meta.w : meta.z);
                if (nxt <= -2) {  // reached a cell
                    if (sub == 0) {
                        const int cell = -2 - nxt;
                        if (glob) {  // (rare) a cell below a node that is not staged: straight to the global counter
                            const int64_t o = (int64_t)t * nrows + (pi - row_lo);
                            cell_of[o] = cell;
                            rank_of[o] = atomicAdd(&cell_count[cell], 1);
                        } else {
                            rk = (cell << 16) | atomicAdd(&lcnt[cell], 1);  // LDS
                        }
                    }
                    cur = -1;
                } else {
                    cur = nxt;
                }
            }
            if (sub == 0 && on) rowbuf[idx] = (uint32_t)rk;
        }
        // the run's points are counted: one global atomic per cell reserves their slots, then every point gets its own
        __syncthreads();
        for (int q = threadIdx.x; q < R + 8; q += blockDim.x) {
            const int c = lcnt[q];
            if (c) {
                lbase[q] = atomicAdd(&cell_count[lcell[q]], c);
                lcnt[q] = 0;
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < cnt_rows; idx += blockDim.x) {
            const uint32_t v = rowbuf[idx];
            if (v == 0xFFFFFFFFu) continue;  // placed through the global counter
            const int lc = (int)(v >> 16);
            const int64_t o = (int64_t)t * nrows + (bucket_rows[base + idx] - row_lo);
            cell_of[o] = lcell[lc];
            rank_of[o] = lbase[lc] + (int)(v & 0xFFFFu);
        }
        __syncthreads();  // rowbuf / lcnt are reused by the next run
    }
}

// cell -> depth of its node in the recorded tree (sample positions with a leaf mark start a cell)
__global__ void k_cell_depths(const uint8_t *__restrict__ leaf_flag, const int32_t *__restrict__ leafscan,
                              const int32_t *__restrict__ leaf_depth, int64_t ps, int32_t *__restrict__ cell_depth) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < ps && leaf_flag[g]) cell_depth[leafscan[g]] = leaf_depth[g];
}

// cells -> work lists: <= fin_small points: one wave per cell; <= fin_max: one workgroup per cell (ids in LDS); longer:
// the global-memory variant.  Cells that are final leaves already go through a finisher too (it writes them in
// canonical id order).  One atomic per wave and class; the order of the lists is immaterial.
__global__ void k_cell_lists(const int32_t *__restrict__ cell_count, const int32_t *__restrict__ cell_start,
                             const int32_t *__restrict__ cell_depth, int n_cells, int fin_small, int fin_max,
                             int32_t *__restrict__ small_list, int32_t *__restrict__ fin_start, int32_t *__restrict__ fin_len,
                             int32_t *__restrict__ fin_depth, int32_t *__restrict__ big_start, int32_t *__restrict__ big_len,
                             int32_t *__restrict__ big_depth, int64_t list_stride,
                             long long *__restrict__ counts /* [0] fin, [1] big, [2] small */) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int len = c < n_cells ? cell_count[c] : 0;
    const int cls = len <= 0 ? -1 : (len <= fin_small ? 2 : (len <= fin_max ? 0 : 1));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const unsigned long long m = __ballot(cls == k);
        if (!m) continue;  // wave-uniform
        long long base = 0;
        if (nnd_lane() == __builtin_ctzll(m)) base = (long long)atomicAdd((unsigned long long *)&counts[k], (unsigned long long)__popcll(m));
        base = ((long long)__shfl((int)(base >> 32), __builtin_ctzll(m), 64) << 32) | (unsigned)__shfl((int)base, __builtin_ctzll(m), 64);
        if (cls == k) {
            const int64_t idx = base + nnd_prefix_popc(m);
            int32_t *st = k == 2 ? small_list : (k == 0 ? fin_start : big_start);
            int32_t *ln = k == 2 ? small_list + list_stride : (k == 0 ? fin_len : big_len);
            int32_t *dp_ = k == 2 ? small_list + 2 * list_stride : (k == 0 ? fin_depth : big_depth);
            st[idx] = cell_start[c];
            ln[idx] = len;
            dp_[idx] = cell_depth[c];
        }
    }
}

__global__ void k_place(const int32_t *__restrict__ cell_of, const int32_t *__restrict__ rank_of,
                        const int32_t *__restrict__ cell_start, int64_t nrows, int64_t row_lo, int32_t *__restrict__ perm) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int64_t slot = (int64_t)blockIdx.y * nrows + r;
    perm[cell_start[cell_of[slot]] + rank_of[slot]] = (int32_t)(row_lo + r);
}

// -------------------------------------------------------------- host side --
static int run_scan(nnd_ctx *ctx, int mode, const int32_t *pos_seg, uint8_t *bytes, int32_t *total_dev, int64_t P, int64_t n,
                    const int32_t *perm = nullptr) {
    int nb = (int)((P + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, mode, pos_seg, bytes, P, ctx->scan_blk, perm,
                       ctx->side_pt, n);
    hipLaunchKernelGGL(k_scan_apply<true>, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, mode == 2 ? 0 : mode, pos_seg, bytes, P,
                       ctx->scan_blk, ctx->scan_out, total_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// exclusive scan of data[0 .. n) in place, the grand total to total_dev[0] (ctx->scan_blk holds the tile sums)
static int scan_i32_inplace(nnd_ctx *ctx, int32_t *data, int64_t n, int32_t *total_dev) {
    if (n <= SCAN_TILE) {
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, data, (int)n, total_dev);
    } else {
        const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
        hipLaunchKernelGGL(k_scan_i32_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, data, n, ctx->scan_blk);
        hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, ctx->scan_blk, nb, total_dev);
        hipLaunchKernelGGL(k_scan_i32_apply, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, data, n, ctx->scan_blk);
    }
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

static size_t fin_smem_bytes(int dp, int cap /* 0: ids in global memory */) {
    const size_t tail = sizeof(float) * (dp + 4) + sizeof(uint16_t) * dp + sizeof(int32_t) * (FIN_STACK * 4 + FIN_WS);
    return (size_t)cap * 9 + tail;
}

// What the level-synchronous passes run on: the whole point set (small n), or the compact sample (recording the tree).
struct forest_view {
    const float *xp;
    const uint16_t *xh;
    const float2 *nr;  // per row: (|x|^2 -- 1 / 0 for normalised rows --, |x - bf16(x)|: the point's share of the screening band)
    int64_t n, P;     // points per tree, n_trees * n
    int leaf_size;    // split while len > leaf_size (rp_trees.py:2188)
    int fin_max;      // children of <= fin_max points leave the passes for k_finish_subtrees (0: never)
    bool record;      // keep hyperplanes and children of every node (routing pass)
    int cur = 0, depth = 0;
    int64_t n_nodes = 0;
    std::vector<int64_t> level_base;  // recording: first node id of every level (+ the total at the end)
    int T = 0;                        // trees in the view (0: ctx->p.n_trees)
    int tree_bias = 0;                // global number of the view's first tree (sharded build: tops split by tree)
    int64_t n_low = 0, high_lo = 0;   // recording: node ids in use are [0, n_low) and [high_lo, node_cap) (k_pack_nodes compacts)
};

// big list (global-memory variant; its nodes join the workgroup list as they shrink) -> workgroup list -> small list
static int launch_finishers(nnd_ctx *ctx, int32_t *perm, int32_t *other, const int32_t *big_start, const int32_t *big_len,
                            const int32_t *big_depth, int depth0, long long n_big, long long n_small = 0, rp_tree_map tm = rp_tree_map{}) {
    const int dp = ctx->dp, angular = ctx->p.metric == NND_METRIC_ALT_COSINE;
    int32_t *fin_start = ctx->seg_child + 2 * ctx->max_segs;  // finisher work list lives behind seg_child
    int32_t *fin_len = fin_start + ctx->max_segs;
    int32_t *fin_depth = fin_len + ctx->max_segs;
    long long *fin_count = ctx->counters + CNT_SCRATCH + 1;
#define FIN_ARGS(st, ln, dpth, d0, cnt) ctx->xp, ctx->xh, ctx->nr2, ctx->p.metric, dp, ctx->n, perm, st, ln, dpth, d0, (int)(cnt), angular, \
                 ctx->tree_seed, ctx->p.max_depth, ctx->p.leaf_size, ctx->leaf_flag
    if (n_small > 0) {  // one wave per cell
        const int32_t *sl = ctx->small_list;
        hipLaunchKernelGGL((k_finish_subtrees<false, 64, FIN_SMALL>), dim3((unsigned)n_small), dim3(64), fin_smem_bytes(dp, FIN_SMALL),
                           ctx->stream, FIN_ARGS(sl, sl + ctx->cell_cap, sl + 2 * ctx->cell_cap, 0, n_small), (int32_t *)nullptr,
                           (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, fin_count, tm, ctx->mean + ctx->dp);
        NND_HIP_CHECK(hipGetLastError());
    }
    if (n_big > 0) {
        hipLaunchKernelGGL((k_finish_subtrees<true, 256, 0>), dim3((unsigned)n_big), dim3(256), fin_smem_bytes(dp, 0), ctx->stream,
                           FIN_ARGS(big_start, big_len, big_depth, depth0, n_big), other, ctx->side, FIN_MAX, fin_start, fin_len,
                           fin_depth, fin_count, tm, ctx->mean + ctx->dp);
        NND_HIP_CHECK(hipGetLastError());
    }
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 34, fin_count, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const long long nfin = ctx->h_pin[34];
    if (nfin > ctx->max_segs) {
        ctx->set_error("rp-forest: %lld finisher segments exceed the allocation of %lld", nfin, (long long)ctx->max_segs);
        return 1;
    }
    if (nfin > 0) {
        hipLaunchKernelGGL((k_finish_subtrees<false, 256, FIN_MAX>), dim3((unsigned)nfin), dim3(256), fin_smem_bytes(dp, FIN_MAX),
                           ctx->stream, FIN_ARGS(fin_start, fin_len, fin_depth, 0, nfin), (int32_t *)nullptr, (uint8_t *)nullptr,
                           FIN_MAX, fin_start, fin_len, fin_depth, fin_count, tm, ctx->mean + ctx->dp);
        NND_HIP_CHECK(hipGetLastError());
    }
#undef FIN_ARGS
    return 0;
}

// The level-synchronous passes on `v`.  Returns 0, 1 (error) or 2 (recording ran out of node slots: caller falls back).
static int forest_levels(nnd_ctx *ctx, forest_view &v) {
    const int64_t n = v.n, P = v.P;
    const int T = v.T > 0 ? v.T : ctx->p.n_trees, dp = ctx->dp, leaf_size = v.leaf_size, max_depth = ctx->p.max_depth;
    const uint32_t pos_bias = (uint32_t)((int64_t)v.tree_bias * n);
    rp_tree_map tm;
    tm.tree_bias = v.tree_bias;
    const int angular = ctx->p.metric == NND_METRIC_ALT_COSINE;
    const int hs = dp + 4;
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
    const int fin_max = v.fin_max;
    int32_t *fin_start = ctx->seg_child + 2 * ctx->max_segs;  // finisher work list lives behind seg_child
    int32_t *fin_len = fin_start + ctx->max_segs;
    int32_t *fin_depth = fin_len + ctx->max_segs;
    int64_t node_base = 0;
    // (a shard's host waits go through its communicator -- abort flag, timeout -- so it keeps the copy + wait form)
    long long *flag_words = (ctx->h_pin_dev && !ctx->wait_hook && !nnd_knob("NND_NO_LEVEL_FLAG")) ? ctx->h_pin_dev + 40 : nullptr;
    NND_HIP_CHECK(hipMemsetAsync(ctx->counters + CNT_SCRATCH + 1, 0, sizeof(long long), ctx->stream));
    if (!v.record && S > 0 && n <= fin_max) {  // small point sets: the roots go straight to the finisher
        std::vector<int32_t> h_s(T), h_l(T), h_d(T, 0);
        for (int t = 0; t < T; t++) { h_s[t] = (int32_t)(t * n); h_l[t] = (int32_t)n; }
        NND_HIP_CHECK(hipMemcpyAsync(fin_start, h_s.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(fin_len, h_l.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
        NND_HIP_CHECK(hipMemcpyAsync(fin_depth, h_d.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
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
        if (v.record && node_base + S > ctx->node_cap) {
            if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
            return 2;
        }
        if (v.record) v.level_base.push_back(node_base);
        // recording: this level's hyperplanes are written straight into the node tables at [node_base, node_base + S)
        float *hyper = v.record ? ctx->node_hf + node_base * hs : ctx->hyper;
        uint16_t *hyper_h = v.record ? ctx->node_hh + node_base * dp : ctx->hyper_h;
        hipLaunchKernelGGL(k_hyperplane, dim3((unsigned)((S + 3) / 4)), dim3(256), 0, ctx->stream, v.xp, dp,
                           ctx->perm[cur], ctx->seg_start[cur], ctx->seg_len[cur], (int)S, angular, ctx->tree_seed, depth,
                           hyper, hs, hyper_h, pos_bias, ctx->mean + ctx->dp);
        // point-major pass: hyperplane table fits in L2 AND enough positions are still active to amortise
        // streaming every row once (it costs n rows regardless of how many positions are active)
        const bool fused = inv_live && (S * (int64_t)dp * 2 <= (int64_t)6 << 20) && (active_pos * 2 >= 3 * n);
        if (fused) {
            hipLaunchKernelGGL(k_margin_fused, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ctx->stream, v.xp, v.xh, v.nr,
                               ctx->p.metric, dp, n, T, ctx->inv, hyper, hs, hyper_h, ctx->tree_seed, depth, ctx->side_pt, pos_bias, ctx->mean + ctx->dp);
        } else {
            inv_live = false;
            hipLaunchKernelGGL(k_margin, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, ctx->stream, v.xp, v.xh, v.nr,
                               ctx->p.metric, dp, ctx->perm[cur], ctx->pos_seg[cur], P, hyper, hs, hyper_h, ctx->tree_seed, depth,
                               ctx->side, pos_bias, ctx->mean + ctx->dp);
        }
        if (run_scan(ctx, fused ? 2 : 0, ctx->pos_seg[cur], ctx->side, scan_total, P, n, ctx->perm[cur])) return 1;
        hipLaunchKernelGGL(k_seg_count, dim3((unsigned)((S + 255) / 256)), dim3(256), 0, ctx->stream, ctx->seg_start[cur],
                           ctx->seg_len[cur], (int)S, ctx->scan_out, scan_total, P, ctx->seg_nleft);
        int child_can_split = (max_depth - (depth + 1)) > 0 ? 1 : 0;
        // Recording: once only a few sample positions are still in splittable nodes the recorded tree stops: the
        // children of this level all become cells, however long (a straggler level costs a full pass over the sample
        // for a handful of nodes; an over-long cell just goes to a workgroup finisher instead of a single wave).
        if (v.record && ctx->early_stop > 0 && active_pos * ctx->early_stop < P) child_can_split = 0;
        hipLaunchKernelGGL(k_children, dim3(1), dim3(256), 0, ctx->stream, ctx->seg_start[cur], ctx->seg_len[cur],
                           ctx->seg_nleft, (int)S, leaf_size, child_can_split, fin_max, depth + 1, ctx->seg_start[1 - cur],
                           ctx->seg_len[1 - cur], ctx->seg_child, ctx->leaf_flag, fin_start, fin_len, fin_depth, ctx->counters,
                           v.record ? ctx->node_child : (int32_t *)nullptr, (int)node_base, (int)(node_base + S), ctx->s_leaf_depth,
                           (int)ctx->node_cap - 1, flag_words, flag_words ? ++ctx->flag_seq : 0);
        hipLaunchKernelGGL(k_scatter, dim3(gridP), dim3(256), 0, ctx->stream, ctx->perm[cur], ctx->pos_seg[cur], ctx->side,
                           ctx->scan_out, ctx->seg_start[cur], ctx->seg_nleft, ctx->seg_child, P, n, ctx->perm[1 - cur],
                           ctx->pos_seg[1 - cur], inv_live ? ctx->inv : (int32_t *)nullptr);
        NND_HIP_CHECK(hipGetLastError());
        // one small hand-over per level: the number of segments that stay in the level-synchronous passes
        static_assert(CNT_LEAVES == CNT_ACTIVE_SEGS + 1 && CNT_SCRATCH == CNT_LEAVES + 1, "counter layout");
        long long *next = ctx->h_pin + 32;  // CNT_ACTIVE_SEGS, CNT_LEAVES, CNT_SCRATCH.. are adjacent; pinned words
        if (flag_words) {  // written by k_children itself (see there): no copy, no stream synchronisation
            next = ctx->h_pin + 40;
            volatile long long *seqw = ctx->h_pin + 46;
            const auto t_spin = std::chrono::steady_clock::now();
            unsigned spins = 0;
            while (*seqw != ctx->flag_seq) {
                if ((++spins & 0xFFFFu) == 0 && std::chrono::steady_clock::now() - t_spin > std::chrono::seconds(30)) {
                    ctx->set_error("rp-forest: the device did not hand over the segment count of level %d within 30 s", depth);
                    return 1;
                }
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        } else {
            NND_HIP_CHECK(hipMemcpyAsync(next, ctx->counters + CNT_ACTIVE_SEGS, 6 * sizeof(long long), hipMemcpyDeviceToHost,
                                         ctx->stream));
            NND_HIP_CHECK(nnd_sync_spin(ctx));
        }
        node_base += S;
        S = next[0];
        active_pos = next[1];
        cur = 1 - cur;
        depth++;
        // Tail of the level loop (whole-set mode): once most positions have been handed over, a level-synchronous pass
        // still costs P positions per kernel for a few hundred segments: the global-memory variant of the finisher
        // takes them (its nodes are split in place until they fit the LDS finisher, whose work list they join).
        if (!v.record && S > 0 && (active_pos * 2 < 3 * n) && next[5] <= BIG_MAX) {  // next[5] = CNT_SCRATCH + 3: longest stayer
            if (launch_finishers(ctx, ctx->perm[cur], ctx->perm[1 - cur], ctx->seg_start[cur], ctx->seg_len[cur], nullptr, depth, S, 0, tm)) return 1;
            v.cur = cur;
            v.depth = depth;
            v.n_nodes = node_base;
            return 0;  // both finishers have been launched
        }
    }
    v.cur = cur;
    v.depth = depth;
    v.n_nodes = node_base;
    if (v.record) {
        v.level_base.push_back(node_base);
        // the subtrees that left the passes are recorded one workgroup each (k_finish_subtrees<.., RECORD>)
        NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 34, ctx->counters + CNT_SCRATCH + 1, sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
        NND_HIP_CHECK(nnd_sync_spin(ctx));
        const long long nfin = ctx->h_pin[34];
        if (nfin > ctx->max_segs || node_base + nfin + 1 >= ctx->node_cap) {
            if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
            return 2;
        }
        v.n_low = node_base;
        v.high_lo = ctx->node_cap;
        if (nfin > 0) {
            int *flags = (int *)(ctx->counters + CNT_SCRATCH + 2);  // [0] id counter, [1] overflow
            NND_HIP_CHECK(hipMemsetAsync(flags, 0, 2 * sizeof(int), ctx->stream));
            rp_record rec{ctx->node_hf, ctx->node_hh, ctx->node_child, ctx->s_leaf_depth, hs, (int)ctx->node_cap - 1,
                          (int)(ctx->node_cap - 1 - nfin), (int)node_base, flags, flags + 1, 0, FIN_MAX};
            // short subtrees: one wave each (a chain of dependent latencies wants concurrency, not width); the rest: a workgroup
            rp_record rs = rec, rl = rec;
            rs.min_len = 0; rs.max_len = FIN_SMALL;
            rl.min_len = FIN_SMALL + 1; rl.max_len = FIN_MAX;
            hipLaunchKernelGGL((k_finish_subtrees<false, 64, FIN_SMALL, true>), dim3((unsigned)nfin), dim3(64), fin_smem_bytes(dp, FIN_SMALL),
                               ctx->stream, v.xp, v.xh, v.nr, ctx->p.metric, dp, n, ctx->perm[cur], fin_start, fin_len, fin_depth, 0,
                               (int)nfin, angular, ctx->tree_seed, max_depth, leaf_size, ctx->leaf_flag, (int32_t *)nullptr,
                               (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, ctx->counters + CNT_SCRATCH + 1, tm, ctx->mean + ctx->dp, rs);
            if (v.fin_max > FIN_SMALL)
                hipLaunchKernelGGL((k_finish_subtrees<false, 256, FIN_MAX, true>), dim3((unsigned)nfin), dim3(256), fin_smem_bytes(dp, FIN_MAX),
                                   ctx->stream, v.xp, v.xh, v.nr, ctx->p.metric, dp, n, ctx->perm[cur], fin_start, fin_len, fin_depth, 0,
                                   (int)nfin, angular, ctx->tree_seed, max_depth, leaf_size, ctx->leaf_flag, (int32_t *)nullptr,
                                   (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, ctx->counters + CNT_SCRATCH + 1, tm, ctx->mean + ctx->dp, rl);
            NND_HIP_CHECK(hipGetLastError());
            NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 38, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            NND_HIP_CHECK(nnd_sync_spin(ctx));
            if (((const int *)(ctx->h_pin + 38))[1]) {  // node tables exhausted
                if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
                return 2;
            }
            // ids taken downwards: the segments' own (node_cap - nfin ..), then one per recorded split below them
            v.high_lo = ctx->node_cap - nfin - ((const int *)(ctx->h_pin + 38))[0];
        }
        v.n_nodes = ctx->node_cap;  // ids are spread over the table: level-synchronous nodes upwards, recorded subtrees downwards
        return 0;
    }
    if (launch_finishers(ctx, ctx->perm[cur], ctx->perm[1 - cur], nullptr, nullptr, nullptr, 0, 0, 0, tm)) return 1;
    return 0;
}

static int device_cus(nnd_ctx *ctx, int *out) {
    static int n_cu_dev[64] = {0};
    int &n_cu = n_cu_dev[ctx->p.device & 63];
    if (n_cu == 0) {
        hipDeviceProp_t prop;
        NND_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->p.device));
        n_cu = prop.multiProcessorCount;
    }
    *out = n_cu;
    return 0;
}

// What a routing pass reads and writes: packed trees (k_pack_nodes), the rows [row_lo, row_lo + nrows) of the prepared point
// set, and per (tree, row): the cell and the slot inside it (cell_of / rank_of at [t * nrows + row - row_lo]).
struct rp_route_io {
    const unsigned char *pack;
    const float *hf;          // f32 hyperplanes of the packed nodes (exact rechecks)
    const int32_t *roots;     // device (T): root node of every tree
    int T;
    int64_t row_lo, nrows;
    int64_t n_cells;          // cells of all T trees (sizes the subtree tables)
    int32_t *cell_count, *cell_of, *rank_of;
    int32_t *code, *bucket_rows, *bucket_ab;  // scratch, T * nrows int32 each (coherent form only)
};

template <int NC, int TB>
static int launch_route(nnd_ctx *ctx, const rp_route_io &io, int n_top, int l_top) {
    auto kern = k_route<NC, TB>;
    const size_t smem = (size_t)n_top * (2 * ctx->dp + 16);
    static bool attr_dev[64] = {false};
    if (!attr_dev[ctx->p.device & 63]) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        attr_dev[ctx->p.device & 63] = true;
    }
    int n_cu = 0;
    if (device_cus(ctx, &n_cu)) return 1;
    int64_t blocks = (ctx->n + 127) / 128;
    if (blocks > 2 * (int64_t)n_cu) blocks = 2 * (int64_t)n_cu;  // persistent: two 512-thread workgroups per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), smem, ctx->stream, ctx->xp, ctx->nr2, ctx->p.metric, ctx->dp, ctx->n,
                       io.T, io.pack, io.hf, ctx->dp + 4, ctx->tree_seed, io.cell_count, io.cell_of, io.rank_of, n_top, l_top, ctx->mean + ctx->dp);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// the plain walk (whole point set, the handle's own trees): kept for the comparison test and as the fallback
static int route_plain(nnd_ctx *ctx, const rp_route_io &io, const forest_view &v) {
    const int dp = ctx->dp;
    // levels whose records fit the route kernel's LDS copy (<= 72 KB: two workgroups per CU)
    int l_top = 0;
    while (l_top + 1 < (int)v.level_base.size() && v.level_base[l_top + 1] * (2 * dp + 16) <= 72 * 1024) l_top++;
    const int n_top = (int)v.level_base[l_top];
    switch (dp / 32) {  // NC = 8-float chunks per lane; trees per batch sized for <= 128 VGPRs
        case 1: return launch_route<1, 4>(ctx, io, n_top, l_top);
        case 2: return launch_route<2, 4>(ctx, io, n_top, l_top);
        case 3: return launch_route<3, 2>(ctx, io, n_top, l_top);
        case 4: return launch_route<4, 2>(ctx, io, n_top, l_top);
        case 5: return launch_route<5, 1>(ctx, io, n_top, l_top);
        case 6: return launch_route<6, 1>(ctx, io, n_top, l_top);
        case 7: return launch_route<7, 1>(ctx, io, n_top, l_top);
        case 8: return launch_route<8, 1>(ctx, io, n_top, l_top);
        default: break;  // wider rows: whole-set passes (nnd_create does not enable routing for them)
    }
    return 2;
}

#ifndef NND_ROUTE_L1
#define NND_ROUTE_L1 6      // levels walked in pass 1: 2^L1 buckets per tree
#endif
#ifndef NND_ROUTE_CHUNK
#define NND_ROUTE_CHUNK 2048  // points per work item of pass 2
#endif
#ifndef NND_ROUTE_RMAX
#define NND_ROUTE_RMAX 384    // most records of a bucket's subtree staged in LDS
#endif

struct rp_route_geom {
    int L1, ns, nb, R, chunk, tg;  // levels / slots per tree / buckets / staged records / run length / trees per pass-1 launch
    int64_t max_items;
    size_t o_top_rec, o_top_node, o_bucket_root, o_bcount, o_bcursor, o_bstart, o_nitems, o_btcnt, o_btnode, o_btchild, o_btrec, o_btcell, o_items, total;
};
static rp_route_geom route_geometry(int dp, int T, int64_t nrows, int64_t n_cells) {
    rp_route_geom g;
    const int rec = 2 * dp + 16;
    // 2^L1 buckets per tree: one more level when the subtrees below 64 buckets would not fit the staged table (10 M-point
    // trees: ~730 nodes per bucket at L1 = 6) -- what does not fit is walked through L2 misses
    g.L1 = NND_ROUTE_L1;
    if (g.L1 < 7 && n_cells / ((int64_t)T << g.L1) > NND_ROUTE_RMAX / 2) g.L1 = 7;
    g.ns = 1 << g.L1;
    g.nb = T * g.ns;
    // a subtree of c cells has c - 1 inner nodes, and the buckets are very uneven (six random splits: a few buckets hold
    // several times the mean, and most of the points): stage up to 6x the mean, what does not fit is walked from L2
    int64_t r = 6 * n_cells / g.nb + 16;
    const int64_t r_lds = (int64_t)150 * 1024 / rec;
    if (r > NND_ROUTE_RMAX) r = NND_ROUTE_RMAX;
    if (r > r_lds) r = r_lds;
    g.R = (int)(r < 64 ? 64 : r);
    g.chunk = NND_ROUTE_CHUNK;
    g.tg = (int)((size_t)140 * 1024 / ((size_t)g.ns * (rec + 8)));
    if (g.tg < 1) g.tg = 1;
    if (g.tg > T) g.tg = T;
    g.max_items = (int64_t)g.nb + (int64_t)T * nrows / g.chunk + 1;
    size_t at = 0;
    auto take = [&](size_t bytes) { const size_t o = at; at += (bytes + 255) & ~(size_t)255; return o; };
    g.o_top_rec = take((size_t)g.nb * rec);
    g.o_top_node = take(sizeof(int32_t) * g.nb);
    g.o_bucket_root = take(sizeof(int32_t) * g.nb);
    g.o_bcount = take(sizeof(int32_t) * g.nb);   // bcount and bcursor are adjacent: one memset
    g.o_bcursor = take(sizeof(int32_t) * g.nb);
    g.o_bstart = take(sizeof(int32_t) * (g.nb + 1));
    g.o_nitems = take(sizeof(int32_t) * 4);
    g.o_btcnt = take(sizeof(int32_t) * g.nb);
    g.o_btnode = take(sizeof(int32_t) * (size_t)g.nb * g.R);
    g.o_btchild = take(sizeof(int32_t) * (size_t)g.nb * g.R * 2);
    g.o_btrec = take((size_t)g.nb * g.R * rec);
    g.o_btcell = take(sizeof(int32_t) * (size_t)g.nb * (g.R + 8));
    g.o_items = take(sizeof(int2) * (size_t)g.max_items);
    g.total = at;
    return g;
}

template <int NC>
static int launch_route_coherent(nnd_ctx *ctx, const rp_route_io &io, const rp_route_geom &g) {
    const int dp = ctx->dp, rec = 2 * dp + 16, hs = dp + 4;
    unsigned char *ws = ctx->route_ws;
    unsigned char *top_rec = ws + g.o_top_rec;
    int32_t *top_node = (int32_t *)(ws + g.o_top_node), *bucket_root = (int32_t *)(ws + g.o_bucket_root);
    int32_t *bcount = (int32_t *)(ws + g.o_bcount), *bcursor = (int32_t *)(ws + g.o_bcursor), *bstart = (int32_t *)(ws + g.o_bstart);
    int32_t *n_items = (int32_t *)(ws + g.o_nitems), *bt_cnt = (int32_t *)(ws + g.o_btcnt), *bt_node = (int32_t *)(ws + g.o_btnode);
    int32_t *bt_child = (int32_t *)(ws + g.o_btchild);
    unsigned char *bt_rec = ws + g.o_btrec;
    int32_t *bt_cell = (int32_t *)(ws + g.o_btcell);
    int2 *items = (int2 *)(ws + g.o_items);
    auto ktop = k_route_top<NC>;
    auto kbkt = k_route_bucket<NC>;
    static bool attr_dev[64] = {false};
    if (!attr_dev[ctx->p.device & 63]) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)ktop, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kbkt, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_dev[ctx->p.device & 63] = true;
    }
    int n_cu = 0;
    if (device_cus(ctx, &n_cu)) return 1;
    NND_HIP_CHECK(hipMemsetAsync(bcount, 0, g.o_bstart - g.o_bcount, ctx->stream));  // counts and cursors
    hipLaunchKernelGGL(k_top_heap, dim3((unsigned)io.T), dim3(256), 0, ctx->stream, io.pack, rec, dp, io.roots, g.L1, top_rec, top_node,
                       bucket_root);
    hipLaunchKernelGGL(k_bucket_tables, dim3((unsigned)g.nb), dim3(64), 0, ctx->stream, io.pack, rec, dp, bucket_root, g.R, bt_cnt, bt_node,
                       bt_child, bt_rec, bt_cell);
    for (int tg0 = 0; tg0 < io.T; tg0 += g.tg) {  // pass 1, a group of trees per launch (the group's top levels fill the LDS)
        const int tgn = io.T - tg0 < g.tg ? io.T - tg0 : g.tg;
        const size_t smem = (size_t)tgn * g.ns * (rec + 8);
        int64_t blocks = (io.nrows + 511) / 512;
        if (blocks > n_cu) blocks = n_cu;  // persistent: one 1024-thread workgroup per CU
        hipLaunchKernelGGL(ktop, dim3((unsigned)blocks), dim3(1024), smem, ctx->stream, ctx->xp, ctx->xh, ctx->nr2, ctx->p.metric, dp, ctx->n, io.row_lo,
                           io.nrows, tg0, tgn, g.L1, top_rec, top_node, io.hf, hs, ctx->tree_seed, io.cell_count, io.cell_of, io.rank_of,
                           io.code, bcount, ctx->mean + ctx->dp);
    }
    hipLaunchKernelGGL(k_bucket_prefix, dim3(1), dim3(256), 0, ctx->stream, bcount, g.nb, g.chunk, bstart, items, n_items, (int)g.max_items);
    hipLaunchKernelGGL(k_bucket_scatter, dim3((unsigned)((io.nrows + 4095) / 4096), (unsigned)io.T), dim3(256), 0, ctx->stream, io.code,
                       io.nrows, io.row_lo, g.L1, bstart, bcursor, io.bucket_rows, ctx->nr2, ctx->p.metric, (uint32_t *)io.bucket_ab);
    const size_t smem2 = (size_t)g.R * rec + sizeof(int32_t) * 3 * (size_t)(g.R + 8) + sizeof(uint32_t) * (size_t)g.chunk;
    int per_cu = 0;
    NND_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kbkt, 1024, smem2));
    if (per_cu < 1) per_cu = 1;
    int64_t blocks2 = (int64_t)n_cu * per_cu;
    if (blocks2 > g.max_items) blocks2 = g.max_items;
    hipLaunchKernelGGL(kbkt, dim3((unsigned)blocks2), dim3(1024), smem2, ctx->stream, ctx->xp, ctx->xh, ctx->nr2, ctx->p.metric, dp, ctx->n,
                       io.row_lo, io.nrows, g.L1, g.chunk, io.pack, io.hf, hs, items, n_items, bstart, io.bucket_rows, (const uint32_t *)io.bucket_ab, g.R, bt_cnt, bt_node,
                       bt_rec, bt_cell, ctx->tree_seed, io.cell_count, io.cell_of, io.rank_of, ctx->mean + ctx->dp);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// pass 1 by bucket, pass 2 from LDS (see "coherent routing" above).  Returns 0 / 1 / 2 (not available for this geometry).
static int route_coherent(nnd_ctx *ctx, const rp_route_io &io) {
    const int dp = ctx->dp;
    if (dp % 32 != 0 || dp > 256 || io.T < 1 || io.T > 4096) return 2;
    const rp_route_geom g = route_geometry(dp, io.T, io.nrows, io.n_cells);
    if (g.total > ctx->route_ws_cap) {  // grow-only
        NND_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->route_ws) { NND_HIP_CHECK(hipFree(ctx->route_ws)); ctx->route_ws = nullptr; }
        ctx->route_ws_cap = 0;
        NND_HIP_CHECK(hipMalloc((void **)&ctx->route_ws, g.total + g.total / 4));
        ctx->route_ws_cap = g.total + g.total / 4;
    }
    switch (dp / 32) {
        case 1: return launch_route_coherent<1>(ctx, io, g);
        case 2: return launch_route_coherent<2>(ctx, io, g);
        case 3: return launch_route_coherent<3>(ctx, io, g);
        case 4: return launch_route_coherent<4>(ctx, io, g);
        case 5: return launch_route_coherent<5>(ctx, io, g);
        case 6: return launch_route_coherent<6>(ctx, io, g);
        case 7: return launch_route_coherent<7>(ctx, io, g);
        case 8: return launch_route_coherent<8>(ctx, io, g);
        default: break;
    }
    return 2;
}

__global__ void k_iota_i32(int32_t *__restrict__ out, int n, int base) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = base + i;
}

// The top of the view's trees from the sample (level passes + recording finisher), then the cells = leaves of the recorded
// trees, numbered in position order (tree-major): ctx->scan_out[p] = cell number of the cell that starts at sample
// position p, ctx->cell_depth filled.  Returns 0, 1, or 2 (fall back to the whole-set passes).
static int forest_tops(nnd_ctx *ctx, forest_view &v, int32_t *n_cells_out) {
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    const int rc = forest_levels(ctx, v);
    if (rc) return rc;
    if (run_scan(ctx, 1, nullptr, ctx->leaf_flag, scan_total, v.P, v.n)) return 1;
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 35, scan_total, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const int32_t n_cells = *(const int32_t *)(ctx->h_pin + 35);
    if (n_cells > ctx->cell_cap) {
        if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
        return 2;
    }
    hipLaunchKernelGGL(k_cell_depths, dim3((unsigned)((v.P + 255) / 256)), dim3(256), 0, ctx->stream, ctx->leaf_flag, ctx->scan_out,
                       ctx->s_leaf_depth, v.P, ctx->cell_depth);
    NND_HIP_CHECK(hipGetLastError());
    *n_cells_out = n_cells;
    return 0;
}

// cells (counts known) -> positions: cell_start = exclusive scan, work lists by size class, points placed, cells finished.
// n_rows_routed rows x T trees were routed (cell_of / rank_of as rp_route_io lays them out); perm is written at
// [0, sum of the counts).
static int forest_place_finish(nnd_ctx *ctx, int32_t n_cells, int T, int64_t row_lo, int64_t nrows, const int32_t *cell_of,
                               const int32_t *rank_of, int64_t P_used, rp_tree_map tm) {
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    // cell_start = exclusive scan of the counts (k_scan_blocks scans in place: copy first)
    NND_HIP_CHECK(hipMemcpyAsync(ctx->cell_start, ctx->cell_count, sizeof(int32_t) * (size_t)n_cells, hipMemcpyDeviceToDevice, ctx->stream));
    if (scan_i32_inplace(ctx, ctx->cell_start, n_cells, scan_total)) return 1;
    NND_HIP_CHECK(hipMemsetAsync(ctx->leaf_flag, 0, (size_t)P_used, ctx->stream));
    long long *counts = ctx->counters + CNT_SCRATCH + 1;  // [0] workgroup list (what launch_finishers reads), [1] big, [2] small
    NND_HIP_CHECK(hipMemsetAsync(counts, 0, 3 * sizeof(long long), ctx->stream));
    int32_t *fin_start = ctx->seg_child + 2 * ctx->max_segs, *fin_len = fin_start + ctx->max_segs, *fin_depth = fin_len + ctx->max_segs;
    int32_t *big_start = ctx->seg_start[0], *big_len = ctx->seg_len[0], *big_depth = ctx->seg_nleft;
    hipLaunchKernelGGL(k_cell_lists, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cell_count, ctx->cell_start,
                       ctx->cell_depth, (int)n_cells, FIN_SMALL, FIN_MAX, ctx->small_list, fin_start, fin_len, fin_depth, big_start,
                       big_len, big_depth, ctx->cell_cap, counts);
    if (cell_of)
        hipLaunchKernelGGL(k_place, dim3((unsigned)((nrows + 255) / 256), (unsigned)T), dim3(256), 0, ctx->stream, cell_of, rank_of,
                           ctx->cell_start, nrows, row_lo, ctx->perm[0]);
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 36, counts + 1, 2 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const long long n_big = ctx->h_pin[36], n_small = ctx->h_pin[37];
    if (n_big > ctx->max_segs) {
        if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
        return 2;
    }
    if (launch_finishers(ctx, ctx->perm[0], ctx->perm[1], big_start, big_len, big_depth, 0, n_big, n_small, tm)) return 1;
    ctx->cur = 0;
    return 0;
}

// sample forest -> routing pass -> cells -> finishers.  Returns 0, 1, or 2 (fall back to the whole-set passes).
static int forest_by_routing(nnd_ctx *ctx, int *levels_out) {
    const int64_t n = ctx->n, P = ctx->P, M = ctx->s_m, Ps = (int64_t)ctx->p.n_trees * M;
    const int T = ctx->p.n_trees, dp = ctx->dp;
    hipLaunchKernelGGL(k_gather_sample, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, ctx->stream, ctx->xp, ctx->xh, ctx->nr2,
                       dp, (int64_t)0, M, ctx->s_stride, ctx->tree_seed, ctx->xs, ctx->xsh, ctx->nr2s);
    // sample subtrees of <= 512 members leave the level-synchronous passes for the (one-wave) recording finisher;
    // 2048 (+ a workgroup class) means 4 fewer levels but a slower finisher: 6.9-7.2 ms vs 6.6 ms per forest at 1 M points
    static const int rec_fin = [] { const char *e = nnd_knob("NND_REC_FIN"); const int r = e ? atoi(e) : FIN_SMALL; return r <= FIN_SMALL ? FIN_SMALL : FIN_MAX; }();
    forest_view v{ctx->xs, ctx->xsh, ctx->nr2s, M, Ps, ctx->cell_leaf, rec_fin, true};
    int32_t n_cells = 0;
    int rc = forest_tops(ctx, v, &n_cells);
    if (rc) return rc;
    if (n_cells + P / (ctx->p.leaf_size + 1) > ctx->max_segs) {
        if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
        return 2;
    }
    NND_HIP_CHECK(hipMemsetAsync(ctx->cell_count, 0, sizeof(int32_t) * (size_t)n_cells, ctx->stream));
    const int64_t n_packed = v.n_low + (ctx->node_cap - v.high_lo);
    rp_pack_map mp{v.n_low, v.high_lo, 0, 0, nullptr};
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((n_packed + 15) / 16)), dim3(256), 0, ctx->stream, ctx->node_hh, ctx->node_hf,
                       dp + 4, ctx->node_child, ctx->scan_out, dp, n_packed, mp, ctx->node_pack, ctx->node_hfc);
    int32_t *roots = ctx->route_roots;
    hipLaunchKernelGGL(k_iota_i32, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, ctx->stream, roots, T, 0);
    // ctx->inv (segment table of the sample passes) and ctx->scan_out (the cell numbers: baked into the records by now)
    // are free: they hold the (tree, point) codes and the bucket-sorted point lists of the coherent form
    rp_route_io io{ctx->node_pack, ctx->node_hfc, roots, T, 0, n, n_cells, ctx->cell_count, ctx->pos_seg[0], ctx->pos_seg[1], ctx->inv, ctx->scan_out, ctx->perm[1]};
    int rrc = 2;
    if (!(ctx->p.flags & NND_FLAG_TEST_ROUTE_PLAIN)) rrc = route_coherent(ctx, io);
    if (rrc == 2) rrc = route_plain(ctx, io, v);
    if (rrc) return rrc;
    rc = forest_place_finish(ctx, n_cells, T, 0, n, ctx->pos_seg[0], ctx->pos_seg[1], P, rp_tree_map{});
    if (rc) return rc;
    *levels_out = v.depth;
    ctx->stats.n_cells = n_cells;
    return 0;
}

// leaf tables of a finished position space [0, P): leaves = runs between leaf marks; tree t's leaves are those from
// position t * n (tree_begin == nullptr) or tree_begin[t] on
static int forest_leaf_tables(nnd_ctx *ctx, int64_t P, int T, const int32_t *tree_begin_dev) {
    const int64_t n = ctx->n;
    const int leaf_size = ctx->p.leaf_size;
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);  // device scratch word(s)
    unsigned gridP = (unsigned)((P + 255) / 256);
    if (run_scan(ctx, 1, nullptr, ctx->leaf_flag, scan_total, P, n)) return 1;
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
    // (every tree's first position starts a leaf, so a leaf never crosses a tree boundary; with tree_begin the clamp of
    // k_leaf_lens to t * n boundaries is switched off by handing it one "tree" of P positions)
    hipLaunchKernelGGL(k_leaf_lens, dim3((unsigned)((nl + 255) / 256)), dim3(256), 0, ctx->stream, ctx->leaf_start,
                       (int64_t)nl, tree_begin_dev ? P : n, P, ctx->leaf_len, max_len_dev);
    hipLaunchKernelGGL(k_tree_leaf_begin, dim3((T + 63) / 64), dim3(64), 0, ctx->stream, ctx->scan_out, T, n, tree_begin_dev, P, nl, ctx->tree_begin_dev);
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
    ctx->forest_gen++;
    return 0;
}

int nnd_launch_forest(nnd_ctx *ctx) {
    const int64_t n = ctx->n, P = ctx->P;
    const int T = ctx->p.n_trees, leaf_size = ctx->p.leaf_size;
    ctx->forest_built = false;
    ctx->n_leaves = 0;
    ctx->max_leaf = leaf_size;
    ctx->tree_leaf_begin.clear();
    ctx->stats.n_cells = 0;
    if (T <= 0) return 0;
    if (P >= (int64_t)0x7FFFFFF0) {
        ctx->set_error("n_trees * n = %lld exceeds the int32 position space", (long long)P);
        return 1;
    }
    int levels = 0, rc = 2;
    if (ctx->s_m > 0) rc = forest_by_routing(ctx, &levels);
    if (nnd_knob("NND_FOREST_DEBUG"))
        fprintf(stderr, "forest: n=%lld T=%d s_m=%lld routing rc=%d levels=%d node_cap=%lld cell_cap=%lld max_segs=%lld\n", (long long)n, T,
                (long long)ctx->s_m, rc, levels, (long long)ctx->node_cap, (long long)ctx->cell_cap, (long long)ctx->max_segs);
    if (rc == 1) return 1;
    if (rc == 2) {  // small point set, very wide rows, or the recorded tree outgrew its tables: whole-set passes
        forest_view v{ctx->xp, ctx->xh, ctx->nr2, n, P, leaf_size, FIN_MAX, false};
        if (forest_levels(ctx, v)) return 1;
        ctx->cur = v.cur;
        levels = v.depth;
    }
    ctx->stats.tree_levels = levels;
    return forest_leaf_tables(ctx, P, T, nullptr);
}


// ---------------------------------------------------------------------------------------------------------------------
// The forest of the row-sharded build, sharded BY CELL (shard.hip drives the sequence; every rank runs it):
//   tops      rank r builds the top of ITS trees (split by tree) on the GLOBAL sample -- hashes keyed by the global tree
//             number and position (forest_view::tree_bias), so the forest does not depend on the number of ranks;
//   pack      the recorded nodes, compacted and rebased, go into this rank's slice of the table of ALL trees
//             (all-gathered by the caller), cells renumbered owner-major;
//   route     every rank routes ITS rows through ALL trees (coherent passes);
//   (the caller sends every (cell, row) pair to the rank that owns the cell)
//   finish    the owner places the rows of its cells, finishes them down to leaves and builds the leaf tables.
// Balanced whatever n_trees mod n_ranks is; no rank touches more than T * n / G point-trees after the tops.
int nnd_forest_sample_gather(nnd_ctx *ctx, int64_t j_lo, int64_t j_hi) {
    if (j_hi <= j_lo) return 0;
    hipLaunchKernelGGL(k_gather_sample, dim3((unsigned)((j_hi - j_lo + 15) / 16)), dim3(256), 0, ctx->stream, ctx->xp, ctx->xh, ctx->nr2,
                       ctx->dp, j_lo, j_hi, ctx->s_stride, ctx->tree_seed, ctx->xs, ctx->xsh, ctx->nr2s);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

__global__ void k_tree_cell_begin(const int32_t *__restrict__ leafscan, int T, int64_t M, int32_t *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) out[t] = leafscan[(int64_t)t * M];
}

int nnd_forest_tops(nnd_ctx *ctx, int T_loc, int tree_bias, nnd_tops_info *out) {
    out->n_packed = 0;
    out->n_cells = 0;
    out->levels = 0;
    out->n_low = out->high_lo = 0;
    if (T_loc <= 0) return 0;
    const int64_t M = ctx->s_m;
    forest_view v{ctx->xs, ctx->xsh, ctx->nr2s, M, (int64_t)T_loc * M, ctx->cell_leaf, FIN_SMALL, true};
    v.T = T_loc;
    v.tree_bias = tree_bias;
    int32_t n_cells = 0;
    const int rc = forest_tops(ctx, v, &n_cells);
    // rc 2: NOT an error of the build -- the single-GPU forest falls back to the whole-set passes here (nnd_launch_forest); the
    // sharded build tells the other ranks and all of them switch to the forest split by tree (shard.hip)
    if (rc == 2) { ctx->set_error("rp-forest (sharded): the recorded tree tops outgrew their tables"); return 2; }
    if (rc) return rc;
    // first cell of every local tree (cells are numbered tree-major): per-tree cell counts for the owner-major renumbering
    int32_t *tcb = ctx->route_roots + 2048;  // scratch words behind the root table
    hipLaunchKernelGGL(k_tree_cell_begin, dim3((T_loc + 63) / 64), dim3(64), 0, ctx->stream, ctx->scan_out, T_loc, M, tcb);
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_tree_begin, tcb, sizeof(int32_t) * T_loc, hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const int32_t *hb = (const int32_t *)ctx->h_tree_begin;
    for (int t = 0; t < T_loc; t++) out->tree_cells[t] = (t + 1 < T_loc ? hb[t + 1] : n_cells) - hb[t];
    out->n_cells = n_cells;
    out->n_low = v.n_low;
    out->high_lo = v.high_lo;
    out->n_packed = v.n_low + (ctx->node_cap - v.high_lo);
    out->levels = v.depth;
    return 0;
}

int nnd_forest_tops_pack(nnd_ctx *ctx, const nnd_tops_info *ti, int64_t node_base, const int32_t *cell_gid_dev, unsigned char *pack_dst,
                         float *hf_dst) {
    if (ti->n_packed <= 0) return 0;
    rp_pack_map mp{ti->n_low, ti->high_lo, (int)node_base, 0, cell_gid_dev};
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((ti->n_packed + 15) / 16)), dim3(256), 0, ctx->stream, ctx->node_hh, ctx->node_hf,
                       ctx->dp + 4, ctx->node_child, ctx->scan_out, ctx->dp, ti->n_packed, mp, pack_dst, hf_dst);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

int nnd_forest_route_rows(nnd_ctx *ctx, const unsigned char *pack_all, const float *hf_all, const int32_t *roots_dev, int T_all,
                          int64_t row_lo, int64_t nrows, int64_t n_cells_all, int32_t *cell_count_all) {
    if (nrows <= 0 || T_all <= 0) return 0;
    if ((int64_t)T_all * nrows > ctx->P) { ctx->set_error("rp-forest (sharded): %d trees x %lld rows exceed the position space of %lld", T_all, (long long)nrows, (long long)ctx->P); return 1; }
    rp_route_io io{pack_all, hf_all, roots_dev, T_all, row_lo, nrows, n_cells_all, cell_count_all, ctx->pos_seg[0], ctx->pos_seg[1], ctx->inv, ctx->scan_out, ctx->perm[1]};
    const int rc = route_coherent(ctx, io);
    if (rc == 2) { ctx->set_error("rp-forest (sharded): the routing passes do not support this row width"); return 1; }
    return rc;
}

// (cell, row) records of this rank's rows, ordered by cell: rec_pos = cell_scan[cell] + the row's slot in the cell
__global__ void k_route_records(const int32_t *__restrict__ cell_of, const int32_t *__restrict__ rank_of, const int32_t *__restrict__ cell_scan,
                                int64_t nrows, int64_t row_lo, int32_t *__restrict__ rec_cell, int32_t *__restrict__ rec_row) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrows) return;
    const int64_t slot = (int64_t)blockIdx.y * nrows + r;
    const int c = cell_of[slot];
    const int at = cell_scan[c] + rank_of[slot];
    rec_cell[at] = c;
    rec_row[at] = (int32_t)(row_lo + r);
}
// records sorted by cell + the record offset of every destination rank's first cell (dest_cell[q], q = 0 .. G)
__global__ void k_dest_offsets(const int32_t *__restrict__ cell_scan, const int32_t *__restrict__ dest_cell, int G, int32_t n_cells_all,
                               const int32_t *__restrict__ total, long long *__restrict__ out) {
    const int q = threadIdx.x;
    if (q > G) return;
    const int c = dest_cell[q];
    const long long at = c < n_cells_all ? cell_scan[c] : total[0];
    out[q] = at;
}
int nnd_forest_route_records(nnd_ctx *ctx, int T_all, int64_t row_lo, int64_t nrows, int32_t n_cells_all, int32_t *cell_count_all /* in: counts, out: exclusive scan */,
                             int32_t *count_copy /* out: the counts (sent to the cells' owners) */,
                             const int32_t *dest_cell_dev, int G, int32_t *rec_cell, int32_t *rec_row, long long *dest_off_dev) {
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    NND_HIP_CHECK(hipMemcpyAsync(count_copy, cell_count_all, sizeof(int32_t) * (size_t)n_cells_all, hipMemcpyDeviceToDevice, ctx->stream));
    if (scan_i32_inplace(ctx, cell_count_all, n_cells_all, scan_total)) return 1;
    if (nrows > 0)
        hipLaunchKernelGGL(k_route_records, dim3((unsigned)((nrows + 255) / 256), (unsigned)T_all), dim3(256), 0, ctx->stream, ctx->pos_seg[0],
                           ctx->pos_seg[1], cell_count_all, nrows, row_lo, rec_cell, rec_row);
    hipLaunchKernelGGL(k_dest_offsets, dim3(1), dim3(128), 0, ctx->stream, cell_count_all, dest_cell_dev, G, n_cells_all, scan_total, dest_off_dev);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// owner side: the received records -> per-cell counts (pass 1) and, once the cells have their positions, the rows (pass 2)
// cell_count[c] = sum over the G source ranks of their count of cell c (count vectors exchanged by the caller: no counting
// pass over the records)
__global__ void k_owner_sum_counts(const int32_t *__restrict__ cnt_src /* (G, n_cells) */, int G, int32_t n_cells, int32_t *__restrict__ cell_count) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cells) return;
    int s = 0;
    for (int q = 0; q < G; q++) s += cnt_src[(size_t)q * n_cells + c];
    cell_count[c] = s;
}
__global__ void k_owner_place(const int32_t *__restrict__ rec_cell, const int32_t *__restrict__ rec_row, int64_t n_rec, int32_t cell_base,
                              const int32_t *__restrict__ cell_start, int32_t *__restrict__ cursor, int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_rec) return;
    const int c = rec_cell[i] - cell_base;
    perm[cell_start[c] + atomicAdd(&cursor[c], 1)] = rec_row[i];
}
__global__ void k_gather_i32(const int32_t *__restrict__ src, const int32_t *__restrict__ map, int n, int32_t *__restrict__ dst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[map[i]];
}
__global__ void k_tree_pos_begin(const int32_t *__restrict__ cell_start, const int32_t *__restrict__ tree_first_cell, int T, int32_t n_cells,
                                 int32_t P_used, int32_t *__restrict__ out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t > T) return;
    const int c = t < T ? tree_first_cell[t] : n_cells;
    out[t] = c < n_cells ? cell_start[c] : P_used;
}

// n_rec records (cell, row) of the cells [cell_base, cell_base + n_cells_own) this rank owns (any order); cell_depth_all:
// the depth of every cell of the build in the numbering depth_map[own cell] points into; tree_first_cell (device, T_all + 1):
// this rank's first own cell of every tree (own numbering).
int nnd_forest_finish_owned(nnd_ctx *ctx, const int32_t *rec_cell, const int32_t *rec_row, int64_t n_rec, int32_t cell_base, int32_t n_cells_own,
                            const int32_t *cnt_src, int G, const int32_t *cell_depth_all, const int32_t *depth_map_dev,
                            const int32_t *tree_first_cell_dev, int T_all) {
    ctx->forest_built = false;
    ctx->n_leaves = 0;
    ctx->max_leaf = ctx->p.leaf_size;
    ctx->tree_leaf_begin.assign((size_t)T_all + 1, 0);
    if (n_rec > ctx->P || n_cells_own > ctx->cell_cap || n_cells_own + n_rec / (ctx->p.leaf_size + 1) > ctx->max_segs) {
        ctx->set_error("rp-forest (sharded): %lld point-trees / %d cells exceed this rank's forest tables (%lld positions)", (long long)n_rec, n_cells_own,
                       (long long)ctx->P);
        return 2;  // (as above: the ranks agree to build this forest split by tree)
    }
    if (n_cells_own <= 0 || n_rec <= 0) {  // (a rank that owns no cell: nothing to seed from)
        ctx->forest_built = true;
        return 0;
    }
    hipLaunchKernelGGL(k_owner_sum_counts, dim3((unsigned)((n_cells_own + 255) / 256)), dim3(256), 0, ctx->stream, cnt_src, G, n_cells_own, ctx->cell_count);
    hipLaunchKernelGGL(k_gather_i32, dim3((unsigned)((n_cells_own + 255) / 256)), dim3(256), 0, ctx->stream, cell_depth_all, depth_map_dev, n_cells_own,
                       ctx->cell_depth);
    // positions: cell_start = exclusive scan of the counts; rows placed through per-cell cursors (the order inside a cell is
    // immaterial: the finisher is order independent)
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    NND_HIP_CHECK(hipMemcpyAsync(ctx->cell_start, ctx->cell_count, sizeof(int32_t) * (size_t)n_cells_own, hipMemcpyDeviceToDevice, ctx->stream));
    if (scan_i32_inplace(ctx, ctx->cell_start, n_cells_own, scan_total)) return 1;
    int32_t *cursor = ctx->small_list;  // (free until k_cell_lists: 3 * cell_cap words)
    NND_HIP_CHECK(hipMemsetAsync(cursor, 0, sizeof(int32_t) * (size_t)n_cells_own, ctx->stream));
    hipLaunchKernelGGL(k_owner_place, dim3((unsigned)((n_rec + 255) / 256)), dim3(256), 0, ctx->stream, rec_cell, rec_row, n_rec, cell_base, ctx->cell_start,
                       cursor, ctx->perm[0]);
    int32_t *tpb = ctx->route_roots + 2048 + 64;  // (T_all + 1) position of every tree's first own cell
    hipLaunchKernelGGL(k_tree_pos_begin, dim3((T_all + 64) / 64), dim3(64), 0, ctx->stream, ctx->cell_start, tree_first_cell_dev, T_all, n_cells_own,
                       (int32_t)n_rec, tpb);
    NND_HIP_CHECK(hipGetLastError());
    rp_tree_map tm;
    tm.tree_begin = tpb;
    tm.n_tree_begin = T_all;
    // (forest_place_finish scans the counts again: cheap, and it keeps one code path for the work lists)
    const int rc = forest_place_finish(ctx, n_cells_own, T_all, 0, 0, nullptr, nullptr, n_rec, tm);
    if (rc == 2) { ctx->set_error("rp-forest (sharded): too many over-long cells"); return 2; }
    if (rc) return rc;
    ctx->stats.n_cells = n_cells_own;
    return forest_leaf_tables(ctx, n_rec, T_all, tpb);
}

// One stable partition step for another builder (hubtree.hip): positions with pos >= 0 and side == 0 move to the front of
// their segment, the others behind them; pos_out = seg_child[2 * segment + side] (or -1 for positions already final).
void nnd_forest_stable_partition(nnd_ctx *ctx, int64_t n, const int32_t *ord, const int32_t *pos, uint8_t *side, const int32_t *seg_start,
                                 const int32_t *seg_len, int n_segs, int32_t *nleft, const int32_t *seg_child, int32_t *ord_out,
                                 int32_t *pos_out) {
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    const int nb = (int)((n + SCAN_TILE - 1) / SCAN_TILE);
    hipLaunchKernelGGL(k_scan_reduce, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, 0, pos, side, n, ctx->scan_blk, (const int32_t *)nullptr,
                       (const uint8_t *)nullptr, n);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, ctx->scan_blk, nb, scan_total);
    hipLaunchKernelGGL(k_scan_apply<false>, dim3(nb), dim3(SCAN_BLOCK), 0, ctx->stream, 0, pos, side, n, ctx->scan_blk, ctx->scan_out,
                       (int32_t *)nullptr);
    hipLaunchKernelGGL(k_seg_count, dim3((unsigned)((n_segs + 255) / 256)), dim3(256), 0, ctx->stream, seg_start, seg_len, n_segs,
                       ctx->scan_out, scan_total, n, nleft);
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, ord, pos, side, ctx->scan_out, seg_start,
                       nleft, seg_child, n, n, ord_out, pos_out, (int32_t *)nullptr);
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
