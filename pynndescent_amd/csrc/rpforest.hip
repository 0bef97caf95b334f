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
    float acc = 0.0f, sq = 0.0f, res = 0.0f;
    for (int j = lane; j < dp; j += 64) {
        float l = xl[j], r = xr[j];
        float v = l - r;
        h[j] = v;
        if (!angular) {
            const uint16_t b = nnd_f32_to_bf16(v);
            hb[j] = b;
            const float e = v - __uint_as_float((uint32_t)b << 16);
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
            const uint16_t b = nnd_f32_to_bf16(v);
            hb[j] = b;
            const float e = v - __uint_as_float((uint32_t)b << 16);
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
__device__ __forceinline__ float rp_band(float xnorm, float rx, float hnorm, float rh) {
    return rx * hnorm + (xnorm + rx) * rh + RP_ACC * hnorm * xnorm + RP_EPS;  // + RP_EPS: outside the band the exact margin is no coin flip either
}
// non-negative f32 -> bf16 bits, rounded UP (packed bounds stay bounds)
__device__ __forceinline__ uint32_t rp_bf16_up(float v) { return (__float_as_uint(v) + 0xFFFFu) >> 16; }
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
#ifdef NND_RP_NORECHECK  // timing experiments only
    if (band < 0.0f) m = rp_exact_quad(xf_row, h, dp, sub) + off;
#else
    if (!(fabsf(m) > band)) m = rp_exact_quad(xf_row, h, dp, sub) + off;  // uniform inside the quad
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
                                                uint8_t *__restrict__ side) {
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
    const float m = rp_quad_sum(acc) + off;
    const float band = rp_band(metric == 0 ? sqrtf(xn) : xn, rx, hnorm, rh);
    const uint8_t sd = rp_side(m, band, xp + pt * dp, h, off, dp, sub, seed, (uint32_t)g, depth);
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
                                                      uint8_t *__restrict__ side_pt) {
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
            const float m = rp_quad_sum(acc[u]) + off;
            const float band = rp_band(xnorm, rx, hnorm, rh);
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
                                                  int32_t *__restrict__ leaf_depth, int node_top) {
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
                                                         rp_record rec = rp_record{}) {
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
    const uint32_t seedt = seed ^ ((uint32_t)((int64_t)a / n) * 0x9E3779B9u);  // per tree
    if (!BIG)
        for (int i = tid; i < len; i += NTHR) ids[i] = perm[a + i];
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
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const uint64_t o1 = rp_shfl_xor_u64(k1, o), o2 = rp_shfl_xor_u64(k2, o);
            rp_top2_push(k1, k2, o1);
            rp_top2_push(k1, k2, o2);
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
        part = nnd_wave_sum_f32(part);
        psq = nnd_wave_sum_f32(psq);
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
            const uint16_t b = nnd_f32_to_bf16(v);
            hb[j] = b;
            const float e = v - __uint_as_float((uint32_t)b << 16);
            pres += e * e;
        }
        pres = nnd_wave_sum_f32(pres);
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
#pragma unroll
            for (int u = 0; u < 2; u++) {
                const float2 nrv = nr[pt[u]];
                xn[u] = nrv.x;
                rxv[u] = nrv.y;
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
                const float m = rp_quad_sum(acc[u]) + off;
                const float band = rp_band(metric == 0 ? sqrtf(xn[u]) : xn[u], rxv[u], hnorm, rhv);
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
            incl = cntl;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
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
                                int dp, int64_t m, int64_t stride, uint32_t seed, float *__restrict__ xs,
                                uint16_t *__restrict__ xsh, float2 *__restrict__ nrs) {
    const int sub = threadIdx.x & 15;
    const int64_t j = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    if (j >= m) return;
    const int64_t i = j * stride + (int64_t)(nnd_hash2(seed ^ 0x7F4A7C15u, (uint32_t)j) % (uint32_t)stride);
    for (int c = sub; c < (dp >> 2); c += 16) ((float4 *)(xs + j * dp))[c] = ((const float4 *)(xp + i * dp))[c];
    for (int c = sub; c < (dp >> 3); c += 16) ((uint4 *)(xsh + j * dp))[c] = ((const uint4 *)(xh + i * dp))[c];
    if (sub == 0) nrs[j] = nr[i];
}

__device__ __forceinline__ uint32_t rp_pack_bf16(float a, float b) {
    return (uint32_t)nnd_f32_to_bf16(a) | ((uint32_t)nnd_f32_to_bf16(b) << 16);
}
__device__ __forceinline__ float rp_dot4f(float4 a, float4 b, float acc) {
    return acc + a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
}

// Recorded nodes are packed for the walk: [dp bf16 hyperplane | f32 offset | bf16 |h| : bf16 |h - bf16(h)| (both rounded
// up) | child 0 | child 1], one record of 2 * dp + 16 bytes per node (a walk step touches ONE contiguous record instead
// of three tables).
__global__ void k_pack_nodes(const uint16_t *__restrict__ node_hh, const float *__restrict__ node_hf, int hs,
                             const int32_t *__restrict__ node_child, int dp, int64_t n_nodes, unsigned char *__restrict__ pack) {
    const int sub = threadIdx.x & 15;
    const int64_t v = (int64_t)blockIdx.x * (blockDim.x >> 4) + (threadIdx.x >> 4);
    if (v >= n_nodes) return;
    const int rec = 2 * dp + 16;
    uint4 *dst = (uint4 *)(pack + v * rec);
    const uint4 *src = (const uint4 *)(node_hh + v * dp);
    for (int c = sub; c < (dp >> 3); c += 16) dst[c] = src[c];
    if (sub == 0) {
        const float *h = node_hf + v * hs;
        dst[dp >> 3] = make_uint4(__float_as_uint(h[dp]), (rp_bf16_up(h[dp + 1]) << 16) | rp_bf16_up(h[dp + 2]),
                                  (uint32_t)node_child[2 * v], (uint32_t)node_child[2 * v + 1]);
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
                                               const float *__restrict__ node_hf, int hs,
                                               const int32_t *__restrict__ leafscan, uint32_t seed,
                                               int32_t *__restrict__ cell_count, int32_t *__restrict__ cell_of,
                                               int32_t *__restrict__ rank_of, int n_top, int l_top) {
    extern __shared__ __attribute__((aligned(16))) unsigned char top_tab[];
    const int rec = 2 * dp + 16;
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
                xq[q] = make_uint4(rp_pack_bf16(xa[q].x, xa[q].y), rp_pack_bf16(xa[q].z, xa[q].w), rp_pack_bf16(xb[q].x, xb[q].y),
                                   rp_pack_bf16(xb[q].z, xb[q].w));
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
                    float m = rp_quad_sum(acc[u]) + off;
                    const float band = rp_band(xnorm, rx, hnorm, rh);
                    if (!(fabsf(m) > band)) {  // inside the bf16 error band: the exact f32 margin decides (quad-uniform)
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
                    if (nxt <= -2) {  // reached a cell: first sample position -2 - nxt -> cell index
                        if (sub == 0) {
                            const int cell = leafscan[-2 - nxt];
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

// One tree per XCD.  k_route above walks every tree from every workgroup, so each XCD's L2 (4 MB) has to hold the node
// records of ALL trees (10 MB at 1 M points, 8 trees): the deep levels miss and come over the fabric -- 10 GB of
// fetches per build, 20 x the rows themselves.  Here the workgroups of XCD x (workgroups are dealt round-robin, so
// that is blockIdx.x & 7) walk only trees x, x + 8, ...: one tree's records (1.3 MB) stay in that XCD's L2, and the
// first l_top levels of the tree -- ALL its nodes at those depths, whoever recorded them -- sit in LDS in heap order
// (slot 1 = root, children of slot s at 2s and 2s + 1), so a walk reads LDS while depth < l_top and L2 after that.
// The price: a point's row is fetched once per tree instead of once -- but only its bf16 copy (the screening operand,
// 2 * dp bytes); the f32 row is touched only when a margin falls inside the bf16 error band.  PB points per quad walk
// in lock step (independent chains to cover the L2 latency).  Same arithmetic, same coins, same cells as k_route.
#ifndef NND_RX_OCC
#define NND_RX_OCC 4  // waves per SIMD the register budget is sized for (two 512-thread workgroups per CU)
#endif
#ifndef NND_RX_LDS_KB
#define NND_RX_LDS_KB 72
#endif
template <int NC, int PB>
__global__ __launch_bounds__(512, NND_RX_OCC) void k_route_xcd(const float *__restrict__ xp, const uint16_t *__restrict__ xh,
                                                   const float2 *__restrict__ nr, int metric, int dp, int64_t n, int n_trees,
                                                   const unsigned char *__restrict__ node_pack,
                                                   const float *__restrict__ node_hf, int hs,
                                                   const int32_t *__restrict__ leafscan, uint32_t seed,
                                                   int32_t *__restrict__ cell_count, int32_t *__restrict__ cell_of,
                                                   int32_t *__restrict__ rank_of, int l_top) {
    extern __shared__ __attribute__((aligned(16))) unsigned char top_tab[];
    const int rec = 2 * dp + 16, r16 = rec >> 4, nslots = 1 << l_top;
    int32_t *top_node = (int32_t *)(top_tab + (size_t)nslots * rec);  // node id of every heap slot (-1: no node there)
    const int xcd = blockIdx.x & 7, bx = blockIdx.x >> 3, nbx = gridDim.x >> 3;
    const int sub = threadIdx.x & 3, qpb = blockDim.x >> 2;
    for (int t = xcd; t < n_trees; t += 8) {
        __syncthreads();  // the previous tree's table is no longer read
        if (threadIdx.x == 0) {
            top_node[0] = -1;
            top_node[1] = t;  // the root of tree t is node t
        }
        __syncthreads();
        for (int l = 0; l + 1 < l_top; l++) {  // children of the slots of level l
            for (int sl = (1 << l) + threadIdx.x; sl < (2 << l); sl += blockDim.x) {
                const int nd = top_node[sl];
                int c0 = -1, c1 = -1;
                if (nd >= 0) {
                    const uint4 meta = *(const uint4 *)(node_pack + (int64_t)nd * rec + 2 * dp);
                    c0 = (int)meta.z;
                    c1 = (int)meta.w;
                }
                top_node[2 * sl] = c0 >= 0 ? c0 : -1;  // cells (<= -2) have no record
                top_node[2 * sl + 1] = c1 >= 0 ? c1 : -1;
            }
            __syncthreads();
        }
        for (int q = threadIdx.x; q < nslots * r16; q += blockDim.x) {
            const int sl = q / r16, wd = q - sl * r16;
            const int nd = top_node[sl];
            if (nd >= 0) ((uint4 *)top_tab)[q] = ((const uint4 *)(node_pack + (int64_t)nd * rec))[wd];
        }
        __syncthreads();
        for (int64_t i0 = (int64_t)bx * qpb * PB; i0 < n; i0 += (int64_t)nbx * qpb * PB) {
            int64_t pi[PB];
            uint4 xq[PB][NC];
            float xnorm[PB], rxv[PB];
            int node[PB], hidx[PB];
#pragma unroll
            for (int u = 0; u < PB; u++) {
                pi[u] = i0 + (int64_t)u * qpb + (threadIdx.x >> 2);
                const bool on = pi[u] < n;
                const int64_t ic = on ? pi[u] : 0;
                const uint4 *row = (const uint4 *)(xh + ic * dp);
#pragma unroll
                for (int q = 0; q < NC; q++) xq[u][q] = row[sub + 4 * q];
                const float2 nrv = nr[ic];
                const float xn = nrv.x;
                rxv[u] = nrv.y;
                xnorm[u] = metric == 0 ? sqrtf(xn) : xn;
                node[u] = on ? t : -1;  // >= 0: current node; -1: this walk is over (whole quad)
                hidx[u] = 1;
            }
            for (int depth = 0;; depth++) {
                bool any = false;
#pragma unroll
                for (int u = 0; u < PB; u++) any |= node[u] >= 0;
                if (!__ballot(any)) break;  // wave-uniform
                uint4 p[PB][NC], meta[PB];
                if (depth < l_top) {  // every live walk is at depth `depth`: heap slot hidx < 2^l_top
#pragma unroll
                    for (int u = 0; u < PB; u++) {
                        const uint4 *r8 = (const uint4 *)(top_tab + (size_t)(node[u] >= 0 ? hidx[u] : 1) * rec);
#pragma unroll
                        for (int q = 0; q < NC; q++) p[u][q] = r8[sub + 4 * q];
                        meta[u] = r8[dp >> 3];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < PB; u++) {
                        const uint4 *r8 = (const uint4 *)(node_pack + (int64_t)(node[u] >= 0 ? node[u] : t) * rec);
#pragma unroll
                        for (int q = 0; q < NC; q++) p[u][q] = r8[sub + 4 * q];
                        meta[u] = r8[dp >> 3];
                    }
                }
                float acc[PB];
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    acc[u] = 0.0f;
#pragma unroll
                    for (int q = 0; q < NC; q++) acc[u] = rp_dot8(xq[u][q], p[u][q], acc[u]);
                }
#pragma unroll
                for (int u = 0; u < PB; u++) {
                    if (node[u] < 0) continue;  // whole quad
                    const float off = __uint_as_float(meta[u].x), hnorm = __uint_as_float(meta[u].y & 0xFFFF0000u),
                                rh = __uint_as_float(meta[u].y << 16);
                    float m = rp_quad_sum(acc[u]) + off;
                    const float band = rp_band(xnorm[u], rxv[u], hnorm, rh);
#ifdef NND_RX_NORECHECK  // timing experiments only
                    if (band < 0.0f) {
#else
                    if (!(fabsf(m) > band)) {  // inside the bf16 error band: the exact f32 margin decides (quad-uniform)
#endif
                        const float4 *h4 = (const float4 *)(node_hf + (int64_t)node[u] * hs);
                        const float4 *x4 = (const float4 *)(xp + pi[u] * dp);
                        float e = 0.0f;
#pragma unroll
                        for (int q = 0; q < NC; q++) {
                            const int c = sub + 4 * q;
                            e = rp_dot4f(x4[2 * c], h4[2 * c], e);
                            e = rp_dot4f(x4[2 * c + 1], h4[2 * c + 1], e);
                        }
                        m = rp_quad_sum(e) + off;
                    }
                    const int64_t slot = (int64_t)t * n + pi[u];
                    int side;
                    if (fabsf(m) < RP_EPS) side = (int)(nnd_hash3(seed ^ 0x5bd1e995u, (uint32_t)slot, (uint32_t)depth) & 1u);  // rp_trees.py:380-385
                    else side = m > 0.0f ? 0 : 1;                                                                             // rp_trees.py:386-391
                    const int nxt = (int)(side ? meta[u].w : meta[u].z);
                    if (nxt <= -2) {  // reached a cell: first sample position -2 - nxt -> cell index
                        if (sub == 0) {
                            const int cell = leafscan[-2 - nxt];
                            cell_of[slot] = cell;
#ifdef NND_RX_NOATOMIC  // timing experiments only
                            rank_of[slot] = 0;
#else
                            rank_of[slot] = atomicAdd(&cell_count[cell], 1);
#endif
                        }
                        node[u] = -1;
                    } else {
                        node[u] = nxt;
                        hidx[u] = 2 * hidx[u] + side;
                    }
                }
            }
        }
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
                        const int32_t *__restrict__ cell_start, int64_t n, int32_t *__restrict__ perm) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t slot = (int64_t)blockIdx.y * n + i;
    perm[cell_start[cell_of[slot]] + rank_of[slot]] = (int32_t)i;
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
};

// big list (global-memory variant; its nodes join the workgroup list as they shrink) -> workgroup list -> small list
static int launch_finishers(nnd_ctx *ctx, int32_t *perm, int32_t *other, const int32_t *big_start, const int32_t *big_len,
                            const int32_t *big_depth, int depth0, long long n_big, long long n_small = 0) {
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
                           (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, fin_count);
        NND_HIP_CHECK(hipGetLastError());
    }
    if (n_big > 0) {
        hipLaunchKernelGGL((k_finish_subtrees<true, 256, 0>), dim3((unsigned)n_big), dim3(256), fin_smem_bytes(dp, 0), ctx->stream,
                           FIN_ARGS(big_start, big_len, big_depth, depth0, n_big), other, ctx->side, FIN_MAX, fin_start, fin_len,
                           fin_depth, fin_count);
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
                           FIN_MAX, fin_start, fin_len, fin_depth, fin_count);
        NND_HIP_CHECK(hipGetLastError());
    }
#undef FIN_ARGS
    return 0;
}

// The level-synchronous passes on `v`.  Returns 0, 1 (error) or 2 (recording ran out of node slots: caller falls back).
static int forest_levels(nnd_ctx *ctx, forest_view &v) {
    const int64_t n = v.n, P = v.P;
    const int T = ctx->p.n_trees, dp = ctx->dp, leaf_size = v.leaf_size, max_depth = ctx->p.max_depth;
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
                           hyper, hs, hyper_h);
        // point-major pass: hyperplane table fits in L2 AND enough positions are still active to amortise
        // streaming every row once (it costs n rows regardless of how many positions are active)
        const bool fused = inv_live && (S * (int64_t)dp * 2 <= (int64_t)6 << 20) && (active_pos * 2 >= 3 * n);
        if (fused) {
            hipLaunchKernelGGL(k_margin_fused, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, ctx->stream, v.xp, v.xh, v.nr,
                               ctx->p.metric, dp, n, T, ctx->inv, hyper, hs, hyper_h, ctx->tree_seed, depth, ctx->side_pt);
        } else {
            inv_live = false;
            hipLaunchKernelGGL(k_margin, dim3((unsigned)((P + 63) / 64)), dim3(256), 0, ctx->stream, v.xp, v.xh, v.nr,
                               ctx->p.metric, dp, ctx->perm[cur], ctx->pos_seg[cur], P, hyper, hs, hyper_h, ctx->tree_seed, depth,
                               ctx->side);
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
                           (int)ctx->node_cap - 1);
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
        node_base += S;
        S = next[0];
        active_pos = next[1];
        cur = 1 - cur;
        depth++;
        // Tail of the level loop (whole-set mode): once most positions have been handed over, a level-synchronous pass
        // still costs P positions per kernel for a few hundred segments: the global-memory variant of the finisher
        // takes them (its nodes are split in place until they fit the LDS finisher, whose work list they join).
        if (!v.record && S > 0 && (active_pos * 2 < 3 * n) && next[5] <= BIG_MAX) {  // next[5] = CNT_SCRATCH + 3: longest stayer
            if (launch_finishers(ctx, ctx->perm[cur], ctx->perm[1 - cur], ctx->seg_start[cur], ctx->seg_len[cur], nullptr, depth, S)) return 1;
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
                               (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, ctx->counters + CNT_SCRATCH + 1, rs);
            if (v.fin_max > FIN_SMALL)
                hipLaunchKernelGGL((k_finish_subtrees<false, 256, FIN_MAX, true>), dim3((unsigned)nfin), dim3(256), fin_smem_bytes(dp, FIN_MAX),
                                   ctx->stream, v.xp, v.xh, v.nr, ctx->p.metric, dp, n, ctx->perm[cur], fin_start, fin_len, fin_depth, 0,
                                   (int)nfin, angular, ctx->tree_seed, max_depth, leaf_size, ctx->leaf_flag, (int32_t *)nullptr,
                                   (uint8_t *)nullptr, FIN_MAX, fin_start, fin_len, fin_depth, ctx->counters + CNT_SCRATCH + 1, rl);
            NND_HIP_CHECK(hipGetLastError());
            NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 38, flags, 2 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
            NND_HIP_CHECK(nnd_sync_spin(ctx));
            if (((const int *)(ctx->h_pin + 38))[1]) {  // node tables exhausted
                if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
                return 2;
            }
        }
        v.n_nodes = ctx->node_cap;  // ids are spread over the table: level-synchronous nodes upwards, recorded subtrees downwards
        return 0;
    }
    if (launch_finishers(ctx, ctx->perm[cur], ctx->perm[1 - cur], nullptr, nullptr, nullptr, 0, 0)) return 1;
    return 0;
}

template <int NC, int TB>
static int launch_route(nnd_ctx *ctx, const int32_t *leafscan, int n_top, int l_top) {
    auto kern = k_route<NC, TB>;
    const size_t smem = (size_t)n_top * (2 * ctx->dp + 16);
    static int n_cu_dev[64] = {0};
    int &n_cu = n_cu_dev[ctx->p.device & 63];
    if (n_cu == 0) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        hipDeviceProp_t prop;
        NND_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->p.device));
        n_cu = prop.multiProcessorCount;
    }
    int64_t blocks = (ctx->n + 127) / 128;
    if (blocks > 2 * (int64_t)n_cu) blocks = 2 * (int64_t)n_cu;  // persistent: two 512-thread workgroups per CU
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), smem, ctx->stream, ctx->xp, ctx->nr2, ctx->p.metric, ctx->dp, ctx->n,
                       ctx->p.n_trees, ctx->node_pack, ctx->node_hf, ctx->dp + 4, leafscan, ctx->tree_seed, ctx->cell_count,
                       ctx->pos_seg[0], ctx->pos_seg[1], n_top, l_top);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int NC, int PB>
static int launch_route_xcd(nnd_ctx *ctx, const int32_t *leafscan) {
    auto kern = k_route_xcd<NC, PB>;
    const int rec = 2 * ctx->dp + 16;
    int l_top = 1;
    while (((size_t)2 << l_top) * (rec + 4) <= NND_RX_LDS_KB * 1024) l_top++;  // 2^l_top heap slots (+ their node ids)
    const size_t smem = ((size_t)1 << l_top) * (rec + 4);
    static int n_cu_dev[64] = {0};
    int &n_cu = n_cu_dev[ctx->p.device & 63];
    if (n_cu == 0) {
        NND_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024));
        hipDeviceProp_t prop;
        NND_HIP_CHECK(hipGetDeviceProperties(&prop, ctx->p.device));
        n_cu = prop.multiProcessorCount;
    }
    int64_t blocks = (NND_RX_OCC / 2) * (int64_t)n_cu;  // persistent: NND_RX_OCC / 2 512-thread workgroups per CU
    blocks = blocks < 8 ? 8 : (blocks & ~(int64_t)7);  // the same number of workgroups on every XCD
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(512), smem, ctx->stream, ctx->xp, ctx->xh, ctx->nr2, ctx->p.metric, ctx->dp,
                       ctx->n, ctx->p.n_trees, ctx->node_pack, ctx->node_hf, ctx->dp + 4, leafscan, ctx->tree_seed, ctx->cell_count,
                       ctx->pos_seg[0], ctx->pos_seg[1], l_top);
    NND_HIP_CHECK(hipGetLastError());
    return 0;
}

// sample forest -> routing pass -> cells -> finishers.  Returns 0, 1, or 2 (fall back to the whole-set passes).
static int forest_by_routing(nnd_ctx *ctx, int *levels_out) {
    const int64_t n = ctx->n, P = ctx->P, M = ctx->s_m, Ps = (int64_t)ctx->p.n_trees * M;
    const int T = ctx->p.n_trees, dp = ctx->dp;
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);
    hipLaunchKernelGGL(k_gather_sample, dim3((unsigned)((M + 15) / 16)), dim3(256), 0, ctx->stream, ctx->xp, ctx->xh, ctx->nr2,
                       dp, M, ctx->s_stride, ctx->tree_seed, ctx->xs, ctx->xsh, ctx->nr2s);
    // sample subtrees of <= 512 members leave the level-synchronous passes for the (one-wave) recording finisher;
    // 2048 (+ a workgroup class) means 4 fewer levels but a slower finisher: 6.9-7.2 ms vs 6.6 ms per forest at 1 M points
    static const int rec_fin = [] { const char *e = nnd_knob("NND_REC_FIN"); const int r = e ? atoi(e) : FIN_SMALL; return r <= FIN_SMALL ? FIN_SMALL : FIN_MAX; }();
    forest_view v{ctx->xs, ctx->xsh, ctx->nr2s, M, Ps, ctx->cell_leaf, rec_fin, true};
    int rc = forest_levels(ctx, v);
    if (rc) return rc;
    // cells = leaves of the recorded trees, numbered in position order (tree-major)
    if (run_scan(ctx, 1, nullptr, ctx->leaf_flag, scan_total, Ps, M)) return 1;
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 35, scan_total, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const int32_t n_cells = *(const int32_t *)(ctx->h_pin + 35);
    if (n_cells > ctx->cell_cap || n_cells + P / (ctx->p.leaf_size + 1) > ctx->max_segs) {
        if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
        return 2;
    }
    hipLaunchKernelGGL(k_cell_depths, dim3((unsigned)((Ps + 255) / 256)), dim3(256), 0, ctx->stream, ctx->leaf_flag, ctx->scan_out,
                       ctx->s_leaf_depth, Ps, ctx->cell_depth);
    NND_HIP_CHECK(hipMemsetAsync(ctx->cell_count, 0, sizeof(int32_t) * (size_t)n_cells, ctx->stream));
    hipLaunchKernelGGL(k_pack_nodes, dim3((unsigned)((v.n_nodes + 15) / 16)), dim3(256), 0, ctx->stream, ctx->node_hh, ctx->node_hf,
                       dp + 4, ctx->node_child, dp, v.n_nodes, ctx->node_pack);
    // levels whose records fit the route kernel's LDS copy (<= 72 KB: two workgroups per CU)
    int l_top = 0;
    while (l_top + 1 < (int)v.level_base.size() && v.level_base[l_top + 1] * (2 * dp + 16) <= 72 * 1024) l_top++;
    const int n_top = (int)v.level_base[l_top];
    int rrc = 2;
    // one tree per XCD (k_route_xcd) measured no faster than all trees per point (2.8-3.2 ms vs 2.8 ms at 1 M points):
    // the walk is bound by its dependent record fetches and rechecks, not by where the records are cached.  Opt-in.
    static const bool route_xcd = [] { const char *e = nnd_knob("NND_ROUTE_XCD"); return e && atoi(e) != 0; }();
    if (route_xcd && T % 8 == 0 && dp % 32 == 0 && dp <= 256) {  // every XCD gets the same number of trees
        switch (dp / 32) {
            case 1: rrc = launch_route_xcd<1, 4>(ctx, ctx->scan_out); break;
            case 2: rrc = launch_route_xcd<2, 2>(ctx, ctx->scan_out); break;
            case 3: rrc = launch_route_xcd<3, 2>(ctx, ctx->scan_out); break;
#ifdef NND_RX_PB
            case 4: rrc = launch_route_xcd<4, NND_RX_PB>(ctx, ctx->scan_out); break;
#else
            case 4: rrc = launch_route_xcd<4, 2>(ctx, ctx->scan_out); break;
#endif
            case 5: rrc = launch_route_xcd<5, 1>(ctx, ctx->scan_out); break;
            case 6: rrc = launch_route_xcd<6, 1>(ctx, ctx->scan_out); break;
            case 7: rrc = launch_route_xcd<7, 1>(ctx, ctx->scan_out); break;
            case 8: rrc = launch_route_xcd<8, 1>(ctx, ctx->scan_out); break;
            default: break;
        }
    } else
    switch (dp / 32) {  // NC = 8-float chunks per lane; trees per batch sized for <= 128 VGPRs
        case 1: rrc = launch_route<1, 4>(ctx, ctx->scan_out, n_top, l_top); break;
        case 2: rrc = launch_route<2, 4>(ctx, ctx->scan_out, n_top, l_top); break;
        case 3: rrc = launch_route<3, 2>(ctx, ctx->scan_out, n_top, l_top); break;
        case 4: rrc = launch_route<4, 2>(ctx, ctx->scan_out, n_top, l_top); break;
        case 5: rrc = launch_route<5, 1>(ctx, ctx->scan_out, n_top, l_top); break;
        case 6: rrc = launch_route<6, 1>(ctx, ctx->scan_out, n_top, l_top); break;
        case 7: rrc = launch_route<7, 1>(ctx, ctx->scan_out, n_top, l_top); break;
        case 8: rrc = launch_route<8, 1>(ctx, ctx->scan_out, n_top, l_top); break;
        default: break;  // wider rows: whole-set passes (nnd_create does not enable routing for them)
    }
    if (rrc) return rrc;
    // cell_start = exclusive scan of the counts (k_scan_blocks scans in place: copy first)
    NND_HIP_CHECK(hipMemcpyAsync(ctx->cell_start, ctx->cell_count, sizeof(int32_t) * (size_t)n_cells, hipMemcpyDeviceToDevice, ctx->stream));
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(256), 0, ctx->stream, ctx->cell_start, (int)n_cells, scan_total);
    NND_HIP_CHECK(hipMemsetAsync(ctx->leaf_flag, 0, (size_t)P, ctx->stream));
    long long *counts = ctx->counters + CNT_SCRATCH + 1;  // [0] workgroup list (what launch_finishers reads), [1] big, [2] small
    NND_HIP_CHECK(hipMemsetAsync(counts, 0, 3 * sizeof(long long), ctx->stream));
    int32_t *fin_start = ctx->seg_child + 2 * ctx->max_segs, *fin_len = fin_start + ctx->max_segs, *fin_depth = fin_len + ctx->max_segs;
    int32_t *big_start = ctx->seg_start[0], *big_len = ctx->seg_len[0], *big_depth = ctx->seg_nleft;
    hipLaunchKernelGGL(k_cell_lists, dim3((unsigned)((n_cells + 255) / 256)), dim3(256), 0, ctx->stream, ctx->cell_count, ctx->cell_start,
                       ctx->cell_depth, (int)n_cells, FIN_SMALL, FIN_MAX, ctx->small_list, fin_start, fin_len, fin_depth, big_start,
                       big_len, big_depth, ctx->cell_cap, counts);
    hipLaunchKernelGGL(k_place, dim3((unsigned)((n + 255) / 256), (unsigned)T), dim3(256), 0, ctx->stream, ctx->pos_seg[0], ctx->pos_seg[1],
                       ctx->cell_start, n, ctx->perm[0]);
    NND_HIP_CHECK(hipGetLastError());
    NND_HIP_CHECK(hipMemcpyAsync(ctx->h_pin + 36, counts + 1, 2 * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    NND_HIP_CHECK(nnd_sync_spin(ctx));
    const long long n_big = ctx->h_pin[36], n_small = ctx->h_pin[37];
    if (n_big > ctx->max_segs) {
        if (nnd_knob("NND_FOREST_DEBUG")) fprintf(stderr, "forest: fallback at line %d\n", __LINE__);
        return 2;
    }
    if (launch_finishers(ctx, ctx->perm[0], ctx->perm[1], big_start, big_len, big_depth, 0, n_big, n_small)) return 1;
    *levels_out = v.depth;
    ctx->cur = 0;
    ctx->stats.n_cells = n_cells;
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
    int32_t *scan_total = (int32_t *)(ctx->counters + CNT_SCRATCH);  // device scratch word(s)
    unsigned gridP = (unsigned)((P + 255) / 256);
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
    // leaf tables
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
