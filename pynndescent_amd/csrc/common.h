// common.h -- shared device helpers for the gfx950 NN-Descent kernels.
// Wave = 64 lanes everywhere (CDNA4); nothing here is portable to 32-wide warps on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define NND_WAVE 64
#define NND_EMPTY_E 0xFFFFFFFFu           // empty k-list slot (reference: index -1, utils.py:153)
#define NND_NEW_BIT 0x80000000u           // "new" flag packed into bit 31 of the neighbour word (utils.py:155 flags)
#define NND_IDX_MASK 0x7FFFFFFFu
#define NND_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull
#define NND_EMPTY_SLOT 0xFFFFFFFFu  // an unarmed reverse-offer slot (sample.hip)
#define NND_FLT_MAX 3.402823466e+38f
// rows of more than 64 neighbours (n_neighbors up to NND_WIDE_K): a lane holds entries lane, 64 + lane, ...; merged through LDS (merge.h)
#define NND_WIDE_K 256
#define NND_WIDE_U (NND_WIDE_K / 64)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------- hashing --
// Counter-based RNG: every random decision is a pure function of (seed, ids), so the build is
// deterministic for a given seed on any launch geometry.  Replaces the sequential Tausworthe
// streams of the reference (utils.py:17-57); only statistical equivalence is required
// (SURVEY.md Appendix A5 "quirk" notes).
__host__ __device__ __forceinline__ uint32_t nnd_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU;
    x ^= x >> 15; x *= 0x846ca68bU;
    x ^= x >> 16;
    return x;
}
// nnd_mix32 is a bijection of the 32-bit words; this is its inverse (each xorshift and each odd multiplier undone in turn)
__host__ __device__ __forceinline__ uint32_t nnd_unmix32(uint32_t x) {
    x ^= x >> 16; x *= 0x43021123U;
    x ^= (x >> 15) ^ (x >> 30); x *= 0x1d69e2a5U;
    x ^= x >> 16;
    return x;
}
__host__ __device__ __forceinline__ uint32_t nnd_hash2(uint32_t seed, uint32_t a) {
    return nnd_mix32(seed ^ nnd_mix32(a + 0x9E3779B9u));
}
__host__ __device__ __forceinline__ uint32_t nnd_hash3(uint32_t seed, uint32_t a, uint32_t b) {
    return nnd_mix32(nnd_hash2(seed, a) ^ nnd_mix32(b * 0x85EBCA6Bu + 0xC2B2AE35u));
}

// ------------------------------------------------------------ wave helpers --
__device__ __forceinline__ int nnd_lane() { return (int)(threadIdx.x & 63); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int nnd_prefix_popc(unsigned long long mask) {
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

__device__ __forceinline__ int nnd_wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float nnd_wave_sum_f32(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double nnd_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// sum over the 16 lanes of an aligned 16-lane group
__device__ __forceinline__ float nnd_group16_sum_f32(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- wave reductions without the LDS crossbar.  __shfl_xor is ds_bpermute_b32: an LDS-pipeline round trip per step, six
// dependent ones per reduction -- ~700 cycles that a latency-bound kernel (one wave per tree cell, a few reductions per
// node) cannot hide.  DPP row rotations are VALU operands: four steps leave every lane with its ROW's result, v_readlane
// fetches the four row results, three more operations combine them.  Every lane must be active (EXEC all ones).
template <int CTRL>
__device__ __forceinline__ int nnd_dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); }
#define NND_DPP_ROW_SHR(n) (0x110 + (n))
#define NND_DPP_ROW_ROR(n) (0x120 + (n))
// the same value in every lane: rows by rotations (lanes of a row may differ in the last bit: lane 0 of the row is taken),
// then (r0 + r1) + (r2 + r3)
__device__ __forceinline__ float nnd_wave_sum_f32_u(float v) {
    v += __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(8)>(__float_as_int(v)));
    v += __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(4)>(__float_as_int(v)));
    v += __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(2)>(__float_as_int(v)));
    v += __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(1)>(__float_as_int(v)));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ float nnd_wave_max_f32_u(float v) {
    v = fmaxf(v, __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(8)>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(4)>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(2)>(__float_as_int(v))));
    v = fmaxf(v, __int_as_float(nnd_dpp_i32<NND_DPP_ROW_ROR(1)>(__float_as_int(v))));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}
__device__ __forceinline__ uint64_t nnd_dpp_min_u64_step(uint64_t k, uint32_t olo, uint32_t ohi) {
    const uint64_t o = ((uint64_t)ohi << 32) | olo;
    return o < k ? o : k;
}
// smallest 64-bit key of the wave, in every lane
__device__ __forceinline__ uint64_t nnd_wave_min_u64_u(uint64_t k) {
    k = nnd_dpp_min_u64_step(k, (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(8)>((int)(uint32_t)k), (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(8)>((int)(uint32_t)(k >> 32)));
    k = nnd_dpp_min_u64_step(k, (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(4)>((int)(uint32_t)k), (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(4)>((int)(uint32_t)(k >> 32)));
    k = nnd_dpp_min_u64_step(k, (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(2)>((int)(uint32_t)k), (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(2)>((int)(uint32_t)(k >> 32)));
    k = nnd_dpp_min_u64_step(k, (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(1)>((int)(uint32_t)k), (uint32_t)nnd_dpp_i32<NND_DPP_ROW_ROR(1)>((int)(uint32_t)(k >> 32)));
    uint64_t best = ~0ull;
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const uint64_t o = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(k >> 32), 16 * r) << 32) |
                           (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)k, 16 * r);
        best = o < best ? o : best;
    }
    return best;
}
// inclusive prefix sum over the lanes of the wave
__device__ __forceinline__ int nnd_wave_incl_scan_i32(int v) {
    v += nnd_dpp_i32<NND_DPP_ROW_SHR(1)>(v);  // (lanes shifted in from outside the row read 0)
    v += nnd_dpp_i32<NND_DPP_ROW_SHR(2)>(v);
    v += nnd_dpp_i32<NND_DPP_ROW_SHR(4)>(v);
    v += nnd_dpp_i32<NND_DPP_ROW_SHR(8)>(v);
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = nnd_lane() >> 4;
    return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}

// Make this wave's LDS writes visible to its own later LDS reads (cross-lane through LDS).
// One wave executes LDS operations in order; this only stops the compiler from reordering
// and waits for outstanding LDS traffic.
__device__ __forceinline__ void nnd_wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// ------------------------------------------------------------ key packing --
// Non-negative floats order like their bit patterns; every distance this code ranks is clamped >= 0.
__device__ __forceinline__ uint64_t nnd_make_key(float d, uint32_t idx) {
    return ((uint64_t)__float_as_uint(d) << 32) | (uint64_t)(idx & NND_IDX_MASK);
}
__device__ __forceinline__ float nnd_key_dist(uint64_t key) { return __uint_as_float((uint32_t)(key >> 32)); }
__device__ __forceinline__ uint32_t nnd_key_idx(uint64_t key) { return (uint32_t)key & NND_IDX_MASK; }

// Screening copies of rows and hyperplanes (rp forest): IEEE half precision of (value * scale), scale a power of two chosen
// by the prep kernel so that the data sits well inside the half range (state.h nnd_ctx::mean[dp]).  11 significant bits:
// the rounding error -- and with it the band of margins that need the exact f32 recheck -- is 8x smaller than with the
// bf16 copies of rounds 1-3 (2.5 % of the margins fell inside that band, and a recheck is a dependent global fetch inside
// a divergent branch: it is priced per wave).  Subnormal results are flushed to zero HERE, so that whatever the packed
// dot-product instruction does with subnormal inputs cannot matter: the residual norms (|x - half(x)|, stored per row and
// per hyperplane) are computed from the value actually stored and keep the band rigorous.  Overflow gives inf, an inf
// residual, an infinite band: such a row is always rechecked exactly.
__device__ __forceinline__ uint16_t nnd_f32_to_h16(float v, float scale) {
    uint16_t b = __builtin_bit_cast(uint16_t, (_Float16)(v * scale));  // round to nearest even
    if ((b & 0x7C00u) == 0) b &= 0x8000u;
    return b;
}
__device__ __forceinline__ float nnd_h16_to_f32(uint16_t b, float inv_scale) { return (float)__builtin_bit_cast(_Float16, b) * inv_scale; }

__device__ __forceinline__ float nnd_clamp_dist(float d) { return d > 0.0f ? d : 0.0f; }

// Gram value -> alt-space distance.
//   euclid: |a|^2 + |b|^2 - 2<a,b>          (reference distances.py:63-91 in Gram form)
//   cosine: rows are pre-normalised, na/nb are 1 (non-zero row) or 0 (zero row):
//           0 if both zero, FLT_MAX if one zero or <a,b> <= 0, else -log2(<a,b>) (distances.py:583-630)
__device__ __forceinline__ float nnd_gram_to_dist(int metric, float g, float na, float nb) {
    if (metric == 0) return nnd_clamp_dist(na + nb - 2.0f * g);
    if (na == 0.0f && nb == 0.0f) return 0.0f;
    if (na == 0.0f || nb == 0.0f || g <= 0.0f) return NND_FLT_MAX;
    return nnd_clamp_dist(-__log2f(g));
}

// Swizzled LDS addressing for row tiles read as MFMA operands.
// A tile row holds DC floats = DC/4 16-byte chunks.  Chunk c of row r is stored at chunk position
// c ^ (r & SWZ_MASK): rows that differ in their low bits land in different 16-byte bank slots, so
// the 16 lanes of an MFMA operand group (16 consecutive rows, same chunk) read conflict-free.
template <int DC>
__device__ __forceinline__ int nnd_swz(int row, int chunk) {
    constexpr int NCH = DC / 4;
    constexpr int MASK = (NCH >= 16) ? 15 : (NCH - 1);
    return row * DC + ((chunk ^ (row & MASK)) << 2);
}

// one atomic per workgroup per counter, spread over stripes (see state.h)
__device__ __forceinline__ void nnd_count(long long *counters, int which, long long v) {
    if (v) atomicAdd((unsigned long long *)&counters[(size_t)(blockIdx.x & 511u) * 16 + which], (unsigned long long)v);
}

// row-sharded build: rank that owns vertex v; bounds[r] <= v < bounds[r + 1]
__device__ __forceinline__ int nnd_owner_of(const int64_t *__restrict__ bounds, int n_ranks, int64_t v) {
    int lo = 0, hi = n_ranks - 1;  // bounds[r] <= v < bounds[r + 1]
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (v >= bounds[mid]) lo = mid; else hi = mid - 1;
    }
    return lo;
}


#define NND_HIP_CHECK(expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            ctx->set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return 1;                                                                               \
        }                                                                                           \
    } while (0)
