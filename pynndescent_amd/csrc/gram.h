// gram.h -- LDS staging of gathered point rows and the f32 MFMA Gram-tile inner loop shared by
// the leaf kernel and the local-join kernel.
//
// The dense |A| x |B| block of pairwise distances is a true Gram contraction G = A . B^T; it is
// the only place MFMA is used.  v_mfma_f32_16x16x4_f32 is exact f32 (an fmaf chain), runs at the
// f32 vector rate and takes ONE VGPR per operand per lane:
//     lane l supplies  A[row = l & 15][kk = l >> 4]  and  B[kk = l >> 4][col = l & 15]
//     lane l receives  D[row = 4 * (l >> 4) + r][col = l & 15]   in register r = 0..3
// K mapping: a 16-byte chunk c of a row (floats 4c..4c+3) is consumed by lane group g = c & 3 in
// four consecutive MFMA steps; rows are staged with an XOR swizzle (common.h nnd_swz) so that the
// ds_read_b128 operand fetches are bank-conflict free.
#pragma once
#include "common.h"

// Stage rows ids[0..nrows) (point ids, -1 = empty -> zeros), floats [c0, c0+cw) of each, into Xs.
template <int DC>
__device__ __forceinline__ void nnd_stage_rows(const float *__restrict__ xp, int dp, const int32_t *ids, int nrows,
                                               int c0, int cw, float *Xs, int tid, int nthreads) {
    const int nch = cw >> 2;
    const int total = nrows * nch;
#pragma unroll 4
    for (int idx = tid; idx < total; idx += nthreads) {
        int r = idx / nch, ch = idx - r * nch;
        int id = ids[r];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0) v = *(const float4 *)(xp + (int64_t)id * dp + c0 + 4 * ch);
        *(float4 *)&Xs[nnd_swz<DC>(r, ch)] = v;
    }
}

// acc[J] += A(rows a_base..a_base+15) . B(rows J*16 + b_base ..)^T over the staged chunk of cw floats.
// tile_on(J) is wave-uniform: tiles that are off are skipped.
template <int DC, int NTILES, typename TileOn>
__device__ __forceinline__ void nnd_gram_chunk(const float *Xs, int a_base, int b_base, int cw, f32x4 (&acc)[NTILES],
                                               TileOn tile_on) {
    const int lane = nnd_lane();
    const int r16 = lane & 15, g = lane >> 4;
    const int nq = cw >> 4;  // 16-float groups: each holds one 16-byte chunk for each of the 4 lane groups
    for (int t = 0; t < nq; t++) {
        const int c = 4 * t + g;
        const float4 a = *(const float4 *)&Xs[nnd_swz<DC>(a_base + r16, c)];
        float4 b[NTILES];
#pragma unroll
        for (int J = 0; J < NTILES; J++)
            if (tile_on(J)) b[J] = *(const float4 *)&Xs[nnd_swz<DC>(b_base + J * 16 + r16, c)];
        // the four k-steps of a chunk, tiles interleaved so consecutive MFMAs use different accumulators
#pragma unroll
        for (int J = 0; J < NTILES; J++)
            if (tile_on(J)) acc[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[J].x, acc[J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J < NTILES; J++)
            if (tile_on(J)) acc[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[J].y, acc[J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J < NTILES; J++)
            if (tile_on(J)) acc[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[J].z, acc[J], 0, 0, 0);
#pragma unroll
        for (int J = 0; J < NTILES; J++)
            if (tile_on(J)) acc[J] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[J].w, acc[J], 0, 0, 0);
    }
}
