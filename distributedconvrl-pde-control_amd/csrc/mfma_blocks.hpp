// mfma_blocks.hpp -- device building blocks shared by the fused fp32 MFMA DDPG passes (mlp_mfma.hip: 3-layer nets,
// mlp_mfma2.hip: 2-layer nets).  v_mfma_f32_16x16x4_f32 tiles: an output tile D[16 rows][16 cols] lives in 4 VGPRs
// per lane (col = lane & 15, row = 4 * (lane >> 4) + r).
#pragma once
// gradient slab accesses: non-temporal by default (round 2); -DPDEC_SLAB_NT=0 / -DPDEC_SLAB_NT_LOAD=0 build the plain forms (A/B builds)
#ifndef PDEC_SLAB_NT
#define PDEC_SLAB_NT 1
#endif
#ifndef PDEC_SLAB_NT_LOAD
#define PDEC_SLAB_NT_LOAD PDEC_SLAB_NT
#endif
#if PDEC_SLAB_NT
#define PDEC_SLAB_STORE(v, p) __builtin_nontemporal_store((v), (p))
#else
#define PDEC_SLAB_STORE(v, p) (*(p) = (v))
#endif
#if PDEC_SLAB_NT_LOAD
#define PDEC_SLAB_LOAD(p) __builtin_nontemporal_load(p)
#else
#define PDEC_SLAB_LOAD(p) (*(p))
#endif

#include "common.hpp"

namespace pdec {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FCOLS 128          // columns per workgroup
#define FTHREADS 512
// Leading dimensions chosen so that the ds_read_b128 operand reads (lane (lr, q) reads row lr at column offset 4q)
// are bank-conflict free: with a row stride of 8 (mod 16) floats the 16-byte slot index is (2 lr + q) mod 16,
// which is distinct over each 16-lane service group of ds_read_b128 (MI355X_MICROARCH.md, LDS table).
#define LDP 72             // leading dim of the 64-column LDS transposition images

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int MT>
__device__ __forceinline__ void relu_(f32x4 (&h)[MT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) h[m][r] = fmaxf(h[m][r], 0.f);
}

// scalar head: b3 + sum_rows w3[row] * h[row][col]; every lane of a column gets the result
template <int MT>
__device__ __forceinline__ float head(const f32x4 (&h)[MT], const float* w3, float b3, int q) {
  float acc = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float* w = w3 + 16 * m + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += w[r] * h[m][r];
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  return acc + b3;
}

// dz[row][col] = w3[row] * g[col] * (h > 0)
template <int MT>
__device__ __forceinline__ void head_bwd(f32x4 (&dz)[MT], const f32x4 (&h)[MT], const float* w3, float g, int q) {
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const float* w = w3 + 16 * m + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) dz[m][r] = h[m][r] > 0.f ? w[r] * g : 0.f;
  }
}

// write a D-layout activation (MT tiles) into a [feature][LDP] LDS image at column cw;
// ones_row >= 0 additionally sets that row to 1 (bias gradient rides the GEMM)
template <int MT>
__device__ __forceinline__ void stage_rows(float* img, const f32x4 (&v)[MT], int cw, int q, int ones_row) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * m + 4 * q + r;
      img[row * LDP + cw] = row == ones_row ? 1.f : v[m][r];
    }
}

// D[i][k] += sum over the 64 staged columns L[i][c] * R[k][c]; wave w takes tile pairs w, w+8, ...
template <int NACC>
__device__ __forceinline__ void gemm_pass(f32x4 (&acc)[NACC], const float* L, const float* R, int nL, int nR, int w,
                                          int lr, int q) {
#pragma unroll
  for (int pp = 0; pp < NACC; ++pp) {
    const int p = w + 8 * pp;
    if (p < nL * nR) {
      const int ti = p / nR, tk = p - ti * nR;
      const float* lrow = L + (16 * ti + lr) * LDP + 4 * q;
      const float* rrow = R + (16 * tk + lr) * LDP + 4 * q;
      f32x4 a = acc[pp];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(lrow + 16 * t);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(rrow + 16 * t);
        a = mfma4(av[0], bv[0], a);
        a = mfma4(av[1], bv[1], a);
        a = mfma4(av[2], bv[2], a);
        a = mfma4(av[3], bv[3], a);
      }
      acc[pp] = a;
    }
    // keep the scheduler from hoisting the next tiles' operand loads over this tile (register pressure: the
    // whole dz2 / h1 / dz1 activation set is live here); the partner wave on the SIMD hides the LDS latency
    __builtin_amdgcn_sched_barrier(0);
  }
}
// the same product with the tiles of a wave taken in PAIRS whose MFMAs alternate: two independent accumulation chains
// per wave, so consecutive MFMAs never wait for the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 (issue
// interval 32).  ldp = leading dimension of both images (8 mod 16 floats: conflict-free ds_read_b128), NT = staged
// columns / 16.
template <int NACC, int NT>
__device__ __forceinline__ void gemm_pass_paired(f32x4 (&acc)[NACC], const float* L, const float* R, int ldp, int nL, int nR,
                                                 int w, int lr, int q) {
#pragma unroll
  for (int pp = 0; pp < NACC; pp += 2) {
    const int p0 = w + 8 * pp, p1 = p0 + 8;
    if (p0 < nL * nR) {
      const int ti0 = p0 / nR, tk0 = p0 - ti0 * nR;
      const float* l0 = L + (16 * ti0 + lr) * ldp + 4 * q;
      const float* r0 = R + (16 * tk0 + lr) * ldp + 4 * q;
      if (pp + 1 < NACC && p1 < nL * nR) {
        const int ti1 = p1 / nR, tk1 = p1 - ti1 * nR;
        const float* l1 = L + (16 * ti1 + lr) * ldp + 4 * q;
        const float* r1 = R + (16 * tk1 + lr) * ldp + 4 * q;
        f32x4 a0 = acc[pp], a1 = acc[pp + 1 < NACC ? pp + 1 : pp];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 av0 = *reinterpret_cast<const f32x4*>(l0 + 16 * t);
          const f32x4 bv0 = *reinterpret_cast<const f32x4*>(r0 + 16 * t);
          const f32x4 av1 = *reinterpret_cast<const f32x4*>(l1 + 16 * t);
          const f32x4 bv1 = *reinterpret_cast<const f32x4*>(r1 + 16 * t);
          a0 = mfma4(av0[0], bv0[0], a0);
          a1 = mfma4(av1[0], bv1[0], a1);
          a0 = mfma4(av0[1], bv0[1], a0);
          a1 = mfma4(av1[1], bv1[1], a1);
          a0 = mfma4(av0[2], bv0[2], a0);
          a1 = mfma4(av1[2], bv1[2], a1);
          a0 = mfma4(av0[3], bv0[3], a0);
          a1 = mfma4(av1[3], bv1[3], a1);
        }
        acc[pp] = a0;
        acc[pp + 1 < NACC ? pp + 1 : pp] = a1;
      } else {
        f32x4 a0 = acc[pp];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(l0 + 16 * t);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(r0 + 16 * t);
          a0 = mfma4(av[0], bv[0], a0);
          a0 = mfma4(av[1], bv[1], a0);
          a0 = mfma4(av[2], bv[2], a0);
          a0 = mfma4(av[3], bv[3], a0);
        }
        acc[pp] = a0;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// gemm_pass_paired for the LAST round of a pass: every tile pair goes to its slab chunks right behind its own MFMAs instead of
// all tiles at the end of the pass -- the 21 MB of dW2 partials then leave the chip spread over the round's ~10 k cycles of
// MFMA work instead of as one burst that the end of the kernel waits for (~4 us of HBM write time at 256 workgroups).
template <int NACC, int NT>
__device__ __forceinline__ void gemm_pass_paired_store(f32x4 (&acc)[NACC], const float* L, const float* R, int ldp, int nL, int nR,
                                                       int w, int lr, int q, float* slab, int nslab, int T0, int l) {
#pragma unroll
  for (int pp = 0; pp < NACC; pp += 2) {
    const int p0 = w + 8 * pp, p1 = p0 + 8;
    if (p0 < nL * nR) {
      const int ti0 = p0 / nR, tk0 = p0 - ti0 * nR;
      const float* l0 = L + (16 * ti0 + lr) * ldp + 4 * q;
      const float* r0 = R + (16 * tk0 + lr) * ldp + 4 * q;
      const bool two = pp + 1 < NACC && p1 < nL * nR;
      const int ti1 = two ? p1 / nR : ti0, tk1 = two ? p1 - ti1 * nR : tk0;
      const float* l1 = L + (16 * ti1 + lr) * ldp + 4 * q;
      const float* r1 = R + (16 * tk1 + lr) * ldp + 4 * q;
      f32x4 a0 = acc[pp], a1 = acc[pp + 1 < NACC ? pp + 1 : pp];
      if (two) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 av0 = *reinterpret_cast<const f32x4*>(l0 + 16 * t);
          const f32x4 bv0 = *reinterpret_cast<const f32x4*>(r0 + 16 * t);
          const f32x4 av1 = *reinterpret_cast<const f32x4*>(l1 + 16 * t);
          const f32x4 bv1 = *reinterpret_cast<const f32x4*>(r1 + 16 * t);
          a0 = mfma4(av0[0], bv0[0], a0);
          a1 = mfma4(av1[0], bv1[0], a1);
          a0 = mfma4(av0[1], bv0[1], a0);
          a1 = mfma4(av1[1], bv1[1], a1);
          a0 = mfma4(av0[2], bv0[2], a0);
          a1 = mfma4(av1[2], bv1[2], a1);
          a0 = mfma4(av0[3], bv0[3], a0);
          a1 = mfma4(av1[3], bv1[3], a1);
        }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(l0 + 16 * t);
          const f32x4 bv = *reinterpret_cast<const f32x4*>(r0 + 16 * t);
          a0 = mfma4(av[0], bv[0], a0);
          a0 = mfma4(av[1], bv[1], a0);
          a0 = mfma4(av[2], bv[2], a0);
          a0 = mfma4(av[3], bv[3], a0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        PDEC_SLAB_STORE(a0[r], &slab[((size_t)(4 * (T0 + p0) + r) * nslab + blockIdx.x) * 64 + l]);
        if (two) PDEC_SLAB_STORE(a1[r], &slab[((size_t)(4 * (T0 + p1) + r) * nslab + blockIdx.x) * 64 + l]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// write a D-layout activation (MT tiles) into a [feature][ldp] LDS image at column cw (any leading dimension)
template <int MT>
__device__ __forceinline__ void stage_rows_ld(float* img, int ldp, const f32x4 (&v)[MT], int cw, int q, int ones_row) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * m + 4 * q + r;
      img[row * ldp + cw] = row == ones_row ? 1.f : v[m][r];
    }
}
// Slab layout (chunk-major): a "chunk" is one accumulator register of one 16x16 output tile = 64 floats in lane
// order; chunk c of workgroup z lives at ((c * nslab + z) * 64).  A wave therefore stores 256 contiguous bytes
// per instruction, and the reduction kernel streams one contiguous nslab * 256 B region per chunk.
// Tiles are numbered T = T0 + ti * nR + tk; element (r, lane) of a tile is row 16 ti + 4 (lane>>4) + r,
// column 16 tk + (lane & 15) of the pass's output matrix.
template <int NACC>
__device__ __forceinline__ void store_pass(const f32x4 (&acc)[NACC], float* slab, int nslab, int T0, int nL, int nR, int w, int l) {
#pragma unroll
  for (int pp = 0; pp < NACC; ++pp) {
    const int p = w + 8 * pp;
    if (p < nL * nR) {
#pragma unroll
      for (int r = 0; r < 4; ++r)      // written once, read once by the reduction kernel: streaming (non-temporal) stores keep
                                       // the 26 MB of partials from sitting dirty in the XCD's L2 at the kernel boundary
        PDEC_SLAB_STORE(acc[pp][r], &slab[((size_t)(4 * (T0 + p) + r) * nslab + blockIdx.x) * 64 + l]);
    }
  }
}
// sum over the 16 lanes of a DPP row (= the 16 columns a wave owns), result in every lane; pure VALU
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // lane^1
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // lane^2
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x124, 0xF, 0xF, true));  // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
  return v;
}
template <int N>
__device__ __forceinline__ void zero_(f32x4 (&a)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// block-wide deterministic sum (fixed tree)
__device__ __forceinline__ float block_sum(float v, float* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < FTHREADS / 64; ++i) r += red[i];
  __syncthreads();
  return r;
}


}  // namespace pdec
