// mlp_mfma2.hip -- fused fp32 MFMA path of the DDPG update for the reference-shaped 2-layer nets
// (`drop_middle_layer = true`, src/PDEagent.jl:30-41: actor Dense(ns,h,relu) -> Dense(h,1,tanh), critic
// Dense(ns+1,H,relu) -> Dense(H,1)) at large batches, e.g. the 2-D Keller-Segel case of BASELINE.json configs[3]:
// a 3x3 x 2 species x 2 steps window (ns = 36) over B*A = 100k columns, critic 37 -> 340 -> 1.  This is the
// "2-D conv as im2col-GEMM" contraction: [H x (ns+1)] x [(ns+1) x columns] on v_mfma_f32_16x16x4_f32.
// Restates src/PDEagent.jl:385-409 in the same two passes as mlp_mfma.hip:
//   ddpg2_critic_kernel : a' = At(s'), qt = Ct([s';a']), q = C([s;a]), loss statistics, dq, critic backward -> slab
//   ddpg2_actor_kernel  : a = A(s), q = C([s;a]), -mean(q), backward through the critic to da, actor backward -> slab
//   finish2_kernel      : deterministic slab sum -> flat gradients (+ ADAM, Polyak when no all-reduce comes between)
// Layout: one workgroup = 8 waves = 128 columns, a wave owns 16 columns.  The first-layer weights live in LDS as a
// padded image [HP][LDK]; the contraction index is blocked by 8 (k = 8 blk + 2 q + t, t = 0,1), so the A operand of
// two consecutive k-steps is ONE ds_read_b64 (row stride 8 KB + 4 floats = 2 x odd 8-byte slots: conflict-free over
// each 32-lane service group) and the B operand is the input row pair the lane loaded from HBM.  The output layer is
// a single row: its forward is a per-lane dot + two lane exchanges, its weight gradient a DPP row sum.  dW1 contracts
// over columns: dz1 and the inputs are transposed once through LDS and multiplied as MFMA tiles (mfma_blocks.hpp).
// The kernels read the FLAT parameter buffers (no padded copy to maintain).
#include "common.hpp"
#include "mlp.hpp"
#include "mfma_blocks.hpp"

namespace pdec {

#define K2MAX 6            // k-blocks of 8: ns + 1 <= 48

struct Net2 {              // flat parameters [W1 [H][K0] row-major, b1 [H], W2 [1][H], b2 [1]]
  const float* p;
  int K0, H, kb;           // kb = ceil(K0 / 8) k-blocks are multiplied; the image row stride is a template constant
};

struct Lds2 {
  float *W1, *b1, *w2, *b2;
};
__host__ __device__ inline int lds2_floats(int HP, int ldk) { return HP * ldk + 2 * HP + 4; }
__device__ __forceinline__ Lds2 carve2(float* base, int HP, int ldk) {
  Lds2 s;
  s.W1 = base; s.b1 = base + HP * ldk; s.w2 = s.b1 + HP; s.b2 = s.w2 + HP;
  return s;
}

// flat parameters -> padded LDS image: the H*K0 weights are streamed with 16-byte loads (8 in flight per lane) and
// scattered to their padded rows; the pad cells are zeroed by separate stores (disjoint addresses, no ordering needed)
__device__ __forceinline__ void load_net2(const Lds2& s, const Net2& n, int HP, int ldk, int tid) {
  const int K0 = n.K0, H = n.H, nw = H * K0;
  const int n4 = ((reinterpret_cast<uintptr_t>(n.p) & 15) == 0) ? nw / 4 : 0;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(n.p);
  auto put = [&](int idx, float v) {
    const int r = idx / K0;
    s.W1[r * ldk + (idx - r * K0)] = v;
  };
  int i = tid;
  for (; i + 7 * FTHREADS < n4; i += 8 * FTHREADS) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = w4[i + u * FTHREADS];
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int e = 0; e < 4; ++e) put(4 * (i + u * FTHREADS) + e, v[u][e]);
  }
  for (; i < n4; i += FTHREADS) {
    const f32x4 v = w4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) put(4 * i + e, v[e]);
  }
  for (int k = 4 * n4 + tid; k < nw; k += FTHREADS) put(k, n.p[k]);
  const int padc = ldk - K0;
  for (int k = tid; k < H * padc; k += FTHREADS) { const int r = k / padc; s.W1[r * ldk + K0 + (k - r * padc)] = 0.f; }
  for (int k = tid; k < (HP - H) * ldk; k += FTHREADS) s.W1[H * ldk + k] = 0.f;
  const float* b1 = n.p + (size_t)nw;
  for (int k = tid; k < HP; k += FTHREADS) {
    s.b1[k] = k < H ? b1[k] : 0.f;
    s.w2[k] = k < H ? b1[H + k] : 0.f;
  }
  if (tid == 0) s.b2[0] = b1[2 * H];
}

// first layer: h = relu(W1 x + b1), tiles kept (D layout)
template <int MT, int KB>
__device__ __forceinline__ void layer1_keep(f32x4 (&h)[MT], const float (&x)[K2MAX][2], const Lds2& s, int lr, int q) {
  constexpr int ldk = 8 * KB + 4;
#pragma unroll
  for (int mo = 0; mo < MT; mo += 2) {
    const bool two = mo + 1 < MT;
    const float* b = s.b1 + 16 * mo + 4 * q;
    f32x4 a0 = {b[0], b[1], b[2], b[3]};
    f32x4 a1 = two ? f32x4{b[16], b[17], b[18], b[19]} : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w0 = s.W1 + (16 * mo + lr) * ldk + 2 * q;
    const float* w1 = w0 + 16 * ldk;
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
        const float2 wa = *reinterpret_cast<const float2*>(w0 + 8 * blk);
        a0 = mfma4(wa.x, x[blk][0], a0);
        if (two) {
          const float2 wb = *reinterpret_cast<const float2*>(w1 + 8 * blk);
          a1 = mfma4(wb.x, x[blk][0], a1);
          a0 = mfma4(wa.y, x[blk][1], a0);
          a1 = mfma4(wb.y, x[blk][1], a1);
        } else {
          a0 = mfma4(wa.y, x[blk][1], a0);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r) a0[r] = fmaxf(a0[r], 0.f);
    h[mo] = a0;
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[r] = fmaxf(a1[r], 0.f);
      h[mo + 1] = a1;
    }
  }
}

// first layer + output row without keeping the tiles: returns w2 . relu(W1 x + b1) + b2 (every lane of a column)
template <int MT, int KB>
__device__ __forceinline__ float layer1_head(const float (&x)[K2MAX][2], const Lds2& s, int lr, int q) {
  constexpr int ldk = 8 * KB + 4;
  float acc = 0.f;
#pragma unroll
  for (int mo = 0; mo < MT; mo += 2) {
    const bool two = mo + 1 < MT;
    const float* b = s.b1 + 16 * mo + 4 * q;
    f32x4 a0 = {b[0], b[1], b[2], b[3]};
    f32x4 a1 = two ? f32x4{b[16], b[17], b[18], b[19]} : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w0 = s.W1 + (16 * mo + lr) * ldk + 2 * q;
    const float* w1 = w0 + 16 * ldk;
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
        const float2 wa = *reinterpret_cast<const float2*>(w0 + 8 * blk);
        a0 = mfma4(wa.x, x[blk][0], a0);
        if (two) {
          const float2 wb = *reinterpret_cast<const float2*>(w1 + 8 * blk);
          a1 = mfma4(wb.x, x[blk][0], a1);
          a0 = mfma4(wa.y, x[blk][1], a0);
          a1 = mfma4(wb.y, x[blk][1], a1);
        } else {
          a0 = mfma4(wa.y, x[blk][1], a0);
        }
      }
    const float* wv = s.w2 + 16 * mo + 4 * q;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc += wv[r] * fmaxf(a0[r], 0.f);
    if (two) {
#pragma unroll
      for (int r = 0; r < 4; ++r) acc += wv[16 + r] * fmaxf(a1[r], 0.f);
    }
  }
  acc += __shfl_xor(acc, 16);
  acc += __shfl_xor(acc, 32);
  return acc + s.b2[0];
}

// critic forward inside the actor pass: q partial = w2 . relu(W1 x + b1) and the input-gradient partial
// da = sum_i W1[i][ns] * relu'(z_i) * w2_i of this lane's rows, tile pair by tile pair (no tile is kept)
template <int MT, int KB>
__device__ __forceinline__ void layer1_head_da(const float (&x)[K2MAX][2], const Lds2& s, int ns, int lr, int q, float& qacc,
                                               float& da) {
  constexpr int ldk = 8 * KB + 4;
#pragma unroll
  for (int mo = 0; mo < MT; mo += 2) {
    const bool two = mo + 1 < MT;
    const float* b = s.b1 + 16 * mo + 4 * q;
    f32x4 a0 = {b[0], b[1], b[2], b[3]};
    f32x4 a1 = two ? f32x4{b[16], b[17], b[18], b[19]} : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* w0 = s.W1 + (16 * mo + lr) * ldk + 2 * q;
    const float* w1 = w0 + 16 * ldk;
#pragma unroll
    for (int blk = 0; blk < KB; ++blk) {
        const float2 wa = *reinterpret_cast<const float2*>(w0 + 8 * blk);
        a0 = mfma4(wa.x, x[blk][0], a0);
        if (two) {
          const float2 wb = *reinterpret_cast<const float2*>(w1 + 8 * blk);
          a1 = mfma4(wb.x, x[blk][0], a1);
          a0 = mfma4(wa.y, x[blk][1], a0);
          a1 = mfma4(wb.y, x[blk][1], a1);
        } else {
          a0 = mfma4(wa.y, x[blk][1], a0);
        }
      }
    {   // unconditional operand loads (a select, not a branch around the load)
      const int row0 = 16 * mo + 4 * q;
      const f32x4 wv = *reinterpret_cast<const f32x4*>(s.w2 + row0);
      const float* wc = s.W1 + row0 * ldk + ns;
      const float c0 = wc[0], c1 = wc[ldk], c2 = wc[2 * ldk], c3 = wc[3 * ldk];
      const float cc[4] = {c0, c1, c2, c3};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        qacc += wv[r] * fmaxf(a0[r], 0.f);
        const float t = cc[r] * wv[r];
        da += a0[r] > 0.f ? t : 0.f;
      }
      if (two) {
        const f32x4 wv1 = *reinterpret_cast<const f32x4*>(s.w2 + row0 + 16);
        const float* wc1 = wc + 16 * ldk;
        const float d0 = wc1[0], d1 = wc1[ldk], d2 = wc1[2 * ldk], d3 = wc1[3 * ldk];
        const float dd[4] = {d0, d1, d2, d3};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          qacc += wv1[r] * fmaxf(a1[r], 0.f);
          const float t = dd[r] * wv1[r];
          da += a1[r] > 0.f ? t : 0.f;
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the scheduler from hoisting every tile pair's operand loads (spills)
  }
}

// input rows of a lane: x[blk][t] = X[8 blk + 2 q + t][col]; rows < ns from `s`, row ns from `extra` (the action), 0 above
__device__ __forceinline__ void load_x2(float (&x)[K2MAX][2], const float* __restrict__ s, size_t col, int ns, int kb, int q,
                                        bool valid) {
#pragma unroll
  for (int blk = 0; blk < K2MAX; ++blk)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 8 * blk + 2 * q + t;
      x[blk][t] = (valid && blk < kb && row < ns) ? s[col * ns + row] : 0.f;
    }
}
__device__ __forceinline__ void set_row2(float (&x)[K2MAX][2], int row, float v, int q) {
#pragma unroll
  for (int blk = 0; blk < K2MAX; ++blk)
#pragma unroll
    for (int t = 0; t < 2; ++t)
      if (8 * blk + 2 * q + t == row) x[blk][t] = v;
}
// write the lane's input rows into a [row][LDP] staging image (row `ones_row` := 1: the bias gradient rides the GEMM)
__device__ __forceinline__ void stage_x2(float* img, const float (&x)[K2MAX][2], int nrows, int cw, int q, int ones_row) {
#pragma unroll
  for (int blk = 0; blk < K2MAX; ++blk)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 8 * blk + 2 * q + t;
      if (row < nrows) img[row * LDP + cw] = row == ones_row ? 1.f : x[blk][t];
    }
}

struct Fused2Args {
  Net2 A, C, At, Ct;
  const float *s, *a, *r, *t, *sn;
  int Bu, ns;
  float gamma;
  int quirk;
  float* slab;
  const float* rbar;          // device scalar: mean reward (quirk) written by rmean_kernel
  int tpw;                    // wave tiles (16 columns) per workgroup: workgroup b owns tiles [b tpw, (b + 1) tpw) (grid2_of)
};

// mean of r in a fixed order (one block): the batch-mean reward of the reference's (1xBu).+(Bu) broadcast
__global__ __launch_bounds__(1024) void rmean_kernel(const float* __restrict__ r, int n, float* __restrict__ out) {
  __builtin_amdgcn_s_setprio(3);      // one short block; beside the update passes (priority 2) it would otherwise starve
  __shared__ float part[1024];
  const int tid = threadIdx.x;
  float acc = 0.f;
  const int n4 = ((reinterpret_cast<uintptr_t>(r) & 15) == 0) ? n / 4 : 0;
  const f32x4* r4 = reinterpret_cast<const f32x4*>(r);
  int i = tid;
  for (; i + 7 * 1024 < n4; i += 8 * 1024) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = r4[i + u * 1024];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
  }
  for (; i < n4; i += 1024) { const f32x4 v = r4[i]; acc += (v[0] + v[1]) + (v[2] + v[3]); }
  for (int k = 4 * n4 + tid; k < n; k += 1024) acc += r[k];
  part[tid] = acc;
  __syncthreads();
  for (int sft = 512; sft > 0; sft >>= 1) {
    if (tid < sft) part[tid] += part[tid + sft];
    __syncthreads();
  }
  if (tid == 0) out[0] = part[0] / (float)n;
}

// slab tiles of a 2-layer net: dW2/db2 as a [16][HP] product (MT tiles, row 0 real) | dW1/db1 [HP][16 nR] | stats chunk
__host__ __device__ inline int nr_of(int K0) { return (K0 + 1 + 15) / 16; }
__host__ __device__ inline int slab2_tiles(int MT, int nR) { return MT + MT * nR; }
static inline size_t slab2_floats(int MT, int nR, int nslab) { return ((size_t)4 * slab2_tiles(MT, nR) + 1) * nslab * 64; }

#define CPWMAX 4           // column chunks (of 128) one workgroup walks through: the weight images are loaded once per
                           // workgroup and the partial gradients of all its chunks share one slab
#define LDQ 40             // leading dim of the 32-column staging images (8 mod 16 floats: conflict-free ds_read_b128)

// D[i][k] += sum over NC16 * 16 staged columns L[i][c] * R[k][c] (images [row][LD]); wave w takes tiles w, w+8, ...
template <int NACC, int LD, int NC16>
__device__ __forceinline__ void gemm_cols(f32x4 (&acc)[NACC], const float* L, const float* R, int nL, int nR, int w, int lr,
                                          int q) {
#pragma unroll
  for (int pp = 0; pp < NACC; ++pp) {
    const int p = w + 8 * pp;
    if (p < nL * nR) {
      const int ti = p / nR, tk = p - ti * nR;
      const float* lrow = L + (16 * ti + lr) * LD + 4 * q;
      const float* rrow = R + (16 * tk + lr) * LD + 4 * q;
      f32x4 a = acc[pp];
#pragma unroll
      for (int t = 0; t < NC16; ++t) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(lrow + 16 * t);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(rrow + 16 * t);
        a = mfma4(av[0], bv[0], a);
        a = mfma4(av[1], bv[1], a);
        a = mfma4(av[2], bv[2], a);
        a = mfma4(av[3], bv[3], a);
      }
      acc[pp] = a;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// output-row gradient dW2/db2 += sum_cols g[col] * [h; 1]: DPP sum over a wave's 16 columns, LDS over the 8 waves,
// accumulated over the workgroup's chunks in dacc [HP] (fixed order -> deterministic)
template <int MT>
__device__ __forceinline__ void out_row_grad(const f32x4 (&h)[MT], float g, int H, float* redA, float* dacc, int tid) {
  const int w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4, HP = 16 * MT;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * m + 4 * q + r;
      const float v = row_sum16((row == H ? 1.f : h[m][r]) * g);
      if (lr == 0) redA[w * HP + row] = v;
    }
  __syncthreads();
  for (int i = tid; i < HP; i += FTHREADS) {
    float a = dacc[i];
#pragma unroll
    for (int ww = 0; ww < FTHREADS / 64; ++ww) a += redA[ww * HP + i];
    dacc[i] = a;
  }
}
__device__ __forceinline__ void store_row(const float* dacc, int HP, float* slab, int nslab, int T0, int tid) {
  for (int i = tid; i < HP; i += FTHREADS)
    slab[((size_t)(4 * (T0 + (i >> 4))) * nslab + blockIdx.x) * 64 + (i & 15)] = dacc[i];   // tile i/16, register 0, lane i%16
}

template <int LD>
__device__ __forceinline__ void stage_x2_ld(float* img, const float (&x)[K2MAX][2], int nrows, int cw, int q, int ones_row) {
#pragma unroll
  for (int blk = 0; blk < K2MAX; ++blk)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = 8 * blk + 2 * q + t;
      if (row < nrows) img[row * LD + cw] = row == ones_row ? 1.f : x[blk][t];
    }
}
template <int MT, int LD>
__device__ __forceinline__ void stage_rows_ld(float* img, const f32x4 (&v)[MT], int cw, int q) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) img[(16 * m + 4 * q + r) * LD + cw] = v[m][r];
}

// ------------------------------------------------------------------ critic pass
// LDS: [critic image: Ct, then C][staging DZ1 [HP][LDQ] + Xaug [16 nR][LDQ] (overlaid by the [8][HP] row reduction)]
//      [target actor image][dW2 accumulator [HP]]
template <int MT, int MTA, int KB>
__global__ __launch_bounds__(FTHREADS) void ddpg2_critic_kernel(Fused2Args g) {
  constexpr int LDK = 8 * KB + 4;
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  constexpr int HP = 16 * MT, HPa = 16 * MTA;
  const int ns = g.ns, K0 = ns + 1, nR = (K0 + 1 + 15) / 16;
  const int imgc = lds2_floats(HP, LDK);
  const int stg = max(HP * LDQ + 16 * nR * LDQ, 8 * HP);
  float* base = smem;
  const Lds2 SC = carve2(base, HP, LDK);
  float* Lm = base + imgc;                     // DZ1 [HP][LDQ]
  float* Rm = Lm + HP * LDQ;                   // Xaug [16 nR][LDQ]
  float* redA = Lm;
  const Lds2 SA = carve2(base + imgc + stg, HPa, LDK);
  float* dacc = SA.W1 + lds2_floats(HPa, LDK); // [HP]
  // Columns in units of WAVE TILES (16 columns): workgroup b owns the tiles [t0, t1) and walks them in chunks of 8 (one per
  // wave).  Round 6: the tiles are dealt evenly over the grid (grid2_of) instead of whole 128-column chunks strided over it --
  // config C4's 784 chunks came as 196 workgroups x 4 chunks (60 CUs idle, x0.77); now 256 workgroups take 24.5 tiles each: three
  // full chunks and one with a single live wave, in which the other waves skip their forward passes and the dW1 GEMM skips the
  // column quarters that hold no live wave.
  const int nt = (g.Bu + 15) / 16, t0 = blockIdx.x * g.tpw, t1 = min(t0 + g.tpw, nt);

  load_net2(SA, g.At, HPa, LDK, tid);
  load_net2(SC, g.Ct, HP, LDK, tid);
  for (int i = tid; i < HP; i += FTHREADS) dacc[i] = 0.f;
  const float rbar = g.quirk ? g.rbar[0] : 0.f;
  __syncthreads();
  // ---- targets of every chunk: a' = At(s'), qt = Ct([s'; a'])
  float tgt[CPWMAX];
#pragma unroll
  for (int j = 0; j < CPWMAX; ++j) {
    const int tile = t0 + 8 * j + w;
    tgt[j] = 0.f;
    if (tile < t1) {
      const int col = tile * 16 + lr;
      const bool valid = col < g.Bu;
      float xn[K2MAX][2];
      load_x2(xn, g.sn, (size_t)col, ns, KB, q, valid);
      const float tv = valid ? g.t[col] : 0.f;
      f32x4 ha[MTA];
      layer1_keep<MTA, KB>(ha, xn, SA, lr, q);
      const float an = tanhf(head<MTA>(ha, SA.w2, SA.b2[0], q));
      set_row2(xn, ns, valid ? an : 0.f, q);
      const float qt = layer1_head<MT, KB>(xn, SC, lr, q);
      tgt[j] = g.gamma * (1.f - tv) * qt;
    }
  }
  __syncthreads();                              // every wave is done with the target critic's image
  load_net2(SC, g.C, HP, LDK, tid);
  __syncthreads();
  // ---- q = C([s; a]), dq, gradients, chunk by chunk
  constexpr int NACC = (MT * 3 + 7) / 8;
  f32x4 acc[NACC];
  zero_(acc);
  float sv[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  const int cw = (w & 1) * 16 + lr;
#pragma unroll 1
  for (int j = 0; j < CPWMAX; ++j) {
    const int tb = t0 + 8 * j;                  // first tile of this chunk
    if (tb >= t1) break;
    const int live = min(8, t1 - tb);           // waves 0 .. live-1 hold columns (uniform over the workgroup)
    const float tgj = j == 0 ? tgt[0] : (j == 1 ? tgt[1] : (j == 2 ? tgt[2] : tgt[3]));
    const int col = (tb + w) * 16 + lr;
    const bool valid = w < live && col < g.Bu;
    float xq[K2MAX][2];
    f32x4 h1[MT];
    float dq = 0.f;
    if (w < live) {
      load_x2(xq, g.s, (size_t)col, ns, KB, q, valid);
      const float av = valid ? g.a[col] : 0.f, rv = valid ? g.r[col] : 0.f;
      set_row2(xq, ns, av, q);
      layer1_keep<MT, KB>(h1, xq, SC, lr, q);
      const float qv = head<MT>(h1, SC.w2, SC.b2[0], q);
      const float c = valid ? tgj - qv : 0.f;
      dq = valid ? -(2.f / (float)g.Bu) * ((g.quirk ? rbar : rv) + c) : 0.f;
      if (valid && q == 0) { sv[0] += c; sv[1] += c * c; sv[2] += rv; sv[3] += rv * rv; sv[4] += (rv + c) * (rv + c); }
    } else {                                    // a wave without columns in this chunk contributes zeros
#pragma unroll
      for (int blk = 0; blk < K2MAX; ++blk) { xq[blk][0] = 0.f; xq[blk][1] = 0.f; }
      zero_(h1);
    }
    __syncthreads();                            // the previous chunk's tiles have been consumed
    out_row_grad<MT>(h1, dq, g.C.H, redA, dacc, tid);      // dW2 / db2
    // dz1 in place; dW1 / db1 += dz1 x [x; 1]^T over the 128 columns (four 32-column quarters)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float* wv = SC.w2 + 16 * m + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) h1[m][r] = h1[m][r] > 0.f ? wv[r] * dq : 0.f;
    }
    for (int qq = 0; 2 * qq < live; ++qq) {     // (a quarter = the columns of waves 2 qq, 2 qq + 1)
      __syncthreads();
      if ((w >> 1) == qq) {
        stage_rows_ld<MT, LDQ>(Lm, h1, cw, q);
        stage_x2_ld<LDQ>(Rm, xq, 16 * nR, cw, q, w < live ? K0 : -1);      // (no bias-gradient ones in a dead wave's columns)
      }
      __syncthreads();
      gemm_cols<NACC, LDQ, 2>(acc, Lm, Rm, MT, nR, w, lr, q);
    }
  }
  const int nslab = gridDim.x;
  store_pass(acc, g.slab, nslab, MT, MT, nR, w, l);
  __syncthreads();
  store_row(dacc, HP, g.slab, nslab, 0, tid);
  // ---- loss statistics
#pragma unroll
  for (int k = 0; k < 5; ++k)
    for (int off = 32; off > 0; off >>= 1) sv[k] += __shfl_xor(sv[k], off);
  float* red5 = Lm;
  if (l == 0)
    for (int k = 0; k < 5; ++k) red5[w * 8 + k] = sv[k];
  __syncthreads();
  if (tid < 8) {
    float* st = g.slab + ((size_t)(4 * slab2_tiles(MT, nR)) * nslab + blockIdx.x) * 64;
    float a = 0.f;
    if (tid < 5)
      for (int ww = 0; ww < FTHREADS / 64; ++ww) a += red5[ww * 8 + tid];
    st[tid] = a;
  }
}

// ------------------------------------------------------------------ actor pass
// LDS: [critic image][actor image][staging DZA1 [HPa][LDP] + Saug [16 nRa][LDP]][row reduction [8][HPa]][dW2a accumulator]
template <int MT, int MTA, int KB>
__global__ __launch_bounds__(FTHREADS) void ddpg2_actor_kernel(Fused2Args g) {
  constexpr int LDK = 8 * KB + 4;
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  constexpr int HP = 16 * MT, HPa = 16 * MTA;
  const int ns = g.ns, nRa = (ns + 1 + 15) / 16;
  float* base = smem;
  const Lds2 SC = carve2(base, HP, LDK);
  const Lds2 SA = carve2(base + lds2_floats(HP, LDK), HPa, LDK);
  float* Lm = SA.W1 + lds2_floats(HPa, LDK);  // DZA1 [HPa][LDP]
  float* Rm = Lm + HPa * LDP;                  // Saug [16 nRa][LDP]
  float* redA = Rm + 16 * nRa * LDP;           // [8][HPa]
  float* dacc = redA + 8 * HPa;                // [HPa]
  float* red = dacc + HPa;                     // [8]
  const int nt = (g.Bu + 15) / 16, t0 = blockIdx.x * g.tpw, t1 = min(t0 + g.tpw, nt);     // wave tiles of this workgroup (critic pass)

  load_net2(SA, g.A, HPa, LDK, tid);
  load_net2(SC, g.C, HP, LDK, tid);
  for (int i = tid; i < HPa; i += FTHREADS) dacc[i] = 0.f;
  __syncthreads();
  f32x4 acc[1];
  zero_(acc);
  float st0 = 0.f;
  const int cw = (w & 3) * 16 + lr;
#pragma unroll 1
  for (int tb = t0; tb < t1; tb += 8) {
    const int live = min(8, t1 - tb);           // waves 0 .. live-1 hold columns (uniform over the workgroup)
    const int col = (tb + w) * 16 + lr;
    const bool valid = w < live && col < g.Bu;
    float xs[K2MAX][2];
    f32x4 ha[MTA];
    float dza2 = 0.f;
    if (w < live) {
      load_x2(xs, g.s, (size_t)col, ns, KB, q, valid);
      layer1_keep<MTA, KB>(ha, xs, SA, lr, q);
      const float aout = tanhf(head<MTA>(ha, SA.w2, SA.b2[0], q));
      float x[K2MAX][2];
#pragma unroll
      for (int blk = 0; blk < K2MAX; ++blk) { x[blk][0] = xs[blk][0]; x[blk][1] = xs[blk][1]; }
      set_row2(x, ns, valid ? aout : 0.f, q);
      // q = C([s; a]) and da = sum_i W1c[i][ns] * relu'(h1_i) * w2c_i * dq, tile by tile (nothing of h1 is kept)
      const float dq = valid ? -1.f / (float)g.Bu : 0.f;
      float qacc = 0.f, da = 0.f;
      layer1_head_da<MT, KB>(x, SC, ns, lr, q, qacc, da);
      qacc += __shfl_xor(qacc, 16);
      qacc += __shfl_xor(qacc, 32);
      da += __shfl_xor(da, 16);
      da += __shfl_xor(da, 32);
      if (valid && q == 0) st0 += qacc + SC.b2[0];
      dza2 = da * dq * (1.f - aout * aout);
    } else {                                    // a wave without columns in this chunk contributes zeros
#pragma unroll
      for (int blk = 0; blk < K2MAX; ++blk) { xs[blk][0] = 0.f; xs[blk][1] = 0.f; }
      zero_(ha);
    }
    __syncthreads();                            // the previous chunk's tiles have been consumed
    out_row_grad<MTA>(ha, dza2, g.A.H, redA, dacc, tid);
#pragma unroll
    for (int m = 0; m < MTA; ++m) {
      const float* wv = SA.w2 + 16 * m + 4 * q;
#pragma unroll
      for (int r = 0; r < 4; ++r) ha[m][r] = ha[m][r] > 0.f ? wv[r] * dza2 : 0.f;
    }
    for (int half = 0; 4 * half < live; ++half) {
      __syncthreads();
      if ((w >> 2) == half) {
        stage_rows<MTA>(Lm, ha, cw, q, -1);
        stage_x2(Rm, xs, 16 * nRa, cw, q, w < live ? ns : -1);
      }
      __syncthreads();
      gemm_pass(acc, Lm, Rm, MTA, nRa, w, lr, q);
    }
  }
  const int nslab = gridDim.x;
  store_pass(acc, g.slab, nslab, MTA, MTA, nRa, w, l);
  __syncthreads();
  store_row(dacc, HPa, g.slab, nslab, 0, tid);
  st0 = block_sum(st0, red, tid);
  if (tid == 0) {
    float* st = g.slab + ((size_t)(4 * slab2_tiles(MTA, nRa)) * nslab + blockIdx.x) * 64;
    st[0] = st0;
    st[1] = st[2] = st[3] = st[4] = 0.f;
  }
}

// ------------------------------------------------------------------ slab reduction (+ ADAM + Polyak)
struct Finish2Args {
  const float* slabs;
  int nslab, K0, H, MT, nR;
  float* grads;
  float scale;
  int mode, Bu, quirk;
  float* loss_out;
  int apply;
  float *p, *m, *v, *pt;
  double eta, b1, b2, eps, omb1p, omb2p;   // omb*p = 1 - beta^t: filled in by the kernels from `bp` (device resident)
  BpArgs bp;
  float rho, omr;
};

// Flux.Optimise.ADAM in Float64 (src/custom_nna.jl:23-24) and dest = rho dest + (1 - rho) src (src/PDEagent.jl:415-417);
// the same arithmetic, without FMA contraction, as adam_kernel / polyak_kernel (mlp.hip)
__device__ __forceinline__ void finish2_param(const Finish2Args& g, int i, float gi) {
#pragma clang fp contract(off)
  const double gd = (double)gi;
  const float mt = (float)(g.b1 * (double)g.m[i] + (1.0 - g.b1) * gd);
  const float vt = (float)(g.b2 * (double)g.v[i] + (1.0 - g.b2) * gd * gd);
  g.m[i] = mt;
  g.v[i] = vt;
  const float delta = (float)((double)mt / g.omb1p / (sqrt((double)vt / g.omb2p) + g.eps) * g.eta);
  const float pn = g.p[i] - delta;
  g.p[i] = pn;
  if (g.pt) g.pt[i] = g.rho * g.pt[i] + g.omr * pn;
}

// one block per chunk: 64 chunk elements x 16 slab groups, fixed-order combine -> deterministic
__global__ __launch_bounds__(1024) void finish2_kernel(Finish2Args g_in) {
  Finish2Args g = g_in;
  if (g.apply) {      // beta powers from device memory; one thread of the grid writes the advanced pair
    g.omb1p = 1.0 - g.bp.cur[0];
    g.omb2p = 1.0 - g.bp.cur[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_advance(g.bp, g.b1, g.b2);
  }
  __shared__ float part[16][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int K0 = g.K0, H = g.H, MT = g.MT, nR = g.nR;
  const int offb1 = H * K0, offW2 = offb1 + H, offb2 = offW2 + H;
  const int c = blockIdx.x, T = c >> 2, r = c & 3;
  int i = -1;
  {
    const int q = tx >> 4, lr = tx & 15;
    if (T < MT) {
      const int row = 4 * q + r, col = 16 * T + lr;
      if (row == 0) i = col < H ? offW2 + col : (col == H ? offb2 : -1);
    } else {
      const int u = T - MT, ti = u / nR, tk = u - ti * nR;
      const int row = 16 * ti + 4 * q + r, col = 16 * tk + lr;
      if (row < H) i = col < K0 ? row * K0 + col : (col == K0 ? offb1 + row : -1);
    }
  }
  if (__syncthreads_or(i >= 0)) {
    float acc = 0.f;
    if (i >= 0) {
      const float* sp = g.slabs + (size_t)c * g.nslab * 64 + tx;
      int z = ty;
      for (; z + 240 < g.nslab; z += 256) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = sp[(size_t)(z + 16 * u) * 64];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
      for (; z < g.nslab; z += 16) acc += sp[(size_t)z * 64];
    }
    part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && i >= 0) {
      float a = 0.f;
      for (int k = 0; k < 16; ++k) a += part[k][tx];
      a *= g.scale;
      g.grads[i] = a;
      if (g.apply) finish2_param(g, i, a);
    }
  }
  if (blockIdx.x == 0 && g.loss_out) {
    __shared__ double red[5][256];
    const int tid = threadIdx.x;
    const float* stc = g.slabs + (size_t)(4 * (MT + MT * nR)) * g.nslab * 64;
    double st[5] = {0, 0, 0, 0, 0};
    if (tid < 256) {
      for (int z = tid; z < g.nslab; z += 256)
        for (int k = 0; k < 5; ++k) st[k] += (double)stc[(size_t)z * 64 + k];
      for (int k = 0; k < 5; ++k) red[k][tid] = st[k];
    }
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
      if (tid < sft)
        for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + sft];
      __syncthreads();
    }
    if (tid == 0) {
      const double inv = 1.0 / g.Bu;
      if (g.mode == 0)
        *g.loss_out = (float)(g.quirk ? red[1][0] * inv + 2.0 * (red[0][0] * inv) * (red[2][0] * inv) + red[3][0] * inv
                                      : red[4][0] * inv);
      else
        *g.loss_out = (float)(-red[0][0] * inv);
    }
  }
}

// ADAM + Polyak from the flat gradient buffer (after an external all-reduce): same arithmetic as the fused finish
__global__ void apply2_kernel(Finish2Args g_in, int n) {
  Finish2Args g = g_in;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  g.omb1p = 1.0 - g.bp.cur[0];
  g.omb2p = 1.0 - g.bp.cur[1];
  if (i == 0) bp_advance(g.bp, g.b1, g.b2);
  if (i < n) finish2_param(g, i, g.grads[i]);
}

// ------------------------------------------------------------------ fused policy act (2-layer actor)
// actions = clamp(actor(state) + randn * act_noise, +-act_limit)   (src/PDEagent.jl:183-207) in ONE launch: first layer
// on MFMA out of the padded LDS image, output row + tanh, exploration noise from the Philox stream of pdec_randn
// (element index = column: both paths draw identical numbers), clamp.  256 threads = 4 waves x 16 columns.
template <int MTA, int KB>
__global__ __launch_bounds__(256) void policy_act2_kernel(Net2 n, const float* __restrict__ state, int cols, float act_noise,
                                                          float lim, int learning, int tanh_out, uint64_t seed, uint64_t offset,
                                                          float* __restrict__ out, const uint64_t* ctr_cur, uint64_t* ctr_next,
                                                          uint64_t ctr_inc) {
  constexpr int LDK = 8 * KB + 4, HPa = 16 * MTA;
  extern __shared__ __align__(16) float smem[];
  if (ctr_cur) {       // device-resident noise counter (pdec_policy_act_rng_dev)
    offset += *ctr_cur;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ctr_next = offset + ctr_inc;
  }
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  const Lds2 SA = carve2(smem, HPa, LDK);
  // image load with 256 threads (load_net2 assumes FTHREADS): plain row copy, the actor is small
  for (int i = tid; i < HPa * LDK; i += 256) {
    const int r = i / LDK, c = i - r * LDK;
    SA.W1[i] = (r < n.H && c < n.K0) ? n.p[(size_t)r * n.K0 + c] : 0.f;
  }
  const float* b1 = n.p + (size_t)n.H * n.K0;
  for (int i = tid; i < HPa; i += 256) { SA.b1[i] = i < n.H ? b1[i] : 0.f; SA.w2[i] = i < n.H ? b1[n.H + i] : 0.f; }
  if (tid == 0) SA.b2[0] = b1[2 * n.H];
  __syncthreads();
  const int c = blockIdx.x * 64 + w * 16 + lr;
  const bool valid = c < cols;
  float x[K2MAX][2];
  load_x2(x, state, (size_t)c, n.K0, KB, q, valid);
  f32x4 ha[MTA];
  layer1_keep<MTA, KB>(ha, x, SA, lr, q);
  float o = head<MTA>(ha, SA.w2, SA.b2[0], q);
  if (!valid || q != 0) return;
  if (tanh_out) o = tanhf(o);
  if (learning) {
    const uint64_t ctr = offset + (uint64_t)(c >> 2);
    uint32_t ph[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
    philox4x32(ph, (uint32_t)seed, (uint32_t)(seed >> 32));
    const int hsel = (c >> 1) & 1;
    const double sc = 1.0 / 4294967296.0;
    const double u1 = ((double)ph[2 * hsel] + 0.5) * sc, u2 = ((double)ph[2 * hsel + 1] + 0.5) * sc;
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
    const float z = (float)((c & 1) ? rad * sin(ang) : rad * cos(ang));
    o += z * act_noise;
  }
  out[c] = fminf(fmaxf(o, -lim), lim);
}

// ------------------------------------------------------------------ host side
static int mt2_of(int H) { return (H + 1 + 15) / 16; }

static bool fused2_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PDEC_DISABLE_FUSED");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

bool fused2_supported(const Mlp* A, const Mlp* C) {
  if (fused2_disabled()) return false;
  if (A->dtype != PDEC_F32 || C->dtype != PDEC_F32 || A->L != 2 || C->L != 2) return false;
  const int ns = A->dims[0];
  if (A->dims[2] != 1 || C->dims[0] != ns + 1 || C->dims[2] != 1 || ns + 2 > 8 * K2MAX) return false;
  if (A->acts[0] != PDEC_ACT_RELU || A->acts[1] != PDEC_ACT_TANH) return false;
  if (C->acts[0] != PDEC_ACT_RELU || C->acts[1] != PDEC_ACT_IDENTITY) return false;
  const int mt = mt2_of(C->dims[1]), mta = mt2_of(A->dims[1]);
  return (mt == 22 || mt == 9) && (mta == 2 || mta == 1);
}

// Workgroups of a pass and wave tiles (16 columns) per workgroup.  Up to one 128-column chunk per CU: one chunk each.  Beyond
// that the tiles are dealt EVENLY over 256 workgroups (one per CU: a pass's LDS images leave room for one), each walking
// ceil(tpw / 8) <= CPWMAX chunks; more than CPWMAX chunks per CU: more workgroups.  (Rounds 3 - 5 dealt whole chunks strided over
// ceil(nchunk / cpw) workgroups: C4's 784 chunks = 196 workgroups x 4.)
static int grid2_of(int Bu, int* tpw) {
  const int nt = (Bu + 15) / 16, nchunk = (nt + 7) / 8;
  if (nchunk <= 256) { *tpw = 8; return nchunk; }
  const int grid = std::max(256, (nchunk + CPWMAX - 1) / CPWMAX);
  *tpw = (nt + grid - 1) / grid;
  return (nt + *tpw - 1) / *tpw;
}

static Net2 net2_of(const Mlp* M) {
  Net2 n;
  n.p = M->params.as<float>();
  n.K0 = M->dims[0]; n.H = M->dims[1];
  n.kb = (n.K0 + 7) / 8;
  return n;
}

template <int MT, int MTA, int KB>
static size_t lds2_bytes(const Fused2Args& g, bool actor_pass) {
  const int HP = 16 * MT, HPa = 16 * MTA, LDK = 8 * KB + 4;
  const int imgc = lds2_floats(HP, LDK), imga = lds2_floats(HPa, LDK);
  size_t f;
  if (!actor_pass) {
    const int nR = nr_of(g.C.K0), stg = std::max(HP * LDQ + 16 * nR * LDQ, 8 * HP);
    f = (size_t)imgc + stg + imga + HP;
  } else {
    const int nRa = nr_of(g.A.K0);
    f = (size_t)imgc + imga + HPa * LDP + 16 * nRa * LDP + 8 * HPa + HPa + 8;
  }
  return f * 4;
}

template <int MT, int MTA, int KB>
static int launch2(Mlp* M, const Fused2Args& g, int grid, bool actor_pass) {
  const size_t lds = lds2_bytes<MT, MTA, KB>(g, actor_pass);
  PDEC_REQUIRE(lds <= 160 * 1024, "fused 2-layer pass needs %zu B of LDS", lds);
  const void* kern = actor_pass ? reinterpret_cast<const void*>(ddpg2_actor_kernel<MT, MTA, KB>)
                                : reinterpret_cast<const void*>(ddpg2_critic_kernel<MT, MTA, KB>);
  static size_t attr_lds[2] = {0, 0};          // per template instantiation and pass
  if (attr_lds[actor_pass] < lds) {
    PDEC_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_lds[actor_pass] = lds;
  }
  ProfScope ps(M, actor_pass ? "ddpg2_actor_fused" : "ddpg2_critic_fused", true);
  for (int rep = 0; rep < ps.reps; ++rep) {
    if (actor_pass) hipLaunchKernelGGL((ddpg2_actor_kernel<MT, MTA, KB>), dim3(grid), dim3(FTHREADS), lds, M->stream, g);
    else hipLaunchKernelGGL((ddpg2_critic_kernel<MT, MTA, KB>), dim3(grid), dim3(FTHREADS), lds, M->stream, g);
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

template <int KB>
static int dispatch2k(Mlp* M, const Fused2Args& g, int grid, bool actor_pass, int mt, int mta) {
  if (mt == 22 && mta == 2) return launch2<22, 2, KB>(M, g, grid, actor_pass);
  if (mt == 22 && mta == 1) return launch2<22, 1, KB>(M, g, grid, actor_pass);
  if (mt == 9 && mta == 2) return launch2<9, 2, KB>(M, g, grid, actor_pass);
  return launch2<9, 1, KB>(M, g, grid, actor_pass);
}
// the number of 8-row k-blocks is a compile-time constant (2, 5 or 6: a smaller net runs the next larger variant
// on zero-padded rows), so the MFMA chains carry no control flow
static int dispatch2(Mlp* M, const Fused2Args& g, int grid, bool actor_pass, int mt, int mta) {
  if (g.C.kb <= 2) return dispatch2k<2>(M, g, grid, actor_pass, mt, mta);
  if (g.C.kb <= 5) return dispatch2k<5>(M, g, grid, actor_pass, mt, mta);
  return dispatch2k<6>(M, g, grid, actor_pass, mt, mta);
}

static int launch_finish2(Mlp* M, Mlp* Mt, int nslab, int MT, int nR, double grad_scale, int mode, int Bu, int quirk,
                          void* loss_dev, const AdamPolyak* ap) {
  Finish2Args g{};
  g.slabs = M->fslab.as<float>(); g.nslab = nslab; g.K0 = M->dims[0]; g.H = M->dims[1]; g.MT = MT; g.nR = nR;
  g.grads = M->grads.as<float>(); g.scale = (float)grad_scale; g.mode = mode; g.Bu = Bu; g.quirk = quirk;
  g.loss_out = (float*)loss_dev;
  g.apply = ap != nullptr;
  if (ap) {
    int rcb = bp_begin(M, ap->b1, ap->b2, &g.bp);
    if (rcb) return rcb;
    g.p = M->params.as<float>(); g.m = M->m.as<float>(); g.v = M->v.as<float>();
    g.eta = ap->eta; g.b1 = ap->b1; g.b2 = ap->b2; g.eps = ap->eps;
    if (Mt && (float)ap->rho == 1.0f) Mt = nullptr;      // frozen targets (the reference as it runs): not touched, see mlp_mfma.hip launch_finish
    if (Mt) {
      g.pt = Mt->params.as<float>();
      const float r = (float)ap->rho;
      g.rho = r; g.omr = 1.0f - r;
    }
  }
  {
    ProfScope ps(M, ap ? "fused2_finish" : "fused2_reduce");
    hipLaunchKernelGGL(finish2_kernel, dim3(4 * slab2_tiles(MT, nR)), dim3(1024), 0, M->stream, g);
  }
  PDEC_HIP(hipGetLastError());
  if (ap) {
    bp_done(M);
    M->fw_dirty = true;
    if (Mt) Mt->fw_dirty = true;
  }
  return PDEC_OK;
}

// 2-layer fp32 actor [ns, h, 1] (relu, tanh | identity), h <= 31, ns <= 48, at least a few hundred columns
bool fused2_act_supported(const Mlp* A, int cols) {
  if (fused2_disabled() || A->dtype != PDEC_F32 || A->L != 2 || A->dims[2] != 1 || cols < 256) return false;
  if (A->acts[0] != PDEC_ACT_RELU || (A->acts[1] != PDEC_ACT_TANH && A->acts[1] != PDEC_ACT_IDENTITY)) return false;
  return A->dims[0] <= 8 * K2MAX && mt2_of(A->dims[1]) <= 2;
}

int fused2_policy_act(Mlp* A, const void* state, int cols, double act_noise, double act_limit, int learning, uint64_t seed,
                      uint64_t offset, void* actions_out, const uint64_t* ctr_cur, uint64_t* ctr_next, uint64_t ctr_inc) {
  const Net2 n = net2_of(A);
  const int mta = mt2_of(A->dims[1]), tanh_out = A->acts[1] == PDEC_ACT_TANH;
  const dim3 grid((cols + 63) / 64), block(256);
  ProfScope ps(A, "policy_act_fused");
#define ACT2(MTA, KB)                                                                                                       \
  hipLaunchKernelGGL((policy_act2_kernel<MTA, KB>), grid, block, (size_t)lds2_floats(16 * MTA, 8 * KB + 4) * 4, A->stream, n, \
                     (const float*)state, cols, (float)act_noise, (float)act_limit, learning, tanh_out, seed, offset,       \
                     (float*)actions_out, ctr_cur, ctr_next, ctr_inc)
  if (n.kb <= 2) { if (mta == 1) ACT2(1, 2); else ACT2(2, 2); }
  else if (n.kb <= 5) { if (mta == 1) ACT2(1, 5); else ACT2(2, 5); }
  else { if (mta == 1) ACT2(1, 6); else ACT2(2, 6); }
#undef ACT2
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

bool fused2_net_supported(const Mlp* M) { return !fused2_disabled() && M->dtype == PDEC_F32 && M->L == 2; }

int fused2_adam_polyak(Mlp* M, Mlp* Mt, const AdamPolyak& ap) {
  Finish2Args g{};
  g.grads = M->grads.as<float>();
  g.apply = 1;
  int rcb = bp_begin(M, ap.b1, ap.b2, &g.bp);
  if (rcb) return rcb;
  g.p = M->params.as<float>(); g.m = M->m.as<float>(); g.v = M->v.as<float>();
  g.eta = ap.eta; g.b1 = ap.b1; g.b2 = ap.b2; g.eps = ap.eps;
  if (Mt && (float)ap.rho == 1.0f) Mt = nullptr;        // frozen targets: not touched
  if (Mt) {
    g.pt = Mt->params.as<float>();
    const float r = (float)ap.rho;
    g.rho = r; g.omr = 1.0f - r;
  }
  {
    ProfScope ps(M, "fused2_apply");
    hipLaunchKernelGGL(apply2_kernel, dim3((M->nparams + 255) / 256), dim3(256), 0, M->stream, g, M->nparams);
  }
  PDEC_HIP(hipGetLastError());
  bp_done(M);
  M->fw_dirty = true;
  if (Mt) Mt->fw_dirty = true;
  return PDEC_OK;
}

int launch_rmean(Mlp* C, const float* r, int n, float** out) {
  float* rb = C->scratch.as<float>() + 40;      // device scalar behind the generic path's statistics and losses
  ProfScope ps(C, "ddpg_rmean");
  hipLaunchKernelGGL(rmean_kernel, dim3(1), dim3(1024), 0, C->stream, r, n, rb);
  PDEC_HIP(hipGetLastError());
  *out = rb;
  return PDEC_OK;
}

int fused2_critic_grads(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                        const void* sn, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev,
                        const AdamPolyak* apply) {
  const int mt = mt2_of(C->dims[1]), mta = mt2_of(A->dims[1]), nR = nr_of(C->dims[0]);
  int tpw = 8;
  const int grid = grid2_of(Bu, &tpw);
  const size_t need = slab2_floats(mt, nR, grid) * 4;
  if (C->fslab.bytes < need) PDEC_HIP(C->fslab.alloc(need));
  Fused2Args g{};
  g.C = net2_of(C); g.At = net2_of(At); g.Ct = net2_of(Ct); g.A = g.At;
  g.s = (const float*)s; g.a = (const float*)a; g.r = (const float*)r; g.t = (const float*)t; g.sn = (const float*)sn;
  g.Bu = Bu; g.ns = A->dims[0]; g.gamma = (float)gamma; g.quirk = quirk; g.tpw = tpw;
  g.slab = C->fslab.as<float>();
  if (quirk && C->rbar_ext) {          // reduced by the producer of r on its own stream (pdec_reward_mean)
    g.rbar = (const float*)C->rbar_ext;
  } else if (quirk) {
    float* rb = nullptr;
    int rcm = launch_rmean(C, (const float*)r, Bu, &rb);
    if (rcm) return rcm;
    g.rbar = rb;
  }
  C->rbar_ext = nullptr;
  int rc = dispatch2(C, g, grid, false, mt, mta);
  if (rc) return rc;
  return launch_finish2(C, apply ? Ct : nullptr, grid, mt, nR, grad_scale, 0, Bu, quirk, loss_dev, apply);
}

int fused2_actor_grads(Mlp* A, Mlp* C, Mlp* At, const void* s, int Bu, double grad_scale, void* loss_dev,
                       const AdamPolyak* apply) {
  const int mt = mt2_of(C->dims[1]), mta = mt2_of(A->dims[1]), nRa = nr_of(A->dims[0]);
  int tpw = 8;
  const int grid = grid2_of(Bu, &tpw);
  const size_t need = slab2_floats(mta, nRa, grid) * 4;
  if (A->fslab.bytes < need) PDEC_HIP(A->fslab.alloc(need));
  Fused2Args g{};
  g.C = net2_of(C); g.A = net2_of(A); g.At = g.A; g.Ct = g.C;
  g.s = (const float*)s; g.Bu = Bu; g.ns = A->dims[0]; g.tpw = tpw;
  g.slab = A->fslab.as<float>();
  int rc = dispatch2(C, g, grid, true, mt, mta);       // on the critic's stream object (shared stream, checked by the caller)
  if (rc) return rc;
  return launch_finish2(A, apply ? At : nullptr, grid, mta, nRa, grad_scale, 1, Bu, 0, loss_dev, apply);
}

}  // namespace pdec

// batch-mean reward as its own (fixed-order, one block) reduction on the caller's stream: the producer of r -- the env
// step -- reduces it beside the update, which then reads one scalar instead of every workgroup summing all of r
extern "C" int pdec_reward_mean(pdec_handle any_handle, const void* r, int n, void* mean_out) {
  using namespace pdec;
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_reward_mean: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(r && mean_out && n >= 1, "pdec_reward_mean: null/empty");
  ProfScope ps(o, "ddpg_rmean");
  hipLaunchKernelGGL(rmean_kernel, dim3(1), dim3(1024), 0, o->stream, (const float*)r, n, (float*)mean_out);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

extern "C" int pdec_ddpg_set_reward_partials(pdec_handle critic, const void* partial_sums, int n) {
  using namespace pdec;
  Mlp* C = lookup_as<Mlp>(critic, Kind::Mlp);
  if (!C) { set_error("pdec_ddpg_set_reward_partials: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(C->dtype == PDEC_F32 && n >= 0, "pdec_ddpg_set_reward_partials: fp32 critics only");
  PDEC_REQUIRE(partial_sums == nullptr || fused_net_supported(C), "pdec_ddpg_set_reward_partials: 3-layer fused critics only");
  C->rpart_ext = (const float*)partial_sums;
  C->rpart_n = n;
  return PDEC_OK;
}

extern "C" int pdec_ddpg_set_reward_mean(pdec_handle critic, const void* mean_dev) {
  using namespace pdec;
  Mlp* C = lookup_as<Mlp>(critic, Kind::Mlp);
  if (!C) { set_error("pdec_ddpg_set_reward_mean: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(C->dtype == PDEC_F32, "pdec_ddpg_set_reward_mean: fp32 critics only");
  C->rbar_ext = mean_dev;
  return PDEC_OK;
}
