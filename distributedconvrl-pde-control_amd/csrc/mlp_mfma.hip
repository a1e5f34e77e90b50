// mlp_mfma.hip -- fused fp32 MFMA path of the DDPG update for 3-layer actor/critic pairs
// (the "conv im2col-GEMM where channels x kernel forms a dense contraction": the critic's
// H x H hidden layer over B*A columns).  Restates src/PDEagent.jl:385-409 in two launches:
//
//   ddpg_critic_fused_kernel : a' = At(s'), qt = Ct([s';a']), q = C([s;a]), loss statistics,
//                              dq, full critic backward (dW1,db1,dW2,db2,dW3,db3) -> per-WG slab
//   ddpg_actor_fused_kernel  : a = A(s), q = C([s;a]) with the UPDATED critic, -mean(q),
//                              backward through the critic to da, full actor backward -> slab
//   fused_reduce_kernel      : deterministic sum of the per-workgroup slabs into the flat
//                              gradient buffer (replicas stay bit-identical), loss finalisation
//
// Design (gfx950): one workgroup = 8 waves = 128 columns, each wave owns 16 columns for the
// whole pass.  v_mfma_f32_16x16x4_f32 (exact f32, 64 FLOP/clk/SIMD = the f32 peak): a layer
// output tile D[16 rows][16 cols] lives in 4 VGPRs per lane (col = lane&15, row = 4*(lane>>4)+r)
// and is fed STRAIGHT back as the B operand of the next layer -- the contraction index is
// permuted (k = 16m + 4*(lane>>4) + r) so no lane movement or LDS round trip is needed; the
// matching A operand (4 consecutive k of one weight row) is one ds_read_b128 from the padded
// weight image in LDS.  Backward-to-input uses a pre-transposed weight image the same way.
// Weight gradients contract over COLUMNS, which sit on the wrong lane axis, so activations
// and dz are transposed once through LDS ([feature][col] images, 64-column halves) and the
// per-workgroup dW tiles are again MFMA products; bias gradients ride along as an appended
// row of ones.  HBM traffic: the 5 input arrays once, one slab per workgroup out.
#include "common.hpp"
#include "mlp.hpp"
#include "mfma_blocks.hpp"

namespace pdec {

#define LDWPAD 8           // weight-image row stride = HP + 8
#define KXP 16             // padded input rows (ns+na+1 <= 16)
#define LDW1 20            // LDS leading dim of W1 images

// padded weight image of one 3-layer net [K0, H, H, 1] (device, floats)
// Round 3: the image is laid out exactly as the passes keep it in LDS -- first the "small" block [W1 [HP][LDW1] | b1 [HP] |
// b2 [HP] | w3 [HP] | b3 [4]], then W2 [HP][LDW] and W2^T [HP][LDW], each block padded to a multiple of 256 floats -- so that
// every block reaches LDS by asynchronous LDS-DMA in whole 1-KiB pieces (dma_even): no register round trip, no remainder.
struct FNet {
  const float* w;   // base
  int K0, H, HP, LDW;
  int oW1, ob1, ob2, ow3, ob3, oW2, oW2T, total;
  int nsmall, nbig;   // padded block sizes (floats, multiples of 256)
};
__host__ __device__ constexpr int pad256(int n) { return (n + 255) / 256 * 256; }
__host__ __device__ constexpr int small_floats(int HP) { return pad256(HP * LDW1 + 3 * HP + 4); }
__host__ __device__ constexpr int big_floats(int HP) { return pad256(HP * (HP + LDWPAD)); }

static FNet make_fnet_layout(int K0, int H) {
  FNet f{};
  f.K0 = K0; f.H = H;
  f.HP = (H + 1 + 15) / 16 * 16;
  f.LDW = f.HP + LDWPAD;
  f.nsmall = small_floats(f.HP); f.nbig = big_floats(f.HP);
  f.oW1 = 0;
  f.ob1 = f.HP * LDW1;
  f.ob2 = f.ob1 + f.HP;
  f.ow3 = f.ob2 + f.HP;
  f.ob3 = f.ow3 + f.HP;
  f.oW2 = f.nsmall;
  f.oW2T = f.oW2 + f.nbig;
  f.total = f.oW2T + f.nbig;
  return f;
}

// builds the padded image from the internal flat parameters [W1 row-major, b1, W2, b2, W3, b3]
__global__ void prep_fused_kernel(const float* __restrict__ p, float* __restrict__ out, FNet f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f.total) return;
  const int K0 = f.K0, H = f.H;
  const int pW1 = 0, pb1 = H * K0, pW2 = pb1 + H, pb2 = pW2 + H * H, pW3 = pb2 + H, pb3 = pW3 + H;
  float v = 0.f;
  if (i < f.ob1) { const int r = i / LDW1, c = i % LDW1; if (r < H && c < K0) v = p[pW1 + r * K0 + c]; }
  else if (i < f.ob2) { const int r = i - f.ob1; if (r < H) v = p[pb1 + r]; }
  else if (i < f.ow3) { const int r = i - f.ob2; if (r < H) v = p[pb2 + r]; }
  else if (i < f.ob3) { const int r = i - f.ow3; if (r < H) v = p[pW3 + r]; }
  else if (i < f.oW2) { if (i == f.ob3) v = p[pb3]; }
  else if (i < f.oW2T) { const int r = (i - f.oW2) / f.LDW, c = (i - f.oW2) % f.LDW; if (r < H && c < H) v = p[pW2 + r * H + c]; }
  else { const int r = (i - f.oW2T) / f.LDW, c = (i - f.oW2T) % f.LDW; if (r < H && c < H) v = p[pW2 + c * H + r]; }
  out[i] = v;
}

// ------------------------------------------------------------------ device building blocks
// LDS image of the "small" part of a net: W1 [HP][LDW1], b1 [HP], b2 [HP], w3 [HP], b3 [4]
struct SmallLds {
  float *W1, *b1, *b2, *w3, *b3;
};
__device__ __forceinline__ SmallLds carve_small(float* base, int HP) {
  SmallLds s;
  s.W1 = base; s.b1 = s.W1 + HP * LDW1; s.b2 = s.b1 + HP; s.w3 = s.b2 + HP; s.b3 = s.w3 + HP;
  return s;
}
// floats of the big region: the padded W2 image (in whole DMA pieces), or the staging images that later overlay it
__host__ __device__ constexpr int wreg_floats(int MT, int MTA) {
  const int HP = 16 * MT, HPa = 16 * MTA;
  int a = big_floats(HP), b = 2 * HP * LDP, c = (32 + 4 * HPa) * 136;      // c: the actor's staging images, LDPA = 136
  int m = a > b ? a : b;
  return m > c ? m : c;
}

// Asynchronous global -> LDS copy of NFL floats (a multiple of 256) by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, no VGPR round trip).  EVERY one of the NW waves issues the same number of instructions, dma_count<NFL, NW>()
// -- a wave whose turn falls beyond the last piece copies the last piece again (same bytes, harmless) -- so that a wave can
// wait for an older copy while younger ones are still in flight with a literal s_waitcnt vmcnt(N) (dma_wait_but<N>).
// The data may be read after that wait + a workgroup barrier.
template <int NFL, int NW>
__host__ __device__ constexpr int dma_count() { return (NFL / 256 + NW - 1) / NW; }
template <int NFL, int NW>
__device__ __forceinline__ void dma_even(float* dst_lds, const float* src, int w, int l) {
  static_assert(NFL % 256 == 0 && NFL > 0, "dma_even copies whole 1-KiB pieces");
  constexpr int NCH = NFL / 256;
#pragma unroll
  for (int j = 0; j < dma_count<NFL, NW>(); ++j) {
    int c = w + NW * j;
    c = c < NCH ? c : NCH - 1;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)c * 256 + l * 4),
                                     (__attribute__((address_space(3))) void*)(dst_lds + (size_t)c * 256), 16, 0, 0);
  }
}
template <int N>
__device__ __forceinline__ void dma_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// h = relu(W1 x + b1): x given as KT register rows (lane holds x[4t+q][col])
template <int MT, int KT>
__device__ __forceinline__ void layer_in(f32x4 (&h)[MT], const float (&x)[KT], const SmallLds& s, int lr, int q) {
#pragma unroll
  for (int mo = 0; mo < MT; ++mo) {
    const float* b = s.b1 + 16 * mo + 4 * q;
    f32x4 acc = {b[0], b[1], b[2], b[3]};
#pragma unroll
    for (int t = 0; t < KT; ++t) acc = mfma4(s.W1[(16 * mo + lr) * LDW1 + 4 * t + q], x[t], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = fmaxf(acc[r], 0.f);
    h[mo] = acc;
  }
}

// out = W in (+ bias): W image row-major [..][ldw] in LDS, 4 consecutive k per ds_read_b128.
// Two output tiles are accumulated at a time with their MFMAs interleaved, so consecutive MFMAs of a wave are
// independent (issue interval 32 cycles instead of the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32).
// Tiles MO0 .. MO1-1 of the output only: a weight image may live in two LDS regions (rows below / above a split), each
// handed in with a base pointer W such that row r of the image is at W + r * ldw.
template <int MTO, int MTI, bool BIAS, int MO0, int MO1>
__device__ __forceinline__ void layer_hh_part(f32x4 (&out)[MTO], const f32x4 (&in)[MTI], const float* W, int ldw,
                                              const float* bias, int lr, int q) {
#ifndef PDEC_LAYER_PAIR
#define PDEC_LAYER_PAIR 1
#endif
#pragma unroll
  for (int mo = MO0; PDEC_LAYER_PAIR && mo + 1 < MO1; mo += 2) {
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
    if (BIAS) {
      const float* b = bias + 16 * mo + 4 * q;
      acc0 = f32x4{b[0], b[1], b[2], b[3]};
      acc1 = f32x4{b[16], b[17], b[18], b[19]};
    }
    const float* wrow0 = W + (16 * mo + lr) * ldw + 4 * q;
    const float* wrow1 = wrow0 + 16 * ldw;
    // operand reads run ONE block ahead of the MFMAs that consume them (explicit double buffer): left to itself the
    // scheduler issues each pair of ds_read_b128 right before its 8 MFMAs and the wave eats the LDS latency every time
    f32x4 wa = *reinterpret_cast<const f32x4*>(wrow0);
    f32x4 wb = *reinterpret_cast<const f32x4*>(wrow1);
#pragma unroll
    for (int m = 0; m < MTI; ++m) {
      f32x4 wan = wa, wbn = wb;
      if (m + 1 < MTI) {
        wan = *reinterpret_cast<const f32x4*>(wrow0 + 16 * (m + 1));
        wbn = *reinterpret_cast<const f32x4*>(wrow1 + 16 * (m + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      acc0 = mfma4(wa[0], in[m][0], acc0);
      acc1 = mfma4(wb[0], in[m][0], acc1);
      acc0 = mfma4(wa[1], in[m][1], acc0);
      acc1 = mfma4(wb[1], in[m][1], acc1);
      acc0 = mfma4(wa[2], in[m][2], acc0);
      acc1 = mfma4(wb[2], in[m][2], acc1);
      acc0 = mfma4(wa[3], in[m][3], acc0);
      acc1 = mfma4(wb[3], in[m][3], acc1);
      __builtin_amdgcn_sched_barrier(0);
      wa = wan;
      wb = wbn;
    }
    out[mo] = acc0;
    out[mo + 1] = acc1;
  }
#pragma unroll
  for (int mo = PDEC_LAYER_PAIR ? MO0 + ((MO1 - MO0) & ~1) : MO0; mo < MO1; ++mo) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (BIAS) {
      const float* b = bias + 16 * mo + 4 * q;
      acc = f32x4{b[0], b[1], b[2], b[3]};
    }
    const float* wrow = W + (16 * mo + lr) * ldw + 4 * q;
#pragma unroll
    for (int m = 0; m < MTI; ++m) {
      const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 16 * m);
      acc = mfma4(wv[0], in[m][0], acc);
      acc = mfma4(wv[1], in[m][1], acc);
      acc = mfma4(wv[2], in[m][2], acc);
      acc = mfma4(wv[3], in[m][3], acc);
    }
    out[mo] = acc;
  }
}
template <int MTO, int MTI, bool BIAS>
__device__ __forceinline__ void layer_hh(f32x4 (&out)[MTO], const f32x4 (&in)[MTI], const float* W, int ldw,
                                         const float* bias, int lr, int q) {
  layer_hh_part<MTO, MTI, BIAS, 0, MTO>(out, in, W, ldw, bias, lr, q);
}

struct FusedArgs {
  FNet A, C, At, Ct;          // A/At unused fields are ignored by the actor pass
  const float *s, *a, *r, *t, *sn;
  int Bu, ns, na;
  float gamma;
  int quirk;
  float* slab;                // chunk-major partial gradients, see store_pass
  const float* rpart;         // != null: per-workgroup reward sums of the producer, nrpart of them (sum / Bu = batch mean)
  int nrpart;
  const float* rbar_dev;      // != null: batch-mean reward already reduced by launch_rmean (batches beyond 32768 columns)
  int prio;                   // wave priority of the pass (PDEC_PRIO_MFMA, default 0)
  unsigned long long* stamps; // diagnostic only (PDEC_STAMPS=1): [gridDim.x][16] s_memtime at phase boundaries


};

// diagnostic phase stamp: wave 0 / lane 0 of each workgroup; a null pointer (the default) costs one scalar branch
#define STAMP(k)                                                                                       \
  do {                                                                                                 \
    if (g.stamps && threadIdx.x == 0) g.stamps[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)

// tiles per workgroup: dW3 [16][HP] (MT tiles) | dW2 [HP][HP] (MT*MT) | dW1 [HP][16] (MT); then one stats chunk
static inline int slab_tiles(int MT) { return MT + MT * MT + MT; }
static inline size_t slab_floats_total(int MT, int nslab) { return ((size_t)4 * slab_tiles(MT) + 1) * nslab * 64; }

// Workgroup barrier for LDS data only: every LDS operation of this wave is complete, then s_barrier.  Unlike __syncthreads()
// -- whose workgroup fence makes hipcc wait vmcnt(0), i.e. drain every LDS-DMA in flight -- it leaves vector-memory
// operations (the weight-image copies, slab stores) outstanding across the barrier.  The "memory" clobber keeps the
// compiler from moving LDS accesses across it.  Data a DMA wrote is visible after dma_wait*() + lds_barrier().
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// block-wide deterministic sum (the fixed tree of block_sum) on raw barriers
__device__ __forceinline__ float block_sum_lds(float v, float* red, int tid) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  lds_barrier();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  lds_barrier();
  float r = 0.f;
  for (int i = 0; i < FTHREADS / 64; ++i) r += red[i];
  lds_barrier();
  return r;
}

// ------------------------------------------------------------------ critic pass
// LDS map (floats): Wreg [wreg_floats] | sc [small(HP)] target critic's small block | sc2 [small(HP)] behaviour critic's |
// sa [small(HPa)] | saW2 [big(HPa)] | red [8 + 8 HP + 64].  Round 3: every weight block arrives by LDS-DMA in whole pieces
// (dma_even) -- round 2 loaded the small blocks through registers, 3.5 k cycles per load_small with nothing to overlap --
// and the behaviour critic's small block is copied during the target phase into its own region, so the phase switch is one
// barrier + the issue of the next big copy.  Copies are waited for in issue order with literal vmcnt counts.
template <int MT, int MTA>
__global__ __launch_bounds__(FTHREADS) __attribute__((amdgpu_num_vgpr(112))) void ddpg_critic_fused_kernel(FusedArgs g) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  constexpr int HP = 16 * MT, HPa = 16 * MTA, LDW = HP + LDWPAD, LDWa = HPa + LDWPAD, NW = FTHREADS / 64;
  constexpr int NS = small_floats(HP), NSa = small_floats(HPa), NB = big_floats(HP), NBa = big_floats(HPa);
  float* Wreg = smem;                                   // [HP][LDW] big weight image / staging images
  float* sc = Wreg + wreg_floats(MT, MTA);              // target critic's small block
  float* sc2 = sc + NS;                                 // behaviour critic's small block
  float* sa = sc2 + NS;                                 // target actor's small block
  float* saW2 = sa + NSa;                               // target actor's W2 [HPa][LDWa]
  float* red = saW2 + NBa;                              // [8] block sums | [8][HP] pass A | [8][8] loss statistics

  const SmallLds SCt = carve_small(sc, HP), SC = carve_small(sc2, HP), SA = carve_small(sa, HPa);
  const int ns = g.ns, na = g.na, K0 = ns + na;
  const int col = blockIdx.x * FCOLS + w * 16 + lr;
  const bool valid = col < g.Bu;
  set_wave_prio(g.prio);


  // ---- phase T: target actor + target critic
  if (g.stamps && threadIdx.x == 0) g.stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memrealtime();
  STAMP(0);
  // Copy schedule.  hipcc drains every LDS-DMA in flight (vmcnt(0)) at the first use of an ordinary global load and at every
  // __syncthreads(), so: the small blocks of the target nets are copied first, the per-column inputs are loaded and CONSUMED
  // (that wait covers the small copies, which are needed now anyway), and only then the long copies start -- the target
  // critic's W2 and the behaviour critic's small block -- which stay in flight across the raw barriers (lds_barrier) of the
  // target actor and layer 1.  No ordinary global load is used again before the last copy of the pass has been waited for.
  dma_even<NS, NW>(sc, g.Ct.w, w, l);
  dma_even<NSa, NW>(sa, g.At.w, w, l);
  dma_even<NBa, NW>(saW2, g.At.w + g.At.oW2, w, l);
  float xn[4], xq[4];                       // input rows 4t+q of s' and of [s; a]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 4 * t + q;
    xn[t] = (valid && row < ns) ? g.sn[(size_t)col * ns + row] : 0.f;
    float v = 0.f;
    if (valid && row < ns) v = g.s[(size_t)col * ns + row];
    else if (valid && row < K0) v = g.a[(size_t)col * na + (row - ns)];
    xq[t] = v;
  }
  float rv = valid ? g.r[col] : 0.f;
  float tv = valid ? g.t[col] : 0.f;
  // mean reward for the reference's (1xBu).+(Bu) broadcast: every workgroup reduces all of r in the
  // same fixed order, so the value is identical everywhere
  float rsum = 0.f;
  if (g.quirk && g.rpart) {
    for (int i = tid; i < g.nrpart; i += FTHREADS) rsum += g.rpart[i];
  } else if (g.quirk && !g.rbar_dev) {   // 16-B loads, 8 in flight per lane: the L2 latency is paid per batch, not per element
    const int n4 = ((reinterpret_cast<uintptr_t>(g.r) & 15) == 0) ? g.Bu / 4 : 0;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(g.r);
    int i = tid;
    for (; i + 7 * FTHREADS < n4; i += 8 * FTHREADS) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = r4[i + u * FTHREADS];
#pragma unroll
      for (int u = 0; u < 8; ++u) rsum += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    for (; i < n4; i += FTHREADS) { const f32x4 v = r4[i]; rsum += (v[0] + v[1]) + (v[2] + v[3]); }
    for (int k = 4 * n4 + tid; k < g.Bu; k += FTHREADS) rsum += g.r[k];
  }
  float rbar_in = g.rbar_dev ? g.rbar_dev[0] : 0.f;
  // every loaded value is consumed HERE (the compiler's wait for them lands in front of this statement)
  asm volatile("" : "+v"(xn[0]), "+v"(xn[1]), "+v"(xn[2]), "+v"(xn[3]), "+v"(xq[0]), "+v"(xq[1]), "+v"(xq[2]), "+v"(xq[3]),
               "+v"(rv), "+v"(tv), "+v"(rsum), "+v"(rbar_in));
  dma_wait();                                             // (the small copies; nothing else is outstanding)
  dma_even<NB, NW>(Wreg, g.Ct.w + g.Ct.oW2, w, l);
  dma_even<NS, NW>(sc2, g.C.w, w, l);
  constexpr int N_SC = dma_count<NS, NW>();
  float rbar = block_sum_lds(rsum, red, tid) / (float)g.Bu;   // raw barriers inside (small blocks visible afterwards)
  if (g.rbar_dev) rbar = rbar_in;
  STAMP(1);

  float x[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = xn[t];
  float tgt;
  {
    f32x4 ha1[MTA], ha2[MTA];
    layer_in<MTA, 4>(ha1, x, SA, lr, q);
    layer_hh<MTA, MTA, true>(ha2, ha1, saW2, LDWa, SA.b2, lr, q);
    relu_<MTA>(ha2);
    const float an = tanhf(head<MTA>(ha2, SA.w3, SA.b3[0], q));      // na == 1
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (4 * t + q == ns) x[t] = valid ? an : 0.f;
    f32x4 h1[MT], h2[MT];
    layer_in<MT, 4>(h1, x, SCt, lr, q);
    dma_wait_but<N_SC>();                                 // the target critic's W2 has landed
    lds_barrier();
    layer_hh<MT, MT, true>(h2, h1, Wreg, LDW, SCt.b2, lr, q);
    relu_<MT>(h2);
    const float qt = head<MT>(h2, SCt.w3, SCt.b3[0], q);
    tgt = g.gamma * (1.f - tv) * qt;
  }
  dma_wait();                                             // the behaviour critic's small block (issued a phase ago)
  lds_barrier();                                        // every wave is done with Wreg; sc2 visible
  STAMP(2);
  // ---- phase Q: behaviour critic forward; its W2 lands behind layer 1
  dma_even<NB, NW>(Wreg, g.C.w + g.C.oW2, w, l);
  STAMP(3);
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = xq[t];
  f32x4 h1[MT], h2[MT];
  layer_in<MT, 4>(h1, x, SC, lr, q);
  dma_wait();
  lds_barrier();
  layer_hh<MT, MT, true>(h2, h1, Wreg, LDW, SC.b2, lr, q);
  relu_<MT>(h2);
  const float qv = head<MT>(h2, SC.w3, SC.b3[0], q);
  const float c = valid ? tgt - qv : 0.f;
  const float dq = valid ? -(2.f / (float)g.Bu) * ((g.quirk ? rbar : rv) + c) : 0.f;
  // loss statistics (one lane per column contributes)
  const bool rep = valid && q == 0;
  float sv[5] = {rep ? c : 0.f, rep ? c * c : 0.f, rep ? rv : 0.f, rep ? rv * rv : 0.f, rep ? (rv + c) * (rv + c) : 0.f};
  const int nslab = gridDim.x;
  float* Lm = Wreg;                 // staging images overlay the big weight region
  float* Rm = Wreg + HP * LDP;
  const int cw = (w & 3) * 16 + lr;

  // ---- pass A: dW3/db3 = dq x [h2; 1]^T.  A single output row: reduce dq*h2 over the 16 columns of each wave with DPP
  // row sums, then over the 8 waves through LDS (fixed order -> deterministic); no MFMA/staging round.  Round 3: dz2 is taken
  // from h2 first and the 36 row sums then overwrite h2 in place with no branch between them (round 2 interleaved each
  // sum with its lane-0 store under an exec mask: 36 serial dependent chains, ~8 k cycles); the five loss statistics ride on
  // the same barrier instead of a block reduction of their own at the end of the kernel.
  lds_barrier();
  STAMP(4);
  dma_even<NB, NW>(Wreg, g.C.w + g.C.oW2T, w, l);     // W2^T for the backward pass lands behind pass A
  f32x4 dz2[MT];
  head_bwd<MT>(dz2, h2, SC.w3, dq, q);
  float a_keep = 0.f;                 // this thread's element of dW3 / of the statistics (stored behind the next barrier)
  {
    float* redA = red + 8;            // [8][HP]
    float* red5 = redA + 8 * HP;      // [8][8]
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = 16 * m + 4 * q + r;
        h2[m][r] = row_sum16((row == g.C.H ? 1.f : h2[m][r]) * dq);
      }
#pragma unroll
    for (int k = 0; k < 5; ++k)
      for (int off = 32; off > 0; off >>= 1) sv[k] += __shfl_xor(sv[k], off);
    if (lr == 0) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) redA[w * HP + 16 * m + 4 * q + r] = h2[m][r];
    }
    if (l == 0)
      for (int k = 0; k < 5; ++k) red5[w * 8 + k] = sv[k];
    lds_barrier();
    if (tid < HP) {
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) a_keep += redA[ww * HP + tid];
    } else if (tid >= FTHREADS - 8) {          // the last eight lanes of the last wave: the statistics chunk
      if (tid - (FTHREADS - 8) < 5)
        for (int ww = 0; ww < NW; ++ww) a_keep += red5[ww * 8 + tid - (FTHREADS - 8)];
    }
  }
  // ---- dh1 = W2^T dz2, dz1
  STAMP(5);
  dma_wait();                         // W2^T (no store is outstanding yet: the slab stores of pass A follow the barrier)
  lds_barrier();
  STAMP(6);
  if (tid < HP)                       // slab position of element (row 0, column tid) of the [16][HP] product: tile tid/16, register 0, lane tid%16
    g.slab[((size_t)(4 * (tid >> 4)) * nslab + blockIdx.x) * 64 + (tid & 15)] = a_keep;
  else if (tid >= FTHREADS - 8)
    g.slab[((size_t)(4 * (2 * MT + MT * MT)) * nslab + blockIdx.x) * 64 + (tid - (FTHREADS - 8))] = a_keep;
  f32x4 dz1[MT];
  layer_hh<MT, MT, false>(dz1, dz2, Wreg, LDW, nullptr, lr, q);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) dz1[m][r] = h1[m][r] > 0.f ? dz1[m][r] : 0.f;
  // ---- pass C first: dW1/db1 = dz1 x [x0; 1]^T in ONE staging round over all 128 columns (dz1 [HP][LDP128] +
  // x [16][LDP128] fill the big region exactly); dz1 is dead afterwards, which leaves pass B the registers for two
  // interleaved accumulation chains per wave
  STAMP(7);
  {
    constexpr int LDP128 = 136;                 // 8 mod 16 floats, like LDP: conflict-free ds_read_b128 operand reads
    static_assert(16 * MT * LDP128 + 16 * LDP128 <= wreg_floats(MT, MTA), "pass C images do not fit the big LDS region");
    float* Lc = Wreg;
    float* Rc = Wreg + HP * LDP128;
    const int cw128 = w * 16 + lr;
    f32x4 accC[((MT + 7) / 8 + 1) & ~1];
    zero_(accC);
    lds_barrier();
    stage_rows_ld<MT>(Lc, LDP128, dz1, cw128, q, -1);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int row = 4 * t + q;
      Rc[row * LDP128 + cw128] = row == K0 ? 1.f : x[t];
    }
    lds_barrier();
    gemm_pass_paired<((MT + 7) / 8 + 1) & ~1, 8>(accC, Lc, Rc, LDP128, MT, 1, w, lr, q);
    store_pass(accC, g.slab, nslab, MT + MT * MT, MT, 1, w, l);
  }
  // ---- pass B: dW2/db2 = dz2 x [h1; 1]^T, two 64-column halves
  STAMP(8);
  {
    f32x4 accB[(MT * MT + 7) / 8];
    zero_(accB);
    lds_barrier();
    if ((w >> 2) == 0) {
      stage_rows<MT>(Lm, dz2, cw, q, -1);
      stage_rows<MT>(Rm, h1, cw, q, g.C.H);
    }
    lds_barrier();
    gemm_pass_paired<(MT * MT + 7) / 8, 4>(accB, Lm, Rm, LDP, MT, MT, w, lr, q);
    lds_barrier();
    if ((w >> 2) == 1) {
      stage_rows<MT>(Lm, dz2, cw, q, -1);
      stage_rows<MT>(Rm, h1, cw, q, g.C.H);
    }
    lds_barrier();
    gemm_pass_paired_store<(MT * MT + 7) / 8, 4>(accB, Lm, Rm, LDP, MT, MT, w, lr, q, g.slab, nslab, MT, l);     // stores behind each tile pair
  }
  STAMP(9);
  STAMP(10);
  // slots 11 / 12: the constant 100 MHz counter at the start / end of the workgroup (shader clock = d s_memtime / d s_memrealtime x 100 MHz)
  if (g.stamps && threadIdx.x == 0) g.stamps[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memrealtime();
}

// ------------------------------------------------------------------ actor pass
// LDS map: Wreg | sc (critic's small block) | sa | saW2 | saW2T | red [8] | X [32][LDW] (MT = 9).  The backward pass needs
// W2^T where the forward had W2: its first 32 rows (two output tiles) are copied into X during the forward, the rest into
// the big region at the switch, behind those two tiles -- round 2 waited for the whole 87.5 KB copy with nothing to do.
#define AX_ROWS 32
template <int MT, int MTA>
__global__ __launch_bounds__(FTHREADS) __attribute__((amdgpu_num_vgpr(112))) void ddpg_actor_fused_kernel(FusedArgs g) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  constexpr int HP = 16 * MT, HPa = 16 * MTA, LDW = HP + LDWPAD, LDWa = HPa + LDWPAD, NW = FTHREADS / 64;
  constexpr int NS = small_floats(HP), NSa = small_floats(HPa), NB = big_floats(HP), NBa = big_floats(HPa);
  constexpr bool TWO = HP > AX_ROWS && (AX_ROWS * LDW) % 256 == 0;     // W2^T in two parts (X + big region)
  constexpr int NX = TWO ? AX_ROWS * LDW : 256;
  float* Wreg = smem;
  float* sc = Wreg + wreg_floats(MT, MTA);
  float* sa = sc + NS;
  float* saW2 = sa + NSa;
  float* saW2T = saW2 + NBa;
  float* red = saW2T + NBa;
  float* X = red + 8;
  const SmallLds SC = carve_small(sc, HP), SA = carve_small(sa, HPa);
  const int ns = g.ns;
  const int col = blockIdx.x * FCOLS + w * 16 + lr;
  const bool valid = col < g.Bu;
  set_wave_prio(g.prio);

  // copy schedule as in the critic pass: small blocks, then the per-column input loaded AND consumed, then the long copy
  dma_even<NSa, NW>(sa, g.A.w, w, l);
  dma_even<NBa, NW>(saW2, g.A.w + g.A.oW2, w, l);
  dma_even<NBa, NW>(saW2T, g.A.w + g.A.oW2T, w, l);
  dma_even<NS, NW>(sc, g.C.w, w, l);
  float xs[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 4 * t + q;
    xs[t] = (valid && row < ns) ? g.s[(size_t)col * ns + row] : 0.f;
  }
  asm volatile("" : "+v"(xs[0]), "+v"(xs[1]), "+v"(xs[2]), "+v"(xs[3]));
  dma_wait();
  dma_even<NB, NW>(Wreg, g.C.w + g.C.oW2, w, l);      // lands behind the actor forward and the critic's first layer
  lds_barrier();
  f32x4 ha1[MTA], ha2[MTA];
  layer_in<MTA, 4>(ha1, xs, SA, lr, q);
  layer_hh<MTA, MTA, true>(ha2, ha1, saW2, LDWa, SA.b2, lr, q);
  relu_<MTA>(ha2);
  const float aout = tanhf(head<MTA>(ha2, SA.w3, SA.b3[0], q));
  float x[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) x[t] = (4 * t + q == ns) ? (valid ? aout : 0.f) : xs[t];
  f32x4 h1[MT], h2[MT];
  layer_in<MT, 4>(h1, x, SC, lr, q);
  unsigned long long relu1 = 0;         // bit 4 m + r = [h1[m][r] > 0]: all the backward pass needs of h1
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) relu1 |= (unsigned long long)(h1[m][r] > 0.f) << (4 * m + r);
  dma_wait();
  lds_barrier();
  if constexpr (TWO) dma_even<NX, NW>(X, g.C.w + g.C.oW2T, w, l);      // rows 0 .. 31 of W2^T, behind the forward
  layer_hh<MT, MT, true>(h2, h1, Wreg, LDW, SC.b2, lr, q);
  relu_<MT>(h2);
  const float qv = head<MT>(h2, SC.w3, SC.b3[0], q);
  float st0 = (valid && q == 0) ? qv : 0.f;
  const float dq = valid ? -1.f / (float)g.Bu : 0.f;
  f32x4 dz2[MT];
  head_bwd<MT>(dz2, h2, SC.w3, dq, q);
  f32x4 dz1[MT];
  if constexpr (TWO) {
    dma_wait();
    lds_barrier();                                       // X visible; every wave is done with the big region
    dma_even<NB - NX, NW>(Wreg + NX, g.C.w + g.C.oW2T + NX, w, l);       // rows 32 .. of W2^T behind the first two tiles
    layer_hh_part<MT, MT, false, 0, AX_ROWS / 16>(dz1, dz2, X, LDW, nullptr, lr, q);
    dma_wait();
    lds_barrier();
    layer_hh_part<MT, MT, false, AX_ROWS / 16, MT>(dz1, dz2, Wreg, LDW, nullptr, lr, q);
  } else {
    lds_barrier();
    dma_even<NB, NW>(Wreg, g.C.w + g.C.oW2T, w, l);
    dma_wait();
    lds_barrier();
    layer_hh<MT, MT, false>(dz1, dz2, Wreg, LDW, nullptr, lr, q);
  }
  // da = sum_i W1[i][ns] * dz1[i]  (gradient w.r.t. the action input row of the critic)
  float da = 0.f;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * m + 4 * q + r;
      const float d = ((relu1 >> (4 * m + r)) & 1) ? dz1[m][r] : 0.f;
      da += SC.W1[row * LDW1 + ns] * d;
    }
  da += __shfl_xor(da, 16);
  da += __shfl_xor(da, 32);
  const float dza3 = da * (1.f - aout * aout);
  f32x4 dza2[MTA], dza1[MTA];
  head_bwd<MTA>(dza2, ha2, SA.w3, dza3, q);
  layer_hh<MTA, MTA, false>(dza1, dza2, saW2T, LDWa, nullptr, lr, q);
#pragma unroll
  for (int m = 0; m < MTA; ++m)
#pragma unroll
    for (int r = 0; r < 4; ++r) dza1[m][r] = ha1[m][r] > 0.f ? dza1[m][r] : 0.f;
  // ---- weight gradients of the actor: three small column contractions over all 128 columns in ONE staging round, one output
  // tile per wave (2 + 4 + 2 tiles at MTA = 2).  (Until round 3: two 64-column halves, the tiles of each product handed to waves
  // 0, 1, ..: waves 0 and 1 carried three dependent MFMA chains per half while waves 4 - 7 idled -- 7.7 k cycles of the pass.)
  // Columns are contracted in the same order as before, so the sums are bit-identical.
  const int nslab = gridDim.x;
  constexpr int LDPA = 136;               // 8 mod 16 floats: conflict-free ds_read_b128 operand reads
  static_assert((32 + 4 * HPa) * LDPA <= wreg_floats(MT, MTA), "the actor's staging images do not fit the big LDS region");
  static_assert(2 * MTA + MTA * MTA <= FTHREADS / 64, "one output tile per wave");
  float* I0 = Wreg;                       // DZ3 [16][LDPA]
  float* I1 = I0 + 16 * LDPA;             // HA2aug [HPa][LDPA]
  float* I2 = I1 + HPa * LDPA;            // DZA2 [HPa][LDPA]
  float* I3 = I2 + HPa * LDPA;            // HA1aug [HPa][LDPA]
  float* I4 = I3 + HPa * LDPA;            // DZA1 [HPa][LDPA]
  float* I5 = I4 + HPa * LDPA;            // Xaug [16][LDPA]
  const int cw = w * 16 + lr;
  lds_barrier();
  for (int i = tid; i < 16 * LDPA; i += FTHREADS) { I0[i] = 0.f; I5[i] = 0.f; }
  lds_barrier();
  if (q == 0) I0[cw] = dza3;
  stage_rows_ld<MTA>(I1, LDPA, ha2, cw, q, g.A.H);
  stage_rows_ld<MTA>(I2, LDPA, dza2, cw, q, -1);
  stage_rows_ld<MTA>(I3, LDPA, ha1, cw, q, g.A.H);
  stage_rows_ld<MTA>(I4, LDPA, dza1, cw, q, -1);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 4 * t + q;
    I5[row * LDPA + cw] = row == ns ? 1.f : xs[t];
  }
  lds_barrier();
  {
    // tile numbering of the slab (store_pass): dW3a tiles 0 .. MTA-1, dW2a MTA .. MTA+MTA^2-1, dW1a the next MTA
    const float *Lp = nullptr, *Rp = nullptr;
    int T = -1;
    if (w < MTA) { Lp = I0; Rp = I1 + 16 * w * LDPA; T = w; }
    else if (w < MTA + MTA * MTA) { const int u = w - MTA, ti = u / MTA, tk = u - ti * MTA; Lp = I2 + 16 * ti * LDPA; Rp = I3 + 16 * tk * LDPA; T = w; }
    else if (w < 2 * MTA + MTA * MTA) { const int u = w - MTA - MTA * MTA; Lp = I4 + 16 * u * LDPA; Rp = I5; T = w; }
    if (T >= 0) {
      const float* lrow = Lp + lr * LDPA + 4 * q;
      const float* rrow = Rp + lr * LDPA + 4 * q;
      f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f32x4 av = *reinterpret_cast<const f32x4*>(lrow + 16 * t);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(rrow + 16 * t);
        a = mfma4(av[0], bv[0], a);
        a = mfma4(av[1], bv[1], a);
        a = mfma4(av[2], bv[2], a);
        a = mfma4(av[3], bv[3], a);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
        PDEC_SLAB_STORE(a[r], &g.slab[((size_t)(4 * T + r) * nslab + blockIdx.x) * 64 + l]);
    }
  }
  st0 = block_sum_lds(st0, red, tid);
  if (tid == 0) {
    float* st = g.slab + ((size_t)(4 * (2 * MTA + MTA * MTA)) * nslab + blockIdx.x) * 64;
    st[0] = st0;
    for (int i = 1; i < 8; ++i) st[i] = 0.f;
  }
}

// ------------------------------------------------------------------ slab reduction + ADAM + Polyak + image refresh
// flat internal gradient layout [W1 [H][K0], b1 [H], W2 [H][H], b2 [H], W3 [1][H], b3]
struct FinishArgs {
  const float* slabs;          // chunk-major per-workgroup partial gradients (store_pass)
  int nslab, K0, H, MT;
  float* grads;                // flat gradient buffer (written when nslab > 0, read when nslab == 0)
  float scale;
  int mode, Bu, quirk;         // loss finalisation (mode 0 critic, 1 actor)
  float* loss_out;
  int apply;                   // != 0: ADAM step on p, Polyak into pt, refresh the padded images
  float *p, *m, *v, *pt, *fw, *fwt;
  float* fwp;                  // published copy of the updated image (double-buffered: a concurrent acting kernel
                               // keeps reading the other copy), may be null
  FNet lay;
  double eta, b1, b2, eps, omb1p, omb2p;   // omb*p = 1 - beta^t: filled in by the kernel from `bp` (device resident)
  BpArgs bp;
  float rho, omr;
  int prio;                    // wave priority (the update chain is the critical path: same level as the passes)
};

// Flux.Optimise.ADAM (Float64 arithmetic, as the broadcast promotes; src/custom_nna.jl:23-24), then
// dest = rho dest + (1-rho) src (src/PDEagent.jl:415-417) for this one parameter, and its copies in the
// padded LDS images the fused passes stage from (so no separate prep launch is needed)
__device__ __forceinline__ void finish_param(const FinishArgs& g, int i, float gi, float p0, float m0, float v0, float pt0) {
  // no FMA contraction: Julia evaluates these broadcasts with separate multiplies and adds, and the two call
  // sites of this function (fused reduce+apply, apply after an all-reduce) must round identically
#pragma clang fp contract(off)
  const double gd = (double)gi;
  const float mt = (float)(g.b1 * (double)m0 + (1.0 - g.b1) * gd);
  const float vt = (float)(g.b2 * (double)v0 + (1.0 - g.b2) * gd * gd);
  g.m[i] = mt;
  g.v[i] = vt;
  const float delta = (float)((double)mt / g.omb1p / (sqrt((double)vt / g.omb2p) + g.eps) * g.eta);
  const float pn = p0 - delta;
  g.p[i] = pn;
  const int K0 = g.K0, H = g.H;
  int o1, o2 = -1, j = i;
  if (j < H * K0) o1 = g.lay.oW1 + (j / K0) * LDW1 + (j % K0);
  else if ((j -= H * K0) < H) o1 = g.lay.ob1 + j;
  else if ((j -= H) < H * H) { o1 = g.lay.oW2 + (j / H) * g.lay.LDW + (j % H); o2 = g.lay.oW2T + (j % H) * g.lay.LDW + (j / H); }
  else if ((j -= H * H) < H) o1 = g.lay.ob2 + j;
  else if ((j -= H) < H) o1 = g.lay.ow3 + j;
  else o1 = g.lay.ob3;
  if (g.fw) { g.fw[o1] = pn; if (o2 >= 0) g.fw[o2] = pn; }
  if (g.fwp) { g.fwp[o1] = pn; if (o2 >= 0) g.fwp[o2] = pn; }
  if (g.pt) {
    const float tn = g.rho * pt0 + g.omr * pn;
    g.pt[i] = tn;
    if (g.fwt) {
      g.fwt[o1] = tn; if (o2 >= 0) g.fwt[o2] = tn;
    }
  }
}

// ROUND-2 FORM, kept as the bit-identity reference of fused_finish_kernel below (PDEC_FINISH_REF=1 selects it; tests only).
// grid: nslab > 0 -> one block per chunk (4 * slab_tiles(MT)); nslab == 0 -> ceil(n / 64) blocks (apply from grads).
// block = 64 chunk elements x 16 slab groups (1024 threads): each thread sums every 16th slab (16 loads in flight,
// a contiguous 256-B row each), then a fixed-order combine over the groups -> deterministic, so data-parallel
// replicas stay bit-identical.
__global__ __launch_bounds__(1024) void fused_finish_ref_kernel(FinishArgs g_in) {
  FinishArgs g = g_in;
  set_wave_prio(g.prio);
  if (g.apply) {      // the beta powers come from device memory; one thread of the grid writes the advanced pair
    g.omb1p = 1.0 - g.bp.cur[0];
    g.omb2p = 1.0 - g.bp.cur[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_advance(g.bp, g.b1, g.b2);
  }
  __shared__ float part[16][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int K0 = g.K0, H = g.H, MT = g.MT;
  const int n = H * K0 + H + H * H + H + H + 1;
  const int offb1 = H * K0, offW2 = offb1 + H, offb2 = offW2 + H * H, offW3 = offb2 + H, offb3 = offW3 + H;
  if (g.nslab > 0) {
    const int c = blockIdx.x, T = c >> 2, r = c & 3;
    int i = -1;   // flat parameter index of this chunk element (-1: padding)
    {
      const int q = tx >> 4, lr = tx & 15;
      if (T < MT) {                                  // dW3 / db3: only row 0 of the [16][HP] product is real
        const int row = 4 * q + r, col = 16 * T + lr;
        if (row == 0) i = col < H ? offW3 + col : (col == H ? offb3 : -1);
      } else if (T < MT + MT * MT) {                 // dW2 / db2
        const int u = T - MT, ti = u / MT, tk = u - ti * MT;
        const int row = 16 * ti + 4 * q + r, col = 16 * tk + lr;
        if (row < H) i = col < H ? offW2 + row * H + col : (col == H ? offb2 + row : -1);
      } else {                                       // dW1 / db1
        const int ti = T - MT - MT * MT;
        const int row = 16 * ti + 4 * q + r, col = lr;
        if (row < H) i = col < K0 ? row * K0 + col : (col == K0 ? offb1 + row : -1);
      }
    }
    if (__syncthreads_or(i >= 0)) {
      float p0 = 0.f, m0 = 0.f, v0 = 0.f, pt0 = 0.f;
      if (g.apply && ty == 0 && i >= 0) {            // issue the parameter loads before the slab stream
        p0 = g.p[i]; m0 = g.m[i]; v0 = g.v[i];
        if (g.pt) pt0 = g.pt[i];
      }
      float acc = 0.f;
      if (i >= 0) {
        const float* sp = g.slabs + (size_t)c * g.nslab * 64 + tx;
        int z = ty;
        for (; z + 240 < g.nslab; z += 256) {
          float v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = PDEC_SLAB_LOAD(&sp[(size_t)(z + 16 * u) * 64]);
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
        if (g.apply) finish_param(g, i, a, p0, m0, v0, pt0);
      }
    }
  } else {
    const int i = blockIdx.x * 64 + tx;
    if (ty == 0 && i < n && g.apply) finish_param(g, i, g.grads[i], g.p[i], g.m[i], g.v[i], g.pt ? g.pt[i] : 0.f);
  }
  if (blockIdx.x == 0 && g.loss_out && g.nslab > 0) {   // block 0 also finalises the loss (fixed-order tree -> deterministic)
    __shared__ double red[5][256];
    const int tid = threadIdx.x;
    const float* stc = g.slabs + (size_t)(4 * (2 * MT + MT * MT)) * g.nslab * 64;   // stats chunk
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
      if (g.mode == 0)   // critic: quirk -> mean(c^2) + 2 mean(c) mean(r) + mean(r^2); else mean((r+c)^2)
        *g.loss_out = (float)(g.quirk ? red[1][0] * inv + 2.0 * (red[0][0] * inv) * (red[2][0] * inv) + red[3][0] * inv
                                      : red[4][0] * inv);
      else               // actor: -mean(q)
        *g.loss_out = (float)(-red[0][0] * inv);
    }
  }
}

// Slab reduction + parameter update, round-3 decomposition.  The summation tree of every gradient element is the one of
// the round-2 kernel above -- 16 partial sums over the slabs z = k, k + 16, k + 32, ... in that order, combined in the
// order k = 0 .. 15 -- so parameters stay bit-identical; what changed is who adds what:
//   * one block per HALF chunk (32 elements x nslab slabs = 128 contiguous bytes per slab row) instead of per chunk:
//     the critic's 369 live chunks (64 KB each) over 256 CUs put two whole chunks on 140 CUs and one on the rest, and a
//     CU streams at a fixed ~10 B/clk, so the launch lasted as long as 128 KB on one CU; 738 halves leave at most 96 KB;
//   * 128 threads per block = 16 slab groups x 8 lanes, every lane loads 16 B (four elements of a slab row) and keeps
//     16 such loads in flight: a quarter of the load instructions for the same bytes;
//   * all blocks are co-resident (792 blocks x 2 waves = 6 waves per CU), so no second round of blocks waits for the first.
// grid: nslab > 0 -> 8 * slab_tiles(MT) blocks; nslab == 0 -> ceil(n / 128) blocks (apply from the gradient buffer).
#define FIN_THREADS 128
__global__ __launch_bounds__(FIN_THREADS) void fused_finish_kernel(FinishArgs g_in) {
  FinishArgs g = g_in;
  set_wave_prio(g.prio);
  if (g.apply) {      // the beta powers come from device memory; one thread of the grid writes the advanced pair
    g.omb1p = 1.0 - g.bp.cur[0];
    g.omb2p = 1.0 - g.bp.cur[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) bp_advance(g.bp, g.b1, g.b2);
  }
  __shared__ float part[16][36];
  const int tid = threadIdx.x;
  const int K0 = g.K0, H = g.H, MT = g.MT;
  const int n = H * K0 + H + H * H + H + H + 1;
  const int offb1 = H * K0, offW2 = offb1 + H, offb2 = offW2 + H * H, offW3 = offb2 + H, offb3 = offW3 + H;
  if (g.nslab > 0) {
    const int c = blockIdx.x >> 1, half = blockIdx.x & 1, T = c >> 2, r = c & 3;
    int i = -1;   // flat parameter index of the chunk element this thread finishes (threads 0..31; -1: padding)
    if (tid < 32) {
      const int l = 32 * half + tid, q = l >> 4, lr = l & 15;
      if (T < MT) {                                  // dW3 / db3: only row 0 of the [16][HP] product is real
        const int row = 4 * q + r, col = 16 * T + lr;
        if (row == 0) i = col < H ? offW3 + col : (col == H ? offb3 : -1);
      } else if (T < MT + MT * MT) {                 // dW2 / db2
        const int u = T - MT, ti = u / MT, tk = u - ti * MT;
        const int row = 16 * ti + 4 * q + r, col = 16 * tk + lr;
        if (row < H) i = col < H ? offW2 + row * H + col : (col == H ? offb2 + row : -1);
      } else {                                       // dW1 / db1
        const int ti = T - MT - MT * MT;
        const int row = 16 * ti + 4 * q + r, col = lr;
        if (row < H) i = col < K0 ? row * K0 + col : (col == K0 ? offb1 + row : -1);
      }
    }
    if (__syncthreads_or(i >= 0)) {
      float p0 = 0.f, m0 = 0.f, v0 = 0.f, pt0 = 0.f;
      if (g.apply && i >= 0) {                       // issue the parameter loads before the slab stream
        p0 = g.p[i]; m0 = g.m[i]; v0 = g.v[i];
        if (g.pt) pt0 = g.pt[i];
      }
      const int tx4 = tid & 7, ty = tid >> 3;        // 16-byte lane inside the 128-byte half row, slab group
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      {
        const f32x4* sp = reinterpret_cast<const f32x4*>(g.slabs + (size_t)c * g.nslab * 64 + 32 * half) + tx4;   // row stride: 16 float4
        int z = ty;
        for (; z + 240 < g.nslab; z += 256) {
          f32x4 v[16];
#pragma unroll
          for (int u = 0; u < 16; ++u) v[u] = PDEC_SLAB_LOAD(&sp[(size_t)(z + 16 * u) * 16]);
#pragma unroll
          for (int u = 0; u < 16; ++u) acc += v[u];
        }
        for (; z < g.nslab; z += 16) acc += sp[(size_t)z * 16];
      }
      *reinterpret_cast<f32x4*>(&part[ty][4 * tx4]) = acc;
      __syncthreads();
      if (i >= 0) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) a += part[k][tid];
        a *= g.scale;
        g.grads[i] = a;
        if (g.apply) finish_param(g, i, a, p0, m0, v0, pt0);
      }
    }
  } else {
    const int i = blockIdx.x * FIN_THREADS + tid;
    if (i < n && g.apply) finish_param(g, i, g.grads[i], g.p[i], g.m[i], g.v[i], g.pt ? g.pt[i] : 0.f);
  }
  if (blockIdx.x == 0 && g.loss_out && g.nslab > 0) {   // block 0 also finalises the loss (fixed-order tree -> deterministic)
    // the round-2 tree: 256 strided partial sums, pairwise tree 128, 64, ..., 1 -- thread t owns partials t and t + 128
    // and starts with their sum (= the tree's first level)
    __shared__ double red[5][FIN_THREADS];
    const float* stc = g.slabs + (size_t)(4 * (2 * MT + MT * MT)) * g.nslab * 64;   // stats chunk
    double st[2][5] = {{0, 0, 0, 0, 0}, {0, 0, 0, 0, 0}};
    for (int hh = 0; hh < 2; ++hh)
      for (int z = tid + 128 * hh; z < g.nslab; z += 256)
        for (int k = 0; k < 5; ++k) st[hh][k] += (double)stc[(size_t)z * 64 + k];
    for (int k = 0; k < 5; ++k) red[k][tid] = st[0][k] + st[1][k];
    __syncthreads();
    for (int sft = 64; sft > 0; sft >>= 1) {
      if (tid < sft)
        for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + sft];
      __syncthreads();
    }
    if (tid == 0) {
      const double inv = 1.0 / g.Bu;
      if (g.mode == 0)   // critic: quirk -> mean(c^2) + 2 mean(c) mean(r) + mean(r^2); else mean((r+c)^2)
        *g.loss_out = (float)(g.quirk ? red[1][0] * inv + 2.0 * (red[0][0] * inv) * (red[2][0] * inv) + red[3][0] * inv
                                      : red[4][0] * inv);
      else               // actor: -mean(q)
        *g.loss_out = (float)(-red[0][0] * inv);
    }
  }
}

// ------------------------------------------------------------------ fused policy act
// actions = clamp(actor(state) + randn * act_noise, +-act_limit)  (src/PDEagent.jl:183-207) in ONE launch on the
// same MFMA building blocks as the update passes (16 columns per wave, layer outputs chained as B operands);
// exploration noise from the same Philox counter stream as pdec_randn (element index = column), so both
// paths draw identical numbers.
#define ACT_THREADS 256
template <int MTA>
__global__ __launch_bounds__(ACT_THREADS) void policy_act_fused_kernel(FNet f, const float* __restrict__ state, int cols, int ns,
                                                                      float act_noise, float lim, int learning, int tanh_out,
                                                                      uint64_t seed, uint64_t offset, float* __restrict__ out,
                                                                      const uint64_t* ctr_cur, uint64_t* ctr_next, uint64_t ctr_inc,
                                                                      int prio) {
  extern __shared__ __align__(16) float smem[];
  if (ctr_cur) {       // noise counter kept on the device (pdec_policy_act_rng_dev): read it, one thread writes the advanced value
    offset += *ctr_cur;
    if (blockIdx.x == 0 && threadIdx.x == 0) *ctr_next = offset + ctr_inc;
  }
  // wave priority: PDEC_PRIO_ACT (default 3: short and latency-critical when nothing else holds the PDE step back)
  set_wave_prio(prio);
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  constexpr int HPa = 16 * MTA, LDWa = HPa + LDWPAD, NSa = small_floats(HPa), NBa = big_floats(HPa);
  float* sa = smem;                                  // small block
  float* saW2 = sa + NSa;                            // W2 [HPa][LDWa]
  const SmallLds SA = carve_small(sa, HPa);
  const int c = blockIdx.x * (ACT_THREADS / 4) + w * 16 + lr;
  const bool valid = c < cols;
  float x[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int row = 4 * t + q;
    x[t] = (valid && row < ns) ? state[(size_t)c * ns + row] : 0.f;
  }
  dma_even<NSa, ACT_THREADS / 64>(sa, f.w, w, l);                 // the image is laid out as it sits in LDS: two copies
  dma_even<NBa, ACT_THREADS / 64>(saW2, f.w + f.oW2, w, l);
  dma_wait();
  __syncthreads();
  f32x4 h1[MTA], h2[MTA];
  layer_in<MTA, 4>(h1, x, SA, lr, q);
  layer_hh<MTA, MTA, true>(h2, h1, saW2, LDWa, SA.b2, lr, q);
  relu_<MTA>(h2);
  float o = head<MTA>(h2, SA.w3, SA.b3[0], q);
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
static int mt_of(int H) { return (H + 1 + 15) / 16; }

static bool fused_disabled() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("PDEC_DISABLE_FUSED");
    v = (e && e[0] == '1') ? 1 : 0;
  }
  return v == 1;
}

// one 3-layer fp32 net [K0, H, H, 1] relu/relu/(tanh|identity) whose padded image the fused kernels can stage
bool fused_net_supported(const Mlp* M) {
  if (fused_disabled() || M->dtype != PDEC_F32 || M->L != 3) return false;
  if (M->dims[1] != M->dims[2] || M->dims[0] + 1 > KXP || M->dims[3] != 1) return false;
  if (M->acts[0] != PDEC_ACT_RELU || M->acts[1] != PDEC_ACT_RELU) return false;
  const int mt = mt_of(M->dims[1]);
  return mt == 9 || mt == 2 || mt == 1;
}

bool fused_supported(const Mlp* A, const Mlp* C) {
  if (fused_disabled()) return false;
  if (A->dtype != PDEC_F32 || C->dtype != PDEC_F32) return false;
  if (A->L != 3 || C->L != 3) return false;
  const int ns = A->dims[0], na = A->dims[3];
  if (na != 1 || C->dims[0] != ns + na || C->dims[3] != 1 || ns + na + 1 > KXP) return false;
  if (A->dims[1] != A->dims[2] || C->dims[1] != C->dims[2]) return false;
  if (A->acts[0] != PDEC_ACT_RELU || A->acts[1] != PDEC_ACT_RELU || A->acts[2] != PDEC_ACT_TANH) return false;
  if (C->acts[0] != PDEC_ACT_RELU || C->acts[1] != PDEC_ACT_RELU || C->acts[2] != PDEC_ACT_IDENTITY) return false;
  const int mt = mt_of(C->dims[1]), mta = mt_of(A->dims[1]);
  return (mt == 9 || mt == 2) && (mta == 2 || mta == 1);
}

static int ensure_prepped(Mlp* M) {
  const FNet f = make_fnet_layout(M->dims[0], M->dims[1]);
  const size_t bytes = (size_t)f.total * 4;
  if (M->fw.bytes < bytes) {
    PDEC_HIP(M->fw.alloc(bytes));
    PDEC_HIP(M->fw_pub[0].alloc(bytes));
    PDEC_HIP(M->fw_pub[1].alloc(bytes));
    M->fw_dirty = true;
  }
  if (M->fw_dirty) {
    ProfScope ps(M, "fused_prep");
    hipLaunchKernelGGL(prep_fused_kernel, dim3((f.total + 255) / 256), dim3(256), 0, M->stream, M->params.as<float>(),
                       M->fw.as<float>(), f);
    PDEC_HIP(hipGetLastError());
    // both published copies (the images the acting kernel reads, see fused_policy_act) restart from the fresh image
    PDEC_HIP(hipMemcpyAsync(M->fw_pub[0].p, M->fw.p, bytes, hipMemcpyDeviceToDevice, M->stream));
    PDEC_HIP(hipMemcpyAsync(M->fw_pub[1].p, M->fw.p, bytes, hipMemcpyDeviceToDevice, M->stream));
    M->fw_dirty = false;
  }
  return PDEC_OK;
}

static FNet fnet_of(const Mlp* M) {
  FNet f = make_fnet_layout(M->dims[0], M->dims[1]);
  f.w = M->fw.as<float>();
  return f;
}

template <int MT, int MTA>
static size_t lds_bytes(bool actor_pass) {
  const int HP = 16 * MT, HPa = 16 * MTA, LDW = HP + LDWPAD;
  size_t f = (size_t)wreg_floats(MT, MTA) + small_floats(HPa) + 8;
  if (actor_pass) {   // sc | sa | saW2 | saW2T | red [8] | X (the first rows of W2^T, see the kernel)
    f += small_floats(HP) + 2 * (size_t)big_floats(HPa);
    if (HP > AX_ROWS && (AX_ROWS * LDW) % 256 == 0) f += (size_t)AX_ROWS * LDW;
  } else {            // sc | sc2 | sa | saW2 | red [8] + [8][HP] pass A + [8][8] loss statistics
    f += 2 * (size_t)small_floats(HP) + big_floats(HPa) + 8 * HP + 64;
  }
  return f * 4;
}

// Phase stamps of the critic pass (diagnostic; PDEC_STAMPS=1 prints them, pdec_debug_critic_stamps arms one launch and returns
// them): per-phase s_memtime deltas averaged over the workgroups, and the shader clock the pass ran at
static int read_stamps(Mlp* C, unsigned long long* dev, int grid, double* out13) {
  PDEC_HIP(hipStreamSynchronize(C->stream));
  std::vector<unsigned long long> h((size_t)grid * 16);
  PDEC_HIP(hipMemcpy(h.data(), dev, h.size() * 8, hipMemcpyDeviceToHost));
  double d[10] = {0}, clk = 0;
  for (int b = 0; b < grid; ++b) {
    for (int k = 0; k < 10; ++k) d[k] += (double)(h[b * 16 + k + 1] - h[b * 16 + k]);
    const double rt = (double)(h[b * 16 + 12] - h[b * 16 + 11]);          // 100 MHz ticks
    if (rt > 0) clk += (double)(h[b * 16 + 10] - h[b * 16]) / rt * 0.1;     // GHz
  }
  double tot = 0;
  for (int k = 0; k < 10; ++k) { out13[k] = d[k] / grid; tot += out13[k]; }
  out13[10] = tot;              // shader cycles per workgroup, first to last stamp
  out13[11] = clk / grid;       // shader clock in GHz during the pass (mean over workgroups)
  out13[12] = grid;
  return PDEC_OK;
}
static int dump_stamps(Mlp* C, unsigned long long* dev, int grid) {
  double o[13];
  int rc = read_stamps(C, dev, grid, o);
  if (rc) return rc;
  fprintf(stderr, "[pdec stamps] critic pass, mean shader cycles per phase over %d WGs:", grid);
  static const char* nm[10] = {"load+rbar", "target", "loadQ", "fwdQ", "passA", "loadW2T", "dz1", "passC", "passB", "stats"};
  for (int k = 0; k < 10; ++k) fprintf(stderr, " %s=%.0f", nm[k], o[k]);
  fprintf(stderr, " | total=%.0f cycles, clock=%.3f GHz\n", o[10], o[11]);
  return PDEC_OK;
}

// (Round 5: the bf16-split forms of the passes -- an experiment of rounds 2 - 4 whose results beside the PDE step were not
// bit-reproducible and whose cause was never found -- are deleted; HISTORY.md §3.2a keeps the record and the numbers.  A process
// that still sets PDEC_SPLIT gets an error, not a silent exact-f32 run.)
static int split_refused() {
  const char* e = getenv("PDEC_SPLIT");
  if (e && e[0] && e[0] != '0') {
    set_error("PDEC_SPLIT=%s: the experimental bf16-split passes no longer exist (HISTORY.md §3.2a); the passes are exact f32", e);
    return PDEC_E_INVALID;
  }
  return PDEC_OK;
}
template <int MT, int MTA>
static int launch_critic(Mlp* C, const FusedArgs& g, int grid) {
  if (int rc = split_refused()) return rc;
  const size_t lds = lds_bytes<MT, MTA>(false);
  auto kern = ddpg_critic_fused_kernel<MT, MTA>;
  static bool attr_set = false;
  if (!attr_set) {
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  static const bool env_stamps = getenv("PDEC_STAMPS") != nullptr;
  const bool want_stamps = env_stamps || C->stamps_armed;
  FusedArgs ga = g;
  if (want_stamps) {
    if (C->stamps.bytes < (size_t)grid * 16 * 8) PDEC_HIP(C->stamps.alloc((size_t)grid * 16 * 8));
    ga.stamps = C->stamps.as<unsigned long long>();
  }
  if (C->prof && C->prof_reps == 1) {
    PDEC_TIMED_LAUNCH(C, "ddpg_critic_fused", kern, dim3(grid), dim3(FTHREADS), lds, ga);
  } else {
    ProfScope ps(C, "ddpg_critic_fused", true);
    for (int rep = 0; rep < ps.reps; ++rep)
      hipLaunchKernelGGL(kern, dim3(grid), dim3(FTHREADS), lds, C->stream, ga);
  }
  PDEC_HIP(hipGetLastError());
  if (C->stamps_armed) {          // pdec_debug_critic_stamps: keep the figures of this launch for the caller
    C->stamps_armed = false;
    return read_stamps(C, ga.stamps, grid, C->stamps_last);
  }
  if (want_stamps) return dump_stamps(C, ga.stamps, grid);
  return PDEC_OK;
}
template <int MT, int MTA>
static int launch_actor(Mlp* C, const FusedArgs& g, int grid) {
  if (int rc = split_refused()) return rc;
  const size_t lds = lds_bytes<MT, MTA>(true);
  auto kern = ddpg_actor_fused_kernel<MT, MTA>;
  static bool attr_set = false;
  if (!attr_set) {
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (C->prof && C->prof_reps == 1) {
    PDEC_TIMED_LAUNCH(C, "ddpg_actor_fused", kern, dim3(grid), dim3(FTHREADS), lds, g);
  } else {
    ProfScope ps(C, "ddpg_actor_fused", true);
    for (int rep = 0; rep < ps.reps; ++rep)
      hipLaunchKernelGGL(kern, dim3(grid), dim3(FTHREADS), lds, C->stream, g);
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static int ensure_slab(Mlp* M, size_t floats) {
  if (M->fslab.bytes < floats * 4) PDEC_HIP(M->fslab.alloc(floats * 4));
  return PDEC_OK;
}

// Slab reduction (nslab > 0) and/or the parameter update (apply): ADAM on M with M->adam_* hyper-parameters,
// Polyak into Mt, refresh of both padded images.  One launch.
static int launch_finish(Mlp* M, Mlp* Mt, const float* slabs, int nslab, int MT, double grad_scale, int mode,
                         int Bu, int quirk, void* loss_dev, const AdamPolyak* ap) {
  FinishArgs g{};
  g.slabs = slabs; g.nslab = nslab; g.K0 = M->dims[0]; g.H = M->dims[1]; g.MT = MT;
  g.grads = M->grads.as<float>(); g.scale = (float)grad_scale; g.mode = mode; g.Bu = Bu; g.quirk = quirk;
  g.loss_out = (float*)loss_dev;
  g.apply = ap != nullptr;
  g.prio = env_prio("PDEC_PRIO_MFMA", 2);
  if (ap) {
    int rcb = bp_begin(M, ap->b1, ap->b2, &g.bp);
    if (rcb) return rcb;
    g.p = M->params.as<float>(); g.m = M->m.as<float>(); g.v = M->v.as<float>();
    g.fw = M->fw.as<float>();
    g.fwp = M->fw_pub[M->pub ^ 1].as<float>();       // written now, read by acting kernels enqueued after this launch
    g.lay = make_fnet_layout(M->dims[0], M->dims[1]);
    g.eta = ap->eta; g.b1 = ap->b1; g.b2 = ap->b2; g.eps = ap->eps;
    // rho == 1 (the reference as it runs: its Polyak loop iterates over an empty parameter list, src/PDEagent.jl:415-417 with
    // src/custom_nna.jl:20 -- agent.py quirk_frozen_targets): the target network is not touched at all, as in the reference
    // (dest = 1 * dest + 0 * src would rewrite it with its own bits, and turn an Inf in src into a NaN in dest)
    if (Mt && (float)ap->rho != 1.0f) {
      g.pt = Mt->params.as<float>(); g.fwt = Mt->fw.as<float>();
      const float r = (float)ap->rho;   // the reference holds p = 0.995f0 and computes (1 - p) in Float32
      g.rho = r; g.omr = 1.0f - r;
    }
  }
  const int n = M->nparams;
  {
    const char* label = ap ? (nslab > 0 ? "fused_finish" : "fused_apply") : "fused_reduce";
    const bool use_ref = getenv("PDEC_FINISH_REF") != nullptr;            // tests only: the round-2 kernel (bit-identity reference)
    // the one-shot completion event of this launch: the apply launch carries the stop event (parameters updated), a reduce-only
    // launch the reduce event (flat gradient ready: what an all-reduce on another stream waits for)
    hipEvent_t& slot = ap ? M->stop_event : M->reduce_event;
    const hipEvent_t ev = slot;
    if (use_ref) {
      const int nblk = nslab > 0 ? 4 * slab_tiles(MT) : (n + 63) / 64;
      hipLaunchKernelGGL(fused_finish_ref_kernel, dim3(nblk), dim3(1024), 0, M->stream, g);
      if (ev) (void)hipEventRecord(ev, M->stream);
    } else {
      const int nblk = nslab > 0 ? 8 * slab_tiles(MT) : (n + FIN_THREADS - 1) / FIN_THREADS;
      if (M->prof) {
        PDEC_TIMED_LAUNCH(M, label, fused_finish_kernel, dim3(nblk), dim3(FIN_THREADS), 0, g);
        if (ev) (void)hipEventRecord(ev, M->stream);
      } else if (ev) {
        // the event rides on this kernel's own dispatch packet (its completion signal): a hipEventRecord behind the
        // launch is a packet of its own that the next kernel of the stream has to wait for (~4.5 us of the update chain)
        hipExtLaunchKernelGGL(fused_finish_kernel, dim3(nblk), dim3(FIN_THREADS), 0, M->stream, nullptr, ev, 0, g);
      } else {
        hipLaunchKernelGGL(fused_finish_kernel, dim3(nblk), dim3(FIN_THREADS), 0, M->stream, g);
      }
    }
    slot = nullptr;     // consumed (the stop event only by a launch that applies the update, the reduce event only by a reduce-only one)
  }
  PDEC_HIP(hipGetLastError());
  if (ap) {
    bp_done(M);
    flip(M->pub);
  }
  return PDEC_OK;
}

int fused_policy_act(Mlp* A, const void* state, int cols, double act_noise, double act_limit, int learning, uint64_t seed,
                     uint64_t offset, void* actions_out, const uint64_t* ctr_cur, uint64_t* ctr_next, uint64_t ctr_inc) {
  int rc = ensure_prepped(A);
  if (rc) return rc;
  FNet f = fnet_of(A);
  // Read the PUBLISHED copy of the image: the update's finish kernel writes the other copy (and the in-place image
  // of the update passes), so an acting kernel may run beside the actor half of the following update.  A copy is
  // rewritten two updates later; callers that overlap acting and updating on different streams must order update
  // t+1 behind the acting kernel of step t-1 (bench.py waits on that event at the start of each update).
  f.w = A->fw_pub[A->pub].as<float>();
  const int mta = mt_of(A->dims[1]);
  const int HPa = 16 * mta;
  const size_t lds = ((size_t)small_floats(HPa) + (size_t)big_floats(HPa)) * 4;
  const int tanh_out = A->acts[2] == PDEC_ACT_TANH;
  PDEC_REQUIRE(A->acts[2] == PDEC_ACT_TANH || A->acts[2] == PDEC_ACT_IDENTITY, "fused act: unsupported output activation");
  PDEC_REQUIRE(mta <= 2 && lds <= 64 * 1024, "fused act: hidden width %d too large", A->dims[1]);
  const int prio = env_prio("PDEC_PRIO_ACT", 3);
  const dim3 grid((cols + 63) / 64), block(ACT_THREADS);
#define ACT_ARGS f, (const float*)state, cols, A->dims[0], (float)act_noise, (float)act_limit, learning, tanh_out, seed, offset, \
                 (float*)actions_out, ctr_cur, ctr_next, ctr_inc, prio
  if (A->prof) {
    if (mta == 1) PDEC_TIMED_LAUNCH(A, "policy_act_fused", policy_act_fused_kernel<1>, grid, block, lds, ACT_ARGS);
    else PDEC_TIMED_LAUNCH(A, "policy_act_fused", policy_act_fused_kernel<2>, grid, block, lds, ACT_ARGS);
  } else if (mta == 1) {
    hipLaunchKernelGGL(policy_act_fused_kernel<1>, grid, block, lds, A->stream, ACT_ARGS);
  } else {
    hipLaunchKernelGGL(policy_act_fused_kernel<2>, grid, block, lds, A->stream, ACT_ARGS);
  }
#undef ACT_ARGS
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// ADAM(M) + Polyak(Mt <- M) + image refresh from the gradient buffer (after an external all-reduce)
int fused_adam_polyak(Mlp* M, Mlp* Mt, const AdamPolyak& ap) {
  int rc;
  if ((rc = ensure_prepped(M)) || (Mt && (rc = ensure_prepped(Mt)))) return rc;
  return launch_finish(M, Mt, nullptr, 0, mt_of(M->dims[1]), 1.0, 0, 1, 0, nullptr, &ap);
}

int fused_critic_grads(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                       const void* sn, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev,
                       const AdamPolyak* apply) {
  int rc;
  if ((rc = ensure_prepped(C)) || (rc = ensure_prepped(At)) || (rc = ensure_prepped(Ct))) return rc;
  const int mt = mt_of(C->dims[1]), mta = mt_of(A->dims[1]);
  const int grid = (Bu + FCOLS - 1) / FCOLS;
  if ((rc = ensure_slab(C, slab_floats_total(mt, grid)))) return rc;
  FusedArgs g{};
  g.C = fnet_of(C); g.At = fnet_of(At); g.Ct = fnet_of(Ct); g.A = g.At;
  g.s = (const float*)s; g.a = (const float*)a; g.r = (const float*)r; g.t = (const float*)t; g.sn = (const float*)sn;
  g.Bu = Bu; g.ns = A->dims[0]; g.na = 1; g.gamma = (float)gamma; g.quirk = quirk;
  g.slab = C->fslab.as<float>();
  g.prio = env_prio("PDEC_PRIO_MFMA", 2);
  if (quirk && C->rpart_ext) {         // the producer of r left one partial sum per workgroup
    g.rpart = C->rpart_ext; g.nrpart = C->rpart_n;
  } else if (quirk && C->rbar_ext) {   // reduced by the producer of r on its own stream (pdec_reward_mean): nothing to sum here
    g.rbar_dev = (const float*)C->rbar_ext;
  } else if (quirk && Bu > 256 * FCOLS) {     // every workgroup summing all of r itself does not scale (C3: 131072 rewards): reduce once
    float* rb = nullptr;
    if ((rc = launch_rmean(C, (const float*)r, Bu, &rb))) return rc;
    g.rbar_dev = rb;
  }
  C->rbar_ext = nullptr;
  C->rpart_ext = nullptr;
  if (mt == 9 && mta == 2) rc = launch_critic<9, 2>(C, g, grid);
  else if (mt == 9 && mta == 1) rc = launch_critic<9, 1>(C, g, grid);
  else if (mt == 2 && mta == 2) rc = launch_critic<2, 2>(C, g, grid);
  else rc = launch_critic<2, 1>(C, g, grid);
  if (rc) return rc;
  return launch_finish(C, apply ? Ct : nullptr, C->fslab.as<float>(), grid, mt, grad_scale, 0, Bu, quirk, loss_dev, apply);
}

int fused_actor_grads(Mlp* A, Mlp* C, Mlp* At, const void* s, int Bu, double grad_scale, void* loss_dev,
                      const AdamPolyak* apply) {
  int rc;
  if ((rc = ensure_prepped(C)) || (rc = ensure_prepped(A)) || (apply && At && (rc = ensure_prepped(At)))) return rc;
  const int mt = mt_of(C->dims[1]), mta = mt_of(A->dims[1]);
  const int grid = (Bu + FCOLS - 1) / FCOLS;
  if ((rc = ensure_slab(A, slab_floats_total(mta, grid)))) return rc;
  FusedArgs g{};
  g.C = fnet_of(C); g.A = fnet_of(A); g.At = g.A; g.Ct = g.C;
  g.s = (const float*)s; g.Bu = Bu; g.ns = A->dims[0]; g.na = 1;
  g.slab = A->fslab.as<float>();
  g.prio = env_prio("PDEC_PRIO_MFMA", 2);
  // the actor pass is launched on the critic's stream object for profiling labels but must follow
  // ADAM(C); both handles share one stream in every caller (checked by the dispatcher)
  if (mt == 9 && mta == 2) rc = launch_actor<9, 2>(C, g, grid);
  else if (mt == 9 && mta == 1) rc = launch_actor<9, 1>(C, g, grid);
  else if (mt == 2 && mta == 2) rc = launch_actor<2, 2>(C, g, grid);
  else rc = launch_actor<2, 1>(C, g, grid);
  if (rc) return rc;
  (void)C;
  return launch_finish(A, apply ? At : nullptr, A->fslab.as<float>(), grid, mta, grad_scale, 1, Bu, 0, loss_dev, apply);
}

}  // namespace pdec
