// mlp.hip -- weight-shared ("convolutional") actor/critic MLPs on gfx950: forward,
// backward, Flux-ADAM, Polyak, policy act and the DDPG update.
//
// Restates (from scratch):
//   create_NNA / Chain(Dense...)            src/PDEagent.jl:14-56
//   CustomNeuralNetworkApproximator         src/custom_nna.jl:7-27
//   (policy::CustomDDPGPolicy)(env)         src/PDEagent.jl:175-209
//   RLBase.update!(policy, batch)           src/PDEagent.jl:363-418
//
// A Dense layer applied to the im2col state matrix [features, columns] is a 1-D/2-D
// convolution with circular padding expressed as a GEMM over columns.  Activations are
// kept feature-major H[f][col] (columns contiguous -> coalesced, one column per lane).
// This file holds the GENERIC path (any widths, fp32 and fp64): LDS-tiled VALU GEMMs with
// fused bias/activation and activation-derivative epilogues, split-K weight gradients with
// a deterministic slab reduction (so data-parallel replicas stay bit-identical).  The
// fp32 MFMA fast path for the wide critic lives in mlp_mfma.hip.
#include "common.hpp"
#include "mlp.hpp"

namespace pdec {

// ------------------------------------------------------------------ tiled GEMM
#define GB_M 64
#define GB_N 64
#define GB_K 16

enum { EPI_STORE = 0, EPI_BIAS_ACT = 1, EPI_MUL_DACT = 2 };

template <class T>
struct GemmArgs {
  int M, N, K, kchunk;
  const T* A; long sam, sak;
  const T* B; long sbk, sbn;
  T* C; long scm, scn, scz;       // scz: slab stride for split-K (blockIdx.z)
  int epi, act;
  const T* bias;                  // EPI_BIAS_ACT: bias[m]
  const T* aux; long sauxm, sauxn;  // EPI_MUL_DACT: activation values at (m,n)
};

template <class T>
__device__ __forceinline__ T apply_act(T z, int act) {
  if (act == PDEC_ACT_RELU) return z > 0 ? z : (T)0;
  if (act == PDEC_ACT_TANH) return (T)tanh((double)z);
  return z;
}
template <>
__device__ __forceinline__ float apply_act<float>(float z, int act) {
  if (act == PDEC_ACT_RELU) return z > 0 ? z : 0.0f;
  if (act == PDEC_ACT_TANH) return tanhf(z);
  return z;
}
// derivative expressed through the activation VALUE a = act(z)
template <class T>
__device__ __forceinline__ T dact_from_value(T a, int act) {
  if (act == PDEC_ACT_RELU) return a > 0 ? (T)1 : (T)0;
  if (act == PDEC_ACT_TANH) return (T)1 - a * a;
  return (T)1;
}

template <class T>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs<T> g) {
  __shared__ T As[GB_K][GB_M + 1];
  __shared__ T Bs[GB_K][GB_N + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int m0 = blockIdx.y * GB_M, n0 = blockIdx.x * GB_N;
  const int kbeg = blockIdx.z * g.kchunk;
  int kend = kbeg + g.kchunk;
  if (kend > g.K) kend = g.K;
  T acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0;
  for (int k0 = kbeg; k0 < kend; k0 += GB_K) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + i * 256;
      int m, k;
      if (g.sak == 1) { k = idx & 15; m = idx >> 4; } else { m = idx & 63; k = idx >> 6; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < g.M && gk < kend) ? g.A[gm * g.sam + gk * g.sak] : (T)0;
      int n, kb;
      if (g.sbn == 1) { n = idx & 63; kb = idx >> 6; } else { kb = idx & 15; n = idx >> 4; }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < g.N && gkb < kend) ? g.B[gkb * g.sbk + gn * g.sbn] : (T)0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < GB_K; ++k) {
      T a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = As[k][ty + 16 * i];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[k][tx + 16 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
    }
    __syncthreads();
  }
  T* C = g.C + (size_t)blockIdx.z * g.scz;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty + 16 * i;
    if (m >= g.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx + 16 * j;
      if (n >= g.N) continue;
      T v = acc[i][j];
      if (g.epi == EPI_BIAS_ACT) v = apply_act<T>(v + g.bias[m], g.act);
      else if (g.epi == EPI_MUL_DACT) v *= dact_from_value<T>(g.aux[m * g.sauxm + n * g.sauxn], g.act);
      C[m * g.scm + n * g.scn] = v;
    }
  }
}

// grad[i] = scale * sum_z slab[z][i]
template <class T>
__global__ void reduce_slabs_kernel(const T* __restrict__ slabs, T* __restrict__ grad, int n, int nz, size_t zstride,
                                    T scale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  T acc = 0;
  for (int z = 0; z < nz; ++z) acc += slabs[(size_t)z * zstride + i];
  grad[i] = acc * scale;
}

// db[m] = scale * sum_c dz[m][c]   (one block per row)
template <class T>
__global__ void rowsum_kernel(const T* __restrict__ dz, T* __restrict__ db, int cols, T scale) {
  __shared__ T red[256];
  const int m = blockIdx.x, tid = threadIdx.x;
  T acc = 0;
  for (int c = tid; c < cols; c += 256) acc += dz[(size_t)m * cols + c];
  red[tid] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) db[m] = red[0] * scale;
}

// X0[f][c] from up to two sources; layout flag 0: [cols][n] (Julia [n,cols]), 1: feature-major [n][cols]
template <class T>
__global__ void pack_kernel(T* __restrict__ X0, int cols, const T* __restrict__ s1, int n1, int l1,
                            const T* __restrict__ s2, int n2, int l2) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int f = 0; f < n1; ++f) X0[(size_t)f * cols + c] = l1 ? s1[(size_t)f * cols + c] : s1[(size_t)c * n1 + f];
  for (int f = 0; f < n2; ++f)
    X0[(size_t)(n1 + f) * cols + c] = l2 ? s2[(size_t)f * cols + c] : s2[(size_t)c * n2 + f];
}

// out[c][f] = H[f][c]
template <class T>
__global__ void unpack_kernel(const T* __restrict__ H, int cols, int n, T* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int f = 0; f < n; ++f) out[(size_t)c * n + f] = H[(size_t)f * cols + c];
}

// dz[f][c] = dy(f,c) * act'(a[f][c]); dy layout flag as in pack
template <class T>
__global__ void dz_init_kernel(T* __restrict__ dz, const T* __restrict__ dy, int ldy, const T* __restrict__ a,
                               int n, int cols, int act) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int f = 0; f < n; ++f) {
    const T d = ldy ? dy[(size_t)f * cols + c] : dy[(size_t)c * n + f];
    dz[(size_t)f * cols + c] = d * dact_from_value<T>(a[(size_t)f * cols + c], act);
  }
}

// Flux.Optimise.ADAM apply! + update!: arithmetic in Float64 (beta, eps, eta are Float64 in
// Flux, so the broadcast promotes), stored back in T.
template <class T>
__global__ void adam_kernel(T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v, int n,
                            double eta, double b1, double b2, double eps, BpArgs bp) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double omb1p = 1.0 - bp.cur[0], omb2p = 1.0 - bp.cur[1];
  if (i == 0) bp_advance(bp, b1, b2);
  if (i >= n) return;
  const double gi = (double)g[i];
  const T mt = (T)(b1 * (double)m[i] + (1.0 - b1) * gi);
  const T vt = (T)(b2 * (double)v[i] + (1.0 - b2) * gi * gi);
  m[i] = mt;
  v[i] = vt;
  const T delta = (T)((double)mt / omb1p / (sqrt((double)vt / omb2p) + eps) * eta);
  p[i] = p[i] - delta;
}

template <class T>
__global__ void polyak_kernel(T* __restrict__ dst, const T* __restrict__ src, int n, T rho, T omr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = rho * dst[i] + omr * src[i];
}

template <class TD, class TS>
__global__ void cast_copy_kernel(TD* __restrict__ dst, const TS* __restrict__ src, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (TD)src[i];
}

// actions[c][f] = clamp(H[f][c] + noise[c][f]*act_noise, +-lim)
// (noise on the first nrows outputs only: the action-memory rows stay noise-free, src/PDEagent.jl:201)
template <class T>
__global__ void act_noise_clamp_kernel(const T* __restrict__ H, const T* __restrict__ noise, int cols, int n, int nrows,
                                       T act_noise, T lim, T* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int f = 0; f < n; ++f) {
    T v = H[(size_t)f * cols + c];
    if (noise && f < nrows) v += noise[(size_t)c * n + f] * act_noise;
    v = v < -lim ? -lim : (v > lim ? lim : v);
    out[(size_t)c * n + f] = v;
  }
}

// ---- counter-based normals: Philox4x32-10 + Box-Muller (philox4x32 in mlp.hpp; replaces randn(rng), PDEagent.jl:201)
template <class T>
__global__ void randn_kernel(T* __restrict__ dst, size_t n, uint64_t seed, uint64_t offset, const uint64_t* ctr_cur = nullptr,
                             uint64_t* ctr_next = nullptr, uint64_t ctr_inc = 0) {
  const size_t q = blockIdx.x * (size_t)blockDim.x + threadIdx.x;  // 4 normals per thread
  if (ctr_cur) {       // device-resident counter (pdec_policy_act_rng_dev)
    offset += *ctr_cur;
    if (q == 0) *ctr_next = offset + ctr_inc;
  }
  if (q * 4 >= n) return;
  const uint64_t ctr = offset + q;
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  const double s = 1.0 / 4294967296.0;
  for (int h = 0; h < 2; ++h) {
    const double u1 = ((double)c[2 * h] + 0.5) * s, u2 = ((double)c[2 * h + 1] + 0.5) * s;
    const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
    const size_t i = q * 4 + 2 * h;
    if (i < n) dst[i] = (T)(rad * cos(ang));
    if (i + 1 < n) dst[i + 1] = (T)(rad * sin(ang));
  }
}

// ---- DDPG loss statistics (single block; deterministic)
// stats[0..4] = mean(c), mean(c^2), mean(r), mean(r^2), mean(q) with c_i = gamma(1-t_i)qt_i - q_i
template <class T>
__global__ void ddpg_stats_kernel(const T* __restrict__ q, const T* __restrict__ qt, const T* __restrict__ r,
                                  const T* __restrict__ t, int n, T gamma, T* __restrict__ stats) {
  __shared__ double red[5][256];
  const int tid = threadIdx.x;
  double a[5] = {0, 0, 0, 0, 0};
  for (int i = tid; i < n; i += 256) {
    const T qi = q[i];
    a[4] += (double)qi;
    if (qt) {
      const T c = gamma * ((T)1 - t[i]) * qt[i] - qi;
      a[0] += (double)c;
      a[1] += (double)c * (double)c;
      a[2] += (double)r[i];
      a[3] += (double)r[i] * (double)r[i];
    }
  }
  for (int k = 0; k < 5; ++k) red[k][tid] = a[k];
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s)
      for (int k = 0; k < 5; ++k) red[k][tid] += red[k][tid + s];
    __syncthreads();
  }
  if (tid < 5) stats[tid] = (T)(red[tid][0] / n);
}

// critic: dq_i = -(2/Bu)(rr + c_i) * scale with rr = mean(r) (quirk) or r_i; writes dz of the
// (identity) output layer directly; loss -> *loss_out
template <class T>
__global__ void ddpg_critic_dq_kernel(const T* __restrict__ q, const T* __restrict__ qt, const T* __restrict__ r,
                                      const T* __restrict__ t, int n, T gamma, int quirk, const T* __restrict__ stats,
                                      T* __restrict__ dq, T* __restrict__ loss_out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && loss_out) {
    // quirk: mean_ij (r_j + c_i)^2 = mean(c^2) + 2 mean(c) mean(r) + mean(r^2)
    *loss_out = quirk ? stats[1] + (T)2 * stats[0] * stats[2] + stats[3] : stats[5];
  }
  if (i >= n) return;
  const T c = gamma * ((T)1 - t[i]) * qt[i] - q[i];
  const T rr = quirk ? stats[2] : r[i];
  dq[i] = -((T)2 / (T)n) * (rr + c);
}
// diagonal loss needs mean((r_i + c_i)^2): stats[5]
template <class T>
__global__ void ddpg_diag_loss_kernel(const T* __restrict__ q, const T* __restrict__ qt, const T* __restrict__ r,
                                      const T* __restrict__ t, int n, T gamma, T* __restrict__ stats) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  double a = 0;
  for (int i = tid; i < n; i += 256) {
    const T e = r[i] + gamma * ((T)1 - t[i]) * qt[i] - q[i];
    a += (double)e * (double)e;
  }
  red[tid] = a;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] += red[tid + s];
    __syncthreads();
  }
  if (tid == 0) stats[5] = (T)(red[0] / n);
}
template <class T>
__global__ void fill_kernel(T* __restrict__ p, int n, T v) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = v;
}
template <class T>
__global__ void neg_copy_kernel(T* __restrict__ dst, const T* __restrict__ src) { *dst = -*src; }

// ------------------------------------------------------------------ Mlp methods
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

template <class T>
static int launch_gemm(Mlp& M, const char* label, GemmArgs<T> g, int nz) {
  dim3 grid(cdiv(g.N, GB_N), cdiv(g.M, GB_M), nz), block(256);
  ProfScope ps(&M, label);
  hipLaunchKernelGGL((gemm_kernel<T>), grid, block, 0, M.stream, g);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int Mlp::init(int dtype_, int L_, const int32_t* dims_, const int32_t* acts_, int max_cols_) {
  dtype = dtype_;
  L = L_;
  max_cols = max_cols_;
  dims.assign(dims_, dims_ + L + 1);
  acts.assign(acts_, acts_ + L);
  w_off.resize(L);
  b_off.resize(L);
  size_t off = 0, maxw = 0;
  int maxd = 0;
  for (int l = 0; l < L; ++l) {
    w_off[l] = off;
    off += (size_t)dims[l] * dims[l + 1];
    b_off[l] = off;
    off += dims[l + 1];
    maxw = std::max(maxw, (size_t)dims[l] * dims[l + 1]);
  }
  for (int l = 0; l <= L; ++l) maxd = std::max(maxd, dims[l]);
  nparams = (int)off;
  const size_t ts = dtype_size(dtype);
  PDEC_HIP(params.alloc(off * ts));
  PDEC_HIP(grads.alloc(off * ts));
  PDEC_HIP(m.alloc(off * ts));
  PDEC_HIP(v.alloc(off * ts));
  PDEC_HIP(hipMemset(params.p, 0, off * ts));
  PDEC_HIP(hipMemset(grads.p, 0, off * ts));
  PDEC_HIP(hipMemset(m.p, 0, off * ts));
  PDEC_HIP(hipMemset(v.p, 0, off * ts));
  PDEC_HIP(hipStreamSynchronize(nullptr));     // the memsets are queued on the null stream; later users run on non-blocking streams
  H.resize(L + 1);
  for (int l = 0; l <= L; ++l) PDEC_HIP(H[l].alloc((size_t)dims[l] * max_cols * ts));
  PDEC_HIP(dz[0].alloc((size_t)maxd * max_cols * ts));
  PDEC_HIP(dz[1].alloc((size_t)maxd * max_cols * ts));
  kchunk = 512;
  nsplit_max = cdiv(max_cols, kchunk);
  PDEC_HIP(slabs.alloc((size_t)nsplit_max * maxw * ts));
  PDEC_HIP(dy.alloc((size_t)dims[L] * max_cols * ts));
  PDEC_HIP(scratch.alloc(64 * 8));
  bp_init = false;       // beta powers initialised on the first adam step (Flux: Float64[beta1, beta2])
  return PDEC_OK;
}

template <class T>
int Mlp::pack(const void* s1, int n1, int l1, const void* s2, int n2, int l2, int cols) {
  PDEC_REQUIRE(n1 + n2 == dims[0], "mlp: input rows %d+%d != %d", n1, n2, dims[0]);
  PDEC_REQUIRE(cols >= 1 && cols <= max_cols, "mlp: cols %d exceeds max_cols %d", cols, max_cols);
  ProfScope ps(this, "mlp_pack");
  hipLaunchKernelGGL((pack_kernel<T>), dim3(cdiv(cols, 256)), dim3(256), 0, stream, H[0].as<T>(), cols, (const T*)s1, n1, l1,
                     (const T*)s2, n2, l2);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

template <class T>
int Mlp::forward(int cols) {
  for (int l = 0; l < L; ++l) {
    GemmArgs<T> g{};
    g.M = dims[l + 1]; g.N = cols; g.K = dims[l]; g.kchunk = g.K;
    g.A = params.as<T>() + w_off[l]; g.sam = dims[l]; g.sak = 1;
    g.B = H[l].as<T>(); g.sbk = cols; g.sbn = 1;
    g.C = H[l + 1].as<T>(); g.scm = cols; g.scn = 1; g.scz = 0;
    g.epi = EPI_BIAS_ACT; g.act = acts[l]; g.bias = params.as<T>() + b_off[l];
    int rc = launch_gemm<T>(*this, "mlp_fwd_gemm", g, 1);
    if (rc) return rc;
  }
  return PDEC_OK;
}

// dy: [out][cols] feature-major (ldy=1) or [cols][out] (ldy=0).  Leaves dX0 (w.r.t. the
// packed input, feature-major [in][cols]) in dx_ptr() when want_dx.
template <class T>
int Mlp::backward(const void* dy, int ldy, int cols, bool want_dw, bool want_dx, double grad_scale) {
  int cur = 0;
  {
    ProfScope ps(this, "mlp_dz_init");
    hipLaunchKernelGGL((dz_init_kernel<T>), dim3(cdiv(cols, 256)), dim3(256), 0, stream, dz[cur].as<T>(), (const T*)dy, ldy,
                       H[L].as<T>(), dims[L], cols, acts[L - 1]);
    PDEC_HIP(hipGetLastError());
  }
  for (int l = L - 1; l >= 0; --l) {
    const int out = dims[l + 1], in = dims[l];
    if (want_dw) {
      const int nz = cdiv(cols, kchunk);
      GemmArgs<T> g{};
      g.M = out; g.N = in; g.K = cols; g.kchunk = kchunk;
      g.A = dz[cur].as<T>(); g.sam = cols; g.sak = 1;
      g.B = H[l].as<T>(); g.sbk = 1; g.sbn = cols;
      g.C = slabs.as<T>(); g.scm = in; g.scn = 1; g.scz = (long)out * in;
      g.epi = EPI_STORE;
      int rc = launch_gemm<T>(*this, "mlp_dw_gemm", g, nz);
      if (rc) return rc;
      ProfScope ps(this, "mlp_dw_reduce");
      hipLaunchKernelGGL((reduce_slabs_kernel<T>), dim3(cdiv(out * in, 256)), dim3(256), 0, stream, slabs.as<T>(),
                         grads.as<T>() + w_off[l], out * in, nz, (size_t)out * in, (T)grad_scale);
      hipLaunchKernelGGL((rowsum_kernel<T>), dim3(out), dim3(256), 0, stream, dz[cur].as<T>(), grads.as<T>() + b_off[l], cols,
                         (T)grad_scale);
      PDEC_HIP(hipGetLastError());
    }
    if (l > 0 || want_dx) {
      GemmArgs<T> g{};
      g.M = in; g.N = cols; g.K = out; g.kchunk = out;
      g.A = params.as<T>() + w_off[l]; g.sam = 1; g.sak = in;   // W^T
      g.B = dz[cur].as<T>(); g.sbk = cols; g.sbn = 1;
      g.C = dz[cur ^ 1].as<T>(); g.scm = cols; g.scn = 1; g.scz = 0;
      if (l > 0) {
        g.epi = EPI_MUL_DACT; g.act = acts[l - 1]; g.aux = H[l].as<T>(); g.sauxm = cols; g.sauxn = 1;
      } else {
        g.epi = EPI_STORE;
      }
      int rc = launch_gemm<T>(*this, "mlp_dx_gemm", g, 1);
      if (rc) return rc;
      cur ^= 1;
    }
  }
  dx_index = cur;
  return PDEC_OK;
}

}  // namespace pdec

namespace pdec {
int bp_begin(Mlp* M, double beta1, double beta2, BpArgs* out) {
  if (!M->bp_init) {
    if (!M->bpd.p) PDEC_HIP(M->bpd.alloc(4 * sizeof(double)));
    const double init[4] = {beta1, beta2, beta1, beta2};
    PDEC_HIP(hipMemcpy(M->bpd.p, init, sizeof(init), hipMemcpyHostToDevice));   // first step only (never inside a capture)
    M->bp_sel = 0;
    M->bp_init = true;
  }
  out->cur = M->bpd.as<double>() + 2 * M->bp_sel;
  out->next = M->bpd.as<double>() + 2 * (M->bp_sel ^ 1);
  return PDEC_OK;
}
}  // namespace pdec

using namespace pdec;

#define GET_MLP(M, h)                               \
  Mlp* M = lookup_as<Mlp>(h, Kind::Mlp);            \
  if (!M) {                                         \
    set_error("%s: not an mlp handle", __func__);   \
    return PDEC_E_HANDLE;                           \
  }

#define DISPATCH(M, expr_f, expr_d) ((M)->dtype == PDEC_F64 ? (expr_d) : (expr_f))

// host conversion between the Julia layout (W column-major [out,in]) and the internal one
template <class T>
static void julia_to_internal(const Mlp& M, const T* src, T* dst) {
  size_t so = 0;
  for (int l = 0; l < M.L; ++l) {
    const int in = M.dims[l], out = M.dims[l + 1];
    for (int o = 0; o < out; ++o)
      for (int i = 0; i < in; ++i) dst[M.w_off[l] + (size_t)o * in + i] = src[so + (size_t)i * out + o];
    so += (size_t)in * out;
    for (int o = 0; o < out; ++o) dst[M.b_off[l] + o] = src[so + o];
    so += out;
  }
}
template <class T>
static void internal_to_julia(const Mlp& M, const T* src, T* dst) {
  size_t so = 0;
  for (int l = 0; l < M.L; ++l) {
    const int in = M.dims[l], out = M.dims[l + 1];
    for (int o = 0; o < out; ++o)
      for (int i = 0; i < in; ++i) dst[so + (size_t)i * out + o] = src[M.w_off[l] + (size_t)o * in + i];
    so += (size_t)in * out;
    for (int o = 0; o < out; ++o) dst[so + o] = src[M.b_off[l] + o];
    so += out;
  }
}

static int set_flat(Mlp* M, DevBuf& buf, const void* host) {
  const size_t ts = dtype_size(M->dtype), n = M->nparams;
  std::vector<unsigned char> tmp(n * ts);
  if (M->dtype == PDEC_F64) julia_to_internal<double>(*M, (const double*)host, (double*)tmp.data());
  else julia_to_internal<float>(*M, (const float*)host, (float*)tmp.data());
  PDEC_HIP(hipMemcpyAsync(buf.p, tmp.data(), n * ts, hipMemcpyHostToDevice, M->stream));
  PDEC_HIP(hipStreamSynchronize(M->stream));
  M->fw_dirty = true;
  return PDEC_OK;
}
static int get_flat(Mlp* M, const DevBuf& buf, void* host) {
  const size_t ts = dtype_size(M->dtype), n = M->nparams;
  std::vector<unsigned char> tmp(n * ts);
  PDEC_HIP(hipMemcpyAsync(tmp.data(), buf.p, n * ts, hipMemcpyDeviceToHost, M->stream));
  PDEC_HIP(hipStreamSynchronize(M->stream));
  if (M->dtype == PDEC_F64) internal_to_julia<double>(*M, (const double*)tmp.data(), (double*)host);
  else internal_to_julia<float>(*M, (const float*)tmp.data(), (float*)host);
  return PDEC_OK;
}

extern "C" {

int pdec_mlp_create(pdec_handle* h, int dtype, int n_layers, const int32_t* dims, const int32_t* acts,
                    const void* params_host, int max_cols) {
  PDEC_REQUIRE(h && dims && acts, "pdec_mlp_create: null");
  PDEC_REQUIRE(dtype == PDEC_F32 || dtype == PDEC_F64, "pdec_mlp_create: bad dtype");
  PDEC_REQUIRE(n_layers >= 1 && n_layers <= 8, "pdec_mlp_create: n_layers %d out of [1,8]", n_layers);
  PDEC_REQUIRE(max_cols >= 1, "pdec_mlp_create: max_cols must be >= 1");
  for (int l = 0; l <= n_layers; ++l) PDEC_REQUIRE(dims[l] >= 1 && dims[l] <= 4096, "pdec_mlp_create: dims[%d]=%d", l, dims[l]);
  for (int l = 0; l < n_layers; ++l) PDEC_REQUIRE(acts[l] >= 0 && acts[l] <= 2, "pdec_mlp_create: acts[%d]=%d", l, acts[l]);
  auto M = std::make_unique<Mlp>();
  int rc = M->init(dtype, n_layers, dims, acts, max_cols);
  if (rc) return rc;
  if (params_host && (rc = set_flat(M.get(), M->params, params_host))) return rc;
  *h = register_object(std::move(M));
  return PDEC_OK;
}

int pdec_mlp_num_params(pdec_handle h, int* n) {
  GET_MLP(M, h);
  PDEC_REQUIRE(n, "null");
  *n = M->nparams;
  return PDEC_OK;
}

int pdec_mlp_set_params(pdec_handle h, const void* params_host) {
  GET_MLP(M, h);
  PDEC_REQUIRE(params_host, "pdec_mlp_set_params: null");
  return set_flat(M, M->params, params_host);
}

int pdec_mlp_get_params(pdec_handle h, void* params_host) {
  GET_MLP(M, h);
  PDEC_REQUIRE(params_host, "pdec_mlp_get_params: null");
  return get_flat(M, M->params, params_host);
}

int pdec_mlp_copy(pdec_handle dst, pdec_handle src) {
  GET_MLP(D, dst);
  Mlp* S = lookup_as<Mlp>(src, Kind::Mlp);
  if (!S) { set_error("pdec_mlp_copy: bad src"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(D->dims == S->dims, "pdec_mlp_copy: shape mismatch");
  const int n = D->nparams;
  dim3 grid(cdiv(n, 256)), block(256);
  // order the copy after both handles' streams' prior work: use dst's stream after syncing src's
  if (S->stream != D->stream) PDEC_HIP(hipStreamSynchronize(S->stream));
  if (D->dtype == S->dtype) {
    PDEC_HIP(hipMemcpyAsync(D->params.p, S->params.p, n * dtype_size(D->dtype), hipMemcpyDeviceToDevice, D->stream));
  } else if (D->dtype == PDEC_F64) {
    hipLaunchKernelGGL((cast_copy_kernel<double, float>), grid, block, 0, D->stream, D->params.as<double>(), S->params.as<float>(), n);
  } else {
    hipLaunchKernelGGL((cast_copy_kernel<float, double>), grid, block, 0, D->stream, D->params.as<float>(), S->params.as<double>(), n);
  }
  PDEC_HIP(hipGetLastError());
  D->fw_dirty = true;
  return PDEC_OK;
}

int pdec_mlp_forward(pdec_handle h, const void* x, int cols, void* y_out) {
  GET_MLP(M, h);
  PDEC_REQUIRE(x && y_out, "pdec_mlp_forward: null");
  int rc = DISPATCH(M, M->pack<float>(x, M->dims[0], 0, nullptr, 0, 0, cols), M->pack<double>(x, M->dims[0], 0, nullptr, 0, 0, cols));
  if (rc) return rc;
  rc = DISPATCH(M, M->forward<float>(cols), M->forward<double>(cols));
  if (rc) return rc;
  const int no = M->dims[M->L];
  dim3 grid(cdiv(cols, 256)), block(256);
  if (M->dtype == PDEC_F64)
    hipLaunchKernelGGL((unpack_kernel<double>), grid, block, 0, M->stream, M->H[M->L].as<double>(), cols, no, (double*)y_out);
  else
    hipLaunchKernelGGL((unpack_kernel<float>), grid, block, 0, M->stream, M->H[M->L].as<float>(), cols, no, (float*)y_out);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_mlp_backward(pdec_handle h, const void* x, const void* dy, int cols, void* dx_out, void* grads_out) {
  GET_MLP(M, h);
  PDEC_REQUIRE(x && dy, "pdec_mlp_backward: null");
  int rc = DISPATCH(M, M->pack<float>(x, M->dims[0], 0, nullptr, 0, 0, cols), M->pack<double>(x, M->dims[0], 0, nullptr, 0, 0, cols));
  if (rc) return rc;
  rc = DISPATCH(M, M->forward<float>(cols), M->forward<double>(cols));
  if (rc) return rc;
  rc = DISPATCH(M, M->backward<float>(dy, 0, cols, true, dx_out != nullptr, 1.0),
                M->backward<double>(dy, 0, cols, true, dx_out != nullptr, 1.0));
  if (rc) return rc;
  dim3 grid(cdiv(cols, 256)), block(256);
  if (dx_out) {
    if (M->dtype == PDEC_F64)
      hipLaunchKernelGGL((unpack_kernel<double>), grid, block, 0, M->stream, M->dz[M->dx_index].as<double>(), cols, M->dims[0], (double*)dx_out);
    else
      hipLaunchKernelGGL((unpack_kernel<float>), grid, block, 0, M->stream, M->dz[M->dx_index].as<float>(), cols, M->dims[0], (float*)dx_out);
    PDEC_HIP(hipGetLastError());
  }
  if (grads_out) {
    // grads_out is a DEVICE buffer in the Julia layout: convert through the host (setup/test path)
    const size_t ts = dtype_size(M->dtype);
    std::vector<unsigned char> tmp((size_t)M->nparams * ts);
    rc = get_flat(M, M->grads, tmp.data());
    if (rc) return rc;
    PDEC_HIP(hipMemcpy(grads_out, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
  }
  return PDEC_OK;
}

int pdec_mlp_grad_buffer(pdec_handle h, void** dptr, int* n) {
  GET_MLP(M, h);
  PDEC_REQUIRE(dptr && n, "null");
  *dptr = M->grads.p;
  *n = M->nparams;
  return PDEC_OK;
}

int pdec_adam_step(pdec_handle h, double eta, double beta1, double beta2, double eps) {
  GET_MLP(M, h);
  BpArgs bp;
  int rc = bp_begin(M, beta1, beta2, &bp);
  if (rc) return rc;
  const int n = M->nparams;
  dim3 grid(cdiv(n, 256)), block(256);
  {
    ProfScope ps(M, "adam");
    if (M->dtype == PDEC_F64)
      hipLaunchKernelGGL((adam_kernel<double>), grid, block, 0, M->stream, M->params.as<double>(), M->grads.as<double>(),
                         M->m.as<double>(), M->v.as<double>(), n, eta, beta1, beta2, eps, bp);
    else
      hipLaunchKernelGGL((adam_kernel<float>), grid, block, 0, M->stream, M->params.as<float>(), M->grads.as<float>(),
                         M->m.as<float>(), M->v.as<float>(), n, eta, beta1, beta2, eps, bp);
  }
  PDEC_HIP(hipGetLastError());
  M->fw_dirty = true;
  bp_done(M);
  return PDEC_OK;
}

int pdec_adam_get_state(pdec_handle h, void* m_host, void* v_host, double* beta_pow2) {
  GET_MLP(M, h);
  int rc;
  if (m_host && (rc = get_flat(M, M->m, m_host))) return rc;
  if (v_host && (rc = get_flat(M, M->v, v_host))) return rc;
  if (beta_pow2) {
    beta_pow2[0] = beta_pow2[1] = -1.0;      // not initialised before the first ADAM step
    if (M->bp_init) {
      PDEC_HIP(hipStreamSynchronize(M->stream));
      PDEC_HIP(hipMemcpy(beta_pow2, M->bpd.as<double>() + 2 * M->bp_sel, 16, hipMemcpyDeviceToHost));
    }
  }
  return PDEC_OK;
}

int pdec_adam_set_state(pdec_handle h, const void* m_host, const void* v_host, const double* beta_pow2) {
  GET_MLP(M, h);
  int rc;
  if (m_host && (rc = set_flat(M, M->m, m_host))) return rc;
  if (v_host && (rc = set_flat(M, M->v, v_host))) return rc;
  if (beta_pow2) {
    if (beta_pow2[0] < 0) {
      M->bp_init = false;
    } else {
      if (!M->bpd.p) PDEC_HIP(M->bpd.alloc(4 * sizeof(double)));
      PDEC_HIP(hipStreamSynchronize(M->stream));
      PDEC_HIP(hipMemcpy(M->bpd.as<double>() + 2 * M->bp_sel, beta_pow2, 16, hipMemcpyHostToDevice));
      M->bp_init = true;
    }
  }
  return PDEC_OK;
}

int pdec_polyak(pdec_handle dst, pdec_handle src, double rho) {
  GET_MLP(D, dst);
  Mlp* S = lookup_as<Mlp>(src, Kind::Mlp);
  if (!S) { set_error("pdec_polyak: bad src"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(D->dims == S->dims && D->dtype == S->dtype, "pdec_polyak: shape/dtype mismatch");
  // rho == 1: dest = 1 * dest + 0 * src -- the reference as it runs (its loop body never executes, include/pdeconv.h at this
  // entry point): the target is left untouched, also where src holds an Inf / NaN that 0 * src would turn into NaN
  // (compared as Float32 for fp64 networks too: the kernels below receive r = (float)rho -- the reference holds p = 0.995f0 --, so a
  // rho that rounds to 1.0f IS the frozen case in their arithmetic: (double)r = 1, (double)(1.0f - r) = 0)
  if ((float)rho == 1.0f) return PDEC_OK;
  const int n = D->nparams;
  dim3 grid(cdiv(n, 256)), block(256);
  ProfScope ps(D, "polyak");
  const float r = (float)rho;  // the reference holds p = 0.995f0 and computes (1 - p) in Float32
  if (D->dtype == PDEC_F64) {
    hipLaunchKernelGGL((polyak_kernel<double>), grid, block, 0, D->stream, D->params.as<double>(), S->params.as<double>(), n,
                       (double)r, (double)(1.0f - r));
  } else {
    hipLaunchKernelGGL((polyak_kernel<float>), grid, block, 0, D->stream, D->params.as<float>(), S->params.as<float>(), n, r, 1.0f - r);
  }
  PDEC_HIP(hipGetLastError());
  D->fw_dirty = true;
  return PDEC_OK;
}

int pdec_policy_act(pdec_handle actor, const void* state, const void* noise, int cols, double act_noise,
                    double act_limit, void* actions_out) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(state && actions_out, "pdec_policy_act: null");
  int rc = DISPATCH(M, M->pack<float>(state, M->dims[0], 0, nullptr, 0, 0, cols), M->pack<double>(state, M->dims[0], 0, nullptr, 0, 0, cols));
  if (rc) return rc;
  rc = DISPATCH(M, M->forward<float>(cols), M->forward<double>(cols));
  if (rc) return rc;
  const int no = M->dims[M->L], nrows = M->noise_rows < 0 ? no : M->noise_rows;
  dim3 grid(cdiv(cols, 256)), block(256);
  ProfScope ps(M, "act_noise_clamp");
  if (M->dtype == PDEC_F64)
    hipLaunchKernelGGL((act_noise_clamp_kernel<double>), grid, block, 0, M->stream, M->H[M->L].as<double>(), (const double*)noise,
                       cols, no, nrows, act_noise, act_limit, (double*)actions_out);
  else
    hipLaunchKernelGGL((act_noise_clamp_kernel<float>), grid, block, 0, M->stream, M->H[M->L].as<float>(), (const float*)noise, cols,
                       no, nrows, (float)act_noise, (float)act_limit, (float*)actions_out);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_randn(pdec_handle any_handle, void* dst, size_t n, int dtype, uint64_t seed, uint64_t offset) {
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_randn: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(dst, "pdec_randn: null");
  if (n == 0) return PDEC_OK;
  dim3 grid(cdiv((long)((n + 3) / 4), 256)), block(256);
  if (dtype == PDEC_F64) hipLaunchKernelGGL((randn_kernel<double>), grid, block, 0, o->stream, (double*)dst, n, seed, offset);
  else hipLaunchKernelGGL((randn_kernel<float>), grid, block, 0, o->stream, (float*)dst, n, seed, offset);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ DDPG
template <class T>
static int critic_grads_t(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                          const void* snext, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev) {
  const int ns = At->dims[0], na = At->dims[At->L];
  PDEC_REQUIRE(C->dims[0] == ns + na && Ct->dims[0] == ns + na && C->dims[C->L] == 1 && Ct->dims[Ct->L] == 1,
               "ddpg: critic must map ns+na -> 1");
  int rc;
  // a' = At(s')                                              src/PDEagent.jl:385
  if ((rc = At->pack<T>(snext, ns, 0, nullptr, 0, 0, Bu))) return rc;
  if ((rc = At->forward<T>(Bu))) return rc;
  // qt = Ct(vcat(s', a'))                                    :386
  if (At->stream != Ct->stream) PDEC_HIP(hipStreamSynchronize(At->stream));
  if ((rc = Ct->pack<T>(snext, ns, 0, At->H[At->L].p, na, 1, Bu))) return rc;
  if ((rc = Ct->forward<T>(Bu))) return rc;
  // q = C(vcat(s, a))                                        :392
  if (Ct->stream != C->stream) PDEC_HIP(hipStreamSynchronize(Ct->stream));
  if ((rc = C->pack<T>(s, ns, 0, a, na, 0, Bu))) return rc;
  if ((rc = C->forward<T>(Bu))) return rc;
  T* stats = C->scratch.as<T>();
  const T* q = C->H[C->L].as<T>();
  const T* qt = Ct->H[Ct->L].as<T>();
  // gamma is Float32 in the reference (y = 0.99f0, KSSetup.jl:64)
  const T gm = (T)(float)gamma;
  {
    ProfScope ps(C, "ddpg_loss");
    hipLaunchKernelGGL((ddpg_stats_kernel<T>), dim3(1), dim3(256), 0, C->stream, q, qt, (const T*)r, (const T*)t, Bu, gm, stats);
    if (!quirk)
      hipLaunchKernelGGL((ddpg_diag_loss_kernel<T>), dim3(1), dim3(256), 0, C->stream, q, qt, (const T*)r, (const T*)t, Bu, gm, stats);
    // dq goes straight into dz of the identity output layer (feature-major [1][Bu])
    hipLaunchKernelGGL((ddpg_critic_dq_kernel<T>), dim3(cdiv(Bu, 256)), dim3(256), 0, C->stream, q, qt, (const T*)r, (const T*)t, Bu,
                       gm, quirk, stats, C->dy_buf<T>(Bu), (T*)loss_dev);
    PDEC_HIP(hipGetLastError());
  }
  return C->backward<T>(C->dy_buf<T>(Bu), 1, Bu, true, false, grad_scale);
}

template <class T>
static int actor_grads_t(Mlp* A, Mlp* C, const void* s, int Bu, double grad_scale, void* loss_dev) {
  const int ns = A->dims[0], na = A->dims[A->L];
  int rc;
  if ((rc = A->pack<T>(s, ns, 0, nullptr, 0, 0, Bu))) return rc;
  if ((rc = A->forward<T>(Bu))) return rc;
  if (A->stream != C->stream) PDEC_HIP(hipStreamSynchronize(A->stream));
  if ((rc = C->pack<T>(s, ns, 0, A->H[A->L].p, na, 1, Bu))) return rc;
  if ((rc = C->forward<T>(Bu))) return rc;
  T* stats = C->scratch.as<T>();
  {
    ProfScope ps(C, "ddpg_loss");
    hipLaunchKernelGGL((ddpg_stats_kernel<T>), dim3(1), dim3(256), 0, C->stream, C->H[C->L].as<T>(), (const T*)nullptr,
                       (const T*)nullptr, (const T*)nullptr, Bu, (T)0, stats);
    if (loss_dev) hipLaunchKernelGGL((neg_copy_kernel<T>), dim3(1), dim3(1), 0, C->stream, (T*)loss_dev, stats + 4);
    hipLaunchKernelGGL((fill_kernel<T>), dim3(cdiv(Bu, 256)), dim3(256), 0, C->stream, C->dy_buf<T>(Bu), Bu, (T)(-1.0 / Bu));
    PDEC_HIP(hipGetLastError());
  }
  // d(-mean q)/d[s;a] through the critic, no critic weight gradients   :402-409
  if ((rc = C->backward<T>(C->dy_buf<T>(Bu), 1, Bu, false, true, 1.0))) return rc;
  if (A->stream != C->stream) PDEC_HIP(hipStreamSynchronize(C->stream));
  const T* dA = C->dz[C->dx_index].as<T>() + (size_t)ns * Bu;  // rows ns.. of dX0 = gradient w.r.t. A(s)
  return A->backward<T>(dA, 1, Bu, true, false, grad_scale);
}

extern "C" {

int pdec_ddpg_critic_grads(pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s,
                           const void* a, const void* r, const void* t, const void* snext, int Bu, double gamma,
                           int quirk, double grad_scale, void* critic_loss_dev) {
  GET_MLP(A, hA);
  GET_MLP(C, hC);
  GET_MLP(At, hAt);
  GET_MLP(Ct, hCt);
  PDEC_REQUIRE(s && a && r && t && snext && Bu >= 1, "pdec_ddpg_critic_grads: null/empty batch");
  PDEC_REQUIRE(A->dtype == C->dtype && At->dtype == C->dtype && Ct->dtype == C->dtype, "ddpg: dtype mismatch");
  PDEC_REQUIRE(At->dims == A->dims && Ct->dims == C->dims, "ddpg: target networks must have the behaviour networks' shapes");
  int rc;
  if (fused_supported(A, C) && A->stream == C->stream && At->stream == C->stream && Ct->stream == C->stream)
    rc = fused_critic_grads(A, C, At, Ct, s, a, r, t, snext, Bu, (double)(float)gamma, quirk, grad_scale, critic_loss_dev, nullptr);
  else if (fused2_supported(A, C) && A->stream == C->stream && At->stream == C->stream && Ct->stream == C->stream)
    rc = fused2_critic_grads(A, C, At, Ct, s, a, r, t, snext, Bu, (double)(float)gamma, quirk, grad_scale, critic_loss_dev, nullptr);
  else
    rc = C->dtype == PDEC_F64
             ? critic_grads_t<double>(A, C, At, Ct, s, a, r, t, snext, Bu, gamma, quirk, grad_scale, critic_loss_dev)
             : critic_grads_t<float>(A, C, At, Ct, s, a, r, t, snext, Bu, gamma, quirk, grad_scale, critic_loss_dev);
  if (rc == PDEC_OK && C->reduce_event) {     // see pdec_ddpg_actor_grads
    PDEC_HIP(hipEventRecord(C->reduce_event, C->stream));
    C->reduce_event = nullptr;
  }
  return rc;
}

int pdec_ddpg_actor_grads(pdec_handle hA, pdec_handle hC, const void* s, int Bu, double grad_scale, void* actor_loss_dev) {
  GET_MLP(A, hA);
  GET_MLP(C, hC);
  PDEC_REQUIRE(s && Bu >= 1, "pdec_ddpg_actor_grads: null/empty batch");
  PDEC_REQUIRE(A->dtype == C->dtype, "ddpg: dtype mismatch");
  int rc;
  if (fused_supported(A, C) && A->stream == C->stream) rc = fused_actor_grads(A, C, nullptr, s, Bu, grad_scale, actor_loss_dev, nullptr);
  else if (fused2_supported(A, C) && A->stream == C->stream) rc = fused2_actor_grads(A, C, nullptr, s, Bu, grad_scale, actor_loss_dev, nullptr);
  else rc = C->dtype == PDEC_F64 ? actor_grads_t<double>(A, C, s, Bu, grad_scale, actor_loss_dev)
                                 : actor_grads_t<float>(A, C, s, Bu, grad_scale, actor_loss_dev);
  // a reduce event (pdec_mlp_set_reduce_event) that the path taken did not put on its reduction launch: recorded behind it
  if (rc == PDEC_OK && A->reduce_event) {
    PDEC_HIP(hipEventRecord(A->reduce_event, A->stream));
    A->reduce_event = nullptr;
  }
  return rc;
}

int pdec_ddpg_update(pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s, const void* a,
                     const void* r, const void* t, const void* snext, int Bu, double gamma, double rho, int quirk,
                     double eta_actor, double eta_critic, double* actor_loss, double* critic_loss) {
  GET_MLP(C, hC);
  GET_MLP(A, hA);
  const size_t ts = dtype_size(C->dtype);
  char* losses = C->scratch.as<char>() + 16 * ts;  // two device scalars behind the stats
  int rc = pdec_ddpg_critic_grads(hA, hC, hAt, hCt, s, a, r, t, snext, Bu, gamma, quirk, 1.0, losses);
  if (rc) return rc;
  if ((rc = pdec_adam_step(hC, eta_critic, 0.9, 0.999, 1e-8))) return rc;           // :400
  if ((rc = pdec_ddpg_actor_grads(hA, hC, s, Bu, 1.0, losses + ts))) return rc;
  if (A->stream != C->stream) PDEC_HIP(hipStreamSynchronize(C->stream));
  if ((rc = pdec_adam_step(hA, eta_actor, 0.9, 0.999, 1e-8))) return rc;            // :412
  if ((rc = pdec_polyak(hAt, hA, rho))) return rc;                                  // :415-417
  if ((rc = pdec_polyak(hCt, hC, rho))) return rc;
  if (actor_loss || critic_loss) {
    unsigned char buf[16];
    PDEC_HIP(hipStreamSynchronize(A->stream));
    PDEC_HIP(hipMemcpyAsync(buf, losses, 2 * ts, hipMemcpyDeviceToHost, C->stream));
    PDEC_HIP(hipStreamSynchronize(C->stream));
    if (C->dtype == PDEC_F64) {
      if (critic_loss) *critic_loss = ((double*)buf)[0];
      if (actor_loss) *actor_loss = ((double*)buf)[1];
    } else {
      if (critic_loss) *critic_loss = ((float*)buf)[0];
      if (actor_loss) *actor_loss = ((float*)buf)[1];
    }
  }
  return PDEC_OK;
}


int pdec_adam_polyak_step(pdec_handle h, pdec_handle h_target, double eta, double beta1, double beta2, double eps,
                          double rho) {
  GET_MLP(M, h);
  Mlp* T = lookup_as<Mlp>(h_target, Kind::Mlp);
  if (!T) { set_error("pdec_adam_polyak_step: bad target handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(T->dims == M->dims && T->dtype == M->dtype, "pdec_adam_polyak_step: shape/dtype mismatch");
  if (fused_net_supported(M) && M->stream == T->stream) {
    const AdamPolyak ap{eta, beta1, beta2, eps, rho};
    return fused_adam_polyak(M, T, ap);
  }
  if (fused2_net_supported(M) && M->stream == T->stream) {
    const AdamPolyak ap{eta, beta1, beta2, eps, rho};
    return fused2_adam_polyak(M, T, ap);
  }
  int rc = pdec_adam_step(h, eta, beta1, beta2, eps);
  if (rc) return rc;
  if (M->stream != T->stream) PDEC_HIP(hipStreamSynchronize(M->stream));
  return pdec_polyak(h_target, h, rho);
}

// phase bit 0: critic half (critic pass, reduce + ADAM(C) + Polyak(Ct)); bit 1: actor half (actor pass with the
// updated critic, reduce + ADAM(A) + Polyak(At)).  The split lets a caller order the actor half behind a
// concurrent reader of the actor's weights (e.g. the acting kernel of the same control step on another stream).
static int ddpg_update_phases(int phase, pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s,
                              const void* a, const void* r, const void* t, const void* snext, int Bu, double gamma,
                              double rho, int quirk, double eta_actor, double eta_critic, void* losses_dev) {
  GET_MLP(A, hA);
  GET_MLP(C, hC);
  GET_MLP(At, hAt);
  GET_MLP(Ct, hCt);
  PDEC_REQUIRE(s && Bu >= 1, "pdec_ddpg_update_async: null/empty batch");
  PDEC_REQUIRE(!(phase & 1) || (a && r && t && snext), "pdec_ddpg_update_async: null batch array");
  PDEC_REQUIRE(A->dtype == C->dtype && At->dtype == C->dtype && Ct->dtype == C->dtype, "ddpg: dtype mismatch");
  PDEC_REQUIRE(At->dims == A->dims && Ct->dims == C->dims, "ddpg: target networks must have the behaviour networks' shapes");
  const size_t ts = dtype_size(C->dtype);
  char* l0 = (char*)losses_dev;
  char* l1 = l0 ? l0 + ts : nullptr;
  const bool one_stream = A->stream == C->stream && At->stream == C->stream && Ct->stream == C->stream;
  int rc;
  if (fused_supported(A, C) && one_stream) {
    // 4 launches: critic pass, reduce+ADAM(C)+Polyak(Ct), actor pass (updated critic), reduce+ADAM(A)+Polyak(At)
    const AdamPolyak apc{eta_critic, 0.9, 0.999, 1e-8, rho}, apa{eta_actor, 0.9, 0.999, 1e-8, rho};
    if ((phase & 1) &&
        (rc = fused_critic_grads(A, C, At, Ct, s, a, r, t, snext, Bu, (double)(float)gamma, quirk, 1.0, l0, &apc)))
      return rc;
    if (phase & 2) return fused_actor_grads(A, C, At, s, Bu, 1.0, l1, &apa);
    return PDEC_OK;
  }
  if (fused2_supported(A, C) && one_stream && Bu >= 64) {      // 2-layer nets, large batches: same 4 launches
    const AdamPolyak apc{eta_critic, 0.9, 0.999, 1e-8, rho}, apa{eta_actor, 0.9, 0.999, 1e-8, rho};
    if ((phase & 1) &&
        (rc = fused2_critic_grads(A, C, At, Ct, s, a, r, t, snext, Bu, (double)(float)gamma, quirk, 1.0, l0, &apc)))
      return rc;
    if (phase & 2) return fused2_actor_grads(A, C, At, s, Bu, 1.0, l1, &apa);
    return PDEC_OK;
  }
  if (phase & 1) {
    if ((rc = pdec_ddpg_critic_grads(hA, hC, hAt, hCt, s, a, r, t, snext, Bu, gamma, quirk, 1.0, l0))) return rc;
    if ((rc = pdec_adam_step(hC, eta_critic, 0.9, 0.999, 1e-8))) return rc;         // :400
    if ((rc = pdec_polyak(hCt, hC, rho))) return rc;                                // :415-417 (critic pair)
  }
  if (phase & 2) {
    if ((rc = pdec_ddpg_actor_grads(hA, hC, s, Bu, 1.0, l1))) return rc;
    if (A->stream != C->stream) PDEC_HIP(hipStreamSynchronize(C->stream));
    if ((rc = pdec_adam_step(hA, eta_actor, 0.9, 0.999, 1e-8))) return rc;          // :412
    if ((rc = pdec_polyak(hAt, hA, rho))) return rc;                                // :415-417 (actor pair)
  }
  return PDEC_OK;
}

int pdec_ddpg_update_async(pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s,
                           const void* a, const void* r, const void* t, const void* snext, int Bu, double gamma,
                           double rho, int quirk, double eta_actor, double eta_critic, void* losses_dev) {
  return ddpg_update_phases(3, hA, hC, hAt, hCt, s, a, r, t, snext, Bu, gamma, rho, quirk, eta_actor, eta_critic, losses_dev);
}
int pdec_ddpg_update_critic_async(pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s,
                                  const void* a, const void* r, const void* t, const void* snext, int Bu, double gamma,
                                  double rho, int quirk, double eta_critic, void* losses_dev) {
  return ddpg_update_phases(1, hA, hC, hAt, hCt, s, a, r, t, snext, Bu, gamma, rho, quirk, 0.0, eta_critic, losses_dev);
}
int pdec_ddpg_update_actor_async(pdec_handle hA, pdec_handle hC, pdec_handle hAt, pdec_handle hCt, const void* s, int Bu,
                                 double rho, double eta_actor, void* losses_dev) {
  return ddpg_update_phases(2, hA, hC, hAt, hCt, s, nullptr, nullptr, nullptr, nullptr, Bu, 0.0, rho, 0, eta_actor, 0.0,
                            losses_dev);
}


}  // extern "C"

// ---- acting for FEW columns in one launch (the reference's own shape: one trajectory, A columns; src/PDEagent.jl:175-209)
// The generic path is pack + one GEMM launch per layer + randn + noise / clamp -- and, when the environment computes in fp64
// while the networks are Float32 (as in the reference), a promoted copy of the parameters first: seven launches of ~5.5 us for
// a few hundred multiply-adds.  This kernel does the same arithmetic in ONE workgroup: parameters of type TP promoted to T on
// the fly (exact), activations [feature][column] in LDS, per element acc = sum_k w x in ascending k from zero, then + bias and
// the activation -- the order of gemm_kernel's epilogue --, the Philox / Box-Muller draw of randn_kernel for element
// column * outputs + row, then v += noise * act_noise and the clamp of act_noise_clamp_kernel.
#define SMALL_ACT_MAXL 4
struct SmallActArgs {
  int L, cols, maxw, learning, nrows;
  int dims[SMALL_ACT_MAXL + 1], acts[SMALL_ACT_MAXL], woff[SMALL_ACT_MAXL], boff[SMALL_ACT_MAXL];
  const void* p;
  double act_noise, lim;
  uint64_t seed, offset;
  const uint64_t* ctr_cur;
  uint64_t* ctr_next;
  uint64_t ctr_inc;
};
// the kernel's body; returns the LDS buffer that holds the finished actions as element i = column * outputs + row (what `out`
// receives), for a caller that goes on with them inside the same launch (step_glue_kernel)
template <class T, class TP>
__device__ __forceinline__ T* small_act_body(const SmallActArgs& g, const T* __restrict__ state, T* __restrict__ out,
                                             unsigned char* smem) {
  T* X0 = reinterpret_cast<T*>(smem);
  T* X1 = X0 + (size_t)g.maxw * g.cols;
  const TP* p = static_cast<const TP*>(g.p);
  const int tid = threadIdx.x, cols = g.cols;
  uint64_t offset = g.offset;
  if (g.ctr_cur) {       // device-resident noise counter (pdec_policy_act_rng_dev)
    offset += *g.ctr_cur;
    if (tid == 0) *g.ctr_next = offset + g.ctr_inc;
  }
  const int ns = g.dims[0];
  for (int i = tid; i < ns * cols; i += 256) {
    const int c = i / ns, k = i - c * ns;
    X0[k * cols + c] = state[i];
  }
  __syncthreads();
  T* xin = X0;
  T* xout = X1;
  for (int l = 0; l < g.L; ++l) {
    const int in = g.dims[l], on = g.dims[l + 1];
    const TP* W = p + g.woff[l];
    const TP* b = p + g.boff[l];
    for (int i = tid; i < on * cols; i += 256) {
      const int j = i / cols, c = i - j * cols;
      T acc = 0;
      for (int k = 0; k < in; ++k) acc += (T)W[j * in + k] * xin[k * cols + c];
      xout[j * cols + c] = apply_act<T>(acc + (T)b[j], g.acts[l]);
    }
    __syncthreads();
    T* t = xin; xin = xout; xout = t;
  }
  const int no = g.dims[g.L];
  const T an = (T)g.act_noise, lim = (T)g.lim;
  for (int i = tid; i < no * cols; i += 256) {       // i = column * outputs + row: the element index of the noise stream
    const int c = i / no, f = i - c * no;
    T v = xin[f * cols + c];
    if (g.learning && f < g.nrows) {
      const uint64_t ctr = offset + (uint64_t)(i >> 2);
      uint32_t ph[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
      philox4x32(ph, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
      const int h = (i >> 1) & 1;
      const double sc = 1.0 / 4294967296.0;
      const double u1 = ((double)ph[2 * h] + 0.5) * sc, u2 = ((double)ph[2 * h + 1] + 0.5) * sc;
      const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
      const T z = (T)((i & 1) ? rad * sin(ang) : rad * cos(ang));
      v += z * an;
    }
    v = v < -lim ? -lim : (v > lim ? lim : v);
    out[i] = v;
    xout[i] = v;
  }
  return xout;
}
template <class T, class TP>
__global__ __launch_bounds__(256) void small_act_kernel(SmallActArgs g, const T* __restrict__ state, T* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char small_act_smem[];
  (void)small_act_body<T, TP>(g, state, out, small_act_smem);
}

// ---- the glue between two control steps of a single-trajectory training loop, ONE launch instead of three (round 6):
//   POST_ACT push of the step that just ran (reward, terminal; src/PDEagent.jl:276-289; raises the episode halt flag when that
//   step ended the episode -- replay_push_rt_kernel), agent(env) for the next step (small_act_kernel; or the zero action of the
//   start policy), PRE_ACT push of the next step's (state, action) (:254-274; replay_push2_kernel: skipped once the episode has
//   ended).  Same arithmetic and the same stores as the three launches; each of them was ~5 us of launch latency on the chain
//   act -> push -> update / env step -> push of the reference-shaped loop.
struct StepGlueArgs {
  float *tr, *tt;              // POST_ACT push: reward / terminal traces, capacity cap_rt, first slot start_rt, n_rt values (0: none)
  const void* r;
  const int32_t* done;
  int cols_per_traj, force;
  long long cap_rt, start_rt, n_rt;
  int act_mode;                // 0: no acting, 1: the actor, 2: the zero action
  SmallActArgs act;
  const void* state;           // [cols][ns] of the environment's type: acting input and the PRE_ACT push's state rows
  void* out;                   // [cols][na]
  float *ts, *ta;              // PRE_ACT push: state / action traces, capacity cap_sa rows, first row start_sa, n_sa rows (0: none)
  int ns, na;
  long long cap_sa, start_sa, n_sa;
  int* halt;
  LaunchSync sync;             // pdec_set_launch_sync on the actor's handle: wait before the first read, signal behind the last store
};
template <class T, class TP>
__global__ __launch_bounds__(256) void step_glue_kernel(StepGlueArgs g) {
  extern __shared__ __align__(16) unsigned char small_act_smem[];
  const int tid = threadIdx.x;
  launch_sync_wait(g.sync);           // (the env step that wrote reward / done / state, on another stream)
  const bool was = g.halt && *g.halt;
  const bool ended = was || (g.halt && g.n_rt && g.done && g.done[0] != 0);
  if (g.n_rt && !was) {
    const T* r = static_cast<const T*>(g.r);
    for (long long i = tid; i < g.n_rt; i += 256) {
      const long long slot = (g.start_rt + i) % g.cap_rt;
      g.tr[slot] = (float)r[i];
      g.tt[slot] = (g.force || (g.done && g.done[i / g.cols_per_traj] != 0)) ? 1.f : 0.f;
    }
  }
  __syncthreads();                    // every thread has read *halt before one of them writes it
  if (tid == 0 && g.halt && !was && ended) *g.halt = 1;
  const T* state = static_cast<const T*>(g.state);
  T* out = static_cast<T*>(g.out);
  const T* acts = nullptr;            // the actions of this launch in LDS, element row * na + c
  if (g.act_mode == 1) {
    acts = small_act_body<T, TP>(g.act, state, out, small_act_smem);
  } else if (g.act_mode == 2) {
    for (long long i = tid; i < (long long)g.act.cols * g.na; i += 256) out[i] = (T)0;
  }
  if (g.n_sa && !ended) {
    __syncthreads();                  // the actions are in LDS
    const long long na_ = g.n_sa * g.ns, nb_ = g.n_sa * g.na;
    for (long long i = tid; i < na_ + nb_; i += 256) {
      if (i < na_) {
        const long long row = i / g.ns, c = i - row * g.ns;
        g.ts[((g.start_sa + row) % g.cap_sa) * g.ns + c] = (float)state[i];
      } else {
        const long long j = i - na_, row = j / g.na, c = j - row * g.na;
        g.ta[((g.start_sa + row) % g.cap_sa) * g.na + c] = acts ? (float)acts[j] : 0.f;
      }
    }
  }
  launch_sync_done(g.sync);
}

// does the single-launch form serve this actor at `cols` columns of type `dtype`?  (the fused MFMA acting kernels keep fp32
// actors at fp32 states; PDEC_SMALL_ACT=0: the generic launch sequence, for A/B tests)
static bool small_act_ok(const Mlp* M, int dtype, int cols) {
  static const bool off = [] { const char* e = getenv("PDEC_SMALL_ACT"); return e && e[0] == '0'; }();
  if (off || M->L > SMALL_ACT_MAXL || cols < 1) return false;
  if (!(M->dtype == dtype || (M->dtype == PDEC_F32 && dtype == PDEC_F64))) return false;
  int maxw = 1;
  for (int l = 0; l <= M->L; ++l) maxw = std::max(maxw, M->dims[l]);
  return (size_t)2 * maxw * cols * dtype_size(dtype) <= 48 * 1024;
}

static int small_act(Mlp* M, int dtype, const void* state, int cols, double act_noise, double act_limit, int learning, uint64_t seed,
                     uint64_t offset, void* actions_out, const uint64_t* ctr_cur, uint64_t* ctr_next, uint64_t ctr_inc) {
  SmallActArgs g{};
  g.L = M->L; g.cols = cols; g.learning = learning;
  g.maxw = 1;
  for (int l = 0; l <= M->L; ++l) { g.dims[l] = M->dims[l]; g.maxw = std::max(g.maxw, M->dims[l]); }
  for (int l = 0; l < M->L; ++l) { g.acts[l] = M->acts[l]; g.woff[l] = (int)M->w_off[l]; g.boff[l] = (int)M->b_off[l]; }
  g.nrows = M->noise_rows < 0 ? M->dims[M->L] : M->noise_rows;
  g.p = M->params.p;
  g.act_noise = act_noise; g.lim = act_limit; g.seed = seed; g.offset = offset;
  g.ctr_cur = ctr_cur; g.ctr_next = ctr_next; g.ctr_inc = ctr_inc;
  const size_t lds = (size_t)2 * g.maxw * cols * dtype_size(dtype);
  ProfScope ps(M, "small_act");
  if (dtype == PDEC_F64 && M->dtype == PDEC_F32)
    hipLaunchKernelGGL((small_act_kernel<double, float>), dim3(1), dim3(256), lds, M->stream, g, (const double*)state, (double*)actions_out);
  else if (dtype == PDEC_F64)
    hipLaunchKernelGGL((small_act_kernel<double, double>), dim3(1), dim3(256), lds, M->stream, g, (const double*)state, (double*)actions_out);
  else
    hipLaunchKernelGGL((small_act_kernel<float, float>), dim3(1), dim3(256), lds, M->stream, g, (const float*)state, (float*)actions_out);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

extern "C" {

static int policy_act_rng_impl(pdec_handle actor, Mlp* M, const void* state, int cols, double act_noise, double act_limit,
                               int learning, uint64_t seed, uint64_t offset, void* actions_out, const uint64_t* ctr_cur,
                               uint64_t* ctr_next, uint64_t ctr_inc) {
  if (fused_net_supported(M) && M->dims[M->L] == 1 && M->dims[1] <= 31)
    return fused_policy_act(M, state, cols, act_noise, act_limit, learning, seed, offset, actions_out, ctr_cur, ctr_next, ctr_inc);
  if (fused2_act_supported(M, cols))
    return fused2_policy_act(M, state, cols, act_noise, act_limit, learning, seed, offset, actions_out, ctr_cur, ctr_next, ctr_inc);
  if (small_act_ok(M, M->dtype, cols))
    return small_act(M, M->dtype, state, cols, act_noise, act_limit, learning, seed, offset, actions_out, ctr_cur, ctr_next, ctr_inc);
  void* noise = nullptr;
  const size_t n = (size_t)cols * M->dims[M->L];
  if (learning || ctr_cur) {
    if (M->noise.bytes < n * dtype_size(M->dtype)) PDEC_HIP(M->noise.alloc(n * dtype_size(M->dtype)));
    const dim3 grid((unsigned)((n + 1023) / 1024)), block(256);
    ProfScope ps(M, "randn");
    if (M->dtype == PDEC_F64)
      hipLaunchKernelGGL((randn_kernel<double>), grid, block, 0, M->stream, M->noise.as<double>(), n, seed, offset, ctr_cur, ctr_next, ctr_inc);
    else
      hipLaunchKernelGGL((randn_kernel<float>), grid, block, 0, M->stream, M->noise.as<float>(), n, seed, offset, ctr_cur, ctr_next, ctr_inc);
    PDEC_HIP(hipGetLastError());
    if (learning) noise = M->noise.p;
  }
  return pdec_policy_act(actor, state, noise, cols, act_noise, act_limit, actions_out);
}

int pdec_policy_act_rng(pdec_handle actor, const void* state, int cols, double act_noise, double act_limit,
                        int learning, uint64_t seed, uint64_t offset, void* actions_out) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(state && actions_out && cols >= 1, "pdec_policy_act_rng: null/empty");
  return policy_act_rng_impl(actor, M, state, cols, act_noise, act_limit, learning, seed, offset, actions_out, nullptr, nullptr, 0);
}

// agent(env) for an environment that computes in `state_dtype` with an actor of another parameter type (the reference: fp64
// fields, Float32 networks) WITHOUT a promoted copy of the actor: *served = 1 and the action is enqueued when the single-launch
// form covers the case, *served = 0 (nothing enqueued) otherwise -- the caller then acts through a promoted clone.
int pdec_policy_act_rng_as(pdec_handle actor, int state_dtype, const void* state, int cols, double act_noise, double act_limit,
                           int learning, uint64_t seed, uint64_t offset, void* actions_out, int* served) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(served, "pdec_policy_act_rng_as: null");
  PDEC_REQUIRE(state_dtype == PDEC_F32 || state_dtype == PDEC_F64, "pdec_policy_act_rng_as: bad dtype %d", state_dtype);
  *served = 0;
  if (!small_act_ok(M, state_dtype, cols) || M->dtype == state_dtype) return PDEC_OK;
  PDEC_REQUIRE(state && actions_out, "pdec_policy_act_rng_as: null");
  *served = 1;
  return small_act(M, state_dtype, state, cols, act_noise, act_limit, learning, seed, offset, actions_out, nullptr, nullptr, 0);
}

// does pdec_step_glue serve this case (see there)?
static bool step_glue_served(const Mlp* M, const Object* o, int dtype, int act_mode, int cols, int64_t n_rt) {
  return o->stream == M->stream && n_rt <= 256 && !(act_mode == 1 && (!small_act_ok(M, dtype, cols) || M->dtype == dtype));
}
int pdec_step_glue_served(pdec_handle actor, pdec_handle trajectory_handle, int dtype, int act_mode, int cols, int64_t n_rt, int* served) {
  GET_MLP(M, actor);
  Object* o = lookup(trajectory_handle);
  if (!o) { set_error("pdec_step_glue_served: bad trajectory handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(served && (dtype == PDEC_F32 || dtype == PDEC_F64), "pdec_step_glue_served: bad argument");
  *served = step_glue_served(M, o, dtype, act_mode, cols, n_rt) ? 1 : 0;
  return PDEC_OK;
}

int pdec_step_glue(pdec_handle actor, pdec_handle trajectory_handle, int dtype, const void* reward, const int32_t* done_flags,
                   int cols_per_traj, int force_terminal, void* reward_trace, void* terminal_trace, int64_t capacity,
                   int64_t start_rt, int64_t n_rt, int act_mode, const void* state, int cols, double act_noise, double act_limit,
                   uint64_t seed, uint64_t offset, void* actions_out, void* state_trace, void* action_trace,
                   int64_t capacity_rows, int64_t start_sa, int64_t n_sa, pdec_handle done_event, int* served) {
  GET_MLP(M, actor);
  Object* o = lookup(trajectory_handle);
  if (!o) { set_error("pdec_step_glue: bad trajectory handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(served, "pdec_step_glue: null");
  PDEC_REQUIRE(dtype == PDEC_F32 || dtype == PDEC_F64, "pdec_step_glue: bad dtype %d", dtype);
  PDEC_REQUIRE(act_mode >= 0 && act_mode <= 2 && n_rt >= 0 && n_sa >= 0, "pdec_step_glue: bad argument");
  *served = 0;
  // served where the three launches it stands for would be the single-workgroup ones: the few-column acting kernel of
  // pdec_policy_act_rng_as (state of another type than the networks), single-block pushes, one stream
  if (!step_glue_served(M, o, dtype, act_mode, cols, n_rt)) {
    const bool had = M->sync.wait || M->sync.done;
    M->sync = LaunchSync{};
    PDEC_REQUIRE(!had, "pdec_step_glue: a launch sync is set (pdec_set_launch_sync) and this case is not served");
    return PDEC_OK;
  }
  PDEC_REQUIRE(!n_rt || (reward && reward_trace && terminal_trace && capacity >= 1 && n_rt <= capacity && cols_per_traj >= 1 && start_rt >= 0),
               "pdec_step_glue: bad POST_ACT push");
  PDEC_REQUIRE(!n_sa || (state && state_trace && action_trace && capacity_rows >= 1 && n_sa <= capacity_rows && start_sa >= 0),
               "pdec_step_glue: bad PRE_ACT push");
  PDEC_REQUIRE(act_mode == 0 || (actions_out && (act_mode == 2 || (state && cols >= 1))), "pdec_step_glue: bad acting arguments");
  PDEC_REQUIRE(act_mode == 0 || !n_sa || n_sa == cols, "pdec_step_glue: the PRE_ACT push takes the acting call's columns");
  StepGlueArgs g{};
  g.tr = (float*)reward_trace; g.tt = (float*)terminal_trace; g.r = reward; g.done = done_flags;
  g.cols_per_traj = cols_per_traj; g.force = force_terminal; g.cap_rt = capacity; g.start_rt = start_rt; g.n_rt = n_rt;
  g.act_mode = act_mode; g.state = state; g.out = actions_out;
  g.ts = (float*)state_trace; g.ta = (float*)action_trace; g.ns = M->dims[0]; g.na = M->dims[M->L];
  g.cap_sa = capacity_rows; g.start_sa = start_sa; g.n_sa = n_sa;
  g.halt = o->halt;
  g.sync = M->sync;
  M->sync = LaunchSync{};
  SmallActArgs& a = g.act;
  a.L = M->L; a.cols = cols; a.learning = 1;
  a.maxw = 1;
  for (int l = 0; l <= M->L; ++l) { a.dims[l] = M->dims[l]; a.maxw = std::max(a.maxw, M->dims[l]); }
  for (int l = 0; l < M->L; ++l) { a.acts[l] = M->acts[l]; a.woff[l] = (int)M->w_off[l]; a.boff[l] = (int)M->b_off[l]; }
  a.nrows = M->noise_rows < 0 ? M->dims[M->L] : M->noise_rows;
  a.p = M->params.p;
  a.act_noise = act_noise; a.lim = act_limit; a.seed = seed; a.offset = offset;
  const size_t lds = act_mode == 1 ? (size_t)2 * a.maxw * cols * dtype_size(dtype) : 16;
  hipEvent_t ev = nullptr;
  if (done_event) {
    ev = pdec::event_native(done_event);
    PDEC_REQUIRE(ev, "pdec_step_glue: bad event handle");
  }
  *served = 1;
  ProfScope ps(M, "step_glue");
  // (done_event rides on the launch as the completion event of its dispatch packet, like pdec_mlp_set_stop_event's)
  if (dtype == PDEC_F64 && M->dtype == PDEC_F32) hipExtLaunchKernelGGL((step_glue_kernel<double, float>), dim3(1), dim3(256), lds, M->stream, nullptr, ev, 0, g);
  else if (dtype == PDEC_F64) hipExtLaunchKernelGGL((step_glue_kernel<double, double>), dim3(1), dim3(256), lds, M->stream, nullptr, ev, 0, g);
  else hipExtLaunchKernelGGL((step_glue_kernel<float, float>), dim3(1), dim3(256), lds, M->stream, nullptr, ev, 0, g);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static int ensure_noise_ctr(Mlp* M) {
  if (!M->noise_ctr.p) {
    PDEC_HIP(M->noise_ctr.alloc(2 * sizeof(uint64_t)));
    // blocking copy, NOT hipMemset: a memset is queued on the null stream and returns; the acting kernel runs on a
    // non-blocking stream that the null stream does not order, so on a busy device it could read the counter first
    const uint64_t zero[2] = {0, 0};
    PDEC_HIP(hipMemcpy(M->noise_ctr.p, zero, sizeof(zero), hipMemcpyHostToDevice));
    M->nc_sel = 0;
  }
  return PDEC_OK;
}

int pdec_policy_act_rng_dev(pdec_handle actor, const void* state, int cols, double act_noise, double act_limit,
                            int learning, uint64_t seed, void* actions_out) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(state && actions_out && cols >= 1, "pdec_policy_act_rng_dev: null/empty");
  int rc = ensure_noise_ctr(M);
  if (rc) return rc;
  uint64_t* c = M->noise_ctr.as<uint64_t>();
  const uint64_t inc = learning ? ((uint64_t)cols * M->dims[M->L] + 3) / 4 : 0;   // counters one call consumes (4 normals each)
  rc = policy_act_rng_impl(actor, M, state, cols, act_noise, act_limit, learning, seed, 0, actions_out, c + M->nc_sel,
                           c + (M->nc_sel ^ 1), inc);
  if (rc) return rc;
  flip(M->nc_sel);
  return PDEC_OK;
}

int pdec_mlp_set_noise_rows(pdec_handle actor, int rows) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(rows == -1 || (rows >= 1 && rows <= M->dims[M->L]), "pdec_mlp_set_noise_rows: %d of %d outputs", rows, M->dims[M->L]);
  M->noise_rows = rows == M->dims[M->L] ? -1 : rows;
  return PDEC_OK;
}

int pdec_mlp_acts_on_published_copy(pdec_handle actor, int* yes) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(yes, "pdec_mlp_acts_on_published_copy: null");
  *yes = (fused_net_supported(M) && M->dims[M->L] == 1 && M->dims[1] <= 31) ? 1 : 0;
  return PDEC_OK;
}

int pdec_noise_counter_set(pdec_handle actor, uint64_t value) {
  GET_MLP(M, actor);
  int rc = ensure_noise_ctr(M);
  if (rc) return rc;
  PDEC_HIP(hipStreamSynchronize(M->stream));
  PDEC_HIP(hipMemcpy(M->noise_ctr.as<uint64_t>() + M->nc_sel, &value, sizeof(value), hipMemcpyHostToDevice));
  return PDEC_OK;
}

int pdec_debug_critic_stamps(pdec_handle critic, int arm, double* out13) {
  GET_MLP(M, critic);
  if (arm) { M->stamps_armed = true; return PDEC_OK; }
  PDEC_REQUIRE(out13, "pdec_debug_critic_stamps: null");
  PDEC_REQUIRE(!M->stamps_armed && M->stamps_last[12] > 0, "pdec_debug_critic_stamps: no fused critic pass has run on this network since it was armed");
  PDEC_HIP(hipStreamSynchronize(M->stream));
  for (int i = 0; i < 13; ++i) out13[i] = M->stamps_last[i];
  return PDEC_OK;
}

int pdec_noise_counter_get(pdec_handle actor, uint64_t* value) {
  GET_MLP(M, actor);
  PDEC_REQUIRE(value, "pdec_noise_counter_get: null");
  int rc = ensure_noise_ctr(M);
  if (rc) return rc;
  PDEC_HIP(hipStreamSynchronize(M->stream));
  PDEC_HIP(hipMemcpy(value, M->noise_ctr.as<uint64_t>() + M->nc_sel, sizeof(*value), hipMemcpyDeviceToHost));
  return PDEC_OK;
}

}  // extern "C"
