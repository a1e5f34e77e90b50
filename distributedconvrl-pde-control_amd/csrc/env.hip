// env.hip -- batched PDE environment step for gfx950: actuator synthesis, time integration,
// sensor read-out, reward, im2col featurize and blow-up flag fused in one launch.
//
// Restates (from scratch, batched over independent trajectories):
//   (env::PDEenv)(action)              src/PDEenv.jl:195-241
//   KS do_step (CNAB2, spectral)       scripts/KS/setup/KSSetup.jl:115-160
//   Keller-Segel f + RK4               scripts/Keller-Segel/setup/KellerSegelSetup.jl:213-239
//   featurize / prepare_action / reward_function   KSSetup.jl:162-245, KellerSegelSetup.jl:241-332
//
// KS kernel design: one workgroup integrates TWO trajectories packed as the real and
// imaginary part of one complex sequence z = u_a + i u_b.  Every operator of the CNAB2
// scheme is either a real diagonal in wave space (A_inv, B), multiplication by the purely
// imaginary diagonal G = -i alpha/2 (linear, so it acts on the packed spectrum directly) or
// the pointwise square in physical space, which acts on Re and Im separately -- so the pair
// never has to be separated and one complex FFT serves two trajectories.  All 2K+3 FFTs of
// a control step run out of LDS (mixed-radix Stockham); per-mode state lives in registers;
// HBM sees only the compulsory traffic (y, action in; y, state, reward, done out).
#include "env.hpp"
#include "mlp.hpp"

namespace pdec {

// ------------------------------------------------------------------ shared device pieces

// dots[r][s] = sum_j Gs[j][s] * y_r[(sn0[s]+j) mod N] for r in {0,1}; yf(r,n) reads LDS.  Threads
// are split into groups that each cover a slice of the window; partials are combined via `part`.
template <class T, class YF>
__device__ __forceinline__ void sense_dots(const EnvDev<T>& e, YF yf, T* dots, T* part, int tid, int nt) {
  const int S = e.S, N = e.N, Wd = e.Wd;
  int ng = nt / S;
  if (ng < 1) ng = 1;
  if (ng > 8) ng = 8;
  const int chunk = (Wd + ng - 1) / ng;
  for (int idx = tid; idx < ng * S; idx += nt) {
    const int grp = idx / S, s = idx - grp * S;
    int j0 = grp * chunk, j1 = j0 + chunk;
    if (j1 > Wd) j1 = Wd;
    int n = e.sn0[s] + j0;
    if (n >= N) n -= N;
    // eight table rows in flight and four independent partial sums per trajectory: the loop used to be one load-to-use
    // latency per row (11 k cycles of the 77 k-cycle C2 step for a 90-row band)
    T p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
    int j = j0;
    for (; j + 8 <= j1; j += 8) {
      T gk[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) gk[u] = e.Gs[(size_t)(j + u) * S + s];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        int nn = n + u;
        nn = nn >= N ? nn - N : nn;
        p0[u & 3] += gk[u] * yf(0, nn);
        p1[u & 3] += gk[u] * yf(1, nn);
      }
      n += 8;
      if (n >= N) n -= N;
    }
    for (; j < j1; ++j) {
      const T gk = e.Gs[(size_t)j * S + s];
      p0[0] += gk * yf(0, n);
      p1[0] += gk * yf(1, n);
      if (++n == N) n = 0;
    }
    part[(grp * 2 + 0) * S + s] = (p0[0] + p0[1]) + (p0[2] + p0[3]);
    part[(grp * 2 + 1) * S + s] = (p1[0] + p1[1]) + (p1[2] + p1[3]);
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * S; idx += nt) {
    T acc = 0;
    for (int grp = 0; grp < ng; ++grp) acc += part[grp * 2 * S + idx];
    dots[idx] = acc;
  }
  __syncthreads();
}

template <class T>
__device__ __forceinline__ T pow_abs(T d, T p) {
  d = d < 0 ? -d : d;
  if (p == (T)2) return d * d;
  if (p == (T)1) return d;
  return d == 0 ? (T)0 : (T)pow((double)d, (double)p);
}
template <>
__device__ __forceinline__ float pow_abs<float>(float d, float p) {
  d = fabsf(d);
  if (p == 2.0f) return d * d;
  if (p == 1.0f) return d;
  return d == 0.0f ? 0.0f : powf(d, p);
}

// reward_function for one trajectory: dots = <y, g_s> of the species the reward looks at
// (returns the sum of the rewards THIS thread wrote, for the optional per-workgroup reward sum)
template <class T>
__device__ __forceinline__ T reward_traj(const EnvDev<T>& e, const T* dots, const T* act, const T* actp,
                                         T* r_out, int tid, int nt) {
  T mine = 0;
  if (!e.mono) {
    for (int a = tid; a < e.A; a += nt) {
      const int s = e.a2s[a];
      const T d = e.r_in_scale * (dots[s] + e.r_offset * e.gsum[s]);
      const T da = act[a] - actp[a];
      const T r = -pow_abs<T>(d, e.r_power) / e.r_denom - e.a_pun * act[a] * act[a] - e.da_pun * da * da;
      r_out[a] = r;
      mine += r;
    }
  } else if (tid == 0) {
    T acc = 0;
    for (int a = 0; a < e.A; ++a) {
      const int s = e.a2s[a];
      const T d = e.r_in_scale * (dots[s] + e.r_offset * e.gsum[s]);
      const T da = act[a] - actp[a];
      acc += -pow_abs<T>(d, e.r_power) / e.r_denom - e.a_pun * act[a] * act[a] - e.da_pun * da * da;
    }
    r_out[0] = acc / (T)e.A;
    mine = acc / (T)e.A;
  }
  return mine;
}

// featurize for one trajectory.  dots: [n_species][S]; state/prev: [A][ns] (or [1][S] mono)
template <class T>
__device__ __forceinline__ void featurize_traj(const EnvDev<T>& e, const T* dots, const T* prev, T* state,
                                               int tid, int nt) {
  if (e.mono) {
    for (int s = tid; s < e.S; s += nt) state[s] = dots[s] * e.sensor_scale;
    return;
  }
  if (e.fmap) {                         // temporal_steps == 1: every row is fresh -- one gather through the map built at creation (the general
    const int tot = e.A * e.ns;         // path below spends ~40 instructions per element on divisions by run-time values)
    for (int idx = tid; idx < tot; idx += nt) state[idx] = dots[e.fmap[idx]] * e.sensor_scale;
    return;
  }
  const int w = e.window / 2;
  const int fresh = e.window * e.n_species;
  for (int idx = tid; idx < e.A * e.ns; idx += nt) {
    const int a = idx / e.ns, rr = idx - a * e.ns;
    T v;
    if (rr < fresh || prev == nullptr) {
      const int r0 = rr % fresh;
      const int sp = r0 / e.window, i = (r0 - sp * e.window) - w;
      int s = (e.a2s[a] - i) % e.S;
      if (s < 0) s += e.S;
      v = dots[sp * e.S + s] * e.sensor_scale;
    } else {
      v = prev[a * e.ns + (rr - fresh)];
    }
    state[idx] = v;
  }
}

// featurize with action memory (cfg.memory_size > 0; KSSetup.jl:190-229 with :216 and :220-226): columns are
// [fresh window rows | the previous state's rows minus its oldest block and its memory rows | memory rows], the memory rows =
// rows 1.. of the action just applied (actg [A][na]; null = reset form, featurize(y0) without env: zeros).  Kept apart from
// featurize_traj so that the fused step kernels stay what they were, instruction for instruction.
template <class T>
__device__ __forceinline__ void featurize_traj_mem(const EnvDev<T>& e, const T* dots, const T* prev, const T* actg, T* state,
                                                   int tid, int nt) {
  const int w = e.window / 2;
  const int fresh = e.window * e.n_species, body = e.ns - e.mem;
  for (int idx = tid; idx < e.A * e.ns; idx += nt) {
    const int a = idx / e.ns, rr = idx - a * e.ns;
    T v;
    if (rr >= body) {
      v = actg ? actg[(size_t)a * e.na + 1 + (rr - body)] : (T)0;
    } else if (rr < fresh || prev == nullptr) {
      const int r0 = rr % fresh;
      const int sp = r0 / e.window, i = (r0 - sp * e.window) - w;
      int sidx = (e.a2s[a] - i) % e.S;
      if (sidx < 0) sidx += e.S;
      v = dots[sp * e.S + sidx] * e.sensor_scale;
    } else {
      v = prev[(size_t)a * e.ns + (rr - fresh)];
    }
    state[idx] = v;
  }
}

// reward + featurize of the TWO trajectories of a workgroup in one pass each (per-actuator agents, temporal_steps == 1): the
// table loads (a2s, gsum, fmap) are shared and the two trajectories' load-to-use latencies overlap instead of following
// each other (3.6 k + 2.2 k cycles of the C2 step as four separate loops).  Same arithmetic per element as reward_traj /
// featurize_traj.  r1 / st1 null: single trajectory.
template <class T>
__device__ __forceinline__ T reward_pair(const EnvDev<T>& e, const T* dots0, const T* dots1, const T* act0, const T* act1,
                                         const T* actp0, const T* actp1, T* r0, T* r1, int tid, int nt) {
  T mine = 0;
  for (int a = tid; a < e.A; a += nt) {
    const int s = e.a2s[a];
    const T off = e.r_offset * e.gsum[s];
    const T d0 = e.r_in_scale * (dots0[s] + off);
    const T da0 = act0[a] - actp0[a];
    const T v0 = -pow_abs<T>(d0, e.r_power) / e.r_denom - e.a_pun * act0[a] * act0[a] - e.da_pun * da0 * da0;
    r0[a] = v0;
    mine += v0;
    if (r1) {
      const T d1 = e.r_in_scale * (dots1[s] + off);
      const T da1 = act1[a] - actp1[a];
      const T v1 = -pow_abs<T>(d1, e.r_power) / e.r_denom - e.a_pun * act1[a] * act1[a] - e.da_pun * da1 * da1;
      r1[a] = v1;
      mine += v1;
    }
  }
  return mine;
}
template <class T>
__device__ __forceinline__ void featurize_pair(const EnvDev<T>& e, const T* dots0, const T* dots1, T* st0, T* st1, int tid, int nt) {
  const int tot = e.A * e.ns;
  for (int idx = tid; idx < tot; idx += nt) {
    const int m = e.fmap[idx];
    st0[idx] = dots0[m] * e.sensor_scale;
    if (st1) st1[idx] = dots1[m] * e.sensor_scale;
  }
}

// p[n] = agent_power * sum_i act[(an0[n]+i) mod A] * GaC[i][n]
template <class T>
__device__ __forceinline__ T actuate_cell(const EnvDev<T>& e, const T* act, int n) {
  T acc = 0;
  int a = e.an0[n];
  for (int i = 0; i < e.Cnt; ++i) {
    acc += act[a] * e.GaC[(size_t)i * e.N + n];
    if (++a == e.A) a = 0;
  }
  return acc * e.agent_power;
}
// two trajectories at once (shared table loads)
template <class T>
__device__ __forceinline__ void actuate_cell2(const EnvDev<T>& e, const T* act0, const T* act1, int n, T& p0, T& p1) {
  T a0 = 0, a1 = 0;
  int a = e.an0[n];
  for (int i = 0; i < e.Cnt; ++i) {
    const T gk = e.GaC[(size_t)i * e.N + n];
    a0 += act0[a] * gk;
    a1 += act1[a] * gk;
    if (++a == e.A) a = 0;
  }
  p0 = a0 * e.agent_power;
  p1 = a1 * e.agent_power;
}

// the KS_MPT cells a lane owns at once: their table rows are independent loads (one per cell and table row in flight
// together, two rows unrolled) instead of one load-to-use latency per cell and row -- 5.4 k -> the C2 step's actuation;
// per cell the sum runs over the rows in the same order as actuate_cell2
template <class T, int M>
__device__ __forceinline__ void actuate_cells(const EnvDev<T>& e, const T* act0, const T* act1, const int (&n)[M], T (&p0)[M],
                                              T (&p1)[M]) {
  int a[M];
  bool ok[M];
#pragma unroll
  for (int j = 0; j < M; ++j) {
    ok[j] = n[j] < e.N;
    a[j] = ok[j] ? e.an0[n[j]] : 0;
    p0[j] = 0; p1[j] = 0;
  }
#pragma unroll 2
  for (int i = 0; i < e.Cnt; ++i) {
    T gk[M];
#pragma unroll
    for (int j = 0; j < M; ++j) gk[j] = ok[j] ? e.GaC[(size_t)i * e.N + n[j]] : (T)0;
#pragma unroll
    for (int j = 0; j < M; ++j) {
      p0[j] += act0[a[j]] * gk[j];
      p1[j] += act1[a[j]] * gk[j];
      if (++a[j] == e.A) a[j] = 0;
    }
  }
#pragma unroll
  for (int j = 0; j < M; ++j) { p0[j] *= e.agent_power; p1[j] *= e.agent_power; }
}

// All cells of both trajectories with FOUR CONSECUTIVE cells per lane: one 16/32-byte load per table row and lane (the rows
// of a lane's cells tid + 64 j are four separate 4-byte loads: 92 loads per lane at C2, 5 k cycles of load-to-use
// latency), results through an LDS scratch [2][N] from which every lane then takes the cells its transform owns.
// Per cell the sum runs over the rows in the order of actuate_cell2.  Needs N % 4 == 0.
template <class T>
__device__ __forceinline__ void actuate_consecutive(const EnvDev<T>& e, const T* act0, const T* act1, T* scratch, int tid, int nt) {
  typedef T T4 __attribute__((ext_vector_type(4)));
  const int N = e.N, A = e.A;
  for (int c0 = 4 * tid; c0 < N; c0 += 4 * nt) {
    int a[4];
    T p0[4] = {0, 0, 0, 0}, p1[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = e.an0[c0 + u];
#pragma unroll 4
    for (int i = 0; i < e.Cnt; ++i) {
      const T4 g = *reinterpret_cast<const T4*>(e.GaC + (size_t)i * N + c0);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p0[u] += act0[a[u]] * g[u];
        p1[u] += act1[a[u]] * g[u];
        if (++a[u] == A) a[u] = 0;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      scratch[c0 + u] = p0[u] * e.agent_power;
      scratch[N + c0 + u] = p1[u] * e.agent_power;
    }
  }
  __syncthreads();
}

template <class T>
__device__ __forceinline__ T block_max(T v, T* red, int tid, int nt) {
  for (int off = 32; off > 0; off >>= 1) {
    T o = __shfl_xor(v, off);
    v = o > v ? o : v;
  }
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  T r = red[0];
  for (int i = 1; i < (nt + 63) / 64; ++i) r = red[i] > r ? red[i] : r;
  __syncthreads();
  return r;
}

// per-column terminal flags for the DDPG batch (every actuator column of a blown-up trajectory is terminal)
template <class T>
__device__ __forceinline__ void write_terminal(const EnvDev<T>& e, int b, bool flag, int tid, int nt) {
  if (!e.term_out) return;
  const int cpt = e.mono ? 1 : e.A;
  for (int a = tid; a < cpt; a += nt) e.term_out[(size_t)b * cpt + a] = flag ? (T)1 : (T)0;
}

// ------------------------------------------------------------------ KS CNAB2 kernel
#define KS_MPT 4  // modes / cells owned per thread: k = tid + j*nt

// FFT engines: transform the 4 values a thread owns (indices tid + j*nt) in place.
// Generic engine: mixed-radix Stockham through LDS (any N = 2^a 3^b 5^c).
template <class T>
struct FftGeneric {
  static constexpr int kThreads = 1024;     // largest workgroup the host launches this engine with
  C2<T>*X, *Y;
  const C2<T>* tw;
  FftPlan pl;
  int N, tid, nt;
  __device__ __forceinline__ void init(unsigned char* smem, const EnvDev<T>& e, int tid_, int nt_) {
    N = e.N; tid = tid_; nt = nt_; pl = e.fft;
    X = reinterpret_cast<C2<T>*>(smem);
    Y = X + N;
    C2<T>* t = Y + N;
    for (int k = tid; k < N; k += nt) t[k] = e.tw[k];
    tw = t;
  }
  static __host__ __device__ size_t lds_complex(int N) { return 3 * (size_t)N; }
  // wave-space mode held in slot j after a forward transform (natural order for this engine)
  __device__ __forceinline__ int mode_index(int j) const { return tid + j * nt; }
  // cell held in slot j in physical space
  __device__ __forceinline__ int phys_index(int j) const { return tid + j * nt; }
  template <int SGN>
  __device__ __forceinline__ void run(C2<T> (&a)[KS_MPT]) {
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int k = tid + j * nt;
      if (k < N) X[k] = a[j];
    }
    C2<T>* R = fft_lds<SGN, T>(X, Y, tw, pl, tid, nt);
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int k = tid + j * nt;
      if (k < N) a[j] = R[k];
    }
    __syncthreads();
  }
  // natural-order complex image of the last result for the sensing stage
  __device__ __forceinline__ C2<T>* publish(const C2<T> (&a)[KS_MPT]) {
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int k = tid + j * nt;
      if (k < N) X[k] = a[j];
    }
    __syncthreads();
    return X;
  }
};

// Radix-4 engine for N = 4^L with nt = N/4 threads: in the Stockham DIF form every stage of
// thread t reads x[t + (N/4) j] -- its own registers for the first stage and conflict-free LDS
// rows afterwards -- and the last stage lands back on the owned indices, so a transform costs
// L-1 LDS round trips and L-1 barriers; all twiddles are per-thread constants held in registers.
template <class T, int L>
struct FftR4 {
  C2<T>* buf[2];
  C2<T> w[L - 1][3];
  int tid, par;
  static constexpr int N = 1 << (2 * L), NT = N / 4;
  static constexpr int kThreads = (NT + 63) / 64 * 64;
  __device__ __forceinline__ void init(unsigned char* smem, const EnvDev<T>& e, int tid_, int) {
    tid = tid_; par = 0;
    buf[0] = reinterpret_cast<C2<T>*>(smem);
    buf[1] = buf[0] + N;
#pragma unroll
    for (int st = 0; st < L - 1; ++st) {
      const int s = 1 << (2 * st);
      const int base = tid & ~(s - 1);          // p*s
#pragma unroll
      for (int k = 1; k < 4; ++k) w[st][k - 1] = e.tw[base * k];
    }
  }
  static __host__ __device__ size_t lds_complex(int) { return 2 * (size_t)N; }
  __device__ __forceinline__ int mode_index(int j) const { return tid + j * NT; }
  __device__ __forceinline__ int phys_index(int j) const { return tid + j * NT; }
  template <int SGN>
  __device__ __forceinline__ void run(C2<T> (&a)[KS_MPT]) {
#pragma unroll
    for (int st = 0; st < L; ++st) {
      if (st > 0) {
        const C2<T>* in = buf[par ^ ((st - 1) & 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = in[tid + NT * j];
      }
      dft_small<4, SGN, T>(a);
      if (st < L - 1) {
        const int s = 1 << (2 * st);
        const int q = tid & (s - 1);
        const int ob = q + 4 * (tid - q);       // q + 4 s p
        C2<T>* out = buf[par ^ (st & 1)];
        out[ob] = a[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) {
          C2<T> tw = w[st][k - 1];
          if (SGN > 0) tw.y = -tw.y;
          out[ob + s * k] = cmul(a[k], tw);
        }
        __syncthreads();
      }
    }
    if ((L - 1) & 1) par ^= 1;                  // next transform starts on the buffer not read last
  }
  __device__ __forceinline__ C2<T>* publish(const C2<T> (&a)[KS_MPT]) {
    __syncthreads();
    C2<T>* X = buf[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) X[tid + NT * j] = a[j];
    __syncthreads();
    return X;
  }
};

// ---- register-resident single-wave engine for N = 256 (64 lanes x 4 points): NO LDS traffic.
// In-place radix-4 decimation in frequency: stage st transforms the index digit that currently lives in
// the register index, then that digit is exchanged with one 2-bit digit of the lane id -- lane bits 5:4 by
// v_permlane32_swap / v_permlane16_swap, bits 3:2 by bank-masked DPP row shifts, bits 1:0 by DPP quad
// permutes -- so the next stage again works on the 4 registers of a lane.  The forward transform leaves mode
// k = (lane>>4) + 4((lane>>2)&3) + 16(lane&3) + 64 j in slot j (digit-reversed); the inverse runs the same
// steps backwards and returns to the natural order n = lane + 64 j.  The CNAB2 update is pointwise in wave
// space, so the permuted order only changes which per-mode constants a lane loads (mode_index).
// Besides being shorter, the transform does not queue behind other kernels' LDS traffic when the PDE step
// shares CUs with the MFMA update passes (measured: the LDS engine slowed 46 -> 140 us there).
__device__ __forceinline__ void lane_swap32(unsigned& a, unsigned& b) {   // a[lanes 32-63] <-> b[lanes 0-31]
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
__device__ __forceinline__ void lane_swap16(unsigned& a, unsigned& b) {   // odd 16-lane rows of a <-> even rows of b
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  a = r[0]; b = r[1];
}
#define PDEC_DPP(old, src, ctrl, bank) (unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(src), ctrl, 0xF, bank, false)
// exchange the register index (0..3) with lane bits 5:4
__device__ __forceinline__ void xch_rows(unsigned (&v)[4]) {
  lane_swap32(v[0], v[2]); lane_swap32(v[1], v[3]);
  lane_swap16(v[0], v[1]); lane_swap16(v[2], v[3]);
}
// ... with lane bits 3:2 (row_ror:8 = lane^8; row_shr:4 / row_shl:4 = lane-4 / lane+4 inside a 16-lane row)
__device__ __forceinline__ void xch_mid(unsigned (&v)[4]) {
  unsigned t;
  t = v[0]; v[0] = PDEC_DPP(v[0], v[2], 0x128, 0xC); v[2] = PDEC_DPP(v[2], t, 0x128, 0x3);
  t = v[1]; v[1] = PDEC_DPP(v[1], v[3], 0x128, 0xC); v[3] = PDEC_DPP(v[3], t, 0x128, 0x3);
  t = v[0]; v[0] = PDEC_DPP(v[0], v[1], 0x114, 0xA); v[1] = PDEC_DPP(v[1], t, 0x104, 0x5);
  t = v[2]; v[2] = PDEC_DPP(v[2], v[3], 0x114, 0xA); v[3] = PDEC_DPP(v[3], t, 0x104, 0x5);
}
// ... with lane bits 1:0 (quad_perm [2,3,0,1] = lane^2, [1,0,3,2] = lane^1)
__device__ __forceinline__ void xch_low(unsigned (&v)[4], bool b1, bool b0) {
  unsigned s, t;
  s = PDEC_DPP(0, v[2], 0x4E, 0xF); t = PDEC_DPP(0, v[0], 0x4E, 0xF); v[0] = b1 ? s : v[0]; v[2] = b1 ? v[2] : t;
  s = PDEC_DPP(0, v[3], 0x4E, 0xF); t = PDEC_DPP(0, v[1], 0x4E, 0xF); v[1] = b1 ? s : v[1]; v[3] = b1 ? v[3] : t;
  s = PDEC_DPP(0, v[1], 0xB1, 0xF); t = PDEC_DPP(0, v[0], 0xB1, 0xF); v[0] = b0 ? s : v[0]; v[1] = b0 ? v[1] : t;
  s = PDEC_DPP(0, v[3], 0xB1, 0xF); t = PDEC_DPP(0, v[2], 0xB1, 0xF); v[2] = b0 ? s : v[2]; v[3] = b0 ? v[3] : t;
}
// apply an exchange to every 32-bit word of the 4 complex values a lane holds
template <int WHICH, class T>
__device__ __forceinline__ void xch_complex(C2<T> (&a)[4], bool b1, bool b0) {
  constexpr int W = sizeof(T) / 4;      // words per real
  unsigned w[2 * W][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned tmp[2 * W];
    __builtin_memcpy(tmp, &a[j], sizeof(C2<T>));
#pragma unroll
    for (int c = 0; c < 2 * W; ++c) w[c][j] = tmp[c];
  }
#pragma unroll
  for (int c = 0; c < 2 * W; ++c) {
    if (WHICH == 2) xch_rows(w[c]);
    else if (WHICH == 1) xch_mid(w[c]);
    else xch_low(w[c], b1, b0);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    unsigned tmp[2 * W];
#pragma unroll
    for (int c = 0; c < 2 * W; ++c) tmp[c] = w[c][j];
    __builtin_memcpy(&a[j], tmp, sizeof(C2<T>));
  }
}

// ---- fp32 fast path: the same exchanges written in place (inline asm), one instruction per moved word.
// One binary step of a digit exchange on register pairs (P_k, Q_k), k = 0..3 (two components x two pairs):
//   newQ = bit ? Q : perm(P),  newP = bit ? perm(Q) : P        (bit = the lane-id bit being exchanged)
// as v_cndmask_b32_dpp (DPP permutes src0; VCC = lane mask of the bit, then its complement).  The builtin form
// above costs ~2.5x the instructions in register copies and separate selects.
#define PDEC_XSTEP(CA, CB, MASK, P0, Q0, P1, Q1, P2, Q2, P3, Q3)                                           \
  {                                                                                                        \
    float n0_, n1_, n2_, n3_;                                                                              \
    asm("s_nop 1\n\t"                                                                                      \
        "s_mov_b64 vcc, %12\n\t"                                                                           \
        "v_cndmask_b32_dpp %8, %0, %1, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %9, %2, %3, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %10, %4, %5, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                          \
        "v_cndmask_b32_dpp %11, %6, %7, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                          \
        "s_mov_b64 vcc, %13\n\t"                                                                           \
        "v_cndmask_b32_dpp %0, %1, %0, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %2, %3, %2, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %4, %5, %4, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %6, %7, %6, vcc " CB " row_mask:0xf bank_mask:0xf"                                \
        : "+v"(P0), "+v"(Q0), "+v"(P1), "+v"(Q1), "+v"(P2), "+v"(Q2), "+v"(P3), "+v"(Q3), "=&v"(n0_), "=&v"(n1_), \
          "=&v"(n2_), "=&v"(n3_)                                                                           \
        : "s"(MASK), "s"(~(MASK))                                                                          \
        : "vcc");                                                                                          \
    Q0 = n0_; Q1 = n1_; Q2 = n2_; Q3 = n3_;                                                                \
  }
// digit = lane bits 3:2 (WHICH 1) or 1:0 (WHICH 0); bits 5:4 (WHICH 2) use the permlane swaps
template <int WHICH>
__device__ __forceinline__ void xch_complex_f32(C2<float> (&a)[4]) {
  if (WHICH == 2) {
    asm("s_nop 1\n\t"
        "v_permlane32_swap_b32 %0, %2\n\t"
        "v_permlane32_swap_b32 %4, %6\n\t"
        "v_permlane32_swap_b32 %1, %3\n\t"
        "v_permlane32_swap_b32 %5, %7\n\t"
        "s_nop 1\n\t"
        "v_permlane16_swap_b32 %0, %1\n\t"
        "v_permlane16_swap_b32 %4, %5\n\t"
        "v_permlane16_swap_b32 %2, %3\n\t"
        "v_permlane16_swap_b32 %6, %7"
        : "+v"(a[0].x), "+v"(a[1].x), "+v"(a[2].x), "+v"(a[3].x), "+v"(a[0].y), "+v"(a[1].y), "+v"(a[2].y), "+v"(a[3].y));
  } else if (WHICH == 1) {
    // bit 3 (lane ^ 8 = row_ror:8), register pairs (0,2), (1,3)
    PDEC_XSTEP("row_ror:8", "row_ror:8", 0xFF00FF00FF00FF00ull, a[0].x, a[2].x, a[1].x, a[3].x, a[0].y, a[2].y, a[1].y, a[3].y)
    // bit 2: lanes with the bit clear read lane + 4 (row_ror:12), lanes with it set read lane - 4 (row_ror:4); pairs (0,1), (2,3)
    PDEC_XSTEP("row_ror:12", "row_ror:4", 0xF0F0F0F0F0F0F0F0ull, a[0].x, a[1].x, a[2].x, a[3].x, a[0].y, a[1].y, a[2].y, a[3].y)
  } else {
    PDEC_XSTEP("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", 0xCCCCCCCCCCCCCCCCull, a[0].x, a[2].x, a[1].x, a[3].x, a[0].y, a[2].y,
               a[1].y, a[3].y)
    PDEC_XSTEP("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", 0xAAAAAAAAAAAAAAAAull, a[0].x, a[1].x, a[2].x, a[3].x, a[0].y, a[1].y,
               a[2].y, a[3].y)
  }
}

// ---- fp32 packed-math butterflies: VOP3P op_sel / neg modifiers give the multiplication by +-i and the complex
// product without any register shuffling (the compiler scalarises these and adds ~60 moves per transform).
typedef float pkf2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ pkf2 pk_add_mi(pkf2 a, pkf2 b) {   // a + (-i) b = (a.x + b.y, a.y - b.x)
  pkf2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ pkf2 pk_add_pi(pkf2 a, pkf2 b) {   // a + i b = (a.x - b.y, a.y + b.x)
  pkf2 r;
  asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <int SGN>   // a * w (SGN < 0) or a * conj(w) (SGN > 0)
__device__ __forceinline__ pkf2 pk_cmul(pkf2 a, pkf2 w) {
  pkf2 t, r;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(a), "v"(w));                 // (a.x w.x, a.y w.x)
  if (SGN < 0)   // (t.x - a.y w.y, t.y + a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  else           // (t.x + a.y w.y, t.y - a.x w.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[0,1,0]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
  return r;
}
template <int SGN>
__device__ __forceinline__ void pk_dft4(C2<float> (&a)[4]) {
  pkf2 v0 = __builtin_bit_cast(pkf2, a[0]), v1 = __builtin_bit_cast(pkf2, a[1]), v2 = __builtin_bit_cast(pkf2, a[2]),
       v3 = __builtin_bit_cast(pkf2, a[3]);
  const pkf2 s02 = v0 + v2, d02 = v0 - v2, s13 = v1 + v3, d13 = v1 - v3;
  v0 = s02 + s13;
  v2 = s02 - s13;
  v1 = SGN < 0 ? pk_add_mi(d02, d13) : pk_add_pi(d02, d13);   // d02 + (-+i) d13
  v3 = SGN < 0 ? pk_add_pi(d02, d13) : pk_add_mi(d02, d13);   // d02 - (-+i) d13
  a[0] = __builtin_bit_cast(C2<float>, v0); a[1] = __builtin_bit_cast(C2<float>, v1);
  a[2] = __builtin_bit_cast(C2<float>, v2); a[3] = __builtin_bit_cast(C2<float>, v3);
}

template <class T>
struct FftWave256 {
  static constexpr int kThreads = 64;
  C2<T>* buf;
  C2<T> w[3][3];       // twiddles of the three inner stages, per lane
  int tid;
  bool b1, b0;
  static constexpr int N = 256, NT = 64;
  __device__ __forceinline__ void init(unsigned char* smem, const EnvDev<T>& e, int tid_, int) {
    tid = tid_;
    b1 = (tid & 2) != 0; b0 = (tid & 1) != 0;
    buf = reinterpret_cast<C2<T>*>(smem);
    const int low[3] = {tid, 4 * (tid & 15), 16 * (tid & 3)};   // k * (index formed by the digits still to transform)
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int k = 1; k < 4; ++k) w[st][k - 1] = e.tw[(k * low[st]) & 255];
  }
  static __host__ __device__ size_t lds_complex(int) { return (size_t)N; }   // only for publish()
  __device__ __forceinline__ int mode_index(int j) const {
    return (tid >> 4) + 4 * ((tid >> 2) & 3) + 16 * (tid & 3) + 64 * j;
  }
  __device__ __forceinline__ int phys_index(int j) const { return tid + NT * j; }
  template <int ST>
  __device__ __forceinline__ void exchange(C2<T> (&a)[4]) {
    if constexpr (sizeof(T) == 4) {
      xch_complex_f32<2 - ST>(reinterpret_cast<C2<float>(&)[4]>(a));
    } else {
      if (ST == 0) xch_complex<2, T>(a, b1, b0);
      else if (ST == 1) xch_complex<1, T>(a, b1, b0);
      else xch_complex<0, T>(a, b1, b0);
    }
  }
  template <int ST, int SGN>
  __device__ __forceinline__ void twiddle(C2<T> (&a)[4]) {
#pragma unroll
    for (int k = 1; k < 4; ++k) {
      if constexpr (sizeof(T) == 4) {
        a[k] = __builtin_bit_cast(C2<T>, pk_cmul<SGN>(__builtin_bit_cast(pkf2, a[k]), __builtin_bit_cast(pkf2, w[ST][k - 1])));
      } else {
        C2<T> tw = w[ST][k - 1];
        if (SGN > 0) tw.y = -tw.y;
        a[k] = cmul(a[k], tw);
      }
    }
  }
  template <int SGN>
  __device__ __forceinline__ void dft4(C2<T> (&a)[4]) {
    if constexpr (sizeof(T) == 4) pk_dft4<SGN>(reinterpret_cast<C2<float>(&)[4]>(a));
    else dft_small<4, SGN, T>(a);
  }
  template <int SGN>
  __device__ __forceinline__ void run(C2<T> (&a)[KS_MPT]) {
    if (SGN < 0) {   // forward: natural -> digit-reversed
      dft4<-1>(a); twiddle<0, -1>(a); exchange<0>(a);
      dft4<-1>(a); twiddle<1, -1>(a); exchange<1>(a);
      dft4<-1>(a); twiddle<2, -1>(a); exchange<2>(a);
      dft4<-1>(a);
    } else {         // inverse: digit-reversed -> natural (unnormalised)
      dft4<+1>(a);
      exchange<2>(a); twiddle<2, +1>(a); dft4<+1>(a);
      exchange<1>(a); twiddle<1, +1>(a); dft4<+1>(a);
      exchange<0>(a); twiddle<0, +1>(a); dft4<+1>(a);
    }
  }
  __device__ __forceinline__ C2<T>* publish(const C2<T> (&a)[KS_MPT]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) buf[tid + NT * j] = a[j];
    __syncthreads();
    return buf;
  }
};

// ---- N = 1024 (BASELINE configs[2]) on four waves: ONE cross-wave radix-4 stage + the single-wave 256-point engine (round 4).
// Thread tid of 256 owns x[tid + 256 j], j = 0..3 -- exactly the inputs of the first radix-4 DIF butterfly.  After that
// butterfly and its twiddle W_1024^(tid k2), slot k2 holds element n1 = tid of the length-256 sub-sequence k2; one LDS round
// trip hands sub-sequence w to wave w as element lane + 64 j in slot j, and the wave transforms it in registers (FftWave256:
// lane-digit exchanges by permlane swaps / DPP, no LDS, no barrier).  A transform therefore costs ONE LDS round trip and ONE
// workgroup barrier where the Stockham engine FftR4<5> needs four of each; the two buffers alternate so the next
// transform's writes need no second barrier.  Forward leaves mode k = w + 4 (perm(lane) + 64 j) in slot j of wave w (the CNAB2
// update is pointwise in wave space: only the per-mode constant loads are permuted); the inverse runs the steps backwards
// and returns to the natural order.
template <class T>
struct FftWave1024 {
  static constexpr int kThreads = 256;
  static constexpr int N = 1024, NT = 256;
  FftWave256<T> core;
  C2<T>* buf[2];
  C2<T> wx[3];          // W_1024^(tid k), k = 1..3
  int tid, lane, wv, par;
  __device__ __forceinline__ void init(unsigned char* smem, const EnvDev<T>& e, int tid_, int) {
    tid = tid_; lane = tid & 63; wv = tid >> 6; par = 0;
    buf[0] = reinterpret_cast<C2<T>*>(smem);
    buf[1] = buf[0] + N;
    core.tid = lane;
    core.b1 = (lane & 2) != 0; core.b0 = (lane & 1) != 0;
    core.buf = buf[0];
    const int low[3] = {lane, 4 * (lane & 15), 16 * (lane & 3)};
#pragma unroll
    for (int st = 0; st < 3; ++st)
#pragma unroll
      for (int k = 1; k < 4; ++k) core.w[st][k - 1] = e.tw[(4 * k * low[st]) & 1023];      // W_256^x = W_1024^(4x)
#pragma unroll
    for (int k = 1; k < 4; ++k) wx[k - 1] = e.tw[(k * tid) & 1023];
  }
  static __host__ __device__ size_t lds_complex(int) { return 2 * (size_t)N; }
  __device__ __forceinline__ int mode_index(int j) const {
    return wv + 4 * ((lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3) + 64 * j);
  }
  __device__ __forceinline__ int phys_index(int j) const { return tid + NT * j; }
  template <int SGN>
  __device__ __forceinline__ void run(C2<T> (&a)[KS_MPT]) {
    C2<T>* X = buf[par];
    par ^= 1;
    if (SGN < 0) {   // forward: natural -> (wave, digit-reversed)
      core.template dft4<-1>(a);
#pragma unroll
      for (int k = 1; k < 4; ++k) a[k] = cmul(a[k], wx[k - 1]);
#pragma unroll
      for (int k = 0; k < 4; ++k) X[k * 256 + tid] = a[k];
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = X[wv * 256 + lane + 64 * j];
      core.template run<-1>(a);
    } else {         // inverse (unnormalised)
      core.template run<+1>(a);
#pragma unroll
      for (int j = 0; j < 4; ++j) X[wv * 256 + lane + 64 * j] = a[j];
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) a[k] = X[k * 256 + tid];
#pragma unroll
      for (int k = 1; k < 4; ++k) {
        C2<T> tw = wx[k - 1];
        tw.y = -tw.y;
        a[k] = cmul(a[k], tw);
      }
      core.template dft4<+1>(a);
    }
  }
  __device__ __forceinline__ C2<T>* publish(const C2<T> (&a)[KS_MPT]) {
    __syncthreads();
    C2<T>* X = buf[0];
#pragma unroll
    for (int j = 0; j < 4; ++j) X[tid + NT * j] = a[j];
    __syncthreads();
    return X;
  }
};

// Compile-time mixed-radix engine for the reference's own grid sizes (KS22: 192 = 4.4.3.4, KS200: 240 = 4.5.3.4,
// KS500: 600 = 4.5.5.2.3).  Stockham stages like FftGeneric, but (i) every radix, stride and index split is a template
// constant (no runtime plan walk, divisions by constants, fully unrolled), and (ii) the plan has radices <= 4 at both
// ends: in PHYSICAL space thread t owns the cells t + (N/R_first) j -- exactly the inputs of its first-stage butterfly
// -- and in WAVE space the modes t + (N/R_last) k -- the outputs of its last-stage butterfly; the inverse runs the plan
// backwards, so it consumes the wave-space layout and lands on the physical one.  A transform therefore costs L-1 LDS
// round trips (the generic engine: L+2) and one butterfly per thread and stage.  A single wave per trajectory pair has
// nothing to hide latency behind, so dependent round trips and instruction count ARE the step time at these sizes.
template <class T, int N_, int L_, int R0, int R1, int R2, int R3, int R4>
struct FftFixed {
  static constexpr int N = N_, L = L_;
  C2<T>* buf[2];
  const C2<T>* tw;
  int tid;
  static constexpr int rad(int i) { return i == 0 ? R0 : (i == 1 ? R1 : (i == 2 ? R2 : (i == 3 ? R3 : R4))); }
  static constexpr int M0 = N / R0, ML = N / rad(L - 1);
  static constexpr int max_m(int i) { return i >= L ? 0 : (N / rad(i) > max_m(i + 1) ? N / rad(i) : max_m(i + 1)); }
  static constexpr int kThreads = (max_m(0) + 63) / 64 * 64;
  __device__ __forceinline__ void init(unsigned char* smem, const EnvDev<T>& e, int tid_, int nt) {
    tid = tid_;
    buf[0] = reinterpret_cast<C2<T>*>(smem);
    buf[1] = buf[0] + N;
    C2<T>* t = buf[1] + N;
    for (int k = tid; k < N; k += nt) t[k] = e.tw[k];
    tw = t;
  }
  static __host__ __device__ size_t lds_complex(int) { return 3 * (size_t)N; }
  __device__ __forceinline__ int mode_index(int j) const { return (tid < ML && j < rad(L - 1)) ? tid + j * ML : N; }
  __device__ __forceinline__ int phys_index(int j) const { return (tid < M0 && j < R0) ? tid + j * M0 : N; }

  // one Stockham stage of radix R on sub-length NN with stride S; FIRST: inputs are the caller's registers,
  // LAST: outputs stay in registers
  template <int R, int SGN, bool FIRST, bool LAST, int NN, int S>
  __device__ __forceinline__ void stage(C2<T> (&a)[KS_MPT], const C2<T>* __restrict__ X, C2<T>* __restrict__ Y) {
    constexpr int m = NN / R, nb = N / R;
    if (tid < nb) {
      const int p = tid / S, q = tid - p * S;
      C2<T> b[R];
#pragma unroll
      for (int j = 0; j < R; ++j) {
        if (FIRST) b[j] = a[j < KS_MPT ? j : 0];
        else b[j] = X[q + S * (p + m * j)];
      }
      C2<T> w[R];
      const int ps = p * S;
      if (!LAST) {
#pragma unroll
        for (int k = 1; k < R; ++k) w[k] = tw[ps * k];
      }
      dft_small<R, SGN, T>(b);
      const int base = q + S * R * p;
#pragma unroll
      for (int k = 0; k < R; ++k) {
        C2<T> v = b[k];
        if (!LAST && k > 0) {
          C2<T> ww = w[k];
          if (SGN > 0) ww.y = -ww.y;
          v = cmul(v, ww);
        }
        if (LAST) a[k < KS_MPT ? k : 0] = v;
        else Y[base + S * k] = v;
      }
    }
  }
  // stage I of the (forward or reversed) plan, sub-length and stride accumulated at compile time
  template <int SGN, int I, int NN, int S>
  __device__ __forceinline__ void walk(C2<T> (&a)[KS_MPT]) {
    if constexpr (I < L) {
      constexpr int R = rad(SGN < 0 ? I : L - 1 - I);
      constexpr bool FIRST = I == 0, LAST = I == L - 1;
      // stage I reads what stage I-1 wrote: buffers alternate, stage 0 writes buf[0]
      stage<R, SGN, FIRST, LAST, NN, S>(a, buf[(I + 1) & 1], buf[I & 1]);
      if (!LAST) __syncthreads();
      walk<SGN, I + 1, NN / R, S * R>(a);
    }
  }
  template <int SGN>
  __device__ __forceinline__ void run(C2<T> (&a)[KS_MPT]) {
    walk<SGN, 0, N, 1>(a);
    __syncthreads();      // the last stage's readers are done before the next transform writes buf[0] again
  }
  __device__ __forceinline__ C2<T>* publish(const C2<T> (&a)[KS_MPT]) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int k = phys_index(j);
      if (k < N) buf[0][k] = a[j];
    }
    __syncthreads();
    return buf[0];
  }
};
template <class T> using FftFixed192 = FftFixed<T, 192, 4, 4, 4, 3, 4, 1>;
template <class T> using FftFixed240 = FftFixed<T, 240, 4, 4, 5, 3, 4, 1>;
template <class T> using FftFixed600 = FftFixed<T, 600, 5, 4, 5, 5, 2, 3>;

// SHARE (pdec_env_set_simd_sharing; fp32 single-wave engine only): the 64-VGPR form of the kernel, see below
// SYNC (pdec_set_launch_sync; single-workgroup launches of the reference's own shapes): wait for the producer of the action
// before anything is read, signal behind the last store -- a separate instantiation, so that the batched kernels keep their code
template <class T, class ENG, bool FUSED, bool SHARE = false, bool SYNC = false>
__global__ void __launch_bounds__(ENG::kThreads, SHARE ? 8 : 1) ks_env_step_kernel(EnvDev<T> e, const T* __restrict__ y_in, const T* __restrict__ p_in,
                                   const T* __restrict__ action, const T* __restrict__ action_prev,
                                   const T* __restrict__ state_prev, T* __restrict__ y_out,
                                   T* __restrict__ p_out, T* __restrict__ state_out,
                                   T* __restrict__ reward_out, int32_t* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x;
  if constexpr (SYNC) launch_sync_wait(e.sync);
  // This kernel is a long dependent chain (63 FFTs) issued by very few waves.  Beside the f32-MFMA update passes (which
  // execute on the vector unit) each of its instructions waits for an MFMA to drain (~2.7x slower), and their eight waves
  // per workgroup wait for each other at barriers; on its own it runs at priority 1, below the passes at 2 (r02f).
  // SHARE: in its register form (86 VGPRs) a wave of this kernel cannot share a SIMD with two waves of the 222-VGPR critic
  // pass (2 x 224 + 86 > 512): step and pass exclude each other per CU, and the step's tail delayed every workgroup of
  // the next pass (75 us in the pipeline against 63 alone, r02j).  The SHARE form keeps the per-mode constants, the
  // constant term and the previous nonlinear term in LDS (LDSC below), is bounded to 64 VGPRs (2 x 224 + 64 = 512) and
  // runs at priority 3: it is over before the next pass needs the registers, and the pass keeps its alone time (r02l:
  // 135 -> 125 us per control step).  Alone the SHARE form is slower (37 vs 29 us: four exposed LDS round trips per
  // sub-step), so only the two-stream training pipeline asks for it.
  set_wave_prio(e.prio);
  ENG eng;
  eng.init(smem_raw, e, tid, nt);
  T* act = reinterpret_cast<T*>(reinterpret_cast<C2<T>*>(smem_raw) + ENG::lds_complex(N));  // [2][A] current
  T* actp = act + 2 * e.A;                // [2][A] previous
  T* dots = actp + 2 * e.A;               // [2][S]
  T* part = dots + 2 * e.S;               // [8][2][S]
  T* red = part + 16 * e.S;               // [16]

  const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
  const bool has1 = b1 < e.B;
  const size_t o0 = (size_t)b0 * N, o1 = (size_t)b1 * N;

  if (FUSED) {
    for (int a = tid; a < e.A; a += nt) {
      act[a] = action[(size_t)b0 * e.A + a];
      act[e.A + a] = has1 ? action[(size_t)b1 * e.A + a] : (T)0;
      actp[a] = action_prev[(size_t)b0 * e.A + a];
      actp[e.A + a] = has1 ? action_prev[(size_t)b1 * e.A + a] : (T)0;
    }
  }
  __syncthreads();

  // LDSC: the per-mode constants, the constant term and the previous nonlinear term live in LDS (lane-private float4
  // slots) instead of 32 registers -- the kernel then fits in 64 VGPRs and a wave of it can share a SIMD with two waves
  // of the 222-VGPR critic pass (2 x 224 + 64 = 512), instead of waiting for / holding up a whole workgroup of it
  constexpr bool LDSC = SHARE;
  typedef T T4v __attribute__((ext_vector_type(4)));
  T4v* cst = reinterpret_cast<T4v*>(smem_raw + ((size_t)(reinterpret_cast<unsigned char*>(red + 16) - smem_raw + 15) & ~(size_t)15));
  C2<T> U[KS_MPT], Nn[KS_MPT], Ck[KS_MPT], v[KS_MPT];
  T kc1[KS_MPT], kc2[KS_MPT], kc3[KS_MPT], kg[KS_MPT];
  // forcing p (packed pair) -> spectrum -> constant term of the CNAB2 update
  T pa4[KS_MPT], pb4[KS_MPT];
  if (FUSED) {
    int n4[KS_MPT];
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) n4[j] = eng.phys_index(j);
    if ((N & 3) == 0 && 2 * N <= 16 * e.S) {     // `part` ([8][2][S], free until the sensor dots) holds the [2][N] scratch
      actuate_consecutive<T>(e, act, act + e.A, part, tid, nt);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        pa4[j] = n4[j] < N ? part[n4[j]] : (T)0;
        pb4[j] = n4[j] < N ? part[N + n4[j]] : (T)0;
      }
    } else {
      actuate_cells<T, KS_MPT>(e, act, act + e.A, n4, pa4, pb4);
    }
  }
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {
    const int n = eng.phys_index(j);
    T pa = 0, pb = 0;
    if (n < N) {
      if (FUSED) {
        pa = pa4[j]; pb = pb4[j];
        if (!has1) pb = 0;
        if (p_out) {
          p_out[o0 + n] = pa;
          if (has1) p_out[o1 + n] = pb;
        }
      } else {
        pa = p_in[o0 + n];
        pb = has1 ? p_in[o1 + n] : (T)0;
      }
    }
    v[j] = mk<T>(pa, pb);
  }
  eng.template run<-1>(v);
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {
    const int k = eng.mode_index(j);
    if (k < N) {
      const C2<T> d = e.dhat[k];
      const T c4 = e.c4[k];
      // (1+i)*dhat: the same real disturbance enters both packed trajectories
      Ck[j] = mk<T>(c4 * v[j].x + (d.x - d.y), c4 * v[j].y + (d.x + d.y));
      kc1[j] = e.c1[k];
      kc2[j] = e.c2[k];
      kc3[j] = e.c3[k];
      kg[j] = e.g[k];
    } else {
      Ck[j] = mk<T>(0, 0);
      kc1[j] = kc2[j] = kc3[j] = kg[j] = 0;
    }
    if constexpr (LDSC) cst[j * nt + tid] = T4v{kc1[j], kc2[j], kc3[j], kg[j]};
  }
  if constexpr (LDSC) {
    cst[4 * nt + tid] = T4v{Ck[0].x, Ck[0].y, Ck[1].x, Ck[1].y};
    cst[5 * nt + tid] = T4v{Ck[2].x, Ck[2].y, Ck[3].x, Ck[3].y};
  }
  // Nn = G * fft(u^2);  u_hat = fft(u)
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {
    const int n = eng.phys_index(j);
    U[j] = n < N ? mk<T>(y_in[o0 + n], has1 ? y_in[o1 + n] : (T)0) : mk<T>(0, 0);
    v[j] = mk<T>(U[j].x * U[j].x, U[j].y * U[j].y);
  }
  eng.template run<-1>(v);
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) Nn[j] = cscale(mul_i<+1, T>(v[j]), kg[j]);   // G = i * (-alpha/2)
  if constexpr (LDSC) {
    cst[6 * nt + tid] = T4v{Nn[0].x, Nn[0].y, Nn[1].x, Nn[1].y};
    cst[7 * nt + tid] = T4v{Nn[2].x, Nn[2].y, Nn[3].x, Nn[3].y};
  }
  eng.template run<-1>(U);
  const T invN = (T)1 / (T)N;
  if constexpr (LDSC) {
    for (int it = 0; it < e.K; ++it) {
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) v[j] = U[j];
      eng.template run<+1>(v);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        const T wr = v[j].x * invN, wi = v[j].y * invN;
        v[j] = mk<T>(wr * wr, wi * wi);
      }
      eng.template run<-1>(v);
#pragma unroll
      for (int h = 0; h < 2; ++h) {          // modes 2h, 2h + 1
        const T4v nn = cst[(6 + h) * nt + tid], ck = cst[(4 + h) * nt + tid];
        T4v nw;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int j = 2 * h + u;
          const T4v c = cst[j * nt + tid];   // c1, c2, c3, g
          const C2<T> n1 = cscale(mul_i<+1, T>(v[j]), c[3]);
          U[j] = mk<T>(c[0] * U[j].x + c[1] * n1.x - c[2] * nn[2 * u] + ck[2 * u],
                       c[0] * U[j].y + c[1] * n1.y - c[2] * nn[2 * u + 1] + ck[2 * u + 1]);
          nw[2 * u] = n1.x; nw[2 * u + 1] = n1.y;
        }
        cst[(6 + h) * nt + tid] = nw;
      }
    }
  } else {
    for (int it = 0; it < e.K; ++it) {
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) v[j] = U[j];
      eng.template run<+1>(v);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        const T wr = v[j].x * invN, wi = v[j].y * invN;
        v[j] = mk<T>(wr * wr, wi * wi);
      }
      eng.template run<-1>(v);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        const C2<T> nn1 = Nn[j];
        Nn[j] = cscale(mul_i<+1, T>(v[j]), kg[j]);
        U[j] = mk<T>(kc1[j] * U[j].x + kc2[j] * Nn[j].x - kc3[j] * nn1.x + Ck[j].x,
                     kc1[j] * U[j].y + kc2[j] * Nn[j].y - kc3[j] * nn1.y + Ck[j].y);
      }
    }
  }
  // y+ = real(ifft(u_hat))
  eng.template run<+1>(U);
  T mx0 = 0, mx1 = 0;
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {
    const int n = eng.phys_index(j);
    U[j] = mk<T>(U[j].x * invN, U[j].y * invN);
    if (n < N) {
      y_out[o0 + n] = U[j].x;
      if (has1) y_out[o1 + n] = U[j].y;
      // blow-up test max|y| > max_value (src/PDEenv.jl:227); a NaN also raises the flag
      // (deliberate deviation: Julia's `NaN > max_value` is false and the run would go on)
      if (!(fabs(U[j].x) <= e.max_value)) mx0 = 1;
      if (!(fabs(U[j].y) <= e.max_value)) mx1 = 1;
    }
  }
  if (done) {
    mx0 = block_max<T>(mx0, red, tid, nt);
    mx1 = block_max<T>(mx1, red, tid, nt);
    if (tid == 0) {
      const bool chk = e.check_max == 1;
      done[b0] = (chk && mx0 > 0) ? 1 : 0;
      if (has1) done[b1] = (chk && mx1 > 0) ? 1 : 0;
    }
    if (FUSED && e.check_max != 2) {
      write_terminal<T>(e, b0, e.check_max == 1 && mx0 > 0, tid, nt);
      if (has1) write_terminal<T>(e, b1, e.check_max == 1 && mx1 > 0, tid, nt);
    }
  }
  if (!FUSED) return;
  const T* Rt = reinterpret_cast<const T*>(eng.publish(U));
  sense_dots<T>(e, [&](int r, int n) { return Rt[2 * n + r]; }, dots, part, tid, nt);
  const int rw = e.mono ? 1 : e.A;             // reward entries per trajectory
  const int sw = e.mono ? e.S : e.A * e.ns;    // state entries per trajectory
  T rmine;
  if (e.fmap && !e.mono) {     // both trajectories in one pass each
    rmine = reward_pair<T>(e, dots, dots + e.S, act, act + e.A, actp, actp + e.A, reward_out + (size_t)b0 * rw,
                           has1 ? reward_out + (size_t)b1 * rw : nullptr, tid, nt);
    featurize_pair<T>(e, dots, dots + e.S, state_out + (size_t)b0 * sw, has1 ? state_out + (size_t)b1 * sw : nullptr, tid, nt);
  } else {
    rmine = reward_traj<T>(e, dots, act, actp, reward_out + (size_t)b0 * rw, tid, nt);
    featurize_traj<T>(e, dots, state_prev ? state_prev + (size_t)b0 * sw : nullptr, state_out + (size_t)b0 * sw, tid, nt);
    if (has1) {
      rmine += reward_traj<T>(e, dots + e.S, act + e.A, actp + e.A, reward_out + (size_t)b1 * rw, tid, nt);
      featurize_traj<T>(e, dots + e.S, state_prev ? state_prev + (size_t)b1 * sw : nullptr,
                        state_out + (size_t)b1 * sw, tid, nt);
    }
  }
  if (e.rsum_out) {
    // per-workgroup reward sum (fixed order: lanes by xor-shuffle, then waves in order) for the batch-mean reward of the
    // DDPG update's reward broadcast -- the critic pass then adds one partial per workgroup instead of re-reading all of r
    float v = (float)rmine;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = (T)v;
    __syncthreads();
    if (tid == 0) {
      float tot = 0.f;
      for (int i = 0; i < (nt + 63) / 64; ++i) tot += (float)red[i];
      e.rsum_out[blockIdx.x] = tot;
    }
  }
  if (done && e.check_max == 2) {
    // check_max_value == "reward" (src/PDEenv.jl:232-237): flag on max|reward|
    __syncthreads();
    if (tid == 0) {
      for (int t = 0; t < (has1 ? 2 : 1); ++t) {
        T m = 0;
        const T* r = reward_out + (size_t)(b0 + t) * rw;
        for (int a = 0; a < rw; ++a)
          if (!(fabs(r[a]) <= e.max_value)) m = 1;
        done[b0 + t] = m > 0 ? 1 : 0;
        write_terminal<T>(e, b0 + t, m > 0, 0, 1);
      }
    }
  }
  if constexpr (SYNC) launch_sync_done(e.sync);
}

// ------------------------------------------------------------------ persistent KS rollout (row F2)
// T control steps of  action = clamp(actor(state) + randn * act_noise);  (env::PDEenv)(action)  in ONE launch
// (src/PDEagent.jl:175-209 + src/PDEenv.jl:195-241 + the KS closures of KSSetup.jl:130-245): the two trajectories of a
// workgroup stay in registers between steps, their sensor dots / state / actions in LDS; nothing returns to HBM between
// steps but the optional log rows PDEhook records.  The actor (a chain of <= 3 Dense layers, widths <= RO_W, one output)
// is evaluated one column per lane on the vector unit from a copy of its parameters in LDS; exploration noise from the
// same Philox element numbering as pdec_policy_act_rng (element = global column, counter offset + t * ceil(cols / 4)).
#define RO_W 32            // widest layer the in-kernel actor holds in registers
struct RollActor {
  const void* params;      // flat [W1 row-major [out][in], b1, W2, b2, ...] in the environment's dtype
  int L, nparams, rows;    // rows = widest layer: the height of an activation plane
  int dims[4], acts[3];
};
template <class T>
struct RollArgs {
  int steps, learning;
  T act_noise, act_limit;
  uint64_t seed, offset;
  T *y, *state, *action;                   // in / out: [B][N], [B][A][ns], [B][A]
  T* reward_sum;                           // optional [B][A]: += every step's reward
  T *log_y, *log_p, *log_action, *log_reward;   // optional [steps][B][...]
  int32_t *done_any, *done_step;           // optional [B]
};

template <class T>
__device__ __forceinline__ T ro_act_fn(T z, int act) {
  if (act == PDEC_ACT_RELU) return z > (T)0 ? z : (T)0;
  if (act == PDEC_ACT_TANH) return (T)tanh((double)z);
  return z;
}
template <>
__device__ __forceinline__ float ro_act_fn<float>(float z, int act) {
  if (act == PDEC_ACT_RELU) return fmaxf(z, 0.f);
  if (act == PDEC_ACT_TANH) return tanhf(z);
  return z;
}

// LDS image of the actor: per layer Wt[din][RO_W] (transposed, outputs zero-padded to RO_W) followed by b[RO_W], so the
// RO_NB consecutive outputs a thread owns are contiguous (broadcast 128-bit reads).
#define RO_NB 16
__host__ __device__ inline int ro_image_elems(const int* dims, int L) {
  int n = 0;
  for (int l = 0; l < L; ++l) n += (dims[l] + 1) * RO_W;
  return n;
}
template <class T>
__device__ __forceinline__ void ro_load_image(const RollActor& A, T* wl, int tid, int nt) {
  const T* src = static_cast<const T*>(A.params);
  int so = 0, dof = 0;
  for (int l = 0; l < A.L; ++l) {
    const int din = A.dims[l], dout = A.dims[l + 1];
    for (int i = tid; i < (din + 1) * RO_W; i += nt) {
      const int r = i / RO_W, o = i - r * RO_W;                    // r < din: weight row, r == din: bias
      wl[dof + i] = o < dout ? (r < din ? src[so + o * din + r] : src[so + din * dout + o]) : (T)0;
    }
    so += din * dout + dout;
    dof += (din + 1) * RO_W;
  }
}

// actor forward for ONE PAIR of adjacent columns per thread (packed v_pk_fma_f32 for fp32): the activations of the pair
// sit in two LDS planes hb[plane][i][slot] private to the thread (bank = lane: conflict-free, no barrier between layers),
// the weights are broadcast 128-bit reads of the image; outputs in blocks of RO_NB accumulators, inputs four at a time so
// the LDS reads of four k-steps are in flight together; k-ordered accumulation like the oracle's W * x + b.
template <class T> struct RoPair {
  typedef T type __attribute__((ext_vector_type(2)));
  typedef T quad __attribute__((ext_vector_type(4), aligned(16)));
};
// NB outputs [ob, ob + NB) of one layer for the thread's column pair: acc = b + sum_i W[.][i] * in[i]
template <class T, int NB>
__device__ __forceinline__ void ro_block(const T* __restrict__ Wt, int din, int dout, int ob, int act,
                                         const typename RoPair<T>::type* pin, typename RoPair<T>::type* pout, int nslot) {
  using T2 = typename RoPair<T>::type;
  using T4 = typename RoPair<T>::quad;
  T2 acc[NB];
  T4 w[4][NB / 4];
#pragma unroll
  for (int v = 0; v < NB / 4; ++v) {
    const T4 b4 = *reinterpret_cast<const T4*>(Wt + din * RO_W + 4 * v);
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[4 * v + r] = T2{b4[r], b4[r]};
  }
  int i = 0;
  for (; i + 4 <= din; i += 4) {
    T2 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = pin[(size_t)(i + u) * nslot];
#pragma unroll
      for (int v = 0; v < NB / 4; ++v) w[u][v] = *reinterpret_cast<const T4*>(Wt + (i + u) * RO_W + 4 * v);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] += w[u][j >> 2][j & 3] * a[u];
  }
  for (; i < din; ++i) {
    const T2 a0 = pin[(size_t)i * nslot];
#pragma unroll
    for (int v = 0; v < NB / 4; ++v) w[0][v] = *reinterpret_cast<const T4*>(Wt + i * RO_W + 4 * v);
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] += w[0][j >> 2][j & 3] * a0;
  }
#pragma unroll
  for (int j = 0; j < NB; ++j)
    if (ob + j < dout) pout[(size_t)(ob + j) * nslot] = T2{ro_act_fn<T>(acc[j].x, act), ro_act_fn<T>(acc[j].y, act)};
}
template <class T>
__device__ __forceinline__ typename RoPair<T>::type ro_actor_pair(const RollActor& A, const T* __restrict__ wl,
                                                                  typename RoPair<T>::type* hb, int slot, int nslot) {
  using T2 = typename RoPair<T>::type;
  T2* pin = hb + slot;
  T2* pout = hb + (size_t)A.rows * nslot + slot;
  int off = 0;
  for (int l = 0; l < A.L; ++l) {
    const int din = A.dims[l], dout = A.dims[l + 1], act = A.acts[l];
    int ob = 0;
    for (; dout - ob > 4; ob += RO_NB) ro_block<T, RO_NB>(wl + off + ob, din, dout, ob, act, pin, pout, nslot);
    if (ob < dout) ro_block<T, 4>(wl + off + ob, din, dout, ob, act, pin, pout, nslot);
    T2* tmp = pin; pin = pout; pout = tmp;
    off += (din + 1) * RO_W;
  }
  return pin[0];
}

template <class T, class ENG>
__global__ void __launch_bounds__(ENG::kThreads) ks_rollout_kernel(EnvDev<T> e, RollActor actor, RollArgs<T> g) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x, A = e.A, ns = e.ns;
  set_wave_prio(e.prio);
  ENG eng;
  eng.init(smem_raw, e, tid, nt);
  T* act = reinterpret_cast<T*>(reinterpret_cast<C2<T>*>(smem_raw) + ENG::lds_complex(N));  // [2][A] current
  T* actp = act + 2 * A;                  // [2][A] previous
  T* dots = actp + 2 * A;                 // [2][S]
  T* part = dots + 2 * e.S;               // [8][2][S]
  T* red = part + 16 * e.S;               // [16]
  T* stl = red + 16;                      // [2][A * ns]  state of both trajectories
  T* rsum = stl + 2 * A * ns;             // [2][A]       accumulated reward
  T* rnow = rsum + 2 * A;                 // [2][A]       this step's reward
  const size_t wl_off = (size_t)(reinterpret_cast<unsigned char*>(rnow + 2 * A) - smem_raw + 15) & ~(size_t)15;
  T* wl = reinterpret_cast<T*>(smem_raw + wl_off);   // actor image, 16-byte aligned rows
  using T2 = typename RoPair<T>::type;
  T2* hb = reinterpret_cast<T2*>(wl + ((ro_image_elems(actor.dims, actor.L) + 3) & ~3));   // [2][rows][nt] column pairs

  const int b0 = 2 * blockIdx.x, b1 = b0 + 1;
  const bool has1 = b1 < e.B;
  const size_t o0 = (size_t)b0 * N, o1 = (size_t)b1 * N;
  const size_t cols = (size_t)e.B * A;

  ro_load_image<T>(actor, wl, tid, nt);
  for (int i = tid; i < A * ns; i += nt) {
    stl[i] = g.state[(size_t)b0 * A * ns + i];
    stl[A * ns + i] = has1 ? g.state[(size_t)b1 * A * ns + i] : (T)0;
  }
  for (int a = tid; a < A; a += nt) {
    act[a] = g.action[(size_t)b0 * A + a];
    act[A + a] = has1 ? g.action[(size_t)b1 * A + a] : (T)0;
    rsum[a] = rsum[A + a] = 0;
  }
  C2<T> U[KS_MPT], Nn[KS_MPT], Ck[KS_MPT], v[KS_MPT];
  T kc1[KS_MPT], kc2[KS_MPT], kc3[KS_MPT], kg[KS_MPT], kc4[KS_MPT];
  C2<T> kd[KS_MPT];
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {      // per-mode constants (mode layout of the engine) and the initial fields
    const int k = eng.mode_index(j);
    const bool ok = k < N;
    kc1[j] = ok ? e.c1[k] : (T)0; kc2[j] = ok ? e.c2[k] : (T)0; kc3[j] = ok ? e.c3[k] : (T)0;
    kc4[j] = ok ? e.c4[k] : (T)0; kg[j] = ok ? e.g[k] : (T)0;
    kd[j] = ok ? e.dhat[k] : mk<T>(0, 0);
    const int n = eng.phys_index(j);
    U[j] = n < N ? mk<T>(g.y[o0 + n], has1 ? g.y[o1 + n] : (T)0) : mk<T>(0, 0);
  }
  int flag0 = 0, flag1 = 0, first0 = -1, first1 = -1;
  const T invN = (T)1 / (T)N;
  __syncthreads();

  for (int t = 0; t < g.steps; ++t) {
    // ---- policy (src/PDEagent.jl:183-207): one pair of adjacent columns of the [2][A] column space per thread
    for (int q0 = 0; q0 < A; q0 += nt) {
      const int q = q0 + tid;
      if (q < A) {
        const int idx0 = 2 * q;
        for (int i = 0; i < ns; ++i) hb[(size_t)i * nt + tid] = T2{stl[(size_t)idx0 * ns + i], stl[(size_t)(idx0 + 1) * ns + i]};
        const T2 o2 = ro_actor_pair<T>(actor, wl, hb, tid, nt);
        uint64_t cprev = ~0ull;
        double rad = 0, ang = 0;
        uint32_t ph[4];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int idx = idx0 + s, r = idx / A, a = idx - r * A;
          T o = s == 0 ? o2.x : o2.y;
          if (g.learning && (r == 0 || has1)) {
            const uint64_t c = (uint64_t)(r == 0 ? b0 : b1) * A + a;          // global column = element of the noise stream
            if ((c >> 2) != cprev) {
              const uint64_t ctr = g.offset + (uint64_t)t * ((cols + 3) / 4) + (c >> 2);
              ph[0] = (uint32_t)ctr; ph[1] = (uint32_t)(ctr >> 32); ph[2] = 0u; ph[3] = 0u;
              philox4x32(ph, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
              cprev = c >> 2;
            }
            if (s == 0 || (c & 1) == 0) {      // the odd element shares the Box-Muller pair of its even neighbour
              const int hsel = (int)((c >> 1) & 1);
              const double sc = 1.0 / 4294967296.0;
              const double u1 = ((double)ph[2 * hsel] + 0.5) * sc, u2 = ((double)ph[2 * hsel + 1] + 0.5) * sc;
              rad = sqrt(-2.0 * log(u1)); ang = 6.283185307179586 * u2;
            }
            o += (T)((c & 1) ? rad * sin(ang) : rad * cos(ang)) * g.act_noise;
          }
          o = o < -g.act_limit ? -g.act_limit : (o > g.act_limit ? g.act_limit : o);
          actp[idx] = act[idx];
          act[idx] = (r == 0 || has1) ? o : (T)0;
        }
      }
    }
    __syncthreads();
    if (g.log_action)
      for (int a = tid; a < A; a += nt) {
        g.log_action[((size_t)t * e.B + b0) * A + a] = act[a];
        if (has1) g.log_action[((size_t)t * e.B + b1) * A + a] = act[A + a];
      }
    // ---- prepare_action -> spectrum -> constant term of the CNAB2 update (KSSetup.jl:231-245, :155)
    T pa4[KS_MPT], pb4[KS_MPT];
    {
      int n4[KS_MPT];
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) n4[j] = eng.phys_index(j);
      if ((N & 3) == 0 && 2 * N <= 16 * e.S) {
        actuate_consecutive<T>(e, act, act + A, part, tid, nt);
#pragma unroll
        for (int j = 0; j < KS_MPT; ++j) {
          pa4[j] = n4[j] < N ? part[n4[j]] : (T)0;
          pb4[j] = n4[j] < N ? part[N + n4[j]] : (T)0;
        }
        __syncthreads();      // `part` is reused by the sensor dots of this step
      } else {
        actuate_cells<T, KS_MPT>(e, act, act + A, n4, pa4, pb4);
      }
    }
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int n = eng.phys_index(j);
      T pa = 0, pb = 0;
      if (n < N) {
        pa = pa4[j]; pb = pb4[j];
        if (!has1) pb = 0;
        if (g.log_p) {
          g.log_p[((size_t)t * e.B + b0) * N + n] = pa;
          if (has1) g.log_p[((size_t)t * e.B + b1) * N + n] = pb;
        }
      }
      v[j] = mk<T>(pa, pb);
    }
    eng.template run<-1>(v);
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j)
      Ck[j] = mk<T>(kc4[j] * v[j].x + (kd[j].x - kd[j].y), kc4[j] * v[j].y + (kd[j].x + kd[j].y));
    // ---- do_step (KSSetup.jl:130-160): Nn = G fft(u^2), u_hat = fft(u), K CNAB2 sub-steps, y+ = real(ifft(u_hat))
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) v[j] = mk<T>(U[j].x * U[j].x, U[j].y * U[j].y);
    eng.template run<-1>(v);
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) Nn[j] = cscale(mul_i<+1, T>(v[j]), kg[j]);
    eng.template run<-1>(U);
    for (int it = 0; it < e.K; ++it) {
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) v[j] = U[j];
      eng.template run<+1>(v);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        const T wr = v[j].x * invN, wi = v[j].y * invN;
        v[j] = mk<T>(wr * wr, wi * wi);
      }
      eng.template run<-1>(v);
#pragma unroll
      for (int j = 0; j < KS_MPT; ++j) {
        const C2<T> nn1 = Nn[j];
        Nn[j] = cscale(mul_i<+1, T>(v[j]), kg[j]);
        U[j] = mk<T>(kc1[j] * U[j].x + kc2[j] * Nn[j].x - kc3[j] * nn1.x + Ck[j].x,
                     kc1[j] * U[j].y + kc2[j] * Nn[j].y - kc3[j] * nn1.y + Ck[j].y);
      }
    }
    eng.template run<+1>(U);
    T mx0 = 0, mx1 = 0;
#pragma unroll
    for (int j = 0; j < KS_MPT; ++j) {
      const int n = eng.phys_index(j);
      U[j] = mk<T>(U[j].x * invN, U[j].y * invN);
      if (n < N) {
        if (g.log_y) {
          g.log_y[((size_t)t * e.B + b0) * N + n] = U[j].x;
          if (has1) g.log_y[((size_t)t * e.B + b1) * N + n] = U[j].y;
        }
        if (!(fabs(U[j].x) <= e.max_value)) mx0 = 1;
        if (!(fabs(U[j].y) <= e.max_value)) mx1 = 1;
      }
    }
    if (e.check_max == 1) {
      mx0 = block_max<T>(mx0, red, tid, nt);
      mx1 = block_max<T>(mx1, red, tid, nt);
      if (mx0 > 0) { flag0 = 1; if (first0 < 0) first0 = t; }
      if (mx1 > 0) { flag1 = 1; if (first1 < 0) first1 = t; }
    }
    // ---- reward (KSSetup.jl:162-178) and featurize (:190-229) from the sensor dots of the new field
    const T* Rt = reinterpret_cast<const T*>(eng.publish(U));
    sense_dots<T>(e, [&](int r, int n) { return Rt[2 * n + r]; }, dots, part, tid, nt);
    if (e.fmap) {
      reward_pair<T>(e, dots, dots + e.S, act, act + A, actp, actp + A, rnow, has1 ? rnow + A : nullptr, tid, nt);
      featurize_pair<T>(e, dots, dots + e.S, stl, has1 ? stl + A * ns : nullptr, tid, nt);
    } else {
      reward_traj<T>(e, dots, act, actp, rnow, tid, nt);
      featurize_traj<T>(e, dots, nullptr, stl, tid, nt);
      if (has1) {
        reward_traj<T>(e, dots + e.S, act + A, actp + A, rnow + A, tid, nt);
        featurize_traj<T>(e, dots + e.S, nullptr, stl + A * ns, tid, nt);
      }
    }
    __syncthreads();
    for (int a = tid; a < A; a += nt) {
      rsum[a] += rnow[a];
      rsum[A + a] += rnow[A + a];
      if (g.log_reward) {
        g.log_reward[((size_t)t * e.B + b0) * A + a] = rnow[a];
        if (has1) g.log_reward[((size_t)t * e.B + b1) * A + a] = rnow[A + a];
      }
    }
    __syncthreads();
  }
  // ---- results back to HBM
#pragma unroll
  for (int j = 0; j < KS_MPT; ++j) {
    const int n = eng.phys_index(j);
    if (n < N) {
      g.y[o0 + n] = U[j].x;
      if (has1) g.y[o1 + n] = U[j].y;
    }
  }
  for (int i = tid; i < A * ns; i += nt) {
    g.state[(size_t)b0 * A * ns + i] = stl[i];
    if (has1) g.state[(size_t)b1 * A * ns + i] = stl[A * ns + i];
  }
  for (int a = tid; a < A; a += nt) {
    g.action[(size_t)b0 * A + a] = act[a];
    if (has1) g.action[(size_t)b1 * A + a] = act[A + a];
    if (g.reward_sum) {
      g.reward_sum[(size_t)b0 * A + a] += rsum[a];
      if (has1) g.reward_sum[(size_t)b1 * A + a] += rsum[A + a];
    }
  }
  if (tid == 0) {
    if (g.done_any) { g.done_any[b0] = flag0; if (has1) g.done_any[b1] = flag1; }
    if (g.done_step) { g.done_step[b0] = first0; if (has1) g.done_step[b1] = first1; }
  }
}

// ------------------------------------------------------------------ Keller-Segel RK4 kernel
// One workgroup per trajectory, one cell per thread; u,v in registers, neighbours through
// LDS with the reference's zero-flux edge fix-up (KellerSegelSetup.jl:220-223).
template <class T>
__device__ __forceinline__ void kseg_rhs(T u, T v, T p, T* su, T* sv, int n, int N, T idx, T idx2, bool live,
                                         T& du, T& dv) {
  __syncthreads();
  if (live) {
    su[n + 1] = u;
    sv[n + 1] = v;
    if (n == 0) {
      su[0] = u;
      sv[0] = v;
    }
    if (n == N - 1) {
      su[N + 1] = u;
      sv[N + 1] = v;
    }
  }
  __syncthreads();
  if (live) {
    const T um = su[n], up = su[n + 2], vm = sv[n], vp = sv[n + 2];
    const T ux = (T)0.5 * idx * (up - um);
    const T uxx = idx2 * um - (T)2 * idx2 * u + idx2 * up;
    const T vx = (T)0.5 * idx * (vp - vm);
    const T vxx = idx2 * vm - (T)2 * idx2 * v + idx2 * vp;
    dv = vxx - v + u + p;
    du = uxx + u - (T)5.6 * ux * vx - (T)5.6 * u * vxx - u * u;
  }
}

template <class T, int MODE>  // MODE 0: fused env step, 1: integrate only, 2: rhs only
__global__ void kseg_env_step_kernel(EnvDev<T> e, const T* __restrict__ y_in, const T* __restrict__ p_in,
                                     const T* __restrict__ action, const T* __restrict__ action_prev,
                                     const T* __restrict__ state_prev, T* __restrict__ y_out,
                                     T* __restrict__ p_out, T* __restrict__ state_out,
                                     T* __restrict__ reward_out, int32_t* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
  T* su = reinterpret_cast<T*>(smem_raw);  // [N+2]
  T* sv = su + N + 2;                      // [N+2]
  T* act = sv + N + 2;                     // [A]
  T* actp = act + e.A;                     // [A]
  T* dots = actp + e.A;                    // [2][S]
  T* part = dots + 2 * e.S;                // [8][2][S]
  T* red = part + 16 * e.S;                // [16]
  const int n = tid;
  const bool live = n < N;
  // y[2, nx] Julia column-major: element (species, cell) at cell*2 + species
  const size_t yo = (size_t)b * 2 * N;
  T u = live ? y_in[yo + 2 * n] : (T)0, v = live ? y_in[yo + 2 * n + 1] : (T)0;
  T p = 0;
  if (MODE == 0) {
    for (int a = tid; a < e.A; a += nt) {
      act[a] = action[(size_t)b * e.A + a];
      actp[a] = action_prev[(size_t)b * e.A + a];
    }
    __syncthreads();
    if (live) {
      p = actuate_cell<T>(e, act, n);
      if (p_out) p_out[(size_t)b * N + n] = p;
    }
  } else if (live) {
    p = p_in[(size_t)b * N + n];
  }
  const T idx = (T)1 / e.dx, idx2 = (T)1 / (e.dx * e.dx);
  if (MODE == 2) {
    T du = 0, dv = 0;
    kseg_rhs<T>(u, v, p, su, sv, n, N, idx, idx2, live, du, dv);
    if (live) {
      y_out[yo + 2 * n] = du;
      y_out[yo + 2 * n + 1] = dv;
    }
    return;
  }
  const T h = e.hstep;
  for (int it = 0; it < e.K; ++it) {
    T k1u = 0, k1v = 0, k2u = 0, k2v = 0, k3u = 0, k3v = 0, k4u = 0, k4v = 0;
    kseg_rhs<T>(u, v, p, su, sv, n, N, idx, idx2, live, k1u, k1v);
    if (e.rk2) {     // PDEenv's built-in integrator (src/PDEenv.jl:208-214): explicit midpoint, `oversampling` sub-steps
      kseg_rhs<T>(u + (T)0.5 * h * k1u, v + (T)0.5 * h * k1v, p, su, sv, n, N, idx, idx2, live, k2u, k2v);
      u = u + h * k2u;
      v = v + h * k2v;
      continue;
    }
    kseg_rhs<T>(u + (T)0.5 * h * k1u, v + (T)0.5 * h * k1v, p, su, sv, n, N, idx, idx2, live, k2u, k2v);
    kseg_rhs<T>(u + (T)0.5 * h * k2u, v + (T)0.5 * h * k2v, p, su, sv, n, N, idx, idx2, live, k3u, k3v);
    kseg_rhs<T>(u + h * k3u, v + h * k3v, p, su, sv, n, N, idx, idx2, live, k4u, k4v);
    u = u + h / (T)6 * (k1u + (T)2 * (k2u + k3u) + k4u);
    v = v + h / (T)6 * (k1v + (T)2 * (k2v + k3v) + k4v);
  }
  if (live) {
    y_out[yo + 2 * n] = u;
    y_out[yo + 2 * n + 1] = v;
  }
  if (done) {
    T m = (live && !(fabs(u) <= e.max_value && fabs(v) <= e.max_value)) ? (T)1 : (T)0;
    m = block_max<T>(m, red, tid, nt);
    if (tid == 0) done[b] = (e.check_max == 1 && m > 0) ? 1 : 0;
    if (MODE == 0 && e.check_max != 2) write_terminal<T>(e, b, e.check_max == 1 && m > 0, tid, nt);
  }
  if (MODE != 0) return;
  __syncthreads();
  if (live) {
    su[n] = u;
    sv[n] = v;
  }
  __syncthreads();
  sense_dots<T>(e, [&](int r, int nn) { return r == 0 ? su[nn] : sv[nn]; }, dots, part, tid, nt);
  reward_traj<T>(e, dots, act, actp, reward_out + (size_t)b * e.A, tid, nt);
  const size_t sw = (size_t)e.A * e.ns;
  featurize_traj<T>(e, dots, state_prev ? state_prev + b * sw : nullptr, state_out + b * sw, tid, nt);
  if (done && e.check_max == 2) {
    __syncthreads();
    if (tid == 0) {
      T m = 0;
      for (int a = 0; a < e.A; ++a)
        if (!(fabs(reward_out[(size_t)b * e.A + a]) <= e.max_value)) m = 1;
      done[b] = m > 0 ? 1 : 0;
      write_terminal<T>(e, b, m > 0, 0, 1);
    }
  }
}

// ------------------------------------------------------------------ persistent Keller-Segel rollout (row F2)
// T control steps of  action = clamp(actor(state) + randn * act_noise);  (env::PDEenv)(action)  in ONE launch for the 1-D
// Keller-Segel environment (src/PDEagent.jl:175-209 + src/PDEenv.jl:195-241 with scripts/Keller-Segel/setup/KellerSegelSetup.jl:
// 213-332): the fields u, v stay in registers (one cell per thread, one workgroup per trajectory), state / action / reward rows
// in LDS, the actor (<= 3 Dense layers of <= RO_W units, 12 -> 20 -> 20 -> 1 in the shipped script) is evaluated in the kernel,
// one thread per (actuator, unit), and the exploration noise is the same Philox stream element for element (column c = b A + a
// of step t) as the acting kernel's, so the launch tracks the step-by-step loop to the actor's summation order.
template <class T>
__global__ void kseg_rollout_kernel(EnvDev<T> e, RollActor actor, RollArgs<T> g) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x, A = e.A, ns = e.ns;
  T* su = reinterpret_cast<T*>(smem_raw);  // [N+2]
  T* sv = su + N + 2;                      // [N+2]
  T* act = sv + N + 2;                     // [A]
  T* actp = act + A;                       // [A]
  T* dots = actp + A;                      // [2][S]
  T* part = dots + 2 * e.S;                // [8][2][S]
  T* red = part + 16 * e.S;                // [16]
  T* stl = red + 16;                       // [2][A * ns]  state of the trajectory, ping-pong (temporal_steps > 1 shifts the old rows)
  T* stn = stl + A * ns;
  T* rsum = stn + A * ns;                  // [A]       accumulated reward
  T* rnow = rsum + A;                      // [A]       this step's reward
  const size_t wl_off = (size_t)(reinterpret_cast<unsigned char*>(rnow + A) - smem_raw + 15) & ~(size_t)15;
  T* wl = reinterpret_cast<T*>(smem_raw + wl_off);   // actor image, 16-byte aligned rows
  T* hb = wl + ((ro_image_elems(actor.dims, actor.L) + 3) & ~3);   // [2][A][RO_W] activations of the actor, ping-pong
  const int n = tid;
  const bool live = n < N;
  const size_t yo = (size_t)b * 2 * N, cols = (size_t)e.B * A;

  ro_load_image<T>(actor, wl, tid, nt);
  for (int i = tid; i < A * ns; i += nt) stl[i] = g.state[(size_t)b * A * ns + i];
  for (int a = tid; a < A; a += nt) {
    act[a] = g.action[(size_t)b * A + a];
    rsum[a] = 0;
  }
  T u = live ? g.y[yo + 2 * n] : (T)0, v = live ? g.y[yo + 2 * n + 1] : (T)0;
  int flag = 0, first = -1;
  const T idx = (T)1 / e.dx, idx2 = (T)1 / (e.dx * e.dx), h = e.hstep;
  __syncthreads();

  for (int t = 0; t < g.steps; ++t) {
    // ---- policy (src/PDEagent.jl:183-207): the A actuator columns share the weights (per-actuator agents); one thread per
    // (actuator, output unit) and layer, k-ordered accumulation like the oracle's W * x + b
    {
      const T* in = stl;
      int istride = ns, off = 0;
      for (int l = 0; l < actor.L; ++l) {
        const int din = actor.dims[l], dout = actor.dims[l + 1], fn = actor.acts[l];
        T* out = hb + (size_t)(l & 1) * A * RO_W;
        const T* Wt = wl + off;
        for (int id = tid; id < A * dout; id += nt) {
          const int a = id / dout, o = id - a * dout;
          T acc = Wt[din * RO_W + o];
          for (int k = 0; k < din; ++k) acc += Wt[k * RO_W + o] * in[a * istride + k];
          out[a * RO_W + o] = ro_act_fn<T>(acc, fn);
        }
        __syncthreads();
        in = out; istride = RO_W;
        off += (din + 1) * RO_W;
      }
      for (int a = tid; a < A; a += nt) {
        T o = in[a * RO_W];
        if (g.learning) {
          const uint64_t c = (uint64_t)b * A + a;                          // global column = element of the noise stream
          const uint64_t ctr = g.offset + (uint64_t)t * ((cols + 3) / 4) + (c >> 2);
          uint32_t ph[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
          philox4x32(ph, (uint32_t)g.seed, (uint32_t)(g.seed >> 32));
          const int hsel = (int)((c >> 1) & 1);                            // an even / odd pair of columns shares one Box-Muller pair
          const double sc = 1.0 / 4294967296.0;
          const double u1 = ((double)ph[2 * hsel] + 0.5) * sc, u2 = ((double)ph[2 * hsel + 1] + 0.5) * sc;
          const double rad = sqrt(-2.0 * log(u1)), ang = 6.283185307179586 * u2;
          o += (T)((c & 1) ? rad * sin(ang) : rad * cos(ang)) * g.act_noise;
        }
        o = o < -g.act_limit ? -g.act_limit : (o > g.act_limit ? g.act_limit : o);
        actp[a] = act[a];
        act[a] = o;
      }
    }
    __syncthreads();
    if (g.log_action)
      for (int a = tid; a < A; a += nt) g.log_action[((size_t)t * e.B + b) * A + a] = act[a];
    // ---- prepare_action (KellerSegelSetup.jl:249-262), then the integrator of the step kernel (same order of operations)
    T p = 0;
    if (live) {
      p = actuate_cell<T>(e, act, n);
      if (g.log_p) g.log_p[((size_t)t * e.B + b) * N + n] = p;
    }
    for (int it = 0; it < e.K; ++it) {
      T k1u = 0, k1v = 0, k2u = 0, k2v = 0, k3u = 0, k3v = 0, k4u = 0, k4v = 0;
      kseg_rhs<T>(u, v, p, su, sv, n, N, idx, idx2, live, k1u, k1v);
      if (e.rk2) {
        kseg_rhs<T>(u + (T)0.5 * h * k1u, v + (T)0.5 * h * k1v, p, su, sv, n, N, idx, idx2, live, k2u, k2v);
        u = u + h * k2u;
        v = v + h * k2v;
        continue;
      }
      kseg_rhs<T>(u + (T)0.5 * h * k1u, v + (T)0.5 * h * k1v, p, su, sv, n, N, idx, idx2, live, k2u, k2v);
      kseg_rhs<T>(u + (T)0.5 * h * k2u, v + (T)0.5 * h * k2v, p, su, sv, n, N, idx, idx2, live, k3u, k3v);
      kseg_rhs<T>(u + h * k3u, v + h * k3v, p, su, sv, n, N, idx, idx2, live, k4u, k4v);
      u = u + h / (T)6 * (k1u + (T)2 * (k2u + k3u) + k4u);
      v = v + h / (T)6 * (k1v + (T)2 * (k2v + k3v) + k4v);
    }
    if (live && g.log_y) {
      g.log_y[((size_t)t * e.B + b) * 2 * N + 2 * n] = u;
      g.log_y[((size_t)t * e.B + b) * 2 * N + 2 * n + 1] = v;
    }
    if (e.check_max == 1) {
      T m = (live && !(fabs(u) <= e.max_value && fabs(v) <= e.max_value)) ? (T)1 : (T)0;
      m = block_max<T>(m, red, tid, nt);
      if (m > 0) { flag = 1; if (first < 0) first = t; }
    }
    // ---- reward and featurize from the sensor dots of the new fields
    __syncthreads();
    if (live) {
      su[n] = u;
      sv[n] = v;
    }
    __syncthreads();
    sense_dots<T>(e, [&](int r, int nn) { return r == 0 ? su[nn] : sv[nn]; }, dots, part, tid, nt);
    reward_traj<T>(e, dots, act, actp, rnow, tid, nt);
    featurize_traj<T>(e, dots, stl, stn, tid, nt);       // new rows on top, the previous state's rows shifted down (temporal stack)
    { T* sw = stl; stl = stn; stn = sw; }
    __syncthreads();
    for (int a = tid; a < A; a += nt) {
      rsum[a] += rnow[a];
      if (g.log_reward) g.log_reward[((size_t)t * e.B + b) * A + a] = rnow[a];
    }
    __syncthreads();
  }
  // ---- results back to HBM
  if (live) {
    g.y[yo + 2 * n] = u;
    g.y[yo + 2 * n + 1] = v;
  }
  for (int i = tid; i < A * ns; i += nt) g.state[(size_t)b * A * ns + i] = stl[i];
  for (int a = tid; a < A; a += nt) {
    g.action[(size_t)b * A + a] = act[a];
    if (g.reward_sum) g.reward_sum[(size_t)b * A + a] += rsum[a];
  }
  if (tid == 0) {
    if (g.done_any) g.done_any[b] = flag;
    if (g.done_step) g.done_step[b] = first;
  }
}

// ------------------------------------------------------------------ KS, RK4 + periodic 5-point finite differences
// The north-star variant u_t = -u u_x - u_xx - u_xxxx + p (+ the disturbance of KSSetup.jl:155) on the stencil table
// the reference defines but never uses (scripts/KS/setup/KSSetup.jl:55-59): d/dx = [0,-1/2,0,1/2,0]/dx,
// d2/dx2 = [0,1,-2,1,0]/dx^2, d4/dx4 = [1,-4,6,-4,1]/dx^4, classical RK4 (src/fluid_rk4.jl:122-132 form) with K
// sub-steps.  It is a DIFFERENT discretisation from the reference's CNAB2 step, so it is pinned by
// its own oracle (oracle/ks.py: rhs_fd / do_step_rk4_fd), not by the golden trajectories.
// One workgroup per trajectory, one cell per thread; neighbours through an LDS line with a periodic halo of 2.
template <class T>
__device__ __forceinline__ T ksfd_rhs(T u, T force, T* su, int n, int N, T i2dx, T idx2, T idx4, bool live) {
  __syncthreads();
  if (live) {
    su[n + 2] = u;
    if (n < 2) su[N + 2 + n] = u;        // right halo = cells 0, 1
    if (n >= N - 2) su[n - (N - 2)] = u; // left halo  = cells N-2, N-1
  }
  __syncthreads();
  T f = 0;
  if (live) {
    const T m2 = su[n], m1 = su[n + 1], p1 = su[n + 3], p2 = su[n + 4];
    const T ux = i2dx * (p1 - m1);
    const T uxx = idx2 * (m1 - (T)2 * u + p1);
    const T uxxxx = idx4 * (m2 - (T)4 * m1 + (T)6 * u - (T)4 * p1 + p2);
    f = -u * ux - uxx - uxxxx + force;
  }
  return f;
}

// per-workgroup reward sum of the RK4 + FD steps (one trajectory per workgroup), for the batch-mean reward of the DDPG update's
// reward broadcast -- the same hand-over as the CNAB2 step's (pdec_env_set_reward_partials_out): lanes by xor-shuffle, then the
// waves in order
template <class T>
__device__ __forceinline__ void ksfd_reward_partial(const EnvDev<T>& e, T rmine, T* red, int tid, int nt) {
  float v = (float)rmine;
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  if (nt <= 64) {
    if (tid == 0) e.rsum_out[blockIdx.x] = v;
    return;
  }
  __syncthreads();
  if ((tid & 63) == 0) red[tid >> 6] = (T)v;
  __syncthreads();
  if (tid == 0) {
    float tot = 0.f;
    for (int i = 0; i < (nt + 63) / 64; ++i) tot += (float)red[i];
    e.rsum_out[blockIdx.x] = tot;
  }
}

template <class T, int MODE>  // MODE 0: fused env step, 1: integrate only, 2: rhs only
__global__ void ksfd_env_step_kernel(EnvDev<T> e, const T* __restrict__ y_in, const T* __restrict__ p_in,
                                     const T* __restrict__ action, const T* __restrict__ action_prev,
                                     const T* __restrict__ state_prev, T* __restrict__ y_out, T* __restrict__ p_out,
                                     T* __restrict__ state_out, T* __restrict__ reward_out, int32_t* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
  T* su = reinterpret_cast<T*>(smem_raw);  // [2][N] (first N+4 used as the halo line; reused as sensing image)
  T* act = su + 2 * N + 4;                 // [A]
  T* actp = act + e.A;                     // [A]
  T* dots = actp + e.A;                    // [2][S]
  T* part = dots + 2 * e.S;                // [8][2][S]
  T* red = part + 16 * e.S;                // [16]
  const int n = tid;
  const bool live = n < N;
  const size_t yo = (size_t)b * N;
  T u = live ? y_in[yo + n] : (T)0;
  T p = 0;
  if (MODE == 0) {
    for (int a = tid; a < e.A; a += nt) {
      act[a] = action[(size_t)b * e.A + a];
      actp[a] = action_prev[(size_t)b * e.A + a];
    }
    __syncthreads();
    if (live) {
      p = actuate_cell<T>(e, act, n);
      if (p_out) p_out[yo + n] = p;
    }
  } else if (live) {
    p = p_in[yo + n];
  }
  // forcing = actuation + disturbance mu cos(2 + pi + x/(Lx/2)), x = dx (n+1)   (KSSetup.jl:36,155)
  const T force = p + (live ? e.dist_mu * (T)cos(2.0 + 3.14159265358979323846 + (double)e.dx * (n + 1) / ((double)e.dx * N / 2)) : (T)0);
  const T i2dx = (T)0.5 / e.dx, idx2 = (T)1 / (e.dx * e.dx), idx4 = idx2 * idx2;
  if (MODE == 2) {
    const T f = ksfd_rhs<T>(u, force, su, n, N, i2dx, idx2, idx4, live);
    if (live) y_out[yo + n] = f;
    return;
  }
  const T h = e.hstep;
  for (int it = 0; it < e.K; ++it) {
    const T k1 = ksfd_rhs<T>(u, force, su, n, N, i2dx, idx2, idx4, live);
    if (e.rk2) {     // PDEenv's built-in integrator (src/PDEenv.jl:208-214): explicit midpoint, `oversampling` sub-steps
      u = u + h * ksfd_rhs<T>(u + (T)0.5 * h * k1, force, su, n, N, i2dx, idx2, idx4, live);
      continue;
    }
    const T k2 = ksfd_rhs<T>(u + (T)0.5 * h * k1, force, su, n, N, i2dx, idx2, idx4, live);
    const T k3 = ksfd_rhs<T>(u + (T)0.5 * h * k2, force, su, n, N, i2dx, idx2, idx4, live);
    const T k4 = ksfd_rhs<T>(u + h * k3, force, su, n, N, i2dx, idx2, idx4, live);
    u = u + h / (T)6 * (k1 + (T)2 * (k2 + k3) + k4);
  }
  if (live) y_out[yo + n] = u;
  if (done) {
    T m = (live && !(fabs(u) <= e.max_value)) ? (T)1 : (T)0;
    m = block_max<T>(m, red, tid, nt);
    if (tid == 0) done[b] = (e.check_max == 1 && m > 0) ? 1 : 0;
    if (MODE == 0 && e.check_max != 2) write_terminal<T>(e, b, e.check_max == 1 && m > 0, tid, nt);
  }
  if (MODE != 0) return;
  __syncthreads();
  if (live) {
    su[n] = u;
    su[N + n] = 0;
  }
  __syncthreads();
  sense_dots<T>(e, [&](int r, int nn) { return su[r * N + nn]; }, dots, part, tid, nt);
  const int rw = e.mono ? 1 : e.A;
  const size_t sw = e.mono ? (size_t)e.S : (size_t)e.A * e.ns;
  const T rmine = reward_traj<T>(e, dots, act, actp, reward_out + (size_t)b * rw, tid, nt);
  featurize_traj<T>(e, dots, state_prev ? state_prev + b * sw : nullptr, state_out + b * sw, tid, nt);
  if (e.rsum_out) ksfd_reward_partial<T>(e, rmine, red, tid, nt);
  if (done && e.check_max == 2) {
    __syncthreads();
    if (tid == 0) {
      T m = 0;
      for (int a = 0; a < rw; ++a)
        if (!(fabs(reward_out[(size_t)b * rw + a]) <= e.max_value)) m = 1;
      done[b] = m > 0 ? 1 : 0;
      write_terminal<T>(e, b, m > 0, 0, 1);
    }
  }
}

// ---- the same step with ONE WAVE per trajectory (N = 64 CPL; used at N = 256, the grid of configs C1 / C2): lane l keeps the CPL consecutive
// cells CPL l .. CPL l + CPL - 1 in registers, the two neighbours on either side come from lanes l -+ 1 (periodic) by four
// lane exchanges per right-hand side -- no LDS line, no workgroup barrier inside the 4 K right-hand sides of a control step
// (the form above: two barriers each) -- and a 64-thread workgroup fits beside the update passes on every CU in one round
// (the 256-thread form: 72 VGPRs on all four SIMDs, one workgroup per CU at a time beside the passes, two rounds at B = 512).
// Same stencils and the same order of operations per cell as ksfd_rhs / the RK4 above.
// value of the same register in lane l - 1 (FROM_BELOW) or l + 1, periodic over the 64 lanes: one DPP wave rotate per 32-bit word
// (gfx9 wave_ror:1 / wave_rol:1) instead of a ds_bpermute round trip through the LDS crossbar
template <bool FROM_BELOW>
__device__ __forceinline__ float lane_neighbour(float x) {
  constexpr int ctrl = FROM_BELOW ? 0x13C : 0x134;      // DPP_WF_RR1 : DPP_WF_RL1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xf, 0xf, false));
}
template <bool FROM_BELOW>
__device__ __forceinline__ double lane_neighbour(double x) {
  constexpr int ctrl = FROM_BELOW ? 0x13C : 0x134;
  const unsigned long long b = __builtin_bit_cast(unsigned long long, x);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, ctrl, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), ctrl, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <class T, int CPL>
__device__ __forceinline__ void ksfd_rhs_wave(const T (&w)[CPL], const T (&force)[CPL], T (&f)[CPL], int up, int dn, T i2dx, T idx2, T idx4) {
  T ext[CPL + 4];
  (void)up; (void)dn;
  ext[0] = lane_neighbour<true>(w[CPL - 2]);
  ext[1] = lane_neighbour<true>(w[CPL - 1]);
  ext[CPL + 2] = lane_neighbour<false>(w[0]);
  ext[CPL + 3] = lane_neighbour<false>(w[1]);
#pragma unroll
  for (int c = 0; c < CPL; ++c) ext[c + 2] = w[c];
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const T m2 = ext[c], m1 = ext[c + 1], u = ext[c + 2], p1 = ext[c + 3], p2 = ext[c + 4];
    const T ux = i2dx * (p1 - m1);
    const T uxx = idx2 * (m1 - (T)2 * u + p1);
    const T uxxxx = idx4 * (m2 - (T)4 * m1 + (T)6 * u - (T)4 * p1 + p2);
    f[c] = -u * ux - uxx - uxxxx + force[c];
  }
}

template <class T, int CPL>
__global__ void __launch_bounds__(64) ksfd_wave_step_kernel(EnvDev<T> e, const T* __restrict__ y_in, const T* __restrict__ action,
                                                            const T* __restrict__ action_prev, const T* __restrict__ state_prev,
                                                            T* __restrict__ y_out, T* __restrict__ p_out, T* __restrict__ state_out,
                                                            T* __restrict__ reward_out, int32_t* __restrict__ done) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = 64, b = blockIdx.x;
  T* su = reinterpret_cast<T*>(smem_raw);  // [2][N] sensing image
  T* act = su + 2 * N + 4;                 // [A]
  T* actp = act + e.A;                     // [A]
  T* dots = actp + e.A;                    // [2][S]
  T* part = dots + 2 * e.S;                // [8][2][S]
  T* red = part + 16 * e.S;                // [16]
  set_wave_prio(e.prio);
  const size_t yo = (size_t)b * N;
  const int n0 = CPL * tid, up = (tid + 63) & 63, dn = (tid + 1) & 63;
  T u[CPL], force[CPL];
#pragma unroll
  for (int c = 0; c < CPL; ++c) u[c] = y_in[yo + n0 + c];
  for (int a = tid; a < e.A; a += nt) {
    act[a] = action[(size_t)b * e.A + a];
    actp[a] = action_prev[(size_t)b * e.A + a];
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    const int n = n0 + c;
    const T p = actuate_cell<T>(e, act, n);
    if (p_out) p_out[yo + n] = p;
    // forcing = actuation + disturbance mu cos(2 + pi + x/(Lx/2)), x = dx (n+1)   (KSSetup.jl:36,155)
    force[c] = p + e.dist_mu * (T)cos(2.0 + 3.14159265358979323846 + (double)e.dx * (n + 1) / ((double)e.dx * N / 2));
  }
  const T i2dx = (T)0.5 / e.dx, idx2 = (T)1 / (e.dx * e.dx), idx4 = idx2 * idx2, h = e.hstep;
  for (int it = 0; it < e.K; ++it) {
    T k1[CPL], k2[CPL], k3[CPL], k4[CPL], w[CPL];
    ksfd_rhs_wave<T, CPL>(u, force, k1, up, dn, i2dx, idx2, idx4);
#pragma unroll
    for (int c = 0; c < CPL; ++c) w[c] = u[c] + (T)0.5 * h * k1[c];
    ksfd_rhs_wave<T, CPL>(w, force, k2, up, dn, i2dx, idx2, idx4);
    if (e.rk2) {     // PDEenv's built-in integrator (src/PDEenv.jl:208-214): explicit midpoint, `oversampling` sub-steps
#pragma unroll
      for (int c = 0; c < CPL; ++c) u[c] = u[c] + h * k2[c];
      continue;
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) w[c] = u[c] + (T)0.5 * h * k2[c];
    ksfd_rhs_wave<T, CPL>(w, force, k3, up, dn, i2dx, idx2, idx4);
#pragma unroll
    for (int c = 0; c < CPL; ++c) w[c] = u[c] + h * k3[c];
    ksfd_rhs_wave<T, CPL>(w, force, k4, up, dn, i2dx, idx2, idx4);
#pragma unroll
    for (int c = 0; c < CPL; ++c) u[c] = u[c] + h / (T)6 * (k1[c] + (T)2 * (k2[c] + k3[c]) + k4[c]);
  }
  T m = 0;
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    y_out[yo + n0 + c] = u[c];
    if (!(fabs(u[c]) <= e.max_value)) m = 1;
  }
  if (done) {
    m = block_max<T>(m, red, tid, nt);
    if (tid == 0) done[b] = (e.check_max == 1 && m > 0) ? 1 : 0;
    if (e.check_max != 2) write_terminal<T>(e, b, e.check_max == 1 && m > 0, tid, nt);
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < CPL; ++c) {
    su[n0 + c] = u[c];
    su[N + n0 + c] = 0;
  }
  __syncthreads();
  sense_dots<T>(e, [&](int r, int nn) { return su[r * N + nn]; }, dots, part, tid, nt);
  const int rw = e.mono ? 1 : e.A;
  const size_t sw = e.mono ? (size_t)e.S : (size_t)e.A * e.ns;
  const T rmine = reward_traj<T>(e, dots, act, actp, reward_out + (size_t)b * rw, tid, nt);
  featurize_traj<T>(e, dots, state_prev ? state_prev + b * sw : nullptr, state_out + b * sw, tid, nt);
  if (e.rsum_out) ksfd_reward_partial<T>(e, rmine, red, tid, nt);
  if (done && e.check_max == 2) {
    __syncthreads();
    if (tid == 0) {
      T mm = 0;
      for (int a = 0; a < rw; ++a)
        if (!(fabs(reward_out[(size_t)b * rw + a]) <= e.max_value)) mm = 1;
      done[b] = mm > 0 ? 1 : 0;
      write_terminal<T>(e, b, mm > 0, 0, 1);
    }
  }
}

// ------------------------------------------------------------------ stand-alone closures
// MODE 0: prepare_action, 1: featurize, 2: reward
template <class T, int MODE>
__global__ void sense_kernel(EnvDev<T> e, const T* __restrict__ y, const T* __restrict__ action,
                             const T* __restrict__ action_prev, const T* __restrict__ state_prev,
                             T* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int N = e.N, tid = threadIdx.x, nt = blockDim.x, b = blockIdx.x;
  T* sy = reinterpret_cast<T*>(smem_raw);  // [2][N]
  T* act = sy + 2 * N;
  T* actp = act + e.A;
  T* dots = actp + e.A;
  T* part = dots + 2 * e.S;
  if (MODE == 0 || MODE == 2) {     // row 0 of every action column (KSSetup.jl:176,241: action[1, i]); na = 1 without action memory
    for (int a = tid; a < e.A; a += nt) {
      act[a] = action[((size_t)b * e.A + a) * e.na];
      actp[a] = MODE == 2 ? action_prev[((size_t)b * e.A + a) * e.na] : (T)0;
    }
    __syncthreads();
  }
  if (MODE == 0) {
    for (int n = tid; n < N; n += nt) out[(size_t)b * N + n] = actuate_cell<T>(e, act, n);
    return;
  }
  const int sp = e.n_species;
  for (int i = tid; i < sp * N; i += nt) {
    // y[sp, N] column-major: (species, cell) at cell*sp + species
    const int n = i / sp, r = i - n * sp;
    sy[r * N + n] = y[(size_t)b * sp * N + i];
  }
  if (sp == 1)
    for (int n = tid; n < N; n += nt) sy[N + n] = 0;
  __syncthreads();
  sense_dots<T>(e, [&](int r, int n) { return sy[r * N + n]; }, dots, part, tid, nt);
  if (MODE == 1) {
    const size_t sw = e.mono ? (size_t)e.S : (size_t)e.A * e.ns;
    if (e.mem > 0)
      featurize_traj_mem<T>(e, dots, state_prev ? state_prev + b * sw : nullptr, action ? action + (size_t)b * e.A * e.na : nullptr,
                            out + b * sw, tid, nt);
    else
      featurize_traj<T>(e, dots, state_prev ? state_prev + b * sw : nullptr, out + b * sw, tid, nt);
  } else {
    const int rw = e.mono ? 1 : e.A;
    reward_traj<T>(e, dots, act, actp, out + (size_t)b * rw, tid, nt);
  }
}

// terminal flag per actuator column from the per-trajectory blow-up flags (composed env step; the fused kernels write it themselves)
template <class T>
__global__ void terminal_from_done_kernel(const int32_t* __restrict__ done, int B, int cpt, T* __restrict__ term) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * cpt) term[i] = (done[i / cpt] & 1) ? (T)1 : (T)0;
}

// ------------------------------------------------------------------ host side
template <class T>
static EnvDev<T> make_dev(const Env& E) {
  const pdec_env_cfg& c = E.cfg;
  EnvDev<T> e;
  e.B = c.B; e.N = c.N; e.S = c.S; e.A = c.A; e.window = c.window; e.temporal = c.temporal_steps;
  e.mono = c.mono; e.K = c.K; e.check_max = c.check_max_value; e.n_species = c.n_species;
  e.mem = c.memory_size; e.na = 1 + c.memory_size;      // (action memory: stand-alone closures + the composed env step only)
  e.ns = c.mono ? c.S : c.window * c.n_species * c.temporal_steps + c.memory_size;
  e.sensor_scale = (T)c.sensor_scale; e.agent_power = (T)c.agent_power;
  e.r_in_scale = (T)c.reward_in_scale; e.r_offset = (T)c.reward_offset; e.r_power = (T)c.reward_power;
  e.r_denom = (T)c.reward_denom; e.a_pun = (T)c.action_punish; e.da_pun = (T)c.delta_action_punish;
  e.max_value = (T)c.max_value;
  e.dx = (T)(c.Lx / c.N);
  e.hstep = (T)(c.dt / c.K);
  e.rk2 = c.integrator == 1;
  e.prio = env_prio("PDEC_PRIO_KS", (E.share_simd && E.r4_log == 1 && c.dtype == PDEC_F32) ? 3 : 1);
  e.dist_mu = (T)c.mu;
  e.Gs = E.Gs.as<T>(); e.sn0 = E.sn0.as<int>(); e.GaC = E.GaC.as<T>(); e.an0 = E.an0.as<int>();
  e.Wd = E.Wd; e.Cnt = E.Cnt;
  e.gsum = E.gsum.as<T>(); e.a2s = E.a2s.as<int>();
  e.fmap = E.fmap.p ? E.fmap.as<int>() : nullptr;
  e.term_out = static_cast<T*>(E.term_out);
  e.rsum_out = E.rsum_out;
  e.c1 = E.c1.as<T>(); e.c2 = E.c2.as<T>(); e.c3 = E.c3.as<T>(); e.c4 = E.c4.as<T>(); e.g = E.g.as<T>();
  e.dhat = E.dhat.as<C2<T>>(); e.tw = E.tw.as<C2<T>>();
  e.fft = E.fft;
  return e;
}

static size_t ks_lds_bytes(const pdec_env_cfg& c, int r4_log) {
  const size_t ts = dtype_size(c.dtype);
  return (r4_log == 1 ? 1 : ((r4_log == 4 || r4_log == 5 || r4_log == 10) ? 2 : 3)) * (size_t)c.N * 2 * ts + (4 * (size_t)c.A + 2 * c.S + 16 * c.S + 16) * ts;
}
static size_t kseg_lds_bytes(const pdec_env_cfg& c) {
  const size_t ts = dtype_size(c.dtype);
  return (2 * ((size_t)c.N + 2) + 2 * c.A + 2 * c.S + 16 * c.S + 16) * ts;
}
static size_t ksfd_lds_bytes(const pdec_env_cfg& c) {
  const size_t ts = dtype_size(c.dtype);
  return (2 * (size_t)c.N + 4 + 2 * c.A + 2 * c.S + 16 * c.S + 16) * ts;
}
static size_t sense_lds_bytes(const pdec_env_cfg& c) {
  const size_t ts = dtype_size(c.dtype);
  return (2 * (size_t)c.N + 2 * c.A + 2 * c.S + 16 * c.S) * ts;
}

#define COMMA ,
template <class T>
static int launch_step(Env& E, bool fused, int mode, const void* y_in, const void* p, const void* action,
                       const void* action_prev, const void* state_prev, void* y_out, void* p_out,
                       void* state_out, void* reward_out, int32_t* done) {
  EnvDev<T> e = make_dev<T>(E);
  const pdec_env_cfg& c = E.cfg;
  const LaunchSync sync = E.sync;
  E.sync = LaunchSync{};
  if (sync.wait || sync.done) {        // served by the SYNC instantiations below, refused everywhere else
    const bool ok = c.pde_kind == PDEC_PDE_KS_CNAB2 && fused && c.B <= 2 && sizeof(T) == 8 && (E.r4_log == 7 || E.r4_log == 8 || E.r4_log == 9) &&
                    !(E.prof && E.prof_reps > 1);
    PDEC_REQUIRE(ok, "pdec_env_step: a launch sync is set (pdec_set_launch_sync) and this step is not the fused single-workgroup fp64 "
                     "KS step of 192 / 240 / 600 cells");
    if constexpr (sizeof(T) == 8) {
      e.sync = sync;
      const dim3 grid1(1), block1(E.nthreads);
      ProfScope ps(&E, "ks_env_step");
#define KS_SYNC_LAUNCH(ENG)                                                                                                       \
  hipLaunchKernelGGL((ks_env_step_kernel<T, ENG, true, false, true>), grid1, block1, E.lds_bytes, E.stream, e, (const T*)y_in,   \
                     (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out, (T*)p_out,           \
                     (T*)state_out, (T*)reward_out, done)
      if (E.r4_log == 7) KS_SYNC_LAUNCH(FftFixed192<T>);
      else if (E.r4_log == 8) KS_SYNC_LAUNCH(FftFixed240<T>);
      else KS_SYNC_LAUNCH(FftFixed600<T>);
#undef KS_SYNC_LAUNCH
      PDEC_HIP(hipGetLastError());
      return PDEC_OK;
    }
  }
  if (c.pde_kind == PDEC_PDE_KS_CNAB2) {
    dim3 grid((c.B + 1) / 2), block(E.nthreads);
    // replay is safe when the step does not run in place (y_out != y_in)
    // the training pipeline's form of the step, profiled in the pipeline (one launch per event pair): timed by the
    // dispatch's own timestamps (PDEC_TIMED_LAUNCH) so that the measurement puts no packets around the kernel
    if (E.r4_log == 1 && fused && E.prof && E.prof_reps == 1) {
      if constexpr (sizeof(T) == 4) {
#define KS_ARGS e, (const T*)y_in, (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out, (T*)p_out, \
                (T*)state_out, (T*)reward_out, done
        if (E.share_simd)
          PDEC_TIMED_LAUNCH(&E, "ks_env_step", (ks_env_step_kernel<T, FftWave256<T>, true, true>), grid, block, E.lds_bytes + 16 + 8 * 16 * 64, KS_ARGS);
        else
          PDEC_TIMED_LAUNCH(&E, "ks_env_step", (ks_env_step_kernel<T, FftWave256<T>, true>), grid, block, E.lds_bytes, KS_ARGS);
#undef KS_ARGS
        PDEC_HIP(hipGetLastError());
        return PDEC_OK;
      }
    }
    ProfScope ps(&E, fused ? "ks_env_step" : "ks_pde_step", y_out != y_in && state_out != state_prev);
#define KS_LAUNCH(ENG, F)                                                                                      \
  for (int rep__ = 0; rep__ < ps.reps; ++rep__)                                                                \
  hipLaunchKernelGGL((ks_env_step_kernel<T, ENG, F>), grid, block, E.lds_bytes, E.stream, e, (const T*)y_in,    \
                     (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out,    \
                     (T*)p_out, (T*)state_out, (T*)reward_out, done)
    if (E.r4_log == 1) {
      if constexpr (sizeof(T) == 4) {
        if (fused && E.share_simd) {     // 64-VGPR form + its lane-private constant slots (8 float4 per lane, 16-byte aligned)
          for (int rep__ = 0; rep__ < ps.reps; ++rep__)
            hipLaunchKernelGGL((ks_env_step_kernel<T, FftWave256<T>, true, true>), grid, block, E.lds_bytes + 16 + 8 * 16 * 64, E.stream, e,
                               (const T*)y_in, (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out,
                               (T*)p_out, (T*)state_out, (T*)reward_out, done);
          PDEC_HIP(hipGetLastError());
          return PDEC_OK;
        }
      }
      if (fused) KS_LAUNCH(FftWave256<T>, true); else KS_LAUNCH(FftWave256<T>, false);
    }
    else if (E.r4_log == 4) { if (fused) KS_LAUNCH(FftR4<T COMMA 4>, true); else KS_LAUNCH(FftR4<T COMMA 4>, false); }
    else if (E.r4_log == 5) { if (fused) KS_LAUNCH(FftR4<T COMMA 5>, true); else KS_LAUNCH(FftR4<T COMMA 5>, false); }
    else if (E.r4_log == 10) { if (fused) KS_LAUNCH(FftWave1024<T>, true); else KS_LAUNCH(FftWave1024<T>, false); }
    else if (E.r4_log == 7) { if (fused) KS_LAUNCH(FftFixed192<T>, true); else KS_LAUNCH(FftFixed192<T>, false); }
    else if (E.r4_log == 8) { if (fused) KS_LAUNCH(FftFixed240<T>, true); else KS_LAUNCH(FftFixed240<T>, false); }
    else if (E.r4_log == 9) { if (fused) KS_LAUNCH(FftFixed600<T>, true); else KS_LAUNCH(FftFixed600<T>, false); }
    else { if (fused) KS_LAUNCH(FftGeneric<T>, true); else KS_LAUNCH(FftGeneric<T>, false); }
#undef KS_LAUNCH
  } else if (c.pde_kind == PDEC_PDE_KSEG_RK4) {
    dim3 grid(c.B), block(E.nthreads);
    ProfScope ps(&E, mode == 0 ? "kseg_env_step" : (mode == 1 ? "kseg_pde_step" : "kseg_rhs"));
#define KSEG_LAUNCH(M)                                                                                        \
  hipLaunchKernelGGL((kseg_env_step_kernel<T, M>), grid, block, E.lds_bytes, E.stream, e, (const T*)y_in,      \
                     (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out,   \
                     (T*)p_out, (T*)state_out, (T*)reward_out, done)
    if (mode == 0) KSEG_LAUNCH(0);
    else if (mode == 1) KSEG_LAUNCH(1);
    else KSEG_LAUNCH(2);
#undef KSEG_LAUNCH
  } else if (c.pde_kind == PDEC_PDE_KS_RK4_FD) {
    dim3 grid(c.B), block(E.nthreads);
    ProfScope ps(&E, mode == 0 ? "ksfd_env_step" : (mode == 1 ? "ksfd_pde_step" : "ksfd_rhs"));
#define KSFD_LAUNCH(M)                                                                                        \
  hipLaunchKernelGGL((ksfd_env_step_kernel<T, M>), grid, block, E.lds_bytes, E.stream, e, (const T*)y_in,      \
                     (const T*)p, (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out,   \
                     (T*)p_out, (T*)state_out, (T*)reward_out, done)
    // fused step at N = 256: one wave per trajectory (ksfd_wave_step_kernel); PDEC_KSFD_LDS=1: the general form
    const bool lds_form = getenv("PDEC_KSFD_LDS") != nullptr;
#define KSFD_WAVE(CPL)                                                                                                 \
  hipLaunchKernelGGL((ksfd_wave_step_kernel<T, CPL>), grid, dim3(64), E.lds_bytes, E.stream, e, (const T*)y_in,           \
                     (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)y_out, (T*)p_out, (T*)state_out,  \
                     (T*)reward_out, done)
    if (mode == 0 && !lds_form && c.N == 256) KSFD_WAVE(4);      // (N = 1024 would need 168 VGPRs per wave: no room beside the passes)
    else if (mode == 0) KSFD_LAUNCH(0);
    else if (mode == 1) KSFD_LAUNCH(1);
    else KSFD_LAUNCH(2);
#undef KSFD_WAVE
#undef KSFD_LAUNCH
  } else {
    set_error("pde_kind %d not implemented", c.pde_kind);
    return PDEC_E_INVALID;
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// ---- persistent rollout: host side
static size_t ks_rollout_lds(const Env& E, const Mlp& A) {
  const pdec_env_cfg& c = E.cfg;
  const size_t ts = dtype_size(c.dtype);
  return E.lds_bytes + ((size_t)2 * c.A * env_ns(c) + 4 * (size_t)c.A + ((ro_image_elems(A.dims.data(), A.L) + 3) & ~3) + (size_t)4 * *std::max_element(A.dims.begin(), A.dims.begin() + A.L + 1) * E.nthreads) * ts + 16;
}
bool ks_rollout_supported(const Env& E, const Mlp& A) {
  const pdec_env_cfg& c = E.cfg;
  const char* off = getenv("PDEC_ROLLOUT_PERSISTENT");
  if (off && off[0] == '0') return false;
  if (c.pde_kind != PDEC_PDE_KS_CNAB2 || c.mono || c.temporal_steps != 1 || c.check_max_value == 2) return false;
  if ((E.nthreads & 1) || A.L < 1 || A.L > 3 || A.dims[A.L] != 1 || A.dtype != c.dtype || A.dims[0] != env_ns(c)) return false;
  for (int l = 0; l <= A.L; ++l)
    if (A.dims[l] > RO_W) return false;
  return ks_rollout_lds(E, A) <= 64 * 1024;
}

template <class T>
static int ks_rollout_launch(Env& E, const Mlp& A, const RollArgs<T>& g) {
  EnvDev<T> e = make_dev<T>(E);
  const pdec_env_cfg& c = E.cfg;
  RollActor ra{};
  ra.params = A.params.p; ra.L = A.L; ra.nparams = A.nparams;
  for (int l = 0; l <= A.L; ++l) { ra.dims[l] = A.dims[l]; ra.rows = std::max(ra.rows, A.dims[l]); }
  for (int l = 0; l < A.L; ++l) ra.acts[l] = A.acts[l];
  const size_t lds = ks_rollout_lds(E, A);
  dim3 grid((c.B + 1) / 2), block(E.nthreads);
  ProfScope ps(&E, "ks_rollout");
#define KS_ROLL(ENG) hipLaunchKernelGGL((ks_rollout_kernel<T, ENG>), grid, block, lds, E.stream, e, ra, g)
  if (E.r4_log == 1) KS_ROLL(FftWave256<T>);
  else if (E.r4_log == 4) KS_ROLL(FftR4<T COMMA 4>);
  else if (E.r4_log == 5) KS_ROLL(FftR4<T COMMA 5>);
  else if (E.r4_log == 10) KS_ROLL(FftWave1024<T>);
  else if (E.r4_log == 7) KS_ROLL(FftFixed192<T>);
  else if (E.r4_log == 8) KS_ROLL(FftFixed240<T>);
  else if (E.r4_log == 9) KS_ROLL(FftFixed600<T>);
  else KS_ROLL(FftGeneric<T>);
#undef KS_ROLL
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int ks_rollout_persistent(Env& E, const Mlp& A, int T, void* y, void* state, void* action, double act_noise, double act_limit,
                          int learning, uint64_t seed, uint64_t offset, void* reward_sum, void* log_y, void* log_p,
                          void* log_action, void* log_reward, int32_t* done_any, int32_t* done_step) {
  if (!ks_rollout_supported(E, A)) { set_error("ks_rollout_persistent: configuration not covered"); return PDEC_E_INVALID; }
  if (E.cfg.dtype == PDEC_F64) {
    RollArgs<double> g{T, learning, act_noise, act_limit, seed, offset, (double*)y, (double*)state, (double*)action, (double*)reward_sum,
                       (double*)log_y, (double*)log_p, (double*)log_action, (double*)log_reward, done_any, done_step};
    return ks_rollout_launch<double>(E, A, g);
  }
  RollArgs<float> g{T, learning, (float)act_noise, (float)act_limit, seed, offset, (float*)y, (float*)state, (float*)action,
                    (float*)reward_sum, (float*)log_y, (float*)log_p, (float*)log_action, (float*)log_reward, done_any, done_step};
  return ks_rollout_launch<float>(E, A, g);
}

// Keller-Segel (1-D): the same service for kseg_rollout_kernel
static size_t kseg_rollout_lds(const Env& E, const Mlp& A) {
  const pdec_env_cfg& c = E.cfg;
  return E.lds_bytes + ((size_t)2 * c.A * env_ns(c) + 2 * (size_t)c.A + ((ro_image_elems(A.dims.data(), A.L) + 3) & ~3) + (size_t)2 * c.A * RO_W) * dtype_size(c.dtype) + 16;
}
bool kseg_rollout_supported(const Env& E, const Mlp& A) {
  const pdec_env_cfg& c = E.cfg;
  const char* off = getenv("PDEC_ROLLOUT_PERSISTENT");
  if (off && off[0] == '0') return false;
  if (c.pde_kind != PDEC_PDE_KSEG_RK4 || c.mono || c.check_max_value == 2) return false;
  if (A.L < 1 || A.L > 3 || A.dims[A.L] != 1 || A.dtype != c.dtype || A.dims[0] != env_ns(c)) return false;
  for (int l = 0; l <= A.L; ++l)
    if (A.dims[l] > RO_W) return false;
  return kseg_rollout_lds(E, A) <= 64 * 1024;
}
template <class T>
static int kseg_rollout_launch(Env& E, const Mlp& A, const RollArgs<T>& g) {
  EnvDev<T> e = make_dev<T>(E);
  RollActor ra{};
  ra.params = A.params.p; ra.L = A.L; ra.nparams = A.nparams;
  for (int l = 0; l <= A.L; ++l) { ra.dims[l] = A.dims[l]; ra.rows = std::max(ra.rows, A.dims[l]); }
  for (int l = 0; l < A.L; ++l) ra.acts[l] = A.acts[l];
  ProfScope ps(&E, "kseg_rollout");
  hipLaunchKernelGGL((kseg_rollout_kernel<T>), dim3(E.cfg.B), dim3(E.nthreads), kseg_rollout_lds(E, A), E.stream, e, ra, g);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}
int kseg_rollout_persistent(Env& E, const Mlp& A, int T, void* y, void* state, void* action, double act_noise, double act_limit,
                            int learning, uint64_t seed, uint64_t offset, void* reward_sum, void* log_y, void* log_p,
                            void* log_action, void* log_reward, int32_t* done_any, int32_t* done_step) {
  if (!kseg_rollout_supported(E, A)) { set_error("kseg_rollout_persistent: configuration not covered"); return PDEC_E_INVALID; }
  if (E.cfg.dtype == PDEC_F64) {
    RollArgs<double> g{T, learning, act_noise, act_limit, seed, offset, (double*)y, (double*)state, (double*)action, (double*)reward_sum,
                       (double*)log_y, (double*)log_p, (double*)log_action, (double*)log_reward, done_any, done_step};
    return kseg_rollout_launch<double>(E, A, g);
  }
  RollArgs<float> g{T, learning, (float)act_noise, (float)act_limit, seed, offset, (float*)y, (float*)state, (float*)action,
                    (float*)reward_sum, (float*)log_y, (float*)log_p, (float*)log_action, (float*)log_reward, done_any, done_step};
  return kseg_rollout_launch<float>(E, A, g);
}

template <class T>
static int launch_sense(Env& E, int mode, const void* y, const void* action, const void* action_prev,
                        const void* state_prev, void* out) {
  EnvDev<T> e = make_dev<T>(E);
  dim3 grid(E.cfg.B), block(128);
  const size_t lds = sense_lds_bytes(E.cfg);
  ProfScope ps(&E, mode == 0 ? "actuate" : (mode == 1 ? "featurize" : "reward"));
#define SENSE_LAUNCH(M)                                                                                 \
  hipLaunchKernelGGL((sense_kernel<T, M>), grid, block, lds, E.stream, e, (const T*)y, (const T*)action, \
                     (const T*)action_prev, (const T*)state_prev, (T*)out)
  if (mode == 0) SENSE_LAUNCH(0);
  else if (mode == 1) SENSE_LAUNCH(1);
  else SENSE_LAUNCH(2);
#undef SENSE_LAUNCH
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// (env::PDEenv)(action) with action memory (cfg.memory_size > 0): the closures one after the other on the environment's
// stream -- prepare_action (row 0 of every action column), the integrator, reward (row 0), featurize (window rows, temporal
// stack, memory rows = rows 1.. of the action).  No shipped script sets memory_size, so this reference surface is served by
// four launches instead of one and the fused step kernels keep their register budgets.  src/PDEenv.jl:195-241.
template <class T>
static int env_step_composed(Env& E, const void* y_in, const void* action, const void* action_prev, const void* state_prev,
                             void* y_out, void* p_out, void* state_out, void* reward_out, int32_t* done) {
  const pdec_env_cfg& c = E.cfg;
  const size_t need = (size_t)c.B * env_p_count(c) * sizeof(T) + (size_t)c.B * sizeof(int32_t) + 64;
  if (E.mem_scratch.bytes < need) PDEC_HIP(E.mem_scratch.alloc(need));
  void* p = p_out ? p_out : E.mem_scratch.p;
  int32_t* flags = done ? done : reinterpret_cast<int32_t*>(E.mem_scratch.as<char>() + (size_t)c.B * env_p_count(c) * sizeof(T));
  int rc;
  if ((rc = launch_sense<T>(E, 0, nullptr, action, nullptr, nullptr, p))) return rc;
  if ((rc = launch_step<T>(E, false, 1, y_in, p, nullptr, nullptr, nullptr, y_out, nullptr, nullptr, nullptr, flags))) return rc;
  if ((rc = launch_sense<T>(E, 2, y_out, action, action_prev, nullptr, reward_out))) return rc;
  if ((rc = launch_sense<T>(E, 1, y_out, action, nullptr, state_prev, state_out))) return rc;
  if (E.term_out) {
    const int n = c.B * c.A;
    hipLaunchKernelGGL((terminal_from_done_kernel<T>), dim3((n + 255) / 256), dim3(256), 0, E.stream, flags, c.B, c.A, (T*)E.term_out);
    PDEC_HIP(hipGetLastError());
  }
  return PDEC_OK;
}

}  // namespace pdec

using namespace pdec;

extern "C" {

int pdec_env_create(pdec_handle* h, const pdec_env_cfg* cfg, const double* sensor_kernels,
                    const double* actuator_kernels, const int32_t* a2s) {
  PDEC_REQUIRE(h && cfg && sensor_kernels && actuator_kernels && a2s, "pdec_env_create: null argument");
  const pdec_env_cfg& c = *cfg;
  PDEC_REQUIRE(c.dtype == PDEC_F32 || c.dtype == PDEC_F64, "pdec_env_create: bad dtype %d", c.dtype);
  PDEC_REQUIRE(c.B >= 1 && c.N >= 4 && c.S >= 1 && c.A >= 1 && c.K >= 1, "pdec_env_create: bad sizes B=%d N=%d S=%d A=%d K=%d",
               c.B, c.N, c.S, c.A, c.K);
  PDEC_REQUIRE(c.window >= 1 && (c.window & 1) && c.temporal_steps >= 1, "pdec_env_create: window must be odd >= 1");
  PDEC_REQUIRE(c.window <= c.S || c.mono, "pdec_env_create: window %d larger than sensor count %d", c.window, c.S);
  PDEC_REQUIRE(c.Lx > 0 && c.dt > 0, "pdec_env_create: Lx and dt must be positive");
  PDEC_REQUIRE(c.integrator == 0 || (c.integrator == 1 && (c.pde_kind == PDEC_PDE_KSEG_RK4 || c.pde_kind == PDEC_PDE_KS_RK4_FD)),
               "pdec_env_create: integrator %d is not available for pde_kind %d", c.integrator, c.pde_kind);
  PDEC_REQUIRE(c.memory_size >= 0 && c.memory_size <= 64, "pdec_env_create: memory_size %d out of range", c.memory_size);
  PDEC_REQUIRE(c.memory_size == 0 || (!c.mono && c.check_max_value != 2),
               "pdec_env_create: action memory is built for the per-actuator environments with check_max_value 0 / 1 "
               "(KSSetup.jl:216-226); the global-agent form and the reward-based blow-up test are not");
  for (int a = 0; a < c.A && !c.mono; ++a)
    PDEC_REQUIRE(a2s[a] >= 0 && a2s[a] < c.S, "pdec_env_create: a2s[%d]=%d out of range", a, a2s[a]);
  auto E = std::make_unique<Env>();
  E->cfg = c;
  const int N = c.N;
  std::vector<int32_t> a2s_h(a2s, a2s + c.A);
  if (c.mono)
    for (int a = 0; a < c.A; ++a) PDEC_REQUIRE(a2s_h[a] >= 0 && a2s_h[a] < c.S, "pdec_env_create: a2s out of range");
  if (c.pde_kind == PDEC_PDE_KS_CNAB2) {
    PDEC_REQUIRE(c.n_species == 1, "KS has one species");
    PDEC_REQUIRE(N % 2 == 0, "KS CNAB2 needs even N (Nyquist slot, KSSetup.jl:115)");
    PDEC_REQUIRE(make_fft_plan(N, E->fft), "N=%d has a prime factor other than 2,3,5", N);
    int nt = ((N + KS_MPT - 1) / KS_MPT + 63) / 64 * 64;
    PDEC_REQUIRE(nt <= 1024, "N=%d too large for the in-LDS KS kernel (max 4096)", N);
    E->nthreads = nt;
    // engines: 1 = single-wave register FFT (N = 256), 10 = four waves x the same + one cross-wave stage (N = 1024),
    // 4 / 5 = radix-4 through LDS (N = 256 / 1024, PDEC_KS_LDS_FFT=1), 0 = generic
    E->r4_log = (N == 256 && !getenv("PDEC_KS_GENERIC_FFT")) ? (getenv("PDEC_KS_LDS_FFT") ? 4 : 1)
                                                              : ((N == 1024 && !getenv("PDEC_KS_GENERIC_FFT")) ? (getenv("PDEC_KS_LDS_FFT") ? 5 : 10) : 0);
    if (E->r4_log == 0 && !getenv("PDEC_KS_GENERIC_FFT")) {
      // engines 7 / 8 / 9: compile-time plans for the grids of the shipped experiments (KS22, KS200, KS500); one
      // butterfly per thread and stage -> nt = the largest N / radix, rounded up to whole waves
      if (N == 192) { E->r4_log = 7; nt = 64; }
      else if (N == 240) { E->r4_log = 8; nt = 128; }
      else if (N == 600) { E->r4_log = 9; nt = 320; }
      E->nthreads = nt;
    }
    E->lds_bytes = ks_lds_bytes(c, E->r4_log);
    PDEC_REQUIRE(E->lds_bytes <= 160 * 1024, "KS kernel needs %zu B of LDS (> 160 KiB)", E->lds_bytes);
    // per-mode constants, scripts/KS/setup/KSSetup.jl:115-123,131-135
    std::vector<double> c1(N), c2(N), c3(N), c4(N), g(N), dh(2 * N), tw(2 * N), dist(N);
    const double hh = c.dt / c.K, dt2 = hh / 2, dt32 = 3 * hh / 2, dx = c.Lx / N;
    for (int k = 0; k < N; ++k) {
      double kx = k < N / 2 ? k : (k == N / 2 ? 0 : k - N);
      double al = 2 * M_PI * kx / c.Lx;
      double L = al * al - al * al * al * al;
      double Ainv = 1.0 / (1.0 - dt2 * L), Bc = 1.0 + dt2 * L;
      c1[k] = Ainv * Bc; c2[k] = Ainv * dt32; c3[k] = Ainv * dt2; c4[k] = Ainv * hh; g[k] = -0.5 * al;
      tw[2 * k] = cos(2 * M_PI * k / N);
      tw[2 * k + 1] = -sin(2 * M_PI * k / N);
      dist[k] = c.mu * cos(2 + M_PI + (dx * (k + 1)) / (c.Lx / 2));  // KSSetup.jl:155
    }
    for (int k = 0; k < N; ++k) {  // O(N^2) host DFT, setup time only
      double re = 0, im = 0;
      if (c.mu != 0.0)
        for (int n = 0; n < N; ++n) {
          long long ph = ((long long)k * n) % N;
          re += dist[n] * tw[2 * ph];
          im += dist[n] * tw[2 * ph + 1];
        }
      dh[2 * k] = hh * re;
      dh[2 * k + 1] = hh * im;
    }
    int rc;
    if ((rc = upload_converted(E->c1, c1.data(), N, c.dtype))) return rc;
    if ((rc = upload_converted(E->c2, c2.data(), N, c.dtype))) return rc;
    if ((rc = upload_converted(E->c3, c3.data(), N, c.dtype))) return rc;
    if ((rc = upload_converted(E->c4, c4.data(), N, c.dtype))) return rc;
    if ((rc = upload_converted(E->g, g.data(), N, c.dtype))) return rc;
    if ((rc = upload_converted(E->dhat, dh.data(), 2 * N, c.dtype))) return rc;
    if ((rc = upload_converted(E->tw, tw.data(), 2 * N, c.dtype))) return rc;
  } else if (c.pde_kind == PDEC_PDE_KS_RK4_FD) {
    PDEC_REQUIRE(c.n_species == 1, "KS has one species");
    int nt = (N + 63) / 64 * 64;
    PDEC_REQUIRE(nt <= 1024, "N=%d too large for the one-cell-per-thread KS finite-difference kernel (max 1024)", N);
    E->nthreads = nt;
    E->lds_bytes = ksfd_lds_bytes(c);
  } else if (c.pde_kind == PDEC_PDE_KSEG_RK4) {
    PDEC_REQUIRE(c.n_species == 2, "Keller-Segel has two species");
    PDEC_REQUIRE(!c.mono, "Keller-Segel has no mono variant");
    int nt = (N + 63) / 64 * 64;
    PDEC_REQUIRE(nt <= 1024, "N=%d too large for the one-cell-per-thread K-S kernel (max 1024)", N);
    E->nthreads = nt;
    E->lds_bytes = kseg_lds_bytes(c);
  } else {
    set_error("pdec_env_create: pde_kind %d not implemented", c.pde_kind);
    return PDEC_E_INVALID;
  }
  PDEC_REQUIRE(sense_lds_bytes(c) <= 160 * 1024, "sensor kernels need too much LDS");
  // tables: circular band form of the dense kernels.  An entry counts as non-zero if it is non-zero
  // in the plan's dtype, so the band product equals the dense product term by term.
  auto nz = [&](double v) { return c.dtype == PDEC_F64 ? v != 0.0 : (float)v != 0.0f; };
  // window of a circular 0/1 pattern: start after the longest run of zeros
  auto window = [&](const std::vector<char>& m, int& start, int& len) {
    const int n = (int)m.size();
    int best = -1, bestpos = 0, run = 0;
    bool any = false;
    for (int i = 0; i < n; ++i) any |= m[i] != 0;
    if (!any) { start = 0; len = 0; return; }
    for (int i = 0; i < 2 * n; ++i) {       // longest zero run on the ring
      if (!m[i % n]) { if (++run > best && run <= n) { best = run; bestpos = i; } }
      else run = 0;
    }
    if (best <= 0) { start = 0; len = n; return; }
    start = (bestpos + 1) % n;
    len = n - best;
  };
  std::vector<double> gs(c.S, 0.0);
  std::vector<int32_t> sn0(c.S), slen(c.S), an0(N), alen(N);
  int Wd = 1, Cnt = 1;
  for (int s = 0; s < c.S; ++s) {
    std::vector<char> m(N);
    for (int n = 0; n < N; ++n) { m[n] = nz(sensor_kernels[(size_t)s * N + n]); gs[s] += sensor_kernels[(size_t)s * N + n]; }
    int st, ln; window(m, st, ln);
    sn0[s] = st; slen[s] = ln; Wd = std::max(Wd, ln);
  }
  for (int n = 0; n < N; ++n) {
    std::vector<char> m(c.A);
    for (int a = 0; a < c.A; ++a) m[a] = nz(actuator_kernels[(size_t)a * N + n]);
    int st, ln; window(m, st, ln);
    an0[n] = st; alen[n] = ln; Cnt = std::max(Cnt, ln);
  }
  std::vector<double> Gs((size_t)Wd * c.S, 0.0), GaC((size_t)Cnt * N, 0.0);
  for (int s = 0; s < c.S; ++s)
    for (int j = 0; j < slen[s]; ++j) Gs[(size_t)j * c.S + s] = sensor_kernels[(size_t)s * N + (sn0[s] + j) % N];
  for (int n = 0; n < N; ++n)
    for (int i = 0; i < alen[n]; ++i) GaC[(size_t)i * N + n] = actuator_kernels[(size_t)((an0[n] + i) % c.A) * N + n];
  E->Wd = Wd; E->Cnt = Cnt;
  int rc;
  if ((rc = upload_converted(E->Gs, Gs.data(), Gs.size(), c.dtype))) return rc;
  if ((rc = upload_converted(E->GaC, GaC.data(), GaC.size(), c.dtype))) return rc;
  if ((rc = upload_converted(E->gsum, gs.data(), gs.size(), c.dtype))) return rc;
  PDEC_HIP(E->sn0.alloc(sizeof(int32_t) * c.S));
  PDEC_HIP(hipMemcpy(E->sn0.p, sn0.data(), sizeof(int32_t) * c.S, hipMemcpyHostToDevice));
  PDEC_HIP(E->an0.alloc(sizeof(int32_t) * N));
  PDEC_HIP(hipMemcpy(E->an0.p, an0.data(), sizeof(int32_t) * N, hipMemcpyHostToDevice));
  PDEC_HIP(E->a2s.alloc(sizeof(int32_t) * c.A));
  PDEC_HIP(hipMemcpy(E->a2s.p, a2s_h.data(), sizeof(int32_t) * c.A, hipMemcpyHostToDevice));
  if (!c.mono && c.temporal_steps == 1 && c.memory_size == 0) {
    // featurize as one gather (KSSetup.jl:190-229 without temporal stacking): state[a][rr] = dots[sp][(a2s[a] - i) mod S]
    const int ns1 = c.window * c.n_species, w = c.window / 2;
    std::vector<int32_t> fm((size_t)c.A * ns1);
    for (int a = 0; a < c.A; ++a)
      for (int rr = 0; rr < ns1; ++rr) {
        const int sp = rr / c.window, i = (rr - sp * c.window) - w;
        int s = (a2s_h[a] - i) % c.S;
        if (s < 0) s += c.S;
        fm[(size_t)a * ns1 + rr] = sp * c.S + s;
      }
    PDEC_HIP(E->fmap.alloc(sizeof(int32_t) * fm.size()));
    PDEC_HIP(hipMemcpy(E->fmap.p, fm.data(), sizeof(int32_t) * fm.size(), hipMemcpyHostToDevice));
  }
  *h = register_object(std::move(E));
  return PDEC_OK;
}

#define GET_ENV(E, h)                              \
  Env* E = lookup_as<Env>(h, Kind::Env);           \
  if (!E) {                                        \
    set_error("%s: not an env handle", __func__);  \
    return PDEC_E_HANDLE;                          \
  }

int pdec_env_set_terminal_out(pdec_handle h, void* terminal_per_column) {
  GET_ENV(E, h);
  PDEC_REQUIRE(E->cfg.pde_kind != PDEC_PDE_FLUID_RK4 || !terminal_per_column,
               "pdec_env_set_terminal_out: not provided for the fluid environment (expand its done[B] flags)");
  E->term_out = terminal_per_column;
  return PDEC_OK;
}

int pdec_env_set_simd_sharing(pdec_handle h, int on, int* effective) {
  Env* E = lookup_as<Env>(h, Kind::Env);
  if (!E) { set_error("pdec_env_set_simd_sharing: bad handle"); return PDEC_E_HANDLE; }
  E->share_simd = on != 0;
  if (effective) *effective = (E->share_simd && E->cfg.pde_kind == PDEC_PDE_KS_CNAB2 && E->r4_log == 1 && E->cfg.dtype == PDEC_F32) ? 1 : 0;
  return PDEC_OK;
}

int pdec_env_set_reward_partials_out(pdec_handle h, void* partial_sums, int* n_partials) {
  Env* E = lookup_as<Env>(h, Kind::Env);
  if (!E) { set_error("pdec_env_set_reward_partials_out: bad handle"); return PDEC_E_HANDLE; }
  const bool ks = E->cfg.pde_kind == PDEC_PDE_KS_CNAB2, ksfd = E->cfg.pde_kind == PDEC_PDE_KS_RK4_FD;
  PDEC_REQUIRE(((ks || ksfd) && E->cfg.memory_size == 0) || partial_sums == nullptr,
               "pdec_env_set_reward_partials_out: provided by the fused KS steps only (use pdec_reward_mean elsewhere)");
  E->rsum_out = (float*)partial_sums;
  // CNAB2: one workgroup integrates two trajectories; RK4 + FD: one trajectory per workgroup
  if (n_partials) *n_partials = ks ? (E->cfg.B + 1) / 2 : E->cfg.B;
  return PDEC_OK;
}

int pdec_env_part_streams(pdec_handle h, int* n) {
  Env* E = lookup_as<Env>(h, Kind::Env);
  if (!E) { set_error("pdec_env_part_streams: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(n, "pdec_env_part_streams: null");
  *n = E->part_streams();
  return PDEC_OK;
}

int pdec_env_set_part_streams(pdec_handle h, void* const* hip_streams, int n) {
  Env* E = lookup_as<Env>(h, Kind::Env);
  if (!E) { set_error("pdec_env_set_part_streams: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(n >= 0 && (n == 0 || hip_streams), "pdec_env_set_part_streams: bad arguments");
  for (int i = 0; i < n; ++i) PDEC_REQUIRE(hip_streams[i], "pdec_env_set_part_streams: stream %d is null", i);
  return E->set_part_streams((const hipStream_t*)hip_streams, n);
}

int pdec_actuate(pdec_handle h, const void* action, void* p_out) {
  GET_ENV(E, h);
  PDEC_REQUIRE(action && p_out, "pdec_actuate: null");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_actuate(*E, action, p_out);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4) return kseg2d_actuate(*E, action, p_out);
  return E->cfg.dtype == PDEC_F64 ? launch_sense<double>(*E, 0, nullptr, action, nullptr, nullptr, p_out)
                                  : launch_sense<float>(*E, 0, nullptr, action, nullptr, nullptr, p_out);
}

int pdec_featurize(pdec_handle h, const void* y, const void* prev_state, void* state_out) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y && state_out, "pdec_featurize: null");
  PDEC_REQUIRE(prev_state != state_out, "pdec_featurize: state_out must not alias prev_state");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_featurize(*E, y, prev_state, state_out);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4) return kseg2d_featurize(*E, y, prev_state, state_out);
  return E->cfg.dtype == PDEC_F64 ? launch_sense<double>(*E, 1, y, nullptr, nullptr, prev_state, state_out)
                                  : launch_sense<float>(*E, 1, y, nullptr, nullptr, prev_state, state_out);
}

int pdec_featurize_action(pdec_handle h, const void* y, const void* prev_state, const void* action, void* state_out) {
  GET_ENV(E, h);
  if (!action || E->cfg.memory_size == 0) return pdec_featurize(h, y, prev_state, state_out);
  PDEC_REQUIRE(y && state_out, "pdec_featurize_action: null");
  PDEC_REQUIRE(prev_state != state_out, "pdec_featurize_action: state_out must not alias prev_state");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_featurize(*E, y, prev_state, state_out, action);
  return E->cfg.dtype == PDEC_F64 ? launch_sense<double>(*E, 1, y, action, nullptr, prev_state, state_out)
                                  : launch_sense<float>(*E, 1, y, action, nullptr, prev_state, state_out);
}

int pdec_reward(pdec_handle h, const void* y, const void* action, const void* action_prev, void* r_out) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y && action && action_prev && r_out, "pdec_reward: null");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_reward(*E, y, action, action_prev, r_out);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4) return kseg2d_reward(*E, y, action, action_prev, r_out);
  return E->cfg.dtype == PDEC_F64 ? launch_sense<double>(*E, 2, y, action, action_prev, nullptr, r_out)
                                  : launch_sense<float>(*E, 2, y, action, action_prev, nullptr, r_out);
}

int pdec_pde_step(pdec_handle h, const void* y_in, const void* p, void* y_out, int32_t* done) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y_in && p && y_out, "pdec_pde_step: null");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_pde_step(*E, y_in, p, y_out, done);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4) return kseg2d_pde_step(*E, y_in, p, y_out, done);
  return E->cfg.dtype == PDEC_F64
             ? launch_step<double>(*E, false, 1, y_in, p, nullptr, nullptr, nullptr, y_out, nullptr, nullptr, nullptr, done)
             : launch_step<float>(*E, false, 1, y_in, p, nullptr, nullptr, nullptr, y_out, nullptr, nullptr, nullptr, done);
}

int pdec_rhs_eval(pdec_handle h, const void* y, const void* p, void* out) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y && p && out, "pdec_rhs_eval: null");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4) return fluid_rhs_eval(*E, y, p, out);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4) return kseg2d_rhs_eval(*E, y, p, out);
  PDEC_REQUIRE(E->cfg.pde_kind == PDEC_PDE_KSEG_RK4 || E->cfg.pde_kind == PDEC_PDE_KS_RK4_FD,
               "pdec_rhs_eval: only RK4-type PDE kinds expose an RHS");
  return E->cfg.dtype == PDEC_F64
             ? launch_step<double>(*E, false, 2, y, p, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr)
             : launch_step<float>(*E, false, 2, y, p, nullptr, nullptr, nullptr, out, nullptr, nullptr, nullptr, nullptr);
}

int pdec_env_step(pdec_handle h, const void* y_in, const void* action, const void* action_prev,
                  const void* state_prev, void* y_out, void* p_out, void* state_out, void* reward_out,
                  int32_t* done) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y_in && action && action_prev && y_out && state_out && reward_out, "pdec_env_step: null");
  PDEC_REQUIRE(!(E->cfg.temporal_steps > 1 && state_prev == state_out),
               "pdec_env_step: state_out must not alias state_prev when temporal_steps > 1");
  if (E->cfg.pde_kind == PDEC_PDE_FLUID_RK4)
    return fluid_env_step(*E, y_in, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done);
  if (E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4)
    return kseg2d_env_step(*E, y_in, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done);
  if (E->cfg.memory_size > 0)
    return E->cfg.dtype == PDEC_F64
               ? env_step_composed<double>(*E, y_in, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done)
               : env_step_composed<float>(*E, y_in, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done);
  return E->cfg.dtype == PDEC_F64
             ? launch_step<double>(*E, true, 0, y_in, nullptr, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done)
             : launch_step<float>(*E, true, 0, y_in, nullptr, action, action_prev, state_prev, y_out, p_out, state_out, reward_out, done);
}

// ---- host-pointer wrappers: stage through one plan-owned device arena, synchronous
static int env_stage(Env* E, size_t bytes) {
  if (E->stage.bytes < bytes) PDEC_HIP(E->stage.alloc(bytes));
  return PDEC_OK;
}

int pdec_pde_step_host(pdec_handle h, const void* y_in, const void* p, void* y_out, int32_t* done) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y_in && p && y_out, "pdec_pde_step_host: null");
  const pdec_env_cfg& c = E->cfg;
  const size_t ts = dtype_size(c.dtype);
  const size_t ny = (size_t)c.B * env_y_count(c) * ts, np = (size_t)c.B * env_p_count(c) * ts;
  int rc = env_stage(E, 2 * ny + np + c.B * sizeof(int32_t) + 64);
  if (rc) return rc;
  char* base = E->stage.as<char>();
  char *dy = base, *dp = base + ny, *dyo = dp + np, *dd = dyo + ny;
  PDEC_HIP(hipMemcpyAsync(dy, y_in, ny, hipMemcpyHostToDevice, E->stream));
  PDEC_HIP(hipMemcpyAsync(dp, p, np, hipMemcpyHostToDevice, E->stream));
  rc = pdec_pde_step(h, dy, dp, dyo, (int32_t*)dd);
  if (rc) return rc;
  PDEC_HIP(hipMemcpyAsync(y_out, dyo, ny, hipMemcpyDeviceToHost, E->stream));
  if (done) PDEC_HIP(hipMemcpyAsync(done, dd, c.B * sizeof(int32_t), hipMemcpyDeviceToHost, E->stream));
  PDEC_HIP(hipStreamSynchronize(E->stream));
  return PDEC_OK;
}

int pdec_env_step_host(pdec_handle h, const void* y_in, const void* action, const void* action_prev,
                       const void* state_prev, void* y_out, void* p_out, void* state_out, void* reward_out,
                       int32_t* done) {
  GET_ENV(E, h);
  PDEC_REQUIRE(y_in && action && action_prev && y_out && state_out && reward_out, "pdec_env_step_host: null");
  const pdec_env_cfg& c = E->cfg;
  const size_t ts = dtype_size(c.dtype);
  const int ns = env_ns(c);
  const size_t ny = (size_t)c.B * env_y_count(c) * ts, np = (size_t)c.B * env_p_count(c) * ts, na = (size_t)c.B * c.A * env_na(c) * ts;
  const size_t nst = (size_t)c.B * (c.mono ? c.S : c.A * ns) * ts, nr = (size_t)c.B * (c.mono ? 1 : c.A) * ts;
  auto al = [](size_t x) { return (x + 63) / 64 * 64; };
  int rc = env_stage(E, 2 * al(ny) + al(np) + 2 * al(na) + 2 * al(nst) + al(nr) + al(c.B * sizeof(int32_t)));
  if (rc) return rc;
  char* q = E->stage.as<char>();
  char* dy = q; q += al(ny);
  char* dyo = q; q += al(ny);
  char* dp = q; q += al(np);
  char* da = q; q += al(na);
  char* dap = q; q += al(na);
  char* dsp = q; q += al(nst);
  char* dso = q; q += al(nst);
  char* dr = q; q += al(nr);
  char* dd = q;
  PDEC_HIP(hipMemcpyAsync(dy, y_in, ny, hipMemcpyHostToDevice, E->stream));
  PDEC_HIP(hipMemcpyAsync(da, action, na, hipMemcpyHostToDevice, E->stream));
  PDEC_HIP(hipMemcpyAsync(dap, action_prev, na, hipMemcpyHostToDevice, E->stream));
  if (state_prev) PDEC_HIP(hipMemcpyAsync(dsp, state_prev, nst, hipMemcpyHostToDevice, E->stream));
  rc = pdec_env_step(h, dy, da, dap, state_prev ? dsp : nullptr, dyo, dp, dso, dr, (int32_t*)dd);
  if (rc) return rc;
  PDEC_HIP(hipMemcpyAsync(y_out, dyo, ny, hipMemcpyDeviceToHost, E->stream));
  if (p_out) PDEC_HIP(hipMemcpyAsync(p_out, dp, np, hipMemcpyDeviceToHost, E->stream));
  PDEC_HIP(hipMemcpyAsync(state_out, dso, nst, hipMemcpyDeviceToHost, E->stream));
  PDEC_HIP(hipMemcpyAsync(reward_out, dr, nr, hipMemcpyDeviceToHost, E->stream));
  if (done) PDEC_HIP(hipMemcpyAsync(done, dd, c.B * sizeof(int32_t), hipMemcpyDeviceToHost, E->stream));
  PDEC_HIP(hipStreamSynchronize(E->stream));
  return PDEC_OK;
}

}  // extern "C"
