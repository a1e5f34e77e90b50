// replay.hip -- the trajectory glue of src/PDEagent.jl:237-340 on the device, the per-trajectory
// same-step reset of a batched environment, the episode initialisers of the 1-D setups (row F4) and HIP-graph capture
// of a whole pipelined control step (row F2).
//
// The reference keeps a CircularArraySARTTrajectory on the host and pushes one (s, a) / (r, terminal) entry per actuator
// and control step (:254-289), pops the dummy entry of an episode end (:237-252), and draws minibatch indices with
// rand(rng, 1:length(t) - A, batch_size), fetching s' at index + A (:317-340).  Here the four traces are device arrays
// (fp32, as RL.jl keeps them, :112-117) and each stage is ONE small kernel; the host only keeps the two entry counters.
#include "common.hpp"
#include "env.hpp"
#include "mlp.hpp"

namespace pdec {

// ------------------------------------------------------------------ pushes
// rows start .. start + n - 1 (mod cap) of two circular traces; the sources are [n][wa] / [n][wb] of type T
template <class T>
__global__ void replay_push2_kernel(float* __restrict__ ta, int wa, const T* __restrict__ sa, float* __restrict__ tb, int wb,
                                    const T* __restrict__ sb, long long cap, long long start, long long n,
                                    const int* __restrict__ halt) {
  if (halt && *halt) return;          // the episode ended at an earlier step of a speculatively issued episode
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long na_ = n * wa, nb_ = n * wb;
  if (i < na_) {
    const long long row = i / wa, c = i - row * wa;
    ta[((start + row) % cap) * wa + c] = (float)sa[i];
  } else if (i < na_ + nb_) {
    const long long j = i - na_, row = j / wb, c = j - row * wb;
    tb[((start + row) % cap) * wb + c] = sb ? (float)sb[j] : 0.f;
  }
}

// reward [n] of type T and the terminal flag of every column: done[column / cols_per_traj] != 0 (or `force`: time-out)
template <class T>
__global__ void replay_push_rt_kernel(float* __restrict__ tr, float* __restrict__ tt, const T* __restrict__ r,
                                      const int32_t* __restrict__ done, int cols_per_traj, int force, long long cap,
                                      long long start, long long n, int* __restrict__ halt) {
  // halt (pdec_set_episode_halt; single-block launches only, checked on the host): nothing is pushed once the episode has
  // ended; the push of the step that ends it -- its terminal transition, src/PDEagent.jl:276-289 -- still happens and raises it
  if (halt && *halt) return;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    const long long slot = (start + i) % cap;
    tr[slot] = (float)r[i];
    tt[slot] = (force || (done && done[i / cols_per_traj] != 0)) ? 1.f : 0.f;
  }
  if (halt) {
    __syncthreads();                  // every thread has read *halt before one of them writes it
    if (threadIdx.x == 0 && done && done[0] != 0) *halt = 1;
  }
}

// ------------------------------------------------------------------ pde_sample + pde_fetch!
struct SampleArgs {
  const float *state, *action, *reward, *terminal;
  int ns, na, Bu, stride;
  long long cap, cap1, base;
  uint32_t hi;
  uint64_t seed, offset;
  float *s, *a, *r, *t, *sn;
  int32_t* slots;     // optional [3][Bu]
};

// draw k of the counter stream (seed, offset): word k % 4 of Philox counter offset + k / 4 -> ind = (word * hi) >> 32
__device__ __forceinline__ uint32_t draw_index(uint64_t seed, uint64_t offset, long long k, uint32_t hi) {
  const uint64_t ctr = offset + (uint64_t)(k >> 2);
  uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
  philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  return (uint32_t)(((uint64_t)c[k & 3] * (uint64_t)hi) >> 32);
}

__global__ void replay_sample_kernel(SampleArgs g) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= g.Bu) return;
  const long long lg = g.base + (long long)draw_index(g.seed, g.offset, k, g.hi);
  const long long is = lg % g.cap1, irt = lg % g.cap, isn = (lg + g.stride) % g.cap1;
  for (int f = 0; f < g.ns; ++f) {
    g.s[(size_t)k * g.ns + f] = g.state[is * g.ns + f];
    g.sn[(size_t)k * g.ns + f] = g.state[isn * g.ns + f];
  }
  for (int f = 0; f < g.na; ++f) g.a[(size_t)k * g.na + f] = g.action[is * g.na + f];
  g.r[k] = g.reward[irt];
  g.t[k] = g.terminal[irt];
  if (g.slots) {
    g.slots[k] = (int32_t)is;
    g.slots[g.Bu + k] = (int32_t)irt;
    g.slots[2 * g.Bu + k] = (int32_t)isn;
  }
}

// ------------------------------------------------------------------ per-trajectory same-step reset
// rows (one per trajectory) of up to four arrays are overwritten by their initial images where done[b] != 0;
// reward rows are only sanitised (a non-finite reward of a blown-up trajectory becomes 0)
struct ResetArgs {
  const int32_t* done;
  int B, tsize;
  char* dst[3];
  const char* src[3];
  long long row_bytes[3];
  void* reward;
  int reward_len;
};
__global__ void autoreset_kernel(ResetArgs g) {
  const int b = blockIdx.x;
  if (g.done[b] == 0) return;
  for (int k = 0; k < 3; ++k) {
    if (!g.dst[k]) continue;
    const long long n4 = g.row_bytes[k] / 4;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(g.src[k] + (size_t)b * g.row_bytes[k]);
    uint32_t* d = reinterpret_cast<uint32_t*>(g.dst[k] + (size_t)b * g.row_bytes[k]);
    for (long long i = threadIdx.x; i < n4; i += blockDim.x) d[i] = s[i];
  }
  if (g.reward)
    for (int i = threadIdx.x; i < g.reward_len; i += blockDim.x) {
      if (g.tsize == 8) {
        double* r = static_cast<double*>(g.reward) + (size_t)b * g.reward_len + i;
        if (!isfinite(*r)) *r = 0.0;
      } else {
        float* r = static_cast<float*>(g.reward) + (size_t)b * g.reward_len + i;
        if (!isfinite(*r)) *r = 0.f;
      }
    }
}

// ------------------------------------------------------------------ episode initialisers of the 1-D setups
// generate_random_init of scripts/KS/setup/KSSetup.jl:288-298 and scripts/Keller-Segel/setup/KellerSegelSetup.jl:373-384:
//   a ~ U(-1, 1)^(n_species * nsin), a /= |a|_2 (all coefficients together),
//   y0[s][j] = base + sum_{i=1..nsin} a[s * nsin + i - 1] * sin(i * x_j / fdiv),  x_j = j dx (j = 1..N),
//   KS (n_species 1, nsin 8, base 0, fdiv 2 pi): then y0 *= 30 / |y0|_2;  Keller-Segel (2 species, nsin = ceil(Lx / 3),
//   base 1, fdiv = 2 pi Lx / 22): no rescaling.
// Uniform k of trajectory b: word k % 4 of the Philox block (seed; counter offset + b * nblk + k / 4), u = (word + 0.5) / 2^32,
// a = 2 u - 1 (Julia draws from its global RNG, so only the distribution can be reproduced; the oracle restates THIS stream).
// One workgroup per trajectory; memory [N][n_species] (species fastest = Julia y[n_species, nx]).
#define RI_MAXC 64
template <class T>
__global__ void random_init_kernel(T* __restrict__ y, int N, int nsp, int nsin, double dx, double fdiv, double base, double l2_target,
                                   uint64_t seed, uint64_t offset) {
  __shared__ double a[RI_MAXC];
  __shared__ double red[256];
  const int b = blockIdx.x, tid = threadIdx.x, nc = nsp * nsin, nblk = (nc + 3) / 4;
  if (tid == 0) {
    double nrm = 0;
    for (int blk = 0; blk < nblk; ++blk) {
      const uint64_t ctr = offset + (uint64_t)b * nblk + blk;
      uint32_t c[4] = {(uint32_t)ctr, (uint32_t)(ctr >> 32), 0u, 0u};
      philox4x32(c, (uint32_t)seed, (uint32_t)(seed >> 32));
      for (int i = 0; i < 4 && 4 * blk + i < nc; ++i) {
        const double v = 2.0 * (((double)c[i] + 0.5) * (1.0 / 4294967296.0)) - 1.0;
        a[4 * blk + i] = v;
        nrm += v * v;
      }
    }
    nrm = sqrt(nrm);
    for (int i = 0; i < nc; ++i) a[i] /= nrm;
  }
  __syncthreads();
  double scale = 1.0;
  if (l2_target > 0) {      // KS: rescale to |y0|_2 = 30 (one species)
    double acc = 0;
    for (int j = tid; j < N; j += blockDim.x) {
      const double x = (j + 1) * dx;
      double v = 0;
      for (int i = 1; i <= nsin; ++i) v += a[i - 1] * sin(i * x / fdiv);
      acc += v * v;
    }
    red[tid] = acc;
    __syncthreads();
    for (int s = blockDim.x / 2; s > 0; s >>= 1) {
      if (tid < s) red[tid] += red[tid + s];
      __syncthreads();
    }
    scale = l2_target / sqrt(red[0]);
  }
  for (int j = tid; j < N; j += blockDim.x) {
    const double x = (j + 1) * dx;
    for (int sp = 0; sp < nsp; ++sp) {
      double v = 0;
      for (int i = 1; i <= nsin; ++i) v += a[sp * nsin + i - 1] * sin(i * x / fdiv);
      y[((size_t)b * N + j) * nsp + sp] = (T)(base + v * scale);
    }
  }
}

// device-scope events: created without the system-scope fence of a default HIP event (hipEventDisableSystemFence),
// so a record / wait pair between two streams of the SAME device does not flush and invalidate the caches towards the
// host -- the cross-stream hand-offs of the control step sit on its critical path
struct Event : Object {
  hipEvent_t ev = nullptr;
  Event() : Object(Kind::Event) {}
  ~Event() override {
    if (ev) (void)hipEventDestroy(ev);
  }
};

__global__ void sync_probe_wait_kernel(const long long* flag, long long* seen, unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  long long ok = 0;
  while (!(ok = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= 1)) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) break;      // the exit the wait always reaches
  }
  *seen = ok;
}
__global__ void sync_probe_set_kernel(long long* flag) { __hip_atomic_store(flag, 1ll, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }

int* launch_sync_timeout_counter() {
  static int* ctr = [] {
    int* p = nullptr;
    if (hipMalloc(&p, sizeof(int)) != hipSuccess) return (int*)nullptr;
    const int zero = 0;
    if (hipMemcpy(p, &zero, sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return (int*)nullptr;
    return p;
  }();
  return ctr;
}

// the HIP event behind an event handle (nullptr: not an event) -- for launches in other translation units that carry one
hipEvent_t event_native(pdec_handle ev) {
  Event* e = lookup_as<Event>(ev, Kind::Event);
  return e ? e->ev : nullptr;
}

extern std::vector<int*>* g_flip_log;

struct Graph : Object {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<int*> odd_flips;      // slot selectors the captured launches flipped an odd number of times
  bool owed = true;                 // the capture advanced the selectors without running anything: the FIRST launch
                                    // performs that work and must not advance them again
  Graph() : Object(Kind::Graph) {}
  ~Graph() override {
    if (exec) (void)hipGraphExecDestroy(exec);
    if (graph) (void)hipGraphDestroy(graph);
  }
};

}  // namespace pdec

using namespace pdec;

extern "C" {

int pdec_replay_push_sa(pdec_handle any_handle, void* state_trace, void* action_trace, int64_t capacity_rows, int ns, int na,
                        int64_t start, const void* s, const void* a, int64_t n, int dtype) {
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_replay_push_sa: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(state_trace && action_trace && s && capacity_rows >= 1 && ns >= 1 && na >= 1 && start >= 0 && n >= 0 &&
               n <= capacity_rows, "pdec_replay_push_sa: bad argument (n %lld, capacity %lld)", (long long)n, (long long)capacity_rows);
  if (n == 0) return PDEC_OK;
  const long long tot = n * (ns + na);
  const dim3 grid((unsigned)((tot + 255) / 256)), block(256);
  ProfScope ps(o, "replay_push_sa");
  if (dtype == PDEC_F64)
    hipLaunchKernelGGL((replay_push2_kernel<double>), grid, block, 0, o->stream, (float*)state_trace, ns, (const double*)s,
                       (float*)action_trace, na, (const double*)a, (long long)capacity_rows, (long long)start, (long long)n, o->halt);
  else
    hipLaunchKernelGGL((replay_push2_kernel<float>), grid, block, 0, o->stream, (float*)state_trace, ns, (const float*)s,
                       (float*)action_trace, na, (const float*)a, (long long)capacity_rows, (long long)start, (long long)n, o->halt);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_replay_push_rt(pdec_handle any_handle, void* reward_trace, void* terminal_trace, int64_t capacity_rows, int64_t start,
                        const void* r, const int32_t* done_flags, int cols_per_traj, int force_terminal, int64_t n, int dtype) {
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_replay_push_rt: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(reward_trace && terminal_trace && r && capacity_rows >= 1 && start >= 0 && n >= 0 && n <= capacity_rows &&
               cols_per_traj >= 1, "pdec_replay_push_rt: bad argument");
  if (n == 0) return PDEC_OK;
  PDEC_REQUIRE(!o->halt || n <= 256, "pdec_replay_push_rt: an episode halt flag (pdec_set_episode_halt) needs a single-block push (n <= 256)");
  const dim3 grid((unsigned)((n + 255) / 256)), block(256);
  ProfScope ps(o, "replay_push_rt");
  if (dtype == PDEC_F64)
    hipLaunchKernelGGL((replay_push_rt_kernel<double>), grid, block, 0, o->stream, (float*)reward_trace, (float*)terminal_trace,
                       (const double*)r, done_flags, cols_per_traj, force_terminal, (long long)capacity_rows, (long long)start,
                       (long long)n, o->halt);
  else
    hipLaunchKernelGGL((replay_push_rt_kernel<float>), grid, block, 0, o->stream, (float*)reward_trace, (float*)terminal_trace,
                       (const float*)r, done_flags, cols_per_traj, force_terminal, (long long)capacity_rows, (long long)start,
                       (long long)n, o->halt);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// Speculatively issued episodes (run.py): `flag` (device int32, 0 at the episode's start) is attached to the handle; see
// Object::halt.  NULL detaches.
int pdec_set_episode_halt(pdec_handle any_handle, int32_t* flag) {
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_set_episode_halt: bad handle"); return PDEC_E_HANDLE; }
  o->halt = flag;
  return PDEC_OK;
}

// Device-side hand-over for the NEXT launch through `handle` that supports it (pdec_step_glue through the actor's handle; the
// fused single-workgroup KS env step through the environment's): see LaunchSync in common.hpp.  A launch path that does not
// support it fails with PDEC_E_INVALID instead of ignoring it (the other side would wait for nothing).
int pdec_set_launch_sync(pdec_handle handle, const int64_t* wait_flag, int64_t wait_value, int64_t* done_flag, int64_t done_value) {
  Object* o = lookup(handle);
  if (!o) { set_error("pdec_set_launch_sync: bad handle"); return PDEC_E_HANDLE; }
  o->sync.wait = reinterpret_cast<const long long*>(wait_flag); o->sync.wait_val = wait_value;
  o->sync.done = reinterpret_cast<long long*>(done_flag); o->sync.done_val = done_value;
  o->sync.timeouts = (wait_flag || done_flag) ? launch_sync_timeout_counter() : nullptr;
  return PDEC_OK;
}

// Do two streams run side by side?  HIP maps streams onto a few hardware queues (four by default); two streams that share one
// execute in issue order, and a launch that waits in the kernel for a launch queued BEHIND it on the same hardware queue waits
// for nothing.  Probe: a one-thread kernel on `stream_a` waits (up to 20 ms) for a flag that a one-thread kernel issued
// afterwards on `stream_b` raises.  *yes = 1 when it saw the flag.  Synchronises both streams; call it once per pair of streams.
int pdec_streams_run_side_by_side(void* stream_a, void* stream_b, int* yes) {
  PDEC_REQUIRE(yes, "pdec_streams_run_side_by_side: null");
  *yes = 0;
  if (stream_a == stream_b) return PDEC_OK;
  DevBuf buf;
  PDEC_HIP(buf.alloc(2 * sizeof(long long)));
  const long long zero[2] = {0, 0};
  PDEC_HIP(hipMemcpy(buf.p, zero, sizeof(zero), hipMemcpyHostToDevice));
  long long* f = buf.as<long long>();
  hipLaunchKernelGGL(sync_probe_wait_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_a, f, f + 1, 2000000ull);
  hipLaunchKernelGGL(sync_probe_set_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream_b, f);
  PDEC_HIP(hipGetLastError());
  PDEC_HIP(hipStreamSynchronize((hipStream_t)stream_a));
  PDEC_HIP(hipStreamSynchronize((hipStream_t)stream_b));
  long long h[2] = {0, 0};
  PDEC_HIP(hipMemcpy(h, buf.p, sizeof(h), hipMemcpyDeviceToHost));
  *yes = h[1] == 1 ? 1 : 0;
  return PDEC_OK;
}

// waits that gave up since the library was loaded (each one a hand-over that never came: a caller's protocol error)
int pdec_launch_sync_timeouts(int* n) {
  PDEC_REQUIRE(n, "pdec_launch_sync_timeouts: null");
  int* c = launch_sync_timeout_counter();
  PDEC_REQUIRE(c, "pdec_launch_sync_timeouts: no device counter");
  PDEC_HIP(hipMemcpy(n, c, sizeof(int), hipMemcpyDeviceToHost));
  return PDEC_OK;
}

int pdec_replay_sample(pdec_handle any_handle, const void* state_trace, const void* action_trace, const void* reward_trace,
                       const void* terminal_trace, int ns, int na, int64_t capacity, int stride, int64_t n_valid, int64_t n_rt,
                       uint64_t seed, uint64_t offset, int Bu, void* s_out, void* a_out, void* r_out, void* t_out,
                       void* sn_out, int32_t* slots_out) {
  Object* o = lookup(any_handle);
  if (!o) { set_error("pdec_replay_sample: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(state_trace && action_trace && reward_trace && terminal_trace && s_out && a_out && r_out && t_out && sn_out,
               "pdec_replay_sample: null argument");
  const int64_t hi = n_valid - stride;         // inds in 1:length(t)-number_actuators (src/PDEagent.jl:318)
  PDEC_REQUIRE(Bu >= 1 && hi >= 1 && hi < ((int64_t)1 << 32) && capacity >= 1 && stride >= 0 && n_valid <= capacity,
               "pdec_replay_sample: nothing to sample (valid %lld, stride %d)", (long long)n_valid, stride);
  SampleArgs g{};
  g.state = (const float*)state_trace; g.action = (const float*)action_trace;
  g.reward = (const float*)reward_trace; g.terminal = (const float*)terminal_trace;
  g.ns = ns; g.na = na; g.Bu = Bu; g.stride = stride;
  g.cap = capacity; g.cap1 = capacity + stride; g.base = n_rt > capacity ? n_rt - capacity : 0;
  g.hi = (uint32_t)hi; g.seed = seed; g.offset = offset;
  g.s = (float*)s_out; g.a = (float*)a_out; g.r = (float*)r_out; g.t = (float*)t_out; g.sn = (float*)sn_out;
  g.slots = slots_out;
  ProfScope ps(o, "replay_sample");
  hipLaunchKernelGGL(replay_sample_kernel, dim3((Bu + 255) / 256), dim3(256), 0, o->stream, g);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_env_autoreset(pdec_handle henv, const int32_t* done, void* y, const void* y0, void* state, const void* state0,
                       void* action, const void* action0, void* reward) {
  Env* E = lookup_as<Env>(henv, Kind::Env);
  if (!E) { set_error("pdec_env_autoreset: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(done && y && y0, "pdec_env_autoreset: null argument");
  const pdec_env_cfg& c = E->cfg;
  const size_t ts = dtype_size(c.dtype);
  ResetArgs g{};
  g.done = done; g.B = c.B; g.tsize = (int)ts;
  g.dst[0] = (char*)y; g.src[0] = (const char*)y0; g.row_bytes[0] = (long long)(env_y_count(c) * ts);
  if (state && state0) {
    g.dst[1] = (char*)state; g.src[1] = (const char*)state0;
    g.row_bytes[1] = (long long)((c.mono ? (size_t)c.S : (size_t)c.A * env_ns(c)) * ts);
  }
  if (action && action0) {
    g.dst[2] = (char*)action; g.src[2] = (const char*)action0; g.row_bytes[2] = (long long)((size_t)c.A * env_na(c) * ts);
  }
  g.reward = reward; g.reward_len = c.mono ? 1 : c.A;
  ProfScope ps(E, "env_autoreset");
  hipLaunchKernelGGL(autoreset_kernel, dim3(c.B), dim3(256), 0, E->stream, g);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_env_random_init(pdec_handle henv, uint64_t seed, uint64_t offset, void* y0_out) {
  Env* E = lookup_as<Env>(henv, Kind::Env);
  if (!E) { set_error("pdec_env_random_init: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(y0_out, "pdec_env_random_init: null");
  const pdec_env_cfg& c = E->cfg;
  int nsp, nsin;
  double fdiv, base, l2;
  const double two_pi = 6.283185307179586;
  if (c.pde_kind == PDEC_PDE_KS_CNAB2 || c.pde_kind == PDEC_PDE_KS_RK4_FD) {
    nsp = 1; nsin = 8; fdiv = two_pi; base = 0.0; l2 = 30.0;                                  // KSSetup.jl:288-298
  } else if (c.pde_kind == PDEC_PDE_KSEG_RK4) {
    nsp = 2; nsin = (int)ceil(c.Lx / 3.0); fdiv = two_pi * (c.Lx / 22.0); base = 1.0; l2 = 0.0;   // KellerSegelSetup.jl:373-384
  } else {
    set_error("pdec_env_random_init: 1-D setups only (the fluid initialiser is pdec_fluid_ic)");
    return PDEC_E_INVALID;
  }
  PDEC_REQUIRE(nsp * nsin <= RI_MAXC, "pdec_env_random_init: %d sine coefficients exceed the kernel's table", nsp * nsin);
  const double dx = c.Lx / c.N;
  ProfScope ps(E, "env_random_init");
  if (c.dtype == PDEC_F64)
    hipLaunchKernelGGL((random_init_kernel<double>), dim3(c.B), dim3(256), 0, E->stream, (double*)y0_out, c.N, nsp, nsin, dx, fdiv,
                       base, l2, seed, offset);
  else
    hipLaunchKernelGGL((random_init_kernel<float>), dim3(c.B), dim3(256), 0, E->stream, (float*)y0_out, c.N, nsp, nsin, dx, fdiv,
                       base, l2, seed, offset);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// ------------------------------------------------------------------ HIP-graph capture
int pdec_capture_begin(pdec_handle origin) {
  Object* o = lookup(origin);
  if (!o) { set_error("pdec_capture_begin: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(o->stream != nullptr, "pdec_capture_begin: the handle must own a non-null stream (pdec_set_stream): the null stream cannot be captured");
  PDEC_REQUIRE(!o->prof, "pdec_capture_begin: switch the per-kernel event timing off first (pdec_prof_enable(h, 0))");
  PDEC_REQUIRE(g_flip_log == nullptr, "pdec_capture_begin: another capture is already open");
  PDEC_HIP(hipStreamBeginCapture(o->stream, hipStreamCaptureModeRelaxed));
  g_flip_log = new std::vector<int*>();
  return PDEC_OK;
}

int pdec_capture_end(pdec_handle origin, pdec_handle* graph_out) {
  Object* o = lookup(origin);
  if (!o) { set_error("pdec_capture_end: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(graph_out, "pdec_capture_end: null");
  auto g = std::make_unique<Graph>();
  if (g_flip_log) {
    std::map<int*, int> cnt;
    for (int* p : *g_flip_log) ++cnt[p];
    for (auto& kv : cnt)
      if (kv.second & 1) g->odd_flips.push_back(kv.first);
    delete g_flip_log;
    g_flip_log = nullptr;
  }
  PDEC_HIP(hipStreamEndCapture(o->stream, &g->graph));
  PDEC_REQUIRE(g->graph != nullptr, "pdec_capture_end: the capture was invalidated (a forked stream did not join the origin?)");
  PDEC_HIP(hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0));
  g->stream = o->stream;
  *graph_out = register_object(std::move(g));
  return PDEC_OK;
}

int pdec_graph_launch(pdec_handle graph, void* hip_stream) {
  Graph* g = lookup_as<Graph>(graph, Kind::Graph);
  if (!g) { set_error("pdec_graph_launch: bad handle"); return PDEC_E_HANDLE; }
  PDEC_HIP(hipGraphLaunch(g->exec, hip_stream ? (hipStream_t)hip_stream : g->stream));
  if (!g->owed)
    for (int* p : g->odd_flips) *p ^= 1;    // what the captured calls did to the host-side slot selectors
  g->owed = false;
  return PDEC_OK;
}

int pdec_event_create(pdec_handle* ev) {
  PDEC_REQUIRE(ev, "pdec_event_create: null");
  auto e = std::make_unique<Event>();
  PDEC_HIP(hipEventCreateWithFlags(&e->ev, hipEventDisableTiming | hipEventDisableSystemFence));
  *ev = register_object(std::move(e));
  return PDEC_OK;
}
int pdec_event_record(pdec_handle ev, void* hip_stream) {
  Event* e = lookup_as<Event>(ev, Kind::Event);
  if (!e) { set_error("pdec_event_record: bad handle"); return PDEC_E_HANDLE; }
  PDEC_HIP(hipEventRecord(e->ev, (hipStream_t)hip_stream));
  return PDEC_OK;
}
// One-shot: the next slab-reduction / ADAM launch issued on `mlp` (the second kernel of pdec_ddpg_update_critic_async /
// _actor_async on the fused 3-layer path) carries `event` as the completion event of its own dispatch packet -- equivalent
// to pdec_event_record(event, stream) right behind that launch, without the separate packet.  Returns PDEC_E_INVALID for
// networks outside that path (the caller records the event itself).
int pdec_mlp_set_stop_event(pdec_handle mlp, pdec_handle event) {
  Mlp* M = lookup_as<Mlp>(mlp, Kind::Mlp);
  if (!M) { set_error("pdec_mlp_set_stop_event: bad handle"); return PDEC_E_HANDLE; }
  if (!event) { M->stop_event = nullptr; return PDEC_OK; }
  Event* e = lookup_as<Event>(event, Kind::Event);
  if (!e) { set_error("pdec_mlp_set_stop_event: bad event handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(fused_net_supported(M), "pdec_mlp_set_stop_event: fused 3-layer networks only");
  M->stop_event = e->ev;
  return PDEC_OK;
}

// The same for the next reduce-ONLY launch (flat gradient complete): what an all-reduce on another stream waits for.
int pdec_mlp_set_reduce_event(pdec_handle mlp, pdec_handle event) {
  Mlp* M = lookup_as<Mlp>(mlp, Kind::Mlp);
  if (!M) { set_error("pdec_mlp_set_reduce_event: bad handle"); return PDEC_E_HANDLE; }
  if (!event) { M->reduce_event = nullptr; return PDEC_OK; }
  Event* e = lookup_as<Event>(event, Kind::Event);
  if (!e) { set_error("pdec_mlp_set_reduce_event: bad event handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(fused_net_supported(M), "pdec_mlp_set_reduce_event: fused 3-layer networks only");
  M->reduce_event = e->ev;
  return PDEC_OK;
}

// An event set by pdec_mlp_set_stop_event that no launch has consumed (the update took another path) is recorded on the
// net's stream now, so a waiter never sees a stale event; no-op otherwise.
int pdec_mlp_flush_stop_event(pdec_handle mlp) {
  Mlp* M = lookup_as<Mlp>(mlp, Kind::Mlp);
  if (!M) { set_error("pdec_mlp_flush_stop_event: bad handle"); return PDEC_E_HANDLE; }
  if (M->stop_event) {
    PDEC_HIP(hipEventRecord(M->stop_event, M->stream));
    M->stop_event = nullptr;
  }
  return PDEC_OK;
}

int pdec_stream_wait_event(void* hip_stream, pdec_handle ev) {
  Event* e = lookup_as<Event>(ev, Kind::Event);
  if (!e) { set_error("pdec_stream_wait_event: bad handle"); return PDEC_E_HANDLE; }
  PDEC_HIP(hipStreamWaitEvent((hipStream_t)hip_stream, e->ev, 0));
  return PDEC_OK;
}

int pdec_graph_num_nodes(pdec_handle graph, int* n) {
  Graph* g = lookup_as<Graph>(graph, Kind::Graph);
  if (!g) { set_error("pdec_graph_num_nodes: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(n, "pdec_graph_num_nodes: null");
  size_t k = 0;
  PDEC_HIP(hipGraphGetNodes(g->graph, nullptr, &k));
  *n = (int)k;
  return PDEC_OK;
}

}  // extern "C"
