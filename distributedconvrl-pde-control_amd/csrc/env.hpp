// env.hpp -- the environment object shared by env.hip (1-D KS / Keller-Segel kernels) and
// fluid.hip (2-D pseudo-spectral vorticity solver).
#pragma once
#include "common.hpp"
#include "fft_lds.hpp"

namespace pdec {

template <class T>
struct EnvDev {
  int B, N, S, A, ns, window, temporal, mono, K, check_max, n_species, rk2, prio;
  int mem, na;           // action memory rows (cfg.memory_size) and rows per action column, na = 1 + mem: action [B][A][na]
  T sensor_scale, agent_power, r_in_scale, r_offset, r_power, r_denom, a_pun, da_pun, max_value;
  T dx, hstep, dist_mu;  // cell size, RK4 sub-step, KS disturbance amplitude (RK4-FD variant)
  // sensor / actuator kernels as circular BAND tables (exact: every non-zero entry of the dense
  // [S][N] / [A][N] matrices is kept; a kernel whose support is the whole domain gives Wd = N)
  const T* Gs;           // [Wd][S]   Gs[j][s] = g_s[(sn0[s] + j) mod N]     (coalesced over s)
  const int* sn0;        // [S]       first cell of sensor s's window
  const T* GaC;          // [Cnt][N]  GaC[i][n] = ga_{(an0[n]+i) mod A}[n]   (coalesced over n)
  const int* an0;        // [N]       first actuator reaching cell n
  int Wd, Cnt;
  float* rsum_out;       // optional [ceil(B/2)]: per-workgroup sum of the rewards it wrote (pdec_env_set_reward_partials_out)
  T* term_out;           // optional [B][cols per trajectory]: 1.0 where the trajectory blew up (pdec_env_set_terminal_out)
  const T* gsum;         // [S]     sum of each sensor kernel (reward offset term)
  const int* a2s;        // [A]
  const int* fmap;       // [A * ns] or null: state[idx] = dots[fmap[idx]] * sensor_scale (featurize without index arithmetic)
  // KS CNAB2 per-mode constants
  const T *c1, *c2, *c3, *c4, *g;
  const C2<T>* dhat;     // h * fft(mu cos(...))
  const C2<T>* tw;       // exp(-2 pi i k/N)
  FftPlan fft;
  LaunchSync sync;       // pdec_set_launch_sync (SYNC instantiations of the fused KS step only)
};

struct Env : Object {
  pdec_env_cfg cfg;
  DevBuf Gs, sn0, GaC, an0, gsum, a2s, fmap, c1, c2, c3, c4, g, dhat, tw;
  int Wd = 0, Cnt = 0;
  DevBuf stage;  // staging for the _host wrappers
  DevBuf roll;   // ping-pong buffers of pdec_rollout
  DevBuf mem_scratch;   // forcing field + flags of the composed env step (cfg.memory_size > 0)
  void* term_out = nullptr;
  float* rsum_out = nullptr;
  bool share_simd = false;   // pdec_env_set_simd_sharing: launch the 64-VGPR form of the fused KS step
  FftPlan fft;
  int nthreads = 64;
  int r4_log = 0;        // 4 / 5: N = 256 / 1024 use the register-resident radix-4 FFT engine
  size_t lds_bytes = 0;
  Env() : Object(Kind::Env) {}
  // environments that run parts of their batch on streams of their own (fluid.hip, kseg2d.hip): how many such streams the
  // step uses besides the environment's, and the caller's streams to use instead of the library's (pdec_env_set_part_streams)
  virtual int part_streams() const { return 0; }
  virtual int set_part_streams(const hipStream_t*, int) { return PDEC_OK; }
};

// env.hip: T acting + env steps of the KS environment in one persistent launch (see ks_rollout_kernel); returns
// PDEC_E_INVALID without touching anything when the configuration is not covered (the caller then loops per step)
struct Mlp;
bool ks_rollout_supported(const Env& E, const Mlp& A);
bool kseg_rollout_supported(const Env& E, const Mlp& A);
int kseg_rollout_persistent(Env& E, const Mlp& A, int T, void* y, void* state, void* action, double act_noise, double act_limit,
                            int learning, uint64_t seed, uint64_t offset, void* reward_sum, void* log_y, void* log_p,
                            void* log_action, void* log_reward, int32_t* done_any, int32_t* done_step);
int ks_rollout_persistent(Env& E, const Mlp& A, int T, void* y, void* state, void* action, double act_noise, double act_limit,
                          int learning, uint64_t seed, uint64_t offset, void* reward_sum, void* log_y, void* log_p,
                          void* log_action, void* log_reward, int32_t* done_any, int32_t* done_step);

// fluid.hip: 2-D pseudo-spectral vorticity environment (src/fluid_rk4.jl + scripts/Fluid/setup/FluidSetup.jl)
struct FluidEnv;
int fluid_env_step(Env& E, const void* y_in, const void* action, const void* action_prev, const void* state_prev,
                   void* y_out, void* p_out, void* state_out, void* reward_out, int32_t* done);
int fluid_pde_step(Env& E, const void* y_in, const void* p, void* y_out, int32_t* done);
int fluid_rhs_eval(Env& E, const void* y, const void* p, void* out);
int fluid_actuate(Env& E, const void* action, void* p_out);
int fluid_featurize(Env& E, const void* y, const void* state_prev, void* state_out, const void* action = nullptr);
int fluid_reward(Env& E, const void* y, const void* action, const void* action_prev, void* r_out);

// kseg2d.hip: Keller-Segel on a 2-D grid (BASELINE.json configs[3]; the reference's 1-D rules along both axes)
int kseg2d_env_step(Env& E, const void* y_in, const void* action, const void* action_prev, const void* state_prev,
                    void* y_out, void* p_out, void* state_out, void* reward_out, int32_t* done);
int kseg2d_pde_step(Env& E, const void* y_in, const void* p, void* y_out, int32_t* done);
int kseg2d_rhs_eval(Env& E, const void* y, const void* p, void* out);
int kseg2d_actuate(Env& E, const void* action, void* p_out);
int kseg2d_featurize(Env& E, const void* y, const void* state_prev, void* state_out);
int kseg2d_reward(Env& E, const void* y, const void* action, const void* action_prev, void* r_out);

// element counts per trajectory of the arrays that cross the C ABI
inline size_t env_y_count(const pdec_env_cfg& c) {
  if (c.pde_kind == PDEC_PDE_FLUID_RK4) return (size_t)c.N * c.N * 2;
  if (c.pde_kind == PDEC_PDE_KSEG2D_RK4) return (size_t)c.N * c.Ny * 2;
  return (size_t)c.n_species * c.N;
}
inline size_t env_p_count(const pdec_env_cfg& c) {
  if (c.pde_kind == PDEC_PDE_FLUID_RK4) return (size_t)c.N * c.N * 2;
  if (c.pde_kind == PDEC_PDE_KSEG2D_RK4) return (size_t)c.N * c.Ny;
  return (size_t)c.N;
}
inline int env_ns(const pdec_env_cfg& c) {
  if (c.mono) return c.S;
  if (c.pde_kind == PDEC_PDE_FLUID_RK4) return c.window * c.window * c.temporal_steps + c.memory_size;
  if (c.pde_kind == PDEC_PDE_KSEG2D_RK4) return 2 * c.window * c.window * c.temporal_steps;
  return c.window * c.n_species * c.temporal_steps + c.memory_size;
}
inline int env_na(const pdec_env_cfg& c) { return 1 + c.memory_size; }

}  // namespace pdec
