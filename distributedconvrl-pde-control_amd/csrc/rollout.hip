// rollout.hip -- T control steps without the host in the loop.
//
// The reference's run loop (ReinforcementLearning.jl `run`, driven through src/PDEagent.jl:175-209 and
// src/PDEenv.jl:195-241) returns to the host after every control step; for evaluation / data-collection rollouts
// (`testrun`, PDEhook's bestDF logging, src/PDEhook.jl:51-63) nothing on the host depends on the step's result, so
// pdec_rollout enqueues all T steps -- actor forward + exploration noise + clamp, then the fused environment step --
// on the environment's stream in one call, ping-ponging the state buffers, accumulating the rewards and appending
// the per-step log rows on the device.  The caller synchronises once at the end (pdec_sync).
#include "env.hpp"
#include "mlp.hpp"

namespace pdec {

template <class T>
__global__ void rollout_accum_kernel(size_t n, const T* __restrict__ r, T* __restrict__ sum) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) sum[i] += r[i];
}
__global__ void rollout_or_kernel(int n, const int32_t* __restrict__ d, int32_t* __restrict__ acc, int32_t* __restrict__ first,
                                  int step) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && d[i]) {
    acc[i] |= d[i];
    if (first && first[i] < 0) first[i] = step;
  }
}

}  // namespace pdec

using namespace pdec;

extern "C" int pdec_rollout(pdec_handle henv, pdec_handle hactor, int T, void* y, void* state, void* action,
                            double act_noise, double act_limit, int learning, uint64_t seed, uint64_t offset,
                            void* reward_sum, void* log_y, void* log_p, void* log_action, void* log_reward,
                            int32_t* done_any, int32_t* done_step) {
  Env* E = lookup_as<Env>(henv, Kind::Env);
  Mlp* A = lookup_as<Mlp>(hactor, Kind::Mlp);
  if (!E || !A) { set_error("pdec_rollout: bad handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(T >= 1 && y && state && action, "pdec_rollout: null/empty argument");
  const pdec_env_cfg& c = E->cfg;
  PDEC_REQUIRE(A->dtype == c.dtype, "pdec_rollout: the actor and the environment must share one dtype");
  PDEC_REQUIRE(A->stream == E->stream, "pdec_rollout: the actor and the environment must share one stream");
  const size_t ts = dtype_size(c.dtype);
  const int ns = env_ns(c), cols = c.B * (c.mono ? 1 : c.A), na = A->dims[A->L];
  PDEC_REQUIRE(A->dims[0] == (c.mono ? c.S : ns) && cols * na == c.B * c.A * env_na(c),
               "pdec_rollout: actor shape %d -> %d does not match the state/action matrices", A->dims[0], na);
  if (ks_rollout_supported(*E, *A)) {
    // KS: the whole loop in ONE persistent launch -- the trajectories stay in registers / LDS between steps and the actor is
    // evaluated in the kernel (csrc/env.hip: ks_rollout_kernel); reward_sum accumulates, the logs are written per step
    return ks_rollout_persistent(*E, *A, T, y, state, action, act_noise, act_limit, learning, seed, offset, reward_sum, log_y,
                                 log_p, log_action, log_reward, done_any, done_step);
  }
  if (kseg_rollout_supported(*E, *A))     // 1-D Keller-Segel: likewise one launch (csrc/env.hip: kseg_rollout_kernel)
    return kseg_rollout_persistent(*E, *A, T, y, state, action, act_noise, act_limit, learning, seed, offset, reward_sum, log_y,
                                   log_p, log_action, log_reward, done_any, done_step);
  const size_t ny = (size_t)c.B * env_y_count(c) * ts, np = (size_t)c.B * env_p_count(c) * ts, nact = (size_t)c.B * c.A * env_na(c) * ts;
  const size_t nst = (size_t)c.B * (c.mono ? c.S : (size_t)c.A * ns) * ts, nr = (size_t)c.B * (c.mono ? 1 : c.A) * ts;
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  // scratch: second y / state / action buffers, p, reward, done
  const size_t need = al(ny) + al(nst) + al(nact) + al(np) + al(nr) + al(sizeof(int32_t) * c.B);
  if (E->roll.bytes < need) PDEC_HIP(E->roll.alloc(need));
  char* q = E->roll.as<char>();
  char* yb[2] = {(char*)y, q}; q += al(ny);
  char* sb[2] = {(char*)state, q}; q += al(nst);
  char* ab[2] = {(char*)action, q}; q += al(nact);      // ab[cur] = previous action, ab[cur ^ 1] receives the new one
  char* pb = q; q += al(np);
  char* rb = q; q += al(nr);
  int32_t* db = (int32_t*)q;
  if (done_any) PDEC_HIP(hipMemsetAsync(done_any, 0, sizeof(int32_t) * c.B, E->stream));
  if (done_step) PDEC_HIP(hipMemsetAsync(done_step, 0xFF, sizeof(int32_t) * c.B, E->stream));
  int cur = 0;
  for (int t = 0; t < T; ++t) {
    int rc = pdec_policy_act_rng(hactor, sb[cur], cols, act_noise, act_limit, learning, seed,
                                 offset + (uint64_t)t * (((uint64_t)cols * na + 3) / 4), ab[cur ^ 1]);
    if (rc) return rc;
    rc = pdec_env_step(henv, yb[cur], ab[cur ^ 1], ab[cur], sb[cur], yb[cur ^ 1], pb, sb[cur ^ 1], rb, db);
    if (rc) return rc;
    cur ^= 1;
    const size_t nrew = nr / ts;
    if (reward_sum) {
      if (c.dtype == PDEC_F64)
        hipLaunchKernelGGL(rollout_accum_kernel<double>, dim3((unsigned)((nrew + 255) / 256)), dim3(256), 0, E->stream, nrew,
                           (const double*)rb, (double*)reward_sum);
      else
        hipLaunchKernelGGL(rollout_accum_kernel<float>, dim3((unsigned)((nrew + 255) / 256)), dim3(256), 0, E->stream, nrew,
                           (const float*)rb, (float*)reward_sum);
    }
    if (done_any)
      hipLaunchKernelGGL(rollout_or_kernel, dim3((c.B + 255) / 256), dim3(256), 0, E->stream, c.B, db, done_any, done_step, t);
    // PDEhook's per-step rows (src/PDEhook.jl:54-62), appended on the device
    if (log_y) PDEC_HIP(hipMemcpyAsync((char*)log_y + (size_t)t * ny, yb[cur], ny, hipMemcpyDeviceToDevice, E->stream));
    if (log_p) PDEC_HIP(hipMemcpyAsync((char*)log_p + (size_t)t * np, pb, np, hipMemcpyDeviceToDevice, E->stream));
    if (log_action) PDEC_HIP(hipMemcpyAsync((char*)log_action + (size_t)t * nact, ab[cur], nact, hipMemcpyDeviceToDevice, E->stream));
    if (log_reward) PDEC_HIP(hipMemcpyAsync((char*)log_reward + (size_t)t * nr, rb, nr, hipMemcpyDeviceToDevice, E->stream));
  }
  PDEC_HIP(hipGetLastError());
  if (cur == 1) {      // results back into the caller's buffers
    PDEC_HIP(hipMemcpyAsync(y, yb[1], ny, hipMemcpyDeviceToDevice, E->stream));
    PDEC_HIP(hipMemcpyAsync(state, sb[1], nst, hipMemcpyDeviceToDevice, E->stream));
    PDEC_HIP(hipMemcpyAsync(action, ab[1], nact, hipMemcpyDeviceToDevice, E->stream));
  }
  return PDEC_OK;
}
