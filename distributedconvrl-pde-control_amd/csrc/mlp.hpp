// mlp.hpp -- the MLP object shared by mlp.hip (generic path) and mlp_mfma.hip (fp32 MFMA path)
#pragma once
#include <algorithm>
#include <cstdint>

#include "common.hpp"

namespace pdec {

struct Mlp : Object {
  int dtype = PDEC_F32, L = 0, max_cols = 0, nparams = 0;
  std::vector<int> dims, acts;
  std::vector<size_t> w_off, b_off;  // offsets into the flat buffers; W_l row-major [out][in], then b_l
  DevBuf params, grads, m, v;        // flat parameter / gradient / ADAM moment buffers
  // ADAM beta powers (Flux keeps Float64[beta1^t, beta2^t] beside the moments) are DEVICE resident and double-buffered:
  // every ADAM kernel reads slot `bp_sel` and ONE of its threads writes the advanced powers into the other slot, so
  // no launch argument depends on the step count and a captured HIP graph of the update replays unchanged (the
  // host only flips bp_sel, which returns to its captured value after an even number of steps).
  DevBuf bpd;                        // double [2][2]
  int bp_sel = 0;
  bool bp_init = false;
  hipEvent_t stop_event = nullptr;   // one-shot (pdec_mlp_set_stop_event): attached to the next reduction / update launch on this net
  hipEvent_t reduce_event = nullptr; // one-shot (pdec_mlp_set_reduce_event): attached to the next reduce-ONLY launch (flat gradient ready)
  const float* rpart_ext = nullptr;  // per-workgroup reward sums of the producer (pdec_ddpg_set_reward_partials), consumed likewise
  int rpart_n = 0;
  const void* rbar_ext = nullptr;    // batch-mean reward reduced elsewhere (pdec_ddpg_set_reward_mean), consumed by the next critic pass
  DevBuf noise_ctr;                  // uint64 [2]: double-buffered exploration-noise counter of pdec_policy_act_rng_dev
  int nc_sel = 0;
  int noise_rows = -1;   // pdec_mlp_set_noise_rows: exploration noise on the first rows of the output only (-1: all)
  std::vector<DevBuf> H;             // activations, feature-major [dims[l]][cols]
  DevBuf dz[2];                      // ping-pong dL/dz buffers [maxdim][cols]
  DevBuf dy;                         // dL/dy of the output layer [dims[L]][cols]
  DevBuf slabs;                      // split-K partial weight gradients
  DevBuf scratch;                    // loss statistics / device scalars
  int kchunk = 512, nsplit_max = 0, dx_index = 0;
  DevBuf fw, fslab;                  // fused-path padded weight image / per-workgroup gradient slabs
  DevBuf stamps;                     // diagnostic phase stamps (PDEC_STAMPS=1)
  bool stamps_armed = false;         // pdec_debug_critic_stamps: the next critic pass on this net records its stamps ...
  double stamps_last[13] = {0};      // ... here: 10 phase means, total cycles, shader clock (GHz), workgroups
  DevBuf noise;                      // internal exploration-noise buffer (pdec_policy_act_rng fallback)
  bool fw_dirty = true;
  DevBuf fw_pub[2];                  // published copies of the image for concurrent acting kernels
  int pub = 0;

  Mlp() : Object(Kind::Mlp) {}
  int init(int dtype, int L, const int32_t* dims, const int32_t* acts, int max_cols);
  template <class T>
  int pack(const void* s1, int n1, int l1, const void* s2, int n2, int l2, int cols);
  template <class T>
  int forward(int cols);
  template <class T>
  int backward(const void* dy, int ldy, int cols, bool want_dw, bool want_dx, double grad_scale);
  template <class T>
  T* dy_buf(int) { return dy.as<T>(); }
};

// Philox4x32-10 counter-based generator (the exploration noise that replaces randn(rng), src/PDEagent.jl:201)
__device__ __forceinline__ void philox4x32(uint32_t c[4], uint32_t k0, uint32_t k1) {
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// kernel-side view of the beta powers: read `cur` (every thread that needs it), one thread of the grid writes `next`
struct BpArgs {
  const double* cur;
  double* next;
};
// slots for the next ADAM kernel on M (uploads {beta1, beta2} on the first step: Flux initialises the powers with the
// betas themselves); the caller flips M->bp_sel after enqueueing the kernel (bp_done)
int bp_begin(Mlp* M, double beta1, double beta2, BpArgs* out);
inline void bp_done(Mlp* M) { flip(M->bp_sel); }
__device__ __forceinline__ void bp_advance(const BpArgs& a, double b1, double b2, int times = 1) {
  double p0 = a.cur[0], p1 = a.cur[1];
  for (int i = 0; i < times; ++i) { p0 *= b1; p1 *= b2; }
  a.next[0] = p0;
  a.next[1] = p1;
}

// mlp_mfma.hip: fused fp32 MFMA DDPG passes (3-layer actor/critic pairs)
struct AdamPolyak {
  double eta, b1, b2, eps, rho;
};
bool fused_supported(const Mlp* A, const Mlp* C);
bool fused_net_supported(const Mlp* M);
// apply != nullptr: the slab reduction also performs ADAM on the network, Polyak into its target and refreshes
// the padded weight images (single-GPU path: no all-reduce between gradient and update)
int fused_critic_grads(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                       const void* sn, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev,
                       const AdamPolyak* apply);
int fused_actor_grads(Mlp* A, Mlp* C, Mlp* At, const void* s, int Bu, double grad_scale, void* loss_dev,
                      const AdamPolyak* apply);
int fused_adam_polyak(Mlp* M, Mlp* Mt, const AdamPolyak& ap);
// ctr_cur != null: the noise offset is *ctr_cur (+ offset) and one thread writes *ctr_next = that + ctr_inc
int fused_policy_act(Mlp* A, const void* state, int cols, double act_noise, double act_limit, int learning,
                     uint64_t seed, uint64_t offset, void* actions_out, const uint64_t* ctr_cur = nullptr,
                     uint64_t* ctr_next = nullptr, uint64_t ctr_inc = 0);

// mlp_mfma2.hip: the same passes for the reference-shaped 2-layer nets [ns, h, 1] / [ns+1, H, 1] (flat parameters, no image)
// mean of r[0..n) in a fixed order, one block; *out = device scalar owned by C (mlp_mfma2.hip)
int launch_rmean(Mlp* C, const float* r, int n, float** out);
bool fused2_supported(const Mlp* A, const Mlp* C);
bool fused2_net_supported(const Mlp* M);
bool fused2_act_supported(const Mlp* A, int cols);
int fused2_policy_act(Mlp* A, const void* state, int cols, double act_noise, double act_limit, int learning, uint64_t seed,
                      uint64_t offset, void* actions_out, const uint64_t* ctr_cur = nullptr, uint64_t* ctr_next = nullptr,
                      uint64_t ctr_inc = 0);
int fused2_adam_polyak(Mlp* M, Mlp* Mt, const AdamPolyak& ap);
int fused2_critic_grads(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                        const void* sn, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev,
                        const AdamPolyak* apply);
int fused2_actor_grads(Mlp* A, Mlp* C, Mlp* At, const void* s, int Bu, double grad_scale, void* loss_dev,
                       const AdamPolyak* apply);

}  // namespace pdec
