// mlp.hpp -- the MLP object shared by mlp.hip (generic path) and mlp_mfma.hip (fp32 MFMA path)
#pragma once
#include <algorithm>

#include "common.hpp"

namespace pdec {

struct Mlp : Object {
  int dtype = PDEC_F32, L = 0, max_cols = 0, nparams = 0;
  std::vector<int> dims, acts;
  std::vector<size_t> w_off, b_off;  // offsets into the flat buffers; W_l row-major [out][in], then b_l
  DevBuf params, grads, m, v;        // flat parameter / gradient / ADAM moment buffers
  double bp[2];                      // ADAM beta powers (Flux keeps Float64[beta1^t, beta2^t])
  std::vector<DevBuf> H;             // activations, feature-major [dims[l]][cols]
  DevBuf dz[2];                      // ping-pong dL/dz buffers [maxdim][cols]
  DevBuf dy;                         // dL/dy of the output layer [dims[L]][cols]
  DevBuf slabs;                      // split-K partial weight gradients
  DevBuf scratch;                    // loss statistics / device scalars
  int kchunk = 512, nsplit_max = 0, dx_index = 0;
  DevBuf fw, fslab;                  // fused-path padded weight image / per-workgroup gradient slabs
  bool fw_dirty = true;

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

// mlp_mfma.hip: fused fp32 MFMA DDPG passes (3-layer actor/critic pairs)
bool fused_supported(const Mlp* A, const Mlp* C);
int fused_critic_grads(Mlp* A, Mlp* C, Mlp* At, Mlp* Ct, const void* s, const void* a, const void* r, const void* t,
                       const void* sn, int Bu, double gamma, int quirk, double grad_scale, void* loss_dev);
int fused_actor_grads(Mlp* A, Mlp* C, const void* s, int Bu, double grad_scale, void* loss_dev);

}  // namespace pdec
