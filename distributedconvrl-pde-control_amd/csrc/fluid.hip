// fluid.hip -- batched 2-D pseudo-spectral vorticity environment for gfx950 (fp64).
//
// Restates (from scratch, batched over independent trajectories):
//   rk4(f, p, dt)            src/fluid_rk4.jl:122-132      classical RK4, forcing frozen over the step
//   rhs(omghat, p)           src/fluid_rk4.jl:134-143      -nu k^2 w + advection(w) + p
//   advection(omghat)        src/fluid_rk4.jl:145-190      pseudo-spectral Jacobian, 3/2-rule de-aliasing
//   pad / chop               src/fluid_rk4.jl:192-229      index maps only (never materialised here)
//   do_step                  scripts/Fluid/setup/FluidSetup.jl:163-172   K = floor(16 nx dt) RK4 sub-steps
//   featurize / reward_function / prepare_action          FluidSetup.jl:188-261
//
// Design (HBM-bound path, so the point is to move fewer bytes than the reference's 5 padded
// 2-D FFTs per RHS):
//  * real(ifft(A)) = ifft(Herm(A)), Herm(A)[k] = (A[k] + conj(A[-k]))/2, so the four inverse
//    transforms of the reference (u, v, dw/dx, dw/dy; real part taken after each, :169-172)
//    become TWO complex transforms of Z1 = Herm(u^) + i Herm(v^), Z2 = Herm(wx^) + i Herm(wy^).
//    This is exact for any spectrum (Hermitian or not), including the +Nyquist modes the
//    reference keeps (FluidSetup.jl:106-107).
//  * the padded spectra are zero outside n (+1 mirror line) of the p = 3n/2 lines: the first
//    inverse pass only transforms those lines and the last forward pass only produces the n kept
//    ones (pruned 2-D FFT; pad()/chop() are index maps inside the kernels).
//  * the three slow-axis transforms (2 inverse, product, 1 forward) of a tile of columns happen
//    inside ONE kernel out of LDS: the physical-space fields never exist in HBM.  The forward
//    transform of the real product packs two columns as Re/Im of one complex line.
//  * RK4 stage updates and the linear term are fused into the last pass.
//  Per RHS and trajectory the HBM traffic is ~ (3 + 3) n p + 7 n^2 complex values (70 MB at n = 512)
//  instead of 5 * 2 * 2 * p^2 (189 MB) for un-fused library FFTs.
//
// Layout: Julia ComplexF64[ny, nx] column-major = memory [nx][ny] (y fastest); square box (nx = ny = n,
// Lx = Ly) as in every shipped script.  "fast axis" = y, "slow axis" = x.
#include "env.hpp"
#include "wave_fft.hpp"

namespace pdec {

#ifndef FL_NTH
#define FL_NTH 1024
#endif
#define FL_MAXE (3072 / FL_NTH)   // elements per thread of one tile: TL * p <= 3072

__host__ __device__ inline int fl_unpad(int ip, int n, int p) {   // chop(): padded index -> kept index or -1
  if (ip <= n / 2) return ip;
  if (ip >= p - n / 2 + 1) return ip - (p - n);
  return -1;
}
__host__ __device__ inline int fl_pad(int i, int n, int p) { return i <= n / 2 ? i : i + (p - n); }   // pad()
// the nl slow lines carried between the two inverse passes (n kept lines + the mirror of the Nyquist line)
__host__ __device__ inline int fl_line_jp(int s, int n, int p, int nl) { return s <= n / 2 ? s : s + p - nl; }
__host__ __device__ inline int fl_line_of(int jp, int n, int p, int nl) {
  if (jp <= n / 2) return jp;
  if (jp >= p - (nl - n / 2 - 1)) return jp - (p - nl);
  return -1;
}

template <class T>
struct FluidDev {
  int B, n, p, nl, TL, TLn, LS, LSn;
  int wtile;           // W between the two inverse passes is TILE-major, [b][f][p / 8][nl][8] (the persistent x-pass reads a
                       // tile as one contiguous block), instead of line-major [b][f][nl][p]
  T nu, inv2, scale_out, invn2;
  const T* k;          // [n] wavenumbers, [0..n/2, -n/2+1..-1] * 2 pi / L   (FluidSetup.jl:106-107)
  const C2<T>* twp;    // exp(-2 pi i m / p)
  const C2<T>* twn;    // exp(-2 pi i m / n)
  FftPlan plp, pln;
};

// ------------------------------------------------------------------ tile FFT (nlines lines of one length in LDS)
template <int R, int SGN, class T>
__device__ __forceinline__ void tile_stage(const C2<T>* __restrict__ x, C2<T>* __restrict__ y,
                                           const C2<T>* __restrict__ tw, int N, int n, int s, int u0, int tpl) {
  const int m = n / R, nb = N / R;
  const bool pow2 = (s & (s - 1)) == 0;
  const int sh = 31 - __clz(s);
  for (int u = u0; u < nb; u += tpl) {
    int pp, q;
    if (pow2) { pp = u >> sh; q = u & (s - 1); }
    else { pp = u / s; q = u - pp * s; }
    C2<T> a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = x[q + s * (pp + m * j)];
    dft_small<R, SGN, T>(a);
    const int base = q + s * R * pp, ps = pp * s;
    y[base] = a[0];
    // one twiddle read per butterfly; w^2, w^3 by multiplication (the LDS, not the fp64 VALU, is the busy unit here)
    C2<T> w1 = tw[ps];
    if (SGN > 0) w1.y = -w1.y;
    C2<T> w = w1;
#pragma unroll
    for (int k = 1; k < R; ++k) {
      y[base + s * k] = cmul(a[k], w);
      if (k + 1 < R) w = cmul(w, w1);
    }
  }
}

// Transforms `nlines` (power of two <= FL_NTH) lines X[l*LS .. l*LS+N) -> returned buffer (X or Y).
// Issues a barrier before the first stage and after every stage.
template <int SGN, class T>
__device__ __forceinline__ C2<T>* tile_fft(C2<T>* X, C2<T>* Y, const C2<T>* tw, const FftPlan& pl, int LS,
                                           int nlines, int tid) {
  const int tpl = FL_NTH / nlines;
  const int line = tid / tpl, u0 = tid - line * tpl;
  int n = pl.N, s = 1;
  __syncthreads();
  for (int st = 0; st < pl.nstages; ++st) {
    const int r = pl.radix[st];
    const C2<T>* x = X + line * LS;
    C2<T>* y = Y + line * LS;
    if (r == 4) tile_stage<4, SGN, T>(x, y, tw, pl.N, n, s, u0, tpl);
    else if (r == 2) tile_stage<2, SGN, T>(x, y, tw, pl.N, n, s, u0, tpl);
    else if (r == 3) tile_stage<3, SGN, T>(x, y, tw, pl.N, n, s, u0, tpl);
    else tile_stage<5, SGN, T>(x, y, tw, pl.N, n, s, u0, tpl);
    __syncthreads();
    C2<T>* t = X; X = Y; Y = t;
    n /= r;
    s *= r;
  }
  return X;
}

// spectral velocities / vorticity gradients of one mode (src/fluid_rk4.jl:152-161)
template <class T>
__device__ __forceinline__ void fl_spec(C2<T> o, T kx, T ky, bool dc, C2<T>& u, C2<T>& v, C2<T>& wx, C2<T>& wy) {
  const T k2 = kx * kx + ky * ky;
  C2<T> psi = mk<T>(0, 0);
  if (!dc) psi = mk<T>(o.x / k2, o.y / k2);       // psihat = omghat ./ kx2ky2; psihat[1,1] = 0
  u = mk<T>(-ky * psi.y, ky * psi.x);             // uhat =  i ky psihat
  v = mk<T>(kx * psi.y, -kx * psi.x);             // vhat = -i kx psihat
  wx = mk<T>(-kx * o.y, kx * o.x);                // domgdx = i kx omghat
  wy = mk<T>(-ky * o.y, ky * o.x);                // domgdy = i ky omghat
}

// ------------------------------------------------------------------ K1: spectra + pad + Herm + inverse pass along y
// grid (ceil(nl/TL), B).  W[b][f][s][ip], f = 0: Z1, 1: Z2 (unnormalised inverse along y).
template <class T>
__global__ __launch_bounds__(FL_NTH) void fluid_k1_kernel(FluidDev<T> d, const C2<T>* __restrict__ omg,
                                                          C2<T>* __restrict__ W) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  C2<T>* X = reinterpret_cast<C2<T>*>(smem_raw);
  C2<T>* Y = X + d.TL * d.LS;
  C2<T>* tw = Y + d.TL * d.LS;
  const int tid = threadIdx.x, n = d.n, p = d.p, LS = d.LS;
  for (int k = tid; k < p; k += FL_NTH) tw[k] = d.twp[k];
  const int tpl = FL_NTH / d.TL, line = tid / tpl, u0 = tid - line * tpl;
  const int s = blockIdx.x * d.TL + line, b = blockIdx.y;
  const bool live = s < d.nl;
  int j = -1, jm = -1;
  if (live) {
    const int jp = fl_line_jp(s, n, p, d.nl);
    j = fl_unpad(jp, n, p);
    jm = fl_unpad((p - jp) % p, n, p);
  }
  const C2<T>* oj = omg + ((size_t)b * n + (j >= 0 ? j : 0)) * n;
  const C2<T>* om = omg + ((size_t)b * n + (jm >= 0 ? jm : 0)) * n;
  const T kj = j >= 0 ? d.k[j] : (T)0, kjm = jm >= 0 ? d.k[jm] : (T)0;
  C2<T> z2[FL_MAXE];
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int ip = u0 + tpl * e;
    z2[e] = mk<T>(0, 0);
    if (ip < p) {
      const int i = fl_unpad(ip, n, p), im = fl_unpad((p - ip) % p, n, p);
      C2<T> au = mk<T>(0, 0), av = au, ax = au, ay = au, mu = au, mv = au, mx = au, my = au;
      if (j >= 0 && i >= 0) fl_spec<T>(oj[i], kj, d.k[i], i == 0 && j == 0, au, av, ax, ay);
      if (jm >= 0 && im >= 0) fl_spec<T>(om[im], kjm, d.k[im], im == 0 && jm == 0, mu, mv, mx, my);
      // Herm(A) = (A(k) + conj(A(-k))) / 2
      const C2<T> hu = mk<T>((T)0.5 * (au.x + mu.x), (T)0.5 * (au.y - mu.y));
      const C2<T> hv = mk<T>((T)0.5 * (av.x + mv.x), (T)0.5 * (av.y - mv.y));
      const C2<T> hx = mk<T>((T)0.5 * (ax.x + mx.x), (T)0.5 * (ax.y - mx.y));
      const C2<T> hy = mk<T>((T)0.5 * (ay.x + my.x), (T)0.5 * (ay.y - my.y));
      X[line * LS + ip] = mk<T>(hu.x - hv.y, hu.y + hv.x);   // Z1 = Herm(u) + i Herm(v)
      z2[e] = mk<T>(hx.x - hy.y, hx.y + hy.x);               // Z2 = Herm(wx) + i Herm(wy)
    }
  }
  C2<T>* R = tile_fft<+1, T>(X, Y, tw, d.plp, LS, d.TL, tid);
  if (live) {
    C2<T>* w0 = W + (((size_t)b * 2 + 0) * d.nl + s) * p;
#pragma unroll
    for (int e = 0; e < FL_MAXE; ++e) {
      const int ip = u0 + tpl * e;
      if (ip < p) w0[ip] = R[line * LS + ip];
    }
  }
  __syncthreads();
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int ip = u0 + tpl * e;
    if (ip < p) X[line * LS + ip] = z2[e];
  }
  R = tile_fft<+1, T>(X, Y, tw, d.plp, LS, d.TL, tid);
  if (live) {
    C2<T>* w1 = W + (((size_t)b * 2 + 1) * d.nl + s) * p;
#pragma unroll
    for (int e = 0; e < FL_MAXE; ++e) {
      const int ip = u0 + tpl * e;
      if (ip < p) w1[ip] = R[line * LS + ip];
    }
  }
}

// ------------------------------------------------------------------ K2: inverse pass along x, product, forward pass along x
// grid (ceil(p/TL), B).  W2[b][j][ip] = chop_x( FFT_x( -(u wx + v wy) ) ), two columns per complex line.
template <class T>
__global__ __launch_bounds__(FL_NTH) void fluid_k2_kernel(FluidDev<T> d, const C2<T>* __restrict__ W,
                                                          C2<T>* __restrict__ W2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  C2<T>* X = reinterpret_cast<C2<T>*>(smem_raw);
  C2<T>* Y = X + d.TL * d.LS;
  C2<T>* tw = Y + d.TL * d.LS;
  const int tid = threadIdx.x, n = d.n, p = d.p, LS = d.LS, TL = d.TL;
  for (int k = tid; k < p; k += FL_NTH) tw[k] = d.twp[k];
  const int tlsh = 31 - __clz(TL);
  const int tpl = FL_NTH / TL, line = tid / tpl, u0 = tid - line * tpl;
  const int ip0 = blockIdx.x * TL, b = blockIdx.y;
  C2<T> r0[FL_MAXE];
  T wre[FL_MAXE];
#pragma unroll
  for (int f = 0; f < 2; ++f) {
    if (f) __syncthreads();
    const C2<T>* Wf = W + (((size_t)b * 2 + f) * d.nl) * p;
    for (int idx = tid; idx < TL * p; idx += FL_NTH) {       // transposed load, zero fill = pad() along x
      const int t = idx & (TL - 1), jp = idx >> tlsh;
      const int s = fl_line_of(jp, n, p, d.nl), ip = ip0 + t;
      C2<T> v = mk<T>(0, 0);
      if (s >= 0 && ip < p) v = Wf[(size_t)s * p + ip];
      X[t * LS + jp] = v;
    }
    C2<T>* R = tile_fft<+1, T>(X, Y, tw, d.plp, LS, TL, tid);
#pragma unroll
    for (int e = 0; e < FL_MAXE; ++e) {
      const int jp = u0 + tpl * e;
      if (jp < p) {
        const C2<T> v = R[line * LS + jp];
        if (f == 0) r0[e] = v;
        else wre[e] = -(r0[e].x * v.x + r0[e].y * v.y) * d.inv2;   // -(u wx + v wy), both ifft scalings
      }
    }
  }
  __syncthreads();
  const int half = TL / 2;
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int jp = u0 + tpl * e;
    if (jp < p) {
      T* dst = reinterpret_cast<T*>(&X[(line & (half - 1)) * LS + jp]);
      dst[line >= half ? 1 : 0] = wre[e];
    }
  }
  C2<T>* R = tile_fft<-1, T>(X, Y, tw, d.plp, LS, half, tid);
  for (int idx = tid; idx < TL * n; idx += FL_NTH) {
    const int t = idx & (TL - 1), jj = idx >> tlsh;
    const int ip = ip0 + t;
    if (ip >= p) continue;
    const int jp = fl_pad(jj, n, p), jq = (p - jp) % p, tt = t & (half - 1);
    const C2<T> y = R[tt * LS + jp], ym = R[tt * LS + jq];
    C2<T> o;
    if (t < half) o = mk<T>((T)0.5 * (y.x + ym.x), (T)0.5 * (y.y - ym.y));     // (Y + conj(Ym)) / 2
    else o = mk<T>((T)0.5 * (y.y + ym.y), (T)-0.5 * (y.x - ym.x));              // (Y - conj(Ym)) / (2i)
    W2[((size_t)b * n + jj) * p + ip] = o;
  }
}

// ------------------------------------------------------------------ K3: forward pass along y, chop, rhs, RK4 stage
// grid (ceil(n/TL), B).  mode 0: out = rhs;  1: out = f0 + ca k, acc = f0 + cb k;  2: out = f0 + ca k, acc += cb k;
// 4: out = acc + cb k.
template <class T>
__global__ __launch_bounds__(FL_NTH) void fluid_k3_kernel(FluidDev<T> d, const C2<T>* __restrict__ W2,
                                                          const C2<T>* omg_s, const C2<T>* __restrict__ phat,
                                                          const C2<T>* f0, C2<T>* acc, C2<T>* out, int mode, T ca, T cb) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  C2<T>* X = reinterpret_cast<C2<T>*>(smem_raw);
  C2<T>* Y = X + d.TL * d.LS;
  C2<T>* tw = Y + d.TL * d.LS;
  const int tid = threadIdx.x, n = d.n, p = d.p, LS = d.LS;
  for (int k = tid; k < p; k += FL_NTH) tw[k] = d.twp[k];
  const int tpl = FL_NTH / d.TL, line = tid / tpl, u0 = tid - line * tpl;
  const int j = blockIdx.x * d.TL + line, b = blockIdx.y;
  const bool live = j < n;
  const C2<T>* w2 = W2 + ((size_t)b * n + (live ? j : 0)) * p;
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int ip = u0 + tpl * e;
    if (ip < p) X[line * LS + ip] = live ? w2[ip] : mk<T>(0, 0);
  }
  C2<T>* R = tile_fft<-1, T>(X, Y, tw, d.plp, LS, d.TL, tid);
  if (!live) return;
  const T kj = d.k[j];
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int ip = u0 + tpl * e;
    if (ip >= p) continue;
    const int i = fl_unpad(ip, n, p);
    if (i < 0) continue;
    const size_t off = ((size_t)b * n + j) * n + i;
    const T ki = d.k[i], lin = -d.nu * (kj * kj + ki * ki);
    const C2<T> o = omg_s[off], nl = R[line * LS + ip], ph = phat[off];
    const C2<T> k = mk<T>(lin * o.x + d.scale_out * nl.x + ph.x, lin * o.y + d.scale_out * nl.y + ph.y);
    if (mode == 0) {
      out[off] = k;
    } else if (mode == 4) {
      const C2<T> a = acc[off];
      out[off] = mk<T>(a.x + cb * k.x, a.y + cb * k.y);
    } else {
      const C2<T> f = f0[off];
      out[off] = mk<T>(f.x + ca * k.x, f.y + ca * k.y);
      if (mode == 1) acc[off] = mk<T>(f.x + cb * k.x, f.y + cb * k.y);
      else {
        const C2<T> a = acc[off];
        acc[off] = mk<T>(a.x + cb * k.x, a.y + cb * k.y);
      }
    }
  }
}

// ------------------------------------------------------------------ plain 2-D FFT passes of the n x n grid
// (sensing: y = real(ifft(env.y)), FluidSetup.jl:189,206; actuation: fft(p), :260)
template <class T, int SGN, bool REAL_IN>
__global__ __launch_bounds__(FL_NTH) void fluid_fft_fast_kernel(FluidDev<T> d, const void* __restrict__ in,
                                                                C2<T>* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  C2<T>* X = reinterpret_cast<C2<T>*>(smem_raw);
  C2<T>* Y = X + d.TLn * d.LSn;
  C2<T>* tw = Y + d.TLn * d.LSn;
  const int tid = threadIdx.x, n = d.n, LS = d.LSn;
  for (int k = tid; k < n; k += FL_NTH) tw[k] = d.twn[k];
  const int tpl = FL_NTH / d.TLn, line = tid / tpl, u0 = tid - line * tpl;
  const int j = blockIdx.x * d.TLn + line, b = blockIdx.y;
  const bool live = j < n;
  const size_t base = ((size_t)b * n + (live ? j : 0)) * n;
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int i = u0 + tpl * e;
    if (i < n) {
      C2<T> v = mk<T>(0, 0);
      if (live) v = REAL_IN ? mk<T>(static_cast<const T*>(in)[base + i], 0) : static_cast<const C2<T>*>(in)[base + i];
      X[line * LS + i] = v;
    }
  }
  C2<T>* R = tile_fft<SGN, T>(X, Y, tw, d.pln, LS, d.TLn, tid);
  if (!live) return;
#pragma unroll
  for (int e = 0; e < FL_MAXE; ++e) {
    const int i = u0 + tpl * e;
    if (i < n) out[base + i] = R[line * LS + i];
  }
}

template <class T, int SGN, bool REAL_OUT>
__global__ __launch_bounds__(FL_NTH) void fluid_fft_slow_kernel(FluidDev<T> d, const C2<T>* __restrict__ in,
                                                                void* __restrict__ out, T scale) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  C2<T>* X = reinterpret_cast<C2<T>*>(smem_raw);
  C2<T>* Y = X + d.TLn * d.LSn;
  C2<T>* tw = Y + d.TLn * d.LSn;
  const int tid = threadIdx.x, n = d.n, LS = d.LSn, TL = d.TLn;
  for (int k = tid; k < n; k += FL_NTH) tw[k] = d.twn[k];
  const int tlsh = 31 - __clz(TL);
  const int i0 = blockIdx.x * TL, b = blockIdx.y;
  for (int idx = tid; idx < TL * n; idx += FL_NTH) {
    const int t = idx & (TL - 1), j = idx >> tlsh, i = i0 + t;
    X[t * LS + j] = i < n ? in[((size_t)b * n + j) * n + i] : mk<T>(0, 0);
  }
  C2<T>* R = tile_fft<SGN, T>(X, Y, tw, d.pln, LS, TL, tid);
  for (int idx = tid; idx < TL * n; idx += FL_NTH) {
    const int t = idx & (TL - 1), j = idx >> tlsh, i = i0 + t;
    if (i >= n) continue;
    const C2<T> v = R[t * LS + j];
    const size_t off = ((size_t)b * n + j) * n + i;
    if (REAL_OUT) static_cast<T*>(out)[off] = v.x * scale;
    else static_cast<C2<T>*>(out)[off] = mk<T>(v.x * scale, v.y * scale);
  }
}

// ------------------------------------------------------------------ sensing / actuation
// dots[b][s] = <y, gaussians[s]>: one wave per sensor over its BW x BH box   (FluidSetup.jl:196,216)
template <class T>
__global__ __launch_bounds__(256) void fluid_dots_kernel(int n, int S, int BH, int BW, const T* __restrict__ boxes,
                                                         const int* __restrict__ origin, const T* __restrict__ yreal,
                                                         T* __restrict__ dots) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int s = blockIdx.x * 4 + wv, b = blockIdx.y;
  if (s >= S) return;
  const int j0 = origin[2 * s], i0 = origin[2 * s + 1];
  const T* bx = boxes + (size_t)s * BH * BW;
  const T* y = yreal + (size_t)b * n * n;
  T acc = 0;
  for (int e = lane; e < BH * BW; e += 64) {
    const int dj = e / BH, di = e - dj * BH;
    int jj = j0 + dj, ii = i0 + di;
    if (jj >= n) jj -= n;
    if (ii >= n) ii -= n;
    acc += bx[e] * y[(size_t)jj * n + ii];
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) dots[(size_t)b * S + s] = acc;
}

template <class T>
struct FeatArgs {
  int S, A, spa, window, ns, check_max;
  int mem, na;      // action memory (cfg.memory_size): the last mem rows of a state column = rows 1.. of the action [A][na]
  T sensor_scale, r_in_scale, r_power, r_denom, a_pun, da_pun, max_value;
  const int* a2s;
};

// featurize (3x3 circular window of the spa x spa sensor grid, FluidSetup.jl:219-224) + reward (:188-202)
template <class T>
__global__ __launch_bounds__(256) void fluid_feat_kernel(FeatArgs<T> g, const T* __restrict__ dots,
                                                         const T* __restrict__ action, const T* __restrict__ action_prev,
                                                         const T* __restrict__ state_prev, T* __restrict__ state_out,
                                                         T* __restrict__ reward_out, int32_t* __restrict__ done) {
  __shared__ int flag;
  const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;
  const T* dt = dots + (size_t)b * g.S;
  if (tid == 0) flag = 0;
  __syncthreads();
  if (state_out) {
    const int w = g.window / 2, fresh = g.window * g.window;
    const T* prev = state_prev ? state_prev + (size_t)b * g.A * g.ns : nullptr;
    for (int idx = tid; idx < g.A * g.ns; idx += nt) {
      const int a = idx / g.ns, rr = idx - a * g.ns;
      T v;
      if (rr >= g.ns - g.mem) {          // FluidSetup.jl:236-241: env.action[end-(memory_size-1):end, :], zeros without env
        v = action ? action[((size_t)b * g.A + a) * g.na + 1 + (rr - (g.ns - g.mem))] : (T)0;
      } else if (rr < fresh || prev == nullptr) {
        const int r0 = rr % fresh, wi = r0 / g.window - w, wj = r0 % g.window - w;
        const int e = g.a2s[a];
        int row = (e / g.spa - wi) % g.spa, col = (e % g.spa - wj) % g.spa;
        if (row < 0) row += g.spa;
        if (col < 0) col += g.spa;
        v = dt[row * g.spa + col] * g.sensor_scale;
      } else {
        v = prev[a * g.ns + (rr - fresh)];
      }
      state_out[(size_t)b * g.A * g.ns + idx] = v;
    }
  }
  if (reward_out) {
    for (int a = tid; a < g.A; a += nt) {
      const T dd = fabs(g.r_in_scale * dt[g.a2s[a]]);
      const T pw = dd == 0 ? (T)0 : (T)pow((double)dd, (double)g.r_power);
      const T ac = action[((size_t)b * g.A + a) * g.na], da = ac - action_prev[((size_t)b * g.A + a) * g.na];   // row 1 (:200)
      const T r = -fabs(pw / g.r_denom) - g.a_pun * ac * ac - g.da_pun * da * da;
      reward_out[(size_t)b * g.A + a] = r;
      if (g.check_max == 2 && !(fabs(r) <= g.max_value)) atomicOr(&flag, 1);
    }
    __syncthreads();
    if (done && tid == 0 && g.check_max == 2) done[b] = flag;
  }
}

// max|y| > max_value on the spectrum (check_max_value == "y", src/PDEenv.jl:226-231)
template <class T>
__global__ __launch_bounds__(256) void fluid_maxabs_kernel(int nn, T max_value, const C2<T>* __restrict__ y,
                                                           int32_t* __restrict__ done) {
  __shared__ int flag;
  const int b = blockIdx.x;
  if (threadIdx.x == 0) flag = 0;
  __syncthreads();
  int f = 0;
  for (int i = threadIdx.x; i < nn; i += blockDim.x) {
    const C2<T> v = y[(size_t)b * nn + i];
    if (!(hypot(v.x, v.y) <= max_value)) f = 1;
  }
  if (f) atomicOr(&flag, 1);
  __syncthreads();
  if (threadIdx.x == 0) done[b] = flag;
}

// p = sum_a agent_power * action[a] * gaussians_actuators[a]   (FluidSetup.jl:254-258): one 16x16 block of
// cells per workgroup, candidates = the actuators whose box meets the block (CSR built at setup time)
template <class T>
__global__ __launch_bounds__(256) void fluid_actuate_kernel(int n, int A, int BH, int BW, int nb1,
                                                            const T* __restrict__ boxes, const int* __restrict__ origin,
                                                            const int* __restrict__ blk_ptr, const int* __restrict__ blk_idx,
                                                            const T* __restrict__ action, int na, T power, T* __restrict__ preal) {
  const int blk = blockIdx.x, b = blockIdx.y;
  const int bj = blk / nb1, bi = blk - bj * nb1;
  const int di = threadIdx.x & 15, dj = threadIdx.x >> 4;
  const int j = bj * 16 + dj, i = bi * 16 + di;
  if (j >= n || i >= n) return;
  T acc = 0;
  for (int q = blk_ptr[blk]; q < blk_ptr[blk + 1]; ++q) {
    const int a = blk_idx[q];
    int dj2 = j - origin[2 * a], di2 = i - origin[2 * a + 1];
    if (dj2 < 0) dj2 += n;
    if (di2 < 0) di2 += n;
    if (dj2 < BW && di2 < BH) acc += (power * action[((size_t)b * A + a) * na]) * boxes[((size_t)a * BW + dj2) * BH + di2];
  }
  preal[((size_t)b * n + j) * n + i] = acc;
}

#ifndef FL_K2_TC
#define FL_K2_TC 4
#endif
// ------------------------------------------------------------------ wave-FFT variants of K1 / K2 / K3
// Same three passes, but every line transform runs in the registers of ONE wave (wave_fft.hpp): no LDS stage
// round trips, no barriers inside a transform.  LDS is only used to turn coalesced global accesses into the
// permuted (digit-reversed) element order the transforms consume / produce, and for K2's column transposition.
// Used when the padded length p is one of 64/128/192/256/384/512/768 (otherwise the LDS-tile kernels above); 64 and
// 192 run two lines per wave (half-wave transforms).

// the spectral part of K1w for one work item: from the two spectrum lines in LDS (Lj: line j0, Lm: its mirror j1) the
// packed inverse transforms of the carried line t and -- nhalf = 2 -- of its mirror line s_mirror
template <int E, int Q, int LB>
__device__ __forceinline__ void fluid_k1w_body(const FluidDev<double>& d, const C2<double>* Lj, const C2<double>* Lm, int j0, int j1,
                                               int t, int s_mirror, int nhalf, C2<double>* __restrict__ W, int b,
                                               WaveFftD<E, Q, LB>& f, int l, const double* __restrict__ kt) {
  typedef WaveFftD<E, Q, LB> F;
  typedef C2<double> Z;
  const int n = d.n, p = d.p;
  const Z zero = mk<double>(0, 0);
#pragma unroll 1
  for (int half = 0; half < nhalf; ++half) {
    const Z* La = half ? Lm : Lj;     // the line's own spectrum / its mirror's
    const Z* Lb = half ? Lj : Lm;
    const int j = half ? j1 : j0, jm = half ? j0 : j1, s = half ? s_mirror : t;
    const double kj = j >= 0 ? kt[j] : 0.0, kjm = jm >= 0 ? kt[jm] : 0.0;
#pragma unroll 1
    for (int fld = 0; fld < 2; ++fld) {
      Z a[F::R];
#pragma unroll
      for (int jj = 0; jj < F::R; ++jj) {
        const int ip = f.mode_index(jj);
        const int i = fl_unpad(ip, n, p), im = fl_unpad((p - ip) % p, n, p);
        // fl_spec() of the mode and of its mirror, but only the two spectra this field needs (the velocities need
        // psihat = omghat ./ k^2, the vorticity gradients do not)
        const bool va = j >= 0 && i >= 0, vm = jm >= 0 && im >= 0;
        Z oa = zero, om_ = zero;
        double ki = 0.0, kim = 0.0;
        if (va) { oa = La[i]; ki = kt[i]; }
        if (vm) { om_ = Lb[im]; kim = kt[im]; }
        if (fld == 0) {   // Z1 = Herm(u) + i Herm(v),  u = i ky psi, v = -i kx psi
          // one reciprocal serves the four quotients of the pair (k^2 of a mode and of its mirror are the same number;
          // omghat * (1 / k^2) instead of omghat / k^2: <= 1 ulp from the reference's quotient)
          const double k2 = va ? kj * kj + ki * ki : kjm * kjm + kim * kim;
          const double r = 1.0 / k2;
          const Z pa = (va && !(i == 0 && j == 0)) ? mk<double>(oa.x * r, oa.y * r) : zero;
          const Z pm = (vm && !(im == 0 && jm == 0)) ? mk<double>(om_.x * r, om_.y * r) : zero;
          const Z au = mk<double>(-ki * pa.y, ki * pa.x), av = mk<double>(kj * pa.y, -kj * pa.x);
          const Z mu = mk<double>(-kim * pm.y, kim * pm.x), mv = mk<double>(kjm * pm.y, -kjm * pm.x);
          const Z hu = mk<double>(0.5 * (au.x + mu.x), 0.5 * (au.y - mu.y)), hv = mk<double>(0.5 * (av.x + mv.x), 0.5 * (av.y - mv.y));
          a[jj] = mk<double>(hu.x - hv.y, hu.y + hv.x);
        } else {          // Z2 = Herm(wx) + i Herm(wy),  wx = i kx omg, wy = i ky omg
          const Z ax = mk<double>(-kj * oa.y, kj * oa.x), ay = mk<double>(-ki * oa.y, ki * oa.x);
          const Z mx = mk<double>(-kjm * om_.y, kjm * om_.x), my = mk<double>(-kim * om_.y, kim * om_.x);
          const Z hx = mk<double>(0.5 * (ax.x + mx.x), 0.5 * (ax.y - mx.y)), hy = mk<double>(0.5 * (ay.x + my.x), 0.5 * (ay.y - my.y));
          a[jj] = mk<double>(hx.x - hy.y, hx.y + hy.x);
        }
      }
      f.inverse(a);
      if (d.wtile) {
        // tile-major: element x of line s goes to tile x / 8, row s, column x % 8 -- 128-byte pieces that are CONTIGUOUS over
        // the lines of a tile (consecutive waves of a workgroup write consecutive rows), so that the x-pass reads whole tiles
        Z* w = W + ((size_t)b * 2 + fld) * d.nl * p + (size_t)s * 8;
#pragma unroll
        for (int jj = 0; jj < F::R; ++jj) {
          const int x = l + F::LANES * jj;
          w[(size_t)(x >> 3) * d.nl * 8 + (x & 7)] = a[jj];
        }
      } else {
        Z* w = W + (((size_t)b * 2 + fld) * d.nl + s) * p;
#pragma unroll
        for (int jj = 0; jj < F::R; ++jj) w[l + F::LANES * jj] = a[jj];
      }
    }
  }
}

// K1w: one wave per PAIR of carried lines (a line and its mirror, see below).  LDS: per wave the two spectrum lines j and mirror(j) (2 n complex).
template <int E, int Q, int LB>
__global__ __launch_bounds__(256) void fluid_k1w_kernel(FluidDev<double> d, const C2<double>* __restrict__ omg,
                                                        C2<double>* __restrict__ W, int pair) {
  typedef WaveFftD<E, Q, LB> F;
  typedef C2<double> Z;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  // a "line slot" is a wave (LB = 6) or a half wave (LB = 5: two lines per wave); l = position inside the line
  const int lane = threadIdx.x & 63, l = lane & (F::LANES - 1), slot = (threadIdx.x >> 6) * F::LPW + (lane >> LB);
  const int n = d.n, p = d.p;
  Z* Lj = reinterpret_cast<Z*>(smem_raw) + (size_t)slot * 2 * n;
  Z* Lm = Lj + n;
  // the wavenumber table in LDS (round 4): the spectral part reads k of every mode and of its mirror -- 48 dependent global
  // loads per wave item, each with its own wait, in the middle of the fp64 arithmetic
  double* kt = reinterpret_cast<double*>(smem_raw + (size_t)4 * F::LPW * 2 * n * sizeof(Z));
  for (int i = threadIdx.x; i < n; i += 256) kt[i] = d.k[i];
  __syncthreads();
  // work item t = carried line t (jp = t <= n/2) TOGETHER WITH its mirror line (jp' = p - t): both need exactly the
  // spectrum lines j and mirror(j) -- with the roles swapped -- so one load of the two lines serves two output lines
  // (half the reads of omg and half the exposed load latency per transform); line 0 is its own mirror
  // (pair = 0: one carried line per slot, every line loads its two spectrum lines itself -- more, shorter waves: better
  // for the small grids, n = 128: 12.8 vs 15.2 us)
  const int t = blockIdx.x * 4 * F::LPW + slot, b = blockIdx.y;
  if (t >= (pair ? n / 2 + 1 : d.nl)) return;             // whole line slot; no workgroup barrier below this point
  const int jp = fl_line_jp(t, n, p, d.nl), jpm = (p - jp) % p;
  const int s_mirror = fl_line_of(jpm, n, p, d.nl);
  const int j0 = fl_unpad(jp, n, p), j1 = fl_unpad(jpm, n, p);
  const Z zero = mk<double>(0, 0);
  for (int i = l; i < n; i += F::LANES) {
    Lj[i] = j0 >= 0 ? omg[((size_t)b * n + j0) * n + i] : zero;
    Lm[i] = j1 >= 0 ? omg[((size_t)b * n + j1) * n + i] : zero;
  }
  __builtin_amdgcn_wave_barrier();
  F f;
  f.init(d.twp, lane);
  const int nhalf = (pair && s_mirror >= 0 && s_mirror != t) ? 2 : 1;
  fluid_k1w_body<E, Q, LB>(d, Lj, Lm, j0, j1, t, s_mirror, nhalf, W, b, f, l, kt);
}

// K2w: TC columns per workgroup, one wave per column.  LDS: the [TC][p] column tile (transposition + permuted access).
// TC = 4 keeps the tile at 48 KB so that several workgroups share a CU and one's global loads / stores overlap the
// others' transforms (with TC = 8 a CU holds a single workgroup and load -> transform -> store serialise).
template <int E, int Q, int TC, int LB>
__global__ __launch_bounds__(64 * TC >> (6 - LB)) void fluid_k2w_kernel(FluidDev<double> d, const C2<double>* __restrict__ W,
                                                            C2<double>* __restrict__ W2) {
  typedef WaveFftD<E, Q, LB> F;
  typedef C2<double> Z;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  Z* tile = reinterpret_cast<Z*>(smem_raw);
  constexpr int NT = 64 * TC / F::LPW;                      // one line slot (wave or half wave) per column
  const int tid = threadIdx.x, lane = tid & 63, wv = (tid >> 6) * F::LPW + (lane >> LB), n = d.n, p = d.p, LS = p + 1;
  const int ip0 = blockIdx.x * TC, b = blockIdx.y;
  F f;
  f.init(d.twp, lane);
  Z r0[F::R], a[F::R];
  // tile element (t, jp) of a field: W[fld][s(jp)][ip0 + t] or 0 (pad() along x); thread-strided over the TC x p tile
  constexpr int NPF = (TC * F::N + NT - 1) / NT;            // tile elements per thread
  auto tile_src = [&](int fld, int idx) -> Z {
    const int t = idx % TC, jp = idx / TC;
    const int s = fl_line_of(jp, n, p, d.nl), ip = ip0 + t;
    if (s >= 0 && ip < p) return W[(((size_t)b * 2 + fld) * d.nl + s) * p + ip];
    return mk<double>(0, 0);
  };
  for (int idx = tid; idx < TC * p; idx += NT) tile[(idx % TC) * LS + idx / TC] = tile_src(0, idx);
  Z pre[NPF];                                               // field 1 is fetched while field 0 is transformed
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    const int idx = tid + u * NT;
    pre[u] = idx < TC * p ? tile_src(1, idx) : mk<double>(0, 0);
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) r0[jj] = tile[wv * LS + f.mode_index(jj)];
  f.inverse(r0);
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NPF; ++u) {
    const int idx = tid + u * NT;
    if (idx < TC * p) tile[(idx % TC) * LS + idx / TC] = pre[u];
  }
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) a[jj] = tile[wv * LS + f.mode_index(jj)];
  f.inverse(a);
  // -(u wx + v wy), both ifft scalings; forward transform of the real product (one column per wave)
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) a[jj] = mk<double>(-(r0[jj].x * a[jj].x + r0[jj].y * a[jj].y) * d.inv2, 0.0);
  f.forward(a);
  __syncthreads();
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) tile[wv * LS + f.mode_index(jj)] = a[jj];
  __syncthreads();
  for (int idx = tid; idx < TC * n; idx += NT) {        // chop() along x + coalesced store
    const int t = idx % TC, jj = idx / TC, ip = ip0 + t;
    if (ip < p) W2[((size_t)b * n + jj) * p + ip] = tile[t * LS + fl_pad(jj, n, p)];
  }
}

// K2p (round 4): the same x-pass as a PERSISTENT, software-pipelined kernel -- one 8-wave workgroup per CU walks its share of
// the (trajectory, 8-column) tiles; the field tile needed NEXT streams into LDS by LDS-DMA (no VGPR staging) while the waves
// run the line transforms of the current one out of registers.
//   * a tile is TC = 8 columns wide: every global access of W and W2 is a whole 128-byte line (8 complex fp64) -- K2w's TC = 4
//     moves 64-byte half lines -- and only the nl non-zero lines of a column are ever fetched or kept in LDS (pad() is an index map);
//   * LDS: one field region R, [nl rounded up to 8][8] complex (65 KiB at n = 512), an output staging region S, [n][8] (64 KiB), and
//     the radix-Q twiddle table (8 KiB).  A DMA piece (1 KiB per wave instruction) is 8 lines x 8 columns, lane l fetching line
//     8c + (l >> 3), column (l & 7) ^ ((line >> 2) & 7): the column swizzle spreads the 16-byte reads of one wave (lines a
//     multiple of 4 apart in the digit-reversed order the transforms consume) over the banks; S is swizzled the same way;
//   * per tile:  [f0 landed] read column | barrier | DMA f1 -> R | inverse -> r0 | [f1 landed] read column | barrier |
//     DMA f0 of the NEXT tile -> R | inverse -> a | product, forward | column -> S | barrier | 128-byte-line stores of S.
//     f1 has one transform (~5 us) to land, the next f0 two.  Vector-memory operations retire in issue order on vmcnt (loads
//     and stores alike): "f0(next) has landed" = at most the 8 store instructions issued behind its DMA are outstanding; the
//     barriers are raw (s_waitcnt lgkmcnt(0); s_barrier) -- __syncthreads() would drain the prefetch.
//     (First cut: two field regions, outputs stored straight from the transform's registers -- 16-byte pieces of 64 different
//     lines per instruction: 9.4 M partial-line writes per launch cost 23-34 us that no amount of overlap hid.)
// Same arithmetic per column as K2w (bit-identical W2).
__device__ __forceinline__ void k2p_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <int N>
__device__ __forceinline__ void k2p_wait_but() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NPW: DMA pieces per wave and field = ceil(ceil(nl / 8) / 8); NSU: output elements per thread = n * 8 / 512
template <int E, int Q, int NPW, int NSU>
__global__ __launch_bounds__(512) void fluid_k2p_kernel(FluidDev<double> d, const C2<double>* __restrict__ W,
                                                        C2<double>* __restrict__ W2, int ntiles) {
  typedef WaveFftD<E, Q, 6, true> F;      // radix-Q twiddles from an LDS table: the kernel sits near the 256-VGPR limit of 2 waves / SIMD
  typedef C2<double> Z;
  constexpr int TC = 8;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, n = d.n, p = d.p, nl = d.nl;
  const int npieces = (nl + 7) / 8;
  Z* R = reinterpret_cast<Z*>(smem_raw);
  Z* S = R + (size_t)npieces * 64;        // [n][8]
  Z* WQ = S + (size_t)n * TC;             // [(Q - 1) E][64]
  const int tiles_per_b = p / TC;
  F f;
  f.init(d.twp, lane, WQ, wv == 0);
  __syncthreads();
  // LDS element of (line of the mode in slot jj, my column) or -1 (a pad() zero); recomputed where used (a few integer
  // operations per slot against a 768-point transform) instead of held in registers
  auto src_of = [&](int jj) -> int {
    const int sl = fl_line_of(f.mode_index(jj), n, p, nl);
    return sl >= 0 ? sl * TC + (wv ^ ((sl >> 2) & 7)) : -1;
  };
  // the transform's twiddles are ordinary global loads: they must have landed BEFORE the first DMA is issued -- the compiler
  // waits vmcnt(0) at the first use of a global load, which would drain every prefetch in flight behind it
  f.touch();
  auto dma_tile = [&](int fld, int tile) {
    int lv = lane;
    asm volatile("" : "+v"(lv));                         // opaque per call: the offsets below are not hoisted (and spilled)
    const int b = tile / tiles_per_b, ip0 = (tile - b * tiles_per_b) * TC;
    // (W is tile-major for this kernel, FluidDev::wtile: tile ip0 / 8 of a field is the contiguous block [nl][8])
    const char* base = reinterpret_cast<const char*>(W + ((size_t)b * 2 + fld) * nl * p + (size_t)(ip0 >> 3) * nl * 8);      // wave-uniform
#pragma unroll
    for (int j = 0; j < NPW; ++j) {
      // piece c: lines 8c .. 8c+7; this lane's element (recomputed per issue: kept in registers the offsets spill, and a
      // scratch reload is a vector-memory load whose wait drains the DMA queue)
      int c = wv + 8 * j;
      c = c < npieces ? c : npieces - 1;
      int sl = c * 8 + (lv >> 3);
      sl = sl < nl ? sl : nl - 1;                        // rows past nl - 1 of the last piece: never read
      const unsigned off = (unsigned)(sl * 8 + ((lv & 7) ^ ((sl >> 2) & 7))) * (unsigned)sizeof(Z);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                       (__attribute__((address_space(3))) void*)(R + (size_t)c * 64), 16, 0, 0);
    }
  };
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  dma_tile(0, tile);
  const Z zero = mk<double>(0, 0);
  bool first = true;
  for (; tile < ntiles; tile += gridDim.x) {
    const int next = tile + gridDim.x;
    const bool more = next < ntiles;
    Z r0[F::R], a[F::R];
    // ---- field 0 has landed: behind its DMA only the NSU stores of the previous tile were issued
    if (first) k2p_wait_but<0>(); else k2p_wait_but<NSU>();
    first = false;
    k2p_lds_barrier();
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) { const int si = src_of(jj); r0[jj] = si >= 0 ? R[si] : zero; }
    k2p_lds_barrier();                                   // every wave has read field 0
    dma_tile(1, tile);
    f.inverse(r0);
    // ---- field 1 (nothing younger than its DMA is in flight; the stores of the previous tile are older and retire first)
    k2p_wait_but<0>();
    k2p_lds_barrier();
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) { const int si = src_of(jj); a[jj] = si >= 0 ? R[si] : zero; }
    k2p_lds_barrier();                                   // every wave has read field 1
    if (more) dma_tile(0, next);
    f.inverse(a);
    // ---- -(u wx + v wy), both ifft scalings; forward transform of the real product (one column per wave)
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) a[jj] = mk<double>(-(r0[jj].x * a[jj].x + r0[jj].y * a[jj].y) * d.inv2, 0.0);
    f.forward(a);
    // ---- chop() along x: kept modes of my column into S (S was last read two barriers ago), then whole-line stores.
    // (Round 6, measured: the stores issued in the middle of the NEXT tile by all waves, or at its start by waves 4 - 7 only while
    // their SIMD partners transform, make the launch 7 - 14 us LONGER -- HISTORY.md 6.5.)
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) {
      const int kk = fl_unpad(f.mode_index(jj), n, p);
      if (kk >= 0) S[kk * TC + (wv ^ ((kk >> 2) & 7))] = a[jj];
    }
    k2p_lds_barrier();
    const int b = tile / tiles_per_b, ip0 = (tile - b * tiles_per_b) * TC;
    Z* out = W2 + (size_t)b * n * p + ip0;
#pragma unroll
    for (int u = 0; u < NSU; ++u) {
      const int idx = tid + 512 * u, kk = idx >> 3, col = idx & 7;
      out[(size_t)kk * p + col] = S[kk * TC + (col ^ ((kk >> 2) & 7))];
    }
  }
}

// K3w: one wave per kept line j.  LDS: per wave one n-complex line (digit-reversed -> natural for coalesced global access).
template <int E, int Q, int LB>
__global__ __launch_bounds__(256) void fluid_k3w_kernel(FluidDev<double> d, const C2<double>* __restrict__ W2,
                                                        const C2<double>* omg_s, const C2<double>* __restrict__ phat,
                                                        const C2<double>* f0, C2<double>* acc, C2<double>* out, int mode,
                                                        double ca, double cb) {
  typedef WaveFftD<E, Q, LB> F;
  typedef C2<double> Z;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l = lane & (F::LANES - 1), slot = (threadIdx.x >> 6) * F::LPW + (lane >> LB);
  const int n = d.n, p = d.p;
  Z* Ln = reinterpret_cast<Z*>(smem_raw) + (size_t)slot * n;
  const int j = blockIdx.x * 4 * F::LPW + slot, b = blockIdx.y;
  if (j >= n) return;
  F f;
  f.init(d.twp, lane);
  Z a[F::R];
  const Z* w2 = W2 + ((size_t)b * n + j) * p;
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) a[jj] = w2[l + F::LANES * jj];
  f.forward(a);
#pragma unroll
  for (int jj = 0; jj < F::R; ++jj) {
    const int i = fl_unpad(f.mode_index(jj), n, p);          // chop() along y
    if (i >= 0) Ln[i] = a[jj];
  }
  __builtin_amdgcn_wave_barrier();
  const double kj = d.k[j];
  for (int i = l; i < n; i += F::LANES) {
    const size_t off = ((size_t)b * n + j) * n + i;
    const double ki = d.k[i], lin = -d.nu * (kj * kj + ki * ki);
    const Z o = omg_s[off], nlv = Ln[i], ph = phat[off];
    const Z k = mk<double>(lin * o.x + d.scale_out * nlv.x + ph.x, lin * o.y + d.scale_out * nlv.y + ph.y);
    if (mode == 0) {
      out[off] = k;
    } else if (mode == 4) {
      const Z ac = acc[off];
      out[off] = mk<double>(ac.x + cb * k.x, ac.y + cb * k.y);
    } else {
      const Z fv = f0[off];
      out[off] = mk<double>(fv.x + ca * k.x, fv.y + ca * k.y);
      if (mode == 1) acc[off] = mk<double>(fv.x + cb * k.x, fv.y + cb * k.y);
      else {
        const Z ac = acc[off];
        acc[off] = mk<double>(ac.x + cb * k.x, ac.y + cb * k.y);
      }
    }
  }
}

// K31w = K3w of one RK4 stage fused with K1w of the NEXT right-hand side (src/fluid_rk4.jl:122-190 inside rk4, :192-229):
// the stage value K3 writes for a kept line j is exactly the spectrum line K1 reads next, and K1 of a line pair (t, mirror)
// needs only the lines j0, j1 of that pair -- so a wave finishes the stage for its two lines (forward transform along y of
// W2, chop, linear term + forcing, stage update; `out` / `acc` to HBM as K3w does), keeps the two new spectrum lines in
// LDS and runs the K1w body from there.  Saves the re-read of the stage value (4 MB per trajectory and RHS at n = 512)
// and one launch per RHS.  mode 1 / 2 / 4 as in K3w (mode 0, the bare RHS, has no successor).  Pair items only (n >= 256).
template <int E, int Q, int LB>
__global__ __launch_bounds__(256) void fluid_k31w_kernel(FluidDev<double> d, const C2<double>* __restrict__ W2,
                                                         const C2<double>* omg_s, const C2<double>* __restrict__ phat,
                                                         const C2<double>* f0, C2<double>* acc, C2<double>* out, int mode,
                                                         double ca, double cb, C2<double>* __restrict__ W) {
  typedef WaveFftD<E, Q, LB> F;
  typedef C2<double> Z;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int lane = threadIdx.x & 63, l = lane & (F::LANES - 1), slot = (threadIdx.x >> 6) * F::LPW + (lane >> LB);
  const int n = d.n, p = d.p;
  Z* Lj = reinterpret_cast<Z*>(smem_raw) + (size_t)slot * 2 * n;
  Z* Lm = Lj + n;
  const int t = blockIdx.x * 4 * F::LPW + slot, b = blockIdx.y;
  if (t > n / 2) return;
  const int jp = fl_line_jp(t, n, p, d.nl), jpm = (p - jp) % p;
  const int s_mirror = fl_line_of(jpm, n, p, d.nl);
  const int j0 = fl_unpad(jp, n, p), j1 = fl_unpad(jpm, n, p);
  const Z zero = mk<double>(0, 0);
  F f;
  f.init(d.twp, lane);
  // ---- K3 for the kept lines j0 and (if it is another line) j1
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    const int j = h ? j1 : j0;
    Z* Ln = h ? Lm : Lj;
    if (h && (j1 < 0 || j1 == j0)) {
      if (j1 < 0)                                   // the mirror of the Nyquist line lies in the padding: zero spectrum
        for (int i = l; i < n; i += F::LANES) Ln[i] = zero;
      break;
    }
    Z a[F::R];
    const Z* w2 = W2 + ((size_t)b * n + j) * p;
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) a[jj] = w2[l + F::LANES * jj];
    f.forward(a);
#pragma unroll
    for (int jj = 0; jj < F::R; ++jj) {
      const int i = fl_unpad(f.mode_index(jj), n, p);          // chop() along y
      if (i >= 0) Ln[i] = a[jj];
    }
    __builtin_amdgcn_wave_barrier();
    const double kj = d.k[j];
    for (int i = l; i < n; i += F::LANES) {
      const size_t off = ((size_t)b * n + j) * n + i;
      const double ki = d.k[i], lin = -d.nu * (kj * kj + ki * ki);
      const Z o = omg_s[off], nlv = Ln[i], ph = phat[off];
      const Z k = mk<double>(lin * o.x + d.scale_out * nlv.x + ph.x, lin * o.y + d.scale_out * nlv.y + ph.y);
      Z nv;
      if (mode == 4) {
        const Z ac = acc[off];
        nv = mk<double>(ac.x + cb * k.x, ac.y + cb * k.y);
      } else {
        const Z fv = f0[off];
        nv = mk<double>(fv.x + ca * k.x, fv.y + ca * k.y);
        if (mode == 1) acc[off] = mk<double>(fv.x + cb * k.x, fv.y + cb * k.y);
        else {
          const Z ac = acc[off];
          acc[off] = mk<double>(ac.x + cb * k.x, ac.y + cb * k.y);
        }
      }
      out[off] = nv;
      Ln[i] = nv;                                   // the spectrum line of the next right-hand side
    }
  }
  __builtin_amdgcn_wave_barrier();
  // ---- K1 of the next right-hand side from the two lines
  const Z* Lmm = (j1 == j0) ? Lj : Lm;              // line 0 is its own mirror
  const int nhalf = (s_mirror >= 0 && s_mirror != t) ? 2 : 1;
  fluid_k1w_body<E, Q, LB>(d, Lj, Lmm, j0, j1, t, s_mirror, nhalf, W, b, f, l, d.k);
}

// ------------------------------------------------------------------ wave FFT unit-test entry (pdec_debug_wave_fft)
template <int E, int Q, int LB>
__global__ __launch_bounds__(256) void wave_fft_debug_kernel(const C2<double>* __restrict__ in, C2<double>* __restrict__ out,
                                                            const C2<double>* __restrict__ tw, int nlines, int sgn) {
  typedef WaveFftD<E, Q, LB> F;
  const int lane = threadIdx.x & 63, l = lane & (F::LANES - 1);
  const int line = (blockIdx.x * 4 + (threadIdx.x >> 6)) * F::LPW + (lane >> LB);
  if (line >= nlines) return;
  F f;
  f.init(tw, lane);
  C2<double> a[F::R];
  const C2<double>* x = in + (size_t)line * F::N;
  C2<double>* y = out + (size_t)line * F::N;
  if (sgn < 0) {
#pragma unroll
    for (int j = 0; j < F::R; ++j) a[j] = x[l + F::LANES * j];
    f.forward(a);
#pragma unroll
    for (int j = 0; j < F::R; ++j) y[f.mode_index(j)] = a[j];
  } else {
#pragma unroll
    for (int j = 0; j < F::R; ++j) a[j] = x[f.mode_index(j)];
    f.inverse(a);
#pragma unroll
    for (int j = 0; j < F::R; ++j) y[l + F::LANES * j] = a[j];
  }
}

// ------------------------------------------------------------------ host side

// Episode initialiser on the device: omega = sum over vortices of the 9 periodic images of a
// Taylor vortex U/a (2 - r^2/a^2) exp((1 - r^2/a^2)/2)   (src/fluid_rk4.jl:54-69, ic(...) :72-120).  One thread per
// cell, same summation order as the reference (vortex outer, image offsets i, j inner); vort [B][nv][4] = x0, y0, a, U.
template <class T>
__global__ void fluid_ic_kernel(int n, int nv, T Lx, T Ly, const T* __restrict__ vort, T* __restrict__ out) {
  extern __shared__ __align__(16) unsigned char ic_smem[];
  T* sv = reinterpret_cast<T*>(ic_smem);
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 4 * nv; i += blockDim.x) sv[i] = vort[(size_t)b * 4 * nv + i];
  __syncthreads();
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * n) return;
  const int jx = idx / n, iy = idx - jx * n;            // memory [nx][ny]: y is the fast axis
  const T x = (T)jx * (Lx / (T)n), y = (T)iy * (Ly / (T)n);
  T total = 0;
  for (int v = 0; v < nv; ++v) {
    const T x0 = sv[4 * v], y0 = sv[4 * v + 1], a0 = sv[4 * v + 2], U = sv[4 * v + 3];
    T omg = 0;
    for (int i = -1; i <= 1; ++i)
      for (int j = -1; j <= 1; ++j) {
        const T ddx = x - x0 - (T)i * Lx, ddy = y - y0 - (T)j * Ly;
        const T r2 = ddx * ddx + ddy * ddy;
        omg = omg + U / a0 * ((T)2 - r2 / (a0 * a0)) * exp((T)0.5 * ((T)1 - r2 / (a0 * a0)));
      }
    total = total + omg;
  }
  out[(size_t)b * n * n + idx] = total;
}

struct FluidEnv : Env {
  int n = 0, p = 0, nl = 0, TL = 0, TLn = 0, BH = 0, BW = 0, nb1 = 0;
  FftPlan plp, pln;
  DevBuf k, twp, twn, sbox, sorg, abox, aorg, a2s_d, blkptr, blkidx;
  DevBuf W, W2, fs, acc, yreal, tmpc, dots, phat, icv;
  size_t lds_p = 0, lds_n = 0;
  int wave_E = 0, wave_Q = 0;     // != 0: the one-line-per-wave transforms serve the padded length p
  int wave_LB = 6;                // 5: one line per HALF wave (p = 192, 64)
  // round 4: two child environments of B/2 trajectories each (own work arrays, own stream) that fluid_env_step runs side by
  // side -- trajectories are independent, and the kernels of a right-hand side differ in what bounds them (K1 / K2: fp64
  // issue, K3: HBM), so one half's K3 hides under the other half's transforms (n = 512, B = 16: 96.4 -> 100.8 env-steps/s)
  static constexpr int MAXPART = PartStreams::MAX;
  int nparts = 0;
  std::unique_ptr<FluidEnv> half[MAXPART];
  // round 5: the children's streams through the helper both split environments share (common.hpp PartStreams):
  // library-made streams are created at the first split step at the priority level of the environment's stream, as the 2-D
  // Keller-Segel environment does -- not at creation time at the lowest level, before pdec_set_stream had named that stream
  PartStreams ps;
  int part_streams() const override { return nparts >= 2 ? nparts - 1 : 0; }
  // the children exist since pdec_fluid_env_create: the caller's streams replace the library's one for one (all of them, or
  // PDEC_E_INVALID -- a child without a stream of its own would serialise behind another)
  int set_part_streams(const hipStream_t* s, int n) override {
    if (nparts < 2) return PDEC_OK;
    PDEC_REQUIRE(n >= nparts - 1, "pdec_env_set_part_streams: this environment runs %d parts and needs %d streams, got %d", nparts,
                 nparts - 1, n);
    PDEC_HIP(ps.give(s, nparts - 1));
    return PDEC_OK;
  }
  ~FluidEnv() override {
    for (int i = 0; i < MAXPART; ++i) half[i].reset();
  }
};

// does the persistent x-pass (fluid_k2p_kernel) serve this environment?  Decided once: K1 writes W in the layout K2 reads.
static bool k2p_eligible(const FluidEnv& E) {
  static const char* env = getenv("PDEC_FLUID_K2P");
  const int npw = ((E.nl + 7) / 8 + 7) / 8, nsu = E.n * 8 / 512;
  const bool want = env ? env[0] == '1' : E.n >= 256;
  const bool wave64 = E.wave_E != 0 && E.wave_LB == 6;        // the one-line-per-wave plans (wave-FFT kernels K1w / K2p / K3w)
  return want && wave64 && E.p % 8 == 0 && E.cfg.ifpad && E.n * 8 % 512 == 0 && ((npw == 9 && nsu == 8) || (npw == 5 && nsu == 4));
}

static FluidDev<double> fluid_dev(const FluidEnv& E) {
  FluidDev<double> d;
  d.wtile = k2p_eligible(E) ? 1 : 0;
  d.B = E.cfg.B; d.n = E.n; d.p = E.p; d.nl = E.nl; d.TL = E.TL; d.TLn = E.TLn;
  d.LS = E.p + 2; d.LSn = E.n + 2;
  d.nu = E.cfg.nu;
  const double inv = 1.0 / ((double)E.p * E.p);
  d.inv2 = inv * inv;
  d.scale_out = E.cfg.ifpad ? 1.5 * 1.5 : 1.0;        // src/fluid_rk4.jl:176
  d.invn2 = 1.0 / ((double)E.n * E.n);
  d.k = E.k.as<double>(); d.twp = E.twp.as<C2<double>>(); d.twn = E.twn.as<C2<double>>();
  d.plp = E.plp; d.pln = E.pln;
  return d;
}

static int pick_tile(int len) {
  int t = 16;
  while (t > 1 && t * len > FL_NTH * FL_MAXE) t >>= 1;
  return t;
}

template <class K>
static int set_lds(K kern, size_t bytes) {
  PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return PDEC_OK;
}

static int fluid_set_attrs(const FluidEnv& E) {
  int rc;
  if ((rc = set_lds(fluid_k1_kernel<double>, E.lds_p))) return rc;
  if ((rc = set_lds(fluid_k2_kernel<double>, E.lds_p))) return rc;
  if ((rc = set_lds(fluid_k3_kernel<double>, E.lds_p))) return rc;
  if ((rc = set_lds(fluid_fft_fast_kernel<double, +1, false>, E.lds_n))) return rc;
  if ((rc = set_lds(fluid_fft_fast_kernel<double, -1, true>, E.lds_n))) return rc;
  if ((rc = set_lds(fluid_fft_slow_kernel<double, +1, true>, E.lds_n))) return rc;
  if ((rc = set_lds(fluid_fft_slow_kernel<double, -1, false>, E.lds_n))) return rc;
  return PDEC_OK;
}

// K2 of the wave path: the persistent pipelined form (fluid_k2p_kernel) where it is built -- one line per wave, p a multiple of 8,
// the DMA piece count instantiated -- else the tile form.  PDEC_FLUID_K2P=0 / 1 forces the choice.
template <int E, int Q, int LB>
static int fluid_k2_launch(FluidEnv& Ev, const FluidDev<double>& d) {
  typedef C2<double> Z;
  const int B = Ev.cfg.B, p = Ev.p;
  constexpr int LPW = 64 >> LB;
  if constexpr (LB == 6) {
    const int npw = ((Ev.nl + 7) / 8 + 7) / 8;
    // instantiated: n = 512 (nl = 513 -> 9 DMA pieces per wave, 8 output elements per thread) and n = 256 (257 -> 5, 4)
    if (d.wtile) {
      static int ncu = 0;
      if (!ncu) {
        int dev = 0;
        hipDeviceProp_t pr;
        PDEC_HIP(hipGetDevice(&dev));
        PDEC_HIP(hipGetDeviceProperties(&pr, dev));
        ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
      }
      const int ntiles = B * (p / 8);
      const size_t lds = ((size_t)((Ev.nl + 7) / 8) * 64 + (size_t)Ev.n * 8 + (size_t)(Q > 1 ? (Q - 1) * E : 1) * 64) * 16;
      const int grid = ntiles < ncu ? ntiles : ncu;
#define PDEC_K2P(NPW, NSU)                                                                                                     \
  {                                                                                                                            \
    static bool attr = false;                                                                                                  \
    if (!attr) {                                                                                                               \
      PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k2p_kernel<E, Q, NPW, NSU>),                            \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                   \
      attr = true;                                                                                                             \
    }                                                                                                                          \
    hipLaunchKernelGGL((fluid_k2p_kernel<E, Q, NPW, NSU>), dim3(grid), dim3(512), lds, Ev.stream, d, Ev.W.as<Z>(),             \
                       Ev.W2.as<Z>(), ntiles);                                                                                 \
  }
      if (npw == 9) PDEC_K2P(9, 8) else PDEC_K2P(5, 4)
#undef PDEC_K2P
      return PDEC_OK;
    }
  }
  hipLaunchKernelGGL((fluid_k2w_kernel<E, Q, FL_K2_TC, LB>), dim3((p + FL_K2_TC - 1) / FL_K2_TC, B), dim3(64 * FL_K2_TC / LPW),
                     (size_t)FL_K2_TC * (p + 1) * 16, Ev.stream, d, Ev.W.as<Z>(), Ev.W2.as<Z>());
  return PDEC_OK;
}

// one rhs evaluation fused with an RK4 stage update (mode as in fluid_k3_kernel)
template <int E, int Q, int LB>
static int fluid_rhs_launch_wave(FluidEnv& Ev, const FluidDev<double>& d, const void* omg_s, const void* phat, const void* f0,
                                 void* acc, void* out, int mode, double ca, double cb) {
  typedef C2<double> Z;
  const int B = Ev.cfg.B, n = Ev.n;
  constexpr int LPW = 64 >> LB, LPB = 4 * LPW;               // line slots per wave / per 256-thread workgroup
  static bool attr = false;
  if (!attr) {
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k1w_kernel<E, Q, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k2w_kernel<E, Q, FL_K2_TC, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k3w_kernel<E, Q, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  const int pair = n >= 256 ? 1 : 0;      // a line and its mirror per wave (K1w) once there are enough lines to fill the chip
  {
    ProfScope ps(&Ev, "fluid_k1", true);
    for (int r = 0; r < ps.reps; ++r)
      hipLaunchKernelGGL((fluid_k1w_kernel<E, Q, LB>), dim3(((pair ? n / 2 + 1 : Ev.nl) + LPB - 1) / LPB, B), dim3(256),
                         (size_t)LPB * 2 * n * 16 + (size_t)n * 8, Ev.stream, d, (const Z*)omg_s, Ev.W.as<Z>(), pair);
  }
  {
    ProfScope ps(&Ev, "fluid_k2", true);
    for (int r = 0; r < ps.reps; ++r) {
      const int rc = fluid_k2_launch<E, Q, LB>(Ev, d);
      if (rc) return rc;
    }
  }
  {
    ProfScope ps(&Ev, "fluid_k3", mode == 0);
    for (int r = 0; r < ps.reps; ++r)
      hipLaunchKernelGGL((fluid_k3w_kernel<E, Q, LB>), dim3((n + LPB - 1) / LPB, B), dim3(256), (size_t)LPB * n * 16, Ev.stream, d,
                         Ev.W2.as<Z>(), (const Z*)omg_s, (const Z*)phat, (const Z*)f0, (Z*)acc, (Z*)out, mode, ca, cb);
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static int fluid_rhs_launch(FluidEnv& E, const void* omg_s, const void* phat, const void* f0, void* acc, void* out,
                            int mode, double ca, double cb) {
  typedef C2<double> Z;
  const FluidDev<double> d = fluid_dev(E);
  const int B = E.cfg.B;
  if (E.wave_E == 4 && E.wave_Q == 3) return fluid_rhs_launch_wave<4, 3, 6>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 4 && E.wave_Q == 2) return fluid_rhs_launch_wave<4, 2, 6>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 4 && E.wave_Q == 1) return fluid_rhs_launch_wave<4, 1, 6>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 2 && E.wave_Q == 3 && E.wave_LB == 6) return fluid_rhs_launch_wave<2, 3, 6>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 2 && E.wave_Q == 1 && E.wave_LB == 6) return fluid_rhs_launch_wave<2, 1, 6>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 2 && E.wave_Q == 3 && E.wave_LB == 5) return fluid_rhs_launch_wave<2, 3, 5>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  if (E.wave_E == 2 && E.wave_Q == 1 && E.wave_LB == 5) return fluid_rhs_launch_wave<2, 1, 5>(E, d, omg_s, phat, f0, acc, out, mode, ca, cb);
  {
    ProfScope ps(&E, "fluid_k1", true);
    for (int r = 0; r < ps.reps; ++r)
      hipLaunchKernelGGL(fluid_k1_kernel<double>, dim3((E.nl + E.TL - 1) / E.TL, B), dim3(FL_NTH), E.lds_p, E.stream, d,
                         (const Z*)omg_s, E.W.as<Z>());
  }
  {
    ProfScope ps(&E, "fluid_k2", true);
    for (int r = 0; r < ps.reps; ++r)
      hipLaunchKernelGGL(fluid_k2_kernel<double>, dim3((E.p + E.TL - 1) / E.TL, B), dim3(FL_NTH), E.lds_p, E.stream, d,
                         E.W.as<Z>(), E.W2.as<Z>());
  }
  {
    ProfScope ps(&E, "fluid_k3", mode == 0);
    for (int r = 0; r < ps.reps; ++r)
      hipLaunchKernelGGL(fluid_k3_kernel<double>, dim3((E.n + E.TL - 1) / E.TL, B), dim3(FL_NTH), E.lds_p, E.stream, d,
                         E.W2.as<Z>(), (const Z*)omg_s, (const Z*)phat, (const Z*)f0, (Z*)acc, (Z*)out, mode, ca, cb);
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// do_step (FluidSetup.jl:163-172): K sub-steps of rk4 (src/fluid_rk4.jl:122-132), in place on f
// RK4 sub-steps with K3 of every stage fused with K1 of the next right-hand side (fluid_k31w_kernel): K1 once, then
// K2 + K31 per stage, a plain K3 at the very end.  Wave-transform path with line pairs (n >= 256).
template <int E, int Q, int LB>
static int fluid_integrate_wave(FluidEnv& Ev, const FluidDev<double>& d, void* f, const void* phat) {
  typedef C2<double> Z;
  const int B = Ev.cfg.B, n = Ev.n;
  constexpr int LPW = 64 >> LB, LPB = 4 * LPW;
  static bool attr = false;
  if (!attr) {
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k1w_kernel<E, Q, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k2w_kernel<E, Q, FL_K2_TC, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k3w_kernel<E, Q, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    PDEC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fluid_k31w_kernel<E, Q, LB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  const double h = Ev.cfg.dt / Ev.cfg.K;
  Z *fs = Ev.fs.as<Z>(), *acc = Ev.acc.as<Z>(), *fz = (Z*)f;
  const dim3 gpair((n / 2 + 1 + LPB - 1) / LPB, B);
  const size_t lds2 = (size_t)LPB * 2 * n * 16;
  auto k2 = [&]() {
    ProfScope ps(&Ev, "fluid_k2");
    (void)fluid_k2_launch<E, Q, LB>(Ev, d);
  };
  auto k31 = [&](const Z* omg_s, Z* out, int mode, double ca, double cb) {
    ProfScope ps(&Ev, "fluid_k31");
    hipLaunchKernelGGL((fluid_k31w_kernel<E, Q, LB>), gpair, dim3(256), lds2, Ev.stream, d, Ev.W2.as<Z>(), omg_s, (const Z*)phat,
                       (const Z*)fz, acc, out, mode, ca, cb, Ev.W.as<Z>());
  };
  {
    ProfScope ps(&Ev, "fluid_k1");
    hipLaunchKernelGGL((fluid_k1w_kernel<E, Q, LB>), gpair, dim3(256), lds2 + (size_t)n * 8, Ev.stream, d, (const Z*)fz, Ev.W.as<Z>(), 1);
  }
  for (int it = 0; it < Ev.cfg.K; ++it) {
    k2(); k31(fz, fs, 1, 0.5 * h, h / 6.0);
    k2(); k31(fs, fs, 2, 0.5 * h, h / 3.0);
    k2(); k31(fs, fs, 2, h, h / 3.0);
    k2();
    if (it + 1 < Ev.cfg.K) {
      k31(fs, fz, 4, 0.0, h / 6.0);
    } else {
      ProfScope ps(&Ev, "fluid_k3");
      hipLaunchKernelGGL((fluid_k3w_kernel<E, Q, LB>), dim3((n + LPB - 1) / LPB, B), dim3(256), (size_t)LPB * n * 16, Ev.stream, d,
                         Ev.W2.as<Z>(), (const Z*)fs, (const Z*)phat, (const Z*)fz, acc, fz, 4, 0.0, h / 6.0);
    }
  }
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// (A HIP graph of the whole sub-step loop -- 8 K = 320 kernel nodes at the reference's own shape, one trajectory on the 128^2
// grid -- was measured and dropped: 232 against 260 env-steps/s eager.  That loop is not launch-bound: its kernels run
// 17-48 workgroups of ~10 us each on 256 CUs, HISTORY.md.)
static int fluid_integrate(FluidEnv& E, void* f, const void* phat) {
  // measured (B = 16): n = 256: 650 -> 701 env-steps/s fused; n = 512: 76.6 -> 74.5 (the fused waves run six transforms
  // each and the stage's streaming phase no longer overlaps other waves' transforms) -> fused below 512 only
  static const char* fuse_env = getenv("PDEC_FLUID_FUSE");     // 1 / 0 force it on / off
  const bool fuse = fuse_env ? fuse_env[0] == '1' : (E.n >= 256 && E.n < 512);
  if (fuse && E.n >= 256) {
    const FluidDev<double> d = fluid_dev(E);
    if (E.wave_E == 4 && E.wave_Q == 3) return fluid_integrate_wave<4, 3, 6>(E, d, f, phat);
    if (E.wave_E == 4 && E.wave_Q == 2) return fluid_integrate_wave<4, 2, 6>(E, d, f, phat);
    if (E.wave_E == 4 && E.wave_Q == 1) return fluid_integrate_wave<4, 1, 6>(E, d, f, phat);
    if (E.wave_E == 2 && E.wave_Q == 3 && E.wave_LB == 6) return fluid_integrate_wave<2, 3, 6>(E, d, f, phat);
    if (E.wave_E == 2 && E.wave_Q == 1 && E.wave_LB == 6) return fluid_integrate_wave<2, 1, 6>(E, d, f, phat);
  }
  const double h = E.cfg.dt / E.cfg.K;
  void *fs = E.fs.p, *acc = E.acc.p;
  int rc;
  for (int it = 0; it < E.cfg.K; ++it) {
    if ((rc = fluid_rhs_launch(E, f, phat, f, acc, fs, 1, 0.5 * h, h / 6.0))) return rc;
    if ((rc = fluid_rhs_launch(E, fs, phat, f, acc, fs, 2, 0.5 * h, h / 3.0))) return rc;
    if ((rc = fluid_rhs_launch(E, fs, phat, f, acc, fs, 2, h, h / 3.0))) return rc;
    if ((rc = fluid_rhs_launch(E, fs, phat, f, acc, f, 4, 0.0, h / 6.0))) return rc;
  }
  return PDEC_OK;
}

// yreal = real(ifft(y))
static int fluid_to_physical(FluidEnv& E, const void* y) {
  typedef C2<double> Z;
  const FluidDev<double> d = fluid_dev(E);
  const int B = E.cfg.B, gt = (E.n + E.TLn - 1) / E.TLn;
  ProfScope ps(&E, "fluid_ifft2");
  hipLaunchKernelGGL((fluid_fft_fast_kernel<double, +1, false>), dim3(gt, B), dim3(FL_NTH), E.lds_n, E.stream, d, y,
                     E.tmpc.as<Z>());
  hipLaunchKernelGGL((fluid_fft_slow_kernel<double, +1, true>), dim3(gt, B), dim3(FL_NTH), E.lds_n, E.stream, d,
                     E.tmpc.as<Z>(), E.yreal.p, d.invn2);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static int fluid_dots(FluidEnv& E, const void* y) {
  int rc = fluid_to_physical(E, y);
  if (rc) return rc;
  const pdec_env_cfg& c = E.cfg;
  ProfScope ps(&E, "fluid_dots");
  hipLaunchKernelGGL(fluid_dots_kernel<double>, dim3((c.S + 3) / 4, c.B), dim3(256), 0, E.stream, E.n, c.S, E.BH, E.BW,
                     E.sbox.as<double>(), E.sorg.as<int>(), E.yreal.as<double>(), E.dots.as<double>());
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static FeatArgs<double> feat_args(const FluidEnv& E) {
  const pdec_env_cfg& c = E.cfg;
  FeatArgs<double> g;
  g.S = c.S; g.A = c.A; g.spa = c.sensors_per_axis; g.window = c.window;
  g.ns = env_ns(c); g.check_max = c.check_max_value;
  g.mem = c.memory_size; g.na = env_na(c);
  g.sensor_scale = c.sensor_scale; g.r_in_scale = c.reward_in_scale; g.r_power = c.reward_power;
  g.r_denom = c.reward_denom; g.a_pun = c.action_punish; g.da_pun = c.delta_action_punish; g.max_value = c.max_value;
  g.a2s = E.a2s_d.as<int>();
  return g;
}

static int fluid_feat_launch(FluidEnv& E, const void* action, const void* action_prev, const void* state_prev,
                             void* state_out, void* reward_out, int32_t* done) {
  ProfScope ps(&E, "fluid_feat");
  hipLaunchKernelGGL(fluid_feat_kernel<double>, dim3(E.cfg.B), dim3(256), 0, E.stream, feat_args(E), E.dots.as<double>(),
                     (const double*)action, (const double*)action_prev, (const double*)state_prev, (double*)state_out,
                     (double*)reward_out, done);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

static FluidEnv& as_fluid(Env& E) { return static_cast<FluidEnv&>(E); }

int fluid_actuate(Env& E0, const void* action, void* p_out) {
  typedef C2<double> Z;
  FluidEnv& E = as_fluid(E0);
  const pdec_env_cfg& c = E.cfg;
  const FluidDev<double> d = fluid_dev(E);
  const int gt = (E.n + E.TLn - 1) / E.TLn;
  ProfScope ps(&E, "fluid_actuate");
  hipLaunchKernelGGL(fluid_actuate_kernel<double>, dim3(E.nb1 * E.nb1, c.B), dim3(256), 0, E.stream, E.n, c.A, E.BH, E.BW,
                     E.nb1, E.abox.as<double>(), E.aorg.as<int>(), E.blkptr.as<int>(), E.blkidx.as<int>(),
                     (const double*)action, env_na(c), c.agent_power, E.yreal.as<double>());
  hipLaunchKernelGGL((fluid_fft_fast_kernel<double, -1, true>), dim3(gt, c.B), dim3(FL_NTH), E.lds_n, E.stream, d,
                     E.yreal.p, E.tmpc.as<Z>());
  hipLaunchKernelGGL((fluid_fft_slow_kernel<double, -1, false>), dim3(gt, c.B), dim3(FL_NTH), E.lds_n, E.stream, d,
                     E.tmpc.as<Z>(), p_out, 1.0);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int fluid_featurize(Env& E0, const void* y, const void* state_prev, void* state_out, const void* action) {
  FluidEnv& E = as_fluid(E0);
  int rc = fluid_dots(E, y);
  if (rc) return rc;
  return fluid_feat_launch(E, action, nullptr, state_prev, state_out, nullptr, nullptr);   // (action: the memory rows only)
}

int fluid_reward(Env& E0, const void* y, const void* action, const void* action_prev, void* r_out) {
  FluidEnv& E = as_fluid(E0);
  int rc = fluid_dots(E, y);
  if (rc) return rc;
  return fluid_feat_launch(E, action, action_prev, nullptr, nullptr, r_out, nullptr);
}

int fluid_rhs_eval(Env& E0, const void* y, const void* p, void* out) {
  FluidEnv& E = as_fluid(E0);
  return fluid_rhs_launch(E, y, p, nullptr, nullptr, out, 0, 0.0, 0.0);
}

static int fluid_done_y(FluidEnv& E, const void* y, int32_t* done) {
  hipLaunchKernelGGL(fluid_maxabs_kernel<double>, dim3(E.cfg.B), dim3(256), 0, E.stream, E.n * E.n, E.cfg.max_value,
                     (const C2<double>*)y, done);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int fluid_pde_step(Env& E0, const void* y_in, const void* p, void* y_out, int32_t* done) {
  FluidEnv& E = as_fluid(E0);
  const size_t bytes = (size_t)E.cfg.B * E.n * E.n * 16;
  if (y_out != y_in) PDEC_HIP(hipMemcpyAsync(y_out, y_in, bytes, hipMemcpyDeviceToDevice, E.stream));
  int rc = fluid_integrate(E, y_out, p);
  if (rc) return rc;
  if (done) {
    if (E.cfg.check_max_value == 1) return fluid_done_y(E, y_out, done);
    PDEC_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * E.cfg.B, E.stream));
  }
  return PDEC_OK;
}

int fluid_env_step(Env& E0, const void* y_in, const void* action, const void* action_prev, const void* state_prev,
                   void* y_out, void* p_out, void* state_out, void* reward_out, int32_t* done) {
  FluidEnv& E = as_fluid(E0);
  if (E.nparts >= 2 && !E.prof) {
    // the parts of the batch side by side: every argument is batch-major, so a part is a pointer offset.  (Per-kernel
    // timing passes, pdec_prof_enable, take the whole batch on one stream.)
    const size_t nn = (size_t)E.n * E.n, A = E.cfg.A, ns = (size_t)env_ns(E.cfg), na = (size_t)env_na(E.cfg);
    auto off = [](const void* p, size_t bytes) -> const void* { return p ? (const char*)p + bytes : nullptr; };
    auto offm = [](void* p, size_t bytes) -> void* { return p ? (char*)p + bytes : nullptr; };
    {
      bool refused = false;
      PDEC_HIP(E.ps.ensure(E.stream, E.nparts, &refused));
      PDEC_REQUIRE(!refused, "fluid step: its %d part streams do not exist yet and cannot be made while the environment's stream is "
                   "being captured; run one step (or pdec_env_set_part_streams) before the capture, or PDEC_FLUID_SPLIT=0", E.nparts - 1);
    }
    {   // a fork that fails midway has already made some part streams wait: join them before returning
      const hipError_t ef = E.ps.fork(E.stream, E.nparts);
      if (ef != hipSuccess) { (void)E.ps.join(E.stream, E.nparts); PDEC_HIP(ef); }
    }
    int b0 = 0, rc_part = PDEC_OK;
    for (int hh = 0; hh < E.nparts; ++hh) {
      FluidEnv& H = *E.half[hh];
      H.stream = hh == 0 ? E.stream : E.ps.st[hh];
      H.term_out = E.term_out ? (char*)E.term_out + (size_t)b0 * A * 8 : nullptr;
      const int rc = fluid_env_step(H, off(y_in, b0 * nn * 16), off(action, b0 * A * na * 8), off(action_prev, b0 * A * na * 8),
                                    off(state_prev, b0 * A * ns * 8), offm(y_out, b0 * nn * 16), offm(p_out, b0 * nn * 16),
                                    offm(state_out, b0 * A * ns * 8), offm(reward_out, b0 * A * 8), done ? done + b0 : nullptr);
      if (rc) { rc_part = rc; break; }              // (the part streams are joined below on this path too)
      b0 += H.cfg.B;
    }
    const hipError_t ej = E.ps.join(E.stream, E.nparts);
    if (rc_part) return rc_part;
    PDEC_HIP(ej);
    return PDEC_OK;
  }
  void* ph = p_out ? p_out : E.phat.p;
  int rc;
  if ((rc = fluid_actuate(E, action, ph))) return rc;                                   // src/PDEenv.jl:199
  if ((rc = fluid_pde_step(E, y_in, ph, y_out, nullptr))) return rc;                    // :216-218
  if ((rc = fluid_dots(E, y_out))) return rc;
  if (done && E.cfg.check_max_value != 2) {
    if (E.cfg.check_max_value == 1) { if ((rc = fluid_done_y(E, y_out, done))) return rc; }
    else PDEC_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * E.cfg.B, E.stream));
  }
  return fluid_feat_launch(E, action, action_prev, state_prev, state_out, reward_out, done);   // :220-222
}

}  // namespace pdec

using namespace pdec;

static int fluid_make(std::unique_ptr<FluidEnv>& out, const pdec_env_cfg& c, int BH, int BW, const double* sensor_boxes,
                      const int32_t* sensor_origin, const double* actuator_boxes, const int32_t* actuator_origin,
                      const int32_t* a2s) {
  PDEC_REQUIRE(c.pde_kind == PDEC_PDE_FLUID_RK4, "pdec_fluid_env_create: pde_kind must be PDEC_PDE_FLUID_RK4");
  PDEC_REQUIRE(c.dtype == PDEC_F64, "the fluid path computes in fp64 (ComplexF64 in the reference)");
  const int n = c.N;
  PDEC_REQUIRE(c.B >= 1 && n >= 8 && n % 4 == 0 && c.K >= 1, "pdec_fluid_env_create: bad sizes B=%d N=%d K=%d", c.B, n, c.K);
  PDEC_REQUIRE(c.S >= 1 && c.A >= 1 && c.sensors_per_axis >= 1 && c.sensors_per_axis * c.sensors_per_axis == c.S,
               "pdec_fluid_env_create: S must equal sensors_per_axis^2");
  PDEC_REQUIRE(c.window >= 1 && (c.window & 1) && c.window <= c.sensors_per_axis && c.temporal_steps >= 1 && !c.mono,
               "pdec_fluid_env_create: window must be odd and <= sensors_per_axis; no mono variant");
  PDEC_REQUIRE(BH >= 1 && BW >= 1 && BH <= n && BW <= n, "pdec_fluid_env_create: bad box %dx%d", BH, BW);
  PDEC_REQUIRE(c.memory_size >= 0 && c.memory_size <= 64, "pdec_fluid_env_create: memory_size %d out of range", c.memory_size);
  PDEC_REQUIRE(c.Lx > 0 && c.dt > 0, "pdec_fluid_env_create: Lx and dt must be positive");
  for (int a = 0; a < c.A; ++a) PDEC_REQUIRE(a2s[a] >= 0 && a2s[a] < c.S, "pdec_fluid_env_create: a2s[%d] out of range", a);
  for (int s = 0; s < c.S; ++s)
    PDEC_REQUIRE(sensor_origin[2 * s] >= 0 && sensor_origin[2 * s] < n && sensor_origin[2 * s + 1] >= 0 &&
                     sensor_origin[2 * s + 1] < n, "pdec_fluid_env_create: sensor origin %d out of range", s);
  for (int a = 0; a < c.A; ++a)
    PDEC_REQUIRE(actuator_origin[2 * a] >= 0 && actuator_origin[2 * a] < n && actuator_origin[2 * a + 1] >= 0 &&
                     actuator_origin[2 * a + 1] < n, "pdec_fluid_env_create: actuator origin %d out of range", a);
  auto E = std::make_unique<FluidEnv>();
  E->cfg = c;
  E->n = n;
  E->p = c.ifpad ? n * 3 / 2 : n;
  E->nl = c.ifpad ? n + 1 : n;
  E->BH = BH; E->BW = BW;
  PDEC_REQUIRE(E->p <= 1536, "N=%d too large for the in-LDS line FFTs (max 1024 padded / 1536 un-padded)", n);
  PDEC_REQUIRE(make_fft_plan(E->p, E->plp) && make_fft_plan(n, E->pln), "N=%d: sizes must factor into 2,3,5", n);
  if (!getenv("PDEC_FLUID_LDS_FFT")) {     // one-line-per-wave register transforms for the supported padded lengths
    switch (E->p) {
      case 768: E->wave_E = 4; E->wave_Q = 3; break;
      case 512: E->wave_E = 4; E->wave_Q = 2; break;
      case 256: E->wave_E = 4; E->wave_Q = 1; break;
      case 384: E->wave_E = 2; E->wave_Q = 3; break;
      case 128: E->wave_E = 2; E->wave_Q = 1; break;
      case 192: E->wave_E = 2; E->wave_Q = 3; E->wave_LB = 5; break;     // the reference's 128^2 training grid, padded
      case 64: E->wave_E = 2; E->wave_Q = 1; E->wave_LB = 5; break;
      default: break;
    }
  }
  E->TL = pick_tile(E->p);
  E->TLn = pick_tile(n);
  PDEC_REQUIRE(E->TL >= 2, "internal: tile too small");
  E->lds_p = ((size_t)2 * E->TL * (E->p + 2) + E->p) * 16;
  E->lds_n = ((size_t)2 * E->TLn * (n + 2) + n) * 16;
  PDEC_REQUIRE(E->lds_p <= 160 * 1024 && E->lds_n <= 160 * 1024, "fluid kernels need too much LDS");
  std::vector<double> k(n), twp(2 * (size_t)E->p), twn(2 * (size_t)n);
  for (int i = 0; i < n; ++i) k[i] = (i <= n / 2 ? i : i - n) / c.Lx * 2 * M_PI;     // FluidSetup.jl:106
  for (int m = 0; m < E->p; ++m) { twp[2 * m] = cos(2 * M_PI * m / E->p); twp[2 * m + 1] = -sin(2 * M_PI * m / E->p); }
  for (int m = 0; m < n; ++m) { twn[2 * m] = cos(2 * M_PI * m / n); twn[2 * m + 1] = -sin(2 * M_PI * m / n); }
  int rc;
  if ((rc = upload_converted(E->k, k.data(), n, PDEC_F64))) return rc;
  if ((rc = upload_converted(E->twp, twp.data(), twp.size(), PDEC_F64))) return rc;
  if ((rc = upload_converted(E->twn, twn.data(), twn.size(), PDEC_F64))) return rc;
  if ((rc = upload_converted(E->sbox, sensor_boxes, (size_t)c.S * BH * BW, PDEC_F64))) return rc;
  if ((rc = upload_converted(E->abox, actuator_boxes, (size_t)c.A * BH * BW, PDEC_F64))) return rc;
  auto up_i = [](DevBuf& b, const int32_t* src, size_t cnt) -> int {
    PDEC_HIP(b.alloc(sizeof(int32_t) * (cnt ? cnt : 1)));
    if (cnt) PDEC_HIP(hipMemcpy(b.p, src, sizeof(int32_t) * cnt, hipMemcpyHostToDevice));
    return PDEC_OK;
  };
  if ((rc = up_i(E->sorg, sensor_origin, 2 * (size_t)c.S))) return rc;
  if ((rc = up_i(E->aorg, actuator_origin, 2 * (size_t)c.A))) return rc;
  if ((rc = up_i(E->a2s_d, a2s, c.A))) return rc;
  // candidate actuators per 16x16 block of cells
  E->nb1 = (n + 15) / 16;
  std::vector<int32_t> ptr(1, 0), idx;
  auto meets = [&](int o, int len, int c0) {   // ring interval [o, o+len) meets cells [c0, c0+16)
    for (int q = 0; q < 16 && c0 + q < n; ++q) {
      int dlt = c0 + q - o;
      if (dlt < 0) dlt += n;
      if (dlt < len) return true;
    }
    return false;
  };
  for (int bj = 0; bj < E->nb1; ++bj)
    for (int bi = 0; bi < E->nb1; ++bi) {
      for (int a = 0; a < c.A; ++a)
        if (meets(actuator_origin[2 * a], BW, bj * 16) && meets(actuator_origin[2 * a + 1], BH, bi * 16)) idx.push_back(a);
      ptr.push_back((int32_t)idx.size());
    }
  if ((rc = up_i(E->blkptr, ptr.data(), ptr.size()))) return rc;
  if ((rc = up_i(E->blkidx, idx.data(), idx.size()))) return rc;
  const size_t nn = (size_t)n * n, Bz = c.B;
  PDEC_HIP(E->W.alloc(Bz * 2 * E->nl * E->p * 16));
  PDEC_HIP(E->W2.alloc(Bz * n * E->p * 16));
  PDEC_HIP(E->fs.alloc(Bz * nn * 16));
  PDEC_HIP(E->acc.alloc(Bz * nn * 16));
  PDEC_HIP(E->tmpc.alloc(Bz * nn * 16));
  PDEC_HIP(E->phat.alloc(Bz * nn * 16));
  PDEC_HIP(E->yreal.alloc(Bz * nn * 8));
  PDEC_HIP(E->dots.alloc(Bz * c.S * 8));
  if ((rc = fluid_set_attrs(*E))) return rc;
  out = std::move(E);
  return PDEC_OK;
}

extern "C" int pdec_fluid_env_create(pdec_handle* h, const pdec_env_cfg* cfg, int BH, int BW, const double* sensor_boxes,
                                     const int32_t* sensor_origin, const double* actuator_boxes,
                                     const int32_t* actuator_origin, const int32_t* a2s) {
  PDEC_REQUIRE(h && cfg && sensor_boxes && sensor_origin && actuator_boxes && actuator_origin && a2s,
               "pdec_fluid_env_create: null argument");
  std::unique_ptr<FluidEnv> E;
  int rc = fluid_make(E, *cfg, BH, BW, sensor_boxes, sensor_origin, actuator_boxes, actuator_origin, a2s);
  if (rc) return rc;
  // part-batch children for the fused env step where a part still fills the chip (padded 512-point grids and up): two by
  // default; PDEC_FLUID_SPLIT=0 off, 1 / 2 two parts, 3 / 4 that many
  static const char* sp = getenv("PDEC_FLUID_SPLIT");
  int np = sp ? atoi(sp) : ((cfg->N >= 512 && cfg->B >= 8) ? 2 : 0);
  if (np == 1) np = 2;                                  // (PDEC_FLUID_SPLIT=1: on = two parts; 0: off; 3, 4: that many parts)
  np = std::min(std::min(np, (int)FluidEnv::MAXPART), cfg->B);
  if (np >= 2) {
    pdec_env_cfg ch = *cfg;
    int left = cfg->B;
    for (int i = 0; i < np; ++i) {
      ch.B = left / (np - i);
      left -= ch.B;
      if ((rc = fluid_make(E->half[i], ch, BH, BW, sensor_boxes, sensor_origin, actuator_boxes, actuator_origin, a2s))) return rc;
    }
    E->nparts = np;                                    // (streams and events: PartStreams::ensure at the first split step)
  }
  *h = register_object(std::move(E));
  return PDEC_OK;
}

// Unit-test entry for the register-resident wave FFT (wave_fft.hpp): nlines lines of `len` complex doubles, natural
// order in and out, unnormalised forward (sgn < 0) or inverse (sgn > 0).  len in {64, 128, 192, 256, 384, 512, 768}.
extern "C" int pdec_debug_wave_fft(const void* in_dev, void* out_dev, int len, int nlines, int sgn) {
  PDEC_REQUIRE(in_dev && out_dev && nlines >= 1, "pdec_debug_wave_fft: null/empty");
  std::vector<double> tw(2 * (size_t)len);
  for (int m = 0; m < len; ++m) { tw[2 * m] = cos(2 * M_PI * m / len); tw[2 * m + 1] = -sin(2 * M_PI * m / len); }
  DevBuf d;
  int rc = upload_converted(d, tw.data(), tw.size(), PDEC_F64);
  if (rc) return rc;
  const dim3 block(256);
  typedef const C2<double>* CI;
  typedef C2<double>* CO;
#define WFD(E, Q, LB) hipLaunchKernelGGL((wave_fft_debug_kernel<E, Q, LB>), dim3((nlines + 4 * (64 >> LB) - 1) / (4 * (64 >> LB))), block, 0, 0, (CI)in_dev, (CO)out_dev, d.as<C2<double>>(), nlines, sgn)
  if (len == 768) WFD(4, 3, 6);
  else if (len == 512) WFD(4, 2, 6);
  else if (len == 256) WFD(4, 1, 6);
  else if (len == 384) WFD(2, 3, 6);
  else if (len == 128) WFD(2, 1, 6);
  else if (len == 192) WFD(2, 3, 5);
  else if (len == 64) WFD(2, 1, 5);
  else { set_error("pdec_debug_wave_fft: unsupported length %d", len); return PDEC_E_INVALID; }
#undef WFD
  PDEC_HIP(hipGetLastError());
  PDEC_HIP(hipDeviceSynchronize());
  return PDEC_OK;
}

// ic(caseno): spectrum of the sum of Taylor vortices (src/fluid_rk4.jl:72-120 with the random draws made by the
// caller).  vortices: HOST array [B][nv][4] of (x0, y0, a0, U_max); y_out: device [B][nx][ny][2].
extern "C" int pdec_fluid_ic(pdec_handle h, const double* vortices, int nv, void* y_out) {
  Env* E0 = lookup_as<Env>(h, Kind::Env);
  if (!E0 || E0->cfg.pde_kind != PDEC_PDE_FLUID_RK4) { set_error("pdec_fluid_ic: not a fluid env handle"); return PDEC_E_HANDLE; }
  PDEC_REQUIRE(vortices && y_out && nv >= 1 && nv <= 1024, "pdec_fluid_ic: bad arguments (1 <= nv <= 1024)");
  typedef C2<double> Z;
  FluidEnv& E = static_cast<FluidEnv&>(*E0);
  const pdec_env_cfg& c = E.cfg;
  const size_t vb = (size_t)c.B * nv * 4 * sizeof(double);
  if (E.icv.bytes < vb) PDEC_HIP(E.icv.alloc(vb));
  PDEC_HIP(hipMemcpyAsync(E.icv.p, vortices, vb, hipMemcpyHostToDevice, E.stream));
  const FluidDev<double> d = fluid_dev(E);
  const int gt = (E.n + E.TLn - 1) / E.TLn;
  ProfScope ps(&E, "fluid_ic");
  hipLaunchKernelGGL(fluid_ic_kernel<double>, dim3((E.n * E.n + 255) / 256, c.B), dim3(256), (size_t)nv * 4 * sizeof(double), E.stream,
                     E.n, nv, c.Lx, c.Lx, E.icv.as<double>(), E.yreal.as<double>());
  hipLaunchKernelGGL((fluid_fft_fast_kernel<double, -1, true>), dim3(gt, c.B), dim3(FL_NTH), E.lds_n, E.stream, d, E.yreal.p,
                     E.tmpc.as<Z>());
  hipLaunchKernelGGL((fluid_fft_slow_kernel<double, -1, false>), dim3(gt, c.B), dim3(FL_NTH), E.lds_n, E.stream, d,
                     E.tmpc.as<Z>(), y_out, 1.0);
  PDEC_HIP(hipGetLastError());
  PDEC_HIP(hipStreamSynchronize(E.stream));      // the host array may be reused by the caller
  return PDEC_OK;
}
