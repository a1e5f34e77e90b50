// fft_lds.hpp -- in-LDS mixed-radix (2,3,4,5) Stockham auto-sort FFT for one workgroup.
// Unnormalised forward (e^{-2 pi i jk/N}) and unnormalised inverse (caller scales by 1/N),
// i.e. FFTW's conventions, which the reference's constants depend on
// (scripts/KS/setup/KSSetup.jl:124-125,140-158).
#pragma once
#include <hip/hip_runtime.h>

namespace pdec {

template <class T>
struct C2 {
  T x, y;
};
template <class T>
__device__ __forceinline__ C2<T> mk(T x, T y) {
  C2<T> r;
  r.x = x;
  r.y = y;
  return r;
}
template <class T>
__device__ __forceinline__ C2<T> operator+(C2<T> a, C2<T> b) { return mk<T>(a.x + b.x, a.y + b.y); }
template <class T>
__device__ __forceinline__ C2<T> operator-(C2<T> a, C2<T> b) { return mk<T>(a.x - b.x, a.y - b.y); }
template <class T>
__device__ __forceinline__ C2<T> cmul(C2<T> a, C2<T> b) {
  return mk<T>(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
template <class T>
__device__ __forceinline__ C2<T> cscale(C2<T> a, T s) { return mk<T>(a.x * s, a.y * s); }
// multiply by -i (SGN=-1) or +i (SGN=+1)
template <int SGN, class T>
__device__ __forceinline__ C2<T> mul_i(C2<T> a) {
  return SGN < 0 ? mk<T>(a.y, -a.x) : mk<T>(-a.y, a.x);
}

#define PDEC_MAX_STAGES 12

struct FftPlan {
  int N;
  int nstages;
  int radix[PDEC_MAX_STAGES];
};

// factor N into radices {4,2,3,5}; returns false if another prime factor remains
inline bool make_fft_plan(int N, FftPlan& pl) {
  pl.N = N;
  pl.nstages = 0;
  int n = N;
  const int cand[4] = {4, 2, 3, 5};
  for (int ci = 0; ci < 4; ++ci) {
    int r = cand[ci];
    while (n % r == 0 && !(r == 2 && n % 4 == 0)) {
      if (pl.nstages >= PDEC_MAX_STAGES) return false;
      pl.radix[pl.nstages++] = r;
      n /= r;
    }
  }
  return n == 1 && N >= 2;
}

// DFT of R points in registers; SGN=-1 forward, +1 inverse
template <int R, int SGN, class T>
__device__ __forceinline__ void dft_small(C2<T>* a) {
  if (R == 2) {
    C2<T> t = a[0] - a[1];
    a[0] = a[0] + a[1];
    a[1] = t;
  } else if (R == 4) {
    C2<T> s02 = a[0] + a[2], d02 = a[0] - a[2];
    C2<T> s13 = a[1] + a[3], d13 = mul_i<SGN>(a[1] - a[3]);
    a[0] = s02 + s13;
    a[2] = s02 - s13;
    a[1] = d02 + d13;
    a[3] = d02 - d13;
  } else if (R == 3) {
    const T s60 = (T)0.86602540378443864676;
    C2<T> t1 = a[1] + a[2];
    C2<T> m = mk<T>(a[0].x - (T)0.5 * t1.x, a[0].y - (T)0.5 * t1.y);
    C2<T> s = cscale(mul_i<SGN>(a[1] - a[2]), s60);
    a[0] = a[0] + t1;
    a[1] = m + s;
    a[2] = m - s;
  } else if (R == 5) {
    const T c1 = (T)0.30901699437494742410, c2 = (T)-0.80901699437494742410;
    const T s1 = (T)0.95105651629515357212, s2 = (T)0.58778525229247312917;
    C2<T> t14 = a[1] + a[4], d14 = a[1] - a[4];
    C2<T> t23 = a[2] + a[3], d23 = a[2] - a[3];
    C2<T> m1 = mk<T>(a[0].x + c1 * t14.x + c2 * t23.x, a[0].y + c1 * t14.y + c2 * t23.y);
    C2<T> m2 = mk<T>(a[0].x + c2 * t14.x + c1 * t23.x, a[0].y + c2 * t14.y + c1 * t23.y);
    C2<T> q1 = mul_i<SGN>(mk<T>(s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y));
    C2<T> q2 = mul_i<SGN>(mk<T>(s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y));
    a[0] = a[0] + t14 + t23;
    a[1] = m1 + q1;
    a[4] = m1 - q1;
    a[2] = m2 + q2;
    a[3] = m2 - q2;
  }
}

template <int R, int SGN, class T>
__device__ __forceinline__ void fft_stage(const C2<T>* __restrict__ X, C2<T>* __restrict__ Y,
                                          const C2<T>* __restrict__ tw, int N, int n, int s,
                                          int tid, int nthreads) {
  const int m = n / R;
  const int nb = N / R;
  for (int t = tid; t < nb; t += nthreads) {
    const int p = t / s, q = t - p * s;
    C2<T> a[R];
#pragma unroll
    for (int j = 0; j < R; ++j) a[j] = X[q + s * (p + m * j)];
    dft_small<R, SGN, T>(a);
    const int base = q + s * R * p;
    const int ps = p * s;
    Y[base] = a[0];
#pragma unroll
    for (int k = 1; k < R; ++k) {
      C2<T> w = tw[ps * k];
      if (SGN > 0) w.y = -w.y;
      Y[base + s * k] = cmul(a[k], w);
    }
  }
}

// Runs all stages; data starts in X (caller has written it, no barrier needed before the
// call -- one is issued here).  Returns the buffer holding the natural-order result; a
// barrier has been issued after the last stage.  tw[i] = exp(-2 pi i * i / N).
template <int SGN, class T>
__device__ __forceinline__ C2<T>* fft_lds(C2<T>* X, C2<T>* Y, const C2<T>* tw, const FftPlan& pl,
                                          int tid, int nthreads) {
  int n = pl.N, s = 1;
  __syncthreads();
  for (int st = 0; st < pl.nstages; ++st) {
    const int r = pl.radix[st];
    if (r == 4)
      fft_stage<4, SGN, T>(X, Y, tw, pl.N, n, s, tid, nthreads);
    else if (r == 2)
      fft_stage<2, SGN, T>(X, Y, tw, pl.N, n, s, tid, nthreads);
    else if (r == 3)
      fft_stage<3, SGN, T>(X, Y, tw, pl.N, n, s, tid, nthreads);
    else
      fft_stage<5, SGN, T>(X, Y, tw, pl.N, n, s, tid, nthreads);
    __syncthreads();
    C2<T>* t = X;
    X = Y;
    Y = t;
    n /= r;
    s *= r;
  }
  return X;
}

}  // namespace pdec
