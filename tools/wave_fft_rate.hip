// Issue time of the register-resident fp64 line transform (csrc/wave_fft.hpp) with nothing else around it: a workgroup per CU of
// 4 * WPS waves (WPS waves per SIMD) runs `iters` inverse + forward pairs on register data.  Output: ns per transform per SIMD.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Idistributedconvrl-pde-control_amd/csrc tools/wave_fft_rate.hip -o tools/wave_fft_rate
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "wave_fft.hpp"
using namespace pdec;

template <int E, int Q, bool WQM>
__global__ void k(const C2<double>* tw, C2<double>* out, int iters) {
  typedef WaveFftD<E, Q, 6, WQM> F;
  extern __shared__ __align__(16) unsigned char smem[];
  F f;
  f.init(tw, threadIdx.x & 63, reinterpret_cast<C2<double>*>(smem), threadIdx.x < 64);
  __syncthreads();
  C2<double> a[F::R];
#pragma unroll
  for (int j = 0; j < F::R; ++j) a[j] = mk<double>(1e-3 * threadIdx.x + j, 0.5 * j);
  for (int it = 0; it < iters; ++it) {
    f.inverse(a);
#pragma unroll
    for (int j = 0; j < F::R; ++j) a[j] = mk<double>(a[j].x * (1.0 / F::N), a[j].y * (1.0 / F::N));
    f.forward(a);
  }
  C2<double> s = mk<double>(0, 0);
#pragma unroll
  for (int j = 0; j < F::R; ++j) s = s + a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int E, int Q, bool WQM>
void run(int wps) {
  typedef WaveFftD<E, Q, 6, WQM> F;
  const int N = F::N, iters = 2000, threads = 256 * wps, blocks = 256;
  std::vector<C2<double>> tw(N);
  for (int m = 0; m < N; ++m) { tw[m].x = cos(-2 * M_PI * m / N); tw[m].y = sin(-2 * M_PI * m / N); }
  C2<double>*dtw, *out;
  (void)hipMalloc(&dtw, 16 * N); (void)hipMalloc(&out, 16 * threads * blocks);
  (void)hipMemcpy(dtw, tw.data(), 16 * N, hipMemcpyHostToDevice);
  const size_t lds = 100 * 1024;       // one workgroup per CU
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k<E, Q, WQM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k<E, Q, WQM>), dim3(blocks), dim3(threads), lds, 0, dtw, out, 10);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<E, Q, WQM>), dim3(blocks), dim3(threads), lds, 0, dtw, out, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  const double per_wave = ms * 1e6 / (2.0 * iters), per_simd = per_wave / wps;
  printf("N = %d (E %d, Q %d, radix-Q twiddles from %s), %d waves / SIMD: %.0f ns per transform per wave, %.0f ns per SIMD\n", N, E, Q,
         WQM ? "LDS" : "registers", wps, per_wave, per_simd);
  (void)hipFree(dtw); (void)hipFree(out);
}

int main() {
  for (int w : {1, 2}) { run<4, 3, false>(w); run<4, 3, true>(w); run<2, 3, false>(w); run<4, 2, false>(w); run<4, 1, false>(w); }
  return 0;
}
