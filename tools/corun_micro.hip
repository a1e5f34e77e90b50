// Diagnostic microbenchmark (not part of the product): when does a small kernel get CU residency beside a
// kernel whose workgroups (one per CU) hold a lot of LDS / registers?   hipcc --offload-arch=gfx950 -O3 -o corun_micro corun_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int NV>
__global__ void hog(unsigned long long ticks, float* out) {
  extern __shared__ float lds[];
  float v[NV];
  for (int i = 0; i < NV; ++i) v[i] = threadIdx.x * 0.5f + i;
  lds[threadIdx.x] = v[0];
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
    for (int i = 0; i < NV; ++i) v[i] = v[i] * 1.0001f + lds[(threadIdx.x + i) & 255];
    __syncthreads();
  }
  float s = 0;
  for (int i = 0; i < NV; ++i) s += v[i];
  if (s == 12345.f) out[0] = s;
}
typedef float f32x4 __attribute__((ext_vector_type(4)));
// MFMA-dense hog: back-to-back v_mfma_f32_16x16x4_f32 chains, optional LDS reads between them
template <int LDSREADS>
__global__ __launch_bounds__(512) void hog_mfma(unsigned long long ticks, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = i * 1e-6f;
  __syncthreads();
  f32x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  int it = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (LDSREADS) { const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x + it) * 4 + 64 * i) & 16383]); a = v[0]; b = v[1]; }
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
    }
    ++it;
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
__global__ void small(unsigned long long ticks, float* out) {
  __shared__ float l2[1024];
  l2[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  float a = 0;
  while (__builtin_amdgcn_s_memtime() - t0 < ticks) a += l2[(threadIdx.x * 7) & 1023];
  if (a == 12345.f) out[0] = a;
}

template <int NV>
int trial(int lds_kb, int threads, const char* tag) {
  float* d;
  CK(hipMalloc(&d, 4096));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, -1));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog<NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned long long T = 100ull * 1000;   // s_memtime ticks; ~ 100 MHz * ... calibrated below by the event time
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(hog<NV>, dim3(256), dim3(threads), lds_kb * 1024, sa, T, d);
    CK(hipEventRecord(e1, sa));
    CK(hipEventRecord(f0, sb));
    hipLaunchKernelGGL(small, dim3(256), dim3(64), 0, sb, T / 20, d);
    CK(hipEventRecord(f1, sb));
    CK(hipDeviceSynchronize());
  }
  float ta, tb;
  CK(hipEventElapsedTime(&ta, e0, e1));
  CK(hipEventElapsedTime(&tb, f0, f1));
  printf("%-10s hog: %3d KB LDS, %4d threads, ~%d VGPR  -> hog %.1f us, small kernel (alone ~1/20 of hog) %.1f us\n", tag, lds_kb,
         threads, NV + 10, ta * 1e3, tb * 1e3);
  return 0;
}

template <int LDSREADS>
int trial_mfma(int lds_kb, const char* tag) {
  float* d;
  CK(hipMalloc(&d, 4096));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, -1));
  hipEvent_t e0, e1, f0, f1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_mfma<LDSREADS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const unsigned long long T = 100ull * 1000;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0, sa));
    hipLaunchKernelGGL(hog_mfma<LDSREADS>, dim3(256), dim3(512), lds_kb * 1024, sa, T, d);
    CK(hipEventRecord(e1, sa));
    CK(hipEventRecord(f0, sb));
    hipLaunchKernelGGL(small, dim3(256), dim3(64), 0, sb, T / 20, d);
    CK(hipEventRecord(f1, sb));
    CK(hipDeviceSynchronize());
  }
  float ta, tb;
  CK(hipEventElapsedTime(&ta, e0, e1));
  CK(hipEventElapsedTime(&tb, f0, f1));
  printf("%-12s mfma hog: %3d KB LDS, lds reads %d -> hog %.1f us, small kernel %.1f us\n", tag, lds_kb, LDSREADS, ta * 1e3, tb * 1e3);
  return 0;
}

int main() {
  if (trial_mfma<0>(70, "mfma")) return 1;
  if (trial_mfma<0>(104, "mfma")) return 1;
  if (trial_mfma<1>(70, "mfma+lds")) return 1;
  if (trial_mfma<1>(104, "mfma+lds")) return 1;
  for (int kb : {1, 32, 64, 65, 80, 96, 104, 128, 150}) if (trial<8>(kb, 512, "lds-sweep")) return 1;
  for (int th : {256, 512, 1024}) if (trial<8>(104, th, "thr-sweep")) return 1;
  if (trial<100>(1, 512, "vgpr")) return 1;
  if (trial<100>(104, 512, "vgpr+lds")) return 1;
  return 0;
}
