// Diagnostic microbenchmark (not part of the product): how much does a VALU-only latency chain (the KS step's shape:
// one wave per workgroup, 256 workgroups, s_setprio 3) slow down beside an MFMA-dense kernel (one 512-thread workgroup
// per CU, operands from LDS) when the MFMAs are (a) f32-input 16x16x4 (vector rate) or (b) bf16 16x16x32 (matrix core),
// and what each MFMA loop sustains.      hipcc --offload-arch=gfx950 -O3 -o mfma_corun_micro mfma_corun_micro.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(512) void hog_f32(int iters, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (i % 97) * 1e-3f;
  __syncthreads();
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float b = threadIdx.x * 1e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(&lds[((threadIdx.x + it) * 4 + 256 * i) & 16383]);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[0], b, acc[i], 0, 0, 0);
      acc[(i + 1) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[1], b, acc[(i + 1) & 3], 0, 0, 0);
      acc[(i + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[2], b, acc[(i + 2) & 3], 0, 0, 0);
      acc[(i + 3) & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[3], b, acc[(i + 3) & 3], 0, 0, 0);
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
// 16 MFMAs per iteration as above; each bf16 MFMA does 8x the flops of an f32 one (K = 32 vs 4)
__global__ __launch_bounds__(512) void hog_bf16(int iters, float* out) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 16384; i += 512) lds[i] = (i % 97) * 1e-3f;
  __syncthreads();
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 b;
  for (int j = 0; j < 8; ++j) b[j] = (__bf16)(threadIdx.x * 1e-3f + j);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(&lds[((threadIdx.x + it) * 4 + 256 * i + 1024 * u) & 16383]);
        acc[(i + u) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(i + u) & 3], 0, 0, 0);
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.f) out[0] = s;
}
// dependent chain of packed-free scalar FMAs: n * 8 v_fma_f32 per lane, one wave per workgroup
__global__ __launch_bounds__(64) void valu_chain(int n, float* out) {
  __builtin_amdgcn_s_setprio(3);
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) a = __builtin_fmaf(a, b, c);
  }
  if (a == 12345.f) out[0] = a;
}

static float run(hipStream_t sa, hipStream_t sb, int which_hog, int iters, bool with_valu, int nval, float* d, float* t_valu) {
  hipEvent_t e0, e1, f0, f1;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&f0); hipEventCreate(&f1);
  float ta = 0, tb = 0;
  for (int rep = 0; rep < 3; ++rep) {
    if (which_hog >= 0) {
      hipEventRecord(e0, sa);
      if (which_hog == 0) hipLaunchKernelGGL(hog_f32, dim3(256), dim3(512), 100 * 1024, sa, iters, d);
      else hipLaunchKernelGGL(hog_bf16, dim3(256), dim3(512), 100 * 1024, sa, iters, d);
      hipEventRecord(e1, sa);
    }
    if (with_valu) {
      hipEventRecord(f0, sb);
      hipLaunchKernelGGL(valu_chain, dim3(256), dim3(64), 0, sb, nval, d);
      hipEventRecord(f1, sb);
    }
    hipDeviceSynchronize();
  }
  if (which_hog >= 0) hipEventElapsedTime(&ta, e0, e1);
  if (with_valu) hipEventElapsedTime(&tb, f0, f1);
  *t_valu = tb * 1e3f;
  return ta * 1e3f;
}

int main() {
  float* d;
  CK(hipMalloc(&d, 4096));
  hipStream_t sa, sb;
  CK(hipStreamCreate(&sa));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, -1));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_f32), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(hog_bf16), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int iters = 600, nval = 1500;     // hog ~ 16 * 600 MFMAs per wave; chain = 12000 dependent FMAs
  float tv;
  const double nm = 256.0 * 8 * 16.0 * iters;     // MFMAs in the grid
  float t = run(sa, sb, 0, iters, false, 0, d, &tv);
  printf("f32  16x16x4  hog alone : %8.1f us  -> %.1f TFLOP/s\n", t, nm * 2048 / (t * 1e-6) / 1e12);
  t = run(sa, sb, 1, iters, false, 0, d, &tv);
  printf("bf16 16x16x32 hog alone : %8.1f us  -> %.1f TFLOP/s\n", t, nm * 16384 / (t * 1e-6) / 1e12);
  run(sa, sb, -1, 0, true, nval, d, &tv);
  printf("valu chain alone        : %8.1f us\n", tv);
  t = run(sa, sb, 0, iters, true, nval, d, &tv);
  printf("beside f32 hog          : hog %8.1f us, valu chain %8.1f us\n", t, tv);
  t = run(sa, sb, 1, iters * 2, true, nval, d, &tv);
  printf("beside bf16 hog (2x it) : hog %8.1f us, valu chain %8.1f us\n", t, tv);
  return 0;
}
