// bf16x3_micro.hip -- bounded experiment: one 144 x 144 hidden layer of the critic, Y = W X + b, for the
// 16 columns a wave owns, the way the fused DDPG passes run it (8 waves per workgroup, one workgroup per CU, weight image in
// LDS, the input activation X as accumulator-layout tiles in registers), in two operand formats:
//   f32   : v_mfma_f32_16x16x4_f32, A = one f32 per lane from the padded f32 image (what csrc/mlp_mfma.hip does)
//   split : v_mfma_f32_16x16x32_bf16 on bf16 splits -- W = Whi + Wmid (two bf16 images, 16 mantissa bits), X = Xhi + Xmid +
//           Xlo (three splits made in registers, 24 bits), five products accumulated in f32:
//           Whi Xlo + Wmid Xmid + Wmid Xhi + Whi Xmid + Whi Xhi   (the dropped terms are <= 2^-17 relative)
// Prints time per layer call and the error of both against an fp64 host reference.
// Build: hipcc -O3 --offload-arch=gfx950 tools/bf16x3_micro.hip -o tools/bf16x3_micro ; run: tools/bf16x3_micro
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

#define H 144
#define MT 9
#define LDWF 152               // f32 image row stride (floats): conflict-free ds_read_b128
#define KB 5                   // 32-deep k blocks (160 >= 144)
#define RSB 176                // bf16 image row stride (bf16 elements) = 352 B: conflict-free ds_read_b128
#define THREADS 512

#define CHECK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e__), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// ---------------------------------------------------------------- f32 layer (the production form)
template <int REPS>
__global__ __launch_bounds__(THREADS) void layer_f32_kernel(const float* __restrict__ Wimg, const float* __restrict__ X, float* __restrict__ Y) {
  extern __shared__ __align__(16) float smem[];
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  for (int i = tid; i < H * LDWF; i += THREADS) smem[i] = Wimg[i];
  const int col = blockIdx.x * 128 + w * 16 + lr;
  f32x4 in[MT], out[MT];
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 4; ++r) in[m][r] = X[(size_t)col * H + 16 * m + 4 * q + r];
  __syncthreads();
  for (int rep = 0; rep < REPS; ++rep) {
#pragma unroll
    for (int mo = 0; mo + 1 < MT; mo += 2) {
      f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      const float* w0 = smem + (16 * mo + lr) * LDWF + 4 * q;
      const float* w1 = w0 + 16 * LDWF;
      f32x4 wa = *reinterpret_cast<const f32x4*>(w0), wb = *reinterpret_cast<const f32x4*>(w1);
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        f32x4 wan = wa, wbn = wb;
        if (m + 1 < MT) { wan = *reinterpret_cast<const f32x4*>(w0 + 16 * (m + 1)); wbn = *reinterpret_cast<const f32x4*>(w1 + 16 * (m + 1)); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 4; ++r) { acc0 = mfma4(wa[r], in[m][r], acc0); acc1 = mfma4(wb[r], in[m][r], acc1); }
        __builtin_amdgcn_sched_barrier(0);
        wa = wan; wb = wbn;
      }
      out[mo] = acc0; out[mo + 1] = acc1;
    }
    {
      f32x4 acc = {0, 0, 0, 0};
      const float* w0 = smem + (16 * (MT - 1) + lr) * LDWF + 4 * q;
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const f32x4 wv = *reinterpret_cast<const f32x4*>(w0 + 16 * m);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma4(wv[r], in[m][r], acc);
      }
      out[MT - 1] = acc;
    }
    if (rep + 1 < REPS)      // keep the repetitions dependent without changing the numbers that are checked (rep 0 = last)
      for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(in[m]) : "v"(out[m]));
  }
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 4; ++r) Y[(size_t)col * H + 16 * m + 4 * q + r] = out[m][r];
}

// ---------------------------------------------------------------- bf16-split layer
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {      // two f32 -> packed bf16 (round to nearest even), a in the low half
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  bf2 v = __builtin_convertvector(f2{a, b}, bf2);
  return __builtin_bit_cast(unsigned, v);
}
// one accumulator-layout tile pair (8 f32 per lane) -> three bf16 fragments hi / mid / lo
__device__ __forceinline__ void split3(const float (&x)[8], s16x8& hi, s16x8& mid, s16x8& lo) {
  unsigned h[4], m[4], lw[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const float a = x[2 * p], b = x[2 * p + 1];
    h[p] = pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, h[p] << 16), rb = b - __builtin_bit_cast(float, h[p] & 0xffff0000u);
    m[p] = pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, m[p] << 16), sb = rb - __builtin_bit_cast(float, m[p] & 0xffff0000u);
    lw[p] = pk_bf16(sa, sb);
  }
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  hi = __builtin_bit_cast(s16x8, u4{h[0], h[1], h[2], h[3]});
  mid = __builtin_bit_cast(s16x8, u4{m[0], m[1], m[2], m[3]});
  lo = __builtin_bit_cast(s16x8, u4{lw[0], lw[1], lw[2], lw[3]});
}
__device__ __forceinline__ f32x4 mfma_bf(s16x8 a, s16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// image layout: [split 2][row 144][RSB bf16]; element (row, t, q, j) at row * RSB + t * 32 + q * 8 + j holds
// W[row][k = 16 (2 t + (j >> 2)) + 4 q + (j & 3)]  (the k order in which two accumulator tiles sit in a lane group)
template <int REPS, int NPROD>
__global__ __launch_bounds__(THREADS) void layer_split_kernel(const uint16_t* __restrict__ Wimg, const float* __restrict__ X, float* __restrict__ Y) {
  extern __shared__ __align__(16) float smem[];
  uint16_t* img = reinterpret_cast<uint16_t*>(smem);
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, lr = l & 15, q = l >> 4;
  for (int i = tid; i < 2 * H * RSB / 2; i += THREADS) reinterpret_cast<unsigned*>(img)[i] = reinterpret_cast<const unsigned*>(Wimg)[i];
  const int col = blockIdx.x * 128 + w * 16 + lr;
  f32x4 in[MT + 1], out[MT];
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 4; ++r) in[m][r] = X[(size_t)col * H + 16 * m + 4 * q + r];
  in[MT] = f32x4{0, 0, 0, 0};
  __syncthreads();
  const uint16_t* ahi = img + (size_t)lr * RSB + q * 8;
  const uint16_t* amid = ahi + H * RSB;
  for (int rep = 0; rep < REPS; ++rep) {
    s16x8 bh[KB], bm[KB], bl[KB];
#pragma unroll
    for (int t = 0; t < KB; ++t) {
      const float x8[8] = {in[2 * t][0], in[2 * t][1], in[2 * t][2], in[2 * t][3], in[2 * t + 1][0], in[2 * t + 1][1], in[2 * t + 1][2], in[2 * t + 1][3]};
      split3(x8, bh[t], bm[t], bl[t]);
    }
#pragma unroll
    for (int mo = 0; mo < MT; ++mo) {
      f32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int t = 0; t < KB; ++t) {
        const s16x8 ah = *reinterpret_cast<const s16x8*>(ahi + (size_t)16 * mo * RSB + t * 32);
        const s16x8 am = *reinterpret_cast<const s16x8*>(amid + (size_t)16 * mo * RSB + t * 32);
        if (NPROD >= 5) acc = mfma_bf(ah, bl[t], acc);
        if (NPROD >= 4) acc = mfma_bf(am, bm[t], acc);
        if (NPROD >= 3) acc = mfma_bf(am, bh[t], acc);
        if (NPROD >= 2) acc = mfma_bf(ah, bm[t], acc);
        acc = mfma_bf(ah, bh[t], acc);
      }
      out[mo] = acc;
    }
    if (rep + 1 < REPS)
      for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(in[m]) : "v"(out[m]));
  }
  for (int m = 0; m < MT; ++m)
    for (int r = 0; r < 4; ++r) Y[(size_t)col * H + 16 * m + 4 * q + r] = out[m][r];
}

static uint16_t bf16_rn(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  u += 0x7fff + ((u >> 16) & 1);
  return (uint16_t)(u >> 16);
}
static float bf16_f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <class K>
static float time_kernel(K kern, int grid, size_t lds, const void* a, const float* x, float* y, int reps_in_kernel) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kern(grid, lds, a, x, y);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int n = 20;
  for (int i = 0; i < n; ++i) kern(grid, lds, a, x, y);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  return ms / n * 1e3f;     // us per launch
}

int main() {
  const int cols = 32768, grid = cols / 128;
  std::vector<float> W(H * H), X((size_t)cols * H);
  srand(1);
  auto rnd = [] { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (auto& v : W) v = rnd() * 0.17f;            // ~ glorot for 144 x 144
  for (auto& v : X) v = fmaxf(rnd() * 1.5f, 0.f); // relu-like activations
  std::vector<float> imgF((size_t)H * LDWF, 0.f);
  for (int r = 0; r < H; ++r) for (int c = 0; c < H; ++c) imgF[(size_t)r * LDWF + c] = W[r * H + c];
  std::vector<uint16_t> imgB((size_t)2 * H * RSB, 0);
  for (int r = 0; r < H; ++r)
    for (int t = 0; t < KB; ++t) for (int q = 0; q < 4; ++q) for (int j = 0; j < 8; ++j) {
      const int k = 16 * (2 * t + (j >> 2)) + 4 * q + (j & 3);
      const float wv = k < H ? W[r * H + k] : 0.f;
      const uint16_t hi = bf16_rn(wv), mid = bf16_rn(wv - bf16_f(hi));
      imgB[(size_t)r * RSB + t * 32 + q * 8 + j] = hi;
      imgB[(size_t)H * RSB + (size_t)r * RSB + t * 32 + q * 8 + j] = mid;
    }
  float *dF, *dX, *dY; uint16_t* dB;
  CHECK(hipMalloc(&dF, imgF.size() * 4)); CHECK(hipMalloc(&dB, imgB.size() * 2));
  CHECK(hipMalloc(&dX, X.size() * 4)); CHECK(hipMalloc(&dY, X.size() * 4));
  CHECK(hipMemcpy(dF, imgF.data(), imgF.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dB, imgB.data(), imgB.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  const size_t ldsF = (size_t)H * LDWF * 4, ldsB = (size_t)2 * H * RSB * 2;
  // fp64 reference on a sample of columns
  const int ncheck = 512;
  std::vector<double> ref((size_t)ncheck * H);
  double scale = 0;
  for (int c = 0; c < ncheck; ++c) for (int r = 0; r < H; ++r) {
    double s = 0;
    for (int k = 0; k < H; ++k) s += (double)W[r * H + k] * (double)X[(size_t)c * H + k];
    ref[(size_t)c * H + r] = s;
    scale = fmax(scale, fabs(s));
  }
  std::vector<float> Y((size_t)ncheck * H);
  auto err = [&](const char* name) {
    CHECK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
    double e = 0, e2 = 0;
    for (size_t i = 0; i < Y.size(); ++i) { const double d = fabs((double)Y[i] - ref[i]); e = fmax(e, d); e2 += d * d; }
    printf("  %-28s max abs err %.3e  rms %.3e  (max |y| = %.3f)\n", name, e, sqrt(e2 / Y.size()), scale);
  };
#define SETATTR(k, b) CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(b)))
  SETATTR((layer_f32_kernel<1>), ldsF); SETATTR((layer_f32_kernel<16>), ldsF);
  SETATTR((layer_split_kernel<1, 5>), ldsB); SETATTR((layer_split_kernel<16, 5>), ldsB);
  SETATTR((layer_split_kernel<1, 3>), ldsB); SETATTR((layer_split_kernel<16, 3>), ldsB);
  SETATTR((layer_split_kernel<16, 1>), ldsB);
  auto f1 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_f32_kernel<1>), dim3(g), dim3(THREADS), l, 0, (const float*)a, x, y); };
  auto f16 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_f32_kernel<16>), dim3(g), dim3(THREADS), l, 0, (const float*)a, x, y); };
  auto s1 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_split_kernel<1, 5>), dim3(g), dim3(THREADS), l, 0, (const uint16_t*)a, x, y); };
  auto s16 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_split_kernel<16, 5>), dim3(g), dim3(THREADS), l, 0, (const uint16_t*)a, x, y); };
  auto t1 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_split_kernel<1, 3>), dim3(g), dim3(THREADS), l, 0, (const uint16_t*)a, x, y); };
  auto t16 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_split_kernel<16, 3>), dim3(g), dim3(THREADS), l, 0, (const uint16_t*)a, x, y); };
  auto u16 = [](int g, size_t l, const void* a, const float* x, float* y) { hipLaunchKernelGGL((layer_split_kernel<16, 1>), dim3(g), dim3(THREADS), l, 0, (const uint16_t*)a, x, y); };
  printf("one 144x144 layer, %d columns, %d workgroups x 8 waves (16 columns per wave), weight image in LDS\n", cols, grid);
  const float a1 = time_kernel(f1, grid, ldsF, dF, dX, dY, 1), a16 = time_kernel(f16, grid, ldsF, dF, dX, dY, 16);
  f1(grid, ldsF, dF, dX, dY); CHECK(hipDeviceSynchronize());
  printf("f32   16x16x4 f32 : %.2f us per layer (16 in-kernel repetitions: %.1f us, 1: %.1f us)\n", (a16 - a1) / 15, a16, a1);
  err("f32 MFMA");
  const float b1 = time_kernel(s1, grid, ldsB, dB, dX, dY, 1), b16 = time_kernel(s16, grid, ldsB, dB, dX, dY, 16);
  s1(grid, ldsB, dB, dX, dY); CHECK(hipDeviceSynchronize());
  printf("split 5 products  : %.2f us per layer (16: %.1f us, 1: %.1f us)  -> %.2fx the f32 layer\n", (b16 - b1) / 15, b16, b1, (a16 - a1) / (b16 - b1));
  err("bf16 split, 5 products");
  const float c1 = time_kernel(t1, grid, ldsB, dB, dX, dY, 1), c16 = time_kernel(t16, grid, ldsB, dB, dX, dY, 16);
  t1(grid, ldsB, dB, dX, dY); CHECK(hipDeviceSynchronize());
  printf("split 3 products  : %.2f us per layer (16: %.1f us, 1: %.1f us)  -> %.2fx\n", (c16 - c1) / 15, c16, c1, (a16 - a1) / (c16 - c1));
  err("bf16 split, 3 products");
  const float d16 = time_kernel(u16, grid, ldsB, dB, dX, dY, 16);
  printf("split 1 product (conversion + operand reads + 45 MFMAs, timing only): %.2f us per layer\n", (d16 - b1) / 15);
  return 0;
}
