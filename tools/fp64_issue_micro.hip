// fp64 / packed-fp32 / DPP issue-rate micro-benchmark for gfx950 (round 4): cycles per wave-instruction of v_fma_f64, v_add_f64, v_mul_f64 and
// v_cndmask_b32_dpp with 1, 2, 4 waves per SIMD and 1..8 independent chains per wave.  Build: hipcc -O3 --offload-arch=gfx950
// tools/fp64_issue_micro.hip -o tools/fp64_issue_micro; run on the GPU box.  Output: one line per (op, waves/SIMD, chains).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP, int CH>
__global__ void k(double* out, long long* cyc, int iters) {
  double a[CH], b = 1.0000001, c = 1e-9;
  for (int i = 0; i < CH; ++i) a[i] = threadIdx.x * 1e-3 + i;
  unsigned u[CH];
  for (int i = 0; i < CH; ++i) u[i] = threadIdx.x + i;
  __syncthreads();
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (OP == 3) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]));
        if (OP == 4) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(1.0f), "v"(0.5f));
        if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (OP == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
        if (OP == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < CH; ++i) s += a[i] + u[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP, int CH>
void run(const char* name, int wps) {
  const int iters = 20000, threads = 64 * 4 * wps, blocks = 256;
  double* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * threads * blocks); (void)hipMalloc(&cyc, 8 * blocks);
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<OP, CH>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  (void)hipEventRecord(e1, 0);
  (void)hipDeviceSynchronize();
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
  double avg = 0; for (auto v : h) avg += v; avg /= blocks;
  // clock64 = s_memtime (100 MHz-based constant counter on gfx9?) -> report raw ticks per instruction per wave and per SIMD
  const double per_wave = avg / (iters * 8.0 * CH);
  // wall clock: one workgroup per CU, wps waves per SIMD -> instructions per SIMD = iters * 8 * CH * wps
  const double ns_per_instr_simd = ms * 1e6 / (iters * 8.0 * CH * wps);
  printf("%-10s waves/SIMD %d chains %d : %.2f ticks/instr/wave, %.2f ticks/instr/SIMD | wall %.3f ns/instr/SIMD = %.2f cycles at 2.4 GHz\n", name, wps, CH,
         per_wave, per_wave / wps, ns_per_instr_simd, ns_per_instr_simd * 2.4);
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<0, 1>("fma_f64", w); run<0, 2>("fma_f64", w); run<0, 4>("fma_f64", w); run<0, 8>("fma_f64", w);
    run<1, 4>("add_f64", w); run<2, 4>("mul_f64", w); run<3, 4>("mov_dpp", w); run<3, 1>("mov_dpp", w);
    run<4, 4>("fma_f32", w); run<4, 1>("fma_f32", w);
    run<5, 4>("pk_fma_f32", w); run<6, 4>("pk_add_f32", w); run<7, 4>("pk_mul_f32", w); run<5, 1>("pk_fma_f32", w);
  }
  return 0;
}
