// wave_fft.hpp -- register-resident FFT of one line per WAVE (fp64): N = 64 * E * Q points, element n = lane + 64 j
// (or per half wave: N = 32 * 2 * Q, see LB below)
// in register slot j = e + E * qd (e < E: the "exchange digit", 4 or 2; qd < Q: 1, 2 or 3).
//
// Decimation in frequency: radix-Q over qd (in-lane), then Q independent FFTs of size 64 E: radix-E over e, and for
// every 2-bit (E = 4) or 1-bit (E = 2) digit of the lane id: exchange that lane digit with the register digit e
// (v_permlane32/16_swap for lane bits 5, 4; one v_cndmask_b32_dpp per moved 32-bit word for bits 3..0) and run the
// next in-lane radix-E butterfly.  No LDS, no barriers.  The forward transform leaves mode
//     k = qd + Q * (perm(lane) + 64 e),  perm = (l>>4) + 4((l>>2)&3) + 16(l&3)  (E = 4)   or   bitrev6(l)  (E = 2)
// in slot (qd, e); the inverse (decimation in time) consumes that order and returns natural order, so a
// forward -> pointwise -> inverse chain never reorders data.  Same scheme as FftWave256 in env.hip, generalised
// to the 2-D fluid's line lengths (768 / 512 / 384 / 256 / 128) and to fp64.
#pragma once
#include "fft_lds.hpp"

namespace pdec {

// one binary exchange step on four (P, Q) register pairs: newQ = bit ? Q : perm(P), newP = bit ? perm(Q) : P
#define PDEC_WXSTEP(CA, CB, MASK, P0, Q0, P1, Q1, P2, Q2, P3, Q3)                                          \
  {                                                                                                        \
    unsigned n0_, n1_, n2_, n3_;                                                                           \
    asm("s_nop 1\n\t"                                                                                      \
        "s_mov_b64 vcc, %12\n\t"                                                                           \
        "v_cndmask_b32_dpp %8, %0, %1, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %9, %2, %3, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %10, %4, %5, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                          \
        "v_cndmask_b32_dpp %11, %6, %7, vcc " CA " row_mask:0xf bank_mask:0xf\n\t"                          \
        "s_mov_b64 vcc, %13\n\t"                                                                           \
        "v_cndmask_b32_dpp %0, %1, %0, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %2, %3, %2, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %4, %5, %4, vcc " CB " row_mask:0xf bank_mask:0xf\n\t"                           \
        "v_cndmask_b32_dpp %6, %7, %6, vcc " CB " row_mask:0xf bank_mask:0xf"                                \
        : "+v"(P0), "+v"(Q0), "+v"(P1), "+v"(Q1), "+v"(P2), "+v"(Q2), "+v"(P3), "+v"(Q3), "=&v"(n0_), "=&v"(n1_), \
          "=&v"(n2_), "=&v"(n3_)                                                                           \
        : "s"(MASK), "s"(~(MASK))                                                                          \
        : "vcc");                                                                                          \
    Q0 = n0_; Q1 = n1_; Q2 = n2_; Q3 = n3_;                                                                \
  }

// exchange the register bit (P vs Q) with lane bit BIT for the four 32-bit words of a complex double pair
template <int BIT>
__device__ __forceinline__ void wx_bit(unsigned (&p)[4], unsigned (&q)[4]) {
  if (BIT == 5) {
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %4\n\tv_permlane32_swap_b32 %1, %5\n\tv_permlane32_swap_b32 %2, %6\n\t"
        "v_permlane32_swap_b32 %3, %7"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
  } else if (BIT == 4) {
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %4\n\tv_permlane16_swap_b32 %1, %5\n\tv_permlane16_swap_b32 %2, %6\n\t"
        "v_permlane16_swap_b32 %3, %7"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]));
  } else if (BIT == 3) {
    PDEC_WXSTEP("row_ror:8", "row_ror:8", 0xFF00FF00FF00FF00ull, p[0], q[0], p[1], q[1], p[2], q[2], p[3], q[3])
  } else if (BIT == 2) {
    PDEC_WXSTEP("row_ror:12", "row_ror:4", 0xF0F0F0F0F0F0F0F0ull, p[0], q[0], p[1], q[1], p[2], q[2], p[3], q[3])
  } else if (BIT == 1) {
    PDEC_WXSTEP("quad_perm:[2,3,0,1]", "quad_perm:[2,3,0,1]", 0xCCCCCCCCCCCCCCCCull, p[0], q[0], p[1], q[1], p[2], q[2], p[3], q[3])
  } else {
    PDEC_WXSTEP("quad_perm:[1,0,3,2]", "quad_perm:[1,0,3,2]", 0xAAAAAAAAAAAAAAAAull, p[0], q[0], p[1], q[1], p[2], q[2], p[3], q[3])
  }
}
template <int BIT>
__device__ __forceinline__ void wx_pair(C2<double>& a, C2<double>& b) {
  unsigned p[4], q[4];
  __builtin_memcpy(p, &a, 16);
  __builtin_memcpy(q, &b, 16);
  wx_bit<BIT>(p, q);
  __builtin_memcpy(&a, p, 16);
  __builtin_memcpy(&b, q, 16);
}

// LB = 6: one line per wave (64 lanes).  LB = 5 (E = 2 only): one line per HALF wave -- N = 32 * 2 * Q, e.g. 192 for the
// fluid's 128^2 training grid padded by 3/2 -- two independent lines per wave (lanes 0-31 and 32-63); the exchanges
// then only use lane bits 4..0, none of which crosses the halves.
// WQ_MEM: the radix-Q stage twiddles are read from a [(Q - 1) E][LANES] table (LDS) at use instead of held in (Q - 1) E
// complex registers -- 32 VGPRs at E = 4, Q = 3 -- for callers at the register limit (fluid_k2p_kernel).
template <int E, int Q, int LB = 6, bool WQ_MEM = false>
struct WaveFftD {
  static_assert(LB == 6 || (LB == 5 && E == 2), "half-wave lines need the 1-bit exchange digit");
  static constexpr int LANES = 1 << LB, LPW = 64 / LANES;   // lanes per line, lines per wave
  static constexpr int R = E * Q, N = LANES * R, M = LANES * E;
  static constexpr int NST = E == 4 ? LB / 2 : LB;    // lane-digit stages
  C2<double> wq[(Q > 1 && !WQ_MEM) ? (Q - 1) * E : 1];   // radix-Q stage twiddles  tw_N[kq (lane + 64 e)]
  const C2<double>* wq_tab = nullptr;                  // WQ_MEM: the same numbers, element (i, lane) at i * LANES + lane
  __device__ __forceinline__ C2<double> wqv(int i) const {
    if constexpr (WQ_MEM) return wq_tab[i * LANES + lane];
    else return wq[i];
  }
  C2<double> we[E - 1];                               // register-digit stage     tw_M[k lane]
  C2<double> wl[NST > 1 ? (NST - 1) * (E - 1) : 1];   // lane-digit stages (the last one has none)
  int lane;

  // tw: exp(-2 pi i m / N), m < N (global or LDS)
  // tab (WQ_MEM only): the table's storage; fill: this wave writes it (the caller synchronises before the first transform)
  __device__ __forceinline__ void init(const C2<double>* tw, int lane_, C2<double>* tab = nullptr, bool fill = false) {
    lane = lane_ & (LANES - 1);      // position inside the line
    if constexpr (WQ_MEM) {
      wq_tab = tab;
      if (fill) {
#pragma unroll
        for (int kq = 1; kq < Q; ++kq)
#pragma unroll
          for (int e = 0; e < E; ++e) tab[((kq - 1) * E + e) * LANES + lane] = tw[(kq * (lane + LANES * e)) % N];
      }
    } else {
#pragma unroll
      for (int kq = 1; kq < Q; ++kq)
#pragma unroll
        for (int e = 0; e < E; ++e) wq[(kq - 1) * E + e] = tw[(kq * (lane + LANES * e)) % N];
    }
#pragma unroll
    for (int k = 1; k < E; ++k) we[k - 1] = tw[(Q * k * lane) % N];
    int msize = LANES;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) {
      const int msub = msize / E;
      const int low = lane & (msub - 1);
#pragma unroll
      for (int k = 1; k < E; ++k) wl[st * (E - 1) + k - 1] = tw[(Q * k * low * (M / msize)) % N];
      msize = msub;
    }
  }
  // names every twiddle register in an empty asm statement: the compiler places its wait for the global loads of init()
  // HERE instead of at their first use (callers that keep LDS-DMA prefetches in flight need those loads retired before the
  // first DMA is issued, or the first-use wait drains the prefetch queue)
  __device__ __forceinline__ void touch() const {
#pragma unroll
    for (int i = 0; i < ((Q > 1 && !WQ_MEM) ? (Q - 1) * E : 1); ++i) asm volatile("" ::"v"(wq[i].x), "v"(wq[i].y));
#pragma unroll
    for (int i = 0; i < E - 1; ++i) asm volatile("" ::"v"(we[i].x), "v"(we[i].y));
#pragma unroll
    for (int i = 0; i < (NST > 1 ? (NST - 1) * (E - 1) : 1); ++i) asm volatile("" ::"v"(wl[i].x), "v"(wl[i].y));
  }
  // mode index held in slot j after forward()
  __device__ __forceinline__ int mode_index(int j) const {
    const int qd = j / E, e = j - qd * E;
    int perm;
    if (E == 4) perm = (lane >> 4) + 4 * ((lane >> 2) & 3) + 16 * (lane & 3);
    else perm = (int)(__builtin_bitreverse32((unsigned)lane) >> (32 - LB));
    return qd + Q * (perm + LANES * e);
  }
  template <int SGN>
  static __device__ __forceinline__ C2<double> tmul(C2<double> a, C2<double> w) {
    if (SGN > 0) w.y = -w.y;
    return cmul(a, w);
  }
  template <int SGN>
  static __device__ __forceinline__ void dftE(C2<double>* a) { dft_small<E, SGN, double>(a); }
  template <int ST>
  __device__ __forceinline__ void exchange(C2<double>* b) {    // b[0..E): lane digit ST <-> register digit e
    if constexpr (E == 4) {
      constexpr int HI = 5 - 2 * ST, LO = 4 - 2 * ST;
      wx_pair<HI>(b[0], b[2]);
      wx_pair<HI>(b[1], b[3]);
      wx_pair<LO>(b[0], b[1]);
      wx_pair<LO>(b[2], b[3]);
    } else {
      wx_pair<LB - 1 - ST>(b[0], b[1]);
    }
  }
  template <int ST, int SGN>
  __device__ __forceinline__ void stage_fwd(C2<double>* b) {
    exchange<ST>(b);
    dftE<SGN>(b);
    if (ST < NST - 1) {
#pragma unroll
      for (int k = 1; k < E; ++k) b[k] = tmul<SGN>(b[k], wl[ST * (E - 1) + k - 1]);
    }
  }
  template <int ST, int SGN>
  __device__ __forceinline__ void stage_inv(C2<double>* b) {
    if (ST < NST - 1) {
#pragma unroll
      for (int k = 1; k < E; ++k) b[k] = tmul<SGN>(b[k], wl[ST * (E - 1) + k - 1]);
    }
    dftE<SGN>(b);
    exchange<ST>(b);
  }
  // natural order in -> digit-reversed out (unnormalised forward transform, e^{-2 pi i nk/N})
  __device__ __forceinline__ void forward(C2<double> (&a)[R]) {
    if constexpr (Q > 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        C2<double> t[Q];
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) t[qd] = a[qd * E + e];
        dft_small<Q, -1, double>(t);
#pragma unroll
        for (int qd = 1; qd < Q; ++qd) t[qd] = cmul(t[qd], wqv((qd - 1) * E + e));
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) a[qd * E + e] = t[qd];
      }
    }
#pragma unroll
    for (int qd = 0; qd < Q; ++qd) {
      C2<double>* b = &a[qd * E];
      dftE<-1>(b);
#pragma unroll
      for (int k = 1; k < E; ++k) b[k] = cmul(b[k], we[k - 1]);
      stage_fwd<0, -1>(b);
      stage_fwd<1, -1>(b);
      stage_fwd<2, -1>(b);
      if constexpr (NST > 3) { stage_fwd<3, -1>(b); stage_fwd<4, -1>(b); }
      if constexpr (NST > 5) stage_fwd<5, -1>(b);
    }
  }
  // digit-reversed in -> natural order out (unnormalised inverse transform, e^{+2 pi i nk/N})
  __device__ __forceinline__ void inverse(C2<double> (&a)[R]) {
#pragma unroll
    for (int qd = 0; qd < Q; ++qd) {
      C2<double>* b = &a[qd * E];
      if constexpr (NST > 5) stage_inv<5, +1>(b);
      if constexpr (NST > 3) { stage_inv<4, +1>(b); stage_inv<3, +1>(b); }
      stage_inv<2, +1>(b);
      stage_inv<1, +1>(b);
      stage_inv<0, +1>(b);
#pragma unroll
      for (int k = 1; k < E; ++k) b[k] = tmul<+1>(b[k], we[k - 1]);
      dftE<+1>(b);
    }
    if constexpr (Q > 1) {
#pragma unroll
      for (int e = 0; e < E; ++e) {
        C2<double> t[Q];
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) t[qd] = a[qd * E + e];
#pragma unroll
        for (int qd = 1; qd < Q; ++qd) t[qd] = tmul<+1>(t[qd], wqv((qd - 1) * E + e));
        dft_small<Q, +1, double>(t);
#pragma unroll
        for (int qd = 0; qd < Q; ++qd) a[qd * E + e] = t[qd];
      }
    }
  }
};

}  // namespace pdec
