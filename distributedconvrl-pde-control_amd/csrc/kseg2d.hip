// kseg2d.hip -- Keller-Segel on a 2-D ny x nx grid (BASELINE.json configs[3]).
//
// The reference's Keller-Segel is 1-D (scripts/Keller-Segel/setup/KellerSegelSetup.jl); the 2-D extension keeps each
// of its rules along both axes (stated and pinned in oracle/keller_segel2d.py):
//   RHS  :213-232  v' = Lap v - v + u + p;  u' = Lap u + u - 5.6 grad u . grad v - 5.6 u Lap v - u^2, central
//                  differences (:63-64 weights along x and along y), zero-flux edges by ghost = edge cell (:220-223)
//   step :234-239  classical RK4 (src/fluid_rk4.jl:122-132 tableau), K fixed sub-steps, forcing frozen
//   sensing :112-126, :241-316, prepare_action :318-332 with 5x5 boxes of ones on a tensor grid of positions
//
// Data layout in HBM: y [B][ny][nx][2] (u,v interleaved: one 8/16-byte load gives both species of a cell; x is the
// fast axis), p [B][ny][nx], state [B][A][ns], reward/action [B][A].
//
// The integrator is HBM-bound if every RK4 stage goes through memory (4 stages x (read 3 + write 2) fields).  Here
// one launch advances a 64x64 tile by NSUB whole RK4 sub-steps out of LDS: the tile is loaded with a halo of 4*NSUB
// cells (each stage consumes one ring of the halo), y0 / the running sum / the stage value of a thread's cells stay in
// registers, and only the stage value is exchanged through LDS.  Each thread owns strips of 4 consecutive cells of a
// row: the west/east neighbours inside a strip come from registers, the north/south rows are 128-bit LDS reads.
// HBM traffic per sub-step: (1 + 2H/64)^2 reads + 1 write of the tile instead of 4 x (5 fields).
// (Round 4, tried and dropped: the sub-step as a persistent kernel -- one 512-thread workgroup per CU walking its tiles, the
// next tile's region, actuator indices and actions streamed into 66 KiB of LDS staging by LDS-DMA behind the current tile's
// stages, results stored one tile late.  Bit-identical, 75 us per launch against 50: the staging leaves room for ONE
// workgroup per CU, and the four stages -- LDS write, barrier, LDS reads, ~160 packed instructions, barrier -- are latency
// chains that need the 16 waves per CU of two independent workgroups more than they need the HBM latency hidden.)
#include "env.hpp"

namespace pdec {

constexpr int K2_TX = 64;
// tile height / workgroup size per dtype: fp32 64 rows x 512 threads (128 VGPRs -> two workgroups per CU); fp64 needs
// ~250 VGPRs per thread, so 32 rows x 256 threads give two independent workgroups per CU instead of one big one
template <class T> struct K2Tile { static constexpr int TY = sizeof(T) == 4 ? 64 : 32, NT = sizeof(T) == 4 ? 512 : 256; };

template <class T>
struct K2Dev {
  int B, nx, ny, K, Sx, Sy, S, A, hw, ns, window, temporal, check_max;
  T idx, idx2, c56, hstep, sensor_scale, agent_power, r_in_scale, r_offset, r_power, r_denom, a_pun, da_pun, max_value;
  const int* sx;        // [Sx] 0-based centre columns
  const int* sy;        // [Sy] 0-based centre rows
  const int* a2s;       // [A]
  const int* cell_act;  // [ny][nx] actuator whose box covers the cell, -1 = none (boxes do not overlap)
  const int* ftab;      // [A * ns] featurize map: index into a trajectory's box sums [2][S]; bit 31 = a row of the temporal stack
  const int* acnt;      // [A] cells in the actuator's sensor box (reward offset term)
  T* term_out;          // optional [B][A]
};

template <class T>
struct alignas(16) Quad {   // four consecutive cells of a row (u,v pairs)
  C2<T> c[4];
};
template <class T>
struct alignas(16) Quad1 {  // four consecutive scalars
  T c[4];
};

template <class T>
__device__ __forceinline__ T k2_pow_abs(T d, T pw) {
  const T a = fabs(d);
  if (pw == (T)2) return a * a;
  if (pw == (T)1) return a;
  return pow(a, pw);
}

// (u, v) of one cell as a 2-vector: for fp32 the compiler emits the packed VOP3P forms (v_pk_add/mul/fma_f32),
// so the Laplacians, the differences and the RK4 updates of both species cost one instruction
template <class T>
using V2 = T __attribute__((ext_vector_type(2)));
template <class T>
__device__ __forceinline__ V2<T> v2(C2<T> c) { V2<T> r; r.x = c.x; r.y = c.y; return r; }
template <class T>
__device__ __forceinline__ C2<T> c2(V2<T> v) { return mk<T>(v.x, v.y); }

// Right-hand side of TWO adjacent cells at once (x = cell i, y = cell i+1 of a row), one species per vector: every
// operation is element-wise over the pair, so for fp32 the whole expression is packed VOP3P math (v_pk_add/mul/fma_f32,
// two cells per instruction).  Same discretisation as the 1-D kernel kseg_rhs (env.hip) with the constants folded:
// Lap = ((w + e) + (s + n) - 4c)/dx^2, grad u . grad v = ((e-w)_u (e-w)_v + (n-s)_u (n-s)_v) / (4 dx^2),
// u' = Lap u + u (1 - 5.6 Lap v - u) - 5.6 grad u . grad v,  v' = Lap v - v + u + p
template <class T>
__device__ __forceinline__ void k2_rhs_pair(const K2Dev<T>& e, V2<T> u, V2<T> v, V2<T> uw, V2<T> ue, V2<T> us, V2<T> un,
                                            V2<T> vw, V2<T> ve, V2<T> vs, V2<T> vn, V2<T> p, V2<T>& ku, V2<T>& kv) {
  const V2<T> lu = (((uw + ue) + (us + un)) - (T)4 * u) * e.idx2;
  const V2<T> lv = (((vw + ve) + (vs + vn)) - (T)4 * v) * e.idx2;
  const V2<T> dot = (ue - uw) * (ve - vw) + (un - us) * (vn - vs);
  const V2<T> t = ((T)1 - (T)5.6 * lv) - u;
  kv = (lv + p) + (u - v);
  ku = (lu + u * t) - e.c56 * dot;
}

template <class T>
struct alignas(16) QuadT {  // four consecutive scalars of one plane = two cell pairs
  V2<T> a, b;
};

// MODE 0: integrate NSUB sub-steps, 1: right-hand side only (KATs), 2: integrate with the forcing synthesised from the
// action table (p_in = action [B][A]; p = agent_power * action[cell_act]: the [ny][nx] int table is shared by all
// trajectories and stays in L2, so the per-sub-step HBM reads drop from 3 to 2 scalars per cell)
// PROBE (pdec_debug_kseg2d_probe only): the sub-step loop runs `last` times on the tile held in registers -- the instruction mix
// of a time-resident kernel without its halo exchange (the halo of the later repetitions is not refreshed: timing only)
template <class T, int NSUB, int MODE, bool PROBE = false>
__global__ __launch_bounds__(K2Tile<T>::NT, (NSUB == 1 && sizeof(T) == 4) ? 4 : 2) void kseg2d_rk4_kernel(K2Dev<T> e, const C2<T>* __restrict__ y_in,
                                                           const T* __restrict__ p_in, C2<T>* __restrict__ y_out,
                                                           int32_t* __restrict__ done, int last) {
  constexpr int K2_TY = K2Tile<T>::TY, K2_NT = K2Tile<T>::NT;
  constexpr int H = 4 * NSUB, RX = K2_TX + 2 * H, RY = K2_TY + 2 * H, SW = RX / 4, NSTRIP = SW * RY;
  constexpr int NS = (NSTRIP + K2_NT - 1) / K2_NT, RXP = RX;
  extern __shared__ __align__(16) unsigned char k2_smem[];
  // Stage values, one plane per species; a thread's 4 cells = one 128-bit access per plane.  Row pitch = RX, no padding
  // (round 6): strip id lies at dword 4 * id, so the 128-bit accesses of a wave -- own strip, north, south -- are 64
  // CONSECUTIVE 16-byte pieces and conflict-free; with the pitch RX + 4 of rounds 3 - 5 every row end inside a 16-lane
  // group shifted the rest of the group by four banks onto the banks of its first lane (SQ_LDS_BANK_CONFLICT 4.0 M of the
  // 7.9 M per launch; bit-identical, C4 78.1 - 79.2 k -> 80.2 - 81.4 k env-steps/s, HISTORY.md 6.2).  The west / east reads
  // stay single dwords out of these planes at a lane stride of 16 bytes (4-way conflicted): compact per-strip edge arrays
  // remove those conflicts and make the kernel SLOWER (more vector instructions; the LDS wait is latency, not bandwidth).
  T* Su = reinterpret_cast<T*>(k2_smem);        // [RY][RXP]
  T* Sv = Su + RY * RXP;                        // [RY][RXP]
  // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (each with its own L2), so consecutive
  // LOGICAL tiles -- the 16 tiles of a trajectory, which share their halos -- are given to one XCD
  const int ntx = (e.nx + K2_TX - 1) / K2_TX, nty = (e.ny + K2_TY - 1) / K2_TY;
  int lid = blockIdx.x;
  if ((gridDim.x & 7) == 0) lid = (lid & 7) * (gridDim.x >> 3) + (lid >> 3);
  const int b = lid / (ntx * nty), tl = lid - b * (ntx * nty), tyi = tl / ntx, txi = tl - tyi * ntx;
  const int tid = threadIdx.x;
  const int gx0 = txi * K2_TX - H, gy0 = tyi * K2_TY - H;
  // part of the region that lies inside the domain (local coordinates); neighbour indices clamp to it, which IS
  // the zero-flux rule on a domain edge and only feeds halo cells (never used) on an inner tile edge
  const int lox = max(0, -gx0), hix = min(RX, e.nx - gx0) - 1;
  const int loy = max(0, -gy0), hiy = min(RY, e.ny - gy0) - 1;
  const size_t fo = (size_t)b * e.ny * e.nx;

  // per strip: pairs (c0,c1), (c2,c3) of u and of v
  V2<T> u0[NS][2], v0[NS][2], au[NS][2], av[NS][2], cu[NS][2], cv[NS][2], pp[NS][2];
  int sr[NS], sc[NS];
  bool act[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int id = tid + k * K2_NT;
    sr[k] = id / SW;
    sc[k] = 4 * (id - sr[k] * SW);
    act[k] = id < NSTRIP && sr[k] >= loy && sr[k] <= hiy && sc[k] >= lox && sc[k] <= hix;
    if (act[k]) {
      const size_t g = fo + (size_t)(gy0 + sr[k]) * e.nx + (gx0 + sc[k]);
      const Quad<T> q = *reinterpret_cast<const Quad<T>*>(y_in + g);
      QuadT<T> qp;
      if (MODE == 2) {
        const int4 ca = *reinterpret_cast<const int4*>(e.cell_act + (size_t)(gy0 + sr[k]) * e.nx + (gx0 + sc[k]));
        const T* ab = p_in + (size_t)b * e.A;
        qp.a.x = ca.x >= 0 ? e.agent_power * ab[ca.x] : (T)0;
        qp.a.y = ca.y >= 0 ? e.agent_power * ab[ca.y] : (T)0;
        qp.b.x = ca.z >= 0 ? e.agent_power * ab[ca.z] : (T)0;
        qp.b.y = ca.w >= 0 ? e.agent_power * ab[ca.w] : (T)0;
      } else {
        qp = *reinterpret_cast<const QuadT<T>*>(p_in + g);
      }
      u0[k][0].x = q.c[0].x; u0[k][0].y = q.c[1].x; u0[k][1].x = q.c[2].x; u0[k][1].y = q.c[3].x;
      v0[k][0].x = q.c[0].y; v0[k][0].y = q.c[1].y; v0[k][1].x = q.c[2].y; v0[k][1].y = q.c[3].y;
      pp[k][0] = qp.a; pp[k][1] = qp.b;
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) { u0[k][i] = (V2<T>)(T)0; v0[k][i] = (V2<T>)(T)0; pp[k][i] = (V2<T>)(T)0; }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) { cu[k][i] = u0[k][i]; cv[k][i] = v0[k][i]; }
  }
  const T h = e.hstep;
  const int nrep = PROBE ? last : NSUB;
  for (int sub = 0; sub < nrep; ++sub) {
    if (PROBE) {      // keep the compiler from hoisting the strips' LDS offsets out of the runtime loop (42 spilled registers otherwise)
#pragma unroll
      for (int k = 0; k < NS; ++k) asm volatile("" : "+v"(sr[k]), "+v"(sc[k]));
    }
#pragma unroll
    for (int stage = 0; stage < 4; ++stage) {
      // publish the stage values
      if (sub + stage > 0) __syncthreads();
#pragma unroll
      for (int k = 0; k < NS; ++k)
        if (act[k]) {
          QuadT<T> q;
          q.a = cu[k][0]; q.b = cu[k][1];
          *reinterpret_cast<QuadT<T>*>(Su + sr[k] * RXP + sc[k]) = q;
          q.a = cv[k][0]; q.b = cv[k][1];
          *reinterpret_cast<QuadT<T>*>(Sv + sr[k] * RXP + sc[k]) = q;
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (!act[k]) continue;
        const int r = sr[k], c0 = sc[k];
        const int ow = r * RXP + max(c0 - 1, lox), oe = r * RXP + min(c0 + 4, hix);
        const int os = max(r - 1, loy) * RXP + c0, on = min(r + 1, hiy) * RXP + c0;
        const QuadT<T> us = *reinterpret_cast<const QuadT<T>*>(Su + os), un = *reinterpret_cast<const QuadT<T>*>(Su + on);
        const QuadT<T> vs = *reinterpret_cast<const QuadT<T>*>(Sv + os), vn = *reinterpret_cast<const QuadT<T>*>(Sv + on);
        // west / east neighbours of the two pairs: (W,c0) | (c1,c2) | (c3,E)
        V2<T> uA, uM, uB, vA, vM, vB;
        uA.x = Su[ow]; uA.y = cu[k][0].x; uM.x = cu[k][0].y; uM.y = cu[k][1].x; uB.x = cu[k][1].y; uB.y = Su[oe];
        vA.x = Sv[ow]; vA.y = cv[k][0].x; vM.x = cv[k][0].y; vM.y = cv[k][1].x; vB.x = cv[k][1].y; vB.y = Sv[oe];
        V2<T> ku[2], kv[2];
        k2_rhs_pair<T>(e, cu[k][0], cv[k][0], uA, uM, us.a, un.a, vA, vM, vs.a, vn.a, pp[k][0], ku[0], kv[0]);
        k2_rhs_pair<T>(e, cu[k][1], cv[k][1], uM, uB, us.b, un.b, vM, vB, vs.b, vn.b, pp[k][1], ku[1], kv[1]);
        if (MODE == 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) { u0[k][i] = ku[i]; v0[k][i] = kv[i]; }
          continue;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (stage == 0) {
            au[k][i] = ku[i]; av[k][i] = kv[i];
            cu[k][i] = u0[k][i] + ((T)0.5 * h) * ku[i]; cv[k][i] = v0[k][i] + ((T)0.5 * h) * kv[i];
          } else if (stage == 1) {
            au[k][i] = au[k][i] + (T)2 * ku[i]; av[k][i] = av[k][i] + (T)2 * kv[i];
            cu[k][i] = u0[k][i] + ((T)0.5 * h) * ku[i]; cv[k][i] = v0[k][i] + ((T)0.5 * h) * kv[i];
          } else if (stage == 2) {
            au[k][i] = au[k][i] + (T)2 * ku[i]; av[k][i] = av[k][i] + (T)2 * kv[i];
            cu[k][i] = u0[k][i] + h * ku[i]; cv[k][i] = v0[k][i] + h * kv[i];
          } else {
            u0[k][i] = u0[k][i] + (h / (T)6) * (au[k][i] + ku[i]); v0[k][i] = v0[k][i] + (h / (T)6) * (av[k][i] + kv[i]);
            cu[k][i] = u0[k][i]; cv[k][i] = v0[k][i];
          }
        }
      }
      if (MODE == 1) break;
    }
    if (MODE == 1) break;
  }
  // interior of the tile -> HBM
  bool bad = false;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    if (!act[k] || sr[k] < H || sr[k] >= H + K2_TY || sc[k] < H || sc[k] >= H + K2_TX) continue;
    // the address is computed again from the strip's row: otherwise the compiler keeps the prologue's 64-bit offsets alive
    // through the four stages (one spilled register with its reload in front of this store in the MODE 2 form)
    int row = sr[k];
    asm volatile("" : "+v"(row));
    const size_t g = fo + (size_t)(gy0 + row) * e.nx + (gx0 + sc[k]);
    Quad<T> q;
    q.c[0] = mk<T>(u0[k][0].x, v0[k][0].x); q.c[1] = mk<T>(u0[k][0].y, v0[k][0].y);
    q.c[2] = mk<T>(u0[k][1].x, v0[k][1].x); q.c[3] = mk<T>(u0[k][1].y, v0[k][1].y);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      // blow-up test max|y| > max_value (src/PDEenv.jl:227); NaN also raises the flag (as in the 1-D kernels)
      bad |= !(fabs(q.c[i].x) <= e.max_value && fabs(q.c[i].y) <= e.max_value);
    *reinterpret_cast<Quad<T>*>(y_out + g) = q;
  }
  if (MODE != 1 && !PROBE && last && done && e.check_max == 1) {
    if (__any(bad) && (tid & 63) == 0) atomicOr(done + b, 1);
  }
}

// p = sum_i agent_power * action[i] * box_i   (KellerSegelSetup.jl:318-332; boxes do not overlap)
template <class T>
__global__ void kseg2d_actuate_kernel(K2Dev<T> e, const T* __restrict__ action, T* __restrict__ p) {
  const size_t cells = (size_t)e.ny * e.nx, i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cells * e.B) return;
  const size_t b = i / cells, c = i - b * cells;
  const int a = e.cell_act[c];
  p[i] = a >= 0 ? e.agent_power * action[b * e.A + a] : (T)0;
}

// box sums of u and v for every sensor: sums [B][2][S].  One wave per (trajectory, sensor row): the rows of that sensor row's
// boxes are read once, whole and coalesced (four cells = 32 / 64 bytes per lane), into per-column sums in LDS, and each sensor
// adds its own columns -- round 4; one thread per sensor walking its box cell by cell took 45 us at config C4 (67 MB: ~11 us).
// Order of a box's additions: rows top to bottom inside a column, then the columns left to right.
template <class T>
__global__ __launch_bounds__(64) void kseg2d_boxsum_kernel(K2Dev<T> e, const C2<T>* __restrict__ y, T* __restrict__ sums) {
  extern __shared__ __align__(16) unsigned char k2bs_smem[];
  T* cu = reinterpret_cast<T*>(k2bs_smem);          // [nx] column sums of u over the box rows
  T* cv = cu + e.nx;                                // [nx] ... of v
  const int b = blockIdx.x / e.Sy, iy = blockIdx.x - b * e.Sy, lane = threadIdx.x;
  const int cy = e.sy[iy], r0 = max(cy - e.hw, 0), r1 = min(cy + e.hw, e.ny - 1);
  const C2<T>* yb = y + (size_t)b * e.ny * e.nx;
  for (int c = 4 * lane; c < e.nx; c += 256) {      // nx is a multiple of 4
    T su[4] = {0, 0, 0, 0}, sv[4] = {0, 0, 0, 0};
    for (int r = r0; r <= r1; ++r) {
      const Quad<T> q = *reinterpret_cast<const Quad<T>*>(yb + (size_t)r * e.nx + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) { su[i] += q.c[i].x; sv[i] += q.c[i].y; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { cu[c + i] = su[i]; cv[c + i] = sv[i]; }
  }
  __syncthreads();
  for (int ix = lane; ix < e.Sx; ix += 64) {
    const int cx = e.sx[ix], c0 = max(cx - e.hw, 0), c1 = min(cx + e.hw, e.nx - 1);
    T su = 0, sv = 0;
    for (int c = c0; c <= c1; ++c) { su += cu[c]; sv += cv[c]; }
    const int sidx = iy * e.Sx + ix;
    sums[((size_t)b * 2 + 0) * e.S + sidx] = su;
    sums[((size_t)b * 2 + 1) * e.S + sidx] = sv;
  }
}

// reward (KellerSegelSetup.jl:241-257) and featurize (:265-316, 3x3 circular window in the
// scripts/Fluid/setup/FluidSetup.jl:219-223 shift order): one thread per STATE ELEMENT (a, row) of trajectory blockIdx.y, so
// consecutive lanes write consecutive addresses; where an element comes from is a table made with the environment (ftab: the
// per-element index arithmetic -- five divisions by run-time values -- took 73 - 125 us at config C4); the first A threads
// of a trajectory also evaluate the rewards.
template <class T>
__global__ void kseg2d_feat_kernel(K2Dev<T> e, const T* __restrict__ sums, const T* __restrict__ action,
                                   const T* __restrict__ action_prev, const T* __restrict__ state_prev,
                                   T* __restrict__ state_out, T* __restrict__ reward_out, int32_t* __restrict__ done) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, per = e.A * e.ns, b = blockIdx.y;
  if (idx >= per) return;
  const T* sb = sums + (size_t)b * 2 * e.S;
  if (reward_out && idx < e.A) {
    const int i = b * e.A + idx;
    const T d = e.r_in_scale * (sb[e.a2s[idx]] + e.r_offset * (T)e.acnt[idx]);
    const T av = action[i], da = av - action_prev[i];
    const T r = -k2_pow_abs<T>(d, e.r_power) / e.r_denom - e.a_pun * av * av - e.da_pun * da * da;
    reward_out[i] = r;
    if (done && e.check_max == 2 && !(fabs(r) <= e.max_value)) atomicOr(done + b, 1);
  }
  if (state_out) {
    const int t = e.ftab[idx];
    const size_t gi = (size_t)b * per + idx;
    // a row of the temporal stack: the previous state's row `fresh` places up (fresh = 2 window^2); reset form: the fresh value again
    state_out[gi] = (t < 0 && state_prev) ? state_prev[gi - 2 * e.window * e.window] : sb[t & 0x7fffffff] * e.sensor_scale;
  }
}

// per-column terminal flags for the DDPG batch (runs after every kernel that may raise done[b])
template <class T>
__global__ void kseg2d_terminal_kernel(K2Dev<T> e, const int32_t* __restrict__ done) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e.B * e.A) return;
  e.term_out[i] = done[i / e.A] ? (T)1 : (T)0;
}

struct Kseg2dEnv : Env {
  int nx = 0, ny = 0, Sx = 0, Sy = 0, hw = 0, nsub = 1;
  DevBuf sx, sy, a2s_d, cell_act, ftab, acnt, sums, pbuf, ytmp, done_tmp;
  size_t lds1 = 0, lds2 = 0;
  static constexpr int MAXPART = PartStreams::MAX;   // parts of the batch on their own streams during the RK4 sub-steps (k2_integrate)
  PartStreams ps;                                    // common.hpp: the caller's or library-made streams, fork / join events
  int part_streams() const override;
  int set_part_streams(const hipStream_t* s, int n) override;
};

static Kseg2dEnv& as_k2(Env& E) { return static_cast<Kseg2dEnv&>(E); }

template <class T>
static K2Dev<T> k2_dev(const Kseg2dEnv& E) {
  const pdec_env_cfg& c = E.cfg;
  K2Dev<T> d;
  d.B = c.B; d.nx = E.nx; d.ny = E.ny; d.K = c.K; d.Sx = E.Sx; d.Sy = E.Sy; d.S = c.S; d.A = c.A; d.hw = E.hw;
  d.window = c.window; d.temporal = c.temporal_steps; d.ns = 2 * c.window * c.window * c.temporal_steps;
  d.check_max = c.check_max_value;
  const double dx = c.Lx / E.nx;
  d.idx = (T)(1.0 / dx); d.idx2 = (T)(1.0 / (dx * dx)); d.c56 = (T)(5.6 * 0.25 / (dx * dx)); d.hstep = (T)(c.dt / c.K);
  d.sensor_scale = (T)c.sensor_scale; d.agent_power = (T)c.agent_power; d.r_in_scale = (T)c.reward_in_scale;
  d.r_offset = (T)c.reward_offset; d.r_power = (T)c.reward_power; d.r_denom = (T)c.reward_denom;
  d.a_pun = (T)c.action_punish; d.da_pun = (T)c.delta_action_punish; d.max_value = (T)c.max_value;
  d.ftab = E.ftab.as<int>(); d.acnt = E.acnt.as<int>();
  d.sx = E.sx.as<int>(); d.sy = E.sy.as<int>(); d.a2s = E.a2s_d.as<int>(); d.cell_act = E.cell_act.as<int>();
  d.term_out = (T*)E.term_out;
  return d;
}

template <int NSUB>
static constexpr size_t k2_lds(size_t pair_bytes) {   // two planes [RY][RX] of T
  return (size_t)((pair_bytes == 8 ? 64 : 32) + 8 * NSUB) * (K2_TX + 8 * NSUB) * pair_bytes;
}

// b0, nb: the trajectories [b0, b0 + nb) of the batch on stream `st` (the whole batch on the environment's stream by default)
template <class T, int NSUB, int MODE>
static int k2_launch_rk4(Kseg2dEnv& E, const void* y_in, const void* p, void* y_out, int32_t* done, int last, int b0 = 0, int nb = -1,
                         hipStream_t st = nullptr) {
  constexpr int K2_TY = K2Tile<T>::TY, K2_NT = K2Tile<T>::NT;
  if (nb < 0) { nb = E.cfg.B; st = E.stream; }
  const dim3 grid(((E.nx + K2_TX - 1) / K2_TX) * ((E.ny + K2_TY - 1) / K2_TY) * nb);
  const size_t cells = (size_t)E.ny * E.nx;
  const T* pf = (const T*)p + (size_t)b0 * (MODE == 2 ? (size_t)E.cfg.A : cells);
  hipLaunchKernelGGL((kseg2d_rk4_kernel<T, NSUB, MODE>), grid, dim3(K2_NT), k2_lds<NSUB>(2 * sizeof(T)), st, k2_dev<T>(E),
                     (const C2<T>*)y_in + (size_t)b0 * cells, pf, (C2<T>*)y_out + (size_t)b0 * cells, done ? done + b0 : nullptr, last);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

// Parts of the batch that the RK4 sub-steps run on streams of their own (fp32, one sub-step per launch).  Only when each part
// still fills the chip (>= 512 tiles); PDEC_KSEG2D_SPLIT=0 off, 1 / 2 halves, 3 / 4 that many parts.  Default: three parts
// (C4: 43 / 43 / 42 trajectories, 75.2 k env-steps/s against 72.1 k with halves and 71.1 k unsplit; FOUR parts -- with the
// update's five busy streams on the four compute pipes, see PartStreams in common.hpp -- fall to 65 k).
static int k2_parts(const Kseg2dEnv& E) {
  static const char* split_env = getenv("PDEC_KSEG2D_SPLIT");
  const int tiles_all = ((E.nx + K2_TX - 1) / K2_TX) * ((E.ny + K2Tile<float>::TY - 1) / K2Tile<float>::TY) * E.cfg.B;
  int np = split_env ? atoi(split_env) : (tiles_all >= 1536 ? 3 : (tiles_all >= 1024 ? 2 : 0));
  if (np == 1) np = 2;
  np = std::min(std::min(np, (int)Kseg2dEnv::MAXPART), E.cfg.B);
  if (E.ps.given >= 0) np = std::min(np, E.ps.given + 1);     // the caller's part streams: no more parts than it handed over
  return np >= 2 ? np : 0;
}

// the part streams / events of a split step (PartStreams::ensure); refuses under stream capture when they do not exist yet
static int k2_make_part_streams(Kseg2dEnv& E, int np) {
  bool refused = false;
  PDEC_HIP(E.ps.ensure(E.stream, np, &refused));
  PDEC_REQUIRE(!refused, "2-D Keller-Segel step: its %d part streams do not exist yet and cannot be made while the environment's "
               "stream is being captured; run one step (or pdec_env_set_part_streams) before the capture, or PDEC_KSEG2D_SPLIT=0", np - 1);
  return PDEC_OK;
}

int Kseg2dEnv::part_streams() const {
  if (cfg.dtype != PDEC_F32 || nsub != 1) return 0;
  const int np = k2_parts(*this);
  return np ? np - 1 : 0;
}

int Kseg2dEnv::set_part_streams(const hipStream_t* s, int n) {
  PDEC_HIP(ps.give(s, n));
  return PDEC_OK;
}

template <class T>
static int k2_integrate(Kseg2dEnv& E, const void* y_in, const void* p, const void* action, void* y_out, int32_t* done) {
  // K sub-steps, ping-pong between y_out and the scratch field so that the last launch writes y_out; with `action`
  // the forcing is synthesised in the kernel (MODE 2) instead of being read from the p field
  const int K = E.cfg.K, ns = E.nsub;
  const int launches = (K + ns - 1) / ns;
  const size_t bytes = (size_t)E.cfg.B * E.ny * E.nx * 2 * sizeof(T);
  if (launches > 1 && E.ytmp.bytes < bytes) PDEC_HIP(E.ytmp.alloc(bytes));
  if (done) PDEC_HIP(hipMemsetAsync(done, 0, sizeof(int32_t) * E.cfg.B, E.stream));
  ProfScope ps(&E, "kseg2d_rk4");
  const void* src = y_in;
  const void* f = action ? action : p;
  int left = K;
  // Round 4: the batch in parts on their own streams (k2_parts) -- trajectories are independent, so sub-step k + 1 of one part
  // starts while sub-step k of another still drains its last workgroups (a launch is 2 048 workgroups on 512 slots: its tail
  // runs at falling occupancy 32 times per control step).  Same fields bit for bit.  (Per-kernel timing passes,
  // pdec_prof_enable, take the unsplit path: one event pair around the whole sub-step loop of one stream.)
  const int np = sizeof(T) == 4 && ns == 1 && !E.prof ? k2_parts(E) : 0;
  if (np >= 2) {
    { const int rc = k2_make_part_streams(E, np); if (rc) return rc; }
    {   // a fork that fails midway has already made some part streams wait: join them before returning
      const hipError_t ef = E.ps.fork(E.stream, np);
      if (ef != hipSuccess) { (void)E.ps.join(E.stream, np); PDEC_HIP(ef); }
    }
    int rc_launch = PDEC_OK;
    for (int l = 0; l < launches && !rc_launch; ++l) {
      void* dst = ((launches - 1 - l) & 1) ? E.ytmp.p : y_out;
      const int last = l == launches - 1;
      if constexpr (sizeof(T) == 4) {
        int b0 = 0, left_b = E.cfg.B;
        for (int i = 0; i < np; ++i) {
          const int nb = left_b / (np - i);
          hipStream_t st = i == 0 ? E.stream : E.ps.st[i];
          const int rc = action ? k2_launch_rk4<T, 1, 2>(E, src, f, dst, done, last, b0, nb, st) : k2_launch_rk4<T, 1, 0>(E, src, f, dst, done, last, b0, nb, st);
          if (rc) { rc_launch = rc; break; }          // (the part streams are joined below on this path too)
          b0 += nb;
          left_b -= nb;
        }
      }
      src = dst;
    }
    const hipError_t ej = E.ps.join(E.stream, np);
    if (rc_launch) return rc_launch;
    PDEC_HIP(ej);
    return PDEC_OK;
  }
  for (int l = 0; l < launches; ++l) {
    void* dst = ((launches - 1 - l) & 1) ? E.ytmp.p : y_out;
    const int last = l == launches - 1;
    int rc;
    const bool two = sizeof(T) == 4 && left >= 2 && ns == 2;
    if constexpr (sizeof(T) == 4) {
      if (two) rc = action ? k2_launch_rk4<T, 2, 2>(E, src, f, dst, done, last) : k2_launch_rk4<T, 2, 0>(E, src, f, dst, done, last);
      else rc = action ? k2_launch_rk4<T, 1, 2>(E, src, f, dst, done, last) : k2_launch_rk4<T, 1, 0>(E, src, f, dst, done, last);
    } else {
      rc = action ? k2_launch_rk4<T, 1, 2>(E, src, f, dst, done, last) : k2_launch_rk4<T, 1, 0>(E, src, f, dst, done, last);
    }
    left -= two ? 2 : 1;
    if (rc) return rc;
    src = dst;
  }
  return PDEC_OK;
}

template <class T>
static int k2_actuate(Kseg2dEnv& E, const void* action, void* p_out) {
  const size_t n = (size_t)E.cfg.B * E.ny * E.nx;
  hipLaunchKernelGGL(kseg2d_actuate_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, E.stream, k2_dev<T>(E),
                     (const T*)action, (T*)p_out);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

template <class T>
static int k2_sense(Kseg2dEnv& E, const void* y, const void* action, const void* action_prev, const void* state_prev,
                    void* state_out, void* reward_out, int32_t* done) {
  const size_t need = (size_t)E.cfg.B * 2 * E.cfg.S * sizeof(T);
  if (E.sums.bytes < need) PDEC_HIP(E.sums.alloc(need));
  hipLaunchKernelGGL(kseg2d_boxsum_kernel<T>, dim3(E.cfg.B * E.Sy), dim3(64), 2 * (size_t)E.nx * sizeof(T), E.stream, k2_dev<T>(E),
                     (const C2<T>*)y, E.sums.as<T>());
  PDEC_HIP(hipGetLastError());
  const int per = E.cfg.A * (2 * E.cfg.window * E.cfg.window * E.cfg.temporal_steps);
  hipLaunchKernelGGL(kseg2d_feat_kernel<T>, dim3((per + 255) / 256, E.cfg.B), dim3(256), 0, E.stream, k2_dev<T>(E), E.sums.as<T>(),
                     (const T*)action, (const T*)action_prev, (const T*)state_prev, (T*)state_out, (T*)reward_out, done);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

#ifdef PDEC_DEBUG_PROBES
// Timing probe for the time-resident design question (HISTORY.md round 5): the fp32 tile kernel on `nb` trajectories with
// `reps` sub-steps per launch, `iters` launches between two events on the environment's stream -> microseconds per launch.
static int k2_probe(Kseg2dEnv& E, int nb, int reps, int iters, double* us) {
  PDEC_REQUIRE(E.cfg.dtype == PDEC_F32 && nb >= 1 && nb <= E.cfg.B && reps >= 1 && iters >= 1, "pdec_debug_kseg2d_probe: fp32 environments, 1 <= nb <= B");
  const size_t bytes = (size_t)E.cfg.B * E.ny * E.nx * 2 * sizeof(float);
  if (E.ytmp.bytes < bytes) PDEC_HIP(E.ytmp.alloc(bytes));
  DevBuf src, act;
  PDEC_HIP(src.alloc(bytes));
  PDEC_HIP(act.alloc((size_t)E.cfg.B * E.cfg.A * sizeof(float)));
  PDEC_HIP(hipMemsetAsync(src.p, 0, bytes, E.stream));
  PDEC_HIP(hipMemsetAsync(act.p, 0, (size_t)E.cfg.B * E.cfg.A * sizeof(float), E.stream));
  constexpr int K2_TY = K2Tile<float>::TY, K2_NT = K2Tile<float>::NT;
  const dim3 grid(((E.nx + K2_TX - 1) / K2_TX) * ((E.ny + K2_TY - 1) / K2_TY) * nb);
  hipEvent_t e0, e1;
  PDEC_HIP(hipEventCreate(&e0));
  PDEC_HIP(hipEventCreate(&e1));
  for (int it = -2; it < iters; ++it) {
    if (it == 0) PDEC_HIP(hipEventRecord(e0, E.stream));
    hipLaunchKernelGGL((kseg2d_rk4_kernel<float, 1, 2, true>), grid, dim3(K2_NT), k2_lds<1>(2 * sizeof(float)), E.stream, k2_dev<float>(E),
                       src.as<C2<float>>(), act.as<float>(), E.ytmp.as<C2<float>>(), (int32_t*)nullptr, reps);
  }
  PDEC_HIP(hipEventRecord(e1, E.stream));
  PDEC_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  PDEC_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *us = (double)ms * 1e3 / iters;
  return PDEC_OK;
}
#endif

#define K2_DISPATCH(fn, ...) (E.cfg.dtype == PDEC_F64 ? fn<double>(E, __VA_ARGS__) : fn<float>(E, __VA_ARGS__))

int kseg2d_actuate(Env& E0, const void* action, void* p_out) {
  Kseg2dEnv& E = as_k2(E0);
  return K2_DISPATCH(k2_actuate, action, p_out);
}

int kseg2d_featurize(Env& E0, const void* y, const void* state_prev, void* state_out) {
  Kseg2dEnv& E = as_k2(E0);
  return K2_DISPATCH(k2_sense, y, nullptr, nullptr, state_prev, state_out, nullptr, nullptr);
}

int kseg2d_reward(Env& E0, const void* y, const void* action, const void* action_prev, void* r_out) {
  Kseg2dEnv& E = as_k2(E0);
  return K2_DISPATCH(k2_sense, y, action, action_prev, nullptr, nullptr, r_out, nullptr);
}

int kseg2d_rhs_eval(Env& E0, const void* y, const void* p, void* out) {
  Kseg2dEnv& E = as_k2(E0);
  return E.cfg.dtype == PDEC_F64 ? k2_launch_rk4<double, 1, 1>(E, y, p, out, nullptr, 0)
                                 : k2_launch_rk4<float, 1, 1>(E, y, p, out, nullptr, 0);
}

int kseg2d_pde_step(Env& E0, const void* y_in, const void* p, void* y_out, int32_t* done) {
  Kseg2dEnv& E = as_k2(E0);
  PDEC_REQUIRE(y_in != y_out, "kseg2d: y_out must not alias y_in");
  return K2_DISPATCH(k2_integrate, y_in, p, nullptr, y_out, done);
}

int kseg2d_env_step(Env& E0, const void* y_in, const void* action, const void* action_prev, const void* state_prev,
                    void* y_out, void* p_out, void* state_out, void* reward_out, int32_t* done) {
  Kseg2dEnv& E = as_k2(E0);
  const size_t ts = dtype_size(E.cfg.dtype);
  void* ph = p_out;
  if (!ph) {
    const size_t need = (size_t)E.cfg.B * E.ny * E.nx * ts;
    if (E.pbuf.bytes < need) PDEC_HIP(E.pbuf.alloc(need));
    ph = E.pbuf.p;
  }
  int32_t* dn = done;
  if (!dn && E.term_out) {
    if (E.done_tmp.bytes < sizeof(int32_t) * E.cfg.B) PDEC_HIP(E.done_tmp.alloc(sizeof(int32_t) * E.cfg.B));
    dn = E.done_tmp.as<int32_t>();
  }
  int rc;
  if ((rc = kseg2d_actuate(E0, action, ph))) return rc;                                 // src/PDEenv.jl:199
  PDEC_REQUIRE(y_in != y_out, "kseg2d: y_out must not alias y_in");
  // fp64 (HBM-bound): the forcing is synthesised in the kernel from the action table; fp32: reading the p field
  // measured faster than the two-level gather (PDEC_KSEG2D_GATHER=0/1 overrides)
  static const char* gv = getenv("PDEC_KSEG2D_GATHER");
  const bool gather = gv ? gv[0] == '1' : E.cfg.dtype == PDEC_F64;
  if ((rc = K2_DISPATCH(k2_integrate, y_in, ph, gather ? action : nullptr, y_out, dn))) return rc;   // :216-218 (zeroes done)
  if ((rc = K2_DISPATCH(k2_sense, y_out, action, action_prev, state_prev, state_out, reward_out, dn))) return rc;   // :220-222
  if (E.term_out && dn) {
    const int nA = E.cfg.B * E.cfg.A;
    if (E.cfg.dtype == PDEC_F64)
      hipLaunchKernelGGL(kseg2d_terminal_kernel<double>, dim3((nA + 255) / 256), dim3(256), 0, E.stream, k2_dev<double>(E), dn);
    else
      hipLaunchKernelGGL(kseg2d_terminal_kernel<float>, dim3((nA + 255) / 256), dim3(256), 0, E.stream, k2_dev<float>(E), dn);
    PDEC_HIP(hipGetLastError());
  }
  return PDEC_OK;
}

}  // namespace pdec

using namespace pdec;

extern "C" int pdec_kseg2d_env_create(pdec_handle* h, const pdec_env_cfg* cfg, int ny, int Sx, int Sy,
                                      const int32_t* sensor_x, const int32_t* sensor_y, int half_window,
                                      const int32_t* a2s) {
  PDEC_REQUIRE(h && cfg && sensor_x && sensor_y && a2s, "pdec_kseg2d_env_create: null argument");
  const pdec_env_cfg& c = *cfg;
  PDEC_REQUIRE(c.pde_kind == PDEC_PDE_KSEG2D_RK4, "pdec_kseg2d_env_create: pde_kind must be PDEC_PDE_KSEG2D_RK4");
  PDEC_REQUIRE(c.dtype == PDEC_F32 || c.dtype == PDEC_F64, "pdec_kseg2d_env_create: bad dtype %d", c.dtype);
  PDEC_REQUIRE(c.B >= 1 && c.B <= 65535 && c.N >= 4 && c.N % 4 == 0 && ny >= 1 && c.K >= 1,
               "pdec_kseg2d_env_create: bad sizes B=%d nx=%d (multiple of 4) ny=%d K=%d", c.B, c.N, ny, c.K);
  PDEC_REQUIRE(c.n_species == 2 && !c.mono, "Keller-Segel has two species and no mono variant");
  PDEC_REQUIRE(c.memory_size == 0, "pdec_kseg2d_env_create: action memory is not built for the 2-D grid (a configuration of this build, not of the reference)");
  PDEC_REQUIRE(Sx >= 1 && Sy >= 1 && Sx * Sy == c.S && c.A >= 1 && half_window >= 0,
               "pdec_kseg2d_env_create: S must equal Sx*Sy");
  PDEC_REQUIRE(c.window >= 1 && (c.window & 1) && c.temporal_steps >= 1, "pdec_kseg2d_env_create: window must be odd >= 1");
  PDEC_REQUIRE(c.Lx > 0 && c.dt > 0, "pdec_kseg2d_env_create: Lx and dt must be positive");
  for (int i = 0; i < Sx; ++i) PDEC_REQUIRE(sensor_x[i] >= 0 && sensor_x[i] < c.N, "sensor_x[%d] out of range", i);
  for (int i = 0; i < Sy; ++i) PDEC_REQUIRE(sensor_y[i] >= 0 && sensor_y[i] < ny, "sensor_y[%d] out of range", i);
  for (int a = 0; a < c.A; ++a) PDEC_REQUIRE(a2s[a] >= 0 && a2s[a] < c.S, "pdec_kseg2d_env_create: a2s[%d] out of range", a);
  auto E = std::make_unique<Kseg2dEnv>();
  E->cfg = c;
  E->cfg.Ny = ny;
  E->nx = c.N; E->ny = ny; E->Sx = Sx; E->Sy = Sy; E->hw = half_window;
  // One RK4 sub-step per launch (halo 4): measured faster than two fused sub-steps (halo 8, half the HBM traffic but
  // 1.56x instead of 1.27x redundant cells and too many registers for two workgroups per CU); PDEC_KSEG2D_NSUB2=1
  // selects the two-sub-step variant (fp32 only).
  E->nsub = (c.dtype == PDEC_F32 && getenv("PDEC_KSEG2D_NSUB2")) ? 2 : 1;
  std::vector<int32_t> ca((size_t)ny * c.N, -1);
  for (int a = 0; a < c.A; ++a) {
    const int s = a2s[a], iy = s / Sx, ix = s % Sx;
    for (int r = std::max(sensor_y[iy] - half_window, 0); r <= std::min(sensor_y[iy] + half_window, ny - 1); ++r)
      for (int q = std::max(sensor_x[ix] - half_window, 0); q <= std::min(sensor_x[ix] + half_window, c.N - 1); ++q) {
        PDEC_REQUIRE(ca[(size_t)r * c.N + q] < 0, "pdec_kseg2d_env_create: actuator boxes %d and %d overlap", ca[(size_t)r * c.N + q], a);
        ca[(size_t)r * c.N + q] = a;
      }
  }
  auto up_i = [](DevBuf& b, const int32_t* src, size_t cnt) -> int {
    PDEC_HIP(b.alloc(sizeof(int32_t) * cnt));
    PDEC_HIP(hipMemcpy(b.p, src, sizeof(int32_t) * cnt, hipMemcpyHostToDevice));
    return PDEC_OK;
  };
  int rc;
  if ((rc = up_i(E->sx, sensor_x, Sx))) return rc;
  if ((rc = up_i(E->sy, sensor_y, Sy))) return rc;
  if ((rc = up_i(E->a2s_d, a2s, c.A))) return rc;
  if ((rc = up_i(E->cell_act, ca.data(), ca.size()))) return rc;
  {   // featurize map and box cell counts (kseg2d_feat_kernel)
    const int w = c.window / 2, ww = c.window * c.window, fresh = 2 * ww, ns = fresh * c.temporal_steps;
    std::vector<int32_t> ft((size_t)c.A * ns), cnt(c.A);
    for (int a = 0; a < c.A; ++a) {
      const int s = a2s[a], iy = s / Sx, ix = s % Sx;
      const int cy = sensor_y[iy], cx = sensor_x[ix];
      cnt[a] = (std::min(cy + half_window, ny - 1) - std::max(cy - half_window, 0) + 1) *
               (std::min(cx + half_window, c.N - 1) - std::max(cx - half_window, 0) + 1);
      for (int rr = 0; rr < ns; ++rr) {
        const int r0 = rr % fresh, sp = r0 / ww, q = r0 - sp * ww;
        const int di = q / c.window - w, dj = q % c.window - w;
        // circshift(sensors, [di, dj])[iy, ix] = sensors[iy - di, ix - dj]
        const int jy = (((iy - di) % Sy) + Sy) % Sy, jx = (((ix - dj) % Sx) + Sx) % Sx;
        ft[(size_t)a * ns + rr] = (sp * c.S + jy * Sx + jx) | (rr >= fresh ? (int32_t)0x80000000 : 0);
      }
    }
    if ((rc = up_i(E->ftab, ft.data(), ft.size()))) return rc;
    if ((rc = up_i(E->acnt, cnt.data(), cnt.size()))) return rc;
  }
  auto set_lds = [](const void* f, size_t bytes) -> int {
    PDEC_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return PDEC_OK;
  };
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<float, 1, 0>, k2_lds<1>(8)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<float, 2, 0>, k2_lds<2>(8)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<float, 1, 1>, k2_lds<1>(8)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<double, 1, 0>, k2_lds<1>(16)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<double, 1, 1>, k2_lds<1>(16)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<float, 1, 2>, k2_lds<1>(8)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<float, 2, 2>, k2_lds<2>(8)))) return rc;
  if ((rc = set_lds((const void*)kseg2d_rk4_kernel<double, 1, 2>, k2_lds<1>(16)))) return rc;
  // (the library's own part streams are made at the first split step, k2_make_part_streams: by then the environment's stream
  // -- whose priority level they take -- is known, and a caller that cares about where their queues land hands over its own)
  *h = register_object(std::move(E));
  return PDEC_OK;
}

// measurement aid (tools/kseg2d_resident_probe.py; not part of the reference's surface): see k2_probe
extern "C" int pdec_debug_kseg2d_probe(pdec_handle h, int nb, int reps, int iters, double* us_per_launch) {
  Env* E = lookup_as<Env>(h, Kind::Env);
  PDEC_REQUIRE(E && E->cfg.pde_kind == PDEC_PDE_KSEG2D_RK4 && us_per_launch, "pdec_debug_kseg2d_probe: a 2-D Keller-Segel environment handle");
#ifdef PDEC_DEBUG_PROBES
  return k2_probe(as_k2(*E), nb, reps, iters, us_per_launch);
#else
  (void)nb; (void)reps; (void)iters;
  set_error("pdec_debug_kseg2d_probe: this library was built without -DPDEC_DEBUG_PROBES (include/pdeconv_debug.h)");
  return PDEC_E_INVALID;
#endif
}
