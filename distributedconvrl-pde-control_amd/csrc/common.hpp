// common.hpp -- handle registry, error plumbing, per-kernel event timing for libpdeconv.so
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/pdeconv_debug.h"   // (includes pdeconv.h)

namespace pdec {

void set_error(const char* fmt, ...);

#define PDEC_HIP(call)                                                                  \
  do {                                                                                  \
    hipError_t e__ = (call);                                                            \
    if (e__ != hipSuccess) {                                                            \
      pdec::set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                      __LINE__);                                                        \
      return PDEC_E_HIP;                                                                \
    }                                                                                   \
  } while (0)

#define PDEC_REQUIRE(cond, ...)        \
  do {                                 \
    if (!(cond)) {                     \
      pdec::set_error(__VA_ARGS__);    \
      return PDEC_E_INVALID;           \
    }                                  \
  } while (0)

enum class Kind { Env, Mlp, Comm, Graph, Event };

struct ProfEntry {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  int reps = 1;   // launches bracketed by each event pair (replay timing, see ProfScope::reps)
};

// pdec_set_launch_sync: a device-side hand-over between two single-workgroup launches that sit on DIFFERENT streams.  The
// consumer launch spins (one thread, sleeping between polls, bounded by SYNC_TIMEOUT_TICKS of the 100-MHz counter) until
// *wait >= wait_val before it touches its inputs; the producer launch stores done_val to *done when its outputs are written
// (agent-scope release / acquire: the two workgroups may run on different XCDs).  A stream-level event pair costs the chain of
// the single-trajectory training loop ~11 us per hop (signal -> barrier packet of the other queue -> dispatch); this ~2 - 3 us.
struct LaunchSync {
  const long long* wait = nullptr;
  long long wait_val = 0;
  long long* done = nullptr;
  long long done_val = 0;
  int* timeouts = nullptr;       // device counter of waits that gave up (pdec_launch_sync_timeouts)
};
#define PDEC_SYNC_TIMEOUT_TICKS 30000000ull      // 0.3 s: a hand-over that has not come by then never will

// every thread of the workgroup calls it before the launch's first read of what the producer writes
__device__ __forceinline__ void launch_sync_wait(const LaunchSync& s) {
  if (!s.wait) return;
  if (threadIdx.x == 0) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(s.wait, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < s.wait_val) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > PDEC_SYNC_TIMEOUT_TICKS) {     // the exit every wait reaches
        if (s.timeouts) atomicAdd(s.timeouts, 1);
        break;
      }
    }
    __builtin_amdgcn_s_dcache_inv();
  }
  __syncthreads();
}
// every thread calls it behind the launch's last store
__device__ __forceinline__ void launch_sync_done(const LaunchSync& s) {
  if (!s.done) return;
  __syncthreads();               // the workgroup's stores have left for L2
  if (threadIdx.x == 0) __hip_atomic_store(s.done, s.done_val, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

struct Object {
  Kind kind;
  hipStream_t stream = nullptr;
  bool prof = false;
  int prof_reps = 1;   // pdec_prof_enable(h, R > 1): idempotent kernels are launched R times back to back
                       // inside one event pair, so the per-launch figure is free of event-record overhead
                       // and comparable with rocprofv3's kernel-trace duration
  std::map<std::string, ProfEntry> profs;
  // pdec_set_episode_halt: device flag of a speculatively issued episode (run.py, device-side episodes).  The launches that
  // change persistent learner state through this handle -- replay pushes, the small-batch DDPG update -- do nothing once it is
  // raised, and the POST_ACT push raises it when the environment reported the end of the episode (B = 1)
  int* halt = nullptr;
  // pdec_set_launch_sync: one-shot, consumed (and cleared) by the next launch through this handle that supports it
  LaunchSync sync;
  explicit Object(Kind k) : kind(k) {}
  virtual ~Object();
};

// RAII launch timer: records events around a launch when profiling is on
struct ProfScope {
  Object* o;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  const char* name;
  int reps = 1;   // the launch site loops `for (int i = 0; i < ps.reps; ++i) launch` when the kernel is idempotent
  ProfScope(Object* obj, const char* nm, bool idempotent = false) : o(obj), name(nm) {
    if (o->prof) {
      if (idempotent) reps = o->prof_reps;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      (void)hipEventRecord(e0, o->stream);
    }
  }
  ~ProfScope() {
    if (o->prof) {
      (void)hipEventRecord(e1, o->stream);
      ProfEntry& pe = o->profs[name];
      pe.ev.emplace_back(e0, e1);
      pe.reps = reps;
    }
  }
};

// One profiled launch timed by the DISPATCH's own begin / end timestamps (hipExtLaunchKernelGGL attaches the two events to
// the kernel packet): no marker packets in front of and behind the kernel, so the figure is what rocprofv3's kernel trace
// reports.  Event records around the launch (ProfScope) cost the kernel ~6 us of apparent duration in the two-stream
// pipeline (72.6 vs 65.7 us for the critic pass).  Used for the single-launch (reps == 1) profile of the two MFMA passes.
#define PDEC_TIMED_LAUNCH(obj, label, kern, grid, block, lds, ...)                                              \
  do {                                                                                                           \
    hipEvent_t e0__ = nullptr, e1__ = nullptr;                                                                   \
    (void)hipEventCreate(&e0__);                                                                                 \
    (void)hipEventCreate(&e1__);                                                                                 \
    hipExtLaunchKernelGGL(kern, grid, block, lds, (obj)->stream, e0__, e1__, 0, __VA_ARGS__);                    \
    ProfEntry& pe__ = (obj)->profs[label];                                                                       \
    pe__.ev.emplace_back(e0__, e1__);                                                                            \
    pe__.reps = 1;                                                                                               \
  } while (0)

// Host-side slot selectors of double-buffered device state (ADAM beta powers, noise counter, published actor image)
// flip once per launch that uses them.  A launch recorded into a HIP graph flips them at CAPTURE time only, so a
// replay has to repeat the flips: while a capture is open every flip is logged, and pdec_graph_launch re-applies the
// ones that occurred an odd number of times.
void note_flip(int* selector);
inline void flip(int& selector) {
  selector ^= 1;
  note_flip(&selector);
}

pdec_handle register_object(std::unique_ptr<Object> o);
Object* lookup(pdec_handle h);
template <class T>
T* lookup_as(pdec_handle h, Kind k) {
  Object* o = lookup(h);
  if (!o || o->kind != k) return nullptr;
  return static_cast<T*>(o);
}

hipEvent_t event_native(pdec_handle ev);      // replay.hip: the HIP event of an event handle, nullptr otherwise
int* launch_sync_timeout_counter();            // replay.hip: the device counter behind pdec_launch_sync_timeouts (nullptr: allocation failed)

// wave priority from a launch argument (s_setprio takes an immediate)
__device__ __forceinline__ void set_wave_prio(int p) {
  if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else if (p == 3) __builtin_amdgcn_s_setprio(3);
}
// wave priorities, overridable for experiments: PDEC_PRIO_KS (KS env-step kernel, default 1; 3 in its SIMD-sharing form, pdec_env_set_simd_sharing), PDEC_PRIO_MFMA (fused
// 3-layer DDPG passes, default 2)
inline int env_prio(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0] >= '0' && e[0] <= '3') ? e[0] - '0' : dflt;
}

inline size_t dtype_size(int dt) { return dt == PDEC_F64 ? 8 : 4; }

// device buffer with RAII
struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : p(o.p), bytes(o.bytes) {
    o.p = nullptr;
    o.bytes = 0;
  }
  ~DevBuf() { release(); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) return hipSuccess;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) {
      bytes = n;
      // diagnostic: PDEC_POISON=1 fills every fresh library buffer with 0xFF bytes (NaN as float / double), so that a
      // read of memory the library never wrote shows up in the results instead of depending on what was there before
      static const bool poison = getenv("PDEC_POISON") != nullptr;
      if (poison && (e = hipMemset(p, 0xFF, n)) == hipSuccess) e = hipStreamSynchronize(nullptr);
    }
    return e;
  }
  template <class T>
  T* as() const { return static_cast<T*>(p); }
};

// upload a host double array converted to dtype
int upload_converted(DevBuf& dst, const double* src, size_t n, int dtype);

// Streams for the parts of a batch that an environment splits over several streams (fluid.hip, kseg2d.hip); ONE helper for both
// environments.  Part 0 runs on the environment's own stream; parts 1 .. np-1 run on the caller's streams
// (pdec_env_set_part_streams) or, failing that, on streams the library makes AT THE FIRST SPLIT STEP at the priority level of the
// environment's stream -- parts of one level advance evenly (C4: 77 - 78 k env-steps/s against 74 k with the part streams one
// level below); the level is read again when the environment's stream changes (PDEC_PART_LEVEL pins it: -1 / 0 / 1).  WHERE a
// queue lands matters more than its level -- hardware queues sit on the GPU's four compute pipes in the order they are made, and
// two busy queues on one pipe take turns (include/pdeconv.h, pdec_stream_create); a caller that cares makes env / update / part
// streams back to back and hands the part streams over.
// Making streams and events is illegal while the environment's stream is being captured into a HIP graph: ensure() refuses
// with an error then instead of invalidating the capture (run one split step -- or pdec_env_set_part_streams -- before capturing).
// fork() / join() bracket the launches of a split step; join() is to be called on the error path as well, so that no part
// stream is left un-joined behind an early return.
struct PartStreams {
  static constexpr int MAX = 4;
  hipStream_t st[MAX] = {nullptr, nullptr, nullptr, nullptr};     // [0] unused
  bool own[MAX] = {false, false, false, false};                   // made by the library (else the caller's)
  int given = -1;                                                  // >= 0: the caller handed over that many
  int own_level = 0;                                               // priority level of the library-made streams ...
  void* level_of = (void*)-1;                                      // ... read from this environment stream
  hipEvent_t ev_fork = nullptr, ev_join[MAX] = {nullptr, nullptr, nullptr, nullptr};

  ~PartStreams() {
    for (int i = 0; i < MAX; ++i) {
      if (ev_join[i]) (void)hipEventDestroy(ev_join[i]);
      if (st[i] && own[i]) (void)hipStreamDestroy(st[i]);
    }
    if (ev_fork) (void)hipEventDestroy(ev_fork);
  }
  // the caller's streams replace the library's (n of them serve parts 1 .. n)
  hipError_t give(const hipStream_t* s, int n) {
    for (int i = 1; i < MAX; ++i) {
      if (st[i] && own[i]) { hipError_t e = hipStreamDestroy(st[i]); if (e != hipSuccess) return e; }
      own[i] = false;
      st[i] = i - 1 < n ? s[i - 1] : nullptr;
    }
    given = n < MAX - 1 ? n : MAX - 1;
    return hipSuccess;
  }
  // everything parts 1 .. np-1 need exists afterwards; *refused = true: something would have to be created under capture
  hipError_t ensure(hipStream_t env, int np, bool* refused) {
    *refused = false;
    static const char* pinned = getenv("PDEC_PART_LEVEL");
    bool need = !ev_fork;
    for (int i = 1; i < np; ++i) need = need || !st[i] || !ev_join[i];
    const bool relevel = !pinned && level_of != (void*)env;
    if (!need && !relevel) return hipSuccess;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (env) { hipError_t e = hipStreamIsCapturing(env, &cs); if (e != hipSuccess) return e; }
    if (cs != hipStreamCaptureStatusNone) {
      // under capture nothing may be created or destroyed; a stream set that is complete is used as it is (its level was read
      // outside the capture or belongs to the caller)
      if (need) { *refused = true; return hipSuccess; }
      return hipSuccess;
    }
    hipError_t e;
    if (!ev_fork && (e = hipEventCreateWithFlags(&ev_fork, hipEventDisableTiming)) != hipSuccess) return e;
    if (relevel) {
      int level = 0;
      if (env && (e = hipStreamGetPriority(env, &level)) != hipSuccess) return e;      // (the null stream: normal level)
      level_of = (void*)env;
      if (level != own_level) {
        for (int i = 1; i < MAX; ++i)
          if (st[i] && own[i]) { if ((e = hipStreamDestroy(st[i])) != hipSuccess) return e; st[i] = nullptr; own[i] = false; }
        own_level = level;
      }
    }
    for (int i = 1; i < np; ++i) {
      if (!st[i]) {
        int prio = own_level;
        if (pinned) {
          int least = 0, greatest = 0;
          if ((e = hipDeviceGetStreamPriorityRange(&least, &greatest)) != hipSuccess) return e;
          prio = atoi(pinned);
          prio = prio > least ? least : (prio < greatest ? greatest : prio);
        }
        if ((e = hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, prio)) != hipSuccess) return e;
        own[i] = true;
      }
      if (!ev_join[i] && (e = hipEventCreateWithFlags(&ev_join[i], hipEventDisableTiming)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  hipError_t fork(hipStream_t env, int np) {
    hipError_t e = hipEventRecord(ev_fork, env);
    for (int i = 1; i < np && e == hipSuccess; ++i) e = hipStreamWaitEvent(st[i], ev_fork, 0);
    return e;
  }
  hipError_t join(hipStream_t env, int np) {
    hipError_t first = hipSuccess;
    for (int i = 1; i < np; ++i) {
      hipError_t e = hipEventRecord(ev_join[i], st[i]);
      if (e == hipSuccess) e = hipStreamWaitEvent(env, ev_join[i], 0);
      if (e != hipSuccess && first == hipSuccess) first = e;
    }
    return first;
  }
};

}  // namespace pdec
