// runtime.hip -- init/shutdown, handle registry, memory helpers, profiling accessors
#include "common.hpp"

namespace pdec {

static thread_local char g_err[512] = "";
static std::mutex g_mu;
static std::map<pdec_handle, std::unique_ptr<Object>> g_objs;
static pdec_handle g_next = 0x70de0001ull;
static bool g_inited = false;

std::vector<int*>* g_flip_log = nullptr;      // non-null while pdec_capture_begin .. pdec_capture_end is open
void note_flip(int* selector) {
  if (g_flip_log) g_flip_log->push_back(selector);
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

Object::~Object() {
  for (auto& kv : profs)
    for (auto& pr : kv.second.ev) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
}

pdec_handle register_object(std::unique_ptr<Object> o) {
  std::lock_guard<std::mutex> lk(g_mu);
  pdec_handle h = g_next++;
  g_objs[h] = std::move(o);
  return h;
}

Object* lookup(pdec_handle h) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_objs.find(h);
  return it == g_objs.end() ? nullptr : it->second.get();
}

template <class T>
__global__ void convert_kernel(const double* __restrict__ src, T* __restrict__ dst, size_t n) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (T)src[i];
}

int upload_converted(DevBuf& dst, const double* src, size_t n, int dtype) {
  if (n == 0) return PDEC_OK;
  if (dtype == PDEC_F64) {
    PDEC_HIP(dst.alloc(n * 8));
    PDEC_HIP(hipMemcpy(dst.p, src, n * 8, hipMemcpyHostToDevice));
    return PDEC_OK;
  }
  std::vector<float> tmp(n);
  for (size_t i = 0; i < n; ++i) tmp[i] = (float)src[i];
  PDEC_HIP(dst.alloc(n * 4));
  PDEC_HIP(hipMemcpy(dst.p, tmp.data(), n * 4, hipMemcpyHostToDevice));
  return PDEC_OK;
}

// pdec_debug_spin_us (include/pdeconv_debug.h): one wave waits on the constant-rate 100 MHz counter; both exits are certain
// (the counter advances whatever the shader clock does, and the iteration cap ends the loop regardless)
__global__ __launch_bounds__(64) void spin_kernel(long long ticks, long long max_iter, unsigned long long* sink) {
  const long long t0 = (long long)wall_clock64();
  long long i = 0;
  for (; i < max_iter; ++i) {
    if ((long long)wall_clock64() - t0 >= ticks) break;
    __builtin_amdgcn_s_sleep(4);
  }
  if (sink && threadIdx.x == 0) *sink = (unsigned long long)i;
}

}  // namespace pdec

using namespace pdec;

extern "C" {

const char* pdec_last_error(void) { return g_err; }

int pdec_version(void) { return 100; }

int pdec_debug_spin_us(void* hip_stream, double us) {
  PDEC_REQUIRE(us >= 0.0 && us <= 10000.0, "pdec_debug_spin_us: 0 <= us <= 10000");
  if (us == 0.0) return PDEC_OK;
  const long long ticks = (long long)(us * 100.0 + 0.5);           // 100 MHz
  // one iteration (counter read + s_sleep 4 = 256 clocks) lasts >= 0.1 us at any shader clock: 64 iterations per tick is
  // ~600x more than the wait needs and still ends a stalled-counter launch within seconds
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, ticks, ticks * 64 + 1024, (unsigned long long*)nullptr);
  PDEC_HIP(hipGetLastError());
  return PDEC_OK;
}

int pdec_device_count(int* n) {
  PDEC_REQUIRE(n, "pdec_device_count: null");
  int c = 0;
  hipError_t e = hipGetDeviceCount(&c);
  if (e != hipSuccess) {
    *n = 0;
    set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return PDEC_E_NOGPU;
  }
  *n = c;
  return PDEC_OK;
}

int pdec_init(int device_ordinal) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c <= 0) {
    set_error("pdec_init: no HIP device visible");
    return PDEC_E_NOGPU;
  }
  PDEC_REQUIRE(device_ordinal >= 0 && device_ordinal < c, "pdec_init: device %d out of range (%d devices)",
               device_ordinal, c);
  PDEC_HIP(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  PDEC_HIP(hipGetDeviceProperties(&prop, device_ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("pdec_init: device %d is %s; this library only carries gfx950 (MI355X) code objects",
              device_ordinal, prop.gcnArchName);
    return PDEC_E_NOGPU;
  }
  g_inited = true;
  return PDEC_OK;
}

int pdec_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  (void)hipDeviceSynchronize();
  g_objs.clear();
  g_inited = false;
  return PDEC_OK;
}

int pdec_malloc(void** dptr, size_t bytes) {
  PDEC_REQUIRE(dptr, "pdec_malloc: null");
  PDEC_HIP(hipMalloc(dptr, bytes ? bytes : 1));
  return PDEC_OK;
}
int pdec_free(void* dptr) {
  if (dptr) PDEC_HIP(hipFree(dptr));
  return PDEC_OK;
}
int pdec_memcpy_h2d(void* dst, const void* src, size_t bytes) {
  PDEC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return PDEC_OK;
}
int pdec_memcpy_d2h(void* dst, const void* src, size_t bytes) {
  PDEC_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return PDEC_OK;
}
int pdec_memset(void* dptr, int value, size_t bytes) {
  PDEC_HIP(hipMemset(dptr, value, bytes));
  return PDEC_OK;
}

int pdec_set_stream(pdec_handle h, void* s) {
  Object* o = lookup(h);
  if (!o) {
    set_error("pdec_set_stream: bad handle");
    return PDEC_E_HANDLE;
  }
  o->stream = (hipStream_t)s;
  return PDEC_OK;
}

int pdec_stream_create(void** hip_stream, int level) {
  if (!hip_stream) {
    set_error("pdec_stream_create: null out pointer");
    return PDEC_E_INVALID;
  }
  // the null stream's hardware queue first, once per process, if nothing has made it yet: made later it would land BETWEEN
  // the streams a caller makes back to back (include/pdeconv.h), on a pipe of its own choosing
  static std::once_flag null_queue;
  std::call_once(null_queue, [] {
    void* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return;
    (void)hipMemsetAsync(d, 0, 4, nullptr);
    (void)hipStreamSynchronize(nullptr);
    (void)hipFree(d);
  });
  int least = 0, greatest = 0;   // numerically: least >= 0 >= greatest
  PDEC_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
  int prio = level > least ? least : (level < greatest ? greatest : level);
  hipStream_t st = nullptr;
  PDEC_HIP(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio));
  *hip_stream = (void*)st;
  return PDEC_OK;
}

int pdec_stream_destroy(void* hip_stream) {
  if (!hip_stream) return PDEC_OK;
  PDEC_HIP(hipStreamDestroy((hipStream_t)hip_stream));
  return PDEC_OK;
}

int pdec_sync(pdec_handle h) {
  Object* o = lookup(h);
  if (!o) {
    set_error("pdec_sync: bad handle");
    return PDEC_E_HANDLE;
  }
  PDEC_HIP(hipStreamSynchronize(o->stream));
  return PDEC_OK;
}

int pdec_destroy(pdec_handle h) {
  std::unique_ptr<Object> victim;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_objs.find(h);
    if (it == g_objs.end()) {
      set_error("pdec_destroy: bad handle");
      return PDEC_E_HANDLE;
    }
    victim = std::move(it->second);
    g_objs.erase(it);
  }
  (void)hipStreamSynchronize(victim->stream);
  victim.reset();
  return PDEC_OK;
}

int pdec_prof_enable(pdec_handle h, int on) {
  Object* o = lookup(h);
  if (!o) return PDEC_E_HANDLE;
  o->prof = on != 0;
  o->prof_reps = on > 1 ? on : 1;
  return PDEC_OK;
}

int pdec_prof_reset(pdec_handle h) {
  Object* o = lookup(h);
  if (!o) return PDEC_E_HANDLE;
  (void)hipStreamSynchronize(o->stream);
  for (auto& kv : o->profs)
    for (auto& pr : kv.second.ev) {
      (void)hipEventDestroy(pr.first);
      (void)hipEventDestroy(pr.second);
    }
  o->profs.clear();
  return PDEC_OK;
}

int pdec_prof_get(pdec_handle h, const char* name, double* mean_ms, int* count) {
  Object* o = lookup(h);
  if (!o) return PDEC_E_HANDLE;
  PDEC_REQUIRE(name && mean_ms && count, "pdec_prof_get: null");
  auto it = o->profs.find(name);
  if (it == o->profs.end() || it->second.ev.empty()) {
    *mean_ms = 0;
    *count = 0;
    return PDEC_OK;
  }
  double tot = 0;
  int n = 0;
  for (auto& pr : it->second.ev) {
    PDEC_HIP(hipEventSynchronize(pr.second));
    float ms = 0;
    PDEC_HIP(hipEventElapsedTime(&ms, pr.first, pr.second));
    tot += ms;
    ++n;
  }
  *mean_ms = tot / n / it->second.reps;
  *count = n;
  return PDEC_OK;
}

}  // extern "C"
