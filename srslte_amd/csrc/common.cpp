// common.cpp -- device plumbing shared by every entry point of libsrsran_phy_hip.so
#include "coalesce.h"
#include "hip_common.h"

#include <atomic>
#include <mutex>

namespace phyhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...)
{
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

const char* get_error()
{
  return g_err;
}

namespace {
std::atomic<int> g_device{-1}; // -1: never chosen, every thread stays where HIP put it
}

void bind_thread()
{
  static thread_local int bound = -1;
  const int               want  = g_device.load(std::memory_order_relaxed);
  if (want >= 0 && bound != want) {
    (void)hipSetDevice(want);
    bound = want;
  }
}

bool device_available()
{
  bind_thread();
  static std::once_flag once;
  static bool           ok = false;
  std::call_once(once, [] {
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    ok           = (e == hipSuccess && n > 0);
    if (!ok) {
      set_error("no HIP device available (%s); the PHY HIP engine has no CPU fallback",
                e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
      fprintf(stderr, "[srsran_phy_hip] %s\n", g_err);
    }
  });
  if (!ok) {
    set_error("no HIP device available; the PHY HIP engine has no CPU fallback");
  }
  return ok;
}

// ---- submission queues under the handle API (coalesce.h)
namespace {
struct Registry {
  std::mutex                        mu;
  std::map<std::string, Coalescer*> byKey;
};
Registry& registry()
{
  static Registry* r = new Registry; // never destroyed: the queues own HIP resources and outlive static destruction order
  return *r;
}
std::atomic<int> g_coalesce{-1}; // -1: not decided yet (environment), 0 / 1
} // namespace

bool coalescing_enabled()
{
  int v = g_coalesce.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("SRSRAN_HIP_COALESCE");
    v             = (e && e[0] == '0') ? 0 : 1;
    g_coalesce.store(v, std::memory_order_relaxed);
  }
  return v == 1;
}

Coalescer* coalescer_for(const std::string& key, const std::function<Coalescer*()>& make)
{
  Registry&                   r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  auto                        it = r.byKey.find(key);
  if (it != r.byKey.end()) {
    return it->second; // may be nullptr: creation failed before, callers fall back to their private path
  }
  Coalescer* c = make();
  if (c && !c->ok()) {
    delete c;
    c = nullptr;
  }
  r.byKey[key] = c;
  return c;
}

} // namespace phyhip

using namespace phyhip;

extern "C" void srsran_hip_set_coalescing(int enable)
{
  g_coalesce.store(enable ? 1 : 0, std::memory_order_relaxed);
}

extern "C" void srsran_hip_coalesce_stats(uint64_t* nof_batches, uint64_t* nof_units)
{
  uint64_t  b = 0, u = 0;
  Registry& r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  for (auto& kv : r.byKey) {
    if (kv.second) {
      uint64_t bb = 0, uu = 0;
      kv.second->stats(&bb, &uu);
      b += bb;
      u += uu;
    }
  }
  if (nof_batches) {
    *nof_batches = b;
  }
  if (nof_units) {
    *nof_units = u;
  }
}

extern "C" int srsran_hip_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    return 0;
  }
  return n;
}

extern "C" int srsran_hip_set_device(int device)
{
  PHY_HIP_CHECK(hipSetDevice(device), SRSRAN_ERROR);
  g_device.store(device, std::memory_order_relaxed); // worker threads follow (bind_thread)
  return SRSRAN_SUCCESS;
}

extern "C" void* srsran_hip_malloc(size_t bytes)
{
  void* p = nullptr;
  PHY_HIP_CHECK(hipMalloc(&p, bytes ? bytes : 1), nullptr);
  return p;
}

extern "C" void srsran_hip_free(void* dptr)
{
  if (dptr) {
    (void)hipFree(dptr);
  }
}

extern "C" int srsran_hip_memcpy_h2d(void* dst, const void* src, size_t bytes, void* stream)
{
  PHY_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, (hipStream_t)stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_memcpy_d2h(void* dst, const void* src, size_t bytes, void* stream)
{
  PHY_HIP_CHECK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream), SRSRAN_ERROR);
  PHY_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_memset(void* dst, int value, size_t bytes, void* stream)
{
  PHY_HIP_CHECK(hipMemsetAsync(dst, value, bytes, (hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_stream_sync(void* stream)
{
  PHY_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream), SRSRAN_ERROR);
  return SRSRAN_SUCCESS;
}

extern "C" const char* srsran_hip_last_error(void)
{
  return get_error();
}

extern "C" const char* srsran_hip_build_info(void)
{
  return "libsrsran_phy_hip gfx950 (MI355X) " __DATE__ " " __TIME__;
}
