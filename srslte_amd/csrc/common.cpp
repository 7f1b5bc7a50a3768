// common.cpp -- device plumbing shared by every entry point of libsrsran_phy_hip.so
#include "coalesce.h"
#include "hip_common.h"

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

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

// The reference's threading model is one PHY worker per in-flight subframe, and every worker's handles and staging contexts own a stream.  The
// runtime spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (4 unless told otherwise) and streams that share a queue run
// their work one behind the other: three workers decoding a transport block each then take 249 us per call instead of 142, with eight queues
// 146 (tools/probe/seam_threads.c, profiles/r03_seam_threads.txt).  Ask for eight -- before the runtime reads its settings at the process's
// first HIP call, and only if the application or the user has not chosen a value.
__attribute__((constructor)) void more_hardware_queues()
{
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
}
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

// ---- development knobs (hip_common.h)
namespace {
const char* const kKnobEnv[KNOB_COUNT] = {"SRSRAN_HIP_TDEC_VARIANT", "SRSRAN_HIP_PSS_VARIANT", "TDEC_DBG_EXTRACT_ONLY", "LDPC_PCPB", "LDPC_SLOTS", "LDPC_PACKED", "SRSRAN_HIP_TDEC_LAT"};
std::atomic<int>  g_knob[KNOB_COUNT];
std::atomic<bool> g_knob_read[KNOB_COUNT];

int knob_parse(int k, const char* v)
{
  if (!v) {
    return -1;
  }
  switch (k) {
    case KNOB_TDEC_VARIANT:
      return !strcmp(v, "waves1") ? 1 : (!strcmp(v, "persistent") ? 2 : 0);
    case KNOB_PSS_VARIANT:
      return !strcmp(v, "pair") ? 1 : (!strcmp(v, "block") ? 2 : 0);
    case KNOB_TDEC_EXTRACT_ONLY:
      return 1;
    default:
      return atoi(v);
  }
}
} // namespace

int knob(Knob k)
{
  if (!g_knob_read[k].load(std::memory_order_acquire)) {
    g_knob[k].store(knob_parse(k, getenv(kKnobEnv[k])), std::memory_order_relaxed);
    g_knob_read[k].store(true, std::memory_order_release);
  }
  return g_knob[k].load(std::memory_order_relaxed);
}

// ---- submission queues under the handle API (coalesce.h)
namespace {
struct Entry {
  std::mutex                 mu; // creation of this shape's queue
  std::shared_ptr<Coalescer> q;
  bool                       tried = false;
  uint64_t                   last_use = 0;
};
struct Registry {
  std::mutex                                    mu;
  std::map<std::string, std::shared_ptr<Entry>> byKey;
  uint64_t                                      tick = 0;
  uint64_t                                      dead_batches = 0, dead_units = 0; // statistics of evicted queues
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

std::shared_ptr<Coalescer> coalescer_for(const std::string& key, const std::function<Coalescer*()>& make)
{
  Registry&                           r = registry();
  std::shared_ptr<Entry>              e;
  std::vector<std::shared_ptr<Entry>> evicted;
  {
    std::lock_guard<std::mutex> lk(r.mu);
    auto                        it = r.byKey.find(key);
    if (it == r.byKey.end()) {
      // a new shape: make room first -- least recently used entries that only the registry holds
      while (r.byKey.size() >= kMaxShapes) {
        auto victim = r.byKey.end();
        for (auto jt = r.byKey.begin(); jt != r.byKey.end(); ++jt) {
          const bool idle = jt->second.use_count() == 1 && (!jt->second->q || jt->second->q.use_count() == 1);
          if (idle && (victim == r.byKey.end() || jt->second->last_use < victim->second->last_use)) {
            victim = jt;
          }
        }
        if (victim == r.byKey.end()) {
          break; // everything is in use right now: grow past the bound rather than block
        }
        if (victim->second->q) {
          uint64_t b = 0, u = 0;
          victim->second->q->stats(&b, &u);
          r.dead_batches += b;
          r.dead_units += u;
        }
        evicted.push_back(victim->second);
        r.byKey.erase(victim);
      }
      it = r.byKey.emplace(key, std::make_shared<Entry>()).first;
    }
    e           = it->second;
    e->last_use = ++r.tick;
  }
  evicted.clear(); // releases the evicted queues' device resources outside the registry lock
  std::lock_guard<std::mutex> lk(e->mu);
  if (!e->tried) {
    e->tried     = true;
    Coalescer* c = make();
    if (c && !c->ok()) {
      delete c;
      c = nullptr;
    }
    e->q.reset(c);
  }
  return e->q; // may be empty: creation failed before, callers fall back to their private path
}

size_t coalescer_shapes()
{
  Registry&                   r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  size_t                      n = 0;
  for (auto& kv : r.byKey) {
    n += kv.second->q ? 1 : 0;
  }
  return n;
}

} // namespace phyhip

using namespace phyhip;

extern "C" void srsran_hip_set_coalescing(int enable)
{
  g_coalesce.store(enable ? 1 : 0, std::memory_order_relaxed);
}

extern "C" void srsran_hip_coalesce_stats(uint64_t* nof_batches, uint64_t* nof_units)
{
  Registry&                   r = registry();
  std::lock_guard<std::mutex> lk(r.mu);
  uint64_t                    b = r.dead_batches, u = r.dead_units;
  for (auto& kv : r.byKey) {
    if (kv.second->q) {
      uint64_t bb = 0, uu = 0;
      kv.second->q->stats(&bb, &uu);
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

extern "C" int srsran_hip_dev_knob(const char* env_name, const char* value)
{
  for (int k = 0; env_name && k < KNOB_COUNT; k++) {
    if (!strcmp(env_name, kKnobEnv[k])) {
      g_knob[k].store(knob_parse(k, value), std::memory_order_relaxed);
      g_knob_read[k].store(true, std::memory_order_release);
      return SRSRAN_SUCCESS;
    }
  }
  return SRSRAN_ERROR_INVALID_INPUTS;
}

extern "C" uint32_t srsran_hip_coalesce_shapes(void)
{
  return (uint32_t)coalescer_shapes();
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
