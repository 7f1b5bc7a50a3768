// common.cpp -- device plumbing shared by every entry point of libsrsran_phy_hip.so
#include "coalesce.h"
#include "hip_common.h"

#include <atomic>
#include <dlfcn.h>
#include <unistd.h>
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
// The setting only counts when it is in the environment BEFORE the runtime initialises (its first API call in the process).  A host that has used
// HIP before this library is loaded -- a Python process that imported torch and touched the GPU first -- keeps its four queues, and three workers then
// serialise (236-249 us per transport block instead of 143-151, profiles/r03_seam_threads_pass.txt).  The runtime opens /dev/kfd when it
// initialises: if that has happened by the time this constructor runs and nobody had set the variable, say so once (device_available()).
// srslte_amd/__init__.py and bench.py put the variable into the environment themselves, before torch is imported.
bool g_queues_too_late = false;
__attribute__((constructor)) void more_hardware_queues()
{
  if (!getenv("GPU_MAX_HW_QUEUES")) {
    char path[64], target[256];
    for (int fd = 0; fd < 1024 && !g_queues_too_late; fd++) {
      snprintf(path, sizeof(path), "/proc/self/fd/%d", fd);
      const ssize_t n = readlink(path, target, sizeof(target) - 1);
      if (n > 0) {
        target[n] = 0;
        g_queues_too_late = strcmp(target, "/dev/kfd") == 0;
      }
    }
  }
  (void)setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
}
}

namespace {
thread_local int t_device = -1; // srsran_hip_set_thread_device: this thread's device, whatever the process default
thread_local int t_bound  = -1; // the logical device this thread's HIP context was last set to by us

// development aid (SRSRAN_HIP_LOGICAL_DEVICES = n): n logical devices on whatever is installed, logical d -> physical d mod count.  The
// per-device bookkeeping (tags, caches, pools) of a process that spreads its workers over several GPUs can then be exercised on a 1-GPU box.
int physical_count()
{
  static int n = -1;
  if (n < 0) {
    int c = 0;
    n     = (hipGetDeviceCount(&c) == hipSuccess) ? c : 0;
  }
  return n;
}
int logical_count()
{
  const int k = knob(KNOB_LOGICAL_DEVICES);
  const int p = physical_count();
  return (k > 0 && p > 0) ? (k > kMaxDevices ? kMaxDevices : k) : p;
}
int physical_of(int logical)
{
  const int p = physical_count();
  return p > 0 ? logical % p : logical;
}
} // namespace

int current_device()
{
  if (t_device >= 0) {
    return t_device;
  }
  const int g = g_device.load(std::memory_order_relaxed);
  if (g >= 0) {
    return g;
  }
  int d = 0; // nobody chose: the thread is wherever HIP (or the application's own hipSetDevice) put it
  return hipGetDevice(&d) == hipSuccess ? d : 0;
}

void bind_thread()
{
  const int want = t_device >= 0 ? t_device : g_device.load(std::memory_order_relaxed);
  if (want >= 0 && t_bound != want) {
    (void)hipSetDevice(physical_of(want));
    t_bound = want;
  }
}

bool check_device(const DeviceTag& tag, const char* who)
{
  bind_thread();
  const int here = current_device();
  if (tag.dev == here || tag.dev < 0) {
    return true;
  }
  set_error("%s: the object lives on device %d, the calling thread is bound to device %d (srsran_hip_set_thread_device / srsran_hip_set_device)", who, tag.dev,
            here);
  fprintf(stderr, "[srsran_phy_hip] %s\n", g_err);
  return false;
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
    } else if (g_queues_too_late) {
      fprintf(stderr, "[srsran_phy_hip] note: the HIP runtime was initialised before this library was loaded, so GPU_MAX_HW_QUEUES=8 could not be requested: "
                      "with the default of 4 hardware queues more than two concurrent worker threads share queues and run one behind the other.  Set "
                      "GPU_MAX_HW_QUEUES=8 in the environment, or load the library before the first HIP call.\n");
    }
  });
  if (!ok) {
    set_error("no HIP device available; the PHY HIP engine has no CPU fallback");
  }
  return ok;
}

// ---- roctx ranges (hip_common.h)
namespace {
struct RoctxFns {
  int (*push)(const char*) = nullptr;
  int (*pop)()             = nullptr;
};
const RoctxFns* roctx_fns()
{
  static const RoctxFns* fns = [] () -> const RoctxFns* {
    static RoctxFns f;
    const char*     want = getenv("SRSRAN_HIP_ROCTX");
    void*           lib  = nullptr;
    for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"}) {
      lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD); // only what the profiler has already brought in ...
      if (!lib && want && want[0] == '1') {
        lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL); // ... unless asked for
      }
      if (lib) {
        break;
      }
    }
    if (!lib) {
      return nullptr;
    }
    f.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
    f.pop  = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
    return (f.push && f.pop) ? &f : nullptr;
  }();
  return fns;
}
} // namespace

void trace_push(const char* name)
{
  if (const RoctxFns* f = roctx_fns()) {
    (void)f->push(name);
  }
}
void trace_pop()
{
  if (const RoctxFns* f = roctx_fns()) {
    (void)f->pop();
  }
}

// ---- development knobs (hip_common.h)
namespace {
const char* const kKnobEnv[KNOB_COUNT] = {"SRSRAN_HIP_TDEC_VARIANT", "SRSRAN_HIP_PSS_VARIANT", "TDEC_DBG_EXTRACT_ONLY", "LDPC_PCPB", "LDPC_SLOTS", "LDPC_PACKED", "SRSRAN_HIP_TDEC_LAT",
                                          "SRSRAN_HIP_LDPC_C2V_LDS", "SRSRAN_HIP_TCOD_LAT", "SRSRAN_HIP_LOGICAL_DEVICES", "SRSRAN_HIP_TDEC_LAT2"};
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
      return !strcmp(v, "pair") ? 1 : (!strcmp(v, "block") ? 2 : (!strcmp(v, "recompute") ? 3 : 0));
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
  return logical_count();
}

extern "C" int srsran_hip_set_device(int device)
{
  if (device < 0 || device >= logical_count()) {
    set_error("srsran_hip_set_device: no device %d (%d visible)", device, logical_count());
    fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipSetDevice(physical_of(device)), SRSRAN_ERROR);
  g_device.store(device, std::memory_order_relaxed); // the default of every thread that has not bound itself (bind_thread)
  if (t_device < 0) {
    t_bound = device;
  }
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_set_thread_device(int device)
{
  if (device < 0) { // back to the process default
    t_device = -1;
    t_bound  = -1;
    bind_thread();
    return SRSRAN_SUCCESS;
  }
  if (device >= logical_count()) {
    set_error("srsran_hip_set_thread_device: no device %d (%d visible)", device, logical_count());
    fprintf(stderr, "[srsran_phy_hip] %s\n", get_error());
    return SRSRAN_ERROR;
  }
  PHY_HIP_CHECK(hipSetDevice(physical_of(device)), SRSRAN_ERROR);
  t_device = device;
  t_bound  = device;
  return SRSRAN_SUCCESS;
}

extern "C" int srsran_hip_get_thread_device(void)
{
  return current_device();
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

// 1 when the library could ask for its hardware queues in time (or the application chose a value itself), 0 when the runtime was already initialised
extern "C" int srsran_hip_hw_queues_requested_in_time(void)
{
  return g_queues_too_late ? 0 : 1;
}

extern "C" const char* srsran_hip_build_info(void)
{
  return "libsrsran_phy_hip gfx950 (MI355X) " __DATE__ " " __TIME__;
}
