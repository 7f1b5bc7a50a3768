// hip_common.h -- shared host-side helpers of libsrsran_phy_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "srsran_amd/phy_abi.h"
#include "srsran_amd/phy_batch.h"

namespace phyhip {

// last error text (thread local), returned by srsran_hip_last_error()
void        set_error(const char* fmt, ...);
const char* get_error();

// Fails loudly: the product path has no CPU fallback.
#define PHY_HIP_CHECK(expr, retval)                                                                                    \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) {                                                                                            \
      phyhip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));                          \
      fprintf(stderr, "[srsran_phy_hip] %s\n", phyhip::get_error());                                                   \
      return retval;                                                                                                   \
    }                                                                                                                  \
  } while (0)

#define PHY_HIP_CHECK_VOID(expr)                                                                                       \
  do {                                                                                                                 \
    hipError_t _e = (expr);                                                                                            \
    if (_e != hipSuccess) {                                                                                            \
      phyhip::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));                          \
      fprintf(stderr, "[srsran_phy_hip] %s\n", phyhip::get_error());                                                   \
      return;                                                                                                          \
    }                                                                                                                  \
  } while (0)

// true when a HIP device is usable; prints one diagnostic otherwise
bool device_available();

// Devices.  srsran_hip_set_device(d) names the PROCESS's default device (one process per GPU: bench.py, the N-rank launches); a thread that calls
// srsran_hip_set_thread_device(d) is bound to d from then on, whatever the default -- N worker threads of ONE process on N GPUs, the reference's
// own threading model (lib/include/srsran/common/thread_pool.h:48, srsenb nof_phy_threads).  HIP keeps the current device per thread and starts every
// new thread on device 0, so every entry point binds the calling thread first (bind_thread: a thread-local compare after the first call).
// Everything that owns device memory or a stream RECORDS its device (DeviceTag) and every entry point compares it with the calling thread's
// (check_device): a handle created on one device and used from a thread bound to another is refused with an error, never launched.  Process-wide
// caches of device constants and the pools of staging contexts exist once per device (DeviceLocal).
void bind_thread();
int  current_device(); // the (logical) device the calling thread works on
constexpr int kMaxDevices = 16;
struct DeviceTag { // member of everything that owns device memory or a stream: stamped with the constructing thread's device
  int dev;
  DeviceTag() : dev(current_device()) {}
};
// false (error text set, one line on stderr) when `tag` is another device than the calling thread's
bool check_device(const DeviceTag& tag, const char* who);
#define PHY_DEV_GUARD(tag, who, retval)                                                                                \
  do {                                                                                                                 \
    if (!phyhip::check_device((tag), (who))) {                                                                         \
      return retval;                                                                                                   \
    }                                                                                                                  \
  } while (0)
#define PHY_DEV_GUARD_VOID(tag, who)                                                                                   \
  do {                                                                                                                 \
    if (!phyhip::check_device((tag), (who))) {                                                                         \
      return;                                                                                                          \
    }                                                                                                                  \
  } while (0)

// Table uploads (object creation, first use of a new block size / generator).  A hipMemcpy from pageable memory returns when the HOST
// buffer may be reused; a small copy is staged and reaches the device in null-stream order -- and every kernel of this library runs on
// hipStreamNonBlocking streams, which do not order themselves against the null stream.  Nothing documents that such a copy is complete on
// the device when hipMemcpy returns, so upload() makes it so: it returns when the device has the data, whichever stream reads it next.
inline hipError_t upload(void* dst, const void* src, size_t bytes)
{
  const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  return e != hipSuccess ? e : hipDeviceSynchronize();
}

// Host memory that kernels read and write THEMSELVES (single-call staging images, job lists and verdicts of small calls): pinned, mapped into
// the device's address space and explicitly coherent (fine-grained) -- what the host wrote before a launch is what the kernel reads, what a
// kernel wrote is what the host reads after the stream's wait, whatever HIP_HOST_COHERENT says about the default.
inline hipError_t host_image_alloc(void** p, size_t bytes)
{
  return hipHostMalloc(p, bytes, hipHostMallocMapped | hipHostMallocCoherent);
}
template <class T>
inline hipError_t host_image_alloc(T** p, size_t bytes)
{
  return host_image_alloc(reinterpret_cast<void**>(p), bytes);
}

// development knobs: launch-shape alternatives kept in the tree for measurement (profiles/r02_turbo_variants.txt, r02_pss_variants.txt) and a few
// sizing overrides.  The environment variable of a knob is read ONCE (first use); srsran_hip_dev_knob() overrides a knob at run time,
// which is how tests/test_gpu_variants.py switches kernels inside one process.  -1 = not set.
enum Knob {
  KNOB_TDEC_VARIANT = 0,  // SRSRAN_HIP_TDEC_VARIANT: 0 product, 1 "waves1", 2 "persistent"
  KNOB_PSS_VARIANT,       // SRSRAN_HIP_PSS_VARIANT: 0 product ("wave"), 1 "pair", 2 "block"
  KNOB_TDEC_EXTRACT_ONLY, // TDEC_DBG_EXTRACT_ONLY
  KNOB_LDPC_PCPB,         // LDPC_PCPB
  KNOB_LDPC_SLOTS,        // LDPC_SLOTS
  KNOB_LDPC_PACKED,       // LDPC_PACKED
  KNOB_TDEC_LAT,          // SRSRAN_HIP_TDEC_LAT: 0 never use the latency kernel, 1 always (where it exists), unset: by batch size
  KNOB_LDPC_C2V_LDS,      // SRSRAN_HIP_LDPC_C2V_LDS: 0 = small LDPC batches keep their messages in the global slabs
  KNOB_TCOD_LAT,          // SRSRAN_HIP_TCOD_LAT: 0 = never use the one-launch transmit kernel for small batches
  KNOB_LOGICAL_DEVICES,   // SRSRAN_HIP_LOGICAL_DEVICES: n logical devices on the installed ones (development: the per-device bookkeeping on a 1-GPU box)
  KNOB_TDEC_LAT2,         // SRSRAN_HIP_TDEC_LAT2: 0 = never use the two-wave form of the latency kernel, 1 = wherever it exists; unset: by batch size
  KNOB_COUNT
};
int knob(Knob k);

// Per-thread staging contexts (stream, pinned + device images, decoder / plan objects) live in a process-wide POOL: a thread takes one at its first call
// and gives it back when it ends, so (i) what srsran_hip_warmup() -- or the init-time hook srsran_rm_turbo_gentables(), which the reference calls from
// srsran_sch_init (sch.c:166) -- prepared on a short-lived thread is what a PHY worker finds at its first subframe (the reference creates its objects on one
// thread and runs them on others), and (ii) worker churn does not re-create streams and pinned memory.  The pool itself is never destroyed (the HIP runtime
// may be gone by the time static destructors run).
// one T per device, made on first use, never destroyed (they own device memory; the runtime may be gone at static destruction)
template <class T>
T& device_local()
{
  static std::mutex mu;
  static T*         slot[kMaxDevices] = {};
  const int         d = current_device();
  const int         i = (d >= 0 && d < kMaxDevices) ? d : 0;
  std::lock_guard<std::mutex> lk(mu);
  if (!slot[i]) {
    slot[i] = new T;
  }
  return *slot[i];
}
// one T per (thread, device): the thread-local staging contexts of the host-pointer entry points
template <class T>
T& thread_device_local()
{
  struct Holder {
    T* slot[kMaxDevices] = {};
    ~Holder()
    {
      for (T* p : slot) {
        delete p;
      }
    }
  };
  static thread_local Holder h;
  const int                  d = current_device();
  const int                  i = (d >= 0 && d < kMaxDevices) ? d : 0;
  if (!h.slot[i]) {
    h.slot[i] = new T;
  }
  return *h.slot[i];
}

template <class T>
class StagePool {
public:
  static StagePool& get() // the pool of the calling thread's device
  {
    return device_local<StagePool>();
  }
  T* take()
  {
    {
      std::lock_guard<std::mutex> lk(mu_);
      if (!idle_.empty()) {
        T* s = idle_.back();
        idle_.pop_back();
        return s;
      }
    }
    return new T;
  }
  void give(T* s)
  {
    std::lock_guard<std::mutex> lk(mu_);
    idle_.push_back(s);
  }
  size_t idle()
  {
    std::lock_guard<std::mutex> lk(mu_);
    return idle_.size();
  }

private:
  std::mutex      mu_;
  std::vector<T*> idle_;
};
template <class T>
struct StageRef { // one per thread (thread_local): the thread's context on each device it has worked on
  T*            p[kMaxDevices]    = {};
  StagePool<T>* home[kMaxDevices] = {};
  ~StageRef()
  {
    for (int i = 0; i < kMaxDevices; i++) {
      if (p[i]) {
        home[i]->give(p[i]);
      }
    }
  }
  T& get()
  {
    const int d = current_device();
    const int i = (d >= 0 && d < kMaxDevices) ? d : 0;
    if (!p[i]) {
      home[i] = &StagePool<T>::get();
      p[i]    = home[i]->take();
    }
    return *p[i];
  }
};

// roctx ranges around the batch / grant entry points (what pusch.c:368-372's meas_time_en is to the reference): a range shows up in a
// `rocprofv3 --marker-trace` timeline with the kernels of the call under it.  Active when the profiler's roctx library is already in the process
// (rocprofv3 preloads it) or when SRSRAN_HIP_ROCTX=1 asks for it to be loaded; otherwise a pointer test.
void trace_push(const char* name);
void trace_pop();
struct TraceRange {
  explicit TraceRange(const char* name) { trace_push(name); }
  ~TraceRange() { trace_pop(); }
  TraceRange(const TraceRange&)            = delete;
  TraceRange& operator=(const TraceRange&) = delete;
};

static inline uint32_t ceil_div(uint32_t a, uint32_t b)
{
  return (a + b - 1) / b;
}

} // namespace phyhip
