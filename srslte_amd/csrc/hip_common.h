// hip_common.h -- shared host-side helpers of libsrsran_phy_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

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

// The device chosen with srsran_hip_set_device() is the PROCESS's device (one process per GPU): HIP keeps the current device per
// thread and starts every new thread on device 0, so the handle-API entry points bind the calling worker thread to the process's
// device before they touch a stream (a thread-local compare after the first call).
void bind_thread();

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
  KNOB_COUNT
};
int knob(Knob k);

static inline uint32_t ceil_div(uint32_t a, uint32_t b)
{
  return (a + b - 1) / b;
}

} // namespace phyhip
