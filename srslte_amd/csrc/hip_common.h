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

static inline uint32_t ceil_div(uint32_t a, uint32_t b)
{
  return (a + b - 1) / b;
}

} // namespace phyhip
