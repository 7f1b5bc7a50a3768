// common.cpp -- device plumbing shared by every entry point of libsrsran_phy_hip.so
#include "hip_common.h"

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

bool device_available()
{
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

} // namespace phyhip

using namespace phyhip;

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
