// roundtrip_probe.hip -- what ONE host -> device -> host round trip costs on this box, by how the host waits and where its buffers live.
// The handle API and the transport-block seams are round trips (H2D, one or a few kernels, D2H, one host wait); this probe separates the
// fixed cost from the bytes.  Build: hipcc --offload-arch=gfx950 -O3 -o roundtrip_probe roundtrip_probe.hip ; run: ./roundtrip_probe [flags]
//   flags: 0 = hipDeviceScheduleAuto (default), 1 = Spin, 2 = Yield, 4 = BlockingSync
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

__global__ void touch(const int* in, int* out, int n)
{
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[i] = in[i] + 1;
  }
}

static double now_us()
{
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <class F>
static void timeit(const char* what, F f, int reps = 400)
{
  for (int i = 0; i < 20; i++) {
    f();
  }
  std::vector<double> t;
  for (int i = 0; i < reps; i++) {
    double t0 = now_us();
    f();
    t.push_back(now_us() - t0);
  }
  std::sort(t.begin(), t.end());
  printf("  %-78s p50 %7.1f us   p99 %7.1f us   min %7.1f us\n", what, t[t.size() / 2], t[(size_t)(t.size() * 0.99) - 1], t[0]);
}

// a kernel that lasts about `us` microseconds (one small workgroup spinning on the clock): the host wait of N concurrent callers, each with its own
// stream, by hipStreamSynchronize and by polling hipStreamQuery -- does a caller's wait get longer when other threads wait too?
__global__ void spin_us(long long cycles, int* out)
{
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {
  }
  if (out) {
    out[0] = 1;
  }
}

#include <atomic>
#include <thread>
static void threaded(int n_threads, bool poll)
{
  std::vector<std::vector<double>> lat(n_threads);
  std::atomic<int>                 ready{0};
  auto work = [&](int t) {
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    int* d;
    hipMalloc(&d, 64);
    ready++;
    while (ready.load() < n_threads) {
    }
    for (int i = 0; i < 320; i++) {
      const double t0 = now_us();
      hipLaunchKernelGGL(spin_us, dim3(13), dim3(64), 0, st, 100LL * 100, d); // wall_clock64 ticks at 100 MHz: 100 us
      if (poll) {
        while (hipStreamQuery(st) == hipErrorNotReady) {
        }
      } else {
        hipStreamSynchronize(st);
      }
      if (i >= 20) {
        lat[t].push_back(now_us() - t0);
      }
    }
    hipFree(d);
    hipStreamDestroy(st);
  };
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++) {
    th.emplace_back(work, t);
  }
  for (auto& x : th) {
    x.join();
  }
  std::vector<double> all;
  for (auto& l : lat) {
    all.insert(all.end(), l.begin(), l.end());
  }
  std::sort(all.begin(), all.end());
  printf("  %d thread(s), a 100 us kernel each per call, wait by %-22s p50 %7.1f us   p99 %7.1f us\n", n_threads, poll ? "hipStreamQuery polling:" : "hipStreamSynchronize:", all[all.size() / 2],
         all[(size_t)(all.size() * 0.99) - 1]);
}

int main(int argc, char** argv)
{
  unsigned flags = argc > 1 ? (unsigned)atoi(argv[1]) : 0;
  if (hipSetDeviceFlags(flags) != hipSuccess) {
    printf("hipSetDeviceFlags(%u) failed\n", flags);
  }
  printf("device flags %u (0 auto, 1 spin, 2 yield, 4 blocking sync)\n", flags);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  const size_t small = 4096, big = 200 * 1024;
  int *d_in, *d_out, *p_in, *p_out;
  hipMalloc(&d_in, big);
  hipMalloc(&d_out, big);
  hipHostMalloc(&p_in, big);
  hipHostMalloc(&p_out, big);
  int* g_in  = (int*)malloc(big);
  int* g_out = (int*)malloc(big);
  memset(g_in, 1, big);
  memset(p_in, 1, big);
  hipEvent_t ev;
  hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  auto launch = [&](size_t bytes) { hipLaunchKernelGGL(touch, dim3((unsigned)(bytes / 4 + 255) / 256), dim3(256), 0, st, d_in, d_out, (int)(bytes / 4)); };

  timeit("kernel + hipStreamSynchronize", [&] { launch(small); hipStreamSynchronize(st); });
  timeit("kernel + event record + spin on hipEventQuery", [&] { launch(small); hipEventRecord(ev, st); while (hipEventQuery(ev) == hipErrorNotReady) {} });
  timeit("kernel + spin on hipStreamQuery", [&] { launch(small); while (hipStreamQuery(st) == hipErrorNotReady) {} });
  for (size_t bytes : {small, big}) {
    char w[160];
    snprintf(w, sizeof w, "%zu KB pinned: H2D + kernel + D2H + hipStreamSynchronize", bytes / 1024);
    timeit(w, [&] { hipMemcpyAsync(d_in, p_in, bytes, hipMemcpyHostToDevice, st); launch(bytes); hipMemcpyAsync(p_out, d_out, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    snprintf(w, sizeof w, "%zu KB pinned: H2D + kernel + D2H + spin on hipStreamQuery", bytes / 1024);
    timeit(w, [&] { hipMemcpyAsync(d_in, p_in, bytes, hipMemcpyHostToDevice, st); launch(bytes); hipMemcpyAsync(p_out, d_out, bytes, hipMemcpyDeviceToHost, st); while (hipStreamQuery(st) == hipErrorNotReady) {} });
    snprintf(w, sizeof w, "%zu KB pageable: hipMemcpyAsync H2D + kernel + hipMemcpyAsync D2H + sync", bytes / 1024);
    timeit(w, [&] { hipMemcpyAsync(d_in, g_in, bytes, hipMemcpyHostToDevice, st); launch(bytes); hipMemcpyAsync(g_out, d_out, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); });
    snprintf(w, sizeof w, "%zu KB pageable -> memcpy to pinned, H2D + kernel + D2H + sync, memcpy from pinned", bytes / 1024);
    timeit(w, [&] { memcpy(p_in, g_in, bytes); hipMemcpyAsync(d_in, p_in, bytes, hipMemcpyHostToDevice, st); launch(bytes); hipMemcpyAsync(p_out, d_out, bytes, hipMemcpyDeviceToHost, st); hipStreamSynchronize(st); memcpy(g_out, p_out, bytes); });
    snprintf(w, sizeof w, "%zu KB: kernel reads pinned host memory directly, writes it directly (no copies) + sync", bytes / 1024);
    timeit(w, [&] { hipLaunchKernelGGL(touch, dim3((unsigned)(bytes / 4 + 255) / 256), dim3(256), 0, st, p_in, p_out, (int)(bytes / 4)); hipStreamSynchronize(st); });
  }
  timeit("four kernels back to back + sync", [&] { for (int k = 0; k < 4; k++) launch(small); hipStreamSynchronize(st); });
  timeit("hipMemsetAsync + kernel + sync", [&] { hipMemsetAsync(d_out, 0, small, st); launch(small); hipStreamSynchronize(st); });
  for (int n : {1, 3, 8}) {
    threaded(n, false);
    threaded(n, true);
  }
  return 0;
}
