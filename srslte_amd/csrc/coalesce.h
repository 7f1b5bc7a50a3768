// coalesce.h -- per-process submission queues under the unchanged handle API.
//
// The reference's threading contract is one worker thread per in-flight subframe, each with its OWN handles
// (srsenb/src/phy/lte/cc_worker.cc:212-231, lib/include/srsran/common/thread_pool.h:48): N threads call
// srsran_tdec_run_all / srsran_ofdm_rx_sf / srsran_ldpc_decoder_decode_c at the same time on N different handles.  One call is
// one code block or one subframe -- far too little to fill 256 CUs -- so calls of the same SHAPE (same kernel configuration)
// that are in flight together are merged into one batch launch, group-commit style:
//   * a caller queues its request; if nobody is running a batch for this shape it becomes the leader, takes everything that is
//     queued (its own request included), stages the inputs in pinned memory, does ONE upload, ONE launch, ONE download and
//     wakes the others;
//   * callers that arrive while a batch is on the device simply wait in the queue and form the next batch.
// Nobody ever waits for a timer: a lone caller runs at once (batch of one), and the busier the process the larger the batches.
// Results are the batched kernels' results, which the parity tests pin to the oracle per unit, so a call gives the same bytes
// whether it was merged or not.  SRSRAN_HIP_COALESCE=0 turns merging off (every handle then uses its private stream).
#pragma once
#include "hip_common.h"

#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace phyhip {

class Coalescer {
public:
  // runs n units: unit i reads d_in + i * in_stride and writes d_out + i * out_stride (bytes); asynchronous on `st`
  using RunFn = std::function<int(const void* d_in, void* d_out, uint32_t n, hipStream_t st)>;

  Coalescer(size_t in_bytes, size_t out_bytes, uint32_t max_batch, RunFn run) :
    in_bytes_(in_bytes), out_bytes_(out_bytes), in_stride_(stride_of(in_bytes)), out_stride_(stride_of(out_bytes)),
    max_batch_(max_batch), run_(std::move(run))
  {
    ok_ = hipStreamCreateWithFlags(&st_, hipStreamNonBlocking) == hipSuccess && hipMalloc(&d_in_, in_stride_ * max_batch) == hipSuccess &&
          hipMalloc(&d_out_, out_stride_ * max_batch) == hipSuccess && hipHostMalloc(&h_in_, in_stride_ * max_batch) == hipSuccess &&
          hipHostMalloc(&h_out_, out_stride_ * max_batch) == hipSuccess;
  }
  // distance between the units of a batch in the staging buffers: the unit size rounded up to 64 bytes
  static size_t stride_of(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
  bool     ok() const { return ok_; }
  size_t   in_stride() const { return in_stride_; }
  size_t   out_stride() const { return out_stride_; }
  uint32_t max_batch() const { return max_batch_; }

  // blocking: `in` (in_bytes) -> `out` (out_bytes), host memory of the caller.  Returns the run function's code.
  int submit(const void* in, void* out)
  {
    Req                          r{in, out, 0, false};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&r);
    while (!r.done) {
      if (busy_) {
        cv_.wait(lk);
        continue;
      }
      busy_ = true; // leader of the next batch: the oldest requests first (its own is among them unless the queue is very long)
      std::vector<Req*> batch;
      while (!queue_.empty() && batch.size() < max_batch_) {
        batch.push_back(queue_.front());
        queue_.pop_front();
      }
      lk.unlock();
      const int rc = process(batch);
      lk.lock();
      for (Req* b : batch) {
        b->rc   = rc;
        b->done = true;
      }
      n_batches_++;
      n_units_ += batch.size();
      busy_ = false;
      cv_.notify_all();
    }
    return r.rc;
  }
  // statistics for tools/bench_handle.py: batches run and units carried since start
  void stats(uint64_t* batches, uint64_t* units)
  {
    std::lock_guard<std::mutex> lk(mu_);
    *batches = n_batches_;
    *units   = n_units_;
  }

private:
  struct Req {
    const void* in;
    void*       out;
    int         rc;
    bool        done;
  };
  int process(const std::vector<Req*>& batch)
  {
    const uint32_t n = (uint32_t)batch.size();
    for (uint32_t i = 0; i < n; i++) {
      memcpy(static_cast<uint8_t*>(h_in_) + i * in_stride_, batch[i]->in, in_bytes_);
    }
    PHY_HIP_CHECK(hipMemcpyAsync(d_in_, h_in_, in_stride_ * (n - 1) + in_bytes_, hipMemcpyHostToDevice, st_), SRSRAN_ERROR);
    const int rc = run_(d_in_, d_out_, n, st_);
    if (rc != SRSRAN_SUCCESS) {
      (void)hipStreamSynchronize(st_);
      return rc;
    }
    PHY_HIP_CHECK(hipMemcpyAsync(h_out_, d_out_, out_stride_ * (n - 1) + out_bytes_, hipMemcpyDeviceToHost, st_), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipStreamSynchronize(st_), SRSRAN_ERROR);
    for (uint32_t i = 0; i < n; i++) {
      memcpy(batch[i]->out, static_cast<uint8_t*>(h_out_) + i * out_stride_, out_bytes_);
    }
    return SRSRAN_SUCCESS;
  }

  const size_t            in_bytes_, out_bytes_, in_stride_, out_stride_;
  const uint32_t          max_batch_;
  RunFn                   run_;
  hipStream_t             st_    = nullptr;
  void *                  d_in_ = nullptr, *d_out_ = nullptr, *h_in_ = nullptr, *h_out_ = nullptr;
  bool                    ok_   = false;
  std::mutex              mu_;
  std::condition_variable cv_;
  std::deque<Req*>        queue_;
  bool                    busy_      = false;
  uint64_t                n_batches_ = 0, n_units_ = 0;
};

// process-wide registry: one queue per shape key, created on first use by `make` (which returns nullptr on failure) and kept for
// the life of the process (not destroyed at exit: HIP may already be gone by then)
Coalescer* coalescer_for(const std::string& key, const std::function<Coalescer*()>& make);
bool       coalescing_enabled();

} // namespace phyhip
