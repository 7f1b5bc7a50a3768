// coalesce.h -- per-process submission queues under the unchanged handle API.
//
// The reference's threading contract is one worker thread per in-flight subframe, each with its OWN handles
// (srsenb/src/phy/lte/cc_worker.cc:212-231, lib/include/srsran/common/thread_pool.h:48): N threads call
// srsran_tdec_run_all / srsran_ldpc_decoder_decode_c at the same time on N different handles.  One call is
// one code block or one subframe -- far too little to fill 256 CUs -- so calls of the same SHAPE (same kernel configuration)
// that are in flight together are merged into one batch launch, group-commit style:
//   * a caller queues its request; if one of the queue's few LANES (stream + staging buffers + batch engine) is free it becomes
//     the leader of the next batch on that lane, takes everything that is queued (its own request included), stages the inputs
//     in pinned memory, does ONE upload, ONE launch, ONE download and wakes the others;
//   * callers that arrive while every lane is on the device wait in the queue and form the next batch.
// Nobody ever waits for a timer: with few callers every call runs at once on its own lane (the decoder kernels are latency
// bound -- one wave per code word -- so a handful of small launches overlap perfectly); the busier the process, the larger the
// batches, and the driver sees at most that many submitting threads per shape however many workers there are.
// Lanes per shape: 4 for the turbo decoder; 1 for LDPC, which also keeps its private streams while at most four callers are inside the
// decoder (measured, profiles/r02_bench_handle.json).
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
  // builds the run function of lane `l` (every lane owns its engine: batches of different lanes are on the device together)
  using MakeFn = std::function<RunFn(int lane)>;
  static constexpr int kMaxLanes = 4;

  // n_lanes: batches of this shape that may be on the device together (1 ... kMaxLanes)
  Coalescer(size_t in_bytes, size_t out_bytes, uint32_t max_batch, int n_lanes, const MakeFn& make) :
    in_bytes_(in_bytes), out_bytes_(out_bytes), in_stride_(stride_of(in_bytes)), out_stride_(stride_of(out_bytes)), max_batch_(max_batch)
  {
    ok_ = true;
    n_lanes = n_lanes < 1 ? 1 : (n_lanes > kMaxLanes ? kMaxLanes : n_lanes);
    for (int l = 0; l < n_lanes && ok_; l++) {
      Lane& ln = lanes_[l];
      ok_      = hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&ln.d_in, in_stride_ * max_batch) == hipSuccess &&
            hipMalloc(&ln.d_out, out_stride_ * max_batch) == hipSuccess && hipHostMalloc(&ln.h_in, in_stride_ * max_batch) == hipSuccess &&
            hipHostMalloc(&ln.h_out, out_stride_ * max_batch) == hipSuccess;
      if (ok_) {
        ln.run = make(l);
        ok_    = (bool)ln.run;
      }
      free_.push_back(l);
    }
  }
  // distance between the units of a batch in the staging buffers: the unit size rounded up to 64 bytes
  static size_t stride_of(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
  bool          ok() const { return ok_; }

  // blocking: `in` (in_bytes) -> `out` (out_bytes), host memory of the caller.  Returns the run function's code.
  int submit(const void* in, void* out)
  {
    bind_thread();
    Req                          r{in, out, 0, false, false};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&r);
    while (!r.done) {
      if (r.taken || free_.empty()) {
        cv_.wait(lk); // somebody else carries this request, or every lane is on the device: the queue grows meanwhile
        continue;
      }
      // leader of the next batch on a free lane: the oldest requests first (this one is among them)
      const int l = free_.back();
      free_.pop_back();
      std::vector<Req*> batch;
      while (!queue_.empty() && batch.size() < max_batch_) {
        queue_.front()->taken = true;
        batch.push_back(queue_.front());
        queue_.pop_front();
      }
      lk.unlock();
      const int rc = process(lanes_[l], batch);
      lk.lock();
      for (Req* b : batch) {
        b->rc   = rc;
        b->done = true;
      }
      n_batches_++;
      n_units_ += batch.size();
      free_.push_back(l);
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
    bool        taken;
  };
  struct Lane {
    hipStream_t st = nullptr;
    void *      d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    RunFn       run;
  };
  int process(Lane& ln, const std::vector<Req*>& batch)
  {
    const uint32_t n = (uint32_t)batch.size();
    for (uint32_t i = 0; i < n; i++) {
      memcpy(static_cast<uint8_t*>(ln.h_in) + i * in_stride_, batch[i]->in, in_bytes_);
    }
    PHY_HIP_CHECK(hipMemcpyAsync(ln.d_in, ln.h_in, in_stride_ * (n - 1) + in_bytes_, hipMemcpyHostToDevice, ln.st), SRSRAN_ERROR);
    const int rc = ln.run(ln.d_in, ln.d_out, n, ln.st);
    if (rc != SRSRAN_SUCCESS) {
      (void)hipStreamSynchronize(ln.st);
      return rc;
    }
    PHY_HIP_CHECK(hipMemcpyAsync(ln.h_out, ln.d_out, out_stride_ * (n - 1) + out_bytes_, hipMemcpyDeviceToHost, ln.st), SRSRAN_ERROR);
    PHY_HIP_CHECK(hipStreamSynchronize(ln.st), SRSRAN_ERROR);
    for (uint32_t i = 0; i < n; i++) {
      memcpy(batch[i]->out, static_cast<uint8_t*>(ln.h_out) + i * out_stride_, out_bytes_);
    }
    return SRSRAN_SUCCESS;
  }

  const size_t            in_bytes_, out_bytes_, in_stride_, out_stride_;
  const uint32_t          max_batch_;
  Lane                    lanes_[kMaxLanes];
  std::vector<int>        free_;
  bool                    ok_ = false;
  std::mutex              mu_;
  std::condition_variable cv_;
  std::deque<Req*>        queue_;
  uint64_t                n_batches_ = 0, n_units_ = 0;
};

// process-wide registry: one queue per shape key, created on first use by `make` (which returns nullptr on failure) and kept for
// the life of the process (not destroyed at exit: HIP may already be gone by then)
Coalescer* coalescer_for(const std::string& key, const std::function<Coalescer*()>& make);
bool       coalescing_enabled();

} // namespace phyhip
