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
// A request carries a TAG (run-time parameters that do not change the engine, e.g. the LDPC rate-matched length and CRC): only requests
// with equal tags share a batch, so the registry is keyed on what the kernel configuration needs and nothing else.
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
  // runs n units: unit i reads d_in + i * in_stride and writes d_out + i * out_stride (bytes); asynchronous on `st`.  `tag` is the
  // grouping key the units of this batch share (run-time parameters that do not change the engine: LDPC rate-matched length and CRC)
  using RunFn = std::function<int(const void* d_in, void* d_out, uint32_t n, uint64_t tag, hipStream_t st)>;
  // one lane's engine: the run function and what releases the engine when the queue is evicted from the registry
  struct Engine {
    RunFn                 run;
    std::function<void()> destroy;
  };
  // builds the engine of lane `l` (every lane owns its engine: batches of different lanes are on the device together)
  using MakeFn = std::function<Engine(int lane)>;
  static constexpr int kMaxLanes = 4;

  // n_lanes: batches of this shape that may be on the device together (1 ... kMaxLanes)
  Coalescer(size_t in_bytes, size_t out_bytes, uint32_t max_batch, int n_lanes, const MakeFn& make) :
    in_bytes_(in_bytes), out_bytes_(out_bytes), in_stride_(stride_of(in_bytes)), out_stride_(stride_of(out_bytes)), max_batch_(max_batch)
  {
    bind_thread(); // the lanes' streams and buffers must live on the process's device whichever thread gets here first
    ok_ = true;
    n_lanes = n_lanes < 1 ? 1 : (n_lanes > kMaxLanes ? kMaxLanes : n_lanes);
    for (int l = 0; l < n_lanes && ok_; l++) {
      Lane& ln = lanes_[l];
      ok_      = hipStreamCreateWithFlags(&ln.st, hipStreamNonBlocking) == hipSuccess && hipMalloc(&ln.d_in, in_stride_ * max_batch) == hipSuccess &&
            hipMalloc(&ln.d_out, out_stride_ * max_batch) == hipSuccess && hipHostMalloc(&ln.h_in, in_stride_ * max_batch) == hipSuccess &&
            hipHostMalloc(&ln.h_out, out_stride_ * max_batch) == hipSuccess;
      if (ok_) {
        ln.eng = make(l);
        ok_    = (bool)ln.eng.run;
      }
      free_.push_back(l);
    }
  }
  ~Coalescer()
  {
    for (Lane& ln : lanes_) {
      if (ln.st) {
        (void)hipStreamSynchronize(ln.st);
      }
      if (ln.eng.destroy) {
        ln.eng.destroy();
      }
      (void)hipFree(ln.d_in);
      (void)hipFree(ln.d_out);
      (void)hipHostFree(ln.h_in);
      (void)hipHostFree(ln.h_out);
      if (ln.st) {
        (void)hipStreamDestroy(ln.st);
      }
    }
  }
  Coalescer(const Coalescer&)            = delete;
  Coalescer& operator=(const Coalescer&) = delete;
  // distance between the units of a batch in the staging buffers: the unit size rounded up to 64 bytes
  static size_t stride_of(size_t bytes) { return (bytes + 63) & ~(size_t)63; }
  bool          ok() const { return ok_; }

  // blocking: `in` (in_bytes) -> `out` (out_bytes), host memory of the caller.  Only requests with equal tags share a batch.
  // Returns the run function's code.
  int submit(const void* in, void* out, uint64_t tag = 0)
  {
    bind_thread();
    Req                          r{in, out, tag, 0, false, false};
    std::unique_lock<std::mutex> lk(mu_);
    queue_.push_back(&r);
    while (!r.done) {
      if (r.taken || free_.empty()) {
        cv_.wait(lk); // somebody else carries this request, or every lane is on the device: the queue grows meanwhile
        continue;
      }
      // leader of the next batch on a free lane: the oldest request and everything queued that shares its tag (this thread's own
      // request is among them or leads a later batch)
      const int l = free_.back();
      free_.pop_back();
      std::vector<Req*> batch;
      const uint64_t    t = queue_.front()->tag;
      for (auto it = queue_.begin(); it != queue_.end() && batch.size() < max_batch_;) {
        if ((*it)->tag == t) {
          (*it)->taken = true;
          batch.push_back(*it);
          it = queue_.erase(it);
        } else {
          ++it;
        }
      }
      lk.unlock();
      const int rc = process(lanes_[l], batch, t);
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
    uint64_t    tag;
    int         rc;
    bool        done;
    bool        taken;
  };
  struct Lane {
    hipStream_t st = nullptr;
    void *      d_in = nullptr, *d_out = nullptr, *h_in = nullptr, *h_out = nullptr;
    Engine      eng;
  };
  int process(Lane& ln, const std::vector<Req*>& batch, uint64_t tag)
  {
    const uint32_t n = (uint32_t)batch.size();
    for (uint32_t i = 0; i < n; i++) {
      memcpy(static_cast<uint8_t*>(ln.h_in) + i * in_stride_, batch[i]->in, in_bytes_);
    }
    int rc = SRSRAN_SUCCESS;
    if (hipMemcpyAsync(ln.d_in, ln.h_in, in_stride_ * (n - 1) + in_bytes_, hipMemcpyHostToDevice, ln.st) != hipSuccess) {
      set_error("submission queue: upload failed");
      rc = SRSRAN_ERROR;
    }
    if (rc == SRSRAN_SUCCESS) {
      rc = ln.eng.run(ln.d_in, ln.d_out, n, tag, ln.st);
    }
    if (rc == SRSRAN_SUCCESS && hipMemcpyAsync(ln.h_out, ln.d_out, out_stride_ * (n - 1) + out_bytes_, hipMemcpyDeviceToHost, ln.st) != hipSuccess) {
      set_error("submission queue: download failed");
      rc = SRSRAN_ERROR;
    }
    // whatever happened, the lane's staging buffers are reused by the next batch: nothing may still be in flight on its stream
    if (hipStreamSynchronize(ln.st) != hipSuccess && rc == SRSRAN_SUCCESS) {
      set_error("submission queue: stream synchronisation failed");
      rc = SRSRAN_ERROR;
    }
    if (rc != SRSRAN_SUCCESS) {
      return rc;
    }
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

// process-wide registry: one queue per shape key, created on first use by `make` (which returns nullptr on failure; the failure is
// remembered and callers fall back to their private path).  Creation runs outside the registry lock (under a lock of the entry), so
// submissions of other shapes are never held up by a hipMalloc.  The registry keeps at most kMaxShapes queues: the least recently
// used one that nobody holds is released (streams, staging buffers, engines) when a new shape arrives -- a long-running process that
// walks through many block sizes / lifting sizes does not grow without bound.  The returned reference keeps the queue alive while
// the caller uses it.
constexpr size_t           kMaxShapes = 12;
std::shared_ptr<Coalescer> coalescer_for(const std::string& key, const std::function<Coalescer*()>& make);
bool                       coalescing_enabled();
size_t                     coalescer_shapes(); // live queues (tests)

} // namespace phyhip
