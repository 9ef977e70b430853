// device_array.h -- sxmc::DeviceArray<T>: the host/device mirrored array the reference's callers use
// (it was hemi::Array<T>, from the absent contrib/hemi submodule), implemented natively on the HIP
// runtime through the C ABI of libsxmc_hip.so.  There is no CUDA path and no host-only mode.
//
// Semantics the reference's call sites rely on (e.g. mcmc.cpp:159-198, 351-377,
// pdfz.cpp:23-32, test_pdfz.cpp:98-126):
//   * constructed (n, pinned) with no allocation; both sides are allocated lazily;
//   * each side has a validity flag; read accessors copy from the other side if that one is newer;
//   * writeOnly*Ptr() marks its side as the only valid one WITHOUT copying;
//   * ptr() is read-write on the default side, which is the DEVICE here (as under nvcc);
//   * hostPtr() is read-write on the host; copyFromHost(p, n) re-sizes and fills the host side.
#pragma once

#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "../../../include/sxmc_hip.h"

namespace sxmc {

struct HipError : std::runtime_error {
  explicit HipError(const std::string& what) : std::runtime_error(what) {}
};

inline void check(int rc) {
  if (rc != SXMC_OK) throw HipError(std::string("libsxmc_hip: ") + sxmc_last_error());
}

/** The stream the calling thread's array transfers are ordered on.  Default (null): blocking copies
 *  through the legacy default stream, which wait for every blocking stream of the device -- the
 *  reference's single-chain behaviour.  A thread that drives its own chain on a non-blocking stream sets
 *  this to that stream: its transfers are then enqueued there and waited for there, and do not touch
 *  the chains of other threads. */
inline sxmc_stream_t& transfer_stream() {
  static thread_local sxmc_stream_t s = nullptr;
  return s;
}

/** BLOCK POOL.  hipMalloc / hipFree / hipHostMalloc / hipHostFree are device-wide events (hipFree waits until every
 *  stream of the device is idle), and a walk of the reference's shape allocates about thirty blocks at its start
 *  and frees them at its end -- while holding the device's set-up lock, beside the other chains of an ensemble.
 *  While a PoolScope is alive, the arrays of every thread give their blocks back to this pool instead of freeing them
 *  and take blocks from it instead of allocating: after the first experiments an ensemble allocates nothing.  Blocks
 *  are kept per (device, side) in size classes (powers of two from 256 bytes); a block goes back only from a thread
 *  that is not unwinding (an error path frees for real: the runtime's wait is the only ordering it has left).
 *  The last PoolScope to end frees everything.  Without a PoolScope (a single walk; the reference's shape) arrays
 *  allocate and free as before.  Measured (sxmc::ensemble_lockstep, config 3, 32 experiments of 2 000 steps, alternating
 *  runs with SXMC_BLOCK_POOL=0 / 1 on one box): 9 660-9 840 against 9 210-9 590 steps/s with 2 sets of 4 chains,
 *  8 850-8 950 against 8 630-8 690 with 4 sets of 2 -- about 3 %.  (What else was tried against the set-up cost of short
 *  experiments and did NOT pay: a bound on the graph replays a chain keeps queued, high-priority streams for the
 *  evaluators' own set-up work, a second set of evaluators per lane with fixed bindings and a chain object kept across
 *  experiments.  What a short experiment loses beside walking chains is its ~100 small synchronous launches, each of
 *  which takes its turn behind 170-340 us fill kernels of the other chains: bench_cpp prints the phases.) */
class BlockPool {
 public:
  static BlockPool& instance() {
    static BlockPool* pool = new BlockPool;   // (never destroyed: the HIP runtime may be gone before static destructors run)
    return *pool;
  }
  static size_t size_class(size_t bytes) {
    size_t c = 256;
    while (c < bytes) c <<= 1;
    return c;
  }
  bool active() const { return users_.load(std::memory_order_acquire) > 0; }
  /** a block of at least `bytes` for the calling thread's device, or null */
  void* take(bool host, size_t bytes, int device) {
    std::lock_guard<std::mutex> guard(m_);
    std::vector<void*>& v = free_[std::make_tuple(device, host, size_class(bytes))];
    if (v.empty()) return nullptr;
    void* p = v.back();
    v.pop_back();
    return p;
  }
  void give(bool host, size_t bytes, int device, void* p) {
    std::lock_guard<std::mutex> guard(m_);
    free_[std::make_tuple(device, host, size_class(bytes))].push_back(p);
  }
  void enter() { users_.fetch_add(1, std::memory_order_acq_rel); }
  void leave() {
    if (users_.fetch_sub(1, std::memory_order_acq_rel) == 1) drain();
  }
  /** frees every pooled block (each on its own device); the calling thread's current device is restored */
  void drain() {
    std::map<std::tuple<int, bool, size_t>, std::vector<void*>> all;
    {
      std::lock_guard<std::mutex> guard(m_);
      all.swap(free_);
    }
    int here = -1;
    if (sxmc_get_device(&here) != SXMC_OK) here = -1;
    for (auto& kv : all) {
      if (kv.second.empty()) continue;
      sxmc_set_device(std::get<0>(kv.first));
      for (void* p : kv.second) {
        if (std::get<1>(kv.first)) sxmc_host_free(p);
        else sxmc_free(p);
      }
    }
    if (here >= 0) sxmc_set_device(here);
  }

 private:
  std::mutex m_;
  std::map<std::tuple<int, bool, size_t>, std::vector<void*>> free_;
  std::atomic<int> users_{0};
};

/** While one is alive, DeviceArrays recycle their blocks through the BlockPool (see there).  The ensemble runners hold
 *  one for their duration. */
struct PoolScope {
  static bool enabled() {
    static const bool on = [] {
      const char* e = std::getenv("SXMC_BLOCK_POOL");   // (0: allocate and free as without a pool -- measurement)
      return !(e && e[0] == '0');
    }();
    return on;
  }
  PoolScope() {
    if (enabled()) BlockPool::instance().enter();
  }
  ~PoolScope() {
    if (enabled()) BlockPool::instance().leave();
  }
  PoolScope(const PoolScope&) = delete;
  PoolScope& operator=(const PoolScope&) = delete;
};

template <typename T>
class DeviceArray {
 public:
  explicit DeviceArray(size_t n = 0, bool pinned = false) : n_(n), pinned_(pinned) {}
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  ~DeviceArray() { release(); }

  size_t size() const { return n_; }

  void copyFromHost(const T* src, size_t n) {
    if (n != n_) {
      release();
      n_ = n;
    }
    allocHost();
    if (n_) std::memcpy(host_, src, n_ * sizeof(T));
    host_valid_ = true;
    dev_valid_ = false;
  }

  // ---- host side
  T* hostPtr() {  // read-write
    toHost();
    dev_valid_ = false;
    return host_;
  }
  const T* readOnlyHostPtr() {
    toHost();
    return host_;
  }
  T* writeOnlyHostPtr() {
    allocHost();
    host_valid_ = true;
    dev_valid_ = false;
    return host_;
  }

  // ---- device side (the default side)
  T* ptr() {  // read-write
    toDevice();
    host_valid_ = false;
    return dev_;
  }
  const T* readOnlyPtr() { return readOnlyDevicePtr(); }
  T* writeOnlyPtr() { return writeOnlyDevicePtr(); }
  const T* readOnlyDevicePtr() {
    toDevice();
    return dev_;
  }
  T* writeOnlyDevicePtr() {
    allocDevice();
    dev_valid_ = true;
    host_valid_ = false;
    return dev_;
  }

 private:
  // pooled blocks are whole size classes, so that a block fits every array of its class
  static int current_device() {
    int d = 0;
    check(sxmc_get_device(&d));
    return d;
  }
  void allocHost() {
    if (host_) return;
    void* p = nullptr;
    if (pinned_) {
      BlockPool& pool = BlockPool::instance();
      if (pool.active()) {
        host_device_ = current_device();
        host_pooled_ = true;
        p = pool.take(true, n_ * sizeof(T), host_device_);
        if (!p) check(sxmc_host_alloc(&p, BlockPool::size_class(n_ * sizeof(T))));
      } else {
        check(sxmc_host_alloc(&p, n_ * sizeof(T)));
      }
    } else {
      p = ::operator new(n_ ? n_ * sizeof(T) : 1);
    }
    std::memset(p, 0, n_ * sizeof(T));
    host_ = static_cast<T*>(p);
  }
  void allocDevice() {
    if (dev_) return;
    void* p = nullptr;
    BlockPool& pool = BlockPool::instance();
    if (pool.active()) {
      dev_device_ = current_device();
      dev_pooled_ = true;
      p = pool.take(false, n_ * sizeof(T), dev_device_);
      if (!p) check(sxmc_malloc(&p, BlockPool::size_class(n_ * sizeof(T))));
    } else {
      check(sxmc_malloc(&p, n_ * sizeof(T)));
    }
    dev_ = static_cast<T*>(p);
  }
  void toHost() {
    allocHost();
    if (!host_valid_ && dev_valid_) {
      if (sxmc_stream_t s = transfer_stream()) {
        check(sxmc_memcpy_d2h_async(host_, dev_, n_ * sizeof(T), s));
        check(sxmc_stream_synchronize(s));
      } else {
        check(sxmc_memcpy_d2h(host_, dev_, n_ * sizeof(T)));
      }
    }
    host_valid_ = true;
  }
  void toDevice() {
    allocDevice();
    if (!dev_valid_) {
      allocHost();  // a never-written array uploads zeros
      if (sxmc_stream_t s = transfer_stream()) {
        check(sxmc_memcpy_h2d_async(dev_, host_, n_ * sizeof(T), s));
        check(sxmc_stream_synchronize(s));  // the host side may be rewritten as soon as this returns
      } else {
        check(sxmc_memcpy_h2d(dev_, host_, n_ * sizeof(T)));
      }
    }
    dev_valid_ = true;
  }
  void release() {
    // a pooled block (a whole size class) goes back to the pool while one is active and this thread is not
    // unwinding; otherwise it is freed for real -- and then the runtime waits for the device, as it always did
    BlockPool& pool = BlockPool::instance();
    const bool recycle = pool.active() && std::uncaught_exceptions() == 0;
    if (host_) {
      if (pinned_) {
        if (host_pooled_ && recycle) pool.give(true, n_ * sizeof(T), host_device_, host_);
        else sxmc_host_free(host_);
      } else {
        ::operator delete(host_);
      }
    }
    if (dev_) {
      if (dev_pooled_ && recycle) pool.give(false, n_ * sizeof(T), dev_device_, dev_);
      else sxmc_free(dev_);
    }
    host_ = nullptr;
    dev_ = nullptr;
    host_pooled_ = dev_pooled_ = false;
    host_valid_ = dev_valid_ = false;
  }

  size_t n_;
  bool pinned_;
  T* host_ = nullptr;
  T* dev_ = nullptr;
  bool host_valid_ = false;
  bool dev_valid_ = false;
  bool host_pooled_ = false, dev_pooled_ = false;   // the block is a whole size class (it may go to the pool)
  int host_device_ = 0, dev_device_ = 0;            // ... of this device
};

}  // namespace sxmc
