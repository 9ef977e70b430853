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

#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>

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
  void allocHost() {
    if (host_) return;
    void* p = nullptr;
    if (pinned_) {
      check(sxmc_host_alloc(&p, n_ * sizeof(T)));
    } else {
      p = ::operator new(n_ ? n_ * sizeof(T) : 1);
    }
    std::memset(p, 0, n_ * sizeof(T));
    host_ = static_cast<T*>(p);
  }
  void allocDevice() {
    if (dev_) return;
    void* p = nullptr;
    check(sxmc_malloc(&p, n_ * sizeof(T)));
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
    if (host_) {
      if (pinned_) {
        sxmc_host_free(host_);
      } else {
        ::operator delete(host_);
      }
    }
    if (dev_) sxmc_free(dev_);
    host_ = nullptr;
    dev_ = nullptr;
    host_valid_ = dev_valid_ = false;
  }

  size_t n_;
  bool pinned_;
  T* host_ = nullptr;
  T* dev_ = nullptr;
  bool host_valid_ = false;
  bool dev_valid_ = false;
};

}  // namespace sxmc
