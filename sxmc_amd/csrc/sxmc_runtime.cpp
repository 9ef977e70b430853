// sxmc_runtime.cpp -- host side of libsxmc_hip.so: errors, tracing, device properties, the lazy EvalFinished's bookkeeping, and the
// device / memory / stream / graph / event entry points of the C ABI (include/sxmc_hip.h).  There is no CPU fallback
// anywhere in the library: every evaluation entry point launches gfx950 kernels or fails with an error code.
#include "sxmc_host.h"

using namespace sxhost;

namespace sxhost {

thread_local std::string g_last_error;
thread_local bool t_capturing = false;
thread_local unsigned long long t_capture_epoch = 0;
thread_local std::vector<sxmc_group*> t_capture_groups;

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

// Host-side roctx ranges around the phases of a step (tracing): what rocprofv3 --marker-trace
// shows beside the kernel trace.  Off unless SXMC_ROCTX=1 is in the environment or sxmc_set_tracing(1) was called: a
// step makes three of them, and config 2's step is 21 us.
std::atomic<int> g_tracing{-1};
bool tracing() {
  int t = g_tracing.load(std::memory_order_relaxed);
  if (t < 0) {
    const char* e = std::getenv("SXMC_ROCTX");
    t = (e && e[0] && e[0] != '0') ? 1 : 0;
    g_tracing.store(t, std::memory_order_relaxed);
  }
  return t > 0;
}
int get_props(DeviceProps& p) {
  static thread_local DeviceProps cache;
  static thread_local int cache_dev = -1;
  int dev = 0;
  SX_HIP(hipGetDevice(&dev));
  if (!cache.valid || cache_dev != dev) {
    hipDeviceProp_t prop;
    SX_HIP(hipGetDeviceProperties(&prop, dev));
    cache.cus = prop.multiProcessorCount;
    cache.lds_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
    if (cache.lds_per_cu <= 0) cache.lds_per_cu = 160 * 1024;
    cache.valid = true;
    cache_dev = dev;
  }
  p = cache;
  return SXMC_OK;
}

void free_class(LaunchClass& c) {
  if (c.d_descs) (void)hipFree(c.d_descs);
  if (c.d_descs_sparse) (void)hipFree(c.d_descs_sparse);
  c.d_descs_sparse = nullptr;
  if (c.d_segs) (void)hipFree(c.d_segs);
  if (c.d_blk_off) (void)hipFree(c.d_blk_off);
  c.d_descs = nullptr;
  c.d_segs = nullptr;
  c.d_blk_off = nullptr;
}

// LAZY EvalFinished.  A batch launched on the legacy default stream is ordered, on the device, before everything the
// caller does next through this ABI on that stream or on any blocking stream -- its NLL kernels (mcmc.cpp:314-348),
// blocking copies to the host, the next evaluation.  So sxmc_hist_eval_finished of such a batch does not stop the host:
// it notes that the thread has an unwaited batch, and the wait happens at the first call that could tell -- one that
// names a stream which does NOT order with the legacy stream (created non-blocking), or that synchronises.  The host
// then runs ahead of the device like a caller of the group API does, instead of idling the device once per step while
// it wakes up and launches the rest of the step (measured at BASELINE config 3: ~25 us of ~190).  Never lazy: a batch
// with an output buffer the host can read directly (pinned or managed memory: the results must BE there when
// EvalFinished returns, pdfz.cpp:491-495).  What this cannot cover is device work the caller issues OUTSIDE this ABI on
// a non-blocking stream of its own right after EvalFinished, and a device fault of the batch surfaces at the call that
// waits, not at EvalFinished; sxmc_set_lazy_finish(0) (SXMC_LAZY_FINISH=0) restores the blocking wait.
std::atomic<int> g_lazy_finish{-1};
// "an EvalFinished on this device returned without waiting", per device and for the WHOLE PROCESS: the batch sits on the
// device's legacy stream, and the first call of ANY host thread that could observe the difference waits for it
// (round 4 kept this per thread: another thread's call on a non-blocking stream slipped past it -- ADVICE r4).
std::atomic<bool> g_unsettled[kMaxDevices];
int current_device_slot() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    (void)hipGetLastError();
    dev = 0;
  }
  return dev >= 0 && dev < kMaxDevices ? dev : 0;
}
inline bool unsettled_here() { return g_unsettled[current_device_slot()].load(std::memory_order_acquire); }
bool lazy_finish_enabled() {
  int v = g_lazy_finish.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = std::getenv("SXMC_LAZY_FINISH");
    v = (e && e[0] == '0') ? 0 : 1;
    g_lazy_finish.store(v, std::memory_order_relaxed);
  }
  return v > 0;
}
int settle() {
  const int dev = current_device_slot();
  if (!g_unsettled[dev].exchange(false, std::memory_order_acq_rel)) return SXMC_OK;
  SX_HIP(hipStreamSynchronize(nullptr));
  return SXMC_OK;
}
// Before work is put on stream `s`: does `s` order with the legacy stream by itself?
int settle_for(hipStream_t s) {
  if (s == nullptr || !unsettled_here()) return SXMC_OK;
  unsigned flags = 0;
  if (hipStreamGetFlags(s, &flags) == hipSuccess && !(flags & hipStreamNonBlocking)) return SXMC_OK;
  (void)hipGetLastError();
  return settle();
}
// Can the HOST read this buffer without a copy through the runtime (pinned host memory, managed memory)?  Results in
// such a buffer must be there when EvalFinished returns (pdfz.cpp:491-495 synchronises): no lazy finish for them.
bool host_can_read(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) {
    (void)hipGetLastError();
    return true;      // (not a pointer the runtime knows: be safe, wait)
  }
  return a.type != hipMemoryTypeDevice;
}

}  // namespace sxhost

// For the library's other translation units (sxmc_comm.cpp): what every entry point that puts work on a stream does
// first -- the calling thread's deferred evaluations are launched, and a batch whose EvalFinished did not wait is
// waited for unless `s` orders with it by itself.
int sx_flush_and_order(hipStream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  return SXMC_OK;
}

extern "C" {

const char* sxmc_last_error(void) { return g_last_error.c_str(); }
const char* sxmc_version(void) { return "sxmc_hip 0.1 (gfx950)"; }

int sxmc_device_count(int* count) {
  SX_REQUIRE(count, "null argument");
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    *count = 0;
    return fail(SXMC_ERR_HIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
  }
  return SXMC_OK;
}

int sxmc_set_device(int device) {
  SX_HIP(hipSetDevice(device));
  return SXMC_OK;
}

int sxmc_get_device(int* device) {
  SX_REQUIRE(device, "null argument");
  SX_HIP(hipGetDevice(device));
  return SXMC_OK;
}

int sxmc_device_info(int device, char* name, int* compute_units, size_t* hbm_bytes, int* lds_bytes_per_cu,
                     int* clock_khz) {
  hipDeviceProp_t prop;
  SX_HIP(hipGetDeviceProperties(&prop, device));
  if (name) {
    std::snprintf(name, 256, "%s (%s)", prop.name, prop.gcnArchName);
  }
  if (compute_units) *compute_units = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  if (lds_bytes_per_cu) *lds_bytes_per_cu = (int)prop.maxSharedMemoryPerMultiProcessor;
  if (clock_khz) *clock_khz = prop.clockRate;
  return SXMC_OK;
}

int sxmc_set_tracing(int enable) {
  g_tracing.store(enable ? 1 : 0, std::memory_order_relaxed);
  return SXMC_OK;
}

int sxmc_device_synchronize(void) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_HIP(hipDeviceSynchronize());
  return SXMC_OK;
}

int sxmc_device_pci_bus_id(int device, char* out, size_t out_bytes) {
  SX_REQUIRE(out && out_bytes >= 16, "buffer of at least 16 bytes");
  SX_HIP(hipDeviceGetPCIBusId(out, (int)out_bytes, device));
  return SXMC_OK;
}

int sxmc_mem_info(size_t* free_bytes, size_t* total_bytes) {
  SX_REQUIRE(free_bytes && total_bytes, "null argument");
  SX_HIP(hipMemGetInfo(free_bytes, total_bytes));
  return SXMC_OK;
}

int sxmc_malloc(void** d_ptr, size_t bytes) {
  SX_REQUIRE(d_ptr, "null argument");
  SX_HIP(hipMalloc(d_ptr, bytes ? bytes : 4));
  return SXMC_OK;
}
int sxmc_free(void* d_ptr) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (d_ptr) SX_HIP(hipFree(d_ptr));
  return SXMC_OK;
}
int sxmc_host_alloc(void** h_ptr, size_t bytes) {
  SX_REQUIRE(h_ptr, "null argument");
  SX_HIP(hipHostMalloc(h_ptr, bytes ? bytes : 4, hipHostMallocDefault));
  return SXMC_OK;
}
int sxmc_host_free(void* h_ptr) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (h_ptr) SX_HIP(hipHostFree(h_ptr));
  return SXMC_OK;
}
int sxmc_memcpy_h2d(void* d, const void* h, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
  return SXMC_OK;
}
int sxmc_memcpy_d2h(void* h, const void* d, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(h, d, n, hipMemcpyDeviceToHost));
  return SXMC_OK;
}
int sxmc_memcpy_d2d(void* d, const void* s, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemcpy(d, s, n, hipMemcpyDeviceToDevice));
  return SXMC_OK;
}
int sxmc_memcpy_h2d_async(void* d, const void* h, size_t n, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  if (n) SX_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_memcpy_d2h_async(void* h, const void* d, size_t n, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  if (n) SX_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_memset(void* d, int v, size_t n) {
  SX_FLUSH();
  if (n) SX_HIP(hipMemset(d, v, n));
  return SXMC_OK;
}

int sxmc_stream_create(sxmc_stream_t* s) {
  SX_REQUIRE(s, "null argument");
  hipStream_t st;
  SX_HIP(hipStreamCreate(&st));
  *s = st;
  return SXMC_OK;
}
int sxmc_stream_create_nonblocking(sxmc_stream_t* s) {
  SX_REQUIRE(s, "null argument");
  hipStream_t st;
  SX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  *s = st;
  return SXMC_OK;
}
int sxmc_stream_destroy(sxmc_stream_t s) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  if (s) SX_HIP(hipStreamDestroy((hipStream_t)s));
  return SXMC_OK;
}
int sxmc_stream_synchronize(sxmc_stream_t s) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_HIP(hipStreamSynchronize((hipStream_t)s));
  return SXMC_OK;
}
int sxmc_stream_query(sxmc_stream_t s, int* done) {
  SX_FLUSH();
  if (int rc_ = settle()) return rc_;
  SX_REQUIRE(done, "null argument");
  const hipError_t e = hipStreamQuery((hipStream_t)s);
  if (e == hipSuccess) {
    *done = 1;
    return SXMC_OK;
  }
  if (e == hipErrorNotReady) {
    (void)hipGetLastError();   // (not an error: clear the sticky code)
    *done = 0;
    return SXMC_OK;
  }
  SX_HIP(e);
  return SXMC_OK;
}

int sxmc_graph_begin_capture(sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_REQUIRE(s, "the legacy default stream cannot be captured: pass a created stream");
  SX_REQUIRE(!t_capturing, "a capture is already in progress on this thread");
  g_capture_gate.lock_shared();          // (waits for a batch that is being launched on the legacy stream right now)
  const hipError_t began = hipStreamBeginCapture((hipStream_t)s, hipStreamCaptureModeThreadLocal);
  if (began != hipSuccess) {
    g_capture_gate.unlock_shared();
    SX_HIP(began);
  }
  t_capturing = true;
  t_capture_epoch++;
  t_capture_groups.clear();
  return SXMC_OK;
}
int sxmc_graph_end_capture(sxmc_stream_t s, sxmc_graph_t* out) {
  SX_REQUIRE(s && out, "null argument");
  SX_REQUIRE(t_capturing, "no capture in progress on this thread");
  t_capturing = false;
  for (sxmc_group* g : t_capture_groups) g->prezeroed = 0;  // nothing recorded has run yet
  t_capture_groups.clear();
  hipGraph_t graph = nullptr;
  const hipError_t ended = hipStreamEndCapture((hipStream_t)s, &graph);
  g_capture_gate.unlock_shared();
  SX_HIP(ended);
  if (!graph) return fail(SXMC_ERR_HIP, "hipStreamEndCapture returned no graph (an error ended the capture)");
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(SXMC_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
  *out = exec;
  return SXMC_OK;
}
int sxmc_graph_launch(sxmc_graph_t graph, sxmc_stream_t s, int times) {
  SX_FLUSH();
  SX_ORDER(s);
  TraceRange trace("sxmc: graph replay");
  SX_REQUIRE(graph, "null graph");
  SX_REQUIRE(times >= 0, "negative repeat count");
  for (int i = 0; i < times; i++) SX_HIP(hipGraphLaunch((hipGraphExec_t)graph, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_graph_destroy(sxmc_graph_t graph) {
  if (graph) SX_HIP(hipGraphExecDestroy((hipGraphExec_t)graph));
  return SXMC_OK;
}

int sxmc_event_create(sxmc_event_t* e) {
  SX_REQUIRE(e, "null argument");
  hipEvent_t ev;
  SX_HIP(hipEventCreate(&ev));
  *e = ev;
  return SXMC_OK;
}
int sxmc_event_destroy(sxmc_event_t e) {
  if (e) SX_HIP(hipEventDestroy((hipEvent_t)e));
  return SXMC_OK;
}
int sxmc_event_record(sxmc_event_t e, sxmc_stream_t s) {
  SX_FLUSH();
  SX_ORDER(s);
  SX_HIP(hipEventRecord((hipEvent_t)e, (hipStream_t)s));
  return SXMC_OK;
}
int sxmc_event_synchronize(sxmc_event_t e) {
  SX_HIP(hipEventSynchronize((hipEvent_t)e));
  return SXMC_OK;
}
int sxmc_event_elapsed_ms(sxmc_event_t a, sxmc_event_t b, float* ms) {
  SX_REQUIRE(ms, "null argument");
  SX_HIP(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b));
  return SXMC_OK;
}

}  // extern "C"
