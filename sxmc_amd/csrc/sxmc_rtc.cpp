// sxmc_rtc.cpp -- run-time specialisation of the fill kernels for a group's actual program of systematics.
//
// The fill runs fastest as straight-line code with the program (which systematic writes which column, in which
// order: apply_systematic, /root/reference/src/pdfz.cpp:306-331) fixed at compile time.  The library carries such
// kernels for a handful of common programs (pdfz_kernels.hip: kStaticPrograms); for any other program it compiles
// the SAME kernel template (fill_kernels.inc.h, embedded as text) with hiprtc, once per program and process, and
// launches it through the module API.  A program that cannot be compiled (no hiprtc, a compilation error) falls
// back to the kernel that decodes the program at run time: same results, slower.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <hip/hiprtc.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "rtc_sources.h"
#include "sxmc_device.h"

namespace {

std::mutex g_mutex;
std::map<std::string, hipFunction_t> g_cache;   // key includes the device
std::map<std::string, std::string> g_failed;    // key -> why

std::string join(const char* const* pieces) {
  std::string s;
  for (int i = 0; pieces[i]; i++) s += pieces[i];
  return s;
}

std::string kernel_source(const SxRtcSpec& k) {
  std::string prog = "StaticProg<";
  for (int i = 0; i < k.nops; i++) prog += (i ? ", " : "") + std::to_string(k.ops[i]) + "u";
  prog += ">";
  std::string s = "#include \"fill_kernels.inc.h\"\nusing namespace sxfill;\n";
  if (k.pre_width == 6) {   // bucketed table with a boxed observable (fill_boxed_body)
    s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(const SxSignalDesc* __restrict__ descs, "
         "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
    s += "  fill_boxed_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog +
         ">(descs, segs, blk_off, w, dbg);\n}\n";
    return s;
  }
  if (k.pre_width == 5 && k.sparse_runs) {   // ... its sparse counting over runs
    s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(const SxSignalDesc* __restrict__ descs, "
         "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
    s += "  fill_sparse_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog +
         ", true>(descs, segs, blk_off, w, dbg);\n}\n";
    return s;
  }
  if (k.pre_width == 5 && !k.lds_hist) {     // ... its dense evaluation with a histogram beyond LDS
    s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(const SxSignalDesc* __restrict__ descs, "
         "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
    s += "  SxChainDescs one;\n  one.d[0] = one.d[1] = one.d[2] = one.d[3] = descs;\n";
    s += "  fill_ordered_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog +
         ", 1, false>(one, segs, blk_off, w, dbg);\n}\n";
    return s;
  }
  if (k.pre_width == 5) {   // bucketed table with an ordered observable (fill_ordered_body), 1 to 4 chains
    const std::string targs = std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog + ", " +
                              std::to_string(k.nchain > 1 ? k.nchain : 1);
    if (k.nchain > 1) {
      const int bound = k.max_threads > 0 ? k.max_threads : 1024;
      s += "extern \"C\" __global__ __launch_bounds__(" + std::to_string(bound) + ") void sx_rtc_fill(SxChainDescs chains, "
           "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
      s += "  fill_ordered_body<" + targs + ">(chains, segs, blk_off, w, dbg);\n}\n";
    } else {
      s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(const SxSignalDesc* __restrict__ descs, "
           "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
      s += "  SxChainDescs one;\n  one.d[0] = one.d[1] = one.d[2] = one.d[3] = descs;\n";
      s += "  fill_ordered_body<" + targs + ">(one, segs, blk_off, w, dbg);\n}\n";
    }
    return s;
  }
  if (k.nchain > 1) {
    s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(SxChainDescs chains, "
         "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned) {\n";
    s += "  fill_multi_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog + ", " +
         std::to_string(k.pre_width) + ", " + std::to_string(k.nchain) + ">(chains, segs, blk_off, w);\n}\n";
    return s;
  }
  s += "extern \"C\" __global__ __launch_bounds__(1024) void sx_rtc_fill(const SxSignalDesc* __restrict__ descs, "
       "const SxSegment* __restrict__ segs, const unsigned* __restrict__ blk_off, unsigned w, unsigned dbg) {\n";
  if (k.sparse_runs) {
    s += "  fill_sparse_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " + prog +
         ">(descs, segs, blk_off, w, dbg);\n";
  } else {
    s += "  fill_body<" + std::to_string(k.nobs) + ", " + std::to_string(k.nslot) + ", " +
         (k.lds_hist ? "true" : "false") + ", " + prog + ", " + std::to_string(k.pre_width) +
         ">(descs, segs, blk_off, w, dbg);\n";
  }
  s += "}\n";
  return s;
}

std::string spec_key(const SxRtcSpec& k) {
  std::string s = std::to_string(k.nobs) + "/" + std::to_string(k.nslot) + "/" + std::to_string(k.lds_hist) + "/" +
                  std::to_string(k.pre_width) + "/" + std::to_string(k.sparse_runs) + "/" + std::to_string(k.nchain) + "/" +
                  std::to_string(k.max_threads) + ":";
  for (int i = 0; i < k.nops; i++) s += std::to_string(k.ops[i]) + ",";
  return s;
}

// Compiles the specialisation to a gfx950 code object.  Needs no device.
bool compile(const SxRtcSpec& k, std::vector<char>& code, std::string& err) {
  code.clear();
  const std::string src = kernel_source(k), types = join(kRtcTypesSrc), fill = join(kRtcFillSrc);
  const char* headers[] = {types.c_str(), fill.c_str()};
  const char* names[] = {"sxmc_device_types.h", "fill_kernels.inc.h"};
  hiprtcProgram prog = nullptr;
  hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "sx_rtc_fill.hip", 2, headers, names);
  if (r != HIPRTC_SUCCESS) {
    err = std::string("hiprtcCreateProgram: ") + hiprtcGetErrorString(r);
    return false;
  }
  // the flags of the library's own build (Makefile): double add/mul must round separately (bit-exact bin indices)
  // (SXMC_ARCH: the Makefile's ARCH, so that the run-time kernels are built for what the library was built for)
  std::vector<const char*> opts = {"--offload-arch=" SXMC_ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math"};
  // (a variant build of the library -- make VARIANT=.. EXTRA=-D.. -- compiles its run-time kernels the same way)
#if defined(SXMC_CACHED_LOADS) && SXMC_CACHED_LOADS
  opts.push_back("-DSXMC_CACHED_LOADS=1");
#endif
#if defined(SXMC_MEASURE) && SXMC_MEASURE
  opts.push_back("-DSXMC_MEASURE=1");   // (the measurement build's run-time kernels carry the hooks too)
#endif
  r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
  if (r != HIPRTC_SUCCESS) {
    size_t n = 0;
    (void)hiprtcGetProgramLogSize(prog, &n);
    std::string log(n, '\0');
    if (n) (void)hiprtcGetProgramLog(prog, &log[0]);
    err = std::string("hiprtcCompileProgram: ") + hiprtcGetErrorString(r) + "\n" + log;
    (void)hiprtcDestroyProgram(&prog);
    return false;
  }
  size_t n = 0;
  r = hiprtcGetCodeSize(prog, &n);
  if (r == HIPRTC_SUCCESS) {
    code.resize(n);
    r = hiprtcGetCode(prog, code.data());
  }
  (void)hiprtcDestroyProgram(&prog);
  if (r != HIPRTC_SUCCESS) {
    code.clear();   // (the caller caches `code` per specialisation: a half-filled image must not be loaded on the next device)
    err = std::string("hiprtcGetCode: ") + hiprtcGetErrorString(r);
    return false;
  }
  return true;
}

}  // namespace

bool sx_rtc_compile_only(const SxRtcSpec& k, size_t* code_bytes, std::string* err) {
  std::vector<char> code;
  std::string e;
  const bool ok = compile(k, code, e);
  if (code_bytes) *code_bytes = code.size();
  if (err) *err = e;
  return ok;
}

// The kernel for a specialisation on the current device: from the cache, or compiled and loaded now.
// nullptr (and *err) when it cannot be had; the failure is remembered, so a program is tried once.
void* sx_rtc_get(const SxRtcSpec& k, std::string* err) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    if (err) *err = "no current device";
    return nullptr;
  }
  const std::string key = std::to_string(dev) + "|" + spec_key(k);
  std::lock_guard<std::mutex> lock(g_mutex);
  auto hit = g_cache.find(key);
  if (hit != g_cache.end()) return hit->second;
  auto bad = g_failed.find(key);
  if (bad != g_failed.end()) {
    if (err) *err = bad->second;
    return nullptr;
  }
  // the code object is compiled once per process and specialisation (every GPU of the node is a gfx950); each
  // device loads its own module from it
  static std::map<std::string, std::vector<char>> g_code;
  std::string e;
  hipFunction_t fn = nullptr;
  std::vector<char>& code = g_code[spec_key(k)];
  if (!code.empty() || compile(k, code, e)) {
    hipModule_t mod = nullptr;
    hipError_t he = hipModuleLoadData(&mod, code.data());
    if (he == hipSuccess) he = hipModuleGetFunction(&fn, mod, "sx_rtc_fill");
    if (he != hipSuccess) {
      e = std::string("loading the compiled kernel: ") + hipGetErrorString(he);
      fn = nullptr;
    }
  }
  if (!fn) {
    g_failed[key] = e;
    if (err) *err = e;
    return nullptr;
  }
  g_cache[key] = fn;   // (the module stays loaded for the life of the process)
  return fn;
}

hipError_t sx_rtc_launch(void* fn, int grid, int threads, size_t lds_bytes, const SxSignalDesc* descs,
                         const SxSegment* segs, const unsigned* blk_off, unsigned w, unsigned dbg, hipStream_t s,
                         void* ev_start, void* ev_stop) {
  void* args[] = {(void*)&descs, (void*)&segs, (void*)&blk_off, (void*)&w, (void*)&dbg};
  if (ev_start && ev_stop) {   // profiled launch: the two events carry the dispatch's own begin and end
    return hipExtModuleLaunchKernel((hipFunction_t)fn, (unsigned)grid * (unsigned)threads, 1, 1, (unsigned)threads, 1, 1,
                                    lds_bytes, s, args, nullptr, (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 0);
  }
  return hipModuleLaunchKernel((hipFunction_t)fn, (unsigned)grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds_bytes, s,
                               args, nullptr);
}

hipError_t sx_rtc_launch_multi(void* fn, int grid, int threads, size_t lds_bytes, const SxChainDescsHost& chains,
                               const SxSegment* segs, const unsigned* blk_off, unsigned hist_words, unsigned dbg,
                               hipStream_t s, void* ev_start, void* ev_stop) {
  SxChainDescsHost c = chains;
  void* args[] = {(void*)&c, (void*)&segs, (void*)&blk_off, (void*)&hist_words, (void*)&dbg};
  if (ev_start && ev_stop) {
    return hipExtModuleLaunchKernel((hipFunction_t)fn, (unsigned)grid * (unsigned)threads, 1, 1, (unsigned)threads, 1, 1,
                                    lds_bytes, s, args, nullptr, (hipEvent_t)ev_start, (hipEvent_t)ev_stop, 0);
  }
  return hipModuleLaunchKernel((hipFunction_t)fn, (unsigned)grid, 1, 1, (unsigned)threads, 1, 1, (unsigned)lds_bytes, s,
                               args, nullptr);
}
