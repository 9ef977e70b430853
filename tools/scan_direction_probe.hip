// scan_direction_probe.hip -- does the memory-side cache (MALL, 256 MB) give anything back to a table that is read
// whole, once per launch, launch after launch?  A cyclic scan of more bytes than a cache holds hits nothing under LRU;
// the same scan with its DIRECTION alternating from launch to launch finds the tail of the previous pass still there.
// Grid-stride 16-byte loads, U in flight per lane; passes over `bytes`: every pass forward | alternating.  Not part of
// the product.
//   hipcc -O3 --offload-arch=gfx950 tools/scan_direction_probe.hip -o tools/scan_direction_probe && tools/scan_direction_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <bool NT, int U>
__global__ __launch_bounds__(1024) void scan_kernel(const vfloat4* __restrict__ src, size_t n4, int reverse, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const size_t first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t niter = (n4 - first + stride - 1) / stride;     // (n4 is a multiple of the stride: the same for every lane)
  vfloat4 acc = {0, 0, 0, 0};
  for (size_t it = 0; it + U <= niter; it += U) {
    vfloat4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const size_t k = reverse ? niter - 1 - (it + u) : it + u;
      const vfloat4* p = &src[first + k * stride];
      v[u] = NT ? __builtin_nontemporal_load(p) : *p;
    }
#pragma unroll
    for (int u = 0; u < U; u++) acc += v[u];
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.0f;   // never true: keeps the loads alive
}

template <bool NT>
static void run(const char* label, const vfloat4* d, size_t n4, float* sink, int grid, int block, bool alternate) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  const int passes = 40;
  for (int w = 0; w < 6; w++) scan_kernel<NT, 4><<<grid, block>>>(d, n4, alternate ? (w & 1) : 0, sink);
  (void)hipDeviceSynchronize();
  float best = 1e30f, sum = 0.0f;
  for (int p = 0; p < passes; p++) {
    (void)hipEventRecord(a);
    scan_kernel<NT, 4><<<grid, block>>>(d, n4, alternate ? (p & 1) : 0, sink);
    (void)hipEventRecord(b);
    (void)hipEventSynchronize(b);
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, a, b);
    best = ms < best ? ms : best;
    sum += ms;
  }
  const double gb = 16.0 * (double)n4 / 1e9;
  std::printf("%-34s %s  %7.1f us mean  %7.1f us best  = %5.2f TB/s mean\n", label, alternate ? "alternating" : "forward    ",
              1e3 * sum / passes, 1e3 * best, gb / (sum / passes));
}

int main(int argc, char** argv) {
  const int grid = 512, block = 512;
  const size_t stride = (size_t)grid * block;
  std::vector<double> sizes = {0.405e9, 0.805e9, 0.2e9, 1.6e9};
  if (argc > 1) sizes = {std::atof(argv[1])};
  float* sink = nullptr;
  (void)hipMalloc(&sink, 4);
  for (double bytes : sizes) {
    size_t n4 = (size_t)(bytes / 16.0);
    n4 = (n4 / (4 * stride)) * (4 * stride);
    vfloat4* d = nullptr;
    if (hipMalloc(&d, n4 * 16) != hipSuccess) return 1;
    (void)hipMemset(d, 0, n4 * 16);
    std::printf("---- %.3f GB\n", 16.0 * (double)n4 / 1e9);
    run<true>("non-temporal loads", d, n4, sink, grid, block, false);
    run<true>("non-temporal loads", d, n4, sink, grid, block, true);
    run<false>("plain loads", d, n4, sink, grid, block, false);
    run<false>("plain loads", d, n4, sink, grid, block, true);
    (void)hipFree(d);
  }
  return 0;
}
