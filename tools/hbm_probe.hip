// hbm_probe.hip -- what a plain streaming READ of the fill kernel's size reaches on the box at hand.
// Calibrates the "achievable" side of the roofline in DESIGN.md: grid-stride 16-byte loads (default and
// nontemporal policy) over one buffer, a few shapes, best of several launches.  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_probe.hip -o tools/hbm_probe && tools/hbm_probe [bytes=1.3e9]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float vfloat4 __attribute__((ext_vector_type(4)));

template <bool NT, int UNROLL>
__global__ __launch_bounds__(1024) void read_kernel(const vfloat4* __restrict__ src, size_t n4, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  vfloat4 acc = {0, 0, 0, 0};
  for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
    vfloat4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) v[u] = NT ? __builtin_nontemporal_load(&src[i + u * stride]) : src[i + u * stride];
#pragma unroll
    for (int u = 0; u < UNROLL; u++) acc += v[u];
  }
  for (; i < n4; i += stride) acc += NT ? __builtin_nontemporal_load(&src[i]) : src[i];
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = 1.0f;  // never true: keeps the loads alive
}

// the fill kernel's shape: three 16-byte columns and one 4-byte column read in lockstep, U units in flight
template <int U>
__global__ __launch_bounds__(1024) void columns_kernel(const vfloat4* __restrict__ a, const vfloat4* __restrict__ b,
                                                       const vfloat4* __restrict__ c, const unsigned* __restrict__ p,
                                                       size_t n, float* sink) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  vfloat4 acc = {0, 0, 0, 0};
  unsigned pa = 0;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    vfloat4 va[U], vb[U], vc[U];
    unsigned vp[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      va[u] = __builtin_nontemporal_load(&a[i + u * stride]);
      vb[u] = __builtin_nontemporal_load(&b[i + u * stride]);
      vc[u] = __builtin_nontemporal_load(&c[i + u * stride]);
      vp[u] = __builtin_nontemporal_load(&p[i + u * stride]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      acc += va[u] + vb[u] + vc[u];
      pa += vp[u];
    }
  }
  if (acc.x + acc.y + acc.z + acc.w + (float)pa == 12345.678f) *sink = 1.0f;
}

template <int U>
static double columns_ms(const vfloat4* d, size_t n, float* sink, int grid, int block) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  double best = 1e30;
  for (int r = 0; r < 12; r++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((columns_kernel<U>), dim3(grid), dim3(block), 0, 0, d, d + n, d + 2 * n,
                       reinterpret_cast<const unsigned*>(d + 3 * n), n, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r >= 2 && ms < best) best = ms;
  }
  return best;
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e = (x);                                                           \
    if (e != hipSuccess) {                                                        \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                 \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

template <bool NT, int UNROLL>
static double best_ms(const vfloat4* d, size_t n4, float* sink, int grid, int block, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  double best = 1e30;
  for (int r = 0; r < reps + 2; r++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL((read_kernel<NT, UNROLL>), dim3(grid), dim3(block), 0, 0, d, n4, sink);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (r >= 2 && ms < best) best = ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return best;
}

int main(int argc, char** argv) {
  const double bytes = argc > 1 ? std::atof(argv[1]) : 1.3e9;
  const size_t n4 = (size_t)(bytes / 16);
  vfloat4* d = nullptr;
  float* sink = nullptr;
  CHECK(hipMalloc((void**)&d, n4 * 16));
  CHECK(hipMalloc((void**)&sink, 4));
  CHECK(hipMemset(d, 0, n4 * 16));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("{\"device\": \"%s\", \"compute_units\": %d, \"bytes\": %.0f, \"results\": [", prop.name, cus, (double)n4 * 16);
  bool first = true;
  for (int block : {256, 512, 1024}) {
    for (int per_cu : {1, 2, 4, 8}) {
      if (block * per_cu > 2048) continue;
      const int grid = cus * per_cu;
      const double t0 = best_ms<false, 4>(d, n4, sink, grid, block, 10);
      const double t1 = best_ms<true, 4>(d, n4, sink, grid, block, 10);
      const double t2 = best_ms<true, 2>(d, n4, sink, grid, block, 10);
      const double t8 = best_ms<true, 8>(d, n4, sink, grid, block, 10);
      const double sz = (double)n4 * 16 / 1e12;
      std::printf("%s{\"block\": %d, \"blocks_per_cu\": %d, \"default_TBps\": %.3f, \"nt_TBps\": %.3f, "
                  "\"nt_2_in_flight\": %.3f, \"nt_8_in_flight\": %.3f}",
                  first ? "" : ", ", block, per_cu, sz / (t0 * 1e-3), sz / (t1 * 1e-3), sz / (t2 * 1e-3), sz / (t8 * 1e-3));
      first = false;
    }
  }
  std::printf("], \"columns\": [");
  {  // 3 x 16 B + 4 B per lane per unit = 52 bytes: n units so that the total is `bytes`
    const size_t n = (size_t)(bytes / 52.0);
    first = true;
    for (int block : {256, 512, 1024}) {
      for (int per_cu : {1, 2}) {
        const int grid = cus * per_cu;
        const double t1 = columns_ms<1>(d, n, sink, grid, block);
        const double t2 = columns_ms<2>(d, n, sink, grid, block);
        std::printf("%s{\"block\": %d, \"blocks_per_cu\": %d, \"one_unit_TBps\": %.3f, \"two_units_TBps\": %.3f}",
                    first ? "" : ", ", block, per_cu, 52.0 * n / (t1 * 1e-3) / 1e12, 52.0 * n / (t2 * 1e-3) / 1e12);
        first = false;
      }
    }
  }
  std::printf("]}\n");
  (void)hipFree(d);
  (void)hipFree(sink);
  return 0;
}
