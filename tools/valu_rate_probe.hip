// valu_rate_probe.hip -- issue cost, in cycles per wave64 instruction and SIMD, of the vector instructions the
// fill kernels' per-sample path is made of (v_add_f64, v_mul_f64, v_cmp_*_f64, v_cvt_i32_f64, v_cvt_f64_f32,
// v_mad_i32_i24, v_add_u32), measured on the box at hand.  Calibrates the vector-issue side of the lockstep /
// look-ahead passes in DESIGN.md section 4 (those passes are bound by it, not by HBM).  Not part of the product.
//   hipcc -O3 --offload-arch=gfx950 tools/valu_rate_probe.hip -o tools/valu_rate_probe && tools/valu_rate_probe
// Every wave runs ITER rounds of 8 independent instructions of one kind (inline asm, so nothing is folded or
// fused); WAVES waves per SIMD keep the pipeline full.  cycles = elapsed x clock / (ITER x 8 x WAVES).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define ITER 65536

#define PROBE(NAME, DECL, BODY, SINK)                                                        \
  __global__ __launch_bounds__(256) void NAME(double seed, unsigned* out) {                   \
    DECL;                                                                                     \
    _Pragma("unroll 4") for (int i = 0; i < ITER; i++) {                                       \
      BODY;                                                                                   \
    }                                                                                         \
    if (SINK) out[0] = 1u;                                                                    \
  }

// eight accumulators, one instruction each per round: independent chains
#define EIGHT(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double k = seed * 1.0000001
#define ADD64(n) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##n) : "v"(k));
#define MUL64(n) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##n) : "v"(k));
#define FMA64(n) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a##n) : "v"(k));
PROBE(add_f64, D8, EIGHT(ADD64), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678)
PROBE(mul_f64, D8, EIGHT(MUL64), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678)
PROBE(fma_f64, D8, EIGHT(FMA64), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678)

// compares into a scalar mask (the fill's domain tests): result in vcc, thrown away
#define CMP64(n) asm volatile("v_cmp_ge_f64 vcc, %0, %1" ::"v"(a##n), "v"(k) : "vcc");
PROBE(cmp_f64, D8, EIGHT(CMP64), a0 == 12345.678)

#define I8 int b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0
#define CVTI(n) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(b##n) : "v"(a##n));
PROBE(cvt_i32_f64, D8; I8, EIGHT(CVTI), b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 123456789)

#define F8 float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; double a0, a1, a2, a3, a4, a5, a6, a7
#define CVTD(n) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a##n) : "v"(f##n));
PROBE(cvt_f64_f32, F8, EIGHT(CVTD), a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678)

#define U8 int b0 = (int)seed, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3, b4 = b0 + 4, b5 = b0 + 5, b6 = b0 + 6, b7 = b0 + 7; int m = b0 | 3
#define MAD24(n) asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(b##n) : "v"(m));
#define ADDU(n) asm volatile("v_add_u32 %0, %0, %1" : "+v"(b##n) : "v"(m));
PROBE(mad_i32_i24, U8, EIGHT(MAD24), b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 123456789)
PROBE(add_u32, U8, EIGHT(ADDU), b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 123456789)

#define F8B float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; float kf = f0 * 1.0001f
#define MUL32(n) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f##n) : "v"(kf));
PROBE(mul_f32, F8B, EIGHT(MUL32), f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.678f)

typedef void (*kernel_t)(double, unsigned*);

static double run(kernel_t k, int grid, int block, unsigned* d_out) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a);
  (void)hipEventCreate(&b);
  double best = 1e30;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, 1.5, d_out);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipEventDestroy(a);
  (void)hipEventDestroy(b);
  return best;
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) {
    std::fprintf(stderr, "no device\n");
    return 1;
  }
  unsigned* d_out = nullptr;
  (void)hipMalloc((void**)&d_out, 64);
  const int cus = p.multiProcessorCount;
  const double mhz = p.clockRate / 1000.0;   // kHz -> MHz
  std::printf("%s: %d CUs, %.0f MHz (cycles below assume that clock throughout the launch)\n", p.gcnArchName, cus, mhz);
  struct {
    const char* name;
    kernel_t k;
  } probes[] = {{"v_mul_f32", mul_f32},         {"v_add_u32", add_u32},         {"v_mad_i32_i24", mad_i32_i24},
                {"v_add_f64", add_f64},         {"v_mul_f64", mul_f64},         {"v_fma_f64", fma_f64},
                {"v_cmp_ge_f64 (vcc)", cmp_f64}, {"v_cvt_i32_f64", cvt_i32_f64}, {"v_cvt_f64_f32", cvt_f64_f32}};
  for (int waves : {1, 2, 4}) {   // waves per SIMD: a workgroup of 256 = one wave on each of the CU's 4 SIMDs
    std::printf("-- %d wave(s) per SIMD\n", waves);
    for (auto& pr : probes) {
      const double ms = run(pr.k, cus * waves, 256, d_out);
      const double instr_per_simd = (double)ITER * 8.0 * waves;   // wave instructions each SIMD issued
      const double cycles = ms * 1e-3 * mhz * 1e6 / instr_per_simd;
      std::printf("%-20s %8.3f ms  %6.2f cycles per wave instruction\n", pr.name, ms, cycles);
    }
  }
  (void)hipFree(d_out);
  return 0;
}
