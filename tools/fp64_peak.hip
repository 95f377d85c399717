// fp64_peak.hip - measured FP64 peaks of one MI355X: back-to-back v_mfma_f64_16x16x4_f64 on
// independent accumulators and back-to-back v_fma_f64, W wavefronts per SIMD on every CU.
//   hipcc --offload-arch=gfx950 -O3 -o fp64_peak tools/fp64_peak.hip && ./fp64_peak > profiles/r02_fp64_peak.json
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(256) mfma_loop(double* out, int iters, double a0, double b0) {
  d4_t acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = d4_t{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;   // keeps the chain alive, never true
}

__global__ void __launch_bounds__(256) fma_loop(double* out, int iters, double a0, double b0) {
  double x[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = a0 * (k + 1) + threadIdx.x * 1e-9;
  const double m = 1.0 - b0 * 1e-9, c = b0 * 1e-12;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = __builtin_fma(x[k], m, c);
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += x[k];
  if (s == 12345.678) out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class F> static double time_ms(F&& launch, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch(); (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r) launch();
  (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

int main() {
  hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, 0) != hipSuccess) { std::fprintf(stderr, "no HIP device\n"); return 1; }
  const int cus = pr.multiProcessorCount;
  double* out; (void)hipMalloc(&out, (size_t)cus * 8 * 256 * 8);
  const int iters = 4000;
  std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"runs\": [\n", pr.gcnArchName, cus, pr.clockRate / 1000);
  bool first = true;
  for (int wps = 1; wps <= 2; ++wps) {   // wavefronts per SIMD: one 256-thread block per CU and SIMD-wave
    const int blocks = cus * wps;
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
      double n = (double)blocks * 4 * iters * 8 * 4;   // wave-level MFMAs
      double tf = n * 2048.0 / (ms * 1e-3) / 1e12;
      double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * 4 * wps);
      std::printf("%s {\"kernel\": \"v_mfma_f64_16x16x4_f64 x4 accumulators\", \"waves_per_simd\": %d, \"ms\": %.4f, \"tflops\": %.2f, \"cycles_per_mfma_per_simd_at_2.4GHz\": %.1f}", first ? "" : ",\n", wps, ms, tf, cyc);
      first = false;
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
      double n = (double)blocks * 4 * iters * 8;
      double tf = n * 2048.0 / (ms * 1e-3) / 1e12;
      double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wps);
      std::printf(",\n {\"kernel\": \"v_mfma_f64_16x16x4_f64 dependent chain (1 accumulator)\", \"waves_per_simd\": %d, \"ms\": %.4f, \"tflops\": %.2f, \"cycles_per_mfma_per_simd_at_2.4GHz\": %.1f}", wps, ms, tf, cyc);
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 0.5); }, 5);
      double n = (double)blocks * 4 * iters * 64;   // wave-level v_fma_f64
      double tf = n * 128.0 / (ms * 1e-3) / 1e12;
      std::printf(",\n {\"kernel\": \"v_fma_f64 x8 independent\", \"waves_per_simd\": %d, \"ms\": %.4f, \"tflops\": %.2f}", wps, ms, tf);
    }
  }
  std::printf("\n]}\n");
  (void)hipFree(out);
  return 0;
}
