// lat_probe.hip - latencies of DEPENDENT operations on one wavefront of an MI355X, the numbers the interior point kernels are bound by:
// chains of v_fma_f64, v_rcp_f64 (+ Newton step), v_mfma_f64_16x16x4_f64 on one accumulator, LDS write -> barrier -> read round trips,
// ds_add_f64 on one address / on distinct addresses, and the unit of clock64() (s_memtime) against wall_clock64() (100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o tools/_build/lat_probe tools/lat_probe.hip && tools/_build/lat_probe
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double d4_t __attribute__((ext_vector_type(4)));
constexpr int REP = 4096;

__global__ void __launch_bounds__(64) probe(double* out, long long* t, double a0, double b0) {
  __shared__ double sh[2048];
  const int tid = threadIdx.x;
  for (int k = tid; k < 2048; k += 64) sh[k] = 1.0 + 1e-9 * k;
  __syncthreads();
  double x = a0 + 1e-9 * tid, y = b0;
  long long c0, c1, w0, w1;
  // 0: dependent fma chain
  w0 = wall_clock64(); c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) x = fma(x, y, 1e-9);
  c1 = clock64(); w1 = wall_clock64();
  if (tid == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
  // 1: dependent rcp chain (raw v_rcp_f64)
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) x = __builtin_amdgcn_rcp(x + 1.5);
  c1 = clock64();
  if (tid == 0) t[2] = c1 - c0;
  // 2: rcp + one Newton step
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) { double z = x + 1.5; double r = __builtin_amdgcn_rcp(z); x = fma(r, fma(-z, r, 1.0), r); }
  c1 = clock64();
  if (tid == 0) t[3] = c1 - c0;
  // 3: dependent MFMA chain on one accumulator
  d4_t acc = {x, x, x, x};
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-3, acc, 0, 0, 0);
  c1 = clock64();
  if (tid == 0) t[4] = c1 - c0;
  x += acc[0] + acc[1] + acc[2] + acc[3];
  // 4: LDS round trip: write own slot, read the neighbour's (dependent through the value)
  c0 = clock64();
  for (int i = 0; i < REP; ++i) { sh[tid] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x = sh[(tid + 1) & 63] * 0.999; }
  c1 = clock64();
  if (tid == 0) t[5] = c1 - c0;
  // 5: dependent LDS read chain (pointer chase through values)
  int idx = tid;
  c0 = clock64();
  for (int i = 0; i < REP; ++i) { double v = sh[idx]; idx = ((int)(v * 1e-300) + idx + 1) & 2047; }
  c1 = clock64();
  if (tid == 0) t[6] = c1 - c0;
  x += idx;
  // 6: ds_add_f64, 64 lanes on distinct addresses, back to back (throughput)
  __syncthreads();
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) atomicAdd(&sh[tid + 64 * (i & 15)], 1e-9);
  __syncthreads();
  c1 = clock64();
  if (tid == 0) t[7] = c1 - c0;
  // 7: ds_add_f64, all 64 lanes on ONE address
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) atomicAdd(&sh[i & 15], 1e-9);
  __syncthreads();
  c1 = clock64();
  if (tid == 0) t[8] = c1 - c0;
  // 8: ds_add_f64, 8 lanes per address (the pattern of rows sharing a column)
  c0 = clock64();
#pragma unroll 16
  for (int i = 0; i < REP; ++i) atomicAdd(&sh[(tid >> 3) + 8 * (i & 15)], 1e-9);
  __syncthreads();
  c1 = clock64();
  if (tid == 0) t[9] = c1 - c0;
  // 9: independent fma (4 chains): issue rate
  double x1 = x + 1, x2 = x + 2, x3 = x + 3;
  c0 = clock64();
#pragma unroll 8
  for (int i = 0; i < REP; ++i) { x = fma(x, y, 1e-9); x1 = fma(x1, y, 1e-9); x2 = fma(x2, y, 1e-9); x3 = fma(x3, y, 1e-9); }
  c1 = clock64();
  if (tid == 0) t[10] = c1 - c0;
  x += x1 + x2 + x3 + sh[tid];
  if (x == 12345.678) out[tid] = x;
}

int main() {
  double* out; long long* t; long long h[16];
  hipMalloc(&out, 64 * 8); hipMalloc(&t, 16 * 8); hipMemset(t, 0, 16 * 8);
  for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, t, 1.0000001, 0.9999999); hipDeviceSynchronize(); }
  hipMemcpy(h, t, 16 * 8, hipMemcpyDeviceToHost);
  const double unit_ns = (double)h[1] * 10.0 / (double)h[0];   // wall_clock64 ticks at 100 MHz
  std::printf("clock64 unit: %.3f ns (%.0f MHz)\n", unit_ns, 1e3 / unit_ns);
  const char* nm[] = {"dependent v_fma_f64", "", "dependent v_rcp_f64 (+ add)", "rcp + Newton step (+ add)", "dependent v_mfma_f64_16x16x4_f64", "LDS write -> wave barrier -> read", "dependent ds_read_b64",
                      "ds_add_f64 distinct addresses (per instruction)", "ds_add_f64 one address (per instruction)", "ds_add_f64 8 lanes per address (per instruction)", "4 independent v_fma_f64 (per group of 4)"};
  for (int k : {0, 2, 3, 4, 5, 6, 7, 8, 9, 10}) std::printf("%-52s %8.1f clock64 units = %7.1f ns per step\n", nm[k], (double)h[k] / REP, (double)h[k] / REP * unit_ns);
  return 0;
}
