// how long handing out N work items through ONE atomic counter takes on this device: G workgroups of one wavefront, lane 0 takes an item, the
// wavefront "skips" it (nothing else) - the scan the larger launches of a round make over the nodes that are not theirs.  hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ void __launch_bounds__(64) scan(int* ctr, int n, const unsigned char* cls, int want, unsigned long long* sink) {
  __shared__ int sh;
  unsigned long long acc = 0;
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) sh = atomicAdd(ctr, 1);
    __syncthreads();
    const int k = __builtin_amdgcn_readfirstlane(sh);
    if (k >= n) break;
    if (cls[k] != want) continue;
    acc += k;
  }
  if (threadIdx.x == 0 && acc) atomicAdd(sink, acc);
}
// statistics counters: lane 0 of every wavefront adds to ten adjacent 64-bit counters K times (no value returned)
__global__ void __launch_bounds__(64) stats(unsigned long long* c, int K, int spread) {
  if (threadIdx.x != 0) return;
  for (int k = 0; k < K; ++k)
    for (int q = 0; q < 10; ++q) atomicAdd(&c[(size_t)q * spread], 1ull);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 98304;
  int* ctr; unsigned char* cls; unsigned long long* sink;
  hipMalloc(&ctr, 64 * 4); hipMalloc(&cls, n); hipMalloc(&sink, 8); hipMemset(cls, 0, n); hipMemset(sink, 0, 8);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int G : {64, 256, 1024, 2048, 4096}) {
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
      hipMemset(ctr, 0, 64 * 4); hipDeviceSynchronize();
      hipEventRecord(a); hipLaunchKernelGGL(scan, dim3(G), dim3(64), 0, 0, ctr, n, cls, 1, sink); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    printf("one counter, %d workgroups, %d items: %.3f ms = %.1f ns per hand-out\n", G, n, best, 1e6 * best / n);
  }
  // three kernels on three streams, each with its own counter (16 B apart / 256 B apart)
  hipStream_t s[3]; for (auto& q : s) hipStreamCreate(&q);
  for (int stride : {1, 16}) {
    hipMemset(ctr, 0, 64 * 4); hipDeviceSynchronize();
    hipEventRecord(a, 0);
    hipEvent_t f; hipEventCreate(&f); hipEventRecord(f, 0);
    hipEvent_t j[3];
    for (int q = 0; q < 3; ++q) { hipStreamWaitEvent(s[q], f, 0); hipLaunchKernelGGL(scan, dim3(1024), dim3(64), 0, s[q], ctr + q * stride, n, cls, 1, sink); hipEventCreate(&j[q]); hipEventRecord(j[q], s[q]); hipStreamWaitEvent(0, j[q], 0); }
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("three concurrent scans of %d items, 1024 workgroups each, counters %d B apart: %.3f ms\n", n, stride * 4, ms);
  }
  unsigned long long* sc; hipMalloc(&sc, 10 * 4096 * 8); hipMemset(sc, 0, 10 * 4096 * 8);
  for (int spread : {1, 8, 512, 4096}) {
    hipDeviceSynchronize();
    hipEventRecord(a); hipLaunchKernelGGL(stats, dim3(2048), dim3(64), 0, 0, sc, 48, spread); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("2048 wavefronts x 48 x 10 adds without a return to ten counters %d B apart: %.3f ms = %.2f ns per add\n", spread * 8, ms, 1e6 * ms / (2048.0 * 480));
  }
  return 0;
}
