// how many wavefronts of a kernel with 256 VGPRs, L bytes of dynamic LDS (and, optionally, scratch) are resident per CU on this device:
// 4096 one-wavefront workgroups that spin for ~2 ms; the ones that started within 0.3 ms of the first are the resident set.
// hipcc -O3 --offload-arch=gfx950 -o tools/_build/resident_lab tools/resident_lab.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
template <int SCR>
__global__ void __launch_bounds__(64, 2) spin(unsigned long long* out, int ticks, int sel) {
  extern __shared__ double lds[];
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
  double acc = 0;
  if constexpr (SCR > 0) { volatile double priv[SCR > 0 ? SCR : 1]; for (int k = 0; k < SCR; ++k) priv[k] = k + threadIdx.x; acc = priv[sel & (SCR - 1)]; }
  lds[threadIdx.x] = acc;
  while (__builtin_amdgcn_s_memrealtime() - r0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0) {
    unsigned int hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[3 * blockIdx.x] = r0; out[3 * blockIdx.x + 1] = ((unsigned long long)xcc << 32) | hw; out[3 * blockIdx.x + 2] = (unsigned long long)lds[0];
  }
}
int main(int argc, char** argv) {
  const int G = 4096;
  unsigned long long* out; hipMalloc(&out, G * 24);
  std::vector<unsigned long long> h(3 * G);
  hipFuncSetAttribute((const void*)spin<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipFuncSetAttribute((const void*)spin<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int scr = 0; scr < 2; ++scr)
    for (int L : {1024, 16384, 18432, 19640, 20408, 20480, 22528, 34000}) {
      hipMemset(out, 0, G * 24);
      if (scr) hipLaunchKernelGGL(spin<128>, dim3(G), dim3(64), L, 0, out, 200000, 5); else hipLaunchKernelGGL(spin<0>, dim3(G), dim3(64), L, 0, out, 200000, 5);
      hipDeviceSynchronize();
      hipMemcpy(h.data(), out, G * 24, hipMemcpyDeviceToHost);
      unsigned long long t0 = ~0ull; for (int b = 0; b < G; ++b) if (h[3 * b] && h[3 * b] < t0) t0 = h[3 * b];
      std::map<unsigned long long, int> cu; int early = 0;
      for (int b = 0; b < G; ++b) if (h[3 * b] && h[3 * b] - t0 < 30000) { ++early; unsigned int hw = (unsigned)h[3 * b + 1]; cu[((h[3 * b + 1] >> 32) << 16) | (hw & 0xEF00u)]++; }
      std::map<int, int> hist; for (auto& kv : cu) hist[kv.second]++;
      printf("scratch %d, LDS %5d B: %4d wavefronts resident at once on %zu CUs; CUs by their count:", scr, L, early, cu.size());
      for (auto& kv : hist) printf(" %d x %d", kv.second, kv.first);
      printf("\n");
    }
  return 0;
}
