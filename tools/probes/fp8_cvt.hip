// Probe: the hardware f32 -> fp8 converts of gfx950 (behind a clamp and a NaN select: f32x2_to_e4m3x2_sat /
// f32x2_to_e5m2x2_sat, csrc/common.h) against the software routines the cache-write kernels were pinned with
// (f32_to_e4m3_sat / f32_to_e5m2_sat, bit-exact against the oracle's reshape_and_cache_flash), on ALL 2^32 inputs.
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/fp8_cvt tools/probes/fp8_cvt.hip && /tmp/fp8_cvt
#include <cstdio>
#include "../../vllm-triton-backend_amd/csrc/common.h"

__global__ void compare(unsigned long long* mism, unsigned* first) {
  const unsigned long long base = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 2ull;
  for (unsigned long long i = base; i < (1ull << 32); i += (unsigned long long)gridDim.x * blockDim.x * 2ull) {
    const float a = mi355::bits_to_f32((uint32_t)i), b = mi355::bits_to_f32((uint32_t)i + 1u);
    const uint32_t h4 = mi355::f32x2_to_e4m3x2_sat(a, b), h5 = mi355::f32x2_to_e5m2x2_sat(a, b);
    const uint32_t s4 = mi355::f32_to_e4m3_sat(a) | ((uint32_t)mi355::f32_to_e4m3_sat(b) << 8);
    const uint32_t s5 = mi355::f32_to_e5m2_sat(a) | ((uint32_t)mi355::f32_to_e5m2_sat(b) << 8);
    if (h4 != s4) { if (atomicAdd(&mism[0], 1ull) < 8) { first[0] = (uint32_t)i; first[1] = h4; first[2] = s4; } }
    if (h5 != s5) { if (atomicAdd(&mism[1], 1ull) < 8) { first[3] = (uint32_t)i; first[4] = h5; first[5] = s5; } }
  }
}

int main() {
  unsigned long long* mism; unsigned* first;
  hipMalloc(&mism, 16); hipMalloc(&first, 24); hipMemset(mism, 0, 16); hipMemset(first, 0, 24);
  hipLaunchKernelGGL(compare, dim3(4096), dim3(256), 0, 0, mism, first);
  unsigned long long m[2]; unsigned f[6];
  hipMemcpy(m, mism, 16, hipMemcpyDeviceToHost); hipMemcpy(f, first, 24, hipMemcpyDeviceToHost);
  printf("2^32 inputs (pairs i, i+1): e4m3 mismatching pairs %llu, e5m2 mismatching pairs %llu\n", m[0], m[1]);
  if (m[0]) printf("  e4m3 example: bits 0x%08x hw 0x%04x sw 0x%04x\n", f[0], f[1], f[2]);
  if (m[1]) printf("  e5m2 example: bits 0x%08x hw 0x%04x sw 0x%04x\n", f[3], f[4], f[5]);
  return (m[0] || m[1]) ? 1 : 0;
}
