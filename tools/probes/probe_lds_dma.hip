// Probe: where does global_load_lds_dwordx4 land for M0 values beyond 64 KiB on gfx950 (160 KiB LDS)?
// hipcc --offload-arch=gfx950 -O2 -o probe_lds_dma probe_lds_dma.hip && ./probe_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(64) void probe(const uint32_t* src, uint32_t target, int lds_bytes, int* found) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t* w = (uint32_t*)smem;
  const int lane = threadIdx.x;
  for (int i = lane; i < lds_bytes / 4; i += 64) w[i] = 0;
  __syncthreads();
  const uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)smem;
  glds16(src + lane * 4, __builtin_amdgcn_readfirstlane(base + target));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int first = -1;
  for (int i = 0; i < lds_bytes / 4; ++i)
    if (w[i] == 0xabcd0000u) { first = i * 4; break; }
  if (lane == 0) { found[0] = first; found[1] = (int)base; }
}

int main() {
  const int lds_bytes = 160 * 1024;
  uint32_t* src; int* found;
  hipMalloc(&src, 1024); hipMalloc(&found, 8);
  std::vector<uint32_t> h(256);
  for (int i = 0; i < 256; ++i) h[i] = 0xabcd0000u + i;
  hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  const uint32_t targets[] = {0, 32768, 65520 - 1008, 65536, 65536 + 4096, 98304, 131072, 159 * 1024};
  for (uint32_t t : targets) {
    hipMemset(found, 0xff, 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), lds_bytes, 0, src, t, lds_bytes, found);
    int r[2];
    hipMemcpy(r, found, 8, hipMemcpyDeviceToHost);
    printf("target %7u -> data found at LDS byte %7d (lds base %d) %s\n", t, r[0], r[1], r[0] == (int)t ? "OK" : "MISMATCH");
  }
  return 0;
}
