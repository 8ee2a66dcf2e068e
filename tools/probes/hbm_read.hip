// HBM read reference point for the decode roofline: the fastest plain streaming read of a 2 GiB buffer this chip gives a
// hand-written kernel (16-byte loads, no dependent addressing, no compute), next to which the paged decode kernel's
// GB/s is read. Build: hipcc -O3 --offload-arch=gfx950 tools/probes/hbm_read.hip -o tools/ab/hbm_read
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
  // each workgroup streams one contiguous chunk; a wave's loads cover 1 KiB per instruction
  const size_t per_wg = n16 / gridDim.x;
  const u32x4* p = src + per_wg * blockIdx.x;
  unsigned acc = 0;
  for (size_t i = threadIdx.x; i + (UNROLL - 1) * 256 < per_wg; i += UNROLL * 256) {
    u32x4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = NT ? __builtin_nontemporal_load(p + i + u * 256) : p[i + u * 256];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (acc == 0x12345678u) *sink = acc;   // keeps the loads alive, practically never true
}

// the same stream with an explicit cache policy on the load instruction (sc0 / sc1 / nt bits)
#define POLICY_KERNEL(NAME, BITS)                                                                              \
  __global__ __launch_bounds__(256) void NAME(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {   \
    const size_t per_wg = n16 / gridDim.x;                                                                     \
    const u32x4* p = src + per_wg * blockIdx.x;                                                                \
    unsigned acc = 0;                                                                                          \
    for (size_t i = threadIdx.x; i + 7 * 256 < per_wg; i += 8 * 256) {                                         \
      u32x4 v[8];                                                                                              \
      _Pragma("unroll") for (int u = 0; u < 8; ++u)                                                            \
        asm volatile("global_load_dwordx4 %0, %1, off " BITS : "=v"(v[u]) : "v"(p + i + u * 256));            \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
      _Pragma("unroll") for (int u = 0; u < 8; ++u) { asm volatile("" : "+v"(v[u])); acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w; } \
    }                                                                                                          \
    if (acc == 0x12345678u) *sink = acc;                                                                       \
  }
POLICY_KERNEL(read_plain, "")
POLICY_KERNEL(read_nt, "nt")
POLICY_KERNEL(read_sc0, "sc0")
POLICY_KERNEL(read_sc1, "sc1")
POLICY_KERNEL(read_sc0_sc1, "sc0 sc1")
POLICY_KERNEL(read_sc1_nt, "sc1 nt")
POLICY_KERNEL(read_sc0_nt, "sc0 nt")
POLICY_KERNEL(read_sc0_sc1_nt, "sc0 sc1 nt")

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int UNROLL, bool NT>
static int run(const u32x4* buf, size_t n16, unsigned* sink, int wgs, double& best) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(wgs), dim3(256), 0, 0, buf, n16, sink);
  CHECK(hipDeviceSynchronize());
  const int reps = 20;
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((read_kernel<UNROLL, NT>), dim3(wgs), dim3(256), 0, 0, buf, n16, sink);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, gbs = (double)n16 * 16 / (us * 1e-6) / 1e9;
  printf("  wgs %6d unroll %d %s: %8.1f us  %7.1f GB/s\n", wgs, UNROLL, NT ? "nt" : "  ", us, gbs);
  if (gbs > best) best = gbs;
  return 0;
}

int main() {
  const size_t bytes = (size_t)2 << 30;   // the C3 decode job reads 2.15 GB
  u32x4* buf; unsigned* sink;
  CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4));
  CHECK(hipMemset(buf, 1, bytes));
  CHECK(hipDeviceSynchronize());
  const size_t n16 = bytes / 16;
  double best = 0;
  for (int wgs : {2048, 4096, 8192, 16384, 65536}) {
    if (run<4, false>(buf, n16, sink, wgs, best)) return 1;
    if (run<8, false>(buf, n16, sink, wgs, best)) return 1;
    if (run<8, true>(buf, n16, sink, wgs, best)) return 1;
  }
  {
    typedef void (*kern_t)(const u32x4*, size_t, unsigned*);
    const struct { kern_t k; const char* name; } pol[] = {{read_plain, "(none)"}, {read_nt, "nt"}, {read_sc0, "sc0"}, {read_sc1, "sc1"},
      {read_sc0_sc1, "sc0 sc1"}, {read_sc1_nt, "sc1 nt"}, {read_sc0_nt, "sc0 nt"}, {read_sc0_sc1_nt, "sc0 sc1 nt"}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (const auto& pk : pol) {
      for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(pk.k, dim3(8192), dim3(256), 0, 0, buf, n16, sink);
      CHECK(hipEventRecord(e0));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(pk.k, dim3(8192), dim3(256), 0, 0, buf, n16, sink);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("  policy %-12s: %8.1f us  %7.1f GB/s\n", pk.name, ms * 1e3 / 20, (double)bytes / (ms * 1e-3 / 20) / 1e9);
    }
  }
  printf("best streaming read of 2 GiB: %.1f GB/s = %.3f of 8 TB/s\n", best, best / 8000.0);
  return 0;
}
