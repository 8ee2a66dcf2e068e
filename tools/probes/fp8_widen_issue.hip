// What widening an fp8 tile costs one wave per SIMD (prefill_pw_kernel's KV8 instantiations): shader cycles per group for the
// fp8 -> 16-bit conversions of gfx950 and a 16-byte LDS write, alone and in the shadow of a 16x16x32 matrix instruction.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/fp8_widen_issue.hip -o tools/ab/fp8_widen_issue
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define BODY_LOOP(BODY)                                                                  \
  {                                                                                      \
    unsigned long long t0, t1;                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int i = 0; i < 32; ++i) asm volatile(REP8(BODY) ::: "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", \
      "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
      "v28", "v29", "v30", "v31"); \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if (threadIdx.x == 0) out[k] = t1 - t0;                                              \
    ++k;                                                                                 \
  }
#define M16(acc) "v_mfma_f32_16x16x32_bf16 a[" acc "], a[128:131], a[132:135], a[" acc "]\n\t"
#define ADD(d) "v_add_f32 v" d ", v0, v1\n\t"
#define SCVT(d) "v_cvt_scalef32_pk_bf16_fp8 v" d ", v0, 1.0\n\t"
#define SCVTH(d) "v_cvt_scalef32_pk_bf16_fp8 v" d ", v0, 1.0 op_sel:[1,0,0]\n\t"
#define SCVT32(d) "v_cvt_scalef32_pk_f32_fp8 v[" d "], v0, 1.0\n\t"
#define CVT32(d) "v_cvt_pk_f32_fp8 v[" d "], v0\n\t"
#define CVTB(d) "v_cvt_pk_bf16_f32 v" d ", v0, v1\n\t"
#define PERM(d) "v_perm_b32 v" d ", v0, v1, v2\n\t"
#define WR128 "ds_write_b128 v4, v[8:11]\n\t"

__global__ __launch_bounds__(256) void probe(unsigned long long* out) {
  __shared__ char smem[16384];
  asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0" ::: "v0", "v1", "v2", "v3");
  asm volatile("v_lshlrev_b32 v4, 4, %0" :: "v"(threadIdx.x & 63) : "v4");
  if (threadIdx.x > 100000) smem[threadIdx.x] = 1;
  int k = 0;
  BODY_LOOP(ADD("8"))                                        // 0
  BODY_LOOP(SCVT("8"))                                       // 1
  BODY_LOOP(SCVT("8") SCVTH("9"))                            // 2
  BODY_LOOP(SCVT32("8:9"))                                   // 3
  BODY_LOOP(CVT32("8:9"))                                    // 4
  BODY_LOOP(CVT32("8:9") CVTB("10"))                         // 5
  BODY_LOOP(PERM("8"))                                       // 6
  BODY_LOOP(WR128)                                           // 7
  BODY_LOOP(M16("0:3") M16("4:7"))                           // 8
  BODY_LOOP(M16("0:3") SCVT("8") M16("4:7") SCVT("9"))       // 9
  BODY_LOOP(M16("0:3") SCVT("8") SCVTH("10") M16("4:7") SCVT("9") SCVTH("11"))   // 10
  BODY_LOOP(M16("0:3") ADD("8") ADD("10") M16("4:7") ADD("9") ADD("11"))         // 11
  BODY_LOOP(M16("0:3") WR128 M16("4:7") WR128)               // 12
  BODY_LOOP(M16("0:3") CVT32("8:9") CVTB("10") M16("4:7") CVT32("12:13") CVTB("14"))   // 13
  BODY_LOOP(M16("0:3") PERM("8") PERM("10") M16("4:7") PERM("9") PERM("11"))     // 14
  BODY_LOOP(M16("0:3") "v_exp_f32 v8, v2\n\t" SCVT("9") M16("4:7") "v_exp_f32 v10, v2\n\t" SCVT("11"))   // 15
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

int main() {
  unsigned long long* d;
  if (hipMalloc(&d, 64 * 8) != hipSuccess) return 1;
  unsigned long long h[64];
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), 0, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[] = {"v_add_f32", "v_cvt_scalef32_pk_bf16_fp8", "the pair (lo, hi)", "v_cvt_scalef32_pk_f32_fp8", "v_cvt_pk_f32_fp8", "v_cvt_pk_f32_fp8 + v_cvt_pk_bf16_f32",
                         "v_perm_b32", "ds_write_b128", "2 x 16x16x32 (per 2)", "(16x16x32 + 1 scalef32 cvt) x2", "(16x16x32 + the pair) x2", "(16x16x32 + 2 add) x2",
                         "(16x16x32 + ds_write_b128) x2", "(16x16x32 + cvt_pk_f32_fp8 + cvt_pk_bf16) x2", "(16x16x32 + 2 perm) x2", "(16x16x32 + exp + scalef32 cvt) x2"};
  for (int i = 0; i < 16; ++i) printf("%2d %-46s %8.1f cycles per group\n", i, names[i], (double)h[i] / 256.0);
  return 0;
}
