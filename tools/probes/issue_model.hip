// Issue model of one wave per SIMD on gfx950 (what prefill_pw_kernel's tile loop is made of): shader cycles per
// instruction group for matrix instructions of both shapes alone, dependent at distance 1 / 2 / 8, and with vector /
// transcendental / LDS instructions placed in their shadow. One workgroup of four waves; s_memtime around 32 x 8 groups.
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/issue_model.hip -o tools/ab/issue_model     
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP8(x) x x x x x x x x
#define BODY_LOOP(BODY)                                                                  \
  {                                                                                      \
    unsigned long long t0, t1;                                                           \
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory"); \
    for (int i = 0; i < 32; ++i) asm volatile(REP8(BODY) ::: "memory", "v0", "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", \
      "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", \
      "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47"); \
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); \
    if (threadIdx.x == 0) out[k] = t1 - t0;                                              \
    ++k;                                                                                 \
  }

// registers: v0..v7 vector sources (zeros), v8..v15 results, v16..v47 scratch pairs; a[0:127] accumulators; a[128:143] operands
#define M16(acc) "v_mfma_f32_16x16x32_bf16 a[" acc "], a[128:131], a[132:135], a[" acc "]\n\t"
#define M32(acc) "v_mfma_f32_32x32x16_bf16 a[" acc "], a[128:131], a[132:135], a[" acc "]\n\t"
#define ADD(d) "v_add_f32 v" d ", v0, v1\n\t"
#define EXP(d) "v_exp_f32 v" d ", v2\n\t"
#define PKADD(d) "v_pk_add_f32 v[" d "], v[0:1], v[2:3]\n\t"
#define CVT(d) "v_cvt_pk_bf16_f32 v" d ", v0, v1\n\t"
#define LDS128 "ds_read_b128 v[44:47], v4\n\t"
#define LDSTR "ds_read_b64_tr_b16 v[44:45], v4\n\t"

__global__ __launch_bounds__(256) void probe(unsigned long long* out) {
  __shared__ char smem[16384];
  asm volatile("v_mov_b32 v0, 0\n\tv_mov_b32 v1, 0\n\tv_mov_b32 v2, 0\n\tv_mov_b32 v3, 0\n\tv_mov_b32 v4, 0" ::: "v0", "v1", "v2", "v3", "v4");
  if (threadIdx.x > 100000) smem[threadIdx.x] = 1;
  int k = 0;
  BODY_LOOP(ADD("8"))                                                       // 0: v_add_f32
  BODY_LOOP(EXP("8"))                                                       // 1: v_exp_f32
  BODY_LOOP(PKADD("8:9"))                                                   // 2: v_pk_add_f32
  BODY_LOOP(CVT("8"))                                                       // 3: v_cvt_pk_bf16_f32
  BODY_LOOP(M16("0:3") M16("4:7") M16("8:11") M16("12:15") M16("16:19") M16("20:23") M16("24:27") M16("28:31"))   // 4: 8 independent 16x16x32
  BODY_LOOP(M32("0:15") M32("16:31") M32("32:47") M32("48:63") M32("64:79") M32("80:95") M32("96:111") M32("112:127"))   // 5: 8 independent 32x32x16
  BODY_LOOP(M16("0:3") M16("4:7"))                                          // 6: 16x16x32 chains at distance 2 (x2 per group)
  BODY_LOOP(M16("0:3"))                                                     // 7: 16x16x32 chain at distance 1
  BODY_LOOP(M32("0:15"))                                                    // 8: 32x32x16 chain at distance 1
  BODY_LOOP(M16("0:3") EXP("8") M16("4:7") EXP("9"))                         // 9: (16x16x32 + 1 exp) x2
  BODY_LOOP(M16("0:3") EXP("8") EXP("9") M16("4:7") EXP("10") EXP("11"))     // 10: (16x16x32 + 2 exp) x2
  BODY_LOOP(M16("0:3") ADD("8") ADD("9") ADD("10") M16("4:7") ADD("11") ADD("12") ADD("13"))   // 11: (16x16x32 + 3 add) x2
  BODY_LOOP(M16("0:3") ADD("8") ADD("9") M16("4:7") ADD("11") ADD("12"))     // 12: (16x16x32 + 2 add) x2
  BODY_LOOP(M16("0:3") EXP("8") ADD("9") M16("4:7") EXP("11") ADD("12"))     // 13: (16x16x32 + exp + add) x2
  BODY_LOOP(M16("0:3") EXP("8") ADD("9") ADD("10") M16("4:7") EXP("11") ADD("12") ADD("13"))   // 14: (16x16x32 + exp + 2 add) x2
  BODY_LOOP(M32("0:15") EXP("8") EXP("9"))                                   // 15: 32x32x16 + 2 exp
  BODY_LOOP(M32("0:15") EXP("8") EXP("9") ADD("10") ADD("11") CVT("12"))      // 16: 32x32x16 + 2 exp + 2 add + cvt
  BODY_LOOP(M32("0:15") ADD("8") ADD("9") ADD("10") ADD("11") ADD("12") ADD("13") ADD("14"))   // 17: 32x32x16 + 7 add
  BODY_LOOP(M16("0:3") LDS128 M16("4:7") LDS128)                             // 18: (16x16x32 + ds_read_b128) x2
  BODY_LOOP(M16("0:3") LDSTR LDSTR M16("4:7") LDSTR LDSTR)                   // 19: (16x16x32 + 2 ds_read_b64_tr_b16) x2
  BODY_LOOP(M16("0:3") PKADD("8:9") CVT("10") M16("4:7") PKADD("12:13") CVT("14"))   // 20: (16x16x32 + pk_add + cvt) x2
  BODY_LOOP(M32("0:15") EXP("8") EXP("9") PKADD("10:11") CVT("12") LDSTR)      // 21: 32x32x16 + 2 exp + pk_add + cvt + tr read
  BODY_LOOP(EXP("8") ADD("9"))                                              // 22: exp + add alone
  BODY_LOOP(EXP("8") ADD("9") ADD("10") ADD("11"))                           // 23: exp + 3 add alone
  BODY_LOOP("s_nop 0\n\t")                                                  // 24: s_nop 0
  BODY_LOOP(M16("0:3") "s_nop 0\n\t" M16("4:7") "s_nop 0\n\t")               // 25: (16x16x32 + s_nop 0) x2
  // accumulators in VGPRs, A / B in accumulator registers (the S = K.Q^T chains of the kernel) and the other way round (P.V)
#define M16V(acc) "v_mfma_f32_16x16x32_bf16 v[" acc "], a[128:131], a[132:135], v[" acc "]\n\t"
#define M16P(acc) "v_mfma_f32_16x16x32_bf16 a[" acc "], v[24:27], v[28:31], a[" acc "]\n\t"
  BODY_LOOP(M16V("16:19") M16V("20:23"))                                     // 26: S-style, 2 chains
  BODY_LOOP(M16V("16:19") M16V("20:23") M16V("32:35") M16V("36:39"))          // 27: S-style, 4 chains (per 4)
  BODY_LOOP(M16P("0:3") M16P("4:7"))                                          // 28: P.V-style, 2 chains
  BODY_LOOP(M16V("16:19") EXP("8") M16V("20:23") EXP("9"))                     // 29: S-style 2 chains + exp each
  BODY_LOOP(M16P("0:3") EXP("8") M16P("4:7") "v_cvt_pk_bf16_f32 v9, v0, v1\n\t")   // 30: P.V-style 2 chains + exp / cvt
  BODY_LOOP(M16V("16:19") M16V("20:23") M16V("32:35") M16V("36:39") M16V("40:43") M16V("44:47") M16V("16:19") M16V("20:23"))   // 31: S-style, zero-free chain of 8 slots, 6 accumulators
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
  const char* names[] = {"v_add_f32", "v_exp_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32", "8 independent 16x16x32 (per 8)", "8 independent 32x32x16 (per 8)",
                         "2 chains 16x16x32 (per 2)", "16x16x32 back-to-back chain", "32x32x16 back-to-back chain", "(16x16x32 + exp) x2", "(16x16x32 + 2 exp) x2",
                         "(16x16x32 + 3 add) x2", "(16x16x32 + 2 add) x2", "(16x16x32 + exp + add) x2", "(16x16x32 + exp + 2 add) x2", "32x32x16 + 2 exp",
                         "32x32x16 + 2 exp + 2 add + cvt", "32x32x16 + 7 add", "(16x16x32 + ds_read_b128) x2", "(16x16x32 + 2 tr reads) x2",
                         "(16x16x32 + pk_add + cvt) x2", "32x32x16 + 2 exp + pk_add + cvt + tr read", "exp + add", "exp + 3 add", "s_nop 0", "(16x16x32 + s_nop 0) x2",
                         "S-style 16x16x32 (acc in VGPRs), 2 chains (per 2)", "S-style, 4 chains (per 4)", "P.V-style (A / B in VGPRs), 2 chains (per 2)",
                         "S-style 2 chains + exp each (per 2)", "P.V-style 2 chains + exp / cvt (per 2)", "S-style, 8 in a row over 6 accumulators (per 8)"};
  for (int i = 0; i < 32; ++i) printf("%2d %-46s %8.1f cycles per group\n", i, names[i], (double)h[i] / 256.0);
  return 0;
}
