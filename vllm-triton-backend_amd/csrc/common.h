// Shared device/host helpers for libmi355_attn (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/mi355_attn.h"

// ONE switch for everything that exists for measurements only (round 4, VERDICT r03 hygiene):
//   * compile time - the ablation / stamp / profile builds (-DMI355_PROFILE_WG, -DMI355_PW_STAMP, -DPW_ABL_*, -DMI355_ABLATE_*,
//     -DMI355_DECODE_PF=.., ...) need -DMI355_LAB beside them (tools/build_variant.sh and the profile tools pass it): a
//     product build that carries one of them by accident does not compile;
//   * run time - the environment switches of DESIGN.md section 5 (MI355_PREFILL, MI355_DECODE_TREE, MI355_PW_SLOTS, ...) are
//     read through lab_env(), which sees them only when MI355_LAB=1 is set as well: a production process has no hidden knobs.
#if !defined(MI355_LAB) && (defined(MI355_PROFILE_WG) || defined(MI355_PROFILE_PHASES) || defined(MI355_PW_STAMP) || defined(MI355_PW_SEAM) || \
                            defined(PW_ABL_DMA) || defined(PW_FORCE_FALLBACK) || defined(PW_DMA_SPREAD) || defined(MI355_ABLATE_QK) || defined(MI355_ABLATE_DMA) || \
                            defined(MI355_ABLATE_SOFTMAX) || defined(MI355_ABLATE_PV) || defined(MI355_ABLATE_BARRIER) || defined(MI355_PACKED_ROWSUM) || \
                            defined(MI355_DECODE_PF) || defined(MI355_DECODE_PLAIN_LOADS))
#error "diagnostic / ablation macros are lab builds: add -DMI355_LAB"
#endif
#include <cstdlib>

namespace mi355 {

// getenv for the measurement switches: nullptr unless the process also carries MI355_LAB=1 (read once).
inline const char* lab_env(const char* name) {
  static const bool lab = [] { const char* e = std::getenv("MI355_LAB"); return e && e[0] && e[0] != '0'; }();
  if (!lab) return nullptr;
  const char* e = std::getenv(name);
  return (e && e[0]) ? e : nullptr;          // (a variable set to the empty string is an unset one: `VAR=$x cmd` with an empty x pins nothing)
}

// ---------------------------------------------------------------------------------------------
// element-type tags. Storage is raw bits; arithmetic is always fp32.
// ---------------------------------------------------------------------------------------------
struct f32_t { using storage = float; };
struct f16_t { using storage = uint16_t; };
struct bf16_t { using storage = uint16_t; };
struct e4m3_t { using storage = uint8_t; };  // OCP e4m3fn
struct e5m2_t { using storage = uint8_t; };  // OCP e5m2

__device__ __forceinline__ float bits_to_f32(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f32_to_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// bf16 <-> f32 (round to nearest even, NaN preserved by the hardware convert)
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return bits_to_f32(uint32_t(h) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32 on gfx950
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float f16_to_f32(uint16_t h) {
  return (float)__builtin_bit_cast(_Float16, h);
}
__device__ __forceinline__ uint16_t f32_to_f16(float f) {
  _Float16 h = (_Float16)f;
  return __builtin_bit_cast(uint16_t, h);
}

// OCP fp8 decode in software (exact; used by the generic kernel and to validate the hardware
// converts used by the fast kernels).
__device__ __forceinline__ float e4m3_to_f32(uint8_t v) {
  const uint32_t sign = (uint32_t)(v & 0x80) << 24;
  const uint32_t e = (v >> 3) & 0xF, m = v & 0x7;
  float mag;
  if (e == 0) {
    mag = (float)m * 0.001953125f;  // m * 2^-9 (subnormal)
  } else if (e == 15 && m == 7) {
    mag = __builtin_nanf("");
  } else {
    mag = bits_to_f32(((e + 120u) << 23) | (m << 20));  // bias 7 -> 127
  }
  return bits_to_f32(f32_to_bits(mag) | sign);
}
__device__ __forceinline__ float e5m2_to_f32(uint8_t v) {
  // e5m2 is the top byte of an IEEE half
  return f16_to_f32((uint16_t)v << 8);
}

// saturating (finite) f32 -> fp8, round to nearest even; matches __HIP_SATFINITE semantics. In software: the statement
// of the encoding (the kernels use the hardware form below, which tools/probes/fp8_cvt.hip checks against these).
__device__ __forceinline__ uint8_t f32_to_e4m3_sat(float f) {
  if (f != f) return 0x7F;
  const uint32_t sign = (f32_to_bits(f) >> 24) & 0x80;
  float a = __builtin_fabsf(f);
  if (a == 0.0f) return (uint8_t)sign;          // frexpf(0) has exponent 0: the grid search below is for a > 0 only
  if (a >= 448.0f) return (uint8_t)(sign | 0x7E);
  // scale so that the e4m3 subnormal/normal grid maps onto integers, round with rintf
  // normal: value = (8+m) * 2^(e-10); find e from the float exponent
  int ex;
  (void)__builtin_frexpf(a, &ex);  // a = fr * 2^ex, fr in [0.5,1)  => a in [2^(ex-1), 2^ex)
  int e = ex - 1 + 7;              // biased exponent candidate
  if (e < 1) e = 1;                // subnormal range shares the e==1 step size 2^-9
  const float step = __builtin_ldexpf(1.0f, e - 10);  // spacing of representable values
  float q = __builtin_rintf(a / step);                // in [0,16]
  uint32_t qi = (uint32_t)q;
  uint32_t out;
  if (e == 1 && qi < 8) {
    out = qi;  // subnormal (exp field 0)
  } else {
    if (qi == 16) { qi = 8; e += 1; }
    out = ((uint32_t)e << 3) | (qi - 8);
  }
  if (out > 0x7E) out = 0x7E;
  return (uint8_t)(sign | out);
}
__device__ __forceinline__ uint8_t f32_to_e5m2_sat(float f) {
  if (f != f) return 0x7F;
  const uint32_t sign = (f32_to_bits(f) >> 24) & 0x80;
  float a = __builtin_fabsf(f);
  if (a == 0.0f) return (uint8_t)sign;
  if (a >= 57344.0f) return (uint8_t)(sign | 0x7B);
  int ex;
  (void)__builtin_frexpf(a, &ex);
  int e = ex - 1 + 15;
  if (e < 1) e = 1;
  const float step = __builtin_ldexpf(1.0f, e - 17);  // (4+m) * 2^(e-17)
  float q = __builtin_rintf(a / step);                // in [0,8]
  uint32_t qi = (uint32_t)q;
  uint32_t out;
  if (e == 1 && qi < 4) {
    out = qi;
  } else {
    if (qi == 8) { qi = 4; e += 1; }
    out = ((uint32_t)e << 2) | (qi - 4);
  }
  if (out > 0x7B) out = 0x7B;
  return (uint8_t)(sign | out);
}

// The same two conversions on the hardware convert (v_cvt_pk_fp8_f32 / v_cvt_pk_bf8_f32: OCP formats on gfx950, round
// to nearest even, subnormals included): the magnitude clamped to the format's largest finite value first (SATFINITE),
// NaN -> 0x7F as above. Two values per call, the low and high byte of the result's low half. Bit-identical to
// f32_to_e4m3_sat / f32_to_e5m2_sat on every one of the 2^32 inputs (tools/probes/fp8_cvt.hip, profiles/r03/fp8_cvt.log);
// ~9 instructions per pair instead of ~130.
__device__ __forceinline__ uint32_t f32x2_to_e4m3x2_sat(float a, float b) {
  const float ca = __builtin_fminf(__builtin_fmaxf(a, -448.0f), 448.0f), cb = __builtin_fminf(__builtin_fmaxf(b, -448.0f), 448.0f);
  const uint32_t pk = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(ca, cb, 0, false) & 0xffffu;
  const uint32_t qa = (a != a) ? 0x7Fu : (pk & 0xffu), qb = (b != b) ? 0x7Fu : (pk >> 8);
  return qa | (qb << 8);
}
__device__ __forceinline__ uint32_t f32x2_to_e5m2x2_sat(float a, float b) {
  const float ca = __builtin_fminf(__builtin_fmaxf(a, -57344.0f), 57344.0f), cb = __builtin_fminf(__builtin_fmaxf(b, -57344.0f), 57344.0f);
  const uint32_t pk = (uint32_t)__builtin_amdgcn_cvt_pk_bf8_f32(ca, cb, 0, false) & 0xffffu;
  const uint32_t qa = (a != a) ? 0x7Fu : (pk & 0xffu), qb = (b != b) ? 0x7Fu : (pk >> 8);
  return qa | (qb << 8);
}

typedef __attribute__((ext_vector_type(4))) unsigned int cw_u32x4_t;
// 16 values of a 16-bit type (two 16-byte pieces, in memory order) -> 16 fp8 bytes: the quantising store of
// reshape_and_cache_flash, sat_fp8(x / scale) - a true division like the reference's store (scripts/vllm_utils.py:377-401;
// x * (1 / scale) rounds differently for scales that are no powers of two), skipped for the scale 1.0 of an uncalibrated cache.
template <typename T, typename KVT>
__device__ __forceinline__ cw_u32x4_t quantise_fp8x16(cw_u32x4_t lo, cw_u32x4_t hi, float scale) {
  const bool unit = __builtin_amdgcn_readfirstlane(f32_to_bits(scale)) == 0x3f800000u;
  cw_u32x4_t out;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const uint32_t s0 = w < 2 ? lo[2 * w] : hi[2 * w - 4], s1 = w < 2 ? lo[2 * w + 1] : hi[2 * w - 3];
    float x0, x1, x2, x3;
    if constexpr (__is_same(T, bf16_t)) {
      x0 = bits_to_f32(s0 << 16); x1 = bits_to_f32(s0 & 0xffff0000u); x2 = bits_to_f32(s1 << 16); x3 = bits_to_f32(s1 & 0xffff0000u);
    } else {
      x0 = f16_to_f32((uint16_t)s0); x1 = f16_to_f32((uint16_t)(s0 >> 16)); x2 = f16_to_f32((uint16_t)s1); x3 = f16_to_f32((uint16_t)(s1 >> 16));
    }
    if (!unit) { x0 /= scale; x1 /= scale; x2 /= scale; x3 /= scale; }
    if constexpr (__is_same(KVT, e4m3_t)) out[w] = f32x2_to_e4m3x2_sat(x0, x1) | (f32x2_to_e4m3x2_sat(x2, x3) << 16);
    else out[w] = f32x2_to_e5m2x2_sat(x0, x1) | (f32x2_to_e5m2x2_sat(x2, x3) << 16);
  }
  return out;
}

// two f32 -> one dword of two 16-bit floats with ONE v_cvt_pk_* instruction (the scalar casts in
// f32_to_bf16/f32_to_f16 cost a convert per element plus shift/or to pack)
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
}
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, h2));
}

template <typename T> struct elem;
template <> struct elem<f32_t> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return ((const float*)p)[i]; }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((float*)p)[i] = v; }
  static __device__ __forceinline__ float round(float v) { return v; }
};
template <> struct elem<f16_t> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return f16_to_f32(((const uint16_t*)p)[i]); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_f16(v); }
  static __device__ __forceinline__ float round(float v) { return f16_to_f32(f32_to_f16(v)); }
};
template <> struct elem<bf16_t> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return bf16_to_f32(((const uint16_t*)p)[i]); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint16_t*)p)[i] = f32_to_bf16(v); }
  static __device__ __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};
template <> struct elem<e4m3_t> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return e4m3_to_f32(((const uint8_t*)p)[i]); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint8_t*)p)[i] = (uint8_t)f32x2_to_e4m3x2_sat(v, v); }
};
template <> struct elem<e5m2_t> {
  static __device__ __forceinline__ float load(const void* p, int64_t i) { return e5m2_to_f32(((const uint8_t*)p)[i]); }
  static __device__ __forceinline__ void store(void* p, int64_t i, float v) { ((uint8_t*)p)[i] = (uint8_t)f32x2_to_e5m2x2_sat(v, v); }
};

// ---------------------------------------------------------------------------------------------
// sequence lookup: largest i with cu[i] <= token  (reference: find_seq_idx, token mode,
// LIB/kernels/triton_unified_attention.py:32-52)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int find_seq_by_token(const int32_t* __restrict__ cu, int num_seqs, int token) {
  int left = 0, right = num_seqs;
  while (left < right) {
    const int mid = (left + right) >> 1;
    if (cu[mid] <= token) left = mid + 1; else right = mid;
  }
  return left - 1;
}

// The Q-block form of the search (largest i with cu[i] / block_q + i <= qblock, find_seq_idx :32-52) together with the
// sequence's three words - query start, query length, key length - in ONE memory round trip for batches of up to 63
// sequences: lane i takes cu_seqlens_q[i] and seqused_k[i], the search is a ballot (the left side is non-decreasing in i),
// the words come out with v_readlane. The binary search is ceil(log2 S) + 1 DEPENDENT trips, each ~1-2 us on a loaded chip
// and with nothing of the workgroup in flight meanwhile: at 8 sequences it was 5 us of a Q block's ~10 us prologue
// (tools/wg_profile.py, 8 x 512). Returns the sequence or -1 (a surplus Q block); all results are wave-uniform.
__device__ __forceinline__ int find_seq_and_lengths(const int32_t* __restrict__ cu, const int32_t* __restrict__ seqused, int num_seqs, int qblock,
                                                    int block_q, int lane, int& q_start, int& q_len, int& seq_len) {
  if (num_seqs <= 63) {
    const int c = cu[min(lane, num_seqs)];                        // lanes 0 .. S hold cu[0 .. S]
    const int k = seqused[min(lane, num_seqs - 1)];
    const unsigned long long m = __ballot(lane < num_seqs && c / block_q + lane <= qblock);
    const int seq = __builtin_popcountll(m) - 1;
    if (seq < 0) { q_start = q_len = seq_len = 0; return -1; }
    q_start = __builtin_amdgcn_readlane(c, seq);
    q_len = __builtin_amdgcn_readlane(c, seq + 1) - q_start;
    seq_len = __builtin_amdgcn_readlane(k, seq);
    return seq;
  }
  int left = 0, right = num_seqs;
  while (left < right) {
    const int mid = (left + right) >> 1;
    if (cu[mid] / block_q + mid <= qblock) left = mid + 1; else right = mid;
  }
  const int seq = left - 1;
  if (seq < 0) { q_start = q_len = seq_len = 0; return -1; }
  q_start = cu[seq];
  q_len = cu[seq + 1] - q_start;
  seq_len = seqused[seq];
  return seq;
}

// Head sizes the MFMA kernels are built for; any other multiple of 8 (16 with an fp8 cache) up to 256 runs on the
// next one with the missing columns never loaded (zero operands) and never stored - the reference pads to the
// next power of two the same way (HEAD_SIZE_PADDED, triton_unified_attention.py:353,:912). 0 = not served.
__host__ __device__ inline int padded_head_size(int d, bool fp8_kv) {
  if (d < 16 || d > 256 || d % (fp8_kv ? 16 : 8) != 0) return 0;
  return d <= 64 ? 64 : d <= 128 ? 128 : 256;
}

// wave64 reductions
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Exchange with the lane 32 (16) away. v_permlane{32,16}_swap swaps halves (odd/even rows) of TWO
// registers; hipcc (ROCm 7.2) folds the two results into one when both operands are the same SSA
// value, so the copy is made opaque first.
__device__ __forceinline__ float lane_xor32(float v) {
  uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
  asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  // lanes < 32: r[0] = own, r[1] = lane+32's; lanes >= 32: r[0] = lane-32's, r[1] = own
  return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float lane_xor16(float v) {
  uint32_t a = __builtin_bit_cast(uint32_t, v), b = a;
  asm volatile("" : "+v"(b));
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  // even rows: r[0] = own, r[1] = lane+16's; odd rows: r[0] = lane-16's, r[1] = own
  return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]);
}

// Four wave-uniform 32-bit loads through the scalar cache, returned only after they have landed.
// hipcc falls back to vector loads (and then drains every older vector load with them, because
// vmcnt retires in order) for uniform loads it cannot prove read-only, so block-table lookups in
// a software-pipelined loop go through this instead. `base` and `idx*` must be wave-uniform.
__device__ __forceinline__ void scalar_load4(const int32_t* base, int i0, int i1, int i2, int i3, int& r0, int& r1, int& r2, int& r3) {
  const int o0 = __builtin_amdgcn_readfirstlane(i0 * 4), o1 = __builtin_amdgcn_readfirstlane(i1 * 4);
  const int o2 = __builtin_amdgcn_readfirstlane(i2 * 4), o3 = __builtin_amdgcn_readfirstlane(i3 * 4);
  const uint64_t b = (uint64_t)base;
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint64_t bu = ((uint64_t)bhi << 32) | blo;
  asm volatile(
      "s_load_dword %0, %4, %5\n\t"
      "s_load_dword %1, %4, %6\n\t"
      "s_load_dword %2, %4, %7\n\t"
      "s_load_dword %3, %4, %8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(r0), "=&s"(r1), "=&s"(r2), "=&s"(r3)
      : "s"(bu), "s"(o0), "s"(o1), "s"(o2), "s"(o3)
      : "memory");
}

// Four fp8 values (one 32-bit word) -> four values of the 16-bit type T as two packed words, exact. gfx950 converts a
// PAIR of fp8 straight to packed bf16/f16 in one instruction (v_cvt_scalef32_pk_{bf16,f16}_{fp8,bf8}, scale 1.0):
// half the instructions of fp8 -> f32 -> pack.
template <typename T, typename KVT>
__device__ __forceinline__ void widen_fp8x4(uint32_t in, uint32_t& lo, uint32_t& hi) {
  if constexpr (__is_same(T, bf16_t)) {
    typedef __attribute__((ext_vector_type(2))) __bf16 pk_t;
    if constexpr (__is_same(KVT, e4m3_t)) {
      lo = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(in, 1.0f, false));
      hi = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(in, 1.0f, true));
    } else {
      lo = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(in, 1.0f, false));
      hi = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_bf16_bf8(in, 1.0f, true));
    }
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 pk_t;
    if constexpr (__is_same(KVT, e4m3_t)) {
      lo = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(in, 1.0f, false));
      hi = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_f16_fp8(in, 1.0f, true));
    } else {
      lo = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(in, 1.0f, false));
      hi = __builtin_bit_cast(uint32_t, (pk_t)__builtin_amdgcn_cvt_scalef32_pk_f16_bf8(in, 1.0f, true));
    }
  }
}

// One word at `w` and two words of `base` in the same scalar round trip (all wave-uniform).
__device__ __forceinline__ void scalar_load_word_and_pair(const int32_t* w, const int32_t* base, int i0, int i1, int& rw, int& r0, int& r1) {
  const int o0 = __builtin_amdgcn_readfirstlane(i0 * 4), o1 = __builtin_amdgcn_readfirstlane(i1 * 4);
  const uint64_t b = (uint64_t)base, ww = (uint64_t)w;
  // readfirstlane returns int: take each half through uint32_t or the low half sign-extends into the high one
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t wlo = __builtin_amdgcn_readfirstlane((uint32_t)ww), whi = __builtin_amdgcn_readfirstlane((uint32_t)(ww >> 32));
  const uint64_t bu = ((uint64_t)bhi << 32) | blo, wu = ((uint64_t)whi << 32) | wlo;
  asm volatile(
      "s_load_dword %0, %3, 0x0\n\t"
      "s_load_dword %1, %4, %5\n\t"
      "s_load_dword %2, %4, %6\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(rw), "=&s"(r0), "=&s"(r1)
      : "s"(wu), "s"(bu), "s"(o0), "s"(o1)
      : "memory");
}

// The same plus one more word at `w2` (all wave-uniform), still ONE scalar round trip.
__device__ __forceinline__ void scalar_load_2words_and_pair(const int32_t* w, const int32_t* w2, const int32_t* base, int i0, int i1, int& rw, int& rw2, int& r0, int& r1) {
  const int o0 = __builtin_amdgcn_readfirstlane(i0 * 4), o1 = __builtin_amdgcn_readfirstlane(i1 * 4);
  const uint64_t b = (uint64_t)base, ww = (uint64_t)w, w2w = (uint64_t)w2;
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)b), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
  const uint32_t wlo = __builtin_amdgcn_readfirstlane((uint32_t)ww), whi = __builtin_amdgcn_readfirstlane((uint32_t)(ww >> 32));
  const uint32_t xlo = __builtin_amdgcn_readfirstlane((uint32_t)w2w), xhi = __builtin_amdgcn_readfirstlane((uint32_t)(w2w >> 32));
  const uint64_t bu = ((uint64_t)bhi << 32) | blo, wu = ((uint64_t)whi << 32) | wlo, xu = ((uint64_t)xhi << 32) | xlo;
  asm volatile(
      "s_load_dword %0, %4, 0x0\n\t"
      "s_load_dword %1, %5, 0x0\n\t"
      "s_load_dword %2, %6, %7\n\t"
      "s_load_dword %3, %6, %8\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&s"(rw), "=&s"(rw2), "=&s"(r0), "=&s"(r1)
      : "s"(wu), "s"(xu), "s"(bu), "s"(o0), "s"(o1)
      : "memory");
}

// One LDS-DMA piece: every lane fetches 16 bytes from its own global address and the wave's 1 KiB
// lands contiguously at LDS byte address `lds_dst` (wave-uniform) + lane*16.
// Issued through inline asm ON PURPOSE: for the builtin form hipcc (ROCm 7.2) orders every later
// LDS read behind the DMA with an s_waitcnt vmcnt(0), i.e. the wave sits out the whole HBM/L2 round
// trip before it may touch the OTHER stage, and the prefetch overlaps nothing. With the asm form the
// compiler does not know a load is in flight; the caller retires it with glds_wait_all() before the
// barrier that publishes the stage (cdna_hip_programming.md 5.7 item 1, M0 recipe).
__device__ __forceinline__ void glds16(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
// The same with a SCALAR base: lane l's 16 bytes from sbase + voff land at `lds_dst` + 16 l. (M0 is not saved: the kernels
// that use this form have no other use of it - no indirect register indexing, no GWS / message instructions.)
__device__ __forceinline__ void glds16_s(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)sbase), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32));
  const uint64_t bu = ((uint64_t)bhi << 32) | blo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(bu), "s"(dst) : "memory");
}
// The 4-byte form: lane l's dword lands at `lds_dst` + 4*l (256 bytes per wave-instruction).
__device__ __forceinline__ void glds4(const void* gsrc, uint32_t lds_dst) {
  unsigned keep;
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ void glds_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
  return (uint32_t)(size_t)(__attribute__((address_space(3))) const void*)p;
}

// x * tanh(s / x) (reference: apply_softcap, triton_unified_attention.py:24-29, restated with tanhf
// so that |s/x| > 88 does not overflow)
__device__ __forceinline__ float softcap_fn(float s, float cap) { return cap * tanhf(s / cap); }
// The same for the 16-bit matrix-core kernels, where a score costs issue slots: tanh(z) = 1 - 2 / (1 + e^(2z)) as one
// v_exp, one v_rcp and three more instructions (tanhf is a few dozen). Absolute error ~1e-6 * cap - three orders of
// magnitude inside what a 16-bit P resolves; saturates cleanly (e^(2z) = inf -> 1, 0 -> -1). `k` = 2 * log2(e) / cap.
__device__ __forceinline__ float softcap_fast(float s, float cap, float k) {
  const float e = __builtin_amdgcn_exp2f(s * k);
  return cap - 2.0f * cap * __builtin_amdgcn_rcpf(1.0f + e);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
void set_error(const char* fmt, ...);
void set_kernel_name(const char* name);

// launchers implemented per translation unit; each returns MI355_OK / MI355_ERR_*
int launch_generic(const mi355_attn_params& p, hipStream_t stream);
int launch_cache_write(const mi355_cache_params& p, hipStream_t stream);

bool decode_supported(const mi355_attn_params& p);
int decode_pack_groups(const mi355_attn_params& p);        // 0: not a packed multi-token decode step; 1 / 2: column groups (16 matrix columns each) per wave
int decode_rows_max_q(const mi355_attn_params& p);         // mixed batch: sequences with up to this many query tokens are the decode launch's rows
int decode_pack_shift(const mi355_attn_params& p);         // > 0: a multi-token decode step the PACK decode kernels take (log2 of the tokens per work unit)
bool decode_write_fusable(const mi355_attn_params& p);     // decode step whose cache write can ride the attention launch
size_t decode_workspace_bytes(const mi355_attn_params& p);
int launch_decode(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream);

bool prefill_supported(const mi355_attn_params& p);
// key-split launch context (prefill_mfma.hip: plan_key_splits / launch_prefill_ws); nullptr = one workgroup walks all keys
struct KeySplitCtx { int splits, tiles_per_split; bool wide; int64_t out_split_stride, lse_split_stride; };
// counters: the zero-filled head of the caller's workspace (>= 256 KiB) or null
int launch_prefill(const mi355_attn_params& p, hipStream_t stream, const KeySplitCtx* ks = nullptr, int* counters = nullptr);
size_t prefill_workspace_bytes(const mi355_attn_params& p);                                   // key-split partials, 0 if none
int launch_prefill_ws(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream);
const char* mi355_last_kernel_name();
int launch_merge_partials(const void* part_out, const float* part_lse, int parts, void* out, float* lse, int dtype, int num_tokens,
                          int num_q_heads, int head_size, int64_t out_stride_token, int64_t out_stride_head, int64_t lse_stride_token, hipStream_t stream);
// The first 256 KiB of every workspace (zero on entry, zero on exit): [0, 192 KiB) arrival / ticket counters of the decode
// and prefill kernels, [192 KiB, 256 KiB) one byte per (128-row Q block, KV head) of a prefill call: the rows of an f16
// launch of prefill_pw_kernel whose scores left the range its per-row reference covers are flagged there and computed
// again by the register-staged kernel (a true running maximum) in a launch of its own, which clears the flags it serves.
constexpr size_t kWsCountersBytes = (size_t)192 << 10;
constexpr size_t kWsFixFlagOffset = kWsCountersBytes, kWsFixFlagBytes = (size_t)64 << 10;
// legacy layouts / linear new-token source -> flash-layout scratch cache (repack.hip)
bool repack_supported(const mi355_attn_params& p);
size_t repack_scratch_bytes(const mi355_attn_params& p, size_t head);
mi355_attn_params repacked_params(const mi355_attn_params& p, void* scratch, size_t head);
int launch_repack(const mi355_attn_params& p, void* scratch, size_t head, bool skip_single, hipStream_t stream);
bool prefill_pw_applicable(const mi355_attn_params& p);    // beyond prefill_supported()
int launch_prefill_pw(const mi355_attn_params& p, int key_splits, int64_t out_split_stride, int64_t lse_split_stride, int* counters, hipStream_t stream,
                      uint8_t* fix_flags = nullptr);   // fix_flags: see kWsFixFlagOffset (null: out-of-range rows are recomputed inside the launch)
bool prefill_pw_selected(const mi355_attn_params& p, const KeySplitCtx* ks);   // launch_prefill would hand this call to prefill_pw_kernel
bool prefill_lat_applicable(const mi355_attn_params& p);   // short-prompt (latency) prefill kernel, prefill_lat.hip: beyond prefill_supported()
int launch_prefill_lat(const mi355_attn_params& p, hipStream_t stream);
bool prefill_lat_selected(const mi355_attn_params& p);     // launch_prefill hands this call to prefill_lat_kernel
bool prefill_dma_selected(const mi355_attn_params& p);     // launch_prefill (no key splits) hands this call to an LDS-DMA kernel
bool prefill_runs_without_key_splits(const mi355_attn_params& p);
bool prefill_write_fusable(const mi355_attn_params& p);    // a prefill step whose cache write can ride the attention launch (write_new_kv)

inline int check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return MI355_OK;
  set_error("%s: %s", what, hipGetErrorString(e));
  return MI355_ERR_HIP;
}

// > 64 KiB of dynamic LDS needs hipFuncAttributeMaxDynamicSharedMemorySize, per kernel AND per device. `done` is the
// kernel's own bit mask of devices that have it (one static std::atomic per launcher): set once per device, from any
// thread (two threads racing both set it: harmless), no process-global "already done" flag.
inline int ensure_dynamic_lds(const void* kernel, int bytes, std::atomic<uint64_t>& done, const char* what) {
  int dev = 0;
  const int rc0 = check_hip(hipGetDevice(&dev), "hipGetDevice");
  if (rc0 != MI355_OK) return rc0;
  const uint64_t bit = 1ull << (dev & 63);
  if (dev < 64 && (done.load(std::memory_order_acquire) & bit)) return MI355_OK;
  const int rc = check_hip(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
  if (rc == MI355_OK && dev < 64) done.fetch_or(bit, std::memory_order_release);
  return rc;
}

}  // namespace mi355
