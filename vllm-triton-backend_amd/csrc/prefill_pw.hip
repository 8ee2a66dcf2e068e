// prefill_pw_kernel: the D = 128 prefill fast path. Four waves, 64 query rows each, one wave per SIMD.
//
// Replaces kernel_unified_attention_2d (LIB/kernels/triton_unified_attention.py:275-523) for the 16-bit case
// without soft-cap / ALiBi: bf16 or f16, with or without a sliding window (:474-479), causal or - for
// prefill_flash_attention(causal=False) - not; same semantics as prefill_mfma.hip's kernels. An fp8 cache reaches it
// through the dequantising pass of repack.hip.
//
// Why this shape. prefill_dma_kernel (eight waves of 32 rows, two per SIMD) issues 169 VALU + 106 SALU + 50 LDS
// instructions per 32 MFMAs and keeps the matrix pipes 43 % busy. Here
//   * a wave owns TWO 32-row sub-blocks (A, B): every K fragment and every transposed V fragment read from
//     LDS feeds two MFMAs, and the wave issues HALF the LDS reads, LDS-DMA and barriers per MFMA;
//   * the two sub-blocks run half a tile apart ("ping-pong" inside one wave): while the matrix pipe computes
//     S_A = K.Q_A^T the VALU finishes sub-block B's exponentials, then O_B += V^T.P_B runs beside A's
//     exponentials, and so on - every MFMA has a few independent VALU instructions in its shadow and no
//     phase of the loop is VALU-only or MFMA-only. One wave per SIMD has nobody to hide its bubbles, so the
//     interleave is written out gap by gap (the work that follows each MFMA is fixed in the source);
//   * the whole 512-entry register file: O (128), Q (64) and the current K tile (64) live in accumulator
//     registers that only MFMA / ds_read touch (literal a[..] operands, never seen by the register allocator);
//     S, P, V fragments and the softmax state are ordinary VGPRs;
//   * K/V tiles arrive by LDS-DMA with a SCALAR page base + one constant per-lane offset (global_load_lds with
//     an SGPR address): wave w fetches the 16-key group w of every tile, i.e. one block-table entry per tile and
//     matrix - read through the scalar cache one iteration before it is needed - and no per-lane 64-bit address
//     arithmetic; three-slot rings, K two tiles ahead, V one; counted vmcnt;
//   * a single wave issues in order, so every instruction of the loop costs issue time nobody else fills. bf16
//     only, and NO running maximum in the loop: P = 2^s with the fixed reference 0 has the same relative precision
//     as 2^(s - max) (bf16 and f32 are floating point), S^T starts from the inline constant 0, and neither maxima
//     nor rescaling are computed per tile. What a running maximum protects against - P or a sum leaving the range
//     of the format - is CHECKED once per row when the block is done (2^-64 <= l <= 2^100, O finite); a row that
//     fails (scores beyond +-90 or so in log2 units: not attention as models produce it) is computed again by a
//     slow, plain online-softmax routine in the same launch (pw_row_fallback).
//     (Round 3, 16x16x32 form: the reference is PER ROW - minus the row's largest score over its first sixteen keys, taken
//     once per item by sixteen matrix instructions (set_references) and fed to every score chain as its C operand - so a
//     row's P stays near 1 whatever its scores' level; that is what lets f16 (P <= 65504) run here, with a margin of 2^6
//     and tighter range checks, and what makes the fallback a matter of rows whose scores RISE by more than the format's
//     range after their first sixteen keys.)
//   * the tile loop has two forms: STEADY iterations (tile unmasked for the wave, the groups it fetches for the next
//     tiles whole) carry no mask path and no tail handling and put their scalar address arithmetic behind the first
//     MFMAs; general iterations do the rest (a Q block's last tiles, short sequences).
//   * a workgroup walks SEVERAL work items (about one workgroup per CU; dealt statically for one sequence, by ticket
//     counters in the caller's workspace for several): the next
//     item's query rows and first K/V tiles are requested before the current item's output is normalised and stored,
//     so that chain of round trips runs beside the epilogue instead of in front of an idle CU. Everything the per-item
//     code needs is re-derived per item from opaque copies of the lane index and the kernarg pointer: the tile loop has
//     no register to spare for values hoisted out of the item loop (tools/isa_audit.py must report no compiler
//     accumulator-register or scratch use).
//
//   * FIVE instantiations <T, M16, SW>. The text above describes the 32x32x16 one (<bf16, false, false>: MFMA orientation,
//     in-register softmax layout and LDS swizzles of prefill_dma_kernel, prefill_mfma.hip; kept for A/B). The product is
//     M16 = true, for bf16 and f16, each with (SW) and without the sliding-window mask: both contractions on
//     v_mfma_f32_16x16x32_{bf16,f16} - the chip runs this kernel at its power limit and holds a higher clock under that shape -
//     with what its doubled matrix-instruction count asks for: the row sums on the matrix pipe (l += 1.P^T), the
//     exponentials dealt ONE per 16-cycle gap from the moment a sub-block's first score tile is done, LDS-DMA split over
//     two segments, a V swizzle that matches its transposed reads. One wave per SIMD pays ~4 cycles of issue for every
//     instruction, ~9 for a matrix instruction, with nothing overlapping inside the wave (tools/probes/issue_model.hip).
//   * the tile loop's VGPR state (S, the ring of exponentials, row sums, P) is HAND-OWNED: its asm statements name those
//     registers as inputs and write them; see the state block in the kernel.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(16))) float wf32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int wu32x4_t;
typedef __attribute__((ext_vector_type(4))) float wf32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int wu32x2_t;

constexpr int kPwTile = 64;          // keys per KV tile
constexpr int kPwRows = 256;         // Q-block rows per workgroup
constexpr float kPwLog2e = 1.4426950408889634f;
constexpr float kPwSumLo = 5.421010862427522e-20f;   // 2^-64: below it a row's largest P may be close to the subnormals
constexpr float kPwSumHi = 1.2676506002282294e30f;   // 2^100: above it a P or a partial sum may have overflowed
constexpr float kPwRefMax = 1024.0f;                  // |reference| (log2 units) beyond which a row goes to the f32 routine (rounding of Q')
constexpr float kPwSumLoF16 = 2.44140625e-4f;        // f16: 2^-12 (P's grid ends at 2^-24: below this sum too few bits are left)

// accumulator-register map (owned by the asm statements below)
constexpr int kAO = 0;     // O^T[sb][b]  : kAO + 64 sb + 16 b   (16 registers)
constexpr int kAQ = 128;   // Q'[sb][ks]  : kAQ + 32 sb + 4 ks   (4 registers)
constexpr int kAK = 192;   // K[kb][ks]   : kAK + 32 kb + 4 ks   (4 registers)
// LDS map: K ring (3 x 16 KiB), V ring (3 x 16 KiB)
// then the parking area of the epilogue: 32 rows of 272 bytes per wave (a sub-block of O on its way out). It lies
// outside the rings because the next Q block's first tiles are already landing in them while O leaves.
constexpr int kSlotBytes = 16384, kLdsK = 0, kLdsV = 3 * kSlotBytes, kLdsO = 6 * kSlotBytes;
constexpr int kPwORS = 256 + 16;     // padded row stride of the parked O rows
constexpr int kLdsT = kLdsO + 4 * 32 * kPwORS;   // two ints: the item index wave 0 drew, published to the workgroup
constexpr int kPwLds = kLdsT + 16;
// fp8 cache (KV8 instantiations): behind everything else, a staging area of 4 KiB per wave - its 16-key group of the next K tile
// (2 KiB of fp8) and of the next V tile, as LDS-DMA delivers them
constexpr int kLdsS = kPwLds, kPwLdsF8 = kLdsS + 4 * 4096;
static_assert(kLdsS % 16 == 0 && kPwLdsF8 <= 160 * 1024, "prefill_pw_kernel: LDS map");
constexpr size_t kPwCounterBytes = (size_t)256 << 10;   // the counter region at the head of every workspace (include/mi355_attn.h)

struct PwArgs {
  mi355_attn_params p;
  int group;       // G
  int block_q;     // tokens per Q block = 256 / G
  int page_shift;  // log2(page_size)
  int key_splits;  // a work item (Q block, KV head, s) attends the s-th even share of its Q block's key tiles (prefill_mfma.hip)
  int num_qblocks; // static upper bound of the Q blocks of the batch (num_tokens / block_q + num_seqs)
  int g_inv, bq_shift;     // ceil(2^16 / group): x / group == (x * g_inv) >> 16 for x < 256; log2 of block_q when it is a power of two, else -1
  int* tickets;    // [2 * num_kv_heads] zero on entry, zero on exit (head of the caller's workspace), or null: static deal
  int slots;       // workgroups per KV head: the grid is slots * num_kv_heads workgroups, each walking several items
  int64_t out_split_stride, lse_split_stride;
  uint32_t k_page_stride, k_slot_stride, v_page_stride, v_slot_stride;  // elements; validated < 2^31 on the host
  // f16 launches: a row whose scores left the range (see the epilogue) is not recomputed here, one key at a time, but
  // FLAGGED - byte (q_start / fix_bq + seq + token / fix_bq) * Hk + head: its 128-row Q block in the register-staged
  // kernel's numbering - and that kernel computes the flagged blocks again in the next launch. Null: recompute in place.
  uint8_t* fix_flags;
  int fix_bq;      // tokens per 128-row Q block = 128 / G
};

template <int N> using ic = std::integral_constant<int, N>;
template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F&& f, std::integer_sequence<int, I...>) { (f(ic<I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F&& f) { sfor_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// ---- single instructions the loop is built from. asm volatile: they stay in source order, which IS the schedule.
// hipcc neither pads nor counts anything inside (cdna_hip_programming.md 5.7); the placement rules are in the kernel.
__device__ __forceinline__ float a_exp2(float x) { float r; asm volatile("v_exp_f32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ float a_add(float x, float y) { float r; asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; }
__device__ __forceinline__ void a_exp2x2(float& r0, float& r1, float x0, float x1) {
  asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" : "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1));
}
// the same into hand-owned registers (named as inputs, see the state block of prefill_pw_kernel)
__device__ __forceinline__ void a_exp2x2_ho(float r0, float r1, float x0, float x1) {
  asm volatile("v_exp_f32 %0, %2\n\tv_exp_f32 %1, %3" :: "v"(r0), "v"(r1), "v"(x0), "v"(x1));
}
__device__ __forceinline__ void a_exp_ho(float r, float x) { asm volatile("v_exp_f32 %0, %1" :: "v"(r), "v"(x)); }
// ALiBi (AL instantiation; reference :481-482, S += slope * (key position - context length)): the row's slope times the
// key's position is linear in the key, so its tile- and register-dependent part rides in the chain's C operand (rebuilt
// per tile, see alibi_c16) and only slope * 16 kt is left for key tiles 1..3 of a score tile: one v_fmamk in front of the
// exponential (K = 16, 32, 48 as a literal).
template <int KT> __device__ __forceinline__ void a_alibi_exp_ho(float r, float x, float slope2) {
  static_assert(KT >= 1 && KT <= 3, "key tile 0 needs no step");
  if constexpr (KT == 1) asm volatile("v_fmamk_f32 %0, %2, 0x41800000, %1\n\tv_exp_f32 %0, %0" :: "v"(r), "v"(x), "v"(slope2));
  else if constexpr (KT == 2) asm volatile("v_fmamk_f32 %0, %2, 0x42000000, %1\n\tv_exp_f32 %0, %0" :: "v"(r), "v"(x), "v"(slope2));
  else asm volatile("v_fmamk_f32 %0, %2, 0x42400000, %1\n\tv_exp_f32 %0, %0" :: "v"(r), "v"(x), "v"(slope2));
}
// Soft-capped score -> P (SC instantiation; reference: apply_softcap, triton_unified_attention.py:55-60, before the mask :467-482).
// x = u = s * 2 log2(e) / cap (the query rows are pre-scaled for it), so cap tanh(s / cap) log2(e) = A - B / (1 + 2^u) with
// B = 2 A = 2 cap log2(e), and P = 2^(c - B / (1 + 2^u)) with the row's reference c (A drops out of the softmax; it comes back
// in the lse). Six instructions in place of one, three of them transcendental; x is overwritten with u 2^-100 + c: c for
// any finite score, -inf for a masked one (-inf), which carries the mask through to P = 2^-inf = 0. An independent
// instruction or a wait state sits between each transcendental and the VALU instruction that reads its result.
__device__ __forceinline__ void a_softcap_exp_ho(float r, float x, float c, float inv_b) {
  asm volatile("v_exp_f32 %0, %1\n\tv_fmamk_f32 %1, %1, 0x0d800000, %2\n\tv_fma_f32 %0, %0, %3, %3\n\tv_rcp_f32 %0, %0\n\ts_nop 0\n\t"
               "v_sub_f32 %0, %1, %0\n\tv_exp_f32 %0, %0" :: "v"(r), "v"(x), "v"(c), "v"(inv_b));
}
// ... and the two scores of a word together: the chains interleaved, so that every transcendental's consumer has the other
// chain's instruction in front of it and no wait state is spent (twelve instructions for two scores instead of fourteen).
__device__ __forceinline__ void a_softcap_exp2_ho(float r0, float x0, float r1, float x1, float c, float inv_b) {
  asm volatile("v_exp_f32 %0, %1\n\tv_exp_f32 %2, %3\n\tv_fmamk_f32 %1, %1, 0x0d800000, %4\n\tv_fmamk_f32 %3, %3, 0x0d800000, %4\n\t"
               "v_fma_f32 %0, %0, %5, %5\n\tv_fma_f32 %2, %2, %5, %5\n\tv_rcp_f32 %0, %0\n\tv_rcp_f32 %2, %2\n\t"
               "v_sub_f32 %0, %1, %0\n\tv_sub_f32 %2, %3, %2\n\tv_exp_f32 %0, %0\n\tv_exp_f32 %2, %2"
               :: "v"(r0), "v"(x0), "v"(r1), "v"(x1), "v"(c), "v"(inv_b));
}
// Four scores of one lane (registers r = 0..3 of a 16x16 score tile = four consecutive keys) against a bound, hand-owned
// like the forms above. HI: masked (-> ninf) where r > c; LO: where r < c. Three mask registers in rotation put two
// instructions between every compare and the select that reads it (a VALU-written SGPR wants two wait states before a
// VALU reads it: the compiler's own cmp / s_nop 1 / cndmask per element was 3 issue slots, this is 2).
template <bool LO> __device__ __forceinline__ void pw_mask_quad_ho(const wf32x4_t& sq, int c, float ninf) {
  unsigned long long m0, m1;
  if constexpr (!LO)
    asm volatile("v_cmp_gt_i32 %0, 0, %6\n\tv_cmp_gt_i32 %1, 1, %6\n\tv_cmp_gt_i32 vcc, 2, %6\n\t"
                 "v_cndmask_b32 %2, %2, %7, %0\n\tv_cmp_gt_i32 %0, 3, %6\n\tv_cndmask_b32 %3, %3, %7, %1\n\t"
                 "v_cndmask_b32 %4, %4, %7, vcc\n\tv_cndmask_b32 %5, %5, %7, %0"
                 : "=&s"(m0), "=&s"(m1) : "v"(sq[0]), "v"(sq[1]), "v"(sq[2]), "v"(sq[3]), "v"(c), "v"(ninf) : "vcc");
  else
    asm volatile("v_cmp_lt_i32 %0, 0, %6\n\tv_cmp_lt_i32 %1, 1, %6\n\tv_cmp_lt_i32 vcc, 2, %6\n\t"
                 "v_cndmask_b32 %2, %2, %7, %0\n\tv_cmp_lt_i32 %0, 3, %6\n\tv_cndmask_b32 %3, %3, %7, %1\n\t"
                 "v_cndmask_b32 %4, %4, %7, vcc\n\tv_cndmask_b32 %5, %5, %7, %0"
                 : "=&s"(m0), "=&s"(m1) : "v"(sq[0]), "v"(sq[1]), "v"(sq[2]), "v"(sq[3]), "v"(c), "v"(ninf) : "vcc");
}
// instruction q (0..47) of a sub-block's exponential / pack stream in the 16x16x32 form: kind 0 / 1 = exponential of the
// word's first / second score, 2 = pack; j = the word
struct PwEOp { int kind, j; };
constexpr PwEOp pw_eop(int q) {
  int acc = 0;
  for (int n = 0; n < 32; ++n) {
    const bool is_x = n < 2 || (n < 30 && ((n - 2) & 1) == 0);
    const int j = n < 2 ? n : n >= 30 ? 14 + (n - 30) : is_x ? 2 * (1 + (n - 2) / 4) + (((n - 2) >> 1) & 1) : 2 * ((n - 2) / 4) + (((n - 2) >> 1) & 1);
    const int cnt = is_x ? 2 : 1;
    if (q < acc + cnt) return PwEOp{is_x ? q - acc : 2, j};
    acc += cnt;
  }
  return PwEOp{-1, 0};
}
// which of the 26 stream instructions left for a P.V segment goes into its gap g: the even gaps and 1, 7, 13, 19, 25, 31, 33, 35
constexpr int pw_pv_slot(int g) {
  int n = 0;
  for (int h = 0; h < 36; ++h) {
    const bool used = (h & 1) == 0 || h == 1 || h == 7 || h == 13 || h == 19 || h == 25 || h == 31 || h == 33 || h == 35;
    if (h == g) return used ? n : -1;
    n += used ? 1 : 0;
  }
  return -1;
}
// the K(t+1) fragment read that goes into gap g of segment 4 (fragments 4 .. 15; 0 .. 3 are read in segment 3): the odd
// gaps the stream leaves free, then 7 and 13
constexpr int pw_kread_slot(int g) {
  const int gaps[12] = {3, 5, 7, 9, 11, 13, 15, 17, 21, 23, 27, 29};
  for (int i = 0; i < 12; ++i) if (gaps[i] == g) return 4 + i;
  return -1;
}
// compile-time checks of the three tables above
constexpr bool pw_tables_ok() {
  int exp_lo[16] = {}, exp_hi[16] = {}, pack[16] = {};
  for (int q = 0; q < 48; ++q) {
    const PwEOp op = pw_eop(q);
    if (op.j < 0 || op.j > 15 || op.kind < 0 || op.kind > 2) return false;
    (op.kind == 0 ? exp_lo : op.kind == 1 ? exp_hi : pack)[op.j] = q + 1;
    // a word's exponentials are dealt in the S segment from gap 10 on, one per gap: its score tile (8 (w >> 2) + 7 is the
    // tile's last matrix instruction) must be two instructions behind by then
    if (op.kind < 2 && q < 22 && q + 10 < 8 * (op.j >> 2) + 10) return false;
  }
  for (int w = 0; w < 16; ++w) {
    if (!exp_lo[w] || !exp_hi[w] || !pack[w] || exp_hi[w] != exp_lo[w] + 1 || pack[w] < exp_hi[w] + 2) return false;
    // ring of three: word w + 3's exponentials overwrite word w's registers, after word w's pack
    if (w + 3 < 16 && exp_lo[w + 3] < pack[w]) return false;
  }
  int slots = 0, kfrags = 0;
  for (int g = 0; g < 36; ++g) {
    const int sl = pw_pv_slot(g), kf = pw_kread_slot(g);
    if (sl >= 0) { if (sl != slots) return false; ++slots; }
    if (kf >= 0) { if (kf != 4 + kfrags || g < 0) return false; ++kfrags; }
  }
  return slots == 26 && kfrags == 12;
}
static_assert(pw_tables_ok(), "prefill_pw_kernel: the 16x16x32 form's instruction tables are inconsistent");
// a value the compiler knows nothing about from here on (it stays where it is: no instruction)
template <typename V> __device__ __forceinline__ void pw_launder(V& v) { asm volatile("" : "+v"(v)); }

// -DPW_ABL_MFMA16 (timing experiment only, results are garbage): every P.V matrix instruction as TWO 16x16x32 ones of
// the same total FLOPs, to see what clock the chip holds under that shape (MI355X_MICROARCH.md, DVFS give-back item 7)
#ifdef PW_ABL_MFMA16
#define PW_PV_ASM(MFMA)                                                                                             \
  asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]\n\tv_mfma_f32_16x16x32_bf16 a[%c4:%c5], %0, %1, a[%c4:%c5]" \
               :: "v"(v), "v"(pf), "n"(OA), "n"(OA + 3), "n"(OA + 4), "n"(OA + 7));
#else
#define PW_PV_ASM(MFMA) asm volatile(MFMA " a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(pf), "n"(OA), "n"(OA + 15));
#endif
template <typename T> struct pw_ops;
#define MI355_DEF_PW_OPS(TAG, MFMA, CVT)                                                                          \
  template <> struct pw_ops<TAG> {                                                                                \
    /* S(VGPR) = K(AGPR) . Q(AGPR) */                                                                              \
    template <int KA, int QA> static __device__ __forceinline__ void qk_zero(wf32x16_t& s) {                      \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));  \
    }                                                                                                             \
    /* S(VGPR) += K(AGPR) . Q(AGPR) */                                                                             \
    template <int KA, int QA> static __device__ __forceinline__ void qk_acc(wf32x16_t& s) {                       \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3)); \
    }                                                                                                             \
    /* the same right behind ordinary code that wrote s (the mask): VALU write -> MFMA read of C needs two wait */ \
    /* states, and they must sit INSIDE the statement - the compiler moves a separate s_nop in front of its own */ \
    /* v_cndmask writes (it does not know what the next statement reads) */                                        \
    template <int KA, int QA> static __device__ __forceinline__ void qk_acc_masked(wf32x16_t& s) {                \
      asm volatile("s_nop 1\n\t" MFMA " %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3)); \
    }                                                                                                             \
    /* The forms the tile loop uses: the registers of S, of the row sums and of P are named as INPUTS although the */ \
    /* statement writes them ("hand-owned", see the state block of prefill_pw_kernel). */                           \
    template <int KA, int QA> static __device__ __forceinline__ void qk_zero_ho(const wf32x16_t& s) {             \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], 0" :: "v"(s), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));   \
    }                                                                                                             \
    template <int KA, int QA> static __device__ __forceinline__ void qk_acc_ho(const wf32x16_t& s) {              \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], %0" :: "v"(s), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));  \
    }                                                                                                             \
    static __device__ __forceinline__ void sum_pack_ho(float s0, float s1, uint32_t pk, float lo, float hi) {     \
      asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\t" CVT " %2, %3, %4"                           \
                   :: "v"(s0), "v"(s1), "v"(pk), "v"(lo), "v"(hi));                                                 \
    }                                                                                                             \
    /* O(AGPR) += V(VGPR) . P(VGPR) */                                                                             \
    template <int OA> static __device__ __forceinline__ void pv(const wu32x4_t& v, const wu32x4_t& pf) {          \
      PW_PV_ASM(MFMA)                                                                                              \
    }                                                                                                             \
    static __device__ __forceinline__ uint32_t cvt(float lo, float hi) {                                          \
      uint32_t r; asm volatile(CVT " %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi)); return r;                          \
    }                                                                                                             \
    /* two row sums and one packed pair in ONE statement (hipcc pads between statements, never inside one) */     \
    static __device__ __forceinline__ void sum_pack(float& s0, float& s1, uint32_t& pk, float lo, float hi) {     \
      asm volatile("v_add_f32 %0, %0, %3\n\tv_add_f32 %1, %1, %4\n\t" CVT " %2, %3, %4"                           \
                   : "+v"(s0), "+v"(s1), "=&v"(pk) : "v"(lo), "v"(hi));                                             \
    }                                                                                                             \
  };
MI355_DEF_PW_OPS(bf16_t, "v_mfma_f32_32x32x16_bf16", "v_cvt_pk_bf16_f32")
#undef MI355_DEF_PW_OPS

// The same contractions on v_mfma_f32_16x16x32_{bf16,f16} (M16 instantiation of the kernel): 4-register accumulators.
template <typename T> struct pw_ops16;
#define MI355_DEF_PW_OPS16(TAG, MFMA, CVT, ONES)                                                                    \
  template <> struct pw_ops16<TAG> {                                                                              \
    static constexpr uint32_t kOnes = ONES;          /* a pair of 1.0 */                                            \
    template <int KA, int QA> static __device__ __forceinline__ void qk_zero(wf32x4_t& s) {                       \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], 0" : "=v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));  \
    }                                                                                                             \
    template <int KA, int QA> static __device__ __forceinline__ void qk_acc(wf32x4_t& s) {                        \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], %0" : "+v"(s) : "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3)); \
    }                                                                                                             \
    /* hand-owned forms (see the state block of prefill_pw_kernel): the written registers are named as inputs */  \
    /* a chain starts from the row's reference: C = -m_ref in all four registers (see the reference block) */     \
    template <int KA, int QA> static __device__ __forceinline__ void qk_ref_ho(const wf32x4_t& s, const wf32x4_t& r) { \
      asm volatile(MFMA " %0, a[%c2:%c3], a[%c4:%c5], %1" :: "v"(s), "v"(r), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3)); \
    }                                                                                                             \
    template <int KA, int QA> static __device__ __forceinline__ void qk_acc_ho(const wf32x4_t& s) {               \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], %0" :: "v"(s), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));  \
    }                                                                                                             \
    /* a chain that starts from 0 (soft-cap: the raw score is needed, the reference enters after the cap) */        \
    template <int KA, int QA> static __device__ __forceinline__ void qk_zero_ho(const wf32x4_t& s) {              \
      asm volatile(MFMA " %0, a[%c1:%c2], a[%c3:%c4], 0" :: "v"(s), "n"(KA), "n"(KA + 3), "n"(QA), "n"(QA + 3));   \
    }                                                                                                             \
    /* row sums on the matrix pipe: l += 1 . P^T (A = sixteen rows of ones: every register of l holds the row's */ \
    /* whole sum) */                                                                                               \
    static __device__ __forceinline__ void lsum_ho(const wf32x4_t& l, const wu32x4_t& ones, const wu32x4_t& pf) { \
      asm volatile(MFMA " %0, %1, %2, %0" :: "v"(l), "v"(ones), "v"(pf));                                          \
    }                                                                                                             \
    static __device__ __forceinline__ void pack_ho(uint32_t pk, float lo, float hi) {                             \
      asm volatile(CVT " %0, %1, %2" :: "v"(pk), "v"(lo), "v"(hi));                                                \
    }                                                                                                             \
    template <int OA> static __device__ __forceinline__ void pv(const wu32x4_t& v, const wu32x4_t& pf) {          \
      asm volatile(MFMA " a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(pf), "n"(OA), "n"(OA + 3));               \
    }                                                                                                             \
  };
MI355_DEF_PW_OPS16(bf16_t, "v_mfma_f32_16x16x32_bf16", "v_cvt_pk_bf16_f32", 0x3f803f80u)
MI355_DEF_PW_OPS16(f16_t, "v_mfma_f32_16x16x32_f16", "v_cvt_pk_f16_f32", 0x3c003c00u)
#undef MI355_DEF_PW_OPS16

template <typename T> __device__ __forceinline__ float pw_lo(uint32_t w);
template <typename T> __device__ __forceinline__ float pw_hi(uint32_t w);
template <> __device__ __forceinline__ float pw_lo<bf16_t>(uint32_t w) { return bf16_to_f32((uint16_t)(w & 0xffff)); }
template <> __device__ __forceinline__ float pw_hi<bf16_t>(uint32_t w) { return bf16_to_f32((uint16_t)(w >> 16)); }
template <> __device__ __forceinline__ float pw_lo<f16_t>(uint32_t w) { return f16_to_f32((uint16_t)(w & 0xffff)); }
template <> __device__ __forceinline__ float pw_hi<f16_t>(uint32_t w) { return f16_to_f32((uint16_t)(w >> 16)); }
template <typename T> __device__ __forceinline__ uint32_t pw_pack(float lo, float hi) {
  if constexpr (__is_same(T, bf16_t)) return pack_bf16x2(lo, hi); else return pack_f16x2(lo, hi);
}

template <int IDX> __device__ __forceinline__ void acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a%c0, %1" :: "n"(IDX), "v"(v)); }
template <int IDX> __device__ __forceinline__ void acc_zero() { asm volatile("v_accvgpr_write_b32 a%c0, 0" :: "n"(IDX)); }
// running maximum of magnitudes: m = max(m, |a|, |b|) in one instruction (a NaN operand is dropped, like fmaxf)
__device__ __forceinline__ void pw_amax3(float& m, float a, float b) { asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(a), "v"(b)); }
template <int IDX> __device__ __forceinline__ uint32_t acc_read_u32() { uint32_t r; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r) : "n"(IDX)); return r; }
// 16 bytes from sbase + voff (+ an immediate) straight into four accumulator registers: the destination is a literal in
// the statement, the compiler never sees it (nothing to copy early, no VGPR held while the load is in flight). The
// caller retires it with an s_waitcnt of its own.
template <int AG, int IMM> __device__ __forceinline__ void pw_gload16_acc(uint32_t voff, uint64_t sbase) {
  asm volatile("global_load_dwordx4 a[%c0:%c1], %2, %3 offset:%c4" :: "n"(AG), "n"(AG + 3), "v"(voff), "s"(sbase), "n"(IMM) : "memory");
}
template <int IDX> __device__ __forceinline__ float acc_read() { float r; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r) : "n"(IDX)); return r; }

// LDS -> accumulator registers (K fragments), LDS -> VGPR transposed (V fragments). "memory": LDS accesses the
// compiler emits itself (block-table lookups) stay on their side of these.
template <int AG, int OFF> __device__ __forceinline__ void lds_to_acc_b128(uint32_t addr) {
  asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%c3" :: "v"(addr), "n"(AG), "n"(AG + 3), "n"(OFF) : "memory");
}
template <int OFF> __device__ __forceinline__ wu32x2_t lds_tr_b64(uint32_t addr) {
  wu32x2_t r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
// One LDS-DMA piece with a scalar base: lane l's 16 bytes from sbase + voff land at LDS lds_dst + 16 l.
// (M0 is not saved: nothing else in this kernel uses it - tools/isa_audit.py checks the compiler's side.)
__device__ __forceinline__ void pw_glds16(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// 16-byte global load the compiler does not count: its own vmcnt waits would otherwise also wait for the LDS-DMA
// issued behind it. The caller retires it with a counted s_waitcnt that names the destination.
__device__ __forceinline__ wu32x4_t pw_gload16(const void* ptr) {
  wu32x4_t r;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
  return r;
}
// the same with a scalar base and a 32-bit per-lane byte offset (+ an immediate)
template <int IMM> __device__ __forceinline__ wu32x4_t pw_gload16_s(uint32_t voff, uint64_t sbase) {
  wu32x4_t r;
  asm volatile("global_load_dwordx4 %0, %1, %2 offset:%c3" : "=v"(r) : "v"(voff), "s"(sbase), "n"(IMM) : "memory");
  return r;
}
// one block-table entry through the scalar cache; lands before the iteration-end s_waitcnt lgkmcnt(0) that names it
__device__ __forceinline__ void pw_sload(int& dst, uint64_t base, int byte_off) {
  asm volatile("s_load_dword %0, %1, %2" : "=s"(dst) : "s"(base), "s"(byte_off) : "memory");
}

// fp8 cache: 16 staged bytes of this lane -> VGPRs, and a widened 16-byte chunk -> its place in a ring slot. Fixed in the
// instruction stream like everything else of the loop; the reads are retired by a counted s_waitcnt that names them.
template <int OFF> __device__ __forceinline__ wu32x4_t pw_lds_read128(uint32_t addr) {
  wu32x4_t r;
  asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(r) : "v"(addr), "n"(OFF) : "memory");
  return r;
}
template <int OFF> __device__ __forceinline__ void pw_lds_write128(uint32_t addr, const wu32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1 offset:%c2" :: "v"(addr), "v"(v), "n"(OFF) : "memory");
}
__device__ __forceinline__ void pw_lds_write128_rt(uint32_t addr, const wu32x4_t& v) {
  asm volatile("ds_write_b128 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
// one word of four fp8 -> two packed pairs of T (exact: every fp8 value is a bf16 / f16 value), in place in the stream
template <typename T, typename KVT> __device__ __forceinline__ void pw_widen4(uint32_t w, uint32_t& lo, uint32_t& hi) {
  if constexpr (__is_same(T, bf16_t) && __is_same(KVT, e4m3_t))
    asm volatile("v_cvt_scalef32_pk_bf16_fp8 %0, %2, 1.0\n\tv_cvt_scalef32_pk_bf16_fp8 %1, %2, 1.0 op_sel:[1,0,0]" : "=&v"(lo), "=&v"(hi) : "v"(w));
  else if constexpr (__is_same(T, bf16_t))
    asm volatile("v_cvt_scalef32_pk_bf16_bf8 %0, %2, 1.0\n\tv_cvt_scalef32_pk_bf16_bf8 %1, %2, 1.0 op_sel:[1,0,0]" : "=&v"(lo), "=&v"(hi) : "v"(w));
  else if constexpr (__is_same(KVT, e4m3_t))
    asm volatile("v_cvt_scalef32_pk_f16_fp8 %0, %2, 1.0\n\tv_cvt_scalef32_pk_f16_fp8 %1, %2, 1.0 op_sel:[1,0,0]" : "=&v"(lo), "=&v"(hi) : "v"(w));
  else
    asm volatile("v_cvt_scalef32_pk_f16_bf8 %0, %2, 1.0\n\tv_cvt_scalef32_pk_f16_bf8 %1, %2, 1.0 op_sel:[1,0,0]" : "=&v"(lo), "=&v"(hi) : "v"(w));
}

__device__ __forceinline__ int pw_find_seq(const int32_t* __restrict__ cu, int num_seqs, int qblock, int block_q) {
  int left = 0, right = num_seqs;     // largest i with cu[i] / block_q + i <= qblock (find_seq_idx, :32-52)
  while (left < right) {
    const int mid = (left + right) >> 1;
    if (cu[mid] / block_q + mid <= qblock) left = mid + 1; else right = mid;
  }
  return left - 1;
}

// One query row against keys [key_lo, key_hi), the plain way: all 64 lanes on the row, two head dimensions each, the
// textbook online softmax with a true running maximum in f32 (kernel_unified_attention_2d, :467-510, one key at a
// time). Slow on purpose - it only ever runs for rows whose scores left the range prefill_pw_kernel's fixed
// reference is good for - and independent of everything the fast path keeps on chip: K/V straight from the cache.
template <typename T, int KV8, typename ArgPtr>
__device__ __forceinline__ void pw_row_fallback(ArgPtr kp, const int32_t* bt, const char* kbase, const char* vbase,
                                             int token, int hq, int key_lo, int key_hi, uint16_t* out_base, float* lse_base, int lane, int ctx_len) {
  const bool act = 2 * lane < kp->p.head_size;            // (head size 64: half the lanes; the others add 0 to every score and store nothing)
  const int dl = act ? 2 * lane : 0;
  const uint32_t qw = *(const uint32_t*)((const uint16_t*)kp->p.q + (int64_t)token * kp->p.q_stride_token + (int64_t)hq * kp->p.q_stride_head + dl);
  const float q0 = act ? pw_lo<T>(qw) : 0.0f, q1 = act ? pw_hi<T>(qw) : 0.0f;
  using KVT = std::conditional_t<KV8 == 2, e5m2_t, std::conditional_t<KV8 == 1, e4m3_t, T>>;
  const float k_sc = (KV8 && kp->p.k_scale) ? kp->p.k_scale[0] : 1.0f, v_sc = (KV8 && kp->p.v_scale) ? kp->p.v_scale[0] : 1.0f;   // (fp8 cache: the scales outside the sums, as in the fast path)
  const float scale2 = kp->p.scale * kPwLog2e * k_sc;
  const float cap = kp->p.softcap, cap2 = cap * kPwLog2e;
  const float slope2 = kp->p.alibi_slopes ? kp->p.alibi_slopes[hq] * kPwLog2e : 0.0f;     // ALiBi: + slope * (key position - context length), :481-482
  const int page_mask = kp->p.page_size - 1;
  float m = -INFINITY, l = 0.0f, a0 = 0.0f, a1 = 0.0f;
  for (int j = key_lo; j < key_hi; ++j) {
    const int64_t page = bt[j >> kp->page_shift];
    const int64_t slot = j & page_mask;
    uint32_t kw, vw;
    if constexpr (KV8 == 0) {
      kw = *(const uint32_t*)(kbase + (page * kp->k_page_stride + slot * kp->k_slot_stride) * 2 + 2 * dl);
      vw = *(const uint32_t*)(vbase + (page * kp->v_page_stride + slot * kp->v_slot_stride) * 2 + 2 * dl);
    } else {
      uint32_t hi_unused;
      widen_fp8x4<T, KVT>(*(const uint16_t*)(kbase + (page * kp->k_page_stride + slot * kp->k_slot_stride) + dl), kw, hi_unused);
      widen_fp8x4<T, KVT>(*(const uint16_t*)(vbase + (page * kp->v_page_stride + slot * kp->v_slot_stride) + dl), vw, hi_unused);
    }
    float sc = wave_sum(q0 * pw_lo<T>(kw) + q1 * pw_hi<T>(kw)) * scale2;
    if (cap > 0.0f) sc = cap2 - 2.0f * cap2 / (1.0f + __builtin_amdgcn_exp2f(sc * (2.0f / cap)));   // cap tanh(s / cap) in log2 units: cap2 (1 - 2 / (1 + e^(2 s / cap))), e^(2 s / cap) = 2^(2 sc / cap)
    sc += slope2 * (float)(j - ctx_len);
    const float mn = fmaxf(m, sc);
    const float alpha = __builtin_amdgcn_exp2f(m - mn), pj = __builtin_amdgcn_exp2f(sc - mn);
    const uint32_t pr = pw_pack<T>(pj, pj);                 // P is rounded to V's type before P.V (:508), not for the sum
    l = l * alpha + pj;
    a0 = a0 * alpha + pw_lo<T>(pr) * pw_lo<T>(vw);
    a1 = a1 * alpha + pw_lo<T>(pr) * pw_hi<T>(vw);
    m = mn;
  }
  const float inv = l > 0.0f ? v_sc / l : 0.0f;
  if (act) *(uint32_t*)(out_base + (int64_t)token * kp->p.out_stride_token + (int64_t)hq * kp->p.out_stride_head + dl) = pw_pack<T>(a0 * inv, a1 * inv);
  if (lse_base && lane == 0)
    lse_base[(int64_t)token * kp->p.lse_stride_token + hq] = l > 0.0f ? (m + __builtin_amdgcn_logf(l)) * 0.6931471805599453f : -INFINITY;
}

// Diagnostic build only (-DMI355_PW_STAMP, tools/pw_clock.py): shader-cycle sums of the tile loop's segments.
#ifdef MI355_PW_STAMP
#define PW_SEG_STAMP(idx)                                                                   \
  do {                                                                                      \
    unsigned long long st_now;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_now) :: "memory");        \
    st_sum[idx] += (unsigned)st_now - st_last;                                              \
    st_last = (unsigned)st_now;                                                             \
  } while (0)
#else
#define PW_SEG_STAMP(idx) do { } while (0)
#endif

// M16: both contractions on the 16x16x32 matrix instruction (the chip holds a higher clock under that shape,
// MI355X_MICROARCH.md DVFS give-back item 7). Lane (r16 = lane & 15, g4 = lane >> 4) then owns query rows
// 32 x + 16 rt + r16 (sub-block x, row tile rt) and, of a 16-key tile kt, the keys 16 kt + 4 g4 + r; register maps:
// K[kt][ks] kAK + 16 kt + 4 ks, Q'[x][rt][ks] kAQ + 32 x + 16 rt + 4 ks, O[x][rt][db] kAO + 64 x + 32 rt + 4 db.
// SW (M16 only): sliding window. A Q block's tile range starts at the window of its first token (the reference's 2D kernel
// only masks, :474-479; prefill_mfma_kernel tightens the same way), the tiles at the window's lower edge are general
// iterations with the lower bound in their mask, and the steady stretch lies between the two masked ends.
// KV8 (M16, plain, D = 128): the cache holds fp8 (1 = e4m3, 2 = e5m2; reference :434-455 dequantises on load). LDS-DMA cannot
// convert, so a tile's 16-key group travels as fp8 into a STAGING area of its wave (2 KiB per matrix, half the pieces), and one
// iteration later the same wave widens it - 16 fp8 per lane -> two 16-byte chunks, v_cvt_scalef32_pk_{bf16,f16}_{fp8,bf8}, exact -
// and writes it where the 16-bit LDS-DMA would have put it, swizzle included: rings, fragment reads and the schedule of the
// matrix instructions stay as they are. The cache's scales move out of the tiles: k_scale into Q', v_scale into 1 / l.
// The fetch side runs one tile further ahead (FO): iteration t requests K(t+4) / V(t+3) and converts K(t+3) / V(t+2).
template <typename T, bool M16, bool SW, bool SC = false, bool AL = false, int D = 128, int KV8 = 0>
__global__ __launch_bounds__(256, 1) void prefill_pw_kernel(const PwArgs a) {
  static_assert(KV8 == 0 || (M16 && !SW && !SC && !AL && D == 128), "fp8 cache: the plain 16x16x32 instantiation at head size 128");
  constexpr int EB = KV8 ? 1 : 2;              // bytes per cache element
  constexpr int FO = KV8 ? 1 : 0;              // tiles the fetch side runs further ahead
  using KVT = std::conditional_t<KV8 == 2, e5m2_t, std::conditional_t<KV8 == 1, e4m3_t, T>>;
  static_assert(M16 || !SW, "the sliding window is built into the 16x16x32 instantiation only");
  static_assert(M16 || !SC, "soft-cap is built into the 16x16x32 instantiation only");
  static_assert(!AL || (M16 && !SW && !SC), "ALiBi: the plain 16x16x32 instantiation only");
  // D = 64 / 96: the SAME geometry with partly empty rows - a key row still owns a 256-byte LDS row and sixteen chunk
  // positions, of which the swizzle fills eight / twelve with the row's 16-byte chunks (the other lanes of a row's LDS-DMA
  // re-read its last chunk: same address, coalesced, never read back); the last k-steps of every score chain, the last
  // output tiles and their fragment reads simply do not exist. Every instruction table, ring and seam stays as it is.
  static_assert(D == 128 || ((D == 64 || D == 80 || D == 96) && M16 && !SC && !AL && (!SW || D == 96)), "head sizes 64, 80 and 96: the plain 16x16x32 instantiation (96: or with a sliding window)");
  // k-steps of a score chain (D = 80: the third covers 16 of its 32 columns - lane groups 0 and 1 of the query rows' fragment,
  // the other two hold zeros and meet re-read, finite key chunks), 16-column output tiles, a row's last chunk
  constexpr int kTailG = (D % 32) / 8, kKS = D / 32 + (kTailG ? 1 : 0), kDB = D / 16, kCM = D / 8 - 1;
  using ops = pw_ops<bf16_t>;                  // the 32x32x16 form exists for bf16 only (its fixed reference 0 needs bf16's exponent range)
  using ops16 = pw_ops16<T>;
  constexpr int ROWB = 256;                    // bytes per key row (D = 128), 16 chunks of 16 B
  static_assert(M16 || __is_same(T, bf16_t), "P = 2^s with the reference 0 needs the exponent range of bf16");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  asm volatile("" ::: "a255");                 // the kernel owns all 256 accumulator registers
  const mi355_attn_params& p = a.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.group, BQ = a.block_q;
  const int qr = lane & 31, half = lane >> 5;
  // The per-item code (setup, query loads, epilogue) derives its lane-dependent values from an OPAQUE copy of the lane
  // index, refreshed per item: otherwise the compiler hoists those values (row -> token / head divisions, addresses)
  // out of the item loop, where they would have to stay in registers through the tile loop - which has none to spare
  // (it spills to accumulator registers, and those belong to the asm statements).
  // For the same reason that code reads the kernel arguments through an opaque pointer to the kernarg segment (scalar
  // loads, a few hundred cycles, once per item): what it needs of them is then not held in SGPRs through the tile loop.
  typedef const PwArgs __attribute__((address_space(4))) * KernArgs;
  struct SeamArgs {                    // read in one batch of scalar loads at the start of an item's set-up
    const uint16_t* q; uint16_t* out; float* lse;
    const int32_t *cu, *sk, *bt;
    int* tickets;
    int q_st, q_sh, out_st, out_sh;    // elements; the host admits [0, 2^22)
    int64_t lse_st, bt_stride, out_split_stride, lse_split_stride;
    int num_seqs, key_splits, num_qblocks, slots, G, BQ, g_inv, bq_shift, page_shift, skip_decodes, only_decodes, num_kv_heads;
    int window, non_causal;
  };
  int lane_o = lane;
  KernArgs kp = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
  SeamArgs sa;
  auto refresh_lane = [&]() __attribute__((always_inline)) {
    lane_o = lane;
    asm volatile("" : "+v"(lane_o));
    kp = (KernArgs)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    sa.q = (const uint16_t*)kp->p.q; sa.out = (uint16_t*)kp->p.out; sa.lse = kp->p.lse;
    sa.cu = kp->p.cu_seqlens_q; sa.sk = kp->p.seqused_k; sa.bt = kp->p.block_table; sa.tickets = kp->tickets;
    sa.q_st = (int)kp->p.q_stride_token; sa.q_sh = (int)kp->p.q_stride_head; sa.out_st = (int)kp->p.out_stride_token; sa.out_sh = (int)kp->p.out_stride_head;
    sa.lse_st = kp->p.lse_stride_token; sa.bt_stride = kp->p.block_table_stride;
    sa.out_split_stride = kp->out_split_stride; sa.lse_split_stride = kp->lse_split_stride;
    sa.num_seqs = kp->p.num_seqs; sa.key_splits = kp->key_splits; sa.num_qblocks = kp->num_qblocks; sa.slots = kp->slots;
    sa.G = kp->group; sa.BQ = kp->block_q; sa.g_inv = kp->g_inv; sa.bq_shift = kp->bq_shift; sa.page_shift = kp->page_shift;
    sa.skip_decodes = kp->p.skip_decodes; sa.only_decodes = kp->p.only_decodes; sa.num_kv_heads = kp->p.num_kv_heads;
    sa.window = SW ? kp->p.sliding_window : 0; sa.non_causal = kp->p.non_causal;
  };
  refresh_lane();
  // x / G, x % G, x / BQ for x >= 0: shifts when the divisor is a power of two (every GQA ratio in use), else the division
  // x / G and x % G for a row index 0 <= x < 256 (G <= 256): one 24-bit multiply by ceil(2^16 / G) and a shift, branch
  // free. (As "shift if G is a power of two, else divide" hipcc built a diamond of scalar branches around every call:
  // the mask of a diagonal tile alone was ~230 instructions and two dozen branches per sub-block.)
  auto div_g = [&](int x) { return (int)(__umul24((unsigned)x, (unsigned)sa.g_inv) >> 16); };
  auto mod_g = [&](int x) { return x - (int)__umul24((unsigned)div_g(x), (unsigned)sa.G); };
  auto div_bq = [&](int x) { return sa.bq_shift >= 0 ? (x >> sa.bq_shift) : x / sa.BQ; };

  // ---- work items ---------------------------------------------------------------------------------------
  // The grid is `slots` workgroups per KV head (KV head fastest: one head per XCD at Hk = 8, whose L2 then serves
  // that head's K/V to all its Q blocks), about one per CU, and a workgroup walks several items - (Q block, key
  // split) pairs of its head, heaviest Q blocks first, dealt boustrophedon (round j hands item j S + s to slot s
  // for even j, item j S + S-1-s for odd j: with causal weights every slot's total is the same). Walking them in
  // ONE workgroup is what lets the next item's query rows and first K/V tiles load while this item's output is
  // normalised and stored - a fresh workgroup pays that chain of round trips (~5 us) with its CU idle.
  struct Item {
    int seq, q_start, q_len, seq_len, ctx_len, tok0, ksplit, tile_lo, tile_hi, last_group, w_tok_lo, w_tok_hi, rank;
    uint64_t bt64;                     // this sequence's block-table row
    uint16_t* out_base;
    float* lse_base;
  };
  const int head = (int)(blockIdx.x % p.num_kv_heads);
  const int slot = (int)(blockIdx.x / p.num_kv_heads);
  const char* const kbase = (const char*)p.k_cache + (int64_t)head * p.k_stride_head * EB;
  const char* const vbase = (const char*)p.v_cache + (int64_t)head * p.v_stride_head * EB;

  // Sequence of a Q block and its lengths in ONE memory round trip for batches of up to 63 sequences: lane i takes
  // cu_seqlens_q[i] and seqused_k[i], the search (largest i with cu[i] / BQ + i <= qblock, find_seq_idx :32-52) is a
  // ballot. With a single sequence the first block-table entries ride the same trip, speculatively (checked when the
  // first tiles are requested).
  int spec_pg[4] = {0, 0, 0, 0}, spec_off[4] = {0, 0, 0, 0};
  // ... and that trip is taken ONCE per workgroup: the two words stay in a register each across the tile loops (opaque
  // to the compiler, so it neither reloads nor rematerialises them), and an item's set-up is scalar arithmetic on a
  // ballot - 1.1 us of exposed latency per item until round 3.
  // (not in the sliding-window instantiation: its general iterations leave no two registers to carry them)
  constexpr bool kResidentTabs = false;
  int cu_tab = 0, sk_tab = 0;
  if (kResidentTabs && p.num_seqs <= 63) {
    cu_tab = p.cu_seqlens_q[min(lane, p.num_seqs)];
    sk_tab = p.seqused_k[min(lane, p.num_seqs - 1)];
  }
  pw_launder(cu_tab);
  pw_launder(sk_tab);
  // one sequence (every instantiation): its three words as scalars
  int one_q_start = 0, one_q_end = 0, one_seq_len = 0;
  if (!SW && p.num_seqs == 1) {
    one_q_start = __builtin_amdgcn_readfirstlane(p.cu_seqlens_q[0]);
    one_q_end = __builtin_amdgcn_readfirstlane(p.cu_seqlens_q[1]);
    one_seq_len = __builtin_amdgcn_readfirstlane(p.seqused_k[0]);
  }
  asm volatile("" : "+s"(one_q_start), "+s"(one_q_end), "+s"(one_seq_len));
  auto setup_idx = [&](Item& I, int idx) -> bool {       // item idx of this KV head's list; false: it is empty
    {
      const int BQ = sa.BQ;
      const int qb_rank = sa.key_splits == 1 ? idx : idx / sa.key_splits;
      const int qblock = sa.num_qblocks - 1 - qb_rank;      // heaviest first
      I.ksplit = idx - qb_rank * sa.key_splits;
      I.rank = idx * sa.num_kv_heads + head;
      const bool spec_bt = sa.num_seqs == 1 && sa.key_splits == 1;
      if (spec_bt) {
        uint64_t b0 = (uint64_t)sa.bt;
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b0), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b0 >> 32));
        b0 = ((uint64_t)hi << 32) | lo;
        // (ordinary scalar loads - the block table is not written during the launch: the compiler may park SGPRs in
        // vector lanes around here, and it would park the destination of an asm load before the entry has landed)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          spec_off[i] = __builtin_amdgcn_readfirstlane(min(((i * 4 + wave) << 4) >> sa.page_shift, (int)sa.bt_stride - 1) << 2);
          spec_pg[i] = *(const __attribute__((address_space(4))) int*)(b0 + (uint32_t)spec_off[i]);
        }
      } else {       // (defined on every path: a value carried from the previous item would have to live through its tile loop)
#pragma unroll
        for (int i = 0; i < 4; ++i) { spec_off[i] = 0; spec_pg[i] = 0; }
      }
      int seq, q_start, q_len, seq_len;
      const int qr_o = lane_o & 31;
      // (not in the sliding-window instantiation: three more scalars held across its tile loops cost it 4 %)
      if (!SW && sa.num_seqs == 1) {
        seq = qblock >= 0 ? 0 : -1;
        q_start = one_q_start; q_len = one_q_end - one_q_start; seq_len = one_seq_len;
      } else if (sa.num_seqs <= 63) {
        const int cu_v = kResidentTabs ? cu_tab : sa.cu[min(lane_o, sa.num_seqs)];
        const int sk_v = kResidentTabs ? sk_tab : sa.sk[min(lane_o, sa.num_seqs - 1)];
        const unsigned long long le = __ballot(lane_o < sa.num_seqs && div_bq(cu_v) + lane_o <= qblock);
        seq = __builtin_popcountll(le) - 1;
        const int sq = max(seq, 0);
        q_start = __builtin_amdgcn_readlane(cu_v, sq);
        q_len = __builtin_amdgcn_readlane(cu_v, sq + 1) - q_start;
        seq_len = __builtin_amdgcn_readlane(sk_v, sq);
      } else {
        seq = pw_find_seq(sa.cu, sa.num_seqs, qblock, BQ);
        const int sq = max(seq, 0);
        q_start = sa.cu[sq];
        q_len = sa.cu[sq + 1] - q_start;
        seq_len = sa.sk[sq];
      }
      if (seq < 0) return false;
      const int qb_local = qblock - (div_bq(q_start) + seq);
      if (qb_local * BQ >= q_len || q_len <= sa.skip_decodes || (sa.only_decodes && q_len > sa.only_decodes)) return false;
      I.seq = seq; I.q_start = q_start; I.q_len = q_len; I.seq_len = seq_len;
      // non-causal (prefill_flash_attention(causal=False)): every row sees all seq_len keys - which is what every
      // visibility formula below gives for a context that already covers the whole sequence
      I.ctx_len = sa.non_causal ? seq_len : seq_len - q_len;
      I.tok0 = qb_local * BQ;
      I.w_tok_lo = I.tok0 + div_g(wave * 64);
      I.w_tok_hi = min(I.tok0 + div_g(wave * 64 + 63), q_len - 1);
      const int wg_tok_hi = min(I.tok0 + BQ - 1, q_len - 1);
      const int n_keys_wg = max(0, min(I.ctx_len + wg_tok_hi + 1, seq_len));
      I.last_group = (max(n_keys_wg, 1) - 1) >> 4;                 // last 16-key group this Q block can see
      I.tile_lo = 0;
      I.tile_hi = (n_keys_wg + kPwTile - 1) / kPwTile;
      if (SW && sa.window > 0) I.tile_lo = min(max(I.ctx_len + I.tok0 - sa.window + 1, 0) >> 6, I.tile_hi);   // first key the block's first token sees
      if (sa.key_splits > 1) {                    // key-split launch: an even share of this Q block's tiles
        const int tps = (I.tile_hi - I.tile_lo + sa.key_splits - 1) / sa.key_splits;
        I.tile_lo = min(I.tile_lo + I.ksplit * tps, I.tile_hi);
        I.tile_hi = min(I.tile_hi, I.tile_lo + tps);
      }
      I.out_base = sa.out + (int64_t)I.ksplit * sa.out_split_stride;
      I.lse_base = sa.lse ? sa.lse + (int64_t)I.ksplit * sa.lse_split_stride : nullptr;
      {
        const uint64_t b = (uint64_t)(sa.bt + (int64_t)seq * sa.bt_stride);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
        I.bt64 = ((uint64_t)hi << 32) | lo;
      }
      return true;
    }
  };

  // ---- which item next -----------------------------------------------------------------------------------
  // Static deal (no ticket counters): round j hands item j S + s to slot s for even j, j S + S-1-s for odd j.
  // Dynamic deal: the first item of slot s is item s; every further one is S + a ticket drawn from a per-head counter in
  // the caller's workspace (relaxed agent-scope fetch_add by wave 0, lane 0). The ticket is drawn LATE - when the item's
  // steady tiles are done and only its last few remain (a draw at the start of an item hands out the second items at
  // time zero in arrival order, i.e. a random static deal: 146 us instead of 119 at 1 x 4096) - early enough to be back
  // at the seam, and handed to the other waves through two alternating LDS words behind the barrier the seam has
  // anyway. CUs that run slower (the clock differs by up to 15 % between them under the power limit) simply take fewer
  // items. Every workgroup ends by drawing one ticket past the list; the last one to finish zeroes the counters.
  const bool dynamic = sa.tickets != nullptr;
  int round = 0;               // static deal
  int ticket_v = 0;            // dynamic deal: wave 0 / lane 0's draw, in flight or landed
  int attempt = 0;             // publications so far (workgroup-uniform)
  volatile int* const lds_idx = (volatile int*)(smem + kLdsT);
  auto draw = [&]() __attribute__((always_inline)) {
    if (wave == 0 && lane_o == 0) ticket_v = __hip_atomic_fetch_add(sa.tickets + head, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto publish = [&]() __attribute__((always_inline)) {        // (the compiler waits for the draw here: it landed long ago)
    if (wave == 0 && lane_o == 0) lds_idx[attempt & 1] = sa.slots + ticket_v;
  };
  auto read_published = [&]() __attribute__((always_inline)) {
    const int v = lds_idx[attempt & 1];
    ++attempt;
    return __builtin_amdgcn_readfirstlane(v);
  };
  // the next non-empty item; `idx` = an index already in hand (the slot's first item, or one read behind the seam's
  // barrier) when have_idx. false: the list is exhausted
  auto acquire = [&](Item& I, int idx, bool have_idx) -> bool {
    const int items_per_head = sa.num_qblocks * sa.key_splits;
#pragma unroll
    for (int i = 0; i < 4; ++i) { spec_off[i] = 0; spec_pg[i] = 0; }     // (dead on the "list exhausted" path too: see setup_idx)
    while (true) {
      if (!have_idx) {
        if (dynamic) {             // (an empty item: the next ticket is needed at once)
          draw();
          publish();
          __syncthreads();
          idx = read_published();
        } else {
          idx = round * sa.slots + ((round & 1) ? sa.slots - 1 - slot : slot);
          ++round;
        }
      }
      have_idx = false;
      if (idx >= items_per_head) return false;
      if (setup_idx(I, idx)) return true;
    }
  };

  // this lane's query row of sub-block sb (recomputed where it is needed: nothing of it lives through the tile loop)
  // last visible key of this lane's query row in sub-block sb (two 32-row sub-blocks; M16: sb = 2 x + rt, four 16-row
  // tiles), -1 = a padding row. Recomputed where it is needed (Q conversion, the mask, the output) rather than kept in
  // registers across the tile loop.
  auto row_lim = [&](const Item& I, int sb) __attribute__((always_inline)) {
    int m_row = M16 ? wave * 64 + sb * 16 + (lane_o & 15) : wave * 64 + sb * 32 + (lane_o & 31);
    asm volatile("" : "+v"(m_row));   // (the quotient below is three instructions: not worth a register held across the item loop)
    const int tok = I.tok0 + div_g(m_row);
    return ((m_row < sa.BQ * sa.G) && (tok < I.q_len)) ? min(I.ctx_len + tok, I.seq_len - 1) : -1;
  };
  // first visible key of the same row under a sliding window (keys j with position - j < window)
  auto row_lo = [&](const Item& I, int sb) __attribute__((always_inline)) {
    int m_row = wave * 64 + sb * 16 + (lane_o & 15);
    asm volatile("" : "+v"(m_row));
    return I.ctx_len + I.tok0 + div_g(m_row) - sa.window + 1;
  };
  auto row_of = [&](const Item& I, int sb, int& tok_local, int& hq) __attribute__((always_inline)) {
    int m_row = M16 ? wave * 64 + sb * 16 + (lane_o & 15) : wave * 64 + sb * 32 + (lane_o & 31);   // M16: sb = 2 x + rt
    asm volatile("" : "+v"(m_row));
    tok_local = I.tok0 + div_g(m_row);
    hq = head * sa.G + mod_g(m_row);
    return (m_row < sa.BQ * sa.G) && (tok_local < I.q_len);
  };
  // a 64-bit address that is the same for the whole wave, in scalar registers
  auto uniform64 = [&](uint64_t b) {
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return ((uint64_t)hi << 32) | lo;
  };

  // ---- Q rows -> the Q accumulator registers, raw (padding rows read the sequence's last query row and are zeroed on
  // conversion): scalar base, 32-bit lane offset (the host admits strides below 2^22 elements), the destination a
  // literal a[..] range of the asm statement. The previous item's Q' is dead by now (the seam sits behind its last matrix
  // instruction), so the rows land where their scaled form will live - no VGPR carries them across the epilogue (64 of
  // them did until round 3, the peak of the kernel's VGPR demand). Retired by the s_waitcnt vmcnt(0) of
  // zero_o_and_convert_q; vmcnt is in order, so the compiler's own waits only ever over-wait for these.
  // `part` (M16): -1 = all four row tiles, else only row tile `part` - at the seam the loads of the next item are dealt
  // over the four row tiles of the current item's output (see epilogue): 36 load instructions per wave take the CU's
  // address unit 2.4 us when the four waves issue them back to back, time the output's VALU / LDS work can run beside.
  auto issue_q = [&](const Item& I, int part) __attribute__((always_inline)) {
    // scalar address of the block's first token and this KV head's first query head; a lane's row is within 2^31 bytes of it
    const uint64_t qb = uniform64((uint64_t)(sa.q + (int64_t)(I.q_start + I.tok0) * (int64_t)sa.q_st + (int64_t)(head * sa.G) * (int64_t)sa.q_sh));
    if constexpr (M16) {        // a[kAQ + 16 rt4 + 4 ks ..] = Q[row 16 rt4 + r16][32 ks + 8 g4 .. + 7]
      sfor<4>([&](auto RT) {
        constexpr int rt4 = decltype(RT)::value;
        if (part >= 0 && part != rt4) return;
        int tok_local, hq;
        row_of(I, rt4, tok_local, hq);
        const uint32_t off = (uint32_t)(((min(tok_local, I.q_len - 1) - I.tok0) * sa.q_st + (hq - head * sa.G) * sa.q_sh + 8 * (lane_o >> 4)) * 2);
        // (a partial last k-step: the lane groups past the row's end re-read its last piece; their registers are zeroed on conversion)
        const uint32_t off_tail = kTailG ? off - 16u * (uint32_t)max((lane_o >> 4) - (kTailG - 1), 0) : off;
        sfor<kKS>([&](auto KS) { constexpr int ks = decltype(KS)::value; pw_gload16_acc<kAQ + 16 * rt4 + 4 * ks, 64 * ks>((kTailG && ks == kKS - 1) ? off_tail : off, qb); });
      });
    } else if (part <= 0)
    sfor<2>([&](auto SB) {
      constexpr int sb = decltype(SB)::value;
      int tok_local, hq;
      row_of(I, sb, tok_local, hq);
      const uint32_t off = (uint32_t)(((min(tok_local, I.q_len - 1) - I.tok0) * sa.q_st + (hq - head * sa.G) * sa.q_sh + 8 * (lane_o >> 5)) * 2);
      sfor<8>([&](auto KS) { constexpr int ks = decltype(KS)::value; pw_gload16_acc<kAQ + 32 * sb + 4 * ks, 32 * ks>(off, qb); });
    });
  };

  // ---- LDS-DMA constants: wave w stages key rows 16 w .. 16 w + 15 of every tile, four rows per instruction -
  const int page_mask = p.page_size - 1;
  const uint32_t ksb = a.k_slot_stride * EB, vsb = a.v_slot_stride * EB;   // bytes between key rows of a page
  const uint32_t kpb = a.k_page_stride * EB, vpb = a.v_page_stride * EB;   // bytes between pages
  // LDS row R = 4 i + r4 of the group, chunk position c16 holds logical chunk c16 ^ f(R) (swizzle on the source side);
  // rows past the sequence (R > maxr, last group only) re-read row maxr: finite data under a zero probability.
  // (The swizzle terms are recomputed from an opaque lane index at each - rare - call, not kept in eight registers.)
  uint32_t koff[4], voff[4];
  auto set_k_offsets = [&](int maxr) {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int r4 = lo >> 4, c16 = lo & 15;
    if constexpr (KV8) {       // fp8: a key row is 128 bytes, a piece covers eight rows, unswizzled (the swizzle is applied by the widening write)
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int R = 8 * i + (lo >> 3); koff[i] = (uint32_t)(min(R, maxr) * (int)ksb + ((lo & 7) << 4)); }
      koff[2] = koff[3] = 0;
      return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int R = 4 * i + r4; koff[i] = (uint32_t)(min(R, maxr) * (int)ksb + (min(c16 ^ R, kCM) << 4)); }
  };
  auto set_v_offsets = [&](int maxr) {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int r4 = lo >> 4, c16 = lo & 15;
    if constexpr (KV8) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { const int R = 8 * i + (lo >> 3); voff[i] = (uint32_t)(min(R, maxr) * (int)vsb + ((lo & 7) << 4)); }
      voff[2] = voff[3] = 0;
      return;
    }
#pragma unroll
    // V's chunk swizzle follows the transposed read of the instantiation: half a wave of it covers rows 0..3 x four chunks
    // (32x32x16 form: f = 4 (R & 3) | (R >> 2) & 3 keeps the rows apart) or rows 0..7 x two chunks (16x16x32 form: f = 2 (R & 7);
    // the other form's f there puts rows r and r + 4 on the same banks: 34 % of LDS-active cycles were conflicts)
    for (int i = 0; i < 4; ++i) { const int R = 4 * i + r4; const int f = M16 ? 2 * (R & 7) : (((R & 3) << 2) | ((R >> 2) & 3)); voff[i] = (uint32_t)(min(R, maxr) * (int)vsb + (min(c16 ^ f, kCM) << 4)); }
  };
  set_k_offsets(15);
  set_v_offsets(15);
  bool k_tail = false, v_tail = false;           // offsets already clamped for the sequence's last, partial group
  const uint32_t lds_wave = (uint32_t)(wave * 4096);
  // 16-key group of wave w in `tile` (tiles past the share repeat its last tile: never read, keeps the DMA count
  // per iteration fixed), its first key, the byte offset of its block-table entry
  auto group_key0 = [&](const Item& I, int tile) { return min(min(tile, I.tile_hi - 1) * 4 + wave, I.last_group) << 4; };
  auto entry_off = [&](const Item& I, int tile) { return __builtin_amdgcn_readfirstlane((group_key0(I, tile) >> a.page_shift) << 2); };
  // block-table entry -> 64-bit address of the group's first key row
  auto group_base = [&](const Item& I, int tile, int page, auto ISV) {
    constexpr bool isv = decltype(ISV)::value != 0;
    const uint32_t slot0 = (uint32_t)(group_key0(I, tile) & page_mask);
    const uint64_t b = (uint64_t)(isv ? vbase : kbase) + (uint64_t)(uint32_t)page * (isv ? vpb : kpb) + (uint64_t)slot0 * (isv ? vsb : ksb);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return ((uint64_t)hi << 32) | lo;
  };
  // the same for a tile whose group is known to lie inside the share and the sequence (steady iterations below):
  // no clamps, the in-page offset in 32 bits
  const int wave16 = wave * 16;
  auto entry_off_fast = [&](int tile) { return __builtin_amdgcn_readfirstlane(((tile * kPwTile + wave16) >> a.page_shift) << 2); };
  auto group_base_fast = [&](const Item&, int tile, int page, auto ISV) {
    constexpr bool isv = decltype(ISV)::value != 0;
    const uint32_t in_page = (uint32_t)((tile * kPwTile + wave16) & page_mask) * (isv ? vsb : ksb);
    const uint64_t b = (uint64_t)(isv ? vbase : kbase) + (uint64_t)(uint32_t)page * (isv ? vpb : kpb) + in_page;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)b), hi = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32));
    return ((uint64_t)hi << 32) | lo;
  };
  // the sequence ends inside this group (only ever its last one, and every later fetch repeats it): clamp the rows once
  auto tail_check = [&](const Item& I, int tile, auto ISV) {
    constexpr bool isv = decltype(ISV)::value != 0;
    const int maxr = I.seq_len - 1 - group_key0(I, tile);
    if (__builtin_expect(maxr < 15 && !(isv ? v_tail : k_tail), 0)) {
      if (isv) { set_v_offsets(maxr); v_tail = true; } else { set_k_offsets(maxr); k_tail = true; }
    }
  };

  // ---- per-lane LDS read addresses (swizzle folded in) ----------------------------------------------
  // K fragment ks of 32-key block kb: row 32 kb + qr, logical chunk 2 ks + half
  const int r16 = lane & 15, g4 = lane >> 4;
  uint32_t k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = (uint32_t)(kLdsK + qr * ROWB + (((2 * ks + half) ^ (qr & 15)) << 4));
  // M16: K fragment (kt, ks) = rows 16 kt + r16, logical chunk 4 ks + g4 (+ 4096 kt as an immediate)
  uint32_t k_rd16[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd16[ks] = (uint32_t)(kLdsK + r16 * ROWB + (((4 * ks + g4) ^ r16) << 4));
  // V transposed read of k-step sk (16 keys), output block b: row 16 sk + 4 half + q4 (+8), logical byte column
  // 64 b + 32 g1 + 8 pp -> chunk 4 b + 2 g1 + (pp >> 1), sub-offset 8 (pp & 1)
  const int gq1 = (lane >> 4) & 1, li = lane & 15, q4 = li >> 2, pp = li & 3;
  uint32_t v_rd0[4], v_rd1[4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int lc = 4 * b + 2 * gq1 + (pp >> 1);
    const int r0 = 4 * half + q4, r1 = r0 + 8;
    const int f0 = ((r0 & 3) << 2) | ((r0 >> 2) & 3), f1 = ((r1 & 3) << 2) | ((r1 >> 2) & 3);
    v_rd0[b] = (uint32_t)(kLdsV + r0 * ROWB + ((lc ^ f0) << 4) + 8 * (pp & 1));
    v_rd1[b] = (uint32_t)(kLdsV + r1 * ROWB + ((lc ^ f1) << 4) + 8 * (pp & 1));
  }
  // M16: V fragment (db, c) = d tile db (16 columns), keys 32 c + 16 (j >> 2) + 4 g4 + (j & 3): two transposed reads of the
  // 4-row x 16-column blocks at rows 32 c + 4 g4 (and + 16); lane 4 q + p of the 16-lane group addresses row q, columns 4 p..
  uint32_t v_rd16[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) {
    const int row = 4 * g4 + q4, f = 2 * (row & 7);
    v_rd16[db] = (uint32_t)(kLdsV + row * ROWB + (((2 * db + (pp >> 1)) ^ f) << 4) + 8 * (pp & 1));
  }

  // ---- fp8 cache (KV8): where this lane's 16 staged bytes are, and where their two widened chunks go ------------------
  // Piece i of a group = key rows 8 i + (lane >> 3), lane & 7 = the row's 16-byte piece c8 (head dims 16 c8 .. + 15), landing at
  // 1024 i + 16 lane of the destination. Widened, they are the row's logical chunks 2 c8 and 2 c8 + 1, at chunk positions
  // chunk ^ f(row) of the row's 256 bytes in the slot (f: the swizzles of set_k_offsets / set_v_offsets). V's f is even: its
  // second chunk sits 16 bytes behind the first, piece 1 exactly 2048 bytes behind piece 0.
  uint32_t st_rd = 0, kw8[2][2] = {{0, 0}, {0, 0}}, vw8 = 0;
  if constexpr (KV8) {
    const int R0 = lane >> 3, c8 = lane & 7;
    st_rd = (uint32_t)(kLdsS + wave * 4096 + lane * 16);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int e = 0; e < 2; ++e) kw8[i][e] = (uint32_t)(kLdsK + wave * 4096 + (8 * i + R0) * ROWB + (((2 * c8 + e) ^ (8 * i + R0)) << 4));
    vw8 = (uint32_t)(kLdsV + wave * 4096 + R0 * ROWB + (((2 * c8) ^ (2 * (R0 & 7))) << 4));
  }
  // 16 fp8 of one key row -> the row's two chunks
  auto widen16 = [&](const wu32x4_t& in, wu32x4_t& c0, wu32x4_t& c1) __attribute__((always_inline)) {
    uint32_t o[8];
#pragma unroll
    for (int w = 0; w < 4; ++w) pw_widen4<T, KVT>(in[w], o[2 * w], o[2 * w + 1]);
    c0 = wu32x4_t{o[0], o[1], o[2], o[3]};
    c1 = wu32x4_t{o[4], o[5], o[6], o[7]};
  };
  // a staged or freshly landed group (two pieces read from `src` + 1024 i) widened into ring slot byte offset SLOT (compile time)
  auto widen_group = [&](auto ISV, auto SLOT, const wu32x4_t& p0, const wu32x4_t& p1) __attribute__((always_inline)) {
    constexpr bool isv = decltype(ISV)::value != 0;
    constexpr int slot = decltype(SLOT)::value;
    wu32x4_t c0, c1;
    widen16(p0, c0, c1);
    if constexpr (isv) { pw_lds_write128<slot>(vw8, c0); pw_lds_write128<slot + 16>(vw8, c1); }
    else { pw_lds_write128<slot>(kw8[0][0], c0); pw_lds_write128<slot>(kw8[0][1], c1); }
    widen16(p1, c0, c1);
    if constexpr (isv) { pw_lds_write128<slot + 2048>(vw8, c0); pw_lds_write128<slot + 2048 + 16>(vw8, c1); }
    else { pw_lds_write128<slot>(kw8[1][0], c0); pw_lds_write128<slot>(kw8[1][1], c1); }
  };

  // The same in twelve steps of one statement each, for the tile loop (one step per gap): per piece, two words widened, the
  // first chunk written, two words widened, the second chunk written. wq: the chunk under construction.
  uint32_t wq0 = 0, wq1 = 0, wq2 = 0, wq3 = 0;
  auto widen_step = [&](auto ISV, auto SLOT, auto KC, const wu32x4_t& p0, const wu32x4_t& p1) __attribute__((always_inline)) {
    constexpr bool isv = decltype(ISV)::value != 0;
    constexpr int slot = decltype(SLOT)::value, k = decltype(KC)::value, i = k / 6, r = k % 6;
    const wu32x4_t& in = i ? p1 : p0;
    if constexpr (r == 0) pw_widen4<T, KVT>(in[0], wq0, wq1);
    else if constexpr (r == 1) pw_widen4<T, KVT>(in[1], wq2, wq3);
    else if constexpr (r == 3) pw_widen4<T, KVT>(in[2], wq0, wq1);
    else if constexpr (r == 4) pw_widen4<T, KVT>(in[3], wq2, wq3);
    else if constexpr (isv) pw_lds_write128<slot + 2048 * i + (r == 5 ? 16 : 0)>(vw8, wu32x4_t{wq0, wq1, wq2, wq3});
    else pw_lds_write128<slot>(kw8[i][r == 5 ? 1 : 0], wu32x4_t{wq0, wq1, wq2, wq3});
  };

  // ---- an item's first tiles on their way: K0 K1 V0 | K2 V1 (twenty pieces) ------------------------------
  int pg_k = 0, pg_v = 0;                        // block-table entries of K(t+3) / V(t+2) for the coming iteration (V's = K's of one iteration earlier)
  auto issue_first_tiles = [&](const Item& I) __attribute__((always_inline)) {
    if (k_tail) { set_k_offsets(15); k_tail = false; }     // the previous item may have ended inside a group
    if (v_tail) { set_v_offsets(15); v_tail = false; }
    if (I.tile_hi <= I.tile_lo) return;
    int pk0, pk1, pk2, pv0, pv1;
    const int eo0 = entry_off(I, I.tile_lo), eo1 = entry_off(I, I.tile_lo + 1), eo2 = entry_off(I, I.tile_lo + 2), eo3 = entry_off(I, I.tile_lo + 3);
    if (sa.num_seqs == 1 && sa.key_splits == 1 && eo0 == spec_off[0] && eo1 == spec_off[1] && eo2 == spec_off[2] && eo3 == spec_off[3]) {
      pk0 = spec_pg[0]; pk1 = spec_pg[1]; pk2 = spec_pg[2]; pg_k = spec_pg[3];     // the speculative entries are the right ones
    } else {
      scalar_load4((const int32_t*)I.bt64, eo0 >> 2, eo1 >> 2, eo2 >> 2, eo3 >> 2, pk0, pk1, pk2, pg_k);
    }
    pv0 = pk0; pv1 = pk1; pg_v = pk2;            // K and V share the block table
    auto group = [&](int tile, int page, auto ISV, uint32_t lds_dst) {
      constexpr bool isv = decltype(ISV)::value != 0;
      tail_check(I, tile, ISV);
      const uint64_t base = group_base(I, tile, page, ISV);
#pragma unroll
      for (int i = 0; i < (KV8 ? 2 : 4); ++i) pw_glds16(isv ? voff[i] : koff[i], base, lds_dst + lds_wave + i * 1024);
    };
    group(I.tile_lo, pk0, ic<0>{}, kLdsK);
    group(I.tile_lo + 1, pk1, ic<0>{}, kLdsK + kSlotBytes);
    group(I.tile_lo, pv0, ic<1>{}, kLdsV);
    group(I.tile_lo + 2, pk2, ic<0>{}, kLdsK + 2 * kSlotBytes);
    group(I.tile_lo + 1, pv1, ic<1>{}, kLdsV + kSlotBytes);
    if constexpr (KV8) {
      // fp8: the five groups above landed as fp8 in the first half of their wave's region of their slot and are widened in place
      // (widen_first_tiles, once the loads are back); K3 and V2 go to the staging area, where iteration 0 finds them, and the
      // table entries run one tile further ahead
      const int pk3 = pg_k;
      pg_k = *(const __attribute__((address_space(4))) int*)(I.bt64 + (uint32_t)entry_off(I, I.tile_lo + 4));   // (an ordinary load: see setup)
      group(I.tile_lo + 3, pk3, ic<0>{}, kLdsS);
      group(I.tile_lo + 2, pk2, ic<1>{}, kLdsS + 2048);
      pg_v = pk3;
    }
  };
  // fp8: the first five groups, landed (the caller waited for them) as fp8, widened in place: both pieces into registers, then
  // the four chunks over the same 4 KiB of this wave (LDS serves a wave's accesses in order)
  auto widen_first_tiles = [&]() __attribute__((always_inline)) {
    if constexpr (KV8) {
      const uint32_t ip = st_rd - (uint32_t)kLdsS;        // wave * 4096 + lane * 16
      const uint32_t ipv = ip + (uint32_t)kLdsV;          // (the V ring lies beyond what an instruction's offset field reaches)
      auto one = [&](auto ISV, auto SLOT) __attribute__((always_inline)) {
        constexpr int base = decltype(SLOT)::value;
        const uint32_t src = decltype(ISV)::value ? ipv : ip + (uint32_t)kLdsK;
        wu32x4_t p0 = pw_lds_read128<base>(src), p1 = pw_lds_read128<base + 1024>(src);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p0), "+v"(p1) :: "memory");
        widen_group(ISV, SLOT, p0, p1);
      };
      one(ic<0>{}, ic<0>{});
      one(ic<0>{}, ic<kSlotBytes>{});
      one(ic<1>{}, ic<0>{});
      one(ic<0>{}, ic<2 * kSlotBytes>{});
      one(ic<1>{}, ic<kSlotBytes>{});
    }
  };
  // The next item's loads, hung in front of the output's first row tile (hook 0 of epilogue). Dealt over the output's four
  // row tiles they left the address unit to the output's own work - the seam's stamps shrank by 2.4 us - and the launch
  // did not get faster (1 x 4096) or got slower (sliding window, 77 -> 82 us): whatever is issued after the output has
  // begun is still in flight when the next item's tile loop starts, and the wait moves into its first iterations, where
  // all four waves share it at the barrier. (profiles/r03/pw_seam.log)
  auto next_item_loads = [&](const Item& I, auto PART) __attribute__((always_inline)) {
    constexpr int part = decltype(PART)::value;
    if constexpr (part == 0) {
      issue_q(I, -1);
      issue_first_tiles(I);
    }
  };

  // ---- O = 0 and Q' = Q * scale * log2(e), packed, in place in the accumulator registers the raw rows landed in ---
  // (SC: the scores come out as u = s * 2 log2(e) / cap, see a_softcap_exp_ho)
  const float scale2_plain = SC ? p.scale * (2.0f * kPwLog2e) / p.softcap : p.scale * kPwLog2e;
  auto zero_o_and_convert_q = [&](const Item& I, bool zero_o) __attribute__((always_inline)) {
    if (zero_o) sfor<128>([&](auto IC) { acc_zero<kAO + decltype(IC)::value>(); });
    // the item's query rows have landed (and everything older: its first tiles, the previous item's output)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // (fp8 cache: K is widened unscaled, its scale rides in Q' - read per item, a scalar load, rather than held across the loop)
    const float scale2 = KV8 ? scale2_plain * ((kp->p.k_scale != nullptr) ? kp->p.k_scale[0] : 1.0f) : scale2_plain;
    // One row's registers share one scale, 0 for a padding row (its raw words are the sequence's last query row: x 0
    // zeroes them, and were that row not finite, the NaN stays in a row that is never stored). Straight-line: with
    // "valid ? read : 0" per register hipcc wrapped every one of the 64 reads in an exec-mask branch - 940 instructions,
    // ~1.8 us of every seam at one wave per SIMD; 6 per register now.
    sfor<2>([&](auto SB) {
      sfor<(M16 ? 2 : 1)>([&](auto RT) {
        // (M16: registers kAQ + 32 x + 16 rt + 4 ks are Q'[x][rt][ks])
        constexpr int sb = decltype(SB)::value, rt = decltype(RT)::value, n = M16 ? 16 : 32;
        const float sc = row_lim(I, M16 ? 2 * sb + rt : sb) >= 0 ? scale2 : 0.0f;
        const float sc_tail = (kTailG && (lane_o >> 4) >= kTailG) ? 0.0f : sc;          // (D = 80: columns 80 .. 95 of the third k-step do not exist)
        sfor<(M16 ? 4 * kKS : n)>([&](auto E) {
          constexpr int idx = kAQ + 32 * sb + n * rt + decltype(E)::value;
          const uint32_t v = acc_read_u32<idx>();
          const float sce = (M16 && kTailG && decltype(E)::value / 4 == kKS - 1) ? sc_tail : sc;
          acc_write<idx>(pw_pack<T>(pw_lo<T>(v) * sce, pw_hi<T>(v) * sce));
        });
      });
    });
  };

  // ---- state -------------------------------------------------------------------------------------------
  // HAND-OWNED REGISTERS (32x32x16 instantiation). hipcc pads one s_nop between two asm statements when the second reads
  // a register the first wrote and none of its own instructions sits between them - and in the tile loop every
  // instruction is an asm statement and every one costs this wave four issue cycles (tools/probes/issue_model.hip):
  // ~45 pads per tile, 6 % of it. So the tile loop's statements name S, the exponentials' ring, the row sums and P
  // as INPUTS and write them all the same: to the compiler these are values that an empty asm statement defined once
  // per work item (pw_launder: nothing to rematerialise, nothing known about the content) and that are only ever
  // read, so each keeps one home register - which is all the loop needs from the compiler. Its own code touches them
  // in two places: the mask (it writes S, and qk_acc_masked defines S again) and the output (it reads the row sums).
  // tests/test_cpu_host.py counts the pads and the register copies left in the steady loop.
  wf32x16_t S[2][2];          // [sub-block][32-key block]: S^T of the tile in flight
  wu32x4_t pwv[2][4];         // P^T as packed pairs [sub-block][k-step sk]: the B operand of that step
  float er0[3], er1[3];       // exponentials between their v_exp and their sum + pack: a ring of three pairs (word j's are
                              // written at block 2 j - 2 and read at 2 j + 3, word j + 3's written at 2 j + 4; the two
                              // sub-blocks' streams never interleave: B's ends with segment 1, A's runs over segments 2 and 3)
  wu32x4_t vfr[4][4];         // transposed V fragments [output block b][k-step sk] (ordinary outputs of their LDS reads)
  float e0[2][16], e1[2][16]; // (M16)
  float ps0[2] = {0.0f, 0.0f}, ps1[2] = {0.0f, 0.0f};   // running row sums (two chains)
  // M16: S16[x][rt][kt] (keys 16 kt + 4 g4 + r of row 32 x + 16 rt + r16), pw16[x][rt][c] (the B operand of the P.V step
  // over keys 32 c ..: dwords (kt = 2c: r 0,1 | r 2,3 | kt = 2c + 1: r 0,1 | r 2,3)), vfr16[db][c], row sums per (x, rt)
  // (hand-owned as well; the row sums come off the matrix pipe: L16[x][rt] += 1 . P^T, two instructions per tile each)
  // (ONE set of score registers serves both sub-blocks: S_A is written in segment 1 and its last exponential issues
  // before segment 2 ends; S_B is written from segment 3 on and consumed by the end of segment 4 - in steady and general
  // iterations alike. Key tiles are consumed in the order they are produced, so the first chains of a segment write
  // registers whose exponentials issued a segment earlier.)
  wf32x4_t S16[1][2][4];
  wu32x4_t pwv16[2][2][2];
  wu32x4_t vfr16[8][2];
  wf32x4_t L16[2][2];
  wu32x4_t ones16;
  // The rows' REFERENCES (M16): R16[x][rt] = -m_ref of this lane's row in all four registers, the C operand every score
  // chain starts from, so that P = 2^(s - m_ref) comes straight off the exponential. m_ref is an ESTIMATE of the row's
  // largest score - the maximum over the first 16 keys of the item's first tile (set_references, 16 matrix instructions
  // per item) - not a running maximum: softmax is shift invariant, so any reference gives the same result as long as
  // P and its sums stay inside the format, and that is checked per row when the block is done (bf16: scores within
  // ~+-90 of the reference in log2 units; f16: P <= 65504 leaves 16 + kRefMargin above it, and what falls 24 - kRefMargin
  // below it rounds to zero - 2^-18 of the reference term). A fixed reference 0 (rounds 1-2) failed for any row whose
  // scores all sit far from zero; the first keys of a row are where attention sinks live.
  wf32x4_t R16[2][2];
  // (SC: the chains start from 0 and R16[x][rt][0] is the reference c of a_softcap_exp_ho; 1 / B sits in a VGPR because one
  // v_fma takes it as two operands)
  float sc_inv_b = SC ? 1.0f / (2.0f * p.softcap * kPwLog2e) : 0.0f;
  if constexpr (SC) asm volatile("" : "+v"(sc_inv_b));
  // ALiBi state (AL): al_sl[x][rt] = the row's slope in log2 units (a row's head never changes: loaded once), al_base[x][rt] =
  // -m_ref - slope * (the row's own position), so that the bias of the row's own key is 0 and every visible key's bias is
  // <= 0: P = 2^(s - m_ref + slope (j - i)) stays where the plain kernel's P is. R16 becomes the PER-TILE C operand:
  // R16[x][rt][r] = al_base + slope * (the lane's first key of the tile + r), rebuilt before each score segment (alibi_c16).
  float al_sl[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}}, al_base[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
  unsigned al_ref_bad = 0;            // bit 2 x + rt: the row's estimated maximum is beyond kPwRefMax (see the epilogue)
  if constexpr (AL) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const int m_row = wave * 64 + (2 * x + rt) * 16 + (lane & 15);
        const int hq = head * a.group + m_row % a.group;
        al_sl[x][rt] = (m_row < a.block_q * a.group) ? p.alibi_slopes[hq] * kPwLog2e : 0.0f;
        pw_launder(al_sl[x][rt]);
      }
  }
  auto reset_state = [&]() __attribute__((always_inline)) {
    if constexpr (M16) {
#pragma unroll
      for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          L16[x][rt] = wf32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
          pw_launder(L16[x][rt]);
          R16[x][rt] = wf32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
          pw_launder(R16[x][rt]);
          if (x == 0)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) { S16[0][rt][kt] = wf32x4_t{-INFINITY, -INFINITY, -INFINITY, -INFINITY}; pw_launder(S16[0][rt][kt]); }
#pragma unroll
          for (int c = 0; c < 2; ++c) { pwv16[x][rt][c] = wu32x4_t{0, 0, 0, 0}; pw_launder(pwv16[x][rt][c]); }
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) { er0[i] = 0.0f; er1[i] = 0.0f; pw_launder(er0[i]); pw_launder(er1[i]); }
      ones16 = wu32x4_t{ops16::kOnes, ops16::kOnes, ops16::kOnes, ops16::kOnes};      // pairs of 1.0
      pw_launder(ones16);
#pragma unroll
      for (int db = 0; db < 8; ++db) { vfr16[db][0] = wu32x4_t{0, 0, 0, 0}; vfr16[db][1] = wu32x4_t{0, 0, 0, 0}; }
    }
    else {
#pragma unroll
      for (int x = 0; x < 2; ++x) {
        ps0[x] = 0.0f; ps1[x] = 0.0f;
        pw_launder(ps0[x]); pw_launder(ps1[x]);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
          for (int r = 0; r < 16; ++r) S[x][kb][r] = -INFINITY;
          pw_launder(S[x][kb]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { pwv[x][i] = wu32x4_t{0, 0, 0, 0}; pw_launder(pwv[x][i]); }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i) { er0[i] = 0.0f; er1[i] = 0.0f; pw_launder(er0[i]); pw_launder(er1[i]); }
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int sk = 0; sk < 4; ++sk) vfr[b][sk] = wu32x4_t{0, 0, 0, 0};
    }
  };

  // ---- the rows' references (see R16): scores of the item's first tile against its first 16 keys (K fragments kt = 0
  // are in their registers), maximum per row, over the four lane groups that hold a row's keys. Keys the row may not
  // see (causal mask, sliding window) are left in: the reference only has to be in the neighbourhood of the row's
  // scores. f16 leaves kRefMargin powers of two of head room above the estimate.
  constexpr float kRefMargin = __is_same(T, f16_t) ? 6.0f : 0.0f;
  auto set_references = [&](const Item& I) __attribute__((always_inline)) {
    // (temporaries: score registers [rt][x] - nothing of a tile is in them yet; ordinary operands here, so the compiler
    // sees them defined again)
    sfor<2>([&](auto X) __attribute__((always_inline)) {
      sfor<2>([&](auto RT) __attribute__((always_inline)) {
        constexpr int x = decltype(X)::value, rt = decltype(RT)::value;
        sfor<kKS>([&](auto KS) __attribute__((always_inline)) {
          constexpr int ks = decltype(KS)::value, KA = kAK + 4 * ks, QA = kAQ + 32 * x + 16 * rt + 4 * ks;
          if constexpr (ks == 0) ops16::template qk_zero<KA, QA>(S16[0][rt][x]); else ops16::template qk_acc<KA, QA>(S16[0][rt][x]);
        });
      });
    });
    // the matrix pipe's results are readable (the compiler does not know these statements are matrix instructions)
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(S16[0][0][0]), "+v"(S16[0][0][1]), "+v"(S16[0][1][0]), "+v"(S16[0][1][1]));
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const wf32x4_t tv = S16[0][rt][x];
        float m = fmaxf(fmaxf(tv[0], tv[1]), fmaxf(tv[2], tv[3]));
        m = fmaxf(m, lane_xor16(m));
        m = fmaxf(m, lane_xor32(m));
        float r = -(m + kRefMargin);
        // SC: m is the largest u; the reference is what the cap makes of it, c = B / (1 + 2^u) - margin (P = 2^(c - B / (1 + 2^u)))
        if constexpr (SC) r = __builtin_amdgcn_rcpf((1.0f + __builtin_amdgcn_exp2f(m)) * sc_inv_b) - kRefMargin;
        if constexpr (AL) {        // the bias is taken relative to the row's own key (see al_base); R16 is rebuilt per tile
          int m_row = wave * 64 + (2 * x + rt) * 16 + (lane_o & 15);
          asm volatile("" : "+v"(m_row));
          al_base[x][rt] = r - al_sl[x][rt] * (float)(I.ctx_len + I.tok0 + div_g(m_row));
          pw_launder(al_base[x][rt]);
          if (x == 0 && rt == 0) al_ref_bad = 0;
          al_ref_bad |= (fabsf(r) <= kPwRefMax) ? 0u : (1u << (2 * x + rt));
        }
        R16[x][rt] = wf32x4_t{r, r, r, r};
        pw_launder(R16[x][rt]);
      }
    asm volatile("s_nop 1" ::: "memory");        // VALU write -> matrix instruction reading it as C
  };

  Item cur;
#ifdef MI355_PW_STAMP
  unsigned long long st_entry;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_entry) :: "memory");
#endif
  // counters back to zero: the last workgroup of this KV head to get here has seen every other one's final draw land
  auto finish = [&]() __attribute__((always_inline)) {
    if (dynamic && wave == 0 && lane_o == 0) {
      int* const done = sa.tickets + sa.num_kv_heads + head;
      if (__hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == sa.slots - 1) {
        __hip_atomic_store(sa.tickets + head, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  };
  if (!acquire(cur, dynamic ? slot : 0, dynamic)) { finish(); return; }
  issue_q(cur, -1);
  issue_first_tiles(cur);
  zero_o_and_convert_q(cur, true);

  // ---- the pieces of an iteration ------------------------------------------------------------------------
  // MFMA g of S_x = K.Q_x^T: 32-key block g >> 3, k-step g & 7. A chain starts from the constant 0; on a tile that
  // straddles the causal diagonal or the sequence end it starts from 0 / -inf per key instead (the mask enters
  // through the accumulator: S is never modified after its chain).
  auto qk = [&](auto X, auto GC, int t, bool need_mask) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value, g = decltype(GC)::value, kb = g >> 3, ks = g & 7;
    constexpr int KA = kAK + 32 * kb + 4 * ks, QA = kAQ + 32 * x + 4 * ks;
    if constexpr (ks == 0) {
      if (__builtin_expect(!need_mask, 1)) {
        ops::template qk_zero_ho<KA, QA>(S[x][kb]);
      } else {
        const int rel = row_lim(cur, x) - t * kPwTile - 32 * kb - 4 * half;     // visible: (r & 3) + 8 (r >> 2) <= rel
#pragma unroll
        for (int r = 0; r < 16; ++r) S[x][kb][r] = ((r & 3) + 8 * (r >> 2) <= rel) ? 0.0f : -INFINITY;
        ops::template qk_acc_masked<KA, QA>(S[x][kb]);      // (an ordinary read-write operand: S is defined again here)
      }
    } else {
      ops::template qk_acc_ho<KA, QA>(S[x][kb]);
    }
  };
  // MFMA g of O_x += V^T.P_x^T: output block g >> 2, k-step g & 3
  auto pv = [&](auto X, auto GC) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value, g = decltype(GC)::value, b = g >> 2, sk = g & 3;
    ops::template pv<kAO + 64 * x + 16 * b>(vfr[b][sk], pwv[x][sk]);
  };
  // exponentials, row sums and packing of sub-block x: 32 two- or three-instruction statements (per P word: exp exp,
  // and a few blocks later add add cvt) dealt over a window of 28 MFMA gaps. One statement per block: hipcc pads
  // BETWEEN asm statements (a statement that reads a register an earlier one wrote, with none of the compiler's own
  // instructions between them), never inside one; a transcendental's result is read at least a gap after it was
  // written. (Exponentials as plain builtins would draw no padding, but hipcc then moves them next to the asm MFMA
  // whose result they read - it does not know the 12 issue slots that result needs - and measured slower as well.)
  auto estream = [&](auto X, auto WC) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value, w = decltype(WC)::value;
    constexpr int n0 = w * 32 / 28, n1 = (w + 1) * 32 / 28;
#ifdef PW_ABL_E
    return;
#endif
    sfor<n1 - n0>([&](auto NC) __attribute__((always_inline)) {
      constexpr int n = n0 + decltype(NC)::value;
      // block n: X X | X P X P ... X P | P P   (X = exponentials of a word, P = sum + pack of a word)
      constexpr bool is_x = n < 2 || (n < 30 && ((n - 2) & 1) == 0);
      constexpr int j = n < 2 ? n : n >= 30 ? 14 + (n - 30) : is_x ? 2 * (1 + (n - 2) / 4) + (((n - 2) >> 1) & 1) : 2 * ((n - 2) / 4) + (((n - 2) >> 1) & 1);
      constexpr int kb = j >> 3, r = 2 * (j & 7);
      // (see er0 / er1)
      if constexpr (is_x) a_exp2x2_ho(er0[j % 3], er1[j % 3], S[x][kb][r], S[x][kb][r + 1]);
      else ops::sum_pack_ho(ps0[x], ps1[x], pwv[x][j >> 2][j & 3], er0[j % 3], er1[j % 3]);
    });
  };
  // LDS reads. V(t)[b][sk]: two transposed 8-byte reads into one 4-register fragment
  auto vread = [&](auto B, auto SK, auto SLOT) __attribute__((always_inline)) {
    constexpr int b = decltype(B)::value, sk = decltype(SK)::value, off = decltype(SLOT)::value + sk * 16 * ROWB;
#ifdef PW_ABL_LDS
    return;
#endif
    const wu32x2_t v0 = lds_tr_b64<off>(v_rd0[b]);
    const wu32x2_t v1 = lds_tr_b64<off>(v_rd1[b]);
    vfr[b][sk] = wu32x4_t{v0[0], v0[1], v1[0], v1[1]};
  };
  auto kread = [&](auto NC, auto SLOT) __attribute__((always_inline)) {
    constexpr int n = decltype(NC)::value, kb = n >> 3, ks = n & 7;
#ifdef PW_ABL_LDS
    return;
#endif
    lds_to_acc_b128<kAK + 32 * kb + 4 * ks, decltype(SLOT)::value + kb * 32 * ROWB>(k_rd[ks]);
  };

  // ---- the same pieces on the 16x16x32 shape (M16) ----------------------------------------------------------
  // MFMA g (0..31) of S_x: key tile g >> 3, k-step (g >> 1) & 3, row tile g & 1
  // AL: the C operand of sub-block x's score chains for tile t: reference + slope * (position of this lane's key r of key
  // tile 0 - the row's own position); key tiles 1..3 add slope * 16 kt in front of their exponentials (a_alibi_exp_ho).
  auto alibi_c16 = [&](auto X, int t) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value;
    int k0 = t * kPwTile + 4 * g4;
    asm volatile("" : "+v"(k0));
    const float k0f = (float)k0;
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
      const float sl = al_sl[x][rt], b = __builtin_fmaf(sl, k0f, al_base[x][rt]);
      R16[x][rt] = wf32x4_t{b, b + sl, __builtin_fmaf(sl, 2.0f, b), __builtin_fmaf(sl, 3.0f, b)};
    }
    asm volatile("s_nop 1" : "+v"(R16[x][0]), "+v"(R16[x][1]));      // (VALU-written C operands: two wait states before the matrix instruction)
  };
  auto qk16 = [&](auto X, auto GC) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value, g = decltype(GC)::value, kt = g >> 3, ks = (g >> 1) & 3, rt = g & 1;
    constexpr int KA = kAK + 16 * kt + 4 * ks, QA = kAQ + 32 * x + 16 * rt + 4 * ks;
    if constexpr (ks >= kKS) return;              // (D = 64 / 96: a chain has two / three k-steps; the slot stays in the schedule, empty)
    else if constexpr (ks == 0) {
      if constexpr (SC) ops16::template qk_zero_ho<KA, QA>(S16[0][rt][kt]);          // the raw score: the reference enters after the cap
      else ops16::template qk_ref_ho<KA, QA>(S16[0][rt][kt], R16[x][rt]);
    } else ops16::template qk_acc_ho<KA, QA>(S16[0][rt][kt]);
  };
  // The mask of sub-block x, applied to the finished scores (sixteen accumulator set-ups per sub-block: a run-time branch
  // around each, as in the 32x32 form, costs a general iteration a third of its time). Runs right behind the segment
  // that produced S_x: the wait states for its last matrix instructions sit in a statement that names the registers.
  auto mask16 = [&](auto X, int t) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value;
    asm volatile("s_nop 7\n\ts_nop 3"
                 : "+v"(S16[0][0][0]), "+v"(S16[0][0][1]), "+v"(S16[0][0][2]), "+v"(S16[0][0][3]),
                   "+v"(S16[0][1][0]), "+v"(S16[0][1][1]), "+v"(S16[0][1][2]), "+v"(S16[0][1][3]));
    int key0 = t * kPwTile + 4 * g4;        // this lane's first key of key tile 0 (opaque: lane constants folded with it
    asm volatile("" : "+v"(key0));          // would be hoisted out of the item loop, and there is no register for them)
    float ninf = -INFINITY;
    asm volatile("" : "+v"(ninf));
    // which bound this tile crosses for the wave (the two terms of need_mask): the causal diagonal / the sequence's end,
    // or - sliding window - the lower edge; a tile rarely crosses both
    const bool need_hi = !SW || (t * kPwTile + kPwTile - 1 > cur.ctx_len + cur.w_tok_lo) || (t * kPwTile + kPwTile > cur.seq_len);
    const bool need_lo = SW && sa.window > 0 && t * kPwTile < cur.ctx_len + cur.w_tok_hi - sa.window + 1;
    if (need_hi) {
      sfor<2>([&](auto RT) __attribute__((always_inline)) {
        constexpr int rt = decltype(RT)::value;
        const int rlim = row_lim(cur, 2 * x + rt) - key0;        // visible: r <= rlim - 16 kt
        sfor<4>([&](auto KT) __attribute__((always_inline)) {
          constexpr int kt = decltype(KT)::value;
          pw_mask_quad_ho<false>(S16[0][rt][kt], rlim - 16 * kt, ninf);
        });
      });
    }
    if constexpr (SW) {
      if (need_lo) {
        sfor<2>([&](auto RT) __attribute__((always_inline)) {
          constexpr int rt = decltype(RT)::value;
          const int rlow = row_lo(cur, 2 * x + rt) - key0;       // ... and r >= rlow - 16 kt
          sfor<4>([&](auto KT) __attribute__((always_inline)) {
            constexpr int kt = decltype(KT)::value;
            pw_mask_quad_ho<true>(S16[0][rt][kt], rlow - 16 * kt, ninf);
          });
        });
      }
    }
  };
  // MFMA g (0..31) of O_x += V^T.P_x^T: d tile g >> 2, 32-key block (g >> 1) & 1, row tile g & 1
  auto pv16 = [&](auto X, auto GC) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value, g = decltype(GC)::value, db = (g >> 2) & 7, c = (g >> 1) & 1, rt = g & 1;
    if constexpr (g < 32) { if constexpr (db < kDB) ops16::template pv<kAO + 64 * x + 32 * rt + 4 * db>(vfr16[db][c], pwv16[x][rt][c]); }      // (D = 64 / 96: output tiles 0 .. 3 / 0 .. 5)
    else ops16::lsum_ho(L16[x][rt], ones16, pwv16[x][rt][c]);        // g = 32 .. 35: the row sums of this tile
  };
  // Instruction q (0..47) of sub-block x's exponential / pack stream. A 16-cycle matrix instruction leaves this wave ~7
  // cycles of issue in its shadow: one v_exp_f32 fills it, two stall the pipe (tools/probes/issue_model.hip), so the stream
  // is dealt one instruction per gap. Words in key-tile order (w: key tile w >> 2, row tile (w >> 1) & 1, register pair
  // w & 1), blocks X X | X P X P ... | P P as in the 32x32 form: word w's exponentials are instructions 3 w - 2, 3 w - 1,
  // after the score tile 8 (w >> 2) + 10 gaps into the S_x segment (two matrix instructions behind the tile's last one).
  auto eop16 = [&](auto X, auto QC) __attribute__((always_inline)) {
    constexpr int x = decltype(X)::value;
    constexpr PwEOp op = pw_eop(decltype(QC)::value);
    constexpr int w = op.j, kt = w >> 2, rt = (w >> 1) & 1, pr = w & 1;
#ifdef PW_ABL_E
    return;
#endif
    if constexpr (op.kind == 0) {
      if constexpr (SC) { }        // (soft-cap: the word's two scores go together in the second one's slot, a_softcap_exp2_ho)
      else if constexpr (AL && kt > 0) a_alibi_exp_ho<(kt > 0 ? kt : 1)>(er0[w % 3], S16[0][rt][kt][2 * pr], al_sl[x][rt]);
      else a_exp_ho(er0[w % 3], S16[0][rt][kt][2 * pr]);
    } else if constexpr (op.kind == 1) {
      if constexpr (SC) a_softcap_exp2_ho(er0[w % 3], S16[0][rt][kt][2 * pr], er1[w % 3], S16[0][rt][kt][2 * pr + 1], R16[x][rt][0], sc_inv_b);
      else if constexpr (AL && kt > 0) a_alibi_exp_ho<(kt > 0 ? kt : 1)>(er1[w % 3], S16[0][rt][kt][2 * pr + 1], al_sl[x][rt]);
      else a_exp_ho(er1[w % 3], S16[0][rt][kt][2 * pr + 1]);
    }
    else if constexpr (op.kind == 2) ops16::pack_ho(pwv16[x][rt][kt >> 1][2 * (kt & 1) + pr], er0[w % 3], er1[w % 3]);
  };
  // the stream's instructions in a P.V segment's gap g (36 gaps). Steady iterations: 22 went out during the S_x segment,
  // the other 26 take the even gaps and eight odd ones (the odd gaps carry the LDS reads); general iterations: all 48
  // behind the mask, four per three gaps.
  auto estream_pv = [&](auto X, auto GC, auto LATE) __attribute__((always_inline)) {
    constexpr int g = decltype(GC)::value;
    if constexpr (decltype(LATE)::value != 0) {
      constexpr int q0 = g * 48 / 36, q1 = (g + 1) * 48 / 36;
      sfor<q1 - q0>([&](auto I) __attribute__((always_inline)) { eop16(X, ic<q0 + decltype(I)::value>{}); });
    } else {
      constexpr int slot = pw_pv_slot(g);
      if constexpr (slot >= 0) eop16(X, ic<(slot >= 0 ? 22 + slot : 0)>{});
    }
  };
  // V(t) fragment (db, c): the 4 x 16 blocks at rows 32 c + 4 g4 and + 16
  auto vread16 = [&](auto FC, auto SLOT) __attribute__((always_inline)) {
    constexpr int f = decltype(FC)::value, db = f >> 1, c = f & 1, off = decltype(SLOT)::value + c * 32 * ROWB;
#ifdef PW_ABL_LDS
    return;
#endif
    if constexpr (db >= kDB) return;
    const wu32x2_t v0 = lds_tr_b64<off>(v_rd16[db]);
    const wu32x2_t v1 = lds_tr_b64<off + 16 * ROWB>(v_rd16[db]);
    vfr16[db][c] = wu32x4_t{v0[0], v0[1], v1[0], v1[1]};
  };
  auto kread16 = [&](auto NC, auto SLOT) __attribute__((always_inline)) {
    constexpr int n = decltype(NC)::value, kt = n >> 2, ks = n & 3;
#ifdef PW_ABL_LDS
    return;
#endif
    if constexpr (ks >= kKS) return;
    lds_to_acc_b128<kAK + 16 * kt + 4 * ks, decltype(SLOT)::value + kt * 16 * ROWB>(k_rd16[ks]);
  };

#ifdef MI355_PW_STAMP
  unsigned st_sum[6] = {0, 0, 0, 0, 0, 0}, st_last = 0;
#endif
  // One iteration = one KV tile t. Matrix pipe: S_A(t) | O_B += P_B(t-1) | S_B(t) | O_A += P_A(t); beside it, per gap:
  //   seg 1: B(t-1) exponentials (second half) . LDS-DMA of K(t+3), V(t+2)
  //   seg 2: A exponentials . V(t) fragment reads as V(t-1)'s registers retire
  //   seg 3: A exponentials (second half) . K(t+1) fragment reads as K(t)'s registers retire
  //   seg 4: B exponentials
  // IT = (t - tile_lo) % 3 picks the ring slots at compile time.
  // STEADY iterations (see the loops below) know at compile time that tile t needs no mask and that the groups this
  // wave fetches for tiles t+2 .. t+4 lie inside the share and the sequence: no tail handling, no clamps, and the
  // scalar address arithmetic sits BEHIND the first matrix instructions instead of in front of them (the general
  // iteration spends ~55 scalar instructions and five branches between the barrier and its first MFMA).
  auto iteration = [&](auto ITC, auto STC, int t) __attribute__((always_inline)) {
    constexpr int it = decltype(ITC)::value;
    // STC: 1 = steady (see above); 0 = general; 2 (M16) = UNMASKED TAIL: the compute side of a steady iteration (no mask,
    // exponentials dealt from the moment a sub-block's first score tile is done) with the fetch side of a general one
    // (groups clamped to the share and the sequence, rows past the sequence's end redirected), its scalar arithmetic
    // behind the first matrix instructions. A Q block's last four or five tiles lie past what the steady form may
    // fetch blindly, but only the one or two on the causal diagonal need a mask.
    constexpr bool steady = decltype(STC)::value != 0;
    constexpr bool fast_fetch = decltype(STC)::value == 1;
    constexpr int KR = ((it + 1) % 3) * kSlotBytes;            // K(t+1) is read from here
    constexpr int VR = (it % 3) * kSlotBytes;                  // V(t)
    constexpr int KD = kLdsK + (it % 3) * kSlotBytes;          // K(t+3) goes where K(t) was
    constexpr int VD = kLdsV + ((it + 2) % 3) * kSlotBytes;    // V(t+2) goes where V(t-1) was
    bool need_mask = false;
    uint64_t kb64 = 0, vb64 = 0;
    // scalar side of this iteration's LDS-DMA (entries fetched during the previous iteration), and the fetch of the
    // next iteration's entries (they land before the wait that ends this one)
    if constexpr (!steady) {
      need_mask = (t * kPwTile + kPwTile - 1 > cur.ctx_len + cur.w_tok_lo) || (t * kPwTile + kPwTile > cur.seq_len);
      if constexpr (SW) need_mask = need_mask || (sa.window > 0 && t * kPwTile < cur.ctx_len + cur.w_tok_hi - sa.window + 1);   // below the window of the wave's last row
      tail_check(cur, t + 3 + FO, ic<0>{});
      tail_check(cur, t + 2 + FO, ic<1>{});
      kb64 = group_base(cur, t + 3 + FO, pg_k, ic<0>{});
      vb64 = group_base(cur, t + 2 + FO, pg_v, ic<1>{});
      pg_v = pg_k;                                 // V(t+3) lives in the page of K(t+3)
      pw_sload(pg_k, cur.bt64, entry_off(cur, t + 4 + FO));
    }
    // piece j of this iteration's LDS-DMA: 0..3 = K(t+3), 4..7 = V(t+2)
    auto dma_piece = [&](auto JC) __attribute__((always_inline)) {
      constexpr int j = decltype(JC)::value;
#ifndef PW_ABL_DMA
      if constexpr (KV8) {       // fp8: pieces 0, 1 = K(t+4), 2, 3 = V(t+3), into this wave's staging area
        if constexpr (j < 2) pw_glds16(koff[j], kb64, kLdsS + lds_wave + j * 1024);
        else pw_glds16(voff[j - 2], vb64, kLdsS + 2048 + lds_wave + (j - 2) * 1024);
      }
      else if constexpr (j < 4) pw_glds16(koff[j], kb64, KD + lds_wave + j * 1024);
      else pw_glds16(voff[j - 4], vb64, VD + lds_wave + (j - 4) * 1024);
#endif
    };
    // fp8: this wave's groups of K(t+3) / V(t+2), staged by the previous iteration's pieces
    wu32x4_t st8[4];
#ifdef PW_DMA_SPREAD
#define PW_DMA_AT(seg, g) if constexpr ((g) == 1 || (g) == 9) dma_piece(ic<2 * ((seg) - 1) + ((g) == 9)>{})
#else
#define PW_DMA_AT(seg, g) if constexpr ((seg) == 1 && (g) >= 2 && (g) < 10) dma_piece(ic<((seg) == 1 && (g) >= 2 && (g) < 10) ? (g) - 2 : 0>{})
#endif
    if constexpr (M16) {
      // 32 / 36 matrix instructions of 16 cycles per segment. S_A | O_B += P_B(t-1), l_B | S_B | O_A += P_A(t), l_A. Beside them,
      // one instruction per gap where it can be had: A's exponentials from gap 10 of segment 1 on (its first score tile is
      // done) through segment 2, B's from gap 10 of segment 3 through segment 4; the LDS-DMA in gaps 2 .. 9 of segment 1;
      // V(t) fragment f (read by instructions 2 f, 2 f + 1 of segment 2) re-loaded in the odd gaps from 2 f + 3 on; K(t+1)
      // fragments 0 .. 3 in segment 3, the others in the gaps segment 4's stream leaves free.
      if constexpr (AL) alibi_c16(ic<0>{}, t);
      sfor<32>([&](auto GC) __attribute__((always_inline)) {
        constexpr int g = decltype(GC)::value;
        qk16(ic<0>{}, GC);
        if constexpr (steady && g == 0) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (fast_fetch) kb64 = group_base_fast(cur, t + 3 + FO, pg_k, ic<0>{});
          else { tail_check(cur, t + 3 + FO, ic<0>{}); kb64 = group_base(cur, t + 3 + FO, pg_k, ic<0>{}); }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (steady && g == 1) {
          __builtin_amdgcn_sched_barrier(0);
          if constexpr (fast_fetch) vb64 = group_base_fast(cur, t + 2 + FO, pg_v, ic<1>{});
          else { tail_check(cur, t + 2 + FO, ic<1>{}); vb64 = group_base(cur, t + 2 + FO, pg_v, ic<1>{}); }
          asm volatile("s_mov_b32 %0, %1" : "=s"(pg_v) : "s"(__builtin_amdgcn_readfirstlane(pg_k)));      // (a scalar copy: the compiler otherwise parks it in a VGPR and does V's page arithmetic on the VALU, v_mul_lo/hi + two v_readfirstlane)
          if constexpr (!KV8) {        // (fp8: behind the staging reads' wait in gap 4, which would otherwise wait for this load as well)
            if constexpr (fast_fetch) pw_sload(pg_k, cur.bt64, entry_off_fast(t + 4));
            else pw_sload(pg_k, cur.bt64, entry_off(cur, t + 4));
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (KV8) {
          // fp8: the turn-over of the staging area. Gap 0: last iteration's four pieces are back, the two staged groups go to
          // registers; gap 4: they are there, and the next pieces may overwrite them (gaps 4, 6, 8, 10); the widening - sixteen
          // steps of two conversions, a 16-byte write after every second - in the gaps from 5 on.
          if constexpr (g == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            st8[0] = pw_lds_read128<0>(st_rd); st8[1] = pw_lds_read128<1024>(st_rd);
            st8[2] = pw_lds_read128<2048>(st_rd); st8[3] = pw_lds_read128<3072>(st_rd);
          }
          if constexpr (g == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(st8[0]), "+v"(st8[1]), "+v"(st8[2]), "+v"(st8[3]) :: "memory");
          if constexpr (g == 5 && steady) {
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (fast_fetch) pw_sload(pg_k, cur.bt64, entry_off_fast(t + 5));
            else pw_sload(pg_k, cur.bt64, entry_off(cur, t + 5));
            __builtin_amdgcn_sched_barrier(0);
          }
          if constexpr (g >= 4 && g <= 10 && (g & 1) == 0) dma_piece(ic<((g >= 4 && g <= 10) ? (g - 4) / 2 : 0)>{});
          // (placement, profiles/r04/fp8_direct_prefill.log: both groups in one go costs 4 % more; V's steps in segment 3 keep sixteen
          // more registers live and hipcc then parks values in accumulator registers - which belong to the asm statements)
          if constexpr (g >= 5 && g < 17) widen_step(ic<0>{}, ic<KD>{}, ic<(g >= 5 && g < 17) ? g - 5 : 0>{}, st8[0], st8[1]);
          if constexpr (g >= 17 && g < 29) widen_step(ic<1>{}, ic<VD - kLdsV>{}, ic<(g >= 17 && g < 29) ? g - 17 : 0>{}, st8[2], st8[3]);
        }
        // (the four waves issue their pieces at the same time and the CU's address unit takes 64 cycles per round of four:
        // eight pieces in eight consecutive gaps stall the issue - K(t+3) here, V(t+2) in segment 3, every other gap)
        if constexpr (!KV8 && g >= 2 && g < 10 && (g & 1) == 0) dma_piece(ic<((g >= 2 && g < 10) ? (g - 2) / 2 : 0)>{});
        if constexpr (steady && g >= 10) eop16(ic<0>{}, ic<(g >= 10 ? g - 10 : 0)>{});
      });
      PW_SEG_STAMP(1);
      if constexpr (!steady) { if (__builtin_expect(need_mask, 0)) mask16(ic<0>{}, t); }
      sfor<36>([&](auto GC) __attribute__((always_inline)) {
        constexpr int g = decltype(GC)::value;
        pv16(ic<1>{}, GC);
        estream_pv(ic<0>{}, GC, ic<(steady ? 0 : 1)>{});
        if constexpr (g >= 3 && g < 32 && (g & 1) == 1) vread16(ic<(g >= 3 && g < 32) ? ((g - 3) / 2) : 0>{}, ic<VR>{});
      });
      PW_SEG_STAMP(2);
      if constexpr (AL) alibi_c16(ic<1>{}, t);
      sfor<32>([&](auto GC) __attribute__((always_inline)) {
        constexpr int g = decltype(GC)::value;
        qk16(ic<1>{}, GC);
        if constexpr (!KV8 && g < 8 && (g & 1) == 0) dma_piece(ic<(g < 8 ? 4 + g / 2 : 4)>{});
        if constexpr (g == 1) vread16(ic<15>{}, ic<VR>{});
        if constexpr (g >= 3 && g < 10 && (g & 1) == 1) kread16(ic<((g >= 3 && g < 10) ? (g - 3) / 2 : 0)>{}, ic<KR>{});
        if constexpr (steady && g >= 10) eop16(ic<1>{}, ic<(g >= 10 ? g - 10 : 0)>{});
      });
      PW_SEG_STAMP(3);
      if constexpr (!steady) { if (__builtin_expect(need_mask, 0)) mask16(ic<1>{}, t); }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(vfr16[0][0]), "+v"(vfr16[0][1]), "+v"(vfr16[1][0]), "+v"(vfr16[1][1]), "+v"(vfr16[2][0]), "+v"(vfr16[2][1]), "+v"(vfr16[3][0]), "+v"(vfr16[3][1]),
                     "+v"(vfr16[4][0]), "+v"(vfr16[4][1]), "+v"(vfr16[5][0]), "+v"(vfr16[5][1]), "+v"(vfr16[6][0]), "+v"(vfr16[6][1]), "+v"(vfr16[7][0]), "+v"(vfr16[7][1]));
      sfor<36>([&](auto GC) __attribute__((always_inline)) {
        constexpr int g = decltype(GC)::value;
        pv16(ic<0>{}, GC);
        estream_pv(ic<1>{}, GC, ic<(steady ? 0 : 1)>{});
        constexpr int kf = pw_kread_slot(g);
        if constexpr (kf >= 0) kread16(ic<(kf >= 0 ? kf : 0)>{}, ic<KR>{});
      });
      PW_SEG_STAMP(4);
    } else {
    // ---- segment 1 -------------------------------------------------------------------------------------
    sfor<16>([&](auto GC) __attribute__((always_inline)) {
      constexpr int g = decltype(GC)::value;
      qk(ic<0>{}, GC, t, need_mask);
      if constexpr (steady && g == 0) {
        __builtin_amdgcn_sched_barrier(0);
        kb64 = group_base_fast(cur, t + 3, pg_k, ic<0>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (steady && g == 1) {
        __builtin_amdgcn_sched_barrier(0);
        vb64 = group_base_fast(cur, t + 2, pg_v, ic<1>{});
        pg_v = pg_k;
        pw_sload(pg_k, cur.bt64, entry_off_fast(t + 4));
        __builtin_amdgcn_sched_barrier(0);
      }
      estream(ic<1>{}, ic<12 + g>{});
      PW_DMA_AT(1, g);
    });
    PW_SEG_STAMP(1);
    // ---- segment 2 -------------------------------------------------------------------------------------
    sfor<16>([&](auto GC) __attribute__((always_inline)) {
      constexpr int g = decltype(GC)::value;
      pv(ic<1>{}, GC);
      if constexpr (g >= 4) { estream(ic<0>{}, ic<g - 4>{}); vread(ic<((g - 4) >> 2)>{}, ic<((g - 4) & 3)>{}, ic<VR>{}); }
      PW_DMA_AT(2, g);
    });
    PW_SEG_STAMP(2);
    // ---- segment 3 -------------------------------------------------------------------------------------
    sfor<16>([&](auto GC) __attribute__((always_inline)) {
      constexpr int g = decltype(GC)::value;
      qk(ic<1>{}, GC, t, need_mask);
      estream(ic<0>{}, ic<12 + g>{});
      if constexpr (g < 4) vread(ic<3>{}, GC, ic<VR>{});
      if constexpr (g >= 8) kread(ic<g - 8>{}, ic<KR>{});        // K(t+1)[kb 0][ks]: K(t)[kb 0][ks] was last read in gap ks
      PW_DMA_AT(3, g);
    });
    PW_SEG_STAMP(3);
    // ---- segment 4 -------------------------------------------------------------------------------------
    // every V(t) fragment has landed (the last ones were issued twelve gaps ago)
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(vfr[0][0]), "+v"(vfr[0][1]), "+v"(vfr[0][2]), "+v"(vfr[0][3]), "+v"(vfr[1][0]), "+v"(vfr[1][1]), "+v"(vfr[1][2]), "+v"(vfr[1][3]),
                   "+v"(vfr[2][0]), "+v"(vfr[2][1]), "+v"(vfr[2][2]), "+v"(vfr[2][3]), "+v"(vfr[3][0]), "+v"(vfr[3][1]), "+v"(vfr[3][2]), "+v"(vfr[3][3]));
    sfor<16>([&](auto GC) __attribute__((always_inline)) {
      constexpr int g = decltype(GC)::value;
      pv(ic<0>{}, GC);
      if constexpr (g < 8) kread(ic<8 + g>{}, ic<KR>{});         // K(t+1)[kb 1][ks]
      if constexpr (g >= 4) estream(ic<1>{}, ic<g - 4>{});
      PW_DMA_AT(4, g);
    });
    PW_SEG_STAMP(4);
    }
    // K(t+1) is in its registers and the next block-table entries in theirs; K(t+2) and V(t+1) (issued one iteration
    // ago) have landed, this iteration's 8 pieces may stay in flight; everyone is done reading K(t+1)'s and V(t)'s slots
#ifdef PW_ABL_DMA
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(pg_k) :: "memory");
#else
    // (fp8: this iteration's four pieces are waited for where the next one reads them, gap 0 of its first segment)
    if constexpr (KV8) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(pg_k) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_barrier" : "+s"(pg_k) :: "memory");
#endif
    PW_SEG_STAMP(5);
#ifdef MI355_PW_SEAM
    if (t - cur.tile_lo < 2) st_sum[0] = st_last;      // (diagnostic: end of the item's second iteration, tools/pw_first_iterations.py)
#endif
  };

  // ---- an item's output: O / l through this wave's parking rows, whole 256-byte rows out ------------------
  // `hook(ic<part>)`, part 0 .. 3: called in front of each of the four row tiles' output (M16; the other form calls all
  // four first): the seam hangs the next item's loads there.
  auto epilogue = [&](const Item& I, auto hook) __attribute__((always_inline)) {
    const int G = sa.G;
    const uint32_t g_inv = (uint32_t)sa.g_inv;   // m / G == (m * g_inv) >> 16 for m < 256, G <= 256
    const float sc_a = SC ? kp->p.softcap * kPwLog2e : 0.0f;     // soft-cap: the constant A = cap log2(e) that the softmax drops and the lse needs
    const int lane = lane_o, qr = lane_o & 31, half = lane_o >> 5;
    char* ost = smem + kLdsO + wave * (32 * kPwORS);
    const bool wide_store = __builtin_amdgcn_readfirstlane((int)((((uintptr_t)I.out_base & 15) == 0) && (sa.out_st % 8 == 0) && (sa.out_sh % 8 == 0))) != 0;
    const bool whole_block = (I.q_len - I.tok0 >= sa.BQ) && (sa.BQ * G == kPwRows);   // every row of the block is a row of the sequence
    const int key_lo = I.tile_lo * kPwTile;
    const int orow = lane >> 4, och = lane & 15;
    // scalar address of the block's first token, this KV head's first query head; a row is a 32-bit byte offset from it
    typedef __attribute__((address_space(1))) char* gptr_t;
    const gptr_t out0 = (gptr_t)uniform64((uint64_t)(I.out_base + (int64_t)(I.q_start + I.tok0) * (int64_t)sa.out_st + (int64_t)(head * G) * (int64_t)sa.out_sh));
    const uint32_t st_b = (uint32_t)sa.out_st * 2u, sh_b = (uint32_t)sa.out_sh * 2u;
    const uint32_t tok_left = (uint32_t)min(I.q_len - I.tok0, sa.BQ);   // tokens of this block inside the sequence
    const float v_sc8 = (KV8 && kp->p.v_scale != nullptr) ? kp->p.v_scale[0] : 1.0f;
    bool bad[2];                                 // the row left the range the reference-0 arithmetic is good for
    bool bad16[2][2] = {{false, false}, {false, false}};   // M16: per (sub-block, row tile)
    if constexpr (M16) {
      const int r16o = lane_o & 15, g4o = lane_o >> 4;
      sfor<2>([&](auto SB) __attribute__((always_inline)) {
        constexpr int x = decltype(SB)::value;
        float amax2[2] = {0.0f, 0.0f}, l2[2];
        bool ok2[2];
        // one row tile's 32 output registers: scaled, packed, to `store(db, words)`
        auto tile_out = [&](auto RT, float inv, auto store) __attribute__((always_inline)) {
          constexpr int rt = decltype(RT)::value;
          sfor<kDB>([&](auto DB) __attribute__((always_inline)) {
            constexpr int db = decltype(DB)::value, base = kAO + 64 * x + 32 * rt + 4 * db;
            const float o0 = acc_read<base>(), o1 = acc_read<base + 1>(), o2 = acc_read<base + 2>(), o3 = acc_read<base + 3>();
            pw_amax3(amax2[rt], o0, o1);
            pw_amax3(amax2[rt], o2, o3);
            store(DB, wu32x2_t{pw_pack<T>(o0 * inv, o1 * inv), pw_pack<T>(o2 * inv, o3 * inv)});   // d = 16 db + 4 g4 + 0..3
          });
        };
        sfor<2>([&](auto RT) __attribute__((always_inline)) {
          constexpr int rt = decltype(RT)::value;
          hook(ic<2 * x + rt>{});
          const float l = L16[x][rt][0];                     // the row's whole sum (of P as P.V saw it: rounded to bf16)
          int tok_local, hq;
          const bool row_ok = row_of(I, 2 * x + rt, tok_local, hq);
          if (I.lse_base && row_ok && g4o == 0)
            I.lse_base[(int64_t)(I.q_start + tok_local) * sa.lse_st + hq] = l > 0.0f ? (__builtin_amdgcn_logf(l) - (AL ? al_base[x][rt] + al_sl[x][rt] * (float)I.ctx_len : R16[x][rt][0]) + sc_a) * 0.6931471805599453f : -INFINITY;   // P = 2^(score + R) (SC: 2^(capped score - A + R); AL: the bias as the reference counts it, from the context's end)
          float inv = (row_ok && l > 0.0f) ? __builtin_amdgcn_rcpf(l) : 0.0f;   // (1 ulp: the output is rounded to 16 bits next)
          if constexpr (KV8) inv *= v_sc8;                 // fp8 cache: V was widened unscaled
          l2[rt] = l; ok2[rt] = row_ok;
          if (__builtin_expect(wide_store, 1)) {
            tile_out(RT, inv, [&](auto DB, wu32x2_t w2) __attribute__((always_inline)) {
              *(wu32x2_t*)(ost + (16 * rt + r16o) * kPwORS + (16 * decltype(DB)::value + 4 * g4o) * 2) = w2;
            });
          } else {
            uint16_t* op = I.out_base + (int64_t)(I.q_start + tok_local) * (int64_t)sa.out_st + (int64_t)hq * (int64_t)sa.out_sh + 4 * g4o;
            tile_out(RT, inv, [&](auto DB, wu32x2_t w2) __attribute__((always_inline)) {
              if (row_ok) *(wu32x2_t*)(op + 16 * decltype(DB)::value) = w2;
            });
          }
        });
        if (__builtin_expect(wide_store, 1)) {
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          wu32x4_t rows[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) rows[j] = *(const wu32x4_t*)(ost + (4 * j + orow) * kPwORS + och * 16);
          typedef __attribute__((address_space(1))) wu32x4_t* grow_t;
          auto row_off = [&](int j, uint32_t& tq) {
            const uint32_t m = (uint32_t)(wave * 64 + x * 32 + 4 * j + orow);
            tq = __umul24(m, g_inv) >> 16;
            return __umul24(tq, st_b) + __umul24(m - __umul24(tq, (uint32_t)G), sh_b) + (uint32_t)och * 16u;   // (full-rate 24-bit multiplies: strides are below 2^22 elements)
          };
          const bool och_ok = D == 128 || och <= kCM;          // (D = 64 / 96: a row is eight / twelve chunks, the other lanes have nothing to store)
          if (__builtin_expect(whole_block, 1)) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { uint32_t tq; const uint32_t off = row_off(j, tq); if (och_ok) __builtin_nontemporal_store(rows[j], (grow_t)(out0 + off)); }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { uint32_t tq; const uint32_t off = row_off(j, tq); if (och_ok && tq < tok_left) __builtin_nontemporal_store(rows[j], (grow_t)(out0 + off)); }
          }
        }
        // the two row tiles' magnitude checks, after the stores are on their way (four cross-lane steps in one go)
        float am0 = fmaxf(amax2[0], __shfl_xor(amax2[0], 16, 64)), am1 = fmaxf(amax2[1], __shfl_xor(amax2[1], 16, 64));
        am0 = fmaxf(am0, lane_xor32(am0)); am1 = fmaxf(am1, lane_xor32(am1));
        sfor<2>([&](auto RT) __attribute__((always_inline)) {
          constexpr int rt = decltype(RT)::value;
          const bool has_keys = ok2[rt] && I.tile_hi > I.tile_lo && row_lim(I, 2 * x + rt) >= key_lo;
          bad16[x][rt] = has_keys && !(l2[rt] >= (__is_same(T, f16_t) ? kPwSumLoF16 : kPwSumLo) && l2[rt] <= kPwSumHi && (rt ? am1 : am0) < INFINITY);   // (an f16 P that overflowed is an inf in the sum)
          // scores of a magnitude at which the 2^-9 relative rounding of Q' = Q * scale * log2(e) to 16 bits moves the
          // DIFFERENCES between keys by tenths (thousands of log2 units: nothing a model produces): the f32 routine as well
          bad16[x][rt] = bad16[x][rt] || (has_keys && (AL ? ((al_ref_bad >> (2 * x + rt)) & 1u) != 0 : !(fabsf(R16[x][rt][0]) <= kPwRefMax)));
#ifdef PW_FORCE_FALLBACK
          bad16[x][rt] = has_keys;
#endif
        });
      });
    } else {
    sfor<4>([&](auto PART) __attribute__((always_inline)) { hook(PART); });
    sfor<2>([&](auto SB) __attribute__((always_inline)) {
      constexpr int sb = decltype(SB)::value;
      float l = ps0[sb] + ps1[sb];
      l += lane_xor32(l);
      int tok_local, hq;
      const bool row_ok = row_of(I, sb, tok_local, hq);
      if (I.lse_base && row_ok && half == 0)    // P = exp2(score): ln of the sum is the lse
        I.lse_base[(int64_t)(I.q_start + tok_local) * sa.lse_st + hq] = l > 0.0f ? __builtin_amdgcn_logf(l) * 0.6931471805599453f : -INFINITY;
      const float inv = (row_ok && l > 0.0f) ? __builtin_amdgcn_rcpf(l) : 0.0f;   // (1 ulp: the output is rounded to 16 bits next)
      float amax = 0.0f;
      if (__builtin_expect(wide_store, 1)) {
        // the sub-block leaves as whole 256-byte rows, 16 bytes per lane (a lane's own 8-byte pieces touch 32 rows per
        // store): parked in this wave's rows, read back by other lanes (LDS is in order per wave, also against the next
        // sub-block's writes)
        sfor<4>([&](auto B) __attribute__((always_inline)) {
          sfor<4>([&](auto C) __attribute__((always_inline)) {
            constexpr int b = decltype(B)::value, c = decltype(C)::value, base = kAO + 64 * sb + 16 * b + 4 * c;
            const float o0 = acc_read<base>(), o1 = acc_read<base + 1>(), o2 = acc_read<base + 2>(), o3 = acc_read<base + 3>();
            pw_amax3(amax, o0, o1);
            pw_amax3(amax, o2, o3);
            *(wu32x2_t*)(ost + qr * kPwORS + (32 * b + 8 * c + 4 * half) * 2) = wu32x2_t{pw_pack<T>(o0 * inv, o1 * inv), pw_pack<T>(o2 * inv, o3 * inv)};
          });
        });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        wu32x4_t rows[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) rows[j] = *(const wu32x4_t*)(ost + (4 * j + orow) * kPwORS + och * 16);
        typedef __attribute__((address_space(1))) wu32x4_t* grow_t;
        auto row_off = [&](int j, uint32_t& tq) {
          const uint32_t m = (uint32_t)(wave * 64 + sb * 32 + 4 * j + orow);
          tq = __umul24(m, g_inv) >> 16;      // token and query head of row m inside the block
          return __umul24(tq, st_b) + __umul24(m - __umul24(tq, (uint32_t)G), sh_b) + (uint32_t)och * 16u;   // (full-rate 24-bit multiplies: strides are below 2^22 elements)
        };
        if (__builtin_expect(whole_block, 1)) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { uint32_t tq; const uint32_t off = row_off(j, tq); __builtin_nontemporal_store(rows[j], (grow_t)(out0 + off)); }
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) { uint32_t tq; const uint32_t off = row_off(j, tq); if (tq < tok_left) __builtin_nontemporal_store(rows[j], (grow_t)(out0 + off)); }
        }
      } else {
        uint16_t* op = I.out_base + (int64_t)(I.q_start + tok_local) * (int64_t)sa.out_st + (int64_t)hq * (int64_t)sa.out_sh + 4 * half;
        sfor<4>([&](auto B) __attribute__((always_inline)) {
          sfor<4>([&](auto C) __attribute__((always_inline)) {
            constexpr int b = decltype(B)::value, c = decltype(C)::value, base = kAO + 64 * sb + 16 * b + 4 * c;
            const float o0 = acc_read<base>(), o1 = acc_read<base + 1>(), o2 = acc_read<base + 2>(), o3 = acc_read<base + 3>();
            pw_amax3(amax, o0, o1);
            pw_amax3(amax, o2, o3);
            if (row_ok) *(wu32x2_t*)(op + 32 * b + 8 * c) = wu32x2_t{pw_pack<T>(o0 * inv, o1 * inv), pw_pack<T>(o2 * inv, o3 * inv)};
          });
        });
      }
      amax = fmaxf(amax, lane_xor32(amax));
      const bool has_keys = row_ok && I.tile_hi > I.tile_lo && row_lim(I, sb) >= key_lo;
      bad[sb] = has_keys && !(l >= kPwSumLo && l <= kPwSumHi && amax < INFINITY);   // a NaN sum fails both comparisons
#ifdef PW_FORCE_FALLBACK
      bad[sb] = has_keys;                          // diagnostic build: every row through the per-row routine
#endif
    });
    }
    // ---- rows that left the range: computed again, the plain way (never on attention scores as models produce them)
    // (M16: row 32 x + 16 rt + r16's flag sits in lane r16 of every lane group; packed into the same two 32-bit masks)
    const unsigned long long bad_a = M16 ? ((__ballot(bad16[0][0]) & 0xffffull) | ((__ballot(bad16[0][1]) & 0xffffull) << 16)) : __ballot(bad[0]);
    const unsigned long long bad_b = M16 ? ((__ballot(bad16[1][0]) & 0xffffull) | ((__ballot(bad16[1][1]) & 0xffffull) << 16)) : __ballot(bad[1]);
    if (__builtin_expect((bad_a | bad_b) != 0, 0)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores above are done before these rows are written again
      for (int rr = 0; rr < 64; ++rr) {
        const unsigned long long bits = rr < 32 ? bad_a : bad_b;
        if (!((bits >> (rr & 31)) & 1)) continue;             // lane rr & 31 holds the row's flag (both half-waves agree)
        const int m = wave * 64 + rr;
        const int tok = I.tok0 + div_g(m);
        const int key_hi = min(min(I.ctx_len + tok, I.seq_len - 1) + 1, I.tile_hi * kPwTile);
        const int key_lo_row = (SW && sa.window > 0) ? max(key_lo, I.ctx_len + tok - sa.window + 1) : key_lo;
        if (kp->fix_flags) {        // (wave-uniform; several rows of one block write the same byte)
          if (lane == 0) kp->fix_flags[(int64_t)(I.q_start / kp->fix_bq + I.seq + tok / kp->fix_bq) * sa.num_kv_heads + head] = 1;
          continue;
        }
        pw_row_fallback<T, KV8>(kp, (const int32_t*)I.bt64, kbase, vbase, I.q_start + tok, head * G + mod_g(m), key_lo_row, key_hi, I.out_base, I.lse_base, lane, I.ctx_len);
      }
    }
  };

  while (true) {
    reset_state();
    bool drawn = false;
    const int tile_lo = cur.tile_lo, tile_hi = cur.tile_hi;
    // (every load behind the item - its query rows and its first tiles - has landed: zero_o_and_convert_q waited)
    if (tile_hi > tile_lo) {
      widen_first_tiles();           // (fp8 cache: K0 K1 K2 V0 V1 from the fp8 they landed as, each wave its own groups)
      if constexpr (KV8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");
      if constexpr (M16) sfor<16>([&](auto NC) __attribute__((always_inline)) { kread16(NC, ic<0>{}); });
      else sfor<16>([&](auto NC) __attribute__((always_inline)) { kread(NC, ic<0>{}); });
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // slot 0 is re-filled (K(tile_lo + 3)) in the first iteration
      if constexpr (M16) set_references(cur);
      int t = tile_lo;
#ifdef MI355_PW_STAMP
      // diagnostic build only (tools/pw_clock.py): shader cycles and 100 MHz ticks around the tile loop
      unsigned long long st_c0, st_r0;
      asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c0), "=s"(st_r0) :: "memory");
      st_last = (unsigned)st_c0;
      for (int i = 0; i < 6; ++i) st_sum[i] = 0;
#endif
      // Tiles below steady_hi are steady for THIS wave (the bound differs between the waves of a workgroup; both loops
      // have the same barriers, so waves may sit in different ones): unmasked - wholly at or below the wave's first row's
      // last visible key and inside the sequence - with tile t + 4 inside the share and this wave's group of it inside
      // what the Q block can see (then the groups of t + 2, t + 3 are whole). Shifts, not divisions: the terms can be negative.
      const int steady_hi = 1 + min(min(tile_hi - 5 - FO, ((cur.last_group - wave) >> 2) - 4 - FO),
                                    min((cur.ctx_len + cur.w_tok_lo - (kPwTile - 1)) >> 6, (cur.seq_len >> 6) - 1));
      // The workgroup walks tile_hi tiles, but a wave's 64 rows see no key past their last row's limit: the tiles from
      // own_hi on are wholly masked for it (G < 4: up to three of a Q block's last four). It computes nothing for
      // them - it only keeps staging its share of the K/V tiles the other waves still need.
      const int own_hi = cur.w_tok_lo >= cur.q_len ? tile_lo : max(tile_lo, min(tile_hi, (min(cur.ctx_len + cur.w_tok_hi, cur.seq_len - 1) >> 6) + 1));
      // Sliding window: the tiles below the window of the wave's LAST row are general iterations too (lower bound in the
      // mask); they come first, in whole rounds of three so that the steady stretch starts at ring phase 0. One copy of
      // the general loop serves both ends: pass 0 = the window's lower edge, pass 1 = steady stretch + the upper end.
      int pre_hi = tile_lo;
      if constexpr (SW) {
        if (sa.window > 0) {
          const int s_lo = (cur.ctx_len + cur.w_tok_hi - sa.window + 1 + (kPwTile - 1)) >> 6;   // first tile wholly inside every row's window
          pre_hi = tile_lo + 3 * ((max(s_lo - tile_lo, 0) + 2) / 3);
        }
      }
#pragma nounroll
      for (int pass = SW ? 0 : 1; pass < 2; ++pass) {
        if (pass == 1) {
          while (t + 3 <= steady_hi) {                 // three at a time: leaves the ring phase at 0
            iteration(ic<0>{}, ic<1>{}, t);
            iteration(ic<1>{}, ic<1>{}, t + 1);
            iteration(ic<2>{}, ic<1>{}, t + 2);
            t += 3;
          }
          if (dynamic) { draw(); drawn = true; }      // the next item's ticket: a few tiles before this item ends
#ifndef PW_NO_TAIL2
          if constexpr (M16) {
            // unmasked for this wave, but too close to the end of the share or the sequence for the steady form's fetches
            const int unmasked_hi = min(own_hi, 1 + min((cur.ctx_len + cur.w_tok_lo - (kPwTile - 1)) >> 6, (cur.seq_len >> 6) - 1));
            while (t + 3 <= unmasked_hi) {
              iteration(ic<0>{}, ic<2>{}, t);
              iteration(ic<1>{}, ic<2>{}, t + 1);
              iteration(ic<2>{}, ic<2>{}, t + 2);
              t += 3;
            }
          }
#endif
        }
        const int lim = pass == 0 ? min(pre_hi, own_hi) : own_hi;
        while (t < lim) {
          iteration(ic<0>{}, ic<0>{}, t);
          if (++t >= lim) break;
          iteration(ic<1>{}, ic<0>{}, t);
          if (++t >= lim) break;
          iteration(ic<2>{}, ic<0>{}, t);
          ++t;
        }
      }
#ifdef MI355_PW_STAMP
      {
        unsigned long long st_c1, st_r1;
        asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_c1), "=s"(st_r1) :: "memory");
        unsigned long long* dbg = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0);
        if (dbg && tid == 192) {
          unsigned long long* rec = dbg + 12ull * cur.rank;
          rec[0] = st_c1 - st_c0; rec[1] = st_r1 - st_r0; rec[2] = (unsigned long long)(tile_hi - tile_lo);
#ifndef MI355_PW_SEAM
          for (int i = 1; i < 6; ++i) rec[2 + i] = st_sum[i];
#else
          rec[7] = (unsigned long long)(st_sum[0] - (unsigned)st_c0);
#endif
          rec[8] = st_entry; rec[9] = st_r0; rec[10] = st_r1;
        }
      }
#endif
      // drain: sub-block B of the wave's last tile
      if (own_hi > tile_lo) {
        if constexpr (M16) {
          asm volatile("s_nop 1");           // (B's exponentials and packs ended with its iteration)
          sfor<36>([&](auto GC) __attribute__((always_inline)) { pv16(ic<1>{}, GC); });
        } else {
          sfor<16>([&](auto GC) __attribute__((always_inline)) { estream(ic<1>{}, ic<12 + decltype(GC)::value>{}); });
          asm volatile("s_nop 1");
          sfor<16>([&](auto GC) __attribute__((always_inline)) { pv(ic<1>{}, GC); });
        }
      }
      // the tiles this wave only stages for the others: the memory side of an iteration, ring slots at run time
      for (; t < tile_hi; ++t) {
        const int ph = (t - tile_lo) % 3;
        const uint32_t kd = (uint32_t)(kLdsK + ph * kSlotBytes) + lds_wave, vd = (uint32_t)(kLdsV + ((ph + 2) % 3) * kSlotBytes) + lds_wave;
        tail_check(cur, t + 3 + FO, ic<0>{});
        tail_check(cur, t + 2 + FO, ic<1>{});
        const uint64_t kb64 = group_base(cur, t + 3 + FO, pg_k, ic<0>{}), vb64 = group_base(cur, t + 2 + FO, pg_v, ic<1>{});
        pg_v = pg_k;
        pg_k = *(const __attribute__((address_space(4))) int*)(cur.bt64 + (uint32_t)entry_off(cur, t + 4 + FO));   // (an ordinary load: see setup)
        if constexpr (KV8) {       // the staging area's turn-over (see iteration), ring slots at run time
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          wu32x4_t s0 = pw_lds_read128<0>(st_rd), s1 = pw_lds_read128<1024>(st_rd), s2 = pw_lds_read128<2048>(st_rd), s3 = pw_lds_read128<3072>(st_rd);
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3) :: "memory");
#pragma unroll
          for (int j = 0; j < 2; ++j) pw_glds16(koff[j], kb64, kLdsS + lds_wave + j * 1024);
#pragma unroll
          for (int j = 0; j < 2; ++j) pw_glds16(voff[j], vb64, kLdsS + 2048 + lds_wave + j * 1024);
          const uint32_t ko = (uint32_t)(ph * kSlotBytes), vo = (uint32_t)(((ph + 2) % 3) * kSlotBytes);
          wu32x4_t c0, c1;
          widen16(s0, c0, c1); pw_lds_write128_rt(kw8[0][0] + ko, c0); pw_lds_write128_rt(kw8[0][1] + ko, c1);
          widen16(s1, c0, c1); pw_lds_write128_rt(kw8[1][0] + ko, c0); pw_lds_write128_rt(kw8[1][1] + ko, c1);
          widen16(s2, c0, c1); pw_lds_write128_rt(vw8 + vo, c0); pw_lds_write128_rt(vw8 + vo + 16, c1);
          widen16(s3, c0, c1); pw_lds_write128_rt(vw8 + vo + 2048, c0); pw_lds_write128_rt(vw8 + vo + 2048 + 16, c1);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" : "+s"(pg_k) :: "memory");
          continue;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pw_glds16(koff[j], kb64, kd + j * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) pw_glds16(voff[j], vb64, vd + j * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_waitcnt vmcnt(8)\n\ts_barrier" : "+s"(pg_k) :: "memory");
      }
    }

    // ---- seam: the next item's loads go out, then this item's output ----------------------------------------
#ifdef MI355_PW_SEAM
#define PW_SEAM_STAMP(i) do { unsigned long long st_x; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_x) :: "memory"); \
    unsigned long long* dbg_ = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0); \
    if (dbg_ && tid == 192) dbg_[12ull * cur.rank + (i)] = st_x; } while (0)
#else
#define PW_SEAM_STAMP(i) do { } while (0)
#endif
    PW_SEAM_STAMP(3);
    refresh_lane();
    Item nxt;
    if (dynamic) {
      if (!drawn) draw();          // (an item without tiles)
      publish();
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_waitcnt vmcnt(0)" ::: "memory");   // last MFMA results readable; no LDS-DMA of this item in flight
    __syncthreads();                                                          // every wave is done with the rings
    PW_SEAM_STAMP(4);
    const bool more = dynamic ? acquire(nxt, read_published(), true) : acquire(nxt, 0, false);
    PW_SEAM_STAMP(5);
    PW_SEAM_STAMP(6);
    epilogue(cur, [&](auto PART) __attribute__((always_inline)) { if (more) next_item_loads(nxt, PART); });
#ifdef MI355_PW_STAMP
    {
      unsigned long long st_exit;
      asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_exit) :: "memory");
      unsigned long long* dbg = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0);
      if (dbg && tid == 192) dbg[12ull * cur.rank + 11] = st_exit;
      st_entry = st_exit;
    }
#endif
    if (!more) break;
    zero_o_and_convert_q(nxt, true);
    cur = nxt;
  }
  finish();
}


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The kernel's nineteen instantiations are built in THREE translation units of this one source (hipcc spends ~12 s on
// each): prefill_pw.hip itself (PW_TU 0: the plain and sliding-window kernels of both dtypes, the 32x32x16 form, and all
// host code), prefill_pw_feat.hip (PW_TU 1: soft-cap, ALiBi) and prefill_pw_heads.hip (PW_TU 2: head sizes 64 / 80 / 96),
// which define PW_TU and include this file.
#ifndef PW_TU
#define PW_TU 0
#endif

template <typename K>
static int pw_go(K kernel, const PwArgs& a, int num_kv_heads, hipStream_t stream, std::atomic<uint64_t>& opted, int lds = kPwLds) {
  const int rc1 = ensure_dynamic_lds((const void*)kernel, lds, opted, "hipFuncSetAttribute(prefill_pw)");
  if (rc1 != MI355_OK) return rc1;
  hipLaunchKernelGGL(kernel, dim3(a.slots * num_kv_heads), dim3(256), (size_t)lds, stream, a);
  return MI355_OK;
}
template <typename T> int launch_pw_fp8(const PwArgs& a, int num_kv_heads, bool e5m2, hipStream_t stream);                       // prefill_pw_fp8.hip
template <typename T> int launch_pw_feat(const PwArgs& a, int num_kv_heads, bool sw, bool sc, bool al, hipStream_t stream);      // prefill_pw_feat.hip
template <typename T> int launch_pw_heads(const PwArgs& a, int num_kv_heads, int head_size, bool sw, hipStream_t stream);       // prefill_pw_heads.hip

#if PW_TU == 1
template <typename T> int launch_pw_feat(const PwArgs& a, int num_kv_heads, bool sw, bool sc, bool al, hipStream_t stream) {
  static std::atomic<uint64_t> o_al{0}, o_scw{0}, o_sc{0};
  if (al) return pw_go(prefill_pw_kernel<T, true, false, false, true>, a, num_kv_heads, stream, o_al);
  if (sc && sw) return pw_go(prefill_pw_kernel<T, true, true, true>, a, num_kv_heads, stream, o_scw);
  return pw_go(prefill_pw_kernel<T, true, false, true>, a, num_kv_heads, stream, o_sc);
}
template int launch_pw_feat<bf16_t>(const PwArgs&, int, bool, bool, bool, hipStream_t);
template int launch_pw_feat<f16_t>(const PwArgs&, int, bool, bool, bool, hipStream_t);
#endif

#if PW_TU == 3
template <typename T> int launch_pw_fp8(const PwArgs& a, int num_kv_heads, bool e5m2, hipStream_t stream) {
  static std::atomic<uint64_t> o4{0}, o5{0};
  if (e5m2) return pw_go(prefill_pw_kernel<T, true, false, false, false, 128, 2>, a, num_kv_heads, stream, o5, kPwLdsF8);
  return pw_go(prefill_pw_kernel<T, true, false, false, false, 128, 1>, a, num_kv_heads, stream, o4, kPwLdsF8);
}
template int launch_pw_fp8<bf16_t>(const PwArgs&, int, bool, hipStream_t);
template int launch_pw_fp8<f16_t>(const PwArgs&, int, bool, hipStream_t);
#endif

#if PW_TU == 2
template <typename T> int launch_pw_heads(const PwArgs& a, int num_kv_heads, int head_size, bool sw, hipStream_t stream) {
  static std::atomic<uint64_t> o64{0}, o80{0}, o96{0}, o96w{0};
  if (head_size == 64) return pw_go(prefill_pw_kernel<T, true, false, false, false, 64>, a, num_kv_heads, stream, o64);
  if (head_size == 80) return pw_go(prefill_pw_kernel<T, true, false, false, false, 80>, a, num_kv_heads, stream, o80);
  if (sw) return pw_go(prefill_pw_kernel<T, true, true, false, false, 96>, a, num_kv_heads, stream, o96w);
  return pw_go(prefill_pw_kernel<T, true, false, false, false, 96>, a, num_kv_heads, stream, o96);
}
template int launch_pw_heads<bf16_t>(const PwArgs&, int, int, bool, hipStream_t);
template int launch_pw_heads<f16_t>(const PwArgs&, int, int, bool, hipStream_t);
#endif

#if PW_TU == 0

// Preconditions beyond prefill_supported(): bf16 or f16, head size 128, a cache of the query type, G <= 256; a sliding
// window and soft-cap are served, and ALiBi without either (with them: prefill_mfma_kernel's FEAT instantiation).
bool prefill_pw_applicable(const mi355_attn_params& p) {
  // (soft-cap and ALiBi are served since round 3 - the SC and AL instantiations - but ALiBi only by itself)
  const bool feat = p.alibi_slopes != nullptr && (p.softcap > 0.0f || p.sliding_window > 0);
  const int G = p.num_q_heads / p.num_kv_heads;
  // (rows of a Q block are addressed as 32-bit byte offsets from the block's first row: strides below 2^22 elements)
  const int64_t lim = (int64_t)1 << 22;
  const bool strides_ok = p.q_stride_token >= 0 && p.q_stride_token < lim && p.q_stride_head >= 0 && p.q_stride_head < lim &&
                          p.out_stride_token >= 0 && p.out_stride_token < lim && p.out_stride_head >= 0 && p.out_stride_head < lim;
  // (D = 64 / 96: plain; 96 - Phi-3's head size - also with a sliding window)
  const bool d_ok = p.head_size == 128 || ((p.head_size == 64 || p.head_size == 80 || p.head_size == 96) && p.softcap == 0.0f && !p.alibi_slopes && (p.sliding_window <= 0 || p.head_size == 96));
  // an fp8 cache (round 4, the KV8 instantiations: widened inside the kernel, reference :434-455): plain attention at head size 128
  const bool fp8 = p.kv_dtype == MI355_FP8_E4M3 || p.kv_dtype == MI355_FP8_E5M2;
  const bool kv_ok = p.kv_dtype == p.q_dtype || (fp8 && p.head_size == 128 && p.softcap == 0.0f && !p.alibi_slopes && p.sliding_window <= 0);
  return !feat && strides_ok && d_ok && G <= kPwRows && (p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16) && kv_ok;
}

template <typename T>
static int launch_pw_t(const mi355_attn_params& p, int key_splits, int64_t out_split_stride, int64_t lse_split_stride, int* counters, hipStream_t stream,
                       uint8_t* fix_flags) {
  PwArgs a;
  a.p = p;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.block_q = kPwRows / a.group;
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.g_inv = (65536 + a.group - 1) / a.group;
  a.bq_shift = (a.block_q & (a.block_q - 1)) == 0 ? __builtin_ctz((unsigned)a.block_q) : -1;
  a.key_splits = key_splits;
  a.fix_flags = (fix_flags && a.group <= 128) ? fix_flags : nullptr;
  a.fix_bq = std::max(1, 128 / a.group);
  a.out_split_stride = out_split_stride;
  a.lse_split_stride = lse_split_stride;
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  // static upper bound, as the reference (:886-889,:935-943); with ONE sequence the bound is exact without the "+ 1" (its
  // Q blocks are those of num_tokens), and an empty item at the head of the list would shift the boustrophedon deal by one:
  // slot 0 would end up with 3 tiles and every other slot with 67 instead of 65 (1 x 4096)
  a.num_qblocks = p.num_seqs == 1 ? (p.num_tokens + a.block_q - 1) / a.block_q : p.num_tokens / a.block_q + p.num_seqs;
  // one workgroup per CU (it holds exactly one), in whole sets of num_kv_heads; fewer when there are fewer items
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) {
    static std::atomic<int> cu_count[64];
    int c = dev < 64 ? cu_count[dev].load(std::memory_order_relaxed) : 0;
    if (c == 0 && hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && c > 0 && dev < 64)
      cu_count[dev].store(c, std::memory_order_relaxed);
    if (c > 0) cus = c;
  }
  static const int slots_env = [] { const char* e = lab_env("MI355_PW_SLOTS"); return e ? atoi(e) : 0; }();   // measurements only
  const int items_per_head = a.num_qblocks * key_splits;
  // About one workgroup per CU, each walking several items. ONE sequence: the static boustrophedon deal - balanced by
  // construction for causal weights that fall linearly along the list, and better than any greedy deal (1 x 4096: 112.5 us;
  // tickets 115.0; one item per workgroup, dealt by the hardware, 118.5; 1 x 16384: 1574 vs 1587). SEVERAL sequences
  // make the weights a sawtooth, and a static deal loses more than the seams gain (4 x 2048: 143.5 us): there the items
  // are dealt by ticket counters in the zero-filled head of the caller's workspace (2 x 4096: 224.8 us vs 231.6 with one
  // item per workgroup, 16 x 4096 1835 vs 1863, 8 x 2048 262.9 vs 268.7), and without a workspace one item per workgroup.
  static const bool tickets_off = [] { const char* e = lab_env("MI355_PW_TICKETS"); return e && e[0] == '0'; }();   // measurements only
  a.tickets = (counters && !tickets_off && p.num_seqs > 1 && 2 * p.num_kv_heads * sizeof(int) <= kPwCounterBytes) ? counters : nullptr;
  const int per_cu = std::max(1, cus / p.num_kv_heads);
  a.slots = std::max(1, std::min(items_per_head, slots_env > 0 ? slots_env : ((a.tickets || p.num_seqs == 1) ? per_cu : items_per_head)));
  const size_t lds = kPwLds;
  // Which matrix instruction: the 16x16x32 instantiation (the product) draws less power per flop - the chip holds
  // ~2.2-2.3 GHz under it instead of ~2.0-2.1 - at the price of twice as many matrix instructions to issue; with its
  // exponentials dealt one per gap it is ahead on every shape this kernel is chosen for (same box: 1 x 4096 +6.9 %,
  // 1 x 16384 +8.8 %, 16 x 4096 +5 %, 4 x 2048 +3.2 %). MI355_PW_M16=0 pins the 32x32x16 instantiation (A/B, tests).
  static const bool m16_env = [] { const char* e = lab_env("MI355_PW_M16"); return !(e && e[0] == '0'); }();
  const bool sw = p.sliding_window > 0, sc = p.softcap > 0.0f, al = p.alibi_slopes != nullptr;
  const bool d64 = p.head_size == 64, d96 = p.head_size == 96, d80 = p.head_size == 80;
  const bool m16 = m16_env || sw || sc || al || d64 || d96 || d80 || !__is_same(T, bf16_t);      // the 32x32x16 instantiation: bf16, D = 128, no window, no soft-cap, no ALiBi
  int rc_l = MI355_OK;
  const bool fp8 = p.kv_dtype == MI355_FP8_E4M3 || p.kv_dtype == MI355_FP8_E5M2;
  if (fp8) {
    rc_l = launch_pw_fp8<T>(a, p.num_kv_heads, p.kv_dtype == MI355_FP8_E5M2, stream);
  } else if (d64 || d80 || d96) {
    rc_l = launch_pw_heads<T>(a, p.num_kv_heads, p.head_size, sw, stream);
  } else if (al || sc) {
    rc_l = launch_pw_feat<T>(a, p.num_kv_heads, sw, sc, al, stream);
  } else if (m16 && sw) {
    static std::atomic<uint64_t> o{0};
    rc_l = pw_go(prefill_pw_kernel<T, true, true>, a, p.num_kv_heads, stream, o);
  } else if (m16) {
    static std::atomic<uint64_t> o{0};
    rc_l = pw_go(prefill_pw_kernel<T, true, false>, a, p.num_kv_heads, stream, o);
  } else if constexpr (__is_same(T, bf16_t)) {
    static std::atomic<uint64_t> o{0};
    rc_l = pw_go(prefill_pw_kernel<T, false, false>, a, p.num_kv_heads, stream, o);
  }
  if (rc_l != MI355_OK) return rc_l;
  const int rc = check_hip(hipGetLastError(), "prefill_pw_kernel launch");
  if (rc == MI355_OK) set_kernel_name(fp8 ? "prefill_mfma_pw_fp8" : al ? "prefill_mfma_pw_al" : sc ? (sw ? "prefill_mfma_pw_sw_sc" : "prefill_mfma_pw_sc") : sw ? "prefill_mfma_pw_sw" : "prefill_mfma_pw");
  return rc;
}

int launch_prefill_pw(const mi355_attn_params& p, int key_splits, int64_t out_split_stride, int64_t lse_split_stride, int* counters, hipStream_t stream,
                      uint8_t* fix_flags) {
  if (p.q_dtype == MI355_F16) return launch_pw_t<f16_t>(p, key_splits, out_split_stride, lse_split_stride, counters, stream, fix_flags);
  return launch_pw_t<bf16_t>(p, key_splits, out_split_stride, lse_split_stride, counters, stream, nullptr);
}
#endif   // PW_TU == 0

}  // namespace mi355
