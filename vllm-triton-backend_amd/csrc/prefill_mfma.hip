// Chunked / context prefill attention over the paged KV cache for gfx950.
//
// Replaces kernel_unified_attention_2d (LIB/kernels/triton_unified_attention.py:275-523): causal
// attention of a "Q block" (BLOCK_Q consecutive query tokens x all G query heads of one KV head)
// against keys [0, context_len + position], KV fetched through the block table. MFMA-bound.
//
// Structure (one workgroup = 4 waves = 128 Q-block rows, each wave 32 rows; KV tile = 64 keys):
//   * rows are ordered (token, head-in-group) exactly like the reference (offs_m // G, offs_m % G,
//     :343-346), so all G heads of a KV head share every K/V tile (GQA broadcast through LDS);
//   * K/V tiles are fetched HBM -> VGPR with row-shaped 16-byte loads one tile ahead of their use
//     (issued before the MFMA phase, written to LDS after it) and shared by the 4 waves via LDS:
//     K rows padded to 2D+16 bytes (conflict-free ds_read_b128 A-operand reads), V rows padded to
//     2D+64 bytes (conflict-free ds_read_b64_tr_b16 transposed reads);
//   * S^T = K.Q^T with v_mfma_f32_32x32x16 ("swapped" product): lane (q = lane&31, half = lane>>5)
//     then owns one query row: its 16+16 accumulator registers are 32 of the tile's 64 keys, so the
//     online softmax is in-register (one v_permlane32_swap per tile for the row max) and m, l, alpha
//     are lane-local;
//   * O^T += V^T.P^T: the bf16-packed accumulator registers of S^T are directly the B operand
//     (k-slot j of half h = key 16s + 8(j>>2) + 4h + (j&3)); the transposed V read is addressed to
//     deliver the same key order, so P never moves between lanes or through LDS;
//   * softmax in the exp2 domain with scale*log2(e) folded in; causal/window masks only on tiles
//     that straddle a boundary; the KV loop stops at the Q block's last visible key (causal tile
//     skipping, :384-400) and starts at the sliding window's first visible tile (the reference
//     only masks).
// Launch grid mirrors the reference's static upper bound (T / BLOCK_Q + S Q-blocks, :886-889,
// :935-943), heaviest (latest) Q blocks first.
#include <cstdlib>
#include <type_traits>
#include <utility>
#include <cstring>

#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 pbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 pf16x8_t;
typedef __attribute__((ext_vector_type(4))) short ps16x4_t;
typedef __attribute__((ext_vector_type(8))) short ps16x8_t;
typedef __attribute__((ext_vector_type(16))) float pf32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int pu32x4_t;
typedef __attribute__((ext_vector_type(2))) float pf32x2_t;
typedef __attribute__((ext_vector_type(2))) unsigned int pu32x2_t;

constexpr int kBlockM = 128;   // Q-block rows per workgroup
constexpr int kTileN = 64;     // keys per KV tile
constexpr float kLog2eP = 1.4426950408889634f;

struct PrefillArgs {
  mi355_attn_params p;
  int group;      // G
  int block_q;    // tokens per Q block = kBlockM / G
  int page_shift; // log2(page_size)
  int d_valid;    // the real head size; columns d_valid .. D-1 of the kernel's head size are padding
  int kv_same_strides;  // K and V cache strides are equal (two views of one tensor)
  // key-split launch of prefill_mfma_kernel (grid.y = key_splits): workgroup (x, s) attends the s-th of key_splits
  // even shares of its Q block's key tiles only and writes its normalised output and lse to the
  // partial buffers p.out + s * out_split_stride / p.lse + s * lse_split_stride; merge_key_splits_kernel folds them
  int key_splits, tiles_per_key_split;
  int64_t out_split_stride, lse_split_stride;
  uint32_t k_page_stride, k_slot_stride, v_page_stride, v_slot_stride;  // elements; validated < 2^31 on the host
  // fix-up launch behind an f16 prefill_pw_kernel (common.h, kWsFixFlagOffset): one byte per (Q block, KV head); a workgroup
  // whose byte is 0 leaves at once, the others compute their block again (all key splits of it) with a true running
  // maximum. A launch without key splits clears the bytes it serves; with key splits merge_key_splits_kernel does.
  uint8_t* only_flagged;
};

template <typename T> struct pmma;
template <> struct pmma<bf16_t> {
  static __device__ __forceinline__ pf32x16_t run(ps16x8_t a, ps16x8_t b, pf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(pbf16x8_t, a), __builtin_bit_cast(pbf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return pack_bf16x2(lo, hi);
  }
};
template <> struct pmma<f16_t> {
  static __device__ __forceinline__ pf32x16_t run(ps16x8_t a, ps16x8_t b, pf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(pf16x8_t, a), __builtin_bit_cast(pf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return pack_f16x2(lo, hi);
  }
};

// f(integral_constant<0>) ... f(integral_constant<N-1>), in order
template <typename F, int... I>
__device__ __forceinline__ void pm_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void pm_for(F&& f) { pm_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// Head size 256 (round 3): O^T (128 registers per lane) and the Q fragments (64) live in accumulator registers that
// only these statements name - a[0..127] and a[128..191], literal operands the register allocator never sees. As C++
// values hipcc could not give them homes: with the 64 staging registers beside them it spilled ~30 VGPRs and moved several
// hundred values per tile between VGPRs and accumulator registers (440-520 TFLOP/s). The rest of the kernel (S, P, the
// K / V fragments, staging) stays ordinary code and fits the 256 VGPRs; tests/test_cpu_host.py audits the build.
template <typename T> struct pmma_own;
#define MI355_DEF_PMMA_OWN(TAG, MFMA)                                                                              \
  template <> struct pmma_own<TAG> {                                                                              \
    /* S(VGPR) += K(VGPR) . Q(AGPR) */                                                                             \
    template <int QA> static __device__ __forceinline__ void qk(pf32x16_t& s, const pu32x4_t& k) {                \
      asm volatile(MFMA " %0, %1, a[%c2:%c3], %0" : "+v"(s) : "v"(k), "n"(QA), "n"(QA + 3));                       \
    }                                                                                                             \
    /* O(AGPR) += V(VGPR) . P(VGPR) */                                                                             \
    template <int OA> static __device__ __forceinline__ void pv(const ps16x8_t& v, const ps16x8_t& pf) {          \
      asm volatile(MFMA " a[%c2:%c3], %0, %1, a[%c2:%c3]" :: "v"(v), "v"(pf), "n"(OA), "n"(OA + 15));              \
    }                                                                                                             \
  };
MI355_DEF_PMMA_OWN(bf16_t, "v_mfma_f32_32x32x16_bf16")
MI355_DEF_PMMA_OWN(f16_t, "v_mfma_f32_32x32x16_f16")
#undef MI355_DEF_PMMA_OWN
template <int IDX> __device__ __forceinline__ void pm_acc_write(uint32_t v) { asm volatile("v_accvgpr_write_b32 a%c0, %1" :: "n"(IDX), "v"(v)); }
template <int IDX> __device__ __forceinline__ float pm_acc_read() { float r; asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(r) : "n"(IDX)); return r; }
constexpr int kOwnO = 0, kOwnQ = 128;      // accumulator-register map of the hand-owned form

// largest i with cu[i] / block_q + i <= qblock (reference: find_seq_idx in Q-block mode, :32-52)
__device__ __forceinline__ int find_seq_by_qblock(const int32_t* __restrict__ cu, int num_seqs, int qblock, int block_q) {
  int left = 0, right = num_seqs;
  while (left < right) {
    const int mid = (left + right) >> 1;
    if (cu[mid] / block_q + mid <= qblock) left = mid + 1; else right = mid;
  }
  return left - 1;
}

// 16 fp8 values (one 16-byte load) -> 16 values of the query's 16-bit type (two 16-byte LDS pieces); exact
template <typename T, typename KVT>
__device__ __forceinline__ void widen_fp8_piece(pu32x4_t in, pu32x4_t& lo, pu32x4_t& hi) {
  uint32_t o[8];
#pragma unroll
  for (int w = 0; w < 4; ++w) widen_fp8x4<T, KVT>(in[w], o[2 * w], o[2 * w + 1]);
  lo = pu32x4_t{o[0], o[1], o[2], o[3]};
  hi = pu32x4_t{o[4], o[5], o[6], o[7]};
}

// FEAT = soft-cap / ALiBi / sliding window compiled in; the plain instantiation only knows the
// causal + sequence-length mask. KVT = T, or an fp8 type: the cache tile is widened to T on its way
// from the staging registers into LDS (reference dequant `(fp8 -> f32) * scale -> Q dtype`, :434-455;
// the scalar k scale is folded into the softmax scale, the v scale into the output normalisation).
// D = 256 needs the whole register file (O^T alone is 128 accumulator registers per lane) and 141 KiB of
// LDS for its two stages: one workgroup per CU.
template <typename T, typename KVT, int D, bool FEAT>
__global__ __launch_bounds__(256, D <= 128 ? 2 : 1) void prefill_mfma_kernel(const PrefillArgs a) {
  constexpr bool FP8 = !__is_same(T, KVT);
  constexpr int KVB = FP8 ? 1 : 2;           // bytes per cache element
  constexpr int EPP = 16 / KVB;              // elements per 16-byte piece
  constexpr int PPR = D / EPP;               // 16-byte pieces per key row
  constexpr int NLD = PPR / 4;               // loads per thread per tile (64 keys * PPR pieces / 256 threads)
  static_assert(NLD >= 1, "head size too small for the staging pattern");
  constexpr int RSK = D * 2 + 16;            // K row stride in LDS (bytes)
  constexpr int RSV = D * 2 + 64;            // V row stride in LDS (bytes)
  constexpr int KSTEPS = D / 16;             // k-steps of K.Q^T
  constexpr int DBLK = D / 32;               // 32-wide output blocks of P.V

  constexpr int KBUF = kTileN * RSK, VBUF = kTileN * RSV, BUF = KBUF + VBUF;   // one stage: K tile then V tile
  extern __shared__ __attribute__((aligned(16))) char smem[];                    // two stages
  constexpr bool AOWN = D >= 256;          // O^T and Q in hand-owned accumulator registers (see pmma_own)
  if constexpr (AOWN) asm volatile("" ::: "a255");     // the kernel owns all 256 accumulator registers

  const mi355_attn_params& p = a.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.group, BQ = a.block_q;

  // 1-D grid, KV head fastest: workgroups are dealt round-robin to the 8 XCDs, so with Hk = 8 every
  // XCD's L2 serves exactly one KV head's K/V (re-read by all of that head's Q blocks) instead of
  // all of them. Heaviest Q blocks (largest index = longest causal prefix) first.
  // (a fix-up launch deals the Q blocks of ONE head to neighbouring workgroups instead: the flagged blocks are typically
  // all of one query head's - one KV head - and head-fastest would put every one of them on the same XCD, an eighth of the chip)
  const int nqb = (int)(gridDim.x / p.num_kv_heads);
  const int head = a.only_flagged ? (int)(blockIdx.x / nqb) : (int)(blockIdx.x % p.num_kv_heads);
  const int qblock = a.only_flagged ? nqb - 1 - (int)(blockIdx.x % nqb) : nqb - 1 - (int)(blockIdx.x / p.num_kv_heads);
  if (a.only_flagged) {                                    // fix-up launch: only the Q blocks prefill_pw_kernel flagged
    uint8_t* const f = a.only_flagged + (int64_t)qblock * p.num_kv_heads + head;
    if (*(volatile uint8_t*)f == 0) return;
    if (a.key_splits <= 1) {
      __syncthreads();                                     // every wave has read the byte
      if (tid == 0) *f = 0;
    }
  }
  const int seq = find_seq_by_qblock(p.cu_seqlens_q, p.num_seqs, qblock, BQ);
  if (seq < 0) return;
  const int q_start = p.cu_seqlens_q[seq];
  const int q_len = p.cu_seqlens_q[seq + 1] - q_start;
  const int qb_local = qblock - (q_start / BQ + seq);
  if (qb_local * BQ >= q_len) return;                      // surplus program (:338-339)
  if (q_len <= p.skip_decodes) return;                     // (N: rows of sequences with up to N query tokens are another launch's)
  if (p.only_decodes && q_len > p.only_decodes) return;
  const int seq_len = p.seqused_k[seq];
  const int ctx_len = seq_len - q_len;
  const int tok0 = qb_local * BQ;                          // first query token (local) of this Q block

  // ---- this lane's query row -------------------------------------------------------------------
  const int qr = lane & 31, half = lane >> 5;
  const int m_row = wave * 32 + qr;                        // row inside the Q block
  const int tok_local = tok0 + m_row / G;                  // query position inside the sequence's query
  const int hq = head * G + m_row % G;
  const bool row_ok = (m_row < BQ * G) && (tok_local < q_len);
  const int q_abs = ctx_len + tok_local;                   // absolute position; sees keys j <= q_abs

  // wave-uniform bounds of the rows this wave / workgroup owns
  const int w_tok_lo = tok0 + (wave * 32) / G;
  const int w_tok_hi = min(min(tok0 + (wave * 32 + 31) / G, tok0 + BQ - 1), q_len - 1);
  const int wg_tok_hi = min(tok0 + BQ - 1, q_len - 1);
  int n_keys_wg = min(ctx_len + wg_tok_hi + 1, seq_len);   // max_seq_prefix_len (:384-393)
  if (n_keys_wg < 0) n_keys_wg = 0;
  const int wave_keys = min(ctx_len + w_tok_hi + 1, seq_len);  // this wave has nothing to do beyond
  const bool wave_has_rows = w_tok_lo <= w_tok_hi;
  int first_key_wg = 0;
  if (FEAT && p.sliding_window > 0) first_key_wg = max(0, ctx_len + tok0 - p.sliding_window + 1);
  int tile_lo = first_key_wg / kTileN;
  int tile_hi = (n_keys_wg + kTileN - 1) / kTileN;
  // key-split launch: this workgroup's share of the key tiles (possibly empty: it then writes 0 / -inf, which the merge
  // weighs with 0). Masks are functions of absolute positions, so nothing else changes.
  const int ksplit = a.key_splits > 1 ? (int)blockIdx.y : 0;
  if (a.key_splits > 1) {     // an even share of THIS Q block's tiles (the host only knows the longest sequence)
    const int tps = (max(tile_hi - tile_lo, 0) + a.key_splits - 1) / a.key_splits;
    tile_lo = min(tile_lo + ksplit * tps, tile_hi);
    tile_hi = min(tile_hi, tile_lo + tps);
  }
  uint16_t* const out_base = (uint16_t*)p.out + (int64_t)ksplit * a.out_split_stride;
  float* const lse_base = p.lse ? p.lse + (int64_t)ksplit * a.lse_split_stride : nullptr;

  // ---- Q fragments (B operand of S^T = K.Q^T): lane (qr, half) holds Q[row][16ks + 8half .. +7] ----
  ps16x8_t qf[KSTEPS];
  {
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)(q_start + tok_local) * p.q_stride_token + (int64_t)hq * p.q_stride_head + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      pu32x4_t v = {0, 0, 0, 0};
      if (row_ok && 16 * ks + 8 * half < a.d_valid) v = *(const pu32x4_t*)(qp + 16 * ks);
      qf[ks] = __builtin_bit_cast(ps16x8_t, v);
    }
  }
  const float slope = (FEAT && p.alibi_slopes && row_ok) ? p.alibi_slopes[hq] : 0.0f;
  const float k_scale = (FP8 && p.k_scale) ? p.k_scale[0] : 1.0f;
  const float v_scale = (FP8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  const float scale_nat = p.scale * k_scale;   // fp8: K is used un-scaled, its scale moves here
  const float scale2 = scale_nat * kLog2eP;
  const bool plain = !FEAT || (!(p.softcap > 0.0f) && !p.alibi_slopes);
  const float softcap_k = (FEAT && p.softcap > 0.0f) ? 2.0f * kLog2eP / p.softcap : 0.0f;

  // ---- staging: thread t loads pieces t + 256*i of the 64 x PPR tile ------------------------------
  // Load round i of a wave touches exactly one 16-key group, so the page lookup is wave-uniform:
  // scalar loads + scalar address arithmetic, one tile ahead of the vector loads that use it.
  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;
  using kv_elem_t = typename KVT::storage;
  const kv_elem_t* kbase = (const kv_elem_t*)p.k_cache + (int64_t)head * p.k_stride_head;
  const kv_elem_t* vbase = (const kv_elem_t*)p.v_cache + (int64_t)head * p.v_stride_head;
  const int last_group = (max(n_keys_wg, 1) - 1) >> 4;
  const int page_mask = p.page_size - 1;
  // Thread t stages pieces t + 256 i of the 64 x PPR tile. 256 is a multiple of PPR, so piece i of a thread is the
  // same column chunk of key row st_key0 + i*KS: everything below derives from two per-thread values instead of
  // NLD-long arrays (32 registers at D = 256, where the kernel otherwise spills).
  static_assert(256 % PPR == 0, "staging pattern");
  constexpr int KS = 256 / PPR;                                   // key rows between a thread's consecutive pieces
  const int st_key0 = tid / PPR;
  const int st_off0 = (tid % PPR) * EPP;                         // element offset inside the row
  // a piece in the padding columns of a non-built head size loads the row's piece 0 instead (same instruction
  // stream, always a valid address) and is zeroed on its way into LDS
  const bool st_pad0 = st_off0 >= a.d_valid;
  const int st_src_off = st_pad0 ? 0 : st_off0;
  auto st_key_of = [&](int i) { return st_key0 + i * KS; };       // key inside the tile
  auto st_grp_of = [&](int i) { return __builtin_amdgcn_readfirstlane(st_key_of(i) >> 4); };   // its 16-key group (wave-uniform)
  auto k_toff_of = [&](int i) { return (uint32_t)((st_key_of(i) & 15) * (int)p.k_stride_slot + st_src_off); };
  auto v_toff_of = [&](int i) { return (uint32_t)((st_key_of(i) & 15) * (int)p.v_stride_slot + st_src_off); };
  int pg_next[4];                           // physical pages of the NEXT tile's four 16-key groups (SGPRs)
  auto lookup_pages = [&](int tile) {
    int idx[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) idx[g4] = (min(tile * 4 + g4, last_group) << 4) >> a.page_shift;  // stay inside the sequence's pages
    scalar_load4(bt, idx[0], idx[1], idx[2], idx[3], pg_next[0], pg_next[1], pg_next[2], pg_next[3]);
  };
  pu32x4_t kreg[NLD], vreg[NLD];
  auto issue_loads = [&](int tile) {        // uses pg_next, which must hold this tile's pages
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int gi = min(tile * (kTileN / 16) + st_grp_of(i), last_group);
      const int slot0 = (gi << 4) & page_mask;
      int page;
      if constexpr (PPR >= 16) {            // a load round covers one group (D=128) or half of one (D=256)
        page = pg_next[(256 * i) / (16 * PPR)];
      } else if constexpr (PPR == 8) {      // D=64: waves 0-1 / 2-3 of round i take groups 2i / 2i+1
        page = (wave >> 1) ? pg_next[2 * i + 1] : pg_next[2 * i];
      } else {                              // D=32: one group per wave
        page = wave == 0 ? pg_next[0] : wave == 1 ? pg_next[1] : wave == 2 ? pg_next[2] : pg_next[3];
      }
      const kv_elem_t* kp = kbase + ((uint64_t)(uint32_t)page * a.k_page_stride + (uint32_t)slot0 * a.k_slot_stride);
      const kv_elem_t* vp = vbase + ((uint64_t)(uint32_t)page * a.v_page_stride + (uint32_t)slot0 * a.v_slot_stride);
      kreg[i] = *(const pu32x4_t*)(kp + k_toff_of(i));
      vreg[i] = *(const pu32x4_t*)(vp + v_toff_of(i));
    }
  };
  // staging registers -> LDS, K and V separately: at D = 256 the K half is written before the P.V phase (its loads
  // were issued a phase earlier) so that its 32 registers are free while P.V runs - with both halves held to the end
  // of the tile the kernel spills
  auto write_lds_k = [&](char* stage) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu32x4_t kk = kreg[i];
      if (a.d_valid != D) {                   // wave-uniform: the built head sizes skip this
        if (st_pad0) kk = pu32x4_t{0, 0, 0, 0};
      }
      if constexpr (FP8) {
        pu32x4_t lo, hi;
        widen_fp8_piece<T, KVT>(kk, lo, hi);
        *(pu32x4_t*)(stage + st_key_of(i) * RSK + st_off0 * 2) = lo;
        *(pu32x4_t*)(stage + st_key_of(i) * RSK + st_off0 * 2 + 16) = hi;
      } else {
        *(pu32x4_t*)(stage + st_key_of(i) * RSK + st_off0 * 2) = kk;
      }
    }
  };
  auto write_lds_v = [&](int tile, char* stage) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      pu32x4_t v = vreg[i];
      // slots past the sequence hold stale cache contents: keep NaN/Inf out of 0 * V
      if (tile * kTileN + st_key_of(i) >= seq_len) v = pu32x4_t{0, 0, 0, 0};
      if (a.d_valid != D) {
        if (st_pad0) v = pu32x4_t{0, 0, 0, 0};
      }
      if constexpr (FP8) {
        pu32x4_t lo, hi;
        widen_fp8_piece<T, KVT>(v, lo, hi);
        *(pu32x4_t*)(stage + KBUF + st_key_of(i) * RSV + st_off0 * 2) = lo;
        *(pu32x4_t*)(stage + KBUF + st_key_of(i) * RSV + st_off0 * 2 + 16) = hi;
      } else {
        *(pu32x4_t*)(stage + KBUF + st_key_of(i) * RSV + st_off0 * 2) = v;
      }
    }
  };
  auto write_lds = [&](int tile, char* stage) { write_lds_k(stage); write_lds_v(tile, stage); };
  // (head size 256 before round 3: K written early so that its 32 staging registers were free during P.V. With O and Q in
  // accumulator registers there is room, and the early write only cut the loads' flight time to half a tile)
  constexpr bool EARLY_K = false;

  float m_run = -INFINITY, l_run = 0.0f;
  pf32x16_t o_acc[AOWN ? 1 : DBLK];
  if constexpr (AOWN) {
    pm_for<16 * DBLK>([&](auto I) { pm_acc_write<kOwnO + decltype(I)::value>(0u); });
    pm_for<KSTEPS>([&](auto KS) {
      constexpr int ks = decltype(KS)::value;
      const pu32x4_t w = __builtin_bit_cast(pu32x4_t, qf[ks]);
      pm_for<4>([&](auto E) { pm_acc_write<kOwnQ + 4 * ks + decltype(E)::value>(w[decltype(E)::value]); });
    });
  } else {
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[b][r] = 0.0f;
  }

  if (tile_lo < tile_hi) {
    lookup_pages(tile_lo);
    issue_loads(tile_lo);
    if (tile_lo + 1 < tile_hi) lookup_pages(tile_lo + 1);
    write_lds(tile_lo, smem);
  }
  __syncthreads();

  // Make the Q fragments' loads retire HERE: otherwise the compiler's in-loop wait for them
  // (needed on the first iteration only) is a vmcnt(0) that also drains every K/V prefetch.
  if constexpr (!AOWN) {
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(qf[ks]));
  }

  // per-lane offsets of the LDS reads inside a stage
  const int k_rd_off = qr * RSK + half * 16;                                          // + kb*32*RSK + ks*32
  const int gq1 = (lane >> 4) & 1, li = lane & 15;
  const int v_rd_off = KBUF + (4 * half + (li >> 2)) * RSV + (16 * gq1 + 4 * (li & 3)) * 2;  // + sk*16*RSV + db*64

  for (int tile = tile_lo; tile < tile_hi; ++tile) {
    const bool has_next = tile + 1 < tile_hi;
    if (has_next) {
      issue_loads(tile + 1);
      if (tile + 2 < tile_hi) lookup_pages(tile + 2);
    }
    char* stage = smem + ((tile - tile_lo) & 1) * BUF;
    const char* k_rd = stage + k_rd_off;
    const char* v_rd = stage + v_rd_off;

    const int key_base = tile * kTileN;
    bool k_written = false;                    // wave-uniform
    if (wave_has_rows && key_base < wave_keys) {
      // ---- S^T = K . Q^T ---------------------------------------------------------------------------
      // All K fragments of a 32-key block are requested before its first MFMA, and the next block's
      // reads are issued behind the MFMAs that free their registers, so the matrix pipe never waits
      // for a single ds_read round trip (sched_group_barrier pins that order).
      pf32x16_t s_acc[2];
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] = 0.0f;
      // a ring of KW fragments: the whole 32-key block for D <= 128; for D = 256 half of it, or the 64 fragment
      // registers push the kernel over the register file (it spilled ~100 VGPRs)
      constexpr int KW = D >= 256 ? 4 : (KSTEPS < 8 ? KSTEPS : 8);
      pu32x4_t kf[KW];
#pragma unroll
      for (int m = 0; m < KW; ++m)
        kf[m] = AOWN ? *(const pu32x4_t*)(k_rd + (m & 1) * 32 * RSK + (m >> 1) * 32)          // (the hand-owned form alternates the two blocks)
                     : *(const pu32x4_t*)(k_rd + (m / KSTEPS) * 32 * RSK + (m % KSTEPS) * 32);
      __builtin_amdgcn_sched_group_barrier(0x100, KW, 0);       // DS reads
      if constexpr (AOWN) {
        // (asm statements stay in source order: fragment m + KW is requested right behind the instruction that retires
        // fragment m's registers, KW instructions before it is needed)
        // the two 32-key blocks' chains ALTERNATE: an accumulate chain runs at the matrix pipe's full rate only with
        // another instruction between two of its steps (tools/probes/issue_model.hip: distance 2)
        pm_for<2 * KSTEPS>([&](auto M) __attribute__((always_inline)) {
          constexpr int m = decltype(M)::value, kb = m & 1, ks = m >> 1;
          pmma_own<T>::template qk<kOwnQ + 4 * ks>(s_acc[kb], kf[m % KW]);
          if constexpr (m + KW < 2 * KSTEPS)
            kf[m % KW] = *(const pu32x4_t*)(k_rd + ((m + KW) & 1) * 32 * RSK + ((m + KW) >> 1) * 32);
        });
        // the matrix pipe's results are readable (the compiler does not know these statements are matrix instructions)
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" : "+v"(s_acc[0]), "+v"(s_acc[1]));
      } else
#pragma unroll
      for (int m = 0; m < 2 * KSTEPS; ++m) {
        const int kb = m / KSTEPS, ks = m % KSTEPS;
        s_acc[kb] = pmma<T>::run(__builtin_bit_cast(ps16x8_t, kf[m % KW]), qf[ks], s_acc[kb]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
        if (m + KW < 2 * KSTEPS) {
          kf[m % KW] = *(const pu32x4_t*)(k_rd + ((m + KW) / KSTEPS) * 32 * RSK + ((m + KW) % KSTEPS) * 32);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);    // 1 DS read
        }
      }
      // ---- softmax (log2 domain) -------------------------------------------------------------------
      // register r of block kb <-> key key_base + 32kb + (r&3) + 8(r>>2) + 4half
      bool need_mask = (key_base + kTileN - 1 > ctx_len + w_tok_lo) || (key_base + kTileN > seq_len);
      if (FEAT) need_mask = need_mask || (p.sliding_window > 0 && key_base < ctx_len + w_tok_hi - p.sliding_window + 1);
      float mx = -INFINITY;
      float sc = 1.0f;   // factor still to be applied to s_acc inside the exp2 argument
      if (plain && !need_mask) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
        mx *= scale2;      // scale2 > 0: max commutes with the scaling
        sc = scale2;
      } else if (!FEAT || plain && !(p.sliding_window > 0)) {
        // causal / end-of-sequence mask only
        const int lim = row_ok ? min(q_abs, seq_len - 1) : -1;   // last visible key of this row
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key_base + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float x = key <= lim ? s_acc[kb][r] * scale2 : -INFINITY;
            s_acc[kb][r] = x;
            mx = fmaxf(mx, x);
          }
      } else {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key_base + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
            float x = s_acc[kb][r] * scale_nat;
            if (p.softcap > 0.0f) x = softcap_fast(x, p.softcap, softcap_k);
            bool ok = row_ok && key <= q_abs && key < seq_len;
            if (p.sliding_window > 0) ok = ok && (q_abs - key) < p.sliding_window;
            x = ok ? x : -INFINITY;
            if (p.alibi_slopes) x += slope * (float)(key - ctx_len);
            x *= kLog2eP;
            s_acc[kb][r] = x;
            mx = fmaxf(mx, x);
          }
      }
      mx = fmaxf(mx, lane_xor32(mx));  // the other half-wave holds the other 32 keys of this query row
      float m_new = fmaxf(m_run, mx);
      if constexpr (AOWN) {
        // rescaling O is 128 read-multiply-write round trips through the accumulator registers: the row's reference only
        // moves when its maximum has grown by more than 2^8 since (P <= 2^8 then: exact in f32 sums, far inside bf16 / f16;
        // softmax is shift invariant) - the deferral of prefill_dma_kernel
        if (m_run > -INFINITY && mx <= m_run + 8.0f) m_new = m_run;
      }
      if (!(m_new > -INFINITY)) m_new = 0.0f;                     // row fully masked so far (:486-489)
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      float psum = 0.0f;
      ps16x8_t pf[4];                                             // B operands of the 4 k-steps of P.V
#pragma unroll
      for (int kb = 0; kb < 2; ++kb) {
        float e[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          e[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s_acc[kb][r], sc, -m_new));
          psum += e[r];
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const pu32x4_t w = {pmma<T>::pack2(e[8 * s + 0], e[8 * s + 1]), pmma<T>::pack2(e[8 * s + 2], e[8 * s + 3]),
                              pmma<T>::pack2(e[8 * s + 4], e[8 * s + 5]), pmma<T>::pack2(e[8 * s + 6], e[8 * s + 7])};
          pf[2 * kb + s] = __builtin_bit_cast(ps16x8_t, w);
        }
      }
      l_run = l_run * alpha + psum;
      m_run = m_new;
      if (EARLY_K && has_next) {               // the other stage is idle during this whole tile
        write_lds_k(smem + (((tile - tile_lo) & 1) ^ 1) * BUF);
        k_written = true;
      }
      // ---- O^T = alpha * O^T + V^T . P^T -----------------------------------------------------------
      const bool rescale = !__all(alpha == 1.0f);                 // exact: skipped only when no row's max moved
      if (rescale) {
        if constexpr (AOWN) {
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");     // the previous tile's last P.V results are in their registers
          pm_for<16 * DBLK>([&](auto I) __attribute__((always_inline)) {
            constexpr int i = kOwnO + decltype(I)::value;
            pm_acc_write<i>(__builtin_bit_cast(uint32_t, pm_acc_read<i>() * alpha));
          });
        } else {
#pragma unroll
        for (int b = 0; b < DBLK; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[b][r] *= alpha;
        }
      }
      // transposed V reads run VW k-steps (a 32-wide output block; half of one at D = 256, for registers) ahead of
      // the MFMAs that consume them
      constexpr int VW = D >= 256 ? 4 : 8;
      ps16x4_t vt[VW][2];
      auto read_v_step = [&](int m, ps16x4_t (&dst)[2]) {       // m = 4*b + sk
        const char* va = v_rd + (m & 3) * 16 * RSV + (m >> 2) * 64;
        dst[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4_t*)(va));
        dst[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4_t*)(va + 8 * RSV));
      };
#pragma unroll
      for (int m = 0; m < VW; ++m) read_v_step(AOWN ? (4 * (m & 1) + (m >> 1)) : m, vt[m]);     // (hand-owned form: blocks 0, 1 alternate)
      __builtin_amdgcn_sched_group_barrier(0x100, 2 * VW, 0);
      if constexpr (AOWN) {
        asm volatile("s_nop 1" ::: "memory");                   // VALU-written P -> matrix instruction
        // step i of the stream: output blocks 2p, 2p + 1 alternate (accumulate chains at distance 2, see above)
        auto pv_step = [](int i) { const int pr = i >> 3, j = i & 7; return 4 * (2 * pr + (j & 1)) + (j >> 1); };   // -> m = 4 b + sk
        pm_for<4 * DBLK>([&](auto I) __attribute__((always_inline)) {
          constexpr int i = decltype(I)::value, pr = i >> 3, j = i & 7, b = 2 * pr + (j & 1), sk = j >> 1;
          const ps16x4_t v0 = vt[i % VW][0], v1 = vt[i % VW][1];
          const ps16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          pmma_own<T>::template pv<kOwnO + 16 * b>(vf, pf[sk]);
          if constexpr (i + VW < 4 * DBLK) read_v_step(pv_step(i + VW), vt[i % VW]);
        });
      } else
#pragma unroll
      for (int m = 0; m < 4 * DBLK; ++m) {
        const ps16x4_t v0 = vt[m % VW][0], v1 = vt[m % VW][1];
        const ps16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        o_acc[m >> 2] = pmma<T>::run(vf, pf[m & 3], o_acc[m >> 2]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (m + VW < 4 * DBLK) {
          read_v_step(m + VW, vt[m % VW]);
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
    }
    // the other stage was last read one iteration ago and every wave has passed a barrier since
    if (has_next) {
      char* other = smem + (((tile - tile_lo) & 1) ^ 1) * BUF;
      if (!k_written) write_lds_k(other);
      write_lds_v(tile + 1, other);
    }
    __syncthreads();
  }

  // ---- epilogue: O = O^T / l, lane (qr, half) register r of block b <-> d = 32b + (r&3) + 8(r>>2) + 4half
  l_run += lane_xor32(l_run);
  if (lse_base && row_ok && half == 0)    // m_run: row max of the scaled scores in the log2 domain
    lse_base[(int64_t)(q_start + tok_local) * p.lse_stride_token + hq] = l_run > 0.0f ? (m_run + __builtin_amdgcn_logf(l_run)) * 0.6931471805599453f : -INFINITY;
  const float inv = (row_ok && l_run > 0.0f) ? v_scale / l_run : 0.0f;
  // O leaves through LDS as whole rows, 16 bytes per lane, nontemporal (see prefill_dma_kernel's epilogue); the
  // 8-byte pieces of the accumulator layout go out directly only when the output rows are not 16-byte aligned.
  const bool wide_store = (((uintptr_t)out_base & 15) == 0) && (p.out_stride_token % 8 == 0) && (p.out_stride_head % 8 == 0);
  // register 4 c + i of output block b, wherever O lives
  auto o_quad = [&](auto B, auto C, float (&o)[4]) __attribute__((always_inline)) {
    constexpr int b = decltype(B)::value, c = decltype(C)::value;
    if constexpr (AOWN) {
      o[0] = pm_acc_read<kOwnO + 16 * b + 4 * c>(); o[1] = pm_acc_read<kOwnO + 16 * b + 4 * c + 1>();
      o[2] = pm_acc_read<kOwnO + 16 * b + 4 * c + 2>(); o[3] = pm_acc_read<kOwnO + 16 * b + 4 * c + 3>();
    } else {
      o[0] = o_acc[b][4 * c]; o[1] = o_acc[b][4 * c + 1]; o[2] = o_acc[b][4 * c + 2]; o[3] = o_acc[b][4 * c + 3];
    }
  };
  if constexpr (AOWN) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 3" ::: "memory");   // the last P.V results are in their registers
  if (wide_store) {
    constexpr int ORS = D * 2 + 16;                      // padded row stride of the parked rows
    constexpr int CPR = D / 8, RPI = 64 / CPR;           // 16-byte chunks per row, rows per store instruction
    char* ost = smem + wave * (32 * ORS);                // every wave is past the last tile's barrier: the stages are idle
    pm_for<DBLK>([&](auto B) __attribute__((always_inline)) {
      pm_for<4>([&](auto C) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, c = decltype(C)::value;
        float o[4];
        o_quad(B, C, o);
        const pu32x2_t w = {pmma<T>::pack2(o[0] * inv, o[1] * inv), pmma<T>::pack2(o[2] * inv, o[3] * inv)};
        *(pu32x2_t*)(ost + qr * ORS + (32 * b + 8 * c + 4 * half) * 2) = w;
      });
    });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    const int orow = lane / CPR, och = lane % CPR;
#pragma unroll
    for (int j = 0; j < 32 / RPI; ++j) {
      const int r = RPI * j + orow;
      const int m = wave * 32 + r;
      const int tok = tok0 + m / G;
      const pu32x4_t v = *(const pu32x4_t*)(ost + r * ORS + och * 16);
      if (m < BQ * G && tok < q_len && och * 8 < a.d_valid)
        __builtin_nontemporal_store(v, (pu32x4_t*)(out_base + (int64_t)(q_start + tok) * p.out_stride_token +
                                                   (int64_t)(head * G + m % G) * p.out_stride_head + och * 8));
    }
  } else if (row_ok) {
    uint16_t* op = out_base + (int64_t)(q_start + tok_local) * p.out_stride_token + (int64_t)hq * p.out_stride_head + 4 * half;
    pm_for<DBLK>([&](auto B) __attribute__((always_inline)) {
      pm_for<4>([&](auto C) __attribute__((always_inline)) {
        constexpr int b = decltype(B)::value, c = decltype(C)::value;
        float o[4];
        o_quad(B, C, o);
        const pu32x2_t w = {pmma<T>::pack2(o[0] * inv, o[1] * inv), pmma<T>::pack2(o[2] * inv, o[3] * inv)};
        if (32 * b + 8 * c + 4 * half < a.d_valid) *(pu32x2_t*)(op + 32 * b + 8 * c) = w;
      });
    });
  }
}

// =============================================================================================
// prefill_dma_kernel: the D = 128, no-soft-cap/ALiBi/window fast path.
//
// Same decomposition and MFMA orientation as prefill_mfma_kernel, plus three changes that cut the
// per-tile instruction count (the kernel is issue-bound well before it is MFMA-bound):
//   * K/V tiles go HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, no VGPR round trip, no
//     ds_write pass). The DMA writes 1 KiB contiguously per wave-instruction, so rows cannot be
//     padded; bank conflicts are avoided by an XOR swizzle of the 16-byte chunk index instead,
//     applied on the per-lane SOURCE address and again on the read address:
//       K (ds_read_b128 rows):        chunk ^= row & 15
//       V (ds_read_b64_tr_b16 reads): chunk ^= ((row & 3) << 2) | ((row >> 2) & 3)
//     Rows past the sequence are redirected to the sequence's last row (finite data, masked P).
//   * the query is pre-scaled by scale*log2(e) once, and the running max enters through the MFMA
//     accumulator: S^T starts from C = -m_ref, so P = exp2(acc) is ONE instruction per score.
//     m_ref is only moved when some row's tile max exceeds it by more than kDeferThr (log2 units);
//     the softmax is shift-invariant, so results are unchanged up to rounding (P <= 2^kDeferThr).
//   * read addresses are per-lane constants (the swizzle is folded in once), the tile loop is
//     unrolled by two so the LDS stage is an immediate offset.
// =============================================================================================
constexpr float kDeferThr = 8.0f;

// Diagnostic build only (-DMI355_PROFILE_PHASES, tools/phase_profile.py): per-wave cycle sums of the
// tile loop's phases, added into a caller-provided buffer whose address travels in reserved0/1.
// No stamp executes in the product build.
#ifdef MI355_PROFILE_PHASES
#define MI355_STAMP(idx)                                                                   \
  do {                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    unsigned long long t_now;                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_now)::"memory");          \
    __builtin_amdgcn_sched_barrier(0);                                                     \
    prof_sum[idx] += t_now - prof_last;                                                    \
    prof_last = t_now;                                                                     \
  } while (0)
#else
#define MI355_STAMP(idx) do { } while (0)
#endif

// Diagnostic build only (-DMI355_PROFILE_WG, tools/wg_profile.py): life of a workgroup in s_memtime ticks.
#ifdef MI355_PROFILE_WG
#define MI355_WG_STAMP(var)                                                                \
  unsigned long long var;                                                                  \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define MI355_WG_REALTIME(var)                                                             \
  unsigned long long var;                                                                  \
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define MI355_WG_STAMP(var) do { } while (0)
#define MI355_WG_REALTIME(var) do { } while (0)
#endif

// NW waves of 32 rows share the staged K/V tiles (Q block = 32*NW rows), NST stages of 32 KiB.
//   NW 4, NST 2: two workgroups per CU, each staging its own tiles (64 KiB of LDS-DMA per CU per tile round)
//   NW 8, NST 3: one workgroup per CU, HALF the LDS-DMA bytes per MFMA and a two-tile-deep prefetch.
// The kernel's floor is the per-CU global->LDS fill rate (~25 GB/s per CU, MI355X_MICROARCH.md
// ldsdma-fill), not the matrix pipes: tools/phase_profile.py shows the waves waiting on their DMA.
// (A ping-pong variant - waves NW/2.. lagging half a tile so that one group's softmax sits beside the
// other's MFMAs, 4 stages - was built and measured 4 % SLOWER than the plain 8-wave kernel: the chip is
// at its 1.4 kW package power limit under this kernel (tools/clock_watch.py), so re-ordering the same
// work buys nothing; only doing less work per FLOP does. It is not kept.)
// (The row geometry is written for D = 128 and D = 256 - 512-byte rows, a wave's LDS-DMA instruction covers two key rows
// of 32 chunks - but only D = 128 is instantiated: see launch_prefill for what the D = 256 form measured.)
// WR: the instantiation that carries the fused cache write (write_new_kv) - apart, so that the plain one keeps its stream.
template <typename T, int NW, int NST, int D = 128, bool WR = false>
__global__ __launch_bounds__(NW * 64, (NW == 4 && D == 128) ? 2 : 1) void prefill_dma_kernel(const PrefillArgs a) {
  static_assert(D == 128 || D == 256, "rows of 16 or 32 chunks");
  constexpr int ROWB = D * 2;                 // 256-byte rows of 16 chunks of 16 B (D = 256: 512 bytes, 32 chunks)
  constexpr int CPRW = ROWB / 16;             // chunks per row
  constexpr int RPW = 64 / CPRW;              // key rows one wave's LDS-DMA instruction covers (4, or 2)
  constexpr int KBUF = kTileN * ROWB, STAGE = 2 * KBUF;   // K tile then V tile
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int RP = RPW * NW;                // key rows one LDS-DMA piece covers (all waves, RPW rows each)
  constexpr int NP = kTileN / RP;             // pieces per tile: each is one K and one V instruction per wave
  constexpr int PD = NST - 1;                 // prefetch distance in tiles
  constexpr int IPT = 2 * NP;                 // LDS-DMA instructions per tile per wave
  static_assert(NP >= 1 && KSTEPS % NP == 0, "piece interleave");

  extern __shared__ __attribute__((aligned(16))) char smem[];   // two stages
  const mi355_attn_params& p = a.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.group, BQ = a.block_q;
  MI355_WG_STAMP(wg_t0);
  MI355_WG_REALTIME(wg_r0);

  const int head = (int)(blockIdx.x % p.num_kv_heads);
  const int qblock = (int)(gridDim.x / p.num_kv_heads - 1 - blockIdx.x / p.num_kv_heads);
  int q_start, q_len, seq_len;
  const int seq = find_seq_and_lengths(p.cu_seqlens_q, p.seqused_k, p.num_seqs, qblock, BQ, lane, q_start, q_len, seq_len);
  if (seq < 0) return;
  // Prologue order (every step is a memory round trip of ~1-2 us that nothing else on this CU hides when
  // the workgroup owns it): {sequence index, its lengths} in one trip (round 4: a ballot over the first 63 sequences' words
  // instead of a binary search's dependent loads) -> block table -> {first K/V tiles, Q rows} together. The block table is
  // bounded by max_seqlen_k's page count instead of this Q block's own last page.
  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;
  const int bt_last_any = ((max(p.max_seqlen_k, 1) + p.page_size - 1) >> a.page_shift) - 1;
  constexpr bool BT_IN_LDS = NST >= 3;
  const int* bt_lds = (const int*)(smem + NST * STAGE);
  int bt_chunk = 0, bt_cur = 0, bt_nxt = 0;
  if constexpr (BT_IN_LDS) {
    for (int c = wave; c * 64 <= bt_last_any; c += NW) glds4(bt + min(c * 64 + lane, bt_last_any), lds_addr(bt_lds) + c * 256);
  } else {
    bt_cur = bt[min(lane, bt_last_any)];
    bt_nxt = bt[min(64 + lane, bt_last_any)];
  }
  const int qb_local = qblock - (q_start / BQ + seq);
  if (qb_local * BQ >= q_len || q_len <= p.skip_decodes || (p.only_decodes && q_len > p.only_decodes)) {
    if constexpr (BT_IN_LDS) glds_wait_all();   // never leave with a DMA into this workgroup's LDS in flight
    return;
  }
  const int ctx_len = seq_len - q_len;
  const int tok0 = qb_local * BQ;
  MI355_WG_STAMP(wg_ta);   // metadata known


  const int qr = lane & 31, half = lane >> 5;
  const int m_row = wave * 32 + qr;
  const int tok_local = tok0 + m_row / G;
  const int hq = head * G + m_row % G;
  const bool row_ok = (m_row < BQ * G) && (tok_local < q_len);
  const int q_abs = ctx_len + tok_local;
  const int lim = row_ok ? min(q_abs, seq_len - 1) : -1;   // last visible key of this row

  const int w_tok_lo = tok0 + (wave * 32) / G;
  const int w_tok_hi = min(min(tok0 + (wave * 32 + 31) / G, tok0 + BQ - 1), q_len - 1);
  const int wg_tok_hi = min(tok0 + BQ - 1, q_len - 1);
  const int n_keys_wg = max(0, min(ctx_len + wg_tok_hi + 1, seq_len));
  const int wave_keys = min(ctx_len + w_tok_hi + 1, seq_len);
  const bool wave_has_rows = w_tok_lo <= w_tok_hi;
  int tile_lo = 0;
  int tile_hi = (n_keys_wg + kTileN - 1) / kTileN;
  // key-split launch (see prefill_mfma_kernel): an even share of this Q block's tiles, partial output and lse
  const int ksplit = a.key_splits > 1 ? (int)blockIdx.y : 0;
  if (a.key_splits > 1) {
    const int tps = (tile_hi + a.key_splits - 1) / a.key_splits;
    tile_lo = min(ksplit * tps, tile_hi);
    tile_hi = min(tile_hi, tile_lo + tps);
  }
  uint16_t* const out_base = (uint16_t*)p.out + (int64_t)ksplit * a.out_split_stride;
  float* const lse_base = p.lse ? p.lse + (int64_t)ksplit * a.lse_split_stride : nullptr;

  const float scale2 = p.scale * kLog2eP;
  // ---- Q rows: the longest round trip of the prologue (cold HBM), issued as soon as the row is known ----
  pu32x4_t qraw[KSTEPS];
  {
    // padding rows read the sequence's last query row (always a valid address) and are zeroed at the conversion:
    // unconditional loads, no branch per load
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)(q_start + min(tok_local, q_len - 1)) * p.q_stride_token + (int64_t)hq * p.q_stride_head + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) qraw[ks] = *(const pu32x4_t*)(qp + 16 * ks);   // (nontemporal here: -1.5 %)
  }

  // ---- DMA staging constants ----------------------------------------------------------------------
  // lane handles LDS chunk (row = (tid>>4) + 16 i, c = tid & 15) of both tiles; 16-key group i <-> load i
  const char* kbase = (const char*)p.k_cache + (int64_t)head * p.k_stride_head * 2;
  const char* vbase = (const char*)p.v_cache + (int64_t)head * p.v_stride_head * 2;
  const int last_group = (max(n_keys_wg, 1) - 1) >> 4;
  const int page_mask = p.page_size - 1;
  const int rowin = tid / CPRW, ch = tid % CPRW;  // row within a piece (0 .. RP-1), 16-byte chunk
  // piece i covers rows RP i + rowin of the tile: with RP >= 16 a lane's row within its 16-key group is the same for
  // every piece; with RP = 8 (D = 256) odd pieces are the group's second half
  auto rig_of = [&](int i) { return (RP * i + rowin) & 15; };              // row within its 16-key group
  auto grp_of = [&](int i) { return __builtin_amdgcn_readfirstlane((RP * i + rowin) >> 4); };   // wave-uniform: a wave's rows share a group
  auto fk_of = [&](int r) { return r; };
  auto fv_of = [&](int r) { return ((r & 3) << 2) | ((r >> 2) & 3); };
  // (the swizzles flip the low four bits of the chunk index: inside each 256-byte half of a 512-byte row)
  uint32_t k_voff[RP >= 16 ? 1 : 2], v_voff[RP >= 16 ? 1 : 2];
#pragma unroll
  for (int j = 0; j < (RP >= 16 ? 1 : 2); ++j) {
    const int r = rig_of(j);
    k_voff[j] = (uint32_t)(r * (int)a.k_slot_stride * 2 + ((ch ^ fk_of(r)) << 4));
    v_voff[j] = (uint32_t)(r * (int)a.v_slot_stride * 2 + ((ch ^ fv_of(r)) << 4));
  }
  const uint32_t lds_wave = (uint32_t)(wave * 64 * 16);        // + i*RP*ROWB (+KBUF for V) + stage

  const int last_entry = (last_group << 4) >> a.page_shift;
  const uint32_t k_page_bytes = a.k_page_stride * 2, v_page_bytes = a.v_page_stride * 2;
  // Block-table entries of this workgroup's keys (fetch issued at the top of the kernel).
  //   NST == 2: 64 at a time in a VGPR (lane l = entry chunk*64 + l), one chunk ahead, picked with
  //             v_readlane: no scalar-cache round trip inside the loop.
  //   NST >= 3: the whole prefix is staged once in LDS behind the tile stages by LDS-DMA. A VGPR chunk
  //             is a compiler-visible load carried round the loop: hipcc waits for it with vmcnt(0) at
  //             every copy, which would also drain the tile that is meant to stay in flight.
  if constexpr (!BT_IN_LDS) {
    if (tile_lo > 0) {      // key-split launch: this share starts deep in the table, not in the chunks fetched at the top
      bt_chunk = ((min(tile_lo * 4, last_group) << 4) >> a.page_shift) >> 6;
      bt_cur = bt[min(bt_chunk * 64 + lane, bt_last_any)];
      bt_nxt = bt[min((bt_chunk + 1) * 64 + lane, bt_last_any)];
    }
  }
  if constexpr (BT_IN_LDS) {
    // the first tiles' addresses come out of this table; the KSTEPS Q loads issued after it stay in flight
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(KSTEPS) : "memory");
    __syncthreads();
  }
  // NW == 4 (round 4): wave w stages 16-key group w of every tile, both matrices - ONE block-table entry and one scalar
  // base per tile and wave, four K and four V instructions of 1 KiB (rows 4j .. 4j+3 of the group) with constant per-lane
  // offsets and the base as an SGPR pair. The former deal - every wave a quarter of every 16-row piece - cost each wave
  // four page look-ups and 64-bit address sums per tile, ~250 instructions where this costs ~50, and one wave per SIMD
  // pays ~4 cycles for each (prefill_lat.hip: the same change took that kernel from 15.9 to 13.1 us at 1 x 512).
  constexpr bool WAVE_GROUP = D == 128;          // (NW == 8: two waves share a 16-key group, eight rows each, two instructions per matrix)
  constexpr int RPWV = kTileN / NW;              // key rows of a tile one wave stages (16 or 8)
  uint64_t wg_kb = 0, wg_vb = 0;              // this wave's group of the tile being staged: scalar bases
  bool wg_tail = false;
  int wg_key0 = 0, wg_src = 0;                // wg_src: 0 the cache, 1 the linear new-token tensors, 2 both (the group straddles ctx_len)
  constexpr bool fused = WAVE_GROUP && WR;
  const int new_st = (int)p.new_stride_token;
  const char* const knew = (const char*)p.k_new + (int64_t)head * p.new_stride_head * 2;
  const char* const vnew = (const char*)p.v_new + (int64_t)head * p.new_stride_head * 2;
  if constexpr (fused) {
    // this Q block's own tokens into their pages (by slot_mapping when the caller hands one in - negative: not stored - else
    // by position through the block table); nobody in this launch reads them from the cache
    const int tok_end = min(tok0 + BQ, q_len);
    for (int tok = tok0 + (tid >> 4); tok < tok_end; tok += NW * 4) {
      int64_t slot;
      if (p.slot_mapping) slot = p.slot_mapping[q_start + tok];
      else if (p.slot_mapping_i32) slot = p.slot_mapping_i32[q_start + tok];
      else { const int pos = ctx_len + tok; slot = (int64_t)bt[pos >> a.page_shift] * p.page_size + (pos & (p.page_size - 1)); }
      if (slot < 0) continue;
      const int64_t pg = slot >> a.page_shift, sl = slot & (p.page_size - 1);
      const int64_t src = (int64_t)(q_start + tok) * (new_st * 2) + (tid & 15) * 16;
      const pu32x4_t kk = *(const pu32x4_t*)(knew + src), vv = *(const pu32x4_t*)(vnew + src);
      *(pu32x4_t*)((char*)kbase + pg * ((int64_t)a.k_page_stride * 2) + sl * ((int64_t)a.k_slot_stride * 2) + (tid & 15) * 16) = kk;
      *(pu32x4_t*)((char*)vbase + pg * ((int64_t)a.v_page_stride * 2) + sl * ((int64_t)a.v_slot_stride * 2) + (tid & 15) * 16) = vv;
    }
  }
  uint32_t wgk_voff[4], wgv_voff[4];
  if constexpr (WAVE_GROUP) {
#pragma unroll
    for (int j = 0; j < RPWV / 4; ++j) {
      const int r = ((wave * RPWV) & 15) + 4 * j + (lane >> 4), c = lane & 15;       // row within the 16-key group
      wgk_voff[j] = (uint32_t)(r * (int)a.k_slot_stride * 2 + ((c ^ fk_of(r)) << 4));
      wgv_voff[j] = (uint32_t)(r * (int)a.v_slot_stride * 2 + ((c ^ fv_of(r)) << 4));
    }
  }
  auto dma_begin = [&](int tile) __attribute__((always_inline)) {          // call once per tile before its pieces
    if constexpr (!BT_IN_LDS) {
      const int e0 = (min(tile * 4, last_group) << 4) >> a.page_shift;
      if ((e0 >> 6) != bt_chunk) {            // wave-uniform; entries only ever move forward
        bt_chunk = e0 >> 6;
        bt_cur = bt_nxt;
        bt_nxt = bt[min((bt_chunk + 1) * 64 + lane, bt_last_any)];
      }
    }
    if constexpr (WAVE_GROUP) {
      const int gi = min(tile * 4 + ((wave * RPWV) >> 4), last_group);
      wg_key0 = gi << 4;
      const int slot0 = wg_key0 & page_mask;
      uint32_t page;
      if constexpr (BT_IN_LDS) page = (uint32_t)__builtin_amdgcn_readfirstlane(bt_lds[wg_key0 >> a.page_shift]);
      else page = (uint32_t)__builtin_amdgcn_readlane(bt_cur, (wg_key0 >> a.page_shift) & 63);
      const uint64_t k_off = (uint64_t)page * (a.k_page_stride * 2) + (uint64_t)((uint32_t)slot0 * a.k_slot_stride) * 2;
      const uint64_t v_off = a.kv_same_strides ? k_off : (uint64_t)page * (a.v_page_stride * 2) + (uint64_t)((uint32_t)slot0 * a.v_slot_stride) * 2;
      wg_kb = (uint64_t)kbase + k_off;
      wg_vb = (uint64_t)vbase + v_off;
      wg_tail = wg_key0 + 16 > seq_len;       // the sequence ends inside this group -> rows past it fetch its last row
      // fused cache write (write_new_kv, as in prefill_lat_kernel): a group of this call's own tokens comes from the linear
      // key / value tensors - sixteen consecutive rows of [T, Hk, D], another scalar base and row stride; the one group of a
      // sequence that straddles ctx_len takes per-lane addresses
      wg_src = 0;
      if constexpr (fused) if (wg_key0 + 16 > ctx_len) {
        if (wg_key0 >= ctx_len) {
          const uint64_t nb = (uint64_t)(uint32_t)(q_start + wg_key0 - ctx_len) * (uint64_t)(new_st * 2);
          wg_kb = (uint64_t)knew + nb;
          wg_vb = (uint64_t)vnew + nb;
          wg_src = 1;
        } else {
          wg_src = 2;
        }
      }
    }
  };
  // piece i of a tile = key rows RP*i .. RP*i+RP-1: one K and one V LDS-DMA per lane (1 KiB each per wave)
  auto dma_piece = [&](int tile, char* stage, int i) __attribute__((always_inline)) {
    if constexpr (WAVE_GROUP) {             // "piece" i = instruction i of this wave's rows: RPWV wave + 4 i .. + 3 of the tile
      if constexpr (fused) if (wg_src != 0) {   // (fused cache write: rows of the linear tensors, see dma_begin)
        const int rig = ((wave * RPWV) & 15) + 4 * i + (lane >> 4), c = lane & 15;
        const uint32_t dst = lds_addr(stage) + (uint32_t)(wave * (RPWV * ROWB) + i * 1024);
        if (wg_src == 1) {
          const int r = min(rig, max(seq_len - 1 - wg_key0, 0));
          glds16_s((uint32_t)(r * new_st * 2 + ((c ^ fk_of(rig)) << 4)), wg_kb, dst);
          glds16_s((uint32_t)(r * new_st * 2 + ((c ^ fv_of(rig)) << 4)), wg_vb, dst + KBUF);
        } else {
          const int pos = min(wg_key0 + rig, seq_len - 1);
          const bool is_new = pos >= ctx_len;
          const char* ks = is_new ? knew + (int64_t)(q_start + pos - ctx_len) * (new_st * 2) : (const char*)wg_kb + (int64_t)(pos - wg_key0) * ((int)a.k_slot_stride * 2);
          const char* vs = is_new ? vnew + (int64_t)(q_start + pos - ctx_len) * (new_st * 2) : (const char*)wg_vb + (int64_t)(pos - wg_key0) * ((int)a.v_slot_stride * 2);
          glds16(ks + ((c ^ fk_of(rig)) << 4), dst);
          glds16(vs + ((c ^ fv_of(rig)) << 4), dst + KBUF);
        }
        return;
      }
      uint32_t kvo = wgk_voff[i], vvo = wgv_voff[i];
      if (wg_tail) {
        const int rig = ((wave * RPWV) & 15) + 4 * i + (lane >> 4), c = lane & 15, r = min(rig, max(seq_len - 1 - wg_key0, 0));
        kvo = (uint32_t)(r * (int)a.k_slot_stride * 2 + ((c ^ fk_of(rig)) << 4));
        vvo = (uint32_t)(r * (int)a.v_slot_stride * 2 + ((c ^ fv_of(rig)) << 4));
      }
      const uint32_t dst = lds_addr(stage) + (uint32_t)(wave * (RPWV * ROWB) + i * 1024);
      glds16_s(kvo, wg_kb, dst);
      glds16_s(vvo, wg_vb, dst + KBUF);
      return;
    }
    const int gi = min(tile * 4 + grp_of(i), last_group);
    const int key0 = gi << 4;
    const int slot0 = key0 & page_mask;
    int page;
    if constexpr (BT_IN_LDS) page = __builtin_amdgcn_readfirstlane(bt_lds[key0 >> a.page_shift]);
    else page = __builtin_amdgcn_readlane(bt_cur, (key0 >> a.page_shift) & 63);
    uint32_t kvo = k_voff[RP >= 16 ? 0 : (i & 1)], vvo = v_voff[RP >= 16 ? 0 : (i & 1)];
    if (key0 + 16 > seq_len) {            // wave-uniform: the sequence ends inside this group -> rows past it
                                          // fetch its last row instead (never stale cache contents)
      const int rig = rig_of(i);
      const int r = min(rig, max(seq_len - 1 - key0, 0));
      kvo = (uint32_t)(r * (int)a.k_slot_stride * 2 + ((ch ^ fk_of(rig)) << 4));
      vvo = (uint32_t)(r * (int)a.v_slot_stride * 2 + ((ch ^ fv_of(rig)) << 4));
    }
    // K and V caches are two views of one tensor in vLLM: with equal strides (wave-uniform test, one scalar offset
    // serves both) the 64-bit page arithmetic is done once
    const uint64_t k_off = (uint64_t)(uint32_t)page * k_page_bytes + (uint64_t)((uint32_t)slot0 * a.k_slot_stride) * 2;
    const uint64_t v_off = a.kv_same_strides ? k_off : (uint64_t)(uint32_t)page * v_page_bytes + (uint64_t)((uint32_t)slot0 * a.v_slot_stride) * 2;
    glds16(kbase + k_off + kvo, lds_addr(stage) + lds_wave + i * (RP * ROWB));
    glds16(vbase + v_off + vvo, lds_addr(stage) + KBUF + lds_wave + i * (RP * ROWB));   // (vvo carries V's own swizzle)
  };
  auto issue_dma = [&](int tile, char* stage) __attribute__((always_inline)) {
    dma_begin(tile);
#pragma unroll
    for (int i = 0; i < NP; ++i) dma_piece(tile, stage, i);
  };
  // wait until this wave's pieces of tile t+1 have landed: `newer` tiles (t+2 ..) may stay in flight
  auto wait_next_tile = [&](int newer) {
    if (PD >= 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPT) : "memory");
    else if (PD >= 2 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IPT) : "memory");
    else glds_wait_all();
  };

  // ---- per-lane LDS read addresses (swizzle folded in) ---------------------------------------------
  // K fragment ks of 32-key block kb: row 32kb + qr, logical chunk 2ks + half
  // (D = 256: chunks 16 .. 31 are the row's second 256-byte half, same swizzle: + 256 bytes as an immediate, not a register)
  constexpr int KRD = KSTEPS < 8 ? KSTEPS : 8, VRD = DBLK < 4 ? DBLK : 4;
  uint32_t k_rd[KRD];
#pragma unroll
  for (int ks = 0; ks < KRD; ++ks) k_rd[ks] = (uint32_t)(qr * ROWB + (((2 * ks + half) ^ (qr & 15)) << 4));
  // V transposed read of k-step sk (16 keys), output block b: row 16sk + 4half + q4 (+8), logical
  // byte column 64b + 32g1 + 8pp  ->  chunk 4b + 2g1 + (pp>>1), sub-offset 8(pp&1)
  const int gq1 = (lane >> 4) & 1, li = lane & 15, q4 = li >> 2, pp = li & 3;
  uint32_t v_rd0[VRD], v_rd1[VRD];
#pragma unroll
  for (int b = 0; b < VRD; ++b) {
    const int lc = 4 * b + 2 * gq1 + (pp >> 1);
    const int r0 = 4 * half + q4, r1 = r0 + 8;
    const int f0 = ((r0 & 3) << 2) | ((r0 >> 2) & 3), f1 = ((r1 & 3) << 2) | ((r1 >> 2) & 3);
    v_rd0[b] = (uint32_t)(KBUF + r0 * ROWB + ((lc ^ f0) << 4) + 8 * (pp & 1));
    v_rd1[b] = (uint32_t)(KBUF + r1 * ROWB + ((lc ^ f1) << 4) + 8 * (pp & 1));
  }
  // ds_read's immediate offset reaches 64 KiB: stages 0 and 1 are immediates on the addresses above, a stage beyond
  // that costs one v_add per read. (Per-lane addresses of its own for that stage, 16 more VGPRs, measured 0.7 %
  // SLOWER sustained than the adds.)
  auto k_addr = [&](const char* stage, int ks) -> const char* { return stage + k_rd[ks % KRD] + (ks / KRD) * 256; };
  auto v_addr0 = [&](const char* stage, int b) -> const char* { return stage + v_rd0[b % VRD] + (b / VRD) * 256; };
  auto v_addr1 = [&](const char* stage, int b) -> const char* { return stage + v_rd1[b % VRD] + (b / VRD) * 256; };

  float m_ref = 0.0f, l_run = 0.0f;
  bool started = !row_ok;               // padding rows never see a key: do not let them force the slow path
  pf32x16_t cinit;                      // -m_ref in every register: the C operand that starts S^T
#pragma unroll
  for (int r = 0; r < 16; ++r) cinit[r] = 0.0f;
  pf32x16_t o_acc[DBLK];
#pragma unroll
  for (int b = 0; b < DBLK; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) o_acc[b][r] = 0.0f;

  MI355_WG_STAMP(wg_tb);   // block table staged
#pragma unroll
  for (int t = 0; t < PD; ++t)
    if (tile_lo + t < tile_hi) issue_dma(tile_lo + t, smem + t * STAGE);
  // Q fragments, pre-scaled into the log2 domain (the compiler's wait for the Q loads lands here: they
  // are older than the DMA, so it does not wait for the tiles)
  ps16x8_t qf[KSTEPS];
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) {
    const pu32x4_t v = row_ok ? qraw[ks] : pu32x4_t{0, 0, 0, 0};
    pu32x4_t w;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float lo, hi;
      if constexpr (__is_same(T, bf16_t)) {
        lo = bf16_to_f32((uint16_t)(v[e] & 0xffff)); hi = bf16_to_f32((uint16_t)(v[e] >> 16));
      } else {
        lo = f16_to_f32((uint16_t)(v[e] & 0xffff)); hi = f16_to_f32((uint16_t)(v[e] >> 16));
      }
      w[e] = pmma<T>::pack2(lo * scale2, hi * scale2);
    }
    qf[ks] = __builtin_bit_cast(ps16x8_t, w);
  }
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) asm volatile("" : "+v"(qf[ks]));
  wait_next_tile(max(0, min(PD - 1, tile_hi - tile_lo - 1)));              // the first tile has landed
  __syncthreads();
  MI355_WG_STAMP(wg_t1);
#ifdef MI355_PROFILE_PHASES
  unsigned long long prof_sum[6] = {0, 0, 0, 0, 0, 0}, prof_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_last)::"memory");
#endif

  // `prefetch`: issue the NEXT tile's LDS-DMA pieces from inside the Q.K^T MFMA stream (an LDS-DMA
  // instruction costs the issuing wave ~100+ cycles when issued in a burst, far less between MFMAs)
  pf32x16_t s_acc[2];      // S^T of the tile in flight: lives from qk_phase to sm_phase (across a barrier for the lagging wave group)
  ps16x8_t pf[4];          // P^T of that tile as MFMA B operands: from sm_phase to pv_phase
  auto qk_phase = [&](int tile, const char* stage, char* next_stage, bool prefetch) {   // next_stage: where tile+PD goes
    if (prefetch) dma_begin(tile + PD);
    // ---- S^T - m_ref = K . Q'^T + cinit --------------------------------------------------------------
#ifdef MI355_ABLATE_QK
    s_acc[0] = cinit; s_acc[1] = cinit;
    asm volatile("" : "+v"(s_acc[0]), "+v"(s_acc[1]));
#else
    // the fragments of the two 32-key blocks as one stream of 2 KSTEPS reads through a ring of KW registers (a whole
    // block at D = 128; half of one at D = 256, where 16 fragments would be 64 registers)
    constexpr int KW = KSTEPS < 8 ? KSTEPS : 8;
    auto k_frag_addr = [&](int m) { return k_addr(stage, m % KSTEPS) + (m / KSTEPS) * 32 * ROWB; };
    pu32x4_t kf[KW];
#pragma unroll
    for (int m = 0; m < KW; ++m) kf[m] = *(const pu32x4_t*)k_frag_addr(m);
    __builtin_amdgcn_sched_group_barrier(0x100, KW, 0);
#pragma unroll
    for (int m = 0; m < 2 * KSTEPS; ++m) {
      const int kb = m / KSTEPS, ks = m % KSTEPS;
      s_acc[kb] = pmma<T>::run(__builtin_bit_cast(ps16x8_t, kf[m % KW]), qf[ks], ks == 0 ? cinit : s_acc[kb]);
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (m + KW < 2 * KSTEPS) {
        kf[m % KW] = *(const pu32x4_t*)k_frag_addr(m + KW);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
#ifndef MI355_ABLATE_DMA
      if (kb == 0 && prefetch && (ks % (KSTEPS / NP)) == 1) {
        dma_piece(tile + PD, next_stage, ks / (KSTEPS / NP));
      }
#endif
    }
#endif
#ifdef MI355_PROFILE_PHASES
    asm volatile("" :: "v"(s_acc[1][0]));   // the stamp below waits for the QK chains, not just their issue
#endif
    MI355_STAMP(1);
  };
  auto sm_phase = [&](int tile) {
    const int key_base = tile * kTileN;
    // ---- softmax -------------------------------------------------------------------------------------
#ifdef MI355_ABLATE_SOFTMAX
    asm volatile("" :: "v"(s_acc[0]), "v"(s_acc[1]));
#pragma unroll
    for (int i = 0; i < 4; ++i) { pu32x4_t w = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u}; asm volatile("" : "+v"(w)); pf[i] = __builtin_bit_cast(ps16x8_t, w); }
    l_run += 1.0f;
#else
    const bool need_mask = (key_base + kTileN - 1 > ctx_len + w_tok_lo) || (key_base + kTileN > seq_len);
    if (need_mask) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = key_base + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
          s_acc[kb][r] = key <= lim ? s_acc[kb][r] : -INFINITY;
        }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kb][r]);
    mx = fmaxf(mx, lane_xor32(mx));                       // the other half-wave holds the other 32 keys
    float alpha = 1.0f;
    const bool calm = started && mx <= kDeferThr;         // this row keeps its reference max
    if (!__all(calm)) {
      // move the reference of the rows that need it: first visible key, or max grew past the threshold
      const float upd = (!calm && mx > -INFINITY) ? mx : 0.0f;
      // a row's FIRST reference may lie far below 0 (scores under -128 in log2 units): 2^-upd would overflow and turn the
      // row's (still zero) sums into NaN. Nothing has been accumulated for such a row yet: its factor is 1.
      alpha = started ? __builtin_amdgcn_exp2f(-upd) : 1.0f;
      started = started || (mx > -INFINITY);
      m_ref += upd;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_acc[kb][r] -= upd;
#pragma unroll
      for (int r = 0; r < 16; ++r) cinit[r] = -m_ref;
      l_run *= alpha;
#pragma unroll
      for (int b = 0; b < DBLK; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[b][r] *= alpha;
    }
#ifdef MI355_PACKED_ROWSUM
    // row sum in four independent packed partial sums (v_pk_add_f32)
    pf32x2_t ps2[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
#else
    // row sum in four independent scalar chains seeded by the first four terms: packed f32 adds beside MFMAs cost
    // more than the two scalar adds they replace (MI355X_MICROARCH: "an anti-lever beside MFMAs")
    float ps1[4];
#endif
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      float e[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(s_acc[kb][r]);
#ifdef MI355_PACKED_ROWSUM
#pragma unroll
      for (int r = 0; r < 16; r += 2) ps2[(r >> 1) & 3] += pf32x2_t{e[r], e[r + 1]};
#else
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (kb == 0 && r < 4) ps1[r] = e[r];
        else ps1[r & 3] += e[r];
      }
#endif
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const pu32x4_t w = {pmma<T>::pack2(e[8 * s + 0], e[8 * s + 1]), pmma<T>::pack2(e[8 * s + 2], e[8 * s + 3]),
                            pmma<T>::pack2(e[8 * s + 4], e[8 * s + 5]), pmma<T>::pack2(e[8 * s + 6], e[8 * s + 7])};
        pf[2 * kb + s] = __builtin_bit_cast(ps16x8_t, w);
      }
    }
#ifdef MI355_PACKED_ROWSUM
    const pf32x2_t ps = (ps2[0] + ps2[1]) + (ps2[2] + ps2[3]);
    l_run += ps[0] + ps[1];
#else
    l_run += (ps1[0] + ps1[1]) + (ps1[2] + ps1[3]);
#endif
#endif
    MI355_STAMP(2);
  };
  auto pv_phase = [&](const char* stage) {
    // ---- O^T += V^T . P^T ------------------------------------------------------------------------------
#ifdef MI355_ABLATE_PV
    asm volatile("" :: "v"(pf[0]), "v"(pf[1]), "v"(pf[2]), "v"(pf[3]));
#else
    // fragments are assembled at read time (both halves written straight into one 4-VGPR operand)
    ps16x8_t vfr[2][4];
    auto read_v_block = [&](int b, ps16x8_t (&dst)[4]) {
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        const ps16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4_t*)(v_addr0(stage, b) + sk * 16 * ROWB));
        const ps16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ps16x4_t*)(v_addr1(stage, b) + sk * 16 * ROWB));
        dst[sk] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
    };
    read_v_block(0, vfr[0]);
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
    for (int b = 0; b < DBLK; ++b) {
      if (b + 1 < DBLK) read_v_block(b + 1, vfr[(b + 1) & 1]);
#pragma unroll
      for (int sk = 0; sk < 4; ++sk) {
        o_acc[b] = pmma<T>::run(vfr[b & 1][sk], pf[sk], o_acc[b]);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        if (b + 1 < DBLK) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      }
    }
#endif
#ifdef MI355_PROFILE_PHASES
    asm volatile("" :: "v"(o_acc[DBLK - 1][0]));
#endif
    MI355_STAMP(3);
  };
  auto compute_tile = [&](int tile, const char* stage, char* next_stage, bool prefetch) {
    qk_phase(tile, stage, next_stage, prefetch);
    sm_phase(tile);
    pv_phase(stage);
  };

  // tile loop, NST tiles per trip so that the LDS stage is a compile-time offset
  for (int tile = tile_lo; tile < tile_hi; tile += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int t = tile + u;
      if (t < tile_hi) {
        char* cur = smem + u * STAGE;
        char* nxt = smem + ((u + PD) % NST) * STAGE;   // stage of tile t+PD = the one tile t-1 left
        const bool more = t + PD < tile_hi;
        MI355_STAMP(0);
        if (wave_has_rows && t * kTileN < wave_keys) {
          compute_tile(t, cur, nxt, more);
        } else {
#ifndef MI355_ABLATE_DMA
          if (more) issue_dma(t + PD, nxt);   // this wave has no rows left for the tile but still stages its share
#endif
        }
#ifdef MI355_PROFILE_PHASES
        wait_next_tile(max(0, min(t + PD, tile_hi - 1) - (t + 1)));
        MI355_STAMP(4);
#endif
        wait_next_tile(max(0, min(t + PD, tile_hi - 1) - (t + 1)));     // this wave's pieces of tile t+1 have landed ...
#ifndef MI355_ABLATE_BARRIER
        __syncthreads();     // ... and so have everyone else's; stage `cur` is free
#endif
        MI355_STAMP(5);
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------
  MI355_WG_STAMP(wg_t2);
#ifdef MI355_PROFILE_PHASES
  {
    unsigned long long* dbg = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0);
    if (dbg && lane == 0) {
      for (int i = 0; i < 6; ++i) atomicAdd(dbg + i, prof_sum[i]);
      atomicAdd(dbg + 6, 1ull);
    }
  }
#endif
  l_run += lane_xor32(l_run);
  if (lse_base && row_ok && half == 0)    // P = exp2(score - m_ref): the sum is relative to the reference max
    lse_base[(int64_t)(q_start + tok_local) * p.lse_stride_token + hq] = l_run > 0.0f ? (m_ref + __builtin_amdgcn_logf(l_run)) * 0.6931471805599453f : -INFINITY;
  const float inv = (row_ok && l_run > 0.0f) ? 1.0f / l_run : 0.0f;
  // O leaves through LDS: a lane owns 8-byte pieces of one row (the accumulator layout), which as global stores
  // touch 32 rows per instruction. Each wave parks its 32 rows in its own corner of the (now idle) stages and
  // writes them out as whole 256-byte rows, 16 bytes per lane. (Nontemporal stores of the 8-byte pieces were
  // measured 16 % slower over the whole kernel: no write-combining.) Needs 16-byte aligned output rows.
  const bool wide_store = (((uintptr_t)out_base & 15) == 0) && (p.out_stride_token % 8 == 0) && (p.out_stride_head % 8 == 0);
  if (wide_store) {
    constexpr int ORS = ROWB + 16;                       // padded row stride of the parked rows
    char* ost = smem + wave * (32 * ORS);
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const pu32x2_t w = {pmma<T>::pack2(o_acc[b][4 * c + 0] * inv, o_acc[b][4 * c + 1] * inv),
                            pmma<T>::pack2(o_acc[b][4 * c + 2] * inv, o_acc[b][4 * c + 3] * inv)};
        *(pu32x2_t*)(ost + qr * ORS + (32 * b + 8 * c + 4 * half) * 2) = w;
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // same wave reads other lanes' pieces back: LDS is in order per wave
    const int orow = lane / CPRW, och = lane % CPRW;
#pragma unroll
    for (int j = 0; j < 32 / RPW; ++j) {
      const int r = RPW * j + orow;
      const int m = wave * 32 + r;                       // row inside the Q block
      const int tok = tok0 + m / G;
      const pu32x4_t v = *(const pu32x4_t*)(ost + r * ORS + och * 16);
      if (m < BQ * G && tok < q_len)
      {
        pu32x4_t* dst = (pu32x4_t*)(out_base + (int64_t)(q_start + tok) * p.out_stride_token + (int64_t)(head * G + m % G) * p.out_stride_head + och * 8);
        // nontemporal: O is written once and never read here; keeping it out of L2 leaves the cache to K/V and
        // nothing to write back when the kernel ends (+1.7 % at 1 x 4096 and 16 x 4096 over plain stores; sc1
        // write-through the same; the 8-byte pieces of the narrow path must NOT be nontemporal, -16 %)
        __builtin_nontemporal_store(v, dst);
      }
    }
  } else if (row_ok) {
    uint16_t* op = out_base + (int64_t)(q_start + tok_local) * p.out_stride_token + (int64_t)hq * p.out_stride_head + 4 * half;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const pu32x2_t w = {pmma<T>::pack2(o_acc[b][4 * c + 0] * inv, o_acc[b][4 * c + 1] * inv),
                            pmma<T>::pack2(o_acc[b][4 * c + 2] * inv, o_acc[b][4 * c + 3] * inv)};
        *(pu32x2_t*)(op + 32 * b + 8 * c) = w;
      }
  }
#ifdef MI355_PROFILE_WG
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MI355_WG_STAMP(wg_t3);
    MI355_WG_REALTIME(wg_r3);
    unsigned long long* dbg = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0);
    if (dbg && tid == 0) {
      atomicAdd(dbg + 8, wg_t1 - wg_t0);     // prologue: entry -> first tile staged
      atomicAdd(dbg + 15, wg_ta - wg_t0);    //   of which: entry -> sequence metadata known
      atomicAdd(dbg + 7, wg_tb - wg_ta);     //   of which: metadata -> block table staged, Q landed
      atomicAdd(dbg + 9, wg_t2 - wg_t1);     // tile loop
      atomicAdd(dbg + 10, wg_t3 - wg_t2);    // epilogue incl. store drain
      atomicAdd(dbg + 11, (unsigned long long)tile_hi);
      atomicAdd(dbg + 12, 1ull);
      atomicMin(dbg + 13, wg_t0);
      atomicMax(dbg + 14, wg_t3);
      // per-workgroup record for schedule reconstruction: entry, exit, where it ran, how many tiles
      const unsigned hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));     // HW_REG_HW_ID
      const unsigned xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
      unsigned long long* rec = dbg + 16 + 4ull * blockIdx.x;
      // rec[0..1]: s_memrealtime, 100 MHz, one counter for the whole chip
      rec[0] = wg_r0; rec[1] = wg_r3; rec[2] = ((unsigned long long)xcc_id << 32) | hw_id; rec[3] = (unsigned long long)tile_hi;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool paligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

bool prefill_supported(const mi355_attn_params& p) {
  if (!(p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16)) return false;
  const bool fp8_kv = p.kv_dtype == MI355_FP8_E4M3 || p.kv_dtype == MI355_FP8_E5M2;
  if (p.kv_dtype != p.q_dtype && !fp8_kv) return false;
  const int dpad = padded_head_size(p.head_size, fp8_kv);
  if (dpad == 0) return false;
  if ((p.k_new || p.v_new) && !p.write_new_kv) return false;      // (write_new_kv: the fused cache write of prefill_lat_kernel)
  if (p.page_size < 16 || (p.page_size & (p.page_size - 1)) != 0) return false;        // power of two, >= 16
  if (p.k_x != p.head_size || p.k_stride_d != 1 || p.v_stride_d != 1) return false;  // flash layout only
  if (p.k_stride_slot >= (1 << 24) || p.v_stride_slot >= (1 << 24)) return false;      // 32-bit in-page offsets
  if (p.k_stride_page >= (1LL << 31) || p.v_stride_page >= (1LL << 31) || p.k_stride_page < 0 || p.v_stride_page < 0 ||
      p.k_stride_slot < 0 || p.v_stride_slot < 0) return false;
  const int G = p.num_q_heads / p.num_kv_heads;
  if (G > kBlockM) return false;
  if (!paligned16(p.q) || !paligned16(p.k_cache) || !paligned16(p.v_cache)) return false;
  if (((uintptr_t)p.out & 7) != 0) return false;
  if (p.q_stride_token % 8 != 0 || p.q_stride_head % 8 != 0) return false;
  const int64_t kv_align = fp8_kv ? 16 : 8;   // elements per 16 bytes
  const int64_t kv_strides[] = {p.k_stride_page, p.k_stride_slot, p.k_stride_head, p.v_stride_page, p.v_stride_slot, p.v_stride_head};
  for (int64_t s : kv_strides) if (s % kv_align != 0) return false;
  if (p.out_stride_token % 4 != 0 || p.out_stride_head % 4 != 0) return false;
  return true;
}

// ---------------------------------------------------------------------------------------------
// Key-split prefill: few Q blocks over a long context (one sequence's 512-token chunk against 8k keys gives 128
// workgroups for 256 CUs, each walking 128 tiles). The key tiles are then dealt to `splits` workgroups per Q block
// (grid.y), each writes a normalised partial output (query type) and its lse (f32) to the workspace, and
// merge_key_splits_kernel folds them: out = sum_s out_s * exp(lse_s - lse), lse = log sum_s exp(lse_s) - the merge
// of reduce_segments (:804-828) on normalised partials. Both kernels below take it (any feature set, fp8 KV, head
// sizes 64..256); whether and how far to split is chosen from host-known sizes only.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxKeySplits = 8;
constexpr size_t kWsCounterBytes = (size_t)256 << 10;   // head of every workspace: the decode kernels' counters

// LDS bytes of the staged block-table prefix (8-wave kernel): one int per page of the longest sequence, in 256-byte rows
static constexpr size_t bt_lds_max_bytes(int nst) { return (size_t)(160 - 4 - 32 * nst) << 10; }   // what the stages leave of 160 KiB
static size_t prefill_bt_lds_bytes(const mi355_attn_params& p) {
  const size_t entries = ((size_t)std::max(p.max_seqlen_k, 1) + p.page_size - 1) / p.page_size;
  return ((entries + 63) / 64) * 256;
}

struct KeySplitPlan { int splits, tiles_per_split; bool wide; };   // wide: on the 8-wave / 256-row LDS-DMA kernel

static KeySplitPlan plan_key_splits(const mi355_attn_params& p) {
  static const char* env_s = lab_env("MI355_PREFILL_KEY_SPLITS");   // measurements: force a split count (1 = never split)
  // ... or the caller's num_segments (include/mi355_attn.h): 1 = one pass over the whole key range per Q block
  char forced_buf[16];
  const char* env = env_s;
  if (!env && p.num_segments > 0) { snprintf(forced_buf, sizeof(forced_buf), "%d", p.num_segments); env = forced_buf; }
  const int G = p.num_q_heads / p.num_kv_heads, block_q = kBlockM / G;
  const long wgs = ((long)p.num_tokens / block_q + p.num_seqs) * p.num_kv_heads;
  const int tiles = (std::max(p.max_seqlen_k, 1) + kTileN - 1) / kTileN;
  int splits = 1;
  bool wide = false;
  // The 8-wave kernel holds one workgroup per CU and is the faster one once it has >= 512 of them (launch_prefill);
  // with fewer, over >= 4096 keys, key splits give it the items instead of handing the call to the 4-wave kernel:
  // shares of >= 32 tiles, as many as bring the grid to 512 (one sequence, Hq 32 / Hk 8, sustained TFLOP/s, 4-wave kernel
  // with its own split plan -> this: 2048-token chunk at 32k keys 1054 -> 1176; 3072: 991 -> 1113; 1024: 1035 -> 1165;
  // 1024-token chunk at 8k keys 919 -> 988; 512: 858 -> 954).
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  const long wgs8 = ((long)p.num_tokens * G / 256 + p.num_seqs) * p.num_kv_heads;
  // an fp8 cache that prefill_pw_kernel reads itself (its KV8 instantiations, round 4): the wide plan, like a 16-bit cache
  const bool fp8_direct = p.kv_dtype != p.q_dtype && prefill_pw_applicable(p);
  if (env) {
    splits = atoi(env);
  } else if (!feat && p.head_size == 128 && (p.kv_dtype == p.q_dtype || fp8_direct) && wgs8 < 512 && tiles >= 64 &&
             prefill_bt_lds_bytes(p) <= bt_lds_max_bytes(3)) {
    splits = (int)std::min<long>(std::min<long>((512 + wgs8 - 1) / wgs8, tiles / 32), kMaxKeySplits);
    wide = splits > 1;
  }
  if (wide) {
    const int tps8 = (tiles + splits - 1) / splits;
    return {(tiles + tps8 - 1) / tps8, tps8, true};
  }
  // ... and where the wide plan does not split, one pass on that kernel rather than key splits on the register-staged one
  if (fp8_direct && !env && p.max_seqlen_k >= 2048) return {1, tiles, false};
  // the 4-wave kernel (and the register-staged one: features, fp8 KV, other head sizes):
  // two workgroups per CU, >= 8 tiles each (one sequence, Hq 32 / Hk 8: 512-token chunk at 8k keys 167 -> 79 us with 4
  // splits, at 32k keys 655 -> 275; two such chunks 112 -> 79 with 2; a 1024-token chunk at 32k keys 652 -> 534 with 2)
  // (392 / 408 workgroups: +10 % / +8 % with 2 splits; 520: nothing; 1040: -14 %)
  if (!env && wgs < 512 && tiles >= 32 && !p.non_causal) splits = (int)std::min<long>((512 + wgs - 1) / wgs, tiles / 8);
  if (p.non_causal) splits = 1;                       // (only the wide plan above splits a non-causal call: its kernel is the one that serves it)
  splits = std::max(1, std::min(std::min(splits, kMaxKeySplits), tiles));
  const int tps = (tiles + splits - 1) / splits;
  return {(tiles + tps - 1) / tps, tps, false};
}

struct KeySplitLayout { size_t out_off, lse_off, total; int64_t out_split_stride, lse_split_stride; };
static KeySplitLayout key_split_layout(const mi355_attn_params& p, int splits) {
  KeySplitLayout l;
  l.out_split_stride = (int64_t)p.num_tokens * p.num_q_heads * p.head_size;   // elements
  l.lse_split_stride = (int64_t)p.num_tokens * p.num_q_heads;
  l.out_off = kWsCounterBytes;
  l.lse_off = l.out_off + (((size_t)splits * l.out_split_stride * 2 + 255) & ~(size_t)255);
  l.total = l.lse_off + (size_t)splits * l.lse_split_stride * sizeof(float);
  return l;
}

size_t prefill_workspace_bytes(const mi355_attn_params& p) {
  if (!prefill_supported(p)) return 0;
  const int splits = plan_key_splits(p).splits;
  if (splits > 1) return key_split_layout(p, splits).total;
  return prefill_pw_selected(p, nullptr) ? kWsCounterBytes : 0;     // the ticket counters of prefill_pw_kernel's item deal
}

struct MergeArgs {
  mi355_attn_params p;          // out / lse / strides of the caller
  const uint16_t* part_out;     // [splits][T][Hq][D]
  const float* part_lse;        // [splits][T][Hq]
  int splits;
  uint8_t* clear_flags;         // fix-up flags of a key-split f16 launch (PrefillArgs::only_flagged): zeroed here, or null
  int num_flags;
};

// one thread per 8 output elements of one (token, query head) row
template <typename T>
__global__ __launch_bounds__(256) void merge_key_splits_kernel(const MergeArgs a) {
  const mi355_attn_params& p = a.p;
  const int chunks = p.head_size >> 3;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t rows = (int64_t)p.num_tokens * p.num_q_heads;
  if (a.clear_flags && idx < a.num_flags) a.clear_flags[idx] = 0;      // (far fewer flags than threads: (T / block_q + S) * Hk)
  if (idx >= rows * chunks) return;
  const int64_t row = idx / chunks;
  const int c = (int)(idx % chunks);
  const int64_t tok = row / p.num_q_heads;
  const int hq = (int)(row % p.num_q_heads);
  if (p.cu_seqlens_q && tok >= p.cu_seqlens_q[p.num_seqs]) return;   // padding tokens past the last sequence belong to nobody
  if (p.skip_decodes && p.cu_seqlens_q) {      // rows of query_len <= skip_decodes sequences were not computed and must stay untouched
    const int seq = find_seq_by_token(p.cu_seqlens_q, p.num_seqs, (int)tok);
    if (p.cu_seqlens_q[seq + 1] - p.cu_seqlens_q[seq] <= p.skip_decodes) return;
  }
  float lse[kMaxKeySplits], m = -INFINITY;
#pragma unroll
  for (int s = 0; s < kMaxKeySplits; ++s) {
    lse[s] = s < a.splits ? a.part_lse[(int64_t)s * rows + row] : -INFINITY;
    m = fmaxf(m, lse[s]);
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, den = 0.0f;
#pragma unroll
  for (int s = 0; s < kMaxKeySplits; ++s) {
    if (s < a.splits && lse[s] > -INFINITY) {
      const float w = __expf(lse[s] - m);
      den += w;
      const pu32x4_t v = *(const pu32x4_t*)(a.part_out + ((int64_t)s * rows + row) * p.head_size + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float lo, hi;
        if constexpr (__is_same(T, bf16_t)) { lo = bf16_to_f32((uint16_t)(v[e] & 0xffff)); hi = bf16_to_f32((uint16_t)(v[e] >> 16)); }
        else { lo = f16_to_f32((uint16_t)(v[e] & 0xffff)); hi = f16_to_f32((uint16_t)(v[e] >> 16)); }
        acc[2 * e] += w * lo;
        acc[2 * e + 1] += w * hi;
      }
    }
  }
  const float inv = den > 0.0f ? 1.0f / den : 0.0f;        // a row no split saw a key for: 0 (and lse -inf)
  pu32x4_t o;
#pragma unroll
  for (int e = 0; e < 4; ++e) o[e] = pmma<T>::pack2(acc[2 * e] * inv, acc[2 * e + 1] * inv);
  uint16_t* op = (uint16_t*)p.out + tok * p.out_stride_token + (int64_t)hq * p.out_stride_head + 8 * c;
  if ((((uintptr_t)op) & 15) == 0) {
    *(pu32x4_t*)op = o;
  } else {                                                  // out is 8-byte aligned at least (prefill_supported)
    *(pu32x2_t*)op = pu32x2_t{o[0], o[1]};
    *(pu32x2_t*)(op + 4) = pu32x2_t{o[2], o[3]};
  }
  if (p.lse && c == 0) p.lse[tok * p.lse_stride_token + hq] = den > 0.0f ? m + __logf(den) : -INFINITY;
}


// The same merge as an entry point of its own (mi355_merge_attention_partials): the exchange step of a cross-GPU split-KV
// (context-parallel) call hands every rank R normalised partial outputs + lse's of disjoint key ranges.
int launch_merge_partials(const void* part_out, const float* part_lse, int parts, void* out, float* lse, int dtype, int num_tokens,
                          int num_q_heads, int head_size, int64_t out_stride_token, int64_t out_stride_head, int64_t lse_stride_token, hipStream_t stream) {
  if (parts < 1 || parts > kMaxKeySplits) { set_error("merge: %d partial results (1 .. %d are served)", parts, kMaxKeySplits); return MI355_ERR_UNSUPPORTED; }
  if (!(dtype == MI355_BF16 || dtype == MI355_F16) || head_size % 8 != 0 || ((uintptr_t)part_out & 15) != 0 || ((uintptr_t)out & 7) != 0 ||
      out_stride_token % 4 != 0 || out_stride_head % 4 != 0) {
    set_error("merge: bf16 / f16 partials, head size a multiple of 8, 16-byte aligned partials, 8-byte aligned rows of out");
    return MI355_ERR_UNSUPPORTED;
  }
  MergeArgs m;
  memset(&m.p, 0, sizeof(m.p));
  m.p.out = out; m.p.lse = lse;
  m.p.num_tokens = num_tokens; m.p.num_q_heads = num_q_heads; m.p.head_size = head_size;
  m.p.out_stride_token = out_stride_token; m.p.out_stride_head = out_stride_head; m.p.lse_stride_token = lse_stride_token;
  m.part_out = (const uint16_t*)part_out;
  m.part_lse = part_lse;
  m.splits = parts;
  m.clear_flags = nullptr;
  m.num_flags = 0;
  const int64_t work = (int64_t)num_tokens * num_q_heads * (head_size / 8);
  if (work == 0) return MI355_OK;
  const dim3 grid((unsigned)((work + 255) / 256));
  if (dtype == MI355_BF16) hipLaunchKernelGGL(merge_key_splits_kernel<bf16_t>, grid, dim3(256), 0, stream, m);
  else hipLaunchKernelGGL(merge_key_splits_kernel<f16_t>, grid, dim3(256), 0, stream, m);
  return check_hip(hipGetLastError(), "merge_key_splits_kernel launch");
}

// cache element type as a function of the query type
template <typename T> using kv_same = T;
template <typename T> using kv_e4m3 = e4m3_t;
template <typename T> using kv_e5m2 = e5m2_t;

template <typename T, typename KVT, int D, bool FEAT>
static int launch_prefill_t(const mi355_attn_params& p, hipStream_t stream, const KeySplitCtx* ks, uint8_t* only_flagged = nullptr) {
  PrefillArgs a;
  a.p = p;
  a.only_flagged = only_flagged;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.block_q = kBlockM / a.group;
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.d_valid = p.head_size;
  a.kv_same_strides = (p.k_stride_page == p.v_stride_page && p.k_stride_slot == p.v_stride_slot && p.k_stride_head == p.v_stride_head) ? 1 : 0;
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  const int qblocks = p.num_tokens / a.block_q + p.num_seqs;  // static upper bound (:886-889,:935-943)
  a.key_splits = ks ? ks->splits : 1;
  a.tiles_per_key_split = ks ? ks->tiles_per_split : 0;
  a.out_split_stride = ks ? ks->out_split_stride : 0;
  a.lse_split_stride = ks ? ks->lse_split_stride : 0;
  constexpr size_t lds = 2 * (size_t)kTileN * ((D * 2 + 16) + (D * 2 + 64));
  static std::atomic<uint64_t> lds_opt_in{0};
  const int rc0 = ensure_dynamic_lds((const void*)prefill_mfma_kernel<T, KVT, D, FEAT>, (int)lds, lds_opt_in, "hipFuncSetAttribute(prefill)");
  if (rc0 != MI355_OK) return rc0;
  hipLaunchKernelGGL((prefill_mfma_kernel<T, KVT, D, FEAT>), dim3(qblocks * p.num_kv_heads, a.key_splits), dim3(256), lds, stream, a);
  const int rc = check_hip(hipGetLastError(), "prefill_mfma_kernel launch");
  if (rc == MI355_OK) set_kernel_name(!__is_same(T, KVT) ? (FEAT ? "prefill_mfma_fp8_feat" : "prefill_mfma_fp8") : (FEAT ? "prefill_mfma_feat" : "prefill_mfma"));
  return rc;
}

template <typename T, int NW, int NST, int D = 128, bool WR = false>
static int launch_prefill_dma(const mi355_attn_params& p, hipStream_t stream, const KeySplitCtx* ks) {
  PrefillArgs a;
  a.p = p;
  a.only_flagged = nullptr;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.block_q = (NW * 32) / a.group;
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.d_valid = p.head_size;
  a.kv_same_strides = (p.k_stride_page == p.v_stride_page && p.k_stride_slot == p.v_stride_slot && p.k_stride_head == p.v_stride_head) ? 1 : 0;
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  const int qblocks = p.num_tokens / a.block_q + p.num_seqs;
  a.key_splits = ks ? ks->splits : 1;
  a.tiles_per_key_split = ks ? ks->tiles_per_split : 0;
  a.out_split_stride = ks ? ks->out_split_stride : 0;
  a.lse_split_stride = ks ? ks->lse_split_stride : 0;
  size_t lds = (size_t)NST * 2 * kTileN * (2 * D);   // NST stages of K + V tiles, unpadded
  if (NST >= 3) lds += prefill_bt_lds_bytes(p);   // + the block-table prefix (the caller checked that it fits)
  static std::atomic<uint64_t> lds_opt_in{0};
  const int rc0 = ensure_dynamic_lds((const void*)prefill_dma_kernel<T, NW, NST, D, WR>, (int)((size_t)NST * 2 * kTileN * (2 * D) + (NST >= 3 ? bt_lds_max_bytes(NST) : 0)),
                                     lds_opt_in, "hipFuncSetAttribute(prefill_dma)");
  if (rc0 != MI355_OK) return rc0;
  hipLaunchKernelGGL((prefill_dma_kernel<T, NW, NST, D, WR>), dim3(qblocks * p.num_kv_heads, a.key_splits), dim3(NW * 64), lds, stream, a);
  const int rc = check_hip(hipGetLastError(), "prefill_dma_kernel launch");
  if (rc == MI355_OK) set_kernel_name(D == 256 ? "prefill_mfma_d256" : "prefill_mfma");
  return rc;
}

// Whether launch_prefill hands the call to prefill_pw_kernel (prefill_pw.hip): bf16 or f16, D = 128, no soft-cap / ALiBi
// (a sliding window is served), 16-bit cache, and either >= 2048 keys or a key-split plan on the wide kernel. MI355_PREFILL=pw | d8 | d4 | v1 pins a kernel (measurements).
bool prefill_pw_selected(const mi355_attn_params& p, const KeySplitCtx* ks) {
  static const char* variant = lab_env("MI355_PREFILL");
  const bool v1 = variant && variant[0] == 'v' && variant[1] == '1';
  if (!prefill_supported(p) || !prefill_pw_applicable(p) || v1 || (variant && variant[0] != 'p')) return false;
  const bool pinned = variant && variant[0] == 'p';
  if (p.non_causal) return !ks || ks->wide;           // the one matrix-core kernel without the causal diagonal built in
  // (soft-cap and ALiBi: from 512 keys on - the alternative is the register-staged kernel's feature instantiation, which the
  // SC / AL instantiations beat at every short shape measured: 1 x 1024 42 / 31 us against 58, 8 x 512 55 / 44 against 66,
  // 1 x 1536 58 / 40 against 86)
  // (head sizes 96 / 64: the alternative is the register-staged kernel too - 1 x 1024 24 against 30 us, 4 x 1024 45 against 60
  // at 96; 1 x 1536 28 against 31 at 64)
  // (round 4, plain D = 128 with several sequences: from 1536 keys on, from 1024 with sixteen sequences or more - graph replay,
  // us, this kernel | the 8-wave | the 4-wave LDS-DMA kernel after their round-4 change: 4 x 1536 87.4 | 92.9 | 93.4, 16 x 1024 170.6 |
  // 180.6 | 178.2, 8 x 1024 92.8 | 94.8 | 93.9; but 8 x 512 42.7 | 36.5 | 34.0, 32 x 512 137.0 | 121.6 | 116.8)
  const int plain_min = p.num_seqs >= 16 ? 1024 : p.num_seqs >= 2 ? 1536 : 2048;
  const int min_keys = (p.softcap > 0.0f || p.alibi_slopes) ? 512 : (p.head_size == 96 || p.head_size == 80) ? 1024 : p.head_size == 64 ? 1536 : plain_min;
  const bool use_pw = ks ? ks->wide : p.max_seqlen_k >= min_keys;
  return pinned ? (!ks || ks->wide) : use_pw;
}

// launch_prefill (without key splits) hands this call to one of the two LDS-DMA kernels - plain bf16 / f16 attention at
// head size 128 over a 16-bit cache that neither prefill_pw_kernel nor prefill_lat_kernel takes, no kernel pinned otherwise.
// These and prefill_lat_kernel carry the fused cache write (write_new_kv).
bool prefill_dma_selected(const mi355_attn_params& p) {
  if (!prefill_supported(p) || prefill_pw_selected(p, nullptr) || prefill_lat_selected(p)) return false;
  static const char* const variant = lab_env("MI355_PREFILL");
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  if (variant && variant[0] == 'v') return false;
  return !feat && p.head_size == 128 && p.kv_dtype == p.q_dtype && !p.non_causal;
}

bool prefill_runs_without_key_splits(const mi355_attn_params& p) { return plan_key_splits(p).splits <= 1; }

int launch_prefill(const mi355_attn_params& p, hipStream_t stream, const KeySplitCtx* ks, int* counters) {
  if (!prefill_supported(p)) {
    set_error("prefill kernel does not support this configuration");
    return MI355_ERR_UNSUPPORTED;
  }
  const bool bf = p.q_dtype == MI355_BF16;
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  // A/B switches for measurements: MI355_PREFILL=v1 (register-staged), pw / d8 / d4 pin one of the three LDS-DMA kernels
  static const char* variant = lab_env("MI355_PREFILL");
  const bool v1 = variant && variant[0] == 'v' && variant[1] == '1';
  // prefill_pw_kernel (prefill_pw.hip: four waves of 64 rows, one per SIMD; bf16) wherever the 8-wave kernel was the
  // choice, and instead of the 4-wave kernel from 2048 keys on (sustained TFLOP/s, Hq 32 / Hk 8, pw | 8-wave | 4-wave:
  // 1 x 4096 1153 | 1068 | 933, 16 x 4096 1129 | 1032 | 978, 1 x 16384 1325 | 1188 | 1095, 1 x 3072 996 | 915 | 924,
  // 2 x 2048 962 | 920 | 849, 1 x 2048 670 | 624 | 674; below that a workgroup's prologue and epilogue (~9 us at one
  // workgroup per CU) outweigh its few tiles: 4 x 1024 628 | 636 | 663, 8 x 512 419 | 451 | 453, 1 x 1024 274 | 273 | 329).
  if (p.write_new_kv && prefill_pw_selected(p, ks)) { set_error("write_new_kv: a prefill step of this length is not served with a fused cache write (mi355_decode_write_fusable)"); return MI355_ERR_UNSUPPORTED; }
  if (prefill_pw_selected(p, ks)) {
    // f16: P = 2^(s - m_ref) leaves 22 log2 units (15 nats) above a row's reference - its first sixteen keys' maximum - so a
    // retrieval-style row whose needle key scores higher than that overflows. Such rows are FLAGGED by the launch (one byte
    // per 128-row Q block and KV head, in the tail of the workspace's zero-filled head) and their blocks computed again by
    // the register-staged kernel's true running maximum in a launch behind it, which leaves at once where nothing is flagged
    // (~2 us on a 4096-token prompt: f16 only - bf16's +-90 log2 units keep the in-launch per-row routine, which costs the
    // headline nothing). A whole query head of needle rows at 1 x 4096: within 1.2x of the launch without them, where the
    // per-row routine took ~1000x per row (tests/test_gpu_prefill.py::test_f16_rows_whose_scores_rise_late_...).
    const int G = p.num_q_heads / p.num_kv_heads;
    const long n_flags = G <= 128 ? ((long)p.num_tokens / (128 / G) + p.num_seqs) * p.num_kv_heads : 0;
    uint8_t* const flags = (p.q_dtype == MI355_F16 && counters && !p.non_causal && n_flags > 0 && (size_t)n_flags <= kWsFixFlagBytes)
                               ? (uint8_t*)counters + kWsFixFlagOffset : nullptr;
    int rc = launch_prefill_pw(p, ks ? ks->splits : 1, ks ? ks->out_split_stride : 0, ks ? ks->lse_split_stride : 0, counters, stream, flags);
    if (rc != MI355_OK || !flags) return rc;
    static thread_local char pw_name[48];
    snprintf(pw_name, sizeof(pw_name), "%s", mi355_last_kernel_name());
    const int dpad = padded_head_size(p.head_size, false);
    if (p.kv_dtype == MI355_FP8_E4M3) rc = launch_prefill_t<f16_t, e4m3_t, 128, false>(p, stream, ks, flags);        // (an fp8 cache: plain attention at head size 128 only, prefill_pw_applicable)
    else if (p.kv_dtype == MI355_FP8_E5M2) rc = launch_prefill_t<f16_t, e5m2_t, 128, false>(p, stream, ks, flags);
    else if (dpad == 64) rc = feat ? launch_prefill_t<f16_t, f16_t, 64, true>(p, stream, ks, flags) : launch_prefill_t<f16_t, f16_t, 64, false>(p, stream, ks, flags);
    else rc = feat ? launch_prefill_t<f16_t, f16_t, 128, true>(p, stream, ks, flags) : launch_prefill_t<f16_t, f16_t, 128, false>(p, stream, ks, flags);
    if (rc == MI355_OK) set_kernel_name(pw_name);
    return rc;
  }
  // Short prompts (round 4): prefill_lat_kernel (prefill_lat.hip: 64-row Q blocks, waves = 2 row halves x 2 or 4 key parts)
  // where the launch is about one workgroup per CU - the heaviest Q block's chain of key tiles IS the launch, and twice the
  // waves on it halve it - or one sequence of up to ~1500 tokens. One box, graph replay, us per launch (8 waves | 4 waves |
  // the 4-wave 128-row kernel below, which round 4 also taught to issue its LDS-DMA with one scalar base per tile and wave:
  // 1 x 512 16.4 -> 14.4, 8 x 512 37.8 -> 33.5): 1 x 256 9.5 | 9.1 | 10.1, 1 x 512 12.2 | 13.1 | 14.4, 1 x 768 19.5 | 17.6 | 18.8,
  // 1 x 1024 24.2 | 21.6 | 22.5, 1 x 1536 43.9 | 32.3 | 33.8, 4 x 128 tokens over 512 keys 14.1 | 17.5 | 19.0; but 2 x 512 20.3 | 17.0 | 15.4,
  // 3 x 512 30.5 | 22.1 | 19.2, 2 x 1024 48.2 | 33.4 | 31.5, 4 x 512 39.5 | 27.7 | 20.4, 16 x 128 34.1 | 19.8 | 11.7, 8 x 512 75.9 | 48.6 | 33.5
  // (under load a workgroup's prologue - metadata, query rows - queues behind everyone's tile streams: many short Q blocks
  // want the wider ones). MI355_PREFILL=lat pins it wherever it applies.
  if (!ks && prefill_lat_selected(p)) return launch_prefill_lat(p, stream);
  if (p.write_new_kv && (ks || !prefill_dma_selected(p))) { set_error("write_new_kv: this prefill step is not served with a fused cache write (mi355_decode_write_fusable)"); return MI355_ERR_UNSUPPORTED; }
  if (!feat && p.head_size == 128 && !v1 && p.kv_dtype == p.q_dtype) {
    // 8 waves / 256-row Q blocks / 3 stages when that still gives every CU two workgroups' worth of Q blocks
    // (it holds one at a time) and the sequences are long enough to amortise a workgroup's un-overlapped
    // prologue (sustained rates, wide vs narrow: 1 x 4096 1009 vs 888 TFLOP/s, 4 x 4096 1024 vs 968, 1 x 8192 1081 vs
    // 1003, 2 x 2048 896 vs 821; but 4 x 1024 610 vs 655, 1 x 2048 592 vs 660, 1 x 3072 856 vs 882);
    // otherwise 4 waves / 128-row Q blocks / 2 stages, two workgroups per CU.
    // MI355_PREFILL=d4 | d8 pins one of the two for measurements.
    const long wgs8 = ((long)p.num_tokens * (p.num_q_heads / p.num_kv_heads) / 256 + p.num_seqs) * p.num_kv_heads;
    bool wide = wgs8 >= 2 * 256 && p.max_seqlen_k >= 2048;
    if (ks) wide = ks->wide;
    if (variant && variant[0] == 'd') wide = variant[1] == '8';
    if (p.write_new_kv) {
      if (wide && prefill_bt_lds_bytes(p) <= bt_lds_max_bytes(3))
        return bf ? launch_prefill_dma<bf16_t, 8, 3, 128, true>(p, stream, ks) : launch_prefill_dma<f16_t, 8, 3, 128, true>(p, stream, ks);
      return bf ? launch_prefill_dma<bf16_t, 4, 2, 128, true>(p, stream, ks) : launch_prefill_dma<f16_t, 4, 2, 128, true>(p, stream, ks);
    }
    if (wide && prefill_bt_lds_bytes(p) <= bt_lds_max_bytes(3))
      return bf ? launch_prefill_dma<bf16_t, 8, 3>(p, stream, ks) : launch_prefill_dma<f16_t, 8, 3>(p, stream, ks);
    return bf ? launch_prefill_dma<bf16_t, 4, 2>(p, stream, ks) : launch_prefill_dma<f16_t, 4, 2>(p, stream, ks);
  }
  // (Head size 256 on the LDS-DMA form - prefill_dma_kernel<T, 4, 2, 256>, one workgroup per CU - was built in round 3 and
  // measured 112 TFLOP/s against the register-staged kernel's 494: without staging registers hipcc still moves several
  // hundred values per tile between VGPRs and accumulator registers (432 v_accvgpr_read + 225 v_accvgpr_write per tile
  // in the ISA, 750 bytes of scratch). Head size 256 needs the hand-owned register file of prefill_pw_kernel.)
#define MI355_PREFILL_CASE(KV, DD)                                                                        \
  case DD:                                                                                                \
    if (feat) return bf ? launch_prefill_t<bf16_t, KV<bf16_t>, DD, true>(p, stream, ks) : launch_prefill_t<f16_t, KV<f16_t>, DD, true>(p, stream, ks); \
    return bf ? launch_prefill_t<bf16_t, KV<bf16_t>, DD, false>(p, stream, ks) : launch_prefill_t<f16_t, KV<f16_t>, DD, false>(p, stream, ks);
  const int dpad = padded_head_size(p.head_size, p.kv_dtype != p.q_dtype);
  if (p.kv_dtype == MI355_FP8_E4M3) {
    switch (dpad) {
      MI355_PREFILL_CASE(kv_e4m3, 64)
      MI355_PREFILL_CASE(kv_e4m3, 128)
      MI355_PREFILL_CASE(kv_e4m3, 256)
    }
  } else if (p.kv_dtype == MI355_FP8_E5M2) {
    switch (dpad) {
      MI355_PREFILL_CASE(kv_e5m2, 64)
      MI355_PREFILL_CASE(kv_e5m2, 128)
      MI355_PREFILL_CASE(kv_e5m2, 256)
    }
  } else {
    switch (dpad) {
      MI355_PREFILL_CASE(kv_same, 64)
      MI355_PREFILL_CASE(kv_same, 128)
      MI355_PREFILL_CASE(kv_same, 256)
    }
  }
#undef MI355_PREFILL_CASE
  set_error("prefill: head_size %d not built", p.head_size);
  return MI355_ERR_UNSUPPORTED;
}

// Prefill with the caller's workspace at hand: key-split launch + merge when the plan says so, the plain launch otherwise.
int launch_prefill_ws(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!prefill_supported(p)) {
    set_error("prefill kernel does not support this configuration");
    return MI355_ERR_UNSUPPORTED;
  }
  const KeySplitPlan plan = plan_key_splits(p);
  int* const counters = (ws && ws_bytes >= kWsCounterBytes) ? (int*)ws : nullptr;   // zero between calls (decode_splitkv.hip)
  if (plan.splits <= 1) return launch_prefill(p, stream, nullptr, counters);
  const KeySplitLayout lay = key_split_layout(p, plan.splits);
  if (!ws || ws_bytes < lay.total) {
    set_error("workspace of %zu bytes is smaller than the %zu the key-split prefill needs (mi355_attn_workspace_bytes)", ws_bytes, lay.total);
    return MI355_ERR_WORKSPACE;
  }
  mi355_attn_params pp = p;                         // partial outputs: contiguous [T, Hq, D] / [T, Hq] per split
  pp.out = (char*)ws + lay.out_off;
  pp.out_stride_token = (int64_t)p.num_q_heads * p.head_size;
  pp.out_stride_head = p.head_size;
  pp.lse = (float*)((char*)ws + lay.lse_off);
  pp.lse_stride_token = p.num_q_heads;
  const KeySplitCtx ks = {plan.splits, plan.tiles_per_split, plan.wide, lay.out_split_stride, lay.lse_split_stride};
  int rc = launch_prefill(pp, stream, &ks, counters);
  if (rc != MI355_OK) return rc;
  MergeArgs m;
  m.p = p;
  m.part_out = (const uint16_t*)((char*)ws + lay.out_off);
  m.part_lse = (const float*)((char*)ws + lay.lse_off);
  m.splits = plan.splits;
  {   // the fix-up flags of a key-split f16 launch on prefill_pw_kernel (launch_prefill): served by every split, cleared here
    const int G = p.num_q_heads / p.num_kv_heads;
    const long n_flags = G <= 128 ? ((long)p.num_tokens / (128 / G) + p.num_seqs) * p.num_kv_heads : 0;
    const bool used = p.q_dtype == MI355_F16 && counters && !p.non_causal && n_flags > 0 && (size_t)n_flags <= kWsFixFlagBytes && prefill_pw_selected(pp, &ks);
    m.clear_flags = used ? (uint8_t*)counters + kWsFixFlagOffset : nullptr;
    m.num_flags = used ? (int)n_flags : 0;
  }
  const int64_t work = (int64_t)p.num_tokens * p.num_q_heads * (p.head_size / 8);
  const dim3 grid((unsigned)((work + 255) / 256));
  if (p.q_dtype == MI355_BF16) hipLaunchKernelGGL(merge_key_splits_kernel<bf16_t>, grid, dim3(256), 0, stream, m);
  else hipLaunchKernelGGL(merge_key_splits_kernel<f16_t>, grid, dim3(256), 0, stream, m);
  rc = check_hip(hipGetLastError(), "merge_key_splits_kernel launch");
  if (rc == MI355_OK) {
    static thread_local char name[64];
    snprintf(name, sizeof(name), "%s_ksplit", mi355_last_kernel_name());
    set_kernel_name(name);
  }
  return rc;
}

}  // namespace mi355
