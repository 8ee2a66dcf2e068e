// Paged-KV decode for gfx950: split-KV partials + merge.
//
// Replaces kernel_unified_attention_3d + reduce_segments
// (LIB/kernels/triton_unified_attention.py:526-754, :757-836) and, through the same math, the legacy
// paged_attention_2d/3d kernels. HBM-bound: every K/V byte is read exactly once and only
// G = Hq/Hk query rows reuse it, so the design goal is bytes in flight, not FLOPs.
//
// Work decomposition: one WAVE per (query token, KV head, KV split); no inter-wave communication,
// no barriers. A wave walks its KV range in tiles of 32 keys (two 16-key groups; a group never
// straddles a page because page_size is a power of two >= 16):
//   * K and V of the NEXT tile are fetched HBM -> VGPR with row-shaped 16-byte loads (each load
//     instruction covers whole key rows, so every 128-byte line is fetched once) while the current
//     tile is computed; the block-table entries of the tile after that are fetched through the
//     scalar cache, so no dependent load sits in front of the stream;
//   * an fp8 (e4m3fn / e5m2) cache is widened to the query's 16-bit type on the way into LDS
//     (v_cvt_pk_f32_fp8 -> v_cvt_pk_{bf16,f16}_f32; exact, every fp8 value is representable) and the
//     scalar k/v scales are folded into the softmax scale and the output normalisation;
//   * rows are parked in wave-private LDS (row stride 2*D+32 bytes: conflict-free for both read
//     kinds) and read back as MFMA operands: K with ds_read_b128, V with the transposing
//     ds_read_b64_tr_b16;
//   * S^T = K . Q^T and O^T += V^T . P^T on v_mfma_f32_16x16x32_{bf16,f16}. The G query heads of the
//     KV head are the 16 MFMA columns (GQA broadcast costs nothing: K/V are read once for all of
//     them). In this orientation lane (g = lane&15, grp = lane>>4) owns query head g in S, P and O,
//     so the online-softmax state (m, l, alpha) is lane-local; only the running max needs a
//     4-lane exchange (lanes g, g+16, g+32, g+48), done with v_permlane{16,32}_swap.
//   * the accumulator tile of S feeds the P.V MFMA without any lane movement: k-slot j of lane
//     group grp stands for key 4*grp + (j&3) of 16-key group (j>>2), and the transposed V read is
//     addressed to deliver exactly those keys.
// Partials (m, l, un-normalised O) go to a caller-provided workspace and are merged by
// reduce_splits_kernel; with a single split the wave normalises and stores the output itself.
// In `only_decodes` mode the grid runs over SEQUENCES (one query token each), which is how a mixed
// batch hands its decode rows to this kernel without paying for its prefill tokens.
#include <algorithm>
#include <cstdlib>

#include <type_traits>
#include <utility>

#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#ifndef DECODE_TU
#define DECODE_TU 0
#endif
constexpr int kTileKeys = 32;
constexpr int kMaxSplits = 128;
constexpr int kSlotPad = 32;   // floats appended to a D-float partial (m, l, padding to a 128-byte multiple)
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

struct DecodeArgs {
  mi355_attn_params p;
  // split-KV scratch: slot (token*Hq + hq)*num_splits + split = kSlotPad+D floats: un-normalised
  // partial output [0,D), running max in the log2 domain [D], partial sum [D+1]; slots are multiples
  // of 128 bytes so that no cache line is shared between the partials of different merge groups
  float* ws_slots;
  int* ws_cnt;     // [units*Hk] arrival counters of the in-kernel merge; zero on entry, zero on exit
  uint32_t ws_slot_bytes_total;
  int fused_merge; // 1: the last-arriving split of a (unit, KV head) merges in the kernel, no second launch
  int tree;        // 1: two-level merge (round 4). The four waves of a workgroup are four CONSECUTIVE splits of one (unit, KV head,
                   // query-head group): they fold their partials through LDS, the workgroup writes ONE partial (slot rows of
                   // num_splits / 4), and the last workgroup to arrive folds those with all four waves - any split count in one launch
  int num_splits;
  int tiles_per_split;
  int group;       // G = Hq / Hk
  int qgroups;     // cdiv(G, 16): a wave serves 16 of the KV head's query heads; more of them take more waves
  int page_shift;  // log2(page_size)
  int by_seq;      // 1: work items enumerate sequences (only_decodes), 0: query tokens
  int d_valid;     // the real head size; columns d_valid .. D-1 of the kernel's head size are padding
  int unit_is_seq; // 1: every sequence has exactly one query token (max_seqlen_q == 1), so unit == token == sequence
  int pack_tokens; // PACK kernels: query tokens of one sequence per work unit (a power of two <= 16 / G), else 0
  int pack_shift;  // log2(pack_tokens)
  int pack_cps;    // chunks (work units) per sequence: cdiv(max_seqlen_q, pack_tokens)
  uint32_t k_page_stride, k_slot_stride, v_page_stride, v_slot_stride;  // elements; validated on the host
  uint32_t k_dx_stride, v_d_stride;   // legacy v0 layout only: K [page][Hk][D/8][slot][8], V [page][Hk][D][slot]
};

template <typename T> struct mma;
template <> struct mma<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
};
template <> struct mma<f16_t> {
  static __device__ __forceinline__ f32x4_t run(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
};

// 16 fp8 values (one 16-byte load) -> 16 values of the query's 16-bit type (two 16-byte LDS pieces)
template <typename T, typename KVT>
__device__ __forceinline__ void widen_fp8x16(u32x4_t in, u32x4_t& lo, u32x4_t& hi) {
  uint32_t o[8];
#pragma unroll
  for (int w = 0; w < 4; ++w) widen_fp8x4<T, KVT>(in[w], o[2 * w], o[2 * w + 1]);
  lo = u32x4_t{o[0], o[1], o[2], o[3]};
  hi = u32x4_t{o[4], o[5], o[6], o[7]};
}

// (the quantising store of the fused cache write: quantise_fp8x16, common.h)

// max / sum over the four lanes {g, g+16, g+32, g+48}
__device__ __forceinline__ float max_over_lane_groups(float v) {
  v = fmaxf(v, lane_xor32(v));
  return fmaxf(v, lane_xor16(v));
}
__device__ __forceinline__ float sum_over_lane_groups(float v) {
  v += lane_xor32(v);
  return v + lane_xor16(v);
}

// f(integral_constant<0>) ... f(integral_constant<N-1>), in order
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// tiles one split of a row walks: the row's own tiles dealt evenly over the (host-fixed) number of splits
__device__ __forceinline__ int split_tiles(int n_tiles, int num_splits) { return max(1, (n_tiles + num_splits - 1) / num_splits); }

// Shared by both kernels: which token a work unit is, and the key range it may see.
struct RowInfo {
  int token, seq, q_len, ctx_len, q_pos, n_keys, first_key;
  bool valid;
};
__device__ __forceinline__ RowInfo row_info(const mi355_attn_params& p, int by_seq, int unit) {
  RowInfo r;
  r.valid = false;
  r.token = r.seq = r.q_len = r.ctx_len = r.q_pos = r.n_keys = r.first_key = 0;
  if (by_seq) {
    r.seq = unit;
    if (unit >= p.num_seqs) return r;
    r.token = p.cu_seqlens_q[unit];
  } else {
    r.token = unit;
    if (unit >= p.num_tokens) return r;
    r.seq = find_seq_by_token(p.cu_seqlens_q, p.num_seqs, unit);
  }
  const int q_start = p.cu_seqlens_q[r.seq];
  r.q_len = p.cu_seqlens_q[r.seq + 1] - q_start;
  if (by_seq && r.q_len != 1) return r;
  if (r.q_len <= p.skip_decodes) return r;
  if (p.only_decodes && r.q_len > p.only_decodes) return r;
  const int seq_len = p.seqused_k[r.seq];
  r.ctx_len = seq_len - r.q_len;
  r.q_pos = r.token - q_start;
  if (r.q_pos >= r.q_len) return r;   // a token row past the batch's last sequence (padding of a captured graph): untouched, like the reference's (:307-327)
  r.n_keys = max(0, min(r.ctx_len + r.q_pos + 1, seq_len));   // causal
  if (p.sliding_window > 0) r.first_key = max(0, r.ctx_len + r.q_pos - p.sliding_window + 1);  // keep j with q_abs - j < window
  r.valid = true;
  return r;
}

// PAD: the real head size is smaller than D (see padded_head_size); a separate instantiation so that the
// built head sizes keep their branch-free load stream.
// V0: the legacy vLLM v0 cache layout (K [page][Hk][D/8][slot][8], V [page][Hk][D][slot], 16-bit). A (page, head)
// tile is one contiguous block there and its 16-byte pieces are exactly what the kernel wants: a K piece is 8
// consecutive d of one key (parked like a flash-layout piece, only the global offset differs), a V piece is 8
// consecutive keys of one d - V is parked d-major and the P.V operand is read with two plain 8-byte reads, no transpose.
// PACK: multi-token decode steps (speculative decoding / MTP verification: a few query tokens per sequence over a long
// context). The 16 matrix columns of a wave hold the G query heads of pack_tokens = 16 / G (rounded down to a power of
// two) CONSECUTIVE query tokens of one sequence instead of the G heads of one token: the sequence's K/V is streamed
// once per pack_tokens tokens, and only the last tiles differ between the columns (column of token i sees keys up to
// ctx_len + i: a per-lane key limit in the tail tiles' mask). Work units are (sequence, chunk of pack_tokens tokens),
// cdiv(max_seqlen_q, pack_tokens) per sequence - ONE in the dispatched case, so that unit == sequence and the launch
// has no empty units (the reference's Q-block enumeration, triton_unified_attention.py:32-52 with sequence i's blocks
// starting at cu_seqlens_q[i] / BLOCK_Q + i, leaves every other unit empty on a batch of equal query lengths: with
// the workgroups dealt round-robin to the XCDs half of them then stood idle, 64 x 4 tokens x 8192 keys 559 us against
// 380); partials and outputs are addressed per column (token, head).
// PACK = 2: TWO column groups per wave (32 columns: twice the tokens per unit, or G up to 16) - the K / V fragments a
// lane reads from LDS feed two matrix instructions, the score / softmax work per tile doubles (head sizes up to 128).
template <typename T, typename KVT, int D, int WAVES, bool FEAT, bool PAD, bool V0, int PACK = 0>
__global__ __launch_bounds__(WAVES * 64) void decode_splitkv_kernel(const DecodeArgs a) {
  static_assert(!V0 || (__is_same(T, KVT) && !PAD), "the v0 layout path serves same-type 16-bit caches of a built head size");
  static_assert(!PACK || (!PAD && !V0), "packed query tokens: a flash-layout cache of a built head size");
  static_assert(PACK != 2 || !FEAT, "two column groups: plain attention only");
  static_assert(PACK != 2 || D <= 128, "two column groups: O, Q and the K/V tiles in flight fit the register file up to head size 128");
  constexpr int NCG = PACK == 2 ? 2 : 1;            // column groups (16 matrix columns each) per wave
  constexpr bool FP8 = !__is_same(T, KVT);
  // tiles in flight HBM -> VGPR per wave. Rounds 2-3 kept TWO fp8 tiles in flight (half the bytes of a 16-bit tile each);
  // re-measured in round 4 on the kernel as it is now (nt loads, hardware converts), ONE is 3 % faster wherever the fp8 kernel
  // streams - C5 178.8 -> 173.3 us, 64 x 8192 177.5 -> 172.1, 256 x 2048 169.5 -> 164.6, batch 1 .. 4 unchanged
  // (profiles/r04/decode_experiments.log); three: 224 us. -DMI355_DECODE_PF=n (lab build) sets another depth.
#ifdef MI355_DECODE_PF
  constexpr int PF = MI355_DECODE_PF;
#else
  constexpr int PF = 1;
#endif
  constexpr int KVB = FP8 ? 1 : 2;                  // bytes per cache element
  constexpr int PPR = D * KVB / 16;                 // 16-byte pieces per key row in HBM
  constexpr int NLD = (16 * PPR) / 64;              // row-shaped loads per 16-key group per lane
  constexpr int EPP = 16 / KVB;                     // elements per piece
  constexpr int RS = D * 2 + 32;                    // LDS row stride in bytes (16-bit rows)
  constexpr int KSTEPS = D / 32;                    // MFMA k-steps of Q.K^T
  constexpr int DBLK = D / 16;                      // 16-wide output blocks of P.V
  constexpr int RSV0 = 80;                          // v0: bytes per d-row of the parked V tile (32 keys + pad)
  constexpr int LDS_PER_WAVE = 16 * RS + (V0 ? D * RSV0 : 32 * RS);   // K: one 16-key group; V: two groups (or D rows of 32 keys)
  static_assert(NLD >= 1, "head size too small for the row-shaped load");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mi355_attn_params& p = a.p;
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* k_lds = smem + wave_in_wg * LDS_PER_WAVE;
  char* v_lds = k_lds + 16 * RS;

  // ---- which (unit, split, kv head) this wave owns (all wave-uniform) ----------------------------
  const int Hk = p.num_kv_heads;
  const bool tree = PACK == 0 && a.tree;
  static_assert(WAVES == 4, "the two-level merge folds the four waves of a workgroup");
  // (tree: the workgroup is (unit, group of four splits, KV head, query-head group) and its waves are the group's splits)
  const int item = __builtin_amdgcn_readfirstlane(tree ? (int)blockIdx.x : (int)(blockIdx.x * WAVES + wave_in_wg));
  // (the waves of one KV head's query-head groups are neighbours: they stream the same K/V pages through one L2)
  const int hitem = item % (Hk * a.qgroups);
  const int head = hitem / a.qgroups, qg = hitem % a.qgroups;
  const int rest = item / (Hk * a.qgroups);
  const int nsp = tree ? a.num_splits >> 2 : a.num_splits;
  const int split = tree ? (rest % nsp) * 4 + wave_in_wg : rest % nsp;
  const int unit = rest / nsp;
  const int hq0 = head * a.group + 16 * qg;   // first query head of this wave
  const int G = min(16, a.group - 16 * qg);   // query heads of this wave
  const int g = lane & 15, grp = lane >> 4;
  // PACK: column c = 16 * cg + g = (token tq of the unit's chunk, query head c % G)
  int tq[NCG], hq[NCG];
  bool g_ok[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    const int c = 16 * cg + g;
    tq[cg] = PACK ? c / G : 0;
    hq[cg] = hq0 + c - tq[cg] * G;
    g_ok[cg] = PACK ? tq[cg] < a.pack_tokens : g < G;
  }

  // ---- Q fragments: B operand of S^T = K.Q^T: lane (g, grp) holds Q[g][32c + 8grp .. +7] --------
  s16x8_t qf[NCG][KSTEPS];
  auto load_q = [&](int cg, int tok) {   // issued as early as the token is known: rides the same round trip as the lookups (PACK: a per-lane token)
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)tok * p.q_stride_token + (int64_t)hq[cg] * p.q_stride_head + 8 * grp;
#pragma unroll
    for (int c = 0; c < KSTEPS; ++c) {
      u32x4_t v = {0, 0, 0, 0};
      if (g_ok[cg] && (!PAD || 32 * c + 8 * grp < a.d_valid)) v = *(const u32x4_t*)(qp + 32 * c);
      qf[cg][c] = __builtin_bit_cast(s16x8_t, v);
    }
  };

  // Every dependent memory round trip ahead of the first K/V load costs a wave ~2 us on an idle chip
  // and several times that when the chip is streaming, with none of its bytes in flight meanwhile.
  // A pure decode batch (unit == token == sequence) therefore takes ONE scalar round trip for the
  // sequence length AND the first tile's two block-table entries (looked up before the length is
  // known, inside the row's first cdiv(max_seqlen_k, page) entries); the general path searches
  // cu_seqlens_q first.
  RowInfo ri;
  int pg_first[2] = {0, 0};
  int slot_sign = 0;           // write_new_kv: < 0 marks a padding row whose K/V must not reach the cache (triton_attn.py:149-151)
  const bool fast_head = !FEAT && a.unit_is_seq;
  if (fast_head) {
    if (unit >= p.num_seqs || p.skip_decodes) return;
    load_q(0, unit);
    const int t0s = split * a.tiles_per_split;
    // (speculative: the entries are used only if this row's own split size puts the split where the host's did; any
    // index inside the row is safe to read)
    const int last_blk = max(0, min(((p.max_seqlen_k + p.page_size - 1) >> a.page_shift) - 1, (int)p.block_table_stride - 1));
    int seq_len;
    if (p.write_new_kv && (p.slot_mapping || p.slot_mapping_i32)) {
      // the word of this row's slot that carries its sign (int64: the high half) rides the same round trip
      const int32_t* sw = p.slot_mapping ? (const int32_t*)(p.slot_mapping + unit) + 1 : p.slot_mapping_i32 + unit;
      scalar_load_2words_and_pair(p.seqused_k + unit, sw, p.block_table + (int64_t)unit * p.block_table_stride,
                                  min((t0s * 2 * 16) >> a.page_shift, last_blk), min(((t0s * 2 + 1) * 16) >> a.page_shift, last_blk),
                                  seq_len, slot_sign, pg_first[0], pg_first[1]);
    } else
    scalar_load_word_and_pair(p.seqused_k + unit, p.block_table + (int64_t)unit * p.block_table_stride,
                              min((t0s * 2 * 16) >> a.page_shift, last_blk), min(((t0s * 2 + 1) * 16) >> a.page_shift, last_blk),
                              seq_len, pg_first[0], pg_first[1]);
    ri.token = ri.seq = unit;
    ri.q_len = 1; ri.q_pos = 0; ri.first_key = 0;
    ri.ctx_len = seq_len - 1;
    ri.n_keys = max(0, seq_len);
    ri.valid = true;
  } else if constexpr (PACK) {
    ri.seq = a.pack_cps == 1 ? unit : unit / a.pack_cps;
    if (ri.seq >= p.num_seqs) return;
    const int q_start = p.cu_seqlens_q[ri.seq];
    ri.q_len = p.cu_seqlens_q[ri.seq + 1] - q_start;
    ri.q_pos = (unit - ri.seq * a.pack_cps) << a.pack_shift;                       // first token of the chunk
    if (ri.q_pos >= ri.q_len) return;                                              // (a sequence shorter than max_seqlen_q)
    if (p.only_decodes && ri.q_len > p.only_decodes) return;                       // (a mixed batch: the prefill launch's rows)
    const int seq_len = p.seqused_k[ri.seq];
    const int rows = min(a.pack_tokens, ri.q_len - ri.q_pos);
    ri.ctx_len = seq_len - ri.q_len;
    ri.token = q_start + ri.q_pos;
    ri.n_keys = max(0, min(ri.ctx_len + ri.q_pos + rows, seq_len));               // the chunk's last token sees the most
    ri.first_key = (FEAT && p.sliding_window > 0) ? max(0, ri.ctx_len + ri.q_pos - p.sliding_window + 1) : 0;   // the chunk's FIRST token's (the lowest)
    ri.valid = true;
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      g_ok[cg] = g_ok[cg] && tq[cg] < rows;
      load_q(cg, ri.token + (g_ok[cg] ? tq[cg] : 0));
    }
  } else {
    ri = row_info(p, a.by_seq, unit);
    if (!ri.valid) return;
    load_q(0, ri.token);
    if (p.write_new_kv && (p.slot_mapping || p.slot_mapping_i32))
      slot_sign = p.slot_mapping ? ((const int32_t*)(p.slot_mapping + ri.token))[1] : p.slot_mapping_i32[ri.token];
  }
  const int n_keys = ri.n_keys, first_key = ri.first_key, ctx_len = ri.ctx_len;
  // PACK: this lane's columns' tokens, the keys they see / the fewest any column sees (tiles below it need no mask)
  int token[NCG], n_keys_col[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    token[cg] = PACK ? ri.token + (g_ok[cg] ? tq[cg] : 0) : ri.token;
    n_keys_col[cg] = PACK ? max(0, min(ctx_len + ri.q_pos + tq[cg] + 1, n_keys)) : n_keys;
  }
  const int n_keys_min = PACK ? max(0, min(ctx_len + ri.q_pos + 1, n_keys)) : n_keys;
  // sliding window: the first key this lane's column sees / the highest first key of any column (tiles from it on need no window mask)
  const int first_key_col = (PACK && FEAT && p.sliding_window > 0) ? max(0, ctx_len + ri.q_pos + tq[0] - p.sliding_window + 1) : first_key;
  const int first_key_max = (PACK && FEAT && p.sliding_window > 0) ? max(0, n_keys - p.sliding_window) : first_key;

  const int tile_lo = first_key / kTileKeys;
  const int tile_hi = (n_keys + kTileKeys - 1) / kTileKeys;
  // The split COUNT is host arithmetic (capture-stable); how many tiles a split walks is decided here, from THIS row's
  // own length, like the reference's 3D kernel (tiles_per_segment = cdiv(seq_len, NUM_SEGMENTS * TILE), :592): a graph
  // captured at max_model_len and replayed on short sequences still spreads every row over all its splits, and a
  // sequence longer than the host's max_seqlen_k is still read to its end.
  const int tps = split_tiles(tile_hi - tile_lo, a.num_splits);
  const int t0 = tile_lo + split * tps;
  const int t1 = min(t0 + tps, tile_hi);
  const bool direct = a.num_splits == 1;
  const int n_tiles = max(0, tile_hi - tile_lo);
  const int active = min(a.num_splits, (n_tiles + tps - 1) / tps);       // splits of this row that hold a tile
  const bool empty = t0 >= t1;
  // (tree: an empty split whose workgroup holds a partial stays for the workgroup's barriers and its share of the merge)
  if (empty && (!tree || (split & ~3) >= active)) {
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg)
    if (split == 0 && g_ok[cg]) {  // no visible key at all (no split has a tile): the reference returns acc/L = 0/1 = 0
      const int64_t o = (int64_t)token[cg] * p.out_stride_token + (int64_t)hq[cg] * p.out_stride_head;
#pragma unroll
      for (int b = 0; b < DBLK; ++b)
        if (!PAD || 16 * b + 4 * grp < a.d_valid) *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) = u32x2_t{0, 0};
      if (p.lse && grp == 0) p.lse[(int64_t)token[cg] * p.lse_stride_token + hq[cg]] = -INFINITY;
    }
    return;
  }

  const float k_scale = (FP8 && p.k_scale) ? p.k_scale[0] : 1.0f;
  const float v_scale = (FP8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  const float slope = (FEAT && p.alibi_slopes && g_ok[0]) ? p.alibi_slopes[hq[0]] : 0.0f;   // (FEAT kernels have one column group)
  const float scale_nat = p.scale * k_scale;     // fp8: K is used un-scaled, its scale moves here
  const float scale2 = scale_nat * kLog2e;
  const int32_t* bt = p.block_table + (int64_t)ri.seq * p.block_table_stride;
  using kv_elem_t = typename KVT::storage;
  const kv_elem_t* kbase = (const kv_elem_t*)p.k_cache + (int64_t)head * p.k_stride_head;
  const kv_elem_t* vbase = (const kv_elem_t*)p.v_cache + (int64_t)head * p.v_stride_head;
  const int last_group = (n_keys - 1) >> 4;  // last 16-key group that holds a visible key
  const int page_mask = p.page_size - 1;

  // per-lane constants of the row-shaped loads: piece idx = lane + 64*i -> (row = idx / PPR, piece = idx % PPR)
  // PAD: a piece in the padding columns loads the row's piece 0 instead (same instruction stream, always a
  // valid address) and is zeroed on its way into LDS
  int ld_row[NLD], ld_piece[NLD];
  bool ld_pad[NLD];
  uint32_t k_toff[NLD], v_toff[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = lane + 64 * i;
    if constexpr (V0) {
      // K: piece (d-chunk c, key row) at c*dx_stride + row*8; V: piece (d, 8 slots hf) at d*d_stride + 8*hf
      ld_row[i] = idx % 16;
      ld_piece[i] = idx / 16;
      ld_pad[i] = false;
      k_toff[i] = (uint32_t)(ld_piece[i] * (int)a.k_dx_stride + ld_row[i] * (int)a.k_slot_stride);
      v_toff[i] = (uint32_t)((idx / 2) * (int)a.v_d_stride + (idx % 2) * 8);
    } else {
      ld_row[i] = idx / PPR;
      ld_piece[i] = idx % PPR;
      ld_pad[i] = PAD && ld_piece[i] * EPP >= a.d_valid;
      const int src_piece = ld_pad[i] ? 0 : ld_piece[i];
      k_toff[i] = (uint32_t)(ld_row[i] * (int)a.k_slot_stride + src_piece * EPP);
      v_toff[i] = (uint32_t)(ld_row[i] * (int)a.v_slot_stride + src_piece * EPP);
    }
  }

  // block-table entries of the two groups of a tile, via the scalar cache (see scalar_load4)
  int pg[2];
  auto lookup_pages = [&](int tile) {
    const int i0 = (min(tile * 2, last_group) << 4) >> a.page_shift;      // stay inside the sequence's pages
    const int i1 = (min(tile * 2 + 1, last_group) << 4) >> a.page_shift;
    int d0, d1;
    scalar_load4(bt, i0, i1, i0, i1, pg[0], pg[1], d0, d1);
  };
  u32x4_t kreg[PF][2][NLD], vreg[PF][2][NLD];
  auto issue_loads = [&](int tile, u32x4_t (&KR)[2][NLD], u32x4_t (&VR)[2][NLD]) {   // uses pg[] = this tile's pages
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int gi = min(tile * 2 + h, last_group);
      const int slot0 = (gi << 4) & page_mask;
      const kv_elem_t* kp = kbase + ((uint64_t)(uint32_t)pg[h] * a.k_page_stride + (uint32_t)slot0 * a.k_slot_stride);
      const kv_elem_t* vp = vbase + ((uint64_t)(uint32_t)pg[h] * a.v_page_stride + (uint32_t)slot0 * a.v_slot_stride);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
#ifdef MI355_DECODE_PLAIN_LOADS
        KR[h][i] = *(const u32x4_t*)(kp + k_toff[i]);
        VR[h][i] = *(const u32x4_t*)(vp + v_toff[i]);
#else
        // every K/V byte is read once per call: streaming (nt) loads, which also keep the cache lines of the block
        // table and the partials from being pushed out (a plain 2 GiB read stream reaches 6.2 TB/s on this chip,
        // the same stream with nt loads 7.1: tools/probes/hbm_read.hip)
        KR[h][i] = __builtin_nontemporal_load((const u32x4_t*)(kp + k_toff[i]));
        VR[h][i] = __builtin_nontemporal_load((const u32x4_t*)(vp + v_toff[i]));
#endif
      }
    }
  };
  // one loaded 16-byte piece -> LDS row of the 16-bit type
  // (fp8: a piece widens to two 16-byte stores, and ds_write_b128 is serviced eight lanes at a time on 32 banks
  // (MI355X_MICROARCH.md, LDS): the eight pieces of a row, 32 bytes apart, put pieces p and p + 4 on the same banks in
  // both stores - the 2-way conflicts of profiles/r02/pmc_decode_fp8.txt. Taking the two halves of pieces 4 .. 7 in
  // swapped order - two 8-byte loads per piece, the first store then covers eight distinct 16-byte slots - removes them
  // and made the kernel 20 % SLOWER (C5: 183 -> 220 us, profiles/r03/decode_fp8_swap_ab.log): this kernel lives on its
  // VMEM issue slots, not on the LDS array.)
  auto park = [&](char* base, int row, int piece, u32x4_t v) {
    if constexpr (FP8) {
      u32x4_t lo, hi;
      widen_fp8x16<T, KVT>(v, lo, hi);
      *(u32x4_t*)(base + row * RS + piece * 32) = lo;
      *(u32x4_t*)(base + row * RS + piece * 32 + 16) = hi;
    } else {
      *(u32x4_t*)(base + row * RS + piece * 16) = v;
    }
  };

  float m_run[NCG], l_run[NCG];
  f32x4_t o_acc[NCG][DBLK];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
    m_run[cg] = -INFINITY;
    l_run[cg] = 0.0f;
#pragma unroll
    for (int b = 0; b < DBLK; ++b) o_acc[cg][b] = f32x4_t{0, 0, 0, 0};
  }

  if (fast_head && t0 == split * a.tiles_per_split) {
    // looked up with the sequence length, at the tile the host's split size predicted (the usual eager call: lengths
    // near max_seqlen_k); a second group past the sequence re-reads the first
    pg[0] = pg_first[0];
    pg[1] = (t0 * 2 + 1 > last_group) ? pg_first[0] : pg_first[1];
  } else if (!empty) {
    lookup_pages(t0);
  }
#pragma unroll
  for (int u = 0; u < PF; ++u)
    if (t0 + u < t1) {
      issue_loads(t0 + u, kreg[u], vreg[u]);
      if (t0 + u + 1 < t1) lookup_pages(t0 + u + 1);
    }
#pragma unroll
  for (int c = 0; c < KSTEPS; ++c)
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) asm volatile("" : "+v"(qf[cg][c]));   // retire the Q loads before the loop (see prefill kernel)

  auto tile_body = [&](int tile, u32x4_t (&KR)[2][NLD], u32x4_t (&VR)[2][NLD]) __attribute__((always_inline)) {
    // ---- park the current tile's rows in LDS, then refill the registers with a later tile --------
    u32x4_t kcur[2][NLD];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NLD; ++i) kcur[h][i] = KR[h][i];
    const bool tail = (tile * kTileKeys + kTileKeys > n_keys);
    if constexpr (!V0 && !FEAT && !PACK) {
      // Fused cache write (write_new_kv): the wave whose split ends at the sequence's last tile owns the key of the
      // token being decoded. Its row of K and V comes from k_new / v_new instead of from the cache - quantised the way
      // reshape_and_cache_flash would have stored it - is written to its page, and takes part in this tile like any
      // cached row. pg[] still holds this tile's pages: a wave's last lookup is for its last tile.
      if (p.write_new_kv && tile == tile_hi - 1 && t1 == tile_hi) {
        const bool store_row = qg == 0 && slot_sign >= 0;   // the waves of the other query-head groups only take the row; a padding row (slot < 0) is attended over but never stored
        const int r_last = n_keys - 1 - tile * kTileKeys;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int i = 0; i < NLD; ++i)
            if (h * 16 + ld_row[i] == r_last && !(PAD && ld_pad[i])) {
              const uint16_t* kn = (const uint16_t*)p.k_new + (int64_t)token[0] * p.new_stride_token + (int64_t)head * p.new_stride_head + ld_piece[i] * EPP;
              const uint16_t* vn = (const uint16_t*)p.v_new + (int64_t)token[0] * p.new_stride_token + (int64_t)head * p.new_stride_head + ld_piece[i] * EPP;
              u32x4_t kq, vq;
              if constexpr (FP8) {
                kq = quantise_fp8x16<T, KVT>(*(const u32x4_t*)kn, *(const u32x4_t*)(kn + 8), k_scale);
                vq = quantise_fp8x16<T, KVT>(*(const u32x4_t*)vn, *(const u32x4_t*)(vn + 8), v_scale);
              } else {
                kq = *(const u32x4_t*)kn;
                vq = *(const u32x4_t*)vn;
              }
              const int slot0 = ((tile * 2 + h) << 4) & page_mask;
              kv_elem_t* kp = (kv_elem_t*)kbase + ((uint64_t)(uint32_t)pg[h] * a.k_page_stride + (uint32_t)slot0 * a.k_slot_stride);
              kv_elem_t* vp = (kv_elem_t*)vbase + ((uint64_t)(uint32_t)pg[h] * a.v_page_stride + (uint32_t)slot0 * a.v_slot_stride);
              if (store_row) {
                *(u32x4_t*)(kp + k_toff[i]) = kq;
                *(u32x4_t*)(vp + v_toff[i]) = vq;
              }
              KR[h][i] = kq;
              VR[h][i] = vq;
              kcur[h][i] = kq;
            }
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        u32x4_t v = VR[h][i];
        if constexpr (V0) {
          // the piece is 8 consecutive keys of row d: zero the keys past the sequence (stale slots), park d-major
          const int idx = lane + 64 * i, d_row = idx / 2, hf = idx % 2;
          if (tail) {
            const int valid = n_keys - (tile * kTileKeys + h * 16 + hf * 8);     // keys of this piece inside the sequence
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const uint32_t keep = valid >= 2 * w + 2 ? 0xffffffffu : valid == 2 * w + 1 ? 0x0000ffffu : 0u;
              v[w] &= keep;
            }
          }
          *(u32x4_t*)(v_lds + d_row * RSV0 + (h * 16 + hf * 8) * 2) = v;
        } else {
          // rows past the sequence hold stale cache contents: keep NaN/Inf out of 0 * V
          if (tail && (tile * kTileKeys + h * 16 + ld_row[i] >= n_keys)) v = u32x4_t{0, 0, 0, 0};
          if (PAD && ld_pad[i]) v = u32x4_t{0, 0, 0, 0};
          park(v_lds, h * 16 + ld_row[i], ld_piece[i], v);
        }
      }
    if (tile + PF < t1) {
      issue_loads(tile + PF, KR, VR);
      if (tile + PF + 1 < t1) lookup_pages(tile + PF + 1);
    }

    // ---- S^T = K . Q^T, one 16-key group at a time through the K buffer --------------------------
    f32x4_t s[NCG][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) park(k_lds, ld_row[i], ld_piece[i], (PAD && ld_pad[i]) ? u32x4_t{0, 0, 0, 0} : kcur[h][i]);
      f32x4_t acc = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
      // (all of a group's fragment reads are issued before its first matrix instruction: written as one read per
      // k-step in front of its instruction, hipcc serialised read -> wait -> instruction in some builds of this kernel,
      // eight exposed LDS latencies per tile and ~3 % of C3)
      u32x4_t kf[KSTEPS];
#pragma unroll
      for (int c = 0; c < KSTEPS; ++c) kf[c] = *(const u32x4_t*)(k_lds + g * RS + c * 64 + grp * 16);  // lane = key row g of the group
#pragma unroll
      for (int c = 0; c < KSTEPS; ++c) {
        acc = mma<T>::run(__builtin_bit_cast(s16x8_t, kf[c]), qf[0][c], acc);
        if constexpr (NCG > 1) acc1 = mma<T>::run(__builtin_bit_cast(s16x8_t, kf[c]), qf[NCG - 1][c], acc1);
      }
      s[0][h] = acc;
      if constexpr (NCG > 1) s[NCG - 1][h] = acc1;
    }

    // ---- scores -> log2 domain, masks (reference order: scale, softcap, causal, window, +alibi) ---
    float alpha[NCG];
    s16x8_t pf[NCG];
    const bool plain = !FEAT || (!(p.softcap > 0.0f) && !p.alibi_slopes);
    const bool need_mask = (PACK ? tile * kTileKeys + kTileKeys > n_keys_min : tail) || (FEAT && tile * kTileKeys < first_key_max);
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
    float sv[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[h * 4 + r] = s[cg][h][r];
    if (plain && !need_mask) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[j] *= scale2;
    } else if (!FEAT) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = tile * kTileKeys + (j >> 2) * 16 + grp * 4 + (j & 3);
        sv[j] = key < n_keys_col[cg] ? sv[j] * scale2 : -INFINITY;
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = tile * kTileKeys + (j >> 2) * 16 + grp * 4 + (j & 3);
        float x = sv[j] * scale_nat;
        if (p.softcap > 0.0f) x = softcap_fast(x, p.softcap, 2.0f * kLog2e / p.softcap);
        if (key >= n_keys_col[cg] || key < first_key_col) x = -INFINITY;
        if (p.alibi_slopes) x += slope * (float)(key - ctx_len);
        sv[j] = x * kLog2e;
      }
    }
    float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    mx = max_over_lane_groups(mx);
    float m_new = fmaxf(m_run[cg], mx);
    if (!(m_new > -INFINITY)) m_new = 0.0f;   // fully masked so far (:486-489)
    alpha[cg] = __builtin_amdgcn_exp2f(m_run[cg] - m_new);
    float pv[8], psum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pv[j] = __builtin_amdgcn_exp2f(sv[j] - m_new);
      psum += pv[j];
    }
    l_run[cg] = l_run[cg] * alpha[cg] + psum;
    m_run[cg] = m_new;
    // P^T fragment (B operand): k-slot j -> key 4*grp + (j&3) of group (j>>2); rounded to the KV
    // dtype before P.V like the reference (:508)
    pf[cg] = __builtin_bit_cast(s16x8_t, u32x4_t{mma<T>::pack2(pv[0], pv[1]), mma<T>::pack2(pv[2], pv[3]),
                                                 mma<T>::pack2(pv[4], pv[5]), mma<T>::pack2(pv[6], pv[7])});
    }

    // ---- O^T += V^T . P^T -----------------------------------------------------------------------
    bool all_one = alpha[0] == 1.0f;
#pragma unroll
    for (int cg = 1; cg < NCG; ++cg) all_one = all_one && alpha[cg] == 1.0f;
    const bool rescale = !__all(all_one);
#pragma unroll
    for (int b = 0; b < DBLK; ++b) {
      // transposed read: lane 4q+pp of group grp addresses row 4*grp+q, columns 16b+4pp..+3 and
      // receives V[4*grp + e][16b + (lane&15)] in element e
      s16x4_t v0, v1;
      if constexpr (V0) {
        // d-major tile: lane (g, grp) is row d = 16b + g and takes keys 4grp .. 4grp+3 of each 16-key group as they lie
        const char* va = v_lds + (16 * b + g) * RSV0 + (4 * grp) * 2;
        v0 = *(const s16x4_t*)va;
        v1 = *(const s16x4_t*)(va + 32);
      } else {
        const int q4 = g >> 2, pp = g & 3;
        const char* va = v_lds + (grp * 4 + q4) * RS + (16 * b + 4 * pp) * 2;
        v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va));
        v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va + 16 * RS));
      }
      const s16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int cg = 0; cg < NCG; ++cg) {
        if (rescale) {
#pragma unroll
          for (int r = 0; r < 4; ++r) o_acc[cg][b][r] *= alpha[cg];
        }
        o_acc[cg][b] = mma<T>::run(vf, pf[cg], o_acc[cg][b]);
      }
    }
  };

  // (unrolled by construction, not by "#pragma unroll": when the tile body grew past the unroller's size limit - the
  // fused cache write did that - the register sets kreg[u] / vreg[u] were indexed at run time and the fp8 loop, the
  // only one with PF = 2, ran at half its rate: 16 x 32768 keys 370 us against 197)
  if constexpr (PF == 1) {
    for (int tile = t0; tile < t1; ++tile) tile_body(tile, kreg[0], vreg[0]);
  } else
  for (int tile = t0; tile < t1; tile += PF) {
    static_for<PF>([&](auto U) __attribute__((always_inline)) {
      constexpr int u = decltype(U)::value;
      if (tile + u < t1) tile_body(tile + u, kreg[u], vreg[u]);
    });
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  float l_tot[NCG];
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) l_tot[cg] = sum_over_lane_groups(l_run[cg]);
  if constexpr (PACK == 0) {
  if (tree) {
    // ---- two-level merge (reference: reduce_segments, :757-836) -----------------------------------------------------
    // Level 1, inside the workgroup: every wave parks its partial in its own (now idle) LDS region, lane by lane - the
    // four waves share one lane layout (column g, rows 16b + 4grp ..) - and wave w folds the 16-column blocks b = w, w + 4, ..
    // of all of them. One partial per workgroup then leaves for the workspace (or, with four splits in all, the output).
    constexpr int NB = DBLK / 4;
    constexpr int SLOT = D + kSlotPad;
    const int SH = a.num_splits >> 2, split_hi = split >> 2;
    const int nw = min(4, active - 4 * split_hi);                    // waves of this workgroup that hold a partial
    float* const mine = (float*)(smem + wave_in_wg * LDS_PER_WAVE);
    if (!empty) {
#pragma unroll
      for (int b = 0; b < DBLK; ++b) *(f32x4_t*)(mine + (b * 64 + lane) * 4) = o_acc[0][b];
      if (grp == 0) *(f32x2_t*)(mine + DBLK * 256 + 2 * g) = f32x2_t{l_tot[0] > 0.0f ? m_run[0] : -INFINITY, l_tot[0]};
    }
    __syncthreads();
    float m_all = -INFINITY, l_all = 0.0f, wgt[4];
    f32x2_t ml[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ml[i] = i < nw ? *(const f32x2_t*)((const float*)(smem + i * LDS_PER_WAVE) + DBLK * 256 + 2 * g) : f32x2_t{-INFINITY, 0.0f};
      m_all = fmaxf(m_all, ml[i][0]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      wgt[i] = ml[i][0] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ml[i][0] - m_all);
      l_all += ml[i][1] * wgt[i];
    }
    f32x4_t om[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      om[j] = f32x4_t{0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < nw) om[j] += *(const f32x4_t*)((const float*)(smem + i * LDS_PER_WAVE) + ((wave_in_wg + 4 * j) * 64 + lane) * 4) * wgt[i];
    }
    auto store_out = [&](int tok, int hqx, float m_fin, float l_fin, const f32x4_t (&ov)[NB]) {
      if (p.lse && grp == 0 && wave_in_wg == 0)
        p.lse[(int64_t)tok * p.lse_stride_token + hqx] = l_fin > 0.0f ? (m_fin + __builtin_amdgcn_logf(l_fin)) * kLn2 : -INFINITY;
      const float inv = l_fin > 0.0f ? v_scale / l_fin : 0.0f;              // "0 if the overall sum is 0" (:828)
      const int64_t o = (int64_t)tok * p.out_stride_token + (int64_t)hqx * p.out_stride_head;
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int b = wave_in_wg + 4 * j;
        if (!PAD || 16 * b + 4 * grp < a.d_valid) *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) =
            u32x2_t{mma<T>::pack2(ov[j][0] * inv, ov[j][1] * inv), mma<T>::pack2(ov[j][2] * inv, ov[j][3] * inv)};
      }
    };
    if (active <= 4) {                   // this row's tiles fit one workgroup (the others left at once): its fold is the result
      if (g_ok[0]) store_out(token[0], hq[0], m_all, l_all, om);
      return;
    }
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ws_slots, 0, (int)a.ws_slot_bytes_total, 0x00020000);
    const uint32_t slot_g0 = (uint32_t)(((uint32_t)unit * p.num_q_heads + hq0) * SH);     // slot of (g = 0, split group 0)
    if (g_ok[0]) {
      const uint32_t so = ((slot_g0 + (uint32_t)(g * SH + split_hi)) * SLOT) * 4u;
#pragma unroll
      for (int j = 0; j < NB; ++j)
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, om[j]), rsrc, so + (16 * (wave_in_wg + 4 * j) + 4 * grp) * 4, 0, 16);
      if (grp == 0 && wave_in_wg == 0)
        __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__builtin_bit_cast(uint32_t, m_all), __builtin_bit_cast(uint32_t, l_all)}, rsrc, so + D * 4, 0, 16);
    }
    // Level 2: the last workgroup of this (unit, KV head, query-head group) to arrive folds the workgroups' partials, again
    // wave w the blocks b = w, w + 4, ..: every wave drains its stores, the workgroup meets, ONE ticket is drawn.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                 // (also: every wave is done reading the parked partials)
    int* cnt = a.ws_cnt + ((unit * Hk + head) * a.qgroups + qg);
    int* const tk = (int*)smem;
    if (wave_in_wg == 0 && lane == 0) *tk = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(*tk);
    const int active_hi = (active + 3) >> 2;
    if (ticket != active_hi - 1) return;
    // every load below is an sc1 load (bypasses this CU's L1, which may hold lines of an earlier launch); lanes g >= G,
    // idle in the compute layout, take further partials: lane (gm, sub, grp) walks partials sub, sub + NS, .. of head gm
    // with ALL of its loads in flight at once (one round trip up to NS * U2 partials = 128 splits at G <= 4)
    const int Gp = G <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(G - 1)));
    const int NS = 16 / Gp, gm = g & (Gp - 1), sub = g / Gp;
    const bool gm_ok = gm < G;
    float m_acc = -INFINITY, l_acc = 0.0f;
    f32x4_t acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = f32x4_t{0, 0, 0, 0};
    auto fold = [&](float m_in, float l_in, const f32x4_t (&v_in)[NB]) {
      const float m_new = fmaxf(m_acc, m_in);
      const float wa = m_acc == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_acc - m_new);
      const float wb = m_in == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_in - m_new);
      l_acc = l_acc * wa + l_in * wb;
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = acc[j] * wa + v_in[j] * wb;
      m_acc = m_new;
    };
    if (gm_ok) {
      constexpr int U2 = D >= 256 ? 4 : 8;
      const uint32_t s0 = ((slot_g0 + (uint32_t)(gm * SH)) * SLOT) * 4u;
      for (int base = sub; base < active_hi; base += NS * U2) {
        float m_in[U2], l_in[U2];
        f32x4_t v_in[U2][NB];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
          const int sidx = base + u * NS;
          const uint32_t so = sidx < active_hi ? s0 + (uint32_t)sidx * (SLOT * 4u) : 0x80000000u;
          m_in[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, so + D * 4, 0, 16));
          l_in[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, so + D * 4 + 4, 0, 16));
#pragma unroll
          for (int j = 0; j < NB; ++j)
            v_in[u][j] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (16 * (wave_in_wg + 4 * j) + 4 * grp) * 4, 0, 16));
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) fold(base + u * NS < active_hi ? m_in[u] : -INFINITY, l_in[u], v_in[u]);
      }
    }
    for (int off = Gp; off < 16; off <<= 1) {
      const float m_in = __shfl_xor(m_acc, off), l_in = __shfl_xor(l_acc, off);
      f32x4_t v_in[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) v_in[j][r] = __shfl_xor(acc[j][r], off);
      fold(m_in, l_in, v_in);
    }
    if (gm_ok && sub == 0) store_out(ri.token, hq0 + gm, m_acc, l_acc, acc);
    if (wave_in_wg == 0 && lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // zero for the next call
    return;
  }
  }
  if (direct) {
#pragma unroll
    for (int cg = 0; cg < NCG; ++cg) {
      if (!g_ok[cg]) continue;
      if (p.lse && grp == 0)   // m_run is the row max of the scaled scores in the log2 domain
        p.lse[(int64_t)token[cg] * p.lse_stride_token + hq[cg]] = l_tot[cg] > 0.0f ? (m_run[cg] + __builtin_amdgcn_logf(l_tot[cg])) * kLn2 : -INFINITY;
      const float inv = l_tot[cg] > 0.0f ? v_scale / l_tot[cg] : 0.0f;
      const int64_t o = (int64_t)token[cg] * p.out_stride_token + (int64_t)hq[cg] * p.out_stride_head;
#pragma unroll
      for (int b = 0; b < DBLK; ++b)
        if (!PAD || 16 * b + 4 * grp < a.d_valid) *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) =
            u32x2_t{mma<T>::pack2(o_acc[cg][b][0] * inv, o_acc[cg][b][1] * inv), mma<T>::pack2(o_acc[cg][b][2] * inv, o_acc[cg][b][3] * inv)};
    }
    return;
  }
  // Partial -> workspace with WRITE-THROUGH (sc1) 16-byte stores: they are visible to every CU once
  // this wave's vmcnt has drained, without a release fence (cdna_hip_programming.md, Guideline 16 R1).
  constexpr int SLOT = D + kSlotPad;
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(a.ws_slots, 0, (int)a.ws_slot_bytes_total, 0x00020000);
  // slot rows are indexed by work unit: the query token, or the sequence when only decode rows are served
  // (PACK: a row per (unit, token of the chunk), column by column)
  const uint32_t slot_g0 = (uint32_t)(((uint32_t)(PACK ? unit << a.pack_shift : unit) * p.num_q_heads + hq0) * a.num_splits);   // slot of (g = 0, split 0)
  auto slot_of_col = [&](int c) -> uint32_t {
    if constexpr (PACK) { const int t = c / G; return slot_g0 + (uint32_t)((t * p.num_q_heads + (c - t * G)) * a.num_splits); }
    return slot_g0 + (uint32_t)(c * a.num_splits);
  };
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg)
  if (g_ok[cg]) {
    const uint32_t so = ((slot_of_col(16 * cg + g) + split) * SLOT) * 4u;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, o_acc[cg][b]), rsrc, so + (16 * b + 4 * grp) * 4, 0, 16);
    if (grp == 0)
      // (PACK: a split whose tiles lie past a column's last key leaves that column an empty partial)
      __builtin_amdgcn_raw_buffer_store_b64(u32x2_t{__builtin_bit_cast(uint32_t, (PACK && !(l_tot[cg] > 0.0f)) ? -INFINITY : m_run[cg]), __builtin_bit_cast(uint32_t, l_tot[cg])}, rsrc, so + D * 4, 0, 16);
  }
  if (!a.fused_merge) return;

  // ---- in-kernel merge by the last-arriving split of this (unit, KV head) (reference: reduce_segments, :757-836)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // every storing wave drains before it signals
  int* cnt = a.ws_cnt + ((unit * Hk + head) * a.qgroups + qg);
  int ticket = 0;
  if (lane == 0) ticket = __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ticket = __builtin_amdgcn_readfirstlane(ticket);
  if (ticket != active - 1) return;
  // All `active` partials are in memory; every load of them below is an sc1 load (bypasses this CU's
  // L1, which may hold lines of an earlier launch). The compute layout leaves lanes g >= G idle, so
  // the merge folds them in: lane (gm, sub, grp) walks splits sub, sub + NS, ... of head gm with all
  // of a split's loads in flight at once, and the NS partial merges meet through xor shuffles.
  // (m, l are two 4-byte loads: hipcc 7.2 narrows a raw_buffer_load_b64 whose halves are used apart
  // to ONE dword and hands the same register out for both.)
  // (two column groups: one after the other, every lane on its own column - NS = 1)
#pragma unroll
  for (int cg = 0; cg < NCG; ++cg) {
  // columns of this group in use - by THIS unit's tokens: a unit with fewer tokens than the chunk holds (a one-token row
  // of a mixed step) spreads its splits over the idle lanes like the plain kernel does
  const int C = PACK ? min(16, G * min(a.pack_tokens, ri.q_len - ri.q_pos) - 16 * cg) : G;
  const int Gp = NCG > 1 ? 16 : C <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(C - 1)));   // ... rounded up to a power of two
  const int NS = 16 / Gp;
  const int gm = g & (Gp - 1), sub = g / Gp;
  const int cm = 16 * cg + gm;
  const int tqm = PACK ? cm / G : 0;                                          // column cm = (token tqm of the chunk, head cm % G)
  const bool gm_ok = PACK ? (gm < C && tqm < min(a.pack_tokens, ri.q_len - ri.q_pos)) : gm < G;
  const int tokm = ri.token + tqm, hqm = hq0 + cm - tqm * G;
  float m_acc = -INFINITY, l_acc = 0.0f;
  f32x4_t acc[DBLK];
#pragma unroll
  for (int b = 0; b < DBLK; ++b) acc[b] = f32x4_t{0, 0, 0, 0};
  auto fold = [&](float m_in, float l_in, const f32x4_t (&v_in)[DBLK]) {
    const float m_new = fmaxf(m_acc, m_in);
    const float wa = m_acc == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_acc - m_new);
    const float wb = m_in == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_in - m_new);
    l_acc = l_acc * wa + l_in * wb;
#pragma unroll
    for (int b = 0; b < DBLK; ++b) acc[b] = acc[b] * wa + v_in[b] * wb;
    m_acc = m_new;
  };
  if (gm_ok) {
    // U splits' loads are issued before the first is consumed: each is a ~1-2 us L2-miss round trip,
    // so the serial chain is ceil(active / (NS*U)) trips. A split past `active` is given an offset
    // beyond the descriptor's range (the load returns 0) and the weight of an empty partial.
    constexpr int U = D >= 128 ? 2 : 4;
    const uint32_t s0 = (slot_of_col(cm) * SLOT) * 4u;
    for (int base = sub; base < active; base += NS * U) {
      float m_in[U], l_in[U];
      f32x4_t v_in[U][DBLK];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int sidx = base + u * NS;
        const uint32_t so = sidx < active ? s0 + (uint32_t)sidx * (SLOT * 4u) : 0x80000000u;
        m_in[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, so + D * 4, 0, 16));
        l_in[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, so + D * 4 + 4, 0, 16));
#pragma unroll
        for (int b = 0; b < DBLK; ++b)
          v_in[u][b] = __builtin_bit_cast(f32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsrc, so + (16 * b + 4 * grp) * 4, 0, 16));
      }
#pragma unroll
      for (int u = 0; u < U; ++u) fold(base + u * NS < active ? m_in[u] : -INFINITY, l_in[u], v_in[u]);
    }
  }
  for (int off = Gp; off < 16; off <<= 1) {
    const float m_in = __shfl_xor(m_acc, off), l_in = __shfl_xor(l_acc, off);
    f32x4_t v_in[DBLK];
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) v_in[b][r] = __shfl_xor(acc[b][r], off);
    fold(m_in, l_in, v_in);
  }
  if (gm_ok && sub == 0) {
    if (p.lse && grp == 0)
      p.lse[(int64_t)tokm * p.lse_stride_token + hqm] = l_acc > 0.0f ? (m_acc + __builtin_amdgcn_logf(l_acc)) * kLn2 : -INFINITY;
    const float inv = l_acc > 0.0f ? v_scale / l_acc : 0.0f;              // "0 if the overall sum is 0" (:828)
    const int64_t o = (int64_t)tokm * p.out_stride_token + (int64_t)hqm * p.out_stride_head;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
      if (!PAD || 16 * b + 4 * grp < a.d_valid) *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) =
          u32x2_t{mma<T>::pack2(acc[b][0] * inv, acc[b][1] * inv), mma<T>::pack2(acc[b][2] * inv, acc[b][3] * inv)};
  }
  }
  if (lane == 0) __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // leave the counter zero for the next call
}

// Merge of the split partials in a launch of its own (reference: reduce_segments, :757-836), used
// when a (unit, KV head) has more splits than the in-kernel merge reads in one round trip. One
// 256-thread workgroup per (unit, query head): thread (r, col) owns the 16-byte column chunk `col`
// of split rows r, r + R, ...; all of a thread's loads are issued before the first is consumed, so
// the kernel is ONE memory round trip however many splits there are, then an LDS fold of R rows.
// MAXS = the split count the instantiation covers (32 / 64 / 128): a launch with few splits does not pay for the loads
// of many.
template <typename T, bool FP8, int D, int MAXS>
__global__ __launch_bounds__(256) void reduce_splits_kernel(const DecodeArgs a) {
  constexpr int LPS = D / 4;               // lanes per split row (16, 32 or 64)
  constexpr int R = 256 / LPS;             // split rows per pass
  constexpr int NI = MAXS / R;             // passes
  static_assert(MAXS % R == 0 && MAXS <= kMaxSplits, "split rows come in whole passes");
  constexpr int SLOT = D + kSlotPad;
  __shared__ __attribute__((aligned(16))) float red[R][D + 4];   // folded columns, then (m, l)

  const mi355_attn_params& p = a.p;
  const int hq = blockIdx.y, tid = threadIdx.x;
  RowInfo ri;
  if (a.pack_tokens) {   // rows are (unit, token of the unit's chunk); the partials' splits follow the chunk's LAST token
    const int unit = blockIdx.x >> a.pack_shift, tq = blockIdx.x & (a.pack_tokens - 1);
    ri.seq = a.pack_cps == 1 ? unit : unit / a.pack_cps;
    if (ri.seq >= p.num_seqs) return;
    const int q_start = p.cu_seqlens_q[ri.seq], q0 = (unit - ri.seq * a.pack_cps) << a.pack_shift;
    ri.q_len = p.cu_seqlens_q[ri.seq + 1] - q_start;
    ri.q_pos = q0 + tq;
    if (ri.q_pos >= ri.q_len || (p.only_decodes && ri.q_len > p.only_decodes)) return;
    const int seq_len = p.seqused_k[ri.seq];
    ri.ctx_len = seq_len - ri.q_len;
    ri.token = q_start + ri.q_pos;
    ri.first_key = p.sliding_window > 0 ? max(0, ri.ctx_len + q0 - p.sliding_window + 1) : 0;   // the chunk's first token's
    ri.n_keys = max(0, min(ri.ctx_len + min(q0 + a.pack_tokens, ri.q_len), seq_len));
    ri.valid = true;
  } else {
    ri = row_info(p, a.by_seq, blockIdx.x);
    if (!ri.valid) return;
  }
  const int tile_lo = ri.first_key / kTileKeys;
  const int tile_hi = (ri.n_keys + kTileKeys - 1) / kTileKeys;
  const int n_tiles = max(0, tile_hi - tile_lo);
  const int tps = split_tiles(n_tiles, a.num_splits);
  const int active = min(a.num_splits, (n_tiles + tps - 1) / tps);
  const float v_scale = (FP8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  const int r = tid / LPS, col = tid % LPS;

  const float* slot0 = a.ws_slots + ((int64_t)blockIdx.x * p.num_q_heads + hq) * a.num_splits * SLOT;   // row = work unit
  float m_in[NI], l_in[NI];
  f32x4_t v_in[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int s = r + i * R;
    const float* slot = slot0 + (s < active ? s : 0) * SLOT;   // a row past `active` re-reads row 0 with weight 0
    m_in[i] = slot[D];
    l_in[i] = slot[D + 1];
    v_in[i] = ((const f32x4_t*)slot)[col];
  }
  float m_loc = -INFINITY;
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    if (r + i * R >= active) m_in[i] = -INFINITY;
    m_loc = fmaxf(m_loc, m_in[i]);
  }
  float l_loc = 0.0f;
  f32x4_t acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const float w = m_in[i] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_in[i] - m_loc);
    l_loc += l_in[i] * w;
    acc += v_in[i] * w;
  }
  *(f32x4_t*)&red[r][4 * col] = acc;
  if (col == 0) { red[r][D] = m_loc; red[r][D + 1] = l_loc; }
  __syncthreads();
  if (r != 0) return;
  float m_all = -INFINITY;
#pragma unroll
  for (int rr = 0; rr < R; ++rr) m_all = fmaxf(m_all, red[rr][D]);
  float l_all = 0.0f;
  acc = f32x4_t{0, 0, 0, 0};
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const float m_r = red[rr][D];
    const float w = m_r == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(m_r - m_all);
    l_all += red[rr][D + 1] * w;
    acc += *(const f32x4_t*)&red[rr][4 * col] * w;
  }
  const float inv = l_all > 0.0f ? v_scale / l_all : 0.0f;  // "0 if the overall sum is 0" (:828)
  if (p.lse && col == 0) p.lse[(int64_t)ri.token * p.lse_stride_token + hq] = l_all > 0.0f ? (m_all + __builtin_amdgcn_logf(l_all)) * kLn2 : -INFINITY;
  uint16_t* op = (uint16_t*)p.out + (int64_t)ri.token * p.out_stride_token + (int64_t)hq * p.out_stride_head;
  if (4 * col < a.d_valid) *(u32x2_t*)(op + 4 * col) = u32x2_t{mma<T>::pack2(acc[0] * inv, acc[1] * inv), mma<T>::pack2(acc[2] * inv, acc[3] * inv)};
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }
static bool is_fp8_dtype(int d) { return d == MI355_FP8_E4M3 || d == MI355_FP8_E5M2; }

// flash layout [page][slot][Hk][D] (what vLLM V1 uses)
static bool layout_is_flash(const mi355_attn_params& p) { return p.k_x == p.head_size && p.k_stride_d == 1 && p.v_stride_d == 1; }
// legacy v0 layout K [page][Hk][D/8][slot][8], V [page][Hk][D][slot], 16-bit elements, as the reference's legacy ops
// lay it out (LIB/kernels/legacy/triton_paged_decode_attention_2d.py:103-104): slots of a d-chunk / of a d contiguous
static bool layout_is_v0(const mi355_attn_params& p) {
  return p.k_x == 8 && p.k_stride_d == 1 && p.k_stride_slot == 8 && p.k_stride_dx == (int64_t)p.page_size * 8 &&
         p.v_stride_slot == 1 && p.v_stride_d == p.page_size && p.kv_dtype == p.q_dtype &&
         (p.head_size == 64 || p.head_size == 128 || p.head_size == 256);
}

#if DECODE_TU == 0
bool decode_supported(const mi355_attn_params& p) {
  if (!(p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16)) return false;
  if (p.kv_dtype != p.q_dtype && !is_fp8_dtype(p.kv_dtype)) return false;
  if (padded_head_size(p.head_size, is_fp8_dtype(p.kv_dtype)) == 0) return false;
  if ((p.k_new || p.v_new) && !p.write_new_kv) return false;
  if (p.page_size < 16 || (p.page_size & (p.page_size - 1)) != 0) return false;        // power of two, >= 16
  const bool v0 = !layout_is_flash(p) && layout_is_v0(p);
  // (write_new_kv: a decode step, or - only_decodes = 1 - the one-token rows of a mixed step whose prefill launch stores its own)
  if (p.write_new_kv && (v0 || ((p.max_seqlen_q != 1 || p.num_tokens != p.num_seqs) && p.only_decodes != 1) || !p.k_new || !p.v_new || p.new_stride_token % 8 != 0 ||
                         p.new_stride_head % 8 != 0 || ((uintptr_t)p.k_new & 15) != 0 || ((uintptr_t)p.v_new & 15) != 0))
    return false;
  if (!layout_is_flash(p) && !v0) return false;
  if (!aligned16(p.q) || !aligned16(p.k_cache) || !aligned16(p.v_cache)) return false;
  if (((uintptr_t)p.out & 7) != 0) return false;
  const int64_t kv_align = is_fp8_dtype(p.kv_dtype) ? 16 : 8;   // elements per 16 bytes
  const int64_t kv_strides[] = {p.k_stride_page, v0 ? kv_align : p.k_stride_slot, p.k_stride_head, p.v_stride_page, v0 ? kv_align : p.v_stride_slot, p.v_stride_head};
  for (int64_t s : kv_strides) if (s % kv_align != 0 || s < 0) return false;
  if (v0 && (p.k_stride_dx >= (1 << 24) || p.v_stride_d >= (1 << 24))) return false;
  if (p.q_stride_token % 8 != 0 || p.q_stride_head % 8 != 0) return false;
  if (p.out_stride_token % 4 != 0 || p.out_stride_head % 4 != 0) return false;
  if (p.k_stride_slot >= (1 << 24) || p.v_stride_slot >= (1 << 24)) return false;      // 32-bit in-page offsets
  if (p.k_stride_page >= (1LL << 31) || p.v_stride_page >= (1LL << 31)) return false;
  return true;
}
#endif

#if DECODE_TU == 0
bool decode_write_fusable(const mi355_attn_params& p) {
  mi355_attn_params q = p;
  q.write_new_kv = 1;
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;   // (the feature kernels take the general row lookup)
  return !feat && p.kernel_select != MI355_SELECT_GENERIC && p.kernel_select != MI355_SELECT_2D && decode_supported(q);
}
#endif

struct SplitPlan { int num_splits, tiles_per_split; };

// A wave holds 16 query heads of its KV head (the MFMA columns); a KV head with more takes cdiv(G, 16) waves, each
// streaming the head's K/V (neighbouring waves: the second reader hits in L2).
static int query_head_groups(const mi355_attn_params& p) { return (p.num_q_heads / p.num_kv_heads + 15) / 16; }

// Multi-token decode steps on the PACK kernels: plain attention (no window / soft-cap / ALiBi), flash layout, a built
// head size. One column group (16 matrix columns) per wave holds 16 / G tokens (G <= 8), two hold 32 / G (G <= 16,
// head sizes up to 128), both rounded down to a power of two; one group is taken when the longest query fits it.
// MI355_DECODE_PACK=0 switches packing off, =1 keeps it to one column group (A/B).
static int pow2_floor_shift(int x) { return x < 1 ? -1 : 31 - __builtin_clz((unsigned)x); }
// only_decodes = N > 1 (the decode launch of a mixed batch): the rows of sequences with up to N query tokens.
static int pack_max_q(const mi355_attn_params& p) { return p.only_decodes > 1 ? p.only_decodes : p.max_seqlen_q; }
#if DECODE_TU == 0
int decode_pack_groups(const mi355_attn_params& p) {
  if (pack_max_q(p) <= 1 || p.num_tokens <= p.num_seqs || p.only_decodes == 1 || p.skip_decodes || p.write_new_kv) return 0;
  // the decode kernels mask causally (n_keys of a column = ctx + q_pos + 1): only one-token rows are the same under both
  // masks, so a non-causal call never packs several tokens of a sequence
  if (p.non_causal) return 0;
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;   // (one column group only)
  if (!layout_is_flash(p) || p.head_size != padded_head_size(p.head_size, is_fp8_dtype(p.kv_dtype))) return 0;
  const int G = p.num_q_heads / p.num_kv_heads;
  static const char* const e = lab_env("MI355_DECODE_PACK");     // read once per process, like the other switches
  if (e && e[0] == '0') return 0;
  const bool one_ok = G <= 8, two_ok = G <= 16 && p.head_size <= 128 && !feat && !(e && e[0] == '1');
  if (one_ok && (pack_max_q(p) <= (1 << pow2_floor_shift(16 / G)) || !two_ok)) return 1;
  return two_ok ? 2 : 0;
}
#endif
// log2 of the query tokens one work unit holds (0 = not packed)
#if DECODE_TU == 0
int decode_pack_shift(const mi355_attn_params& p) {
  const int groups = decode_pack_groups(p);
  return groups ? pow2_floor_shift(16 * groups / (p.num_q_heads / p.num_kv_heads)) : 0;
}
#endif

static int pack_chunks_per_seq(const mi355_attn_params& p, int ps) { return (pack_max_q(p) + (1 << ps) - 1) >> ps; }
// A mixed batch's rows with up to this many query tokens go to the decode launch (1: only one-token rows). The host
// cannot know whether multi-token decode rows (speculative decoding) ride along with the prefills, and it need not:
// one column group holds 16 / G tokens at the cost of a one-token row.
#if DECODE_TU == 0
int decode_rows_max_q(const mi355_attn_params& p) {
  const int G = p.num_q_heads / p.num_kv_heads;
  if (p.non_causal) return 1;       // (see decode_pack_groups)
  mi355_attn_params q = p;
  q.skip_decodes = 0;
  if (p.decode_rows_hint > 1) {      // the caller knows its decode rows' length: whatever the packed kernels hold of it
    q.only_decodes = p.decode_rows_hint;
    if (decode_pack_groups(q) > 0 && p.decode_rows_hint <= (1 << decode_pack_shift(q))) return p.decode_rows_hint;
  }
  if (G > 8) return 1;
  q.only_decodes = 1 << pow2_floor_shift(16 / G);
  return (q.only_decodes > 1 && decode_pack_groups(q) == 1) ? q.only_decodes : 1;
}
#endif
// work units: sequences (only_decodes), query tokens, or - packed - chunks of query tokens
static long decode_units(const mi355_attn_params& p) {
  const int ps = decode_pack_shift(p);
  if (ps) return (long)p.num_seqs * pack_chunks_per_seq(p, ps);
  return p.only_decodes == 1 ? p.num_seqs : p.num_tokens;
}
// rows of split partials: one per work unit and query head; packed units keep one per token of their chunk
static long partial_rows(const mi355_attn_params& p) { return decode_units(p) << decode_pack_shift(p); }

// Capture-stable split policy: depends only on host-known sizes (units, Hk, max_seqlen_k).
static SplitPlan plan_splits(const mi355_attn_params& p) {
  const int max_tiles = (p.max_seqlen_k + kTileKeys - 1) / kTileKeys;
  if (max_tiles <= 1) return {1, 1};
  int want;
  if (p.num_segments > 0) {
    want = p.num_segments;
  } else {
    // (the decode rows of a mixed batch: the grid runs over all sequences and the prefill ones leave at once, so the split
    // count is sized for the rows that can be decode rows at most - every prefill sequence carries at most max_seqlen_q
    // of the num_tokens - num_seqs tokens beyond one per sequence. C4: 64 sequences, 32 of them decode rows; sized for
    // all 64 the launch ran 2 waves per CU at 3.7 TB/s, profiles/r02/bench_kernel_stats_mixed.csv)
    long units = decode_units(p);
    // (packed chunks: sequences shorter than max_seqlen_q leave units empty; this many are filled at least)
    if (const int ps = decode_pack_shift(p)) units = std::min(units, std::max((long)p.num_seqs, ((long)p.num_tokens + (1 << ps) - 1) >> ps));
    if (p.only_decodes && p.max_seqlen_q > 1 && p.num_tokens > p.num_seqs) {
      const long prefills = ((long)p.num_tokens - p.num_seqs + p.max_seqlen_q - 2) / (p.max_seqlen_q - 1);
      units = std::max(1L, std::min(units - 1, units - prefills));
    }
    const long base = std::max(1L, units * p.num_kv_heads * query_head_groups(p));  // waves with one split each
    // Work items in flight. With streaming loads a 16-bit cache runs best with 4 per CU: more only add partials and
    // a tail (8192 keys: batch 64 -> 2 splits 322 us, 4 splits 332, 8 splits 344; batch 16 -> 8 splits 90, 16: 95;
    // batch 4 at 32768 keys: 92 vs 94). The fp8 kernel spends its time widening, not waiting: it wants 8 per CU
    // (batch 64: 204 us vs 261 with half of them; 16 x 32768 keys, Hq 64: 211 vs 474).
    // MI355_DECODE_TARGET_WAVES overrides the number for sweeps (tools/bench_decode.py).
    static const long target_env = [] { const char* e = lab_env("MI355_DECODE_TARGET_WAVES"); return e ? atol(e) : 0L; }();
    const bool fp8_kv = p.kv_dtype == MI355_FP8_E4M3 || p.kv_dtype == MI355_FP8_E5M2;
    const long target = target_env > 0 ? target_env : 256L * (fp8_kv ? 8 : 4);
    want = (int)((target + base - 1) / base);
    // No floor on the tiles per split and no preference for the in-kernel merge's split count: one wave walks its
    // tiles one memory round trip (~1.3 us) at a time, so below the target the extra items win even when they cost
    // the second launch (graph replay, 8192 keys: batch 8 -> 8 splits merged in the kernel 54 us, 16 splits + reduce
    // launch 45; batch 1 at 512 / 2048 keys: 12.4 -> 7.5 us / 12.0 -> 7.8 us with one tile per split).
  }
  want = std::max(1, std::min(want, std::min(max_tiles, kMaxSplits)));
  const int tps = (max_tiles + want - 1) / want;
  return {(max_tiles + tps - 1) / tps, tps};
}

// Arrival counters of the in-kernel merge: a FIXED region at the head of the workspace, one int per
// (query token | sequence, KV head). Fixed so that the counters of one call never land on bytes an
// earlier call used for partials: the caller zero-fills the workspace once and every call leaves
// the counters at zero. A batch with more (unit, KV head) pairs than fit merges in a second launch.
constexpr size_t kCounterRegionBytes = 256 << 10;
static size_t counters_bytes(const mi355_attn_params&) { return kCounterRegionBytes; }
static bool counters_fit(const mi355_attn_params& p) {
  return (size_t)decode_units(p) * p.num_kv_heads * query_head_groups(p) * sizeof(int) <= kWsCountersBytes;   // (the region's tail belongs to the prefill fix-up flags)
}

// How a launch merges its splits (host arithmetic on host-known sizes, shared by the workspace query and the launch):
//   flat  - the last-arriving split reads every partial of its (unit, KV head) in ONE round trip (<= one_trip splits);
//   tree  - more splits than that on the plain (not packed) kernels: split count rounded up to a multiple of four, the four
//           waves of a workgroup fold their partials through LDS, slot rows of num_splits / 4, the last workgroup folds those;
//   launch - reduce_splits_kernel: packed multi-token steps with many splits, more (unit, KV head) pairs than counters,
//           MI355_DECODE_MERGE_KERNEL=1 (A/B).
struct MergePlan { int num_splits, tiles_per_split, slot_rows; bool flat, tree; };
static MergePlan plan_merge(const mi355_attn_params& p) {
  const SplitPlan sp = plan_splits(p);
  MergePlan m = {sp.num_splits, sp.tiles_per_split, sp.num_splits, false, false};
  if (sp.num_splits <= 1) return m;
  static const bool two_launch = lab_env("MI355_DECODE_MERGE_KERNEL") != nullptr;   // A/B switch: separate merge launch
  if (two_launch || !counters_fit(p)) return m;
  const int pack = decode_pack_groups(p), ps = decode_pack_shift(p);
  const int D = padded_head_size(p.head_size, is_fp8_dtype(p.kv_dtype));
  // in-kernel flat merge only where the last arriver reads its partials in ONE round trip (NS lanes x U
  // loads in flight, see the kernel's epilogue)
  const int G = std::min(p.num_q_heads / p.num_kv_heads, 16) << ps, Gp = G <= 1 ? 1 : 1 << (32 - __builtin_clz((unsigned)(G - 1)));   // columns in use
  // (one column group of packed tokens: two trips at worst - a unit with every column in use - against a merge launch
  // over all token slots; units with fewer tokens walk their splits on more lanes, see the kernel)
  const int one_trip = (pack == 2 ? 1 : pack == 1 ? 2 : 16 / Gp) * (D >= 128 ? 2 : 4);
  if (sp.num_splits <= one_trip) { m.flat = true; return m; }
  // Where the two-level merge pays (round 4, one box): a decode step of a few sequences planned for a long context - what a
  // graph captured at max_model_len is - merges 64 / 128 slot rows per (token, head) in its second launch; one launch with
  // the fold inside is 0.4-0.7 us per layer faster there (tools/e2e_proxy.py: 10.7 -> 10.05 us per token and layer at 600
  // keys, 15.4 -> 15.05 at 13 300; batch 1 at 32768 keys 28.6 -> 27.7). With few splits the merge launch is the cheaper
  // one (batch 1 at 512 keys, 16 splits: 7.4 us against 8.4), and a batch that fills the chip loses its streaming order to
  // the four-splits-per-workgroup deal (C5, 16 splits: 173 -> 205 us). MI355_DECODE_TREE=0 | 1 forces either (A/B).
  static const char* const tree_env = lab_env("MI355_DECODE_TREE");
  const long base = std::max(1L, decode_units(p) * p.num_kv_heads * query_head_groups(p));
  const bool want_tree = tree_env ? tree_env[0] == '1' : (sp.num_splits > 32 && base <= 32);
  if (pack == 0 && want_tree) {
    const int max_tiles = (p.max_seqlen_k + kTileKeys - 1) / kTileKeys;
    m.num_splits = std::min((sp.num_splits + 3) & ~3, kMaxSplits);
    m.tiles_per_split = (max_tiles + m.num_splits - 1) / m.num_splits;
    m.slot_rows = m.num_splits / 4;
    m.tree = true;
  }
  return m;
}

#if DECODE_TU == 0
size_t decode_workspace_bytes(const mi355_attn_params& p) {
  if (!decode_supported(p)) return 0;
  const MergePlan mp = plan_merge(p);
  if (mp.num_splits == 1) return 0;
  const size_t slots = (size_t)partial_rows(p) * p.num_q_heads * mp.slot_rows;
  return counters_bytes(p) + slots * (padded_head_size(p.head_size, is_fp8_dtype(p.kv_dtype)) + kSlotPad) * sizeof(float);
}
#endif

template <typename T, typename KVT, int D, bool FEAT, bool PAD, bool V0, int PACK = 0>
static int launch_decode_t(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  constexpr int WAVES = 4;
  constexpr bool FP8 = !__is_same(T, KVT);
  DecodeArgs a;
  a.p = p;
  const MergePlan sp = plan_merge(p);
  a.num_splits = sp.num_splits;
  a.tiles_per_split = sp.tiles_per_split;
  a.tree = (PACK == 0 && sp.tree) ? 1 : 0;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.qgroups = query_head_groups(p);
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.by_seq = p.only_decodes == 1 ? 1 : 0;     // (only_decodes = N > 1 on a kernel that does not pack: token by token)
  a.d_valid = p.head_size;
  a.unit_is_seq = (!p.only_decodes && p.max_seqlen_q == 1 && p.num_tokens == p.num_seqs) ? 1 : 0;
  a.pack_shift = PACK ? decode_pack_shift(p) : 0;
  a.pack_tokens = PACK ? 1 << a.pack_shift : 0;
  a.pack_cps = PACK ? pack_chunks_per_seq(p, a.pack_shift) : 0;
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  a.k_dx_stride = (uint32_t)p.k_stride_dx; a.v_d_stride = (uint32_t)p.v_stride_d;
  a.ws_slots = nullptr;
  a.ws_cnt = nullptr;
  a.ws_slot_bytes_total = 0;
  a.fused_merge = 0;
  if (sp.num_splits > 1) {
    const size_t slots = (size_t)partial_rows(p) * p.num_q_heads * sp.slot_rows;
    const size_t slot_bytes = slots * (D + kSlotPad) * sizeof(float);
    const size_t need = counters_bytes(p) + slot_bytes;
    if (!ws || ws_bytes < need) {
      set_error("decode needs a %zu-byte workspace, got %zu", need, ws_bytes);
      return MI355_ERR_WORKSPACE;
    }
    if (slot_bytes >= (1ull << 31)) {
      set_error("split-KV scratch of %zu bytes exceeds the 2 GiB buffer-descriptor range; pass fewer segments", slot_bytes);
      return MI355_ERR_UNSUPPORTED;
    }
    a.ws_cnt = (int*)ws;
    a.ws_slots = (float*)((char*)ws + counters_bytes(p));
    a.ws_slot_bytes_total = (uint32_t)slot_bytes;
    a.fused_merge = sp.flat ? 1 : 0;
  }
  const long units = decode_units(p);
  if (units == 0) return MI355_OK;
  const long items = units * sp.num_splits * p.num_kv_heads * a.qgroups;
  const int grid = (int)((items + WAVES - 1) / WAVES);      // (tree: num_splits is a multiple of four - one workgroup per group of four)
  if (PACK && (a.group << a.pack_shift) > 16 * PACK) { set_error("decode: packed columns exceed the wave's"); return MI355_ERR_UNSUPPORTED; }
  const size_t lds = (size_t)WAVES * (16 * (D * 2 + 32) + (V0 ? D * 80 : 32 * (D * 2 + 32)));
  hipLaunchKernelGGL((decode_splitkv_kernel<T, KVT, D, WAVES, FEAT, PAD, V0, PACK>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
  int rc = check_hip(hipGetLastError(), "decode_splitkv_kernel launch");
  if (rc != MI355_OK) return rc;
  if (sp.num_splits > 1 && !a.fused_merge && !a.tree) {
    const dim3 rgrid((unsigned)partial_rows(p), p.num_q_heads);
    if (sp.num_splits <= 32) hipLaunchKernelGGL((reduce_splits_kernel<T, FP8, D, 32>), rgrid, dim3(256), 0, stream, a);
    else if (sp.num_splits <= 64) hipLaunchKernelGGL((reduce_splits_kernel<T, FP8, D, 64>), rgrid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((reduce_splits_kernel<T, FP8, D, 128>), rgrid, dim3(256), 0, stream, a);
    rc = check_hip(hipGetLastError(), "reduce_splits_kernel launch");
  }
  if (rc == MI355_OK)
    set_kernel_name(V0 ? (sp.num_splits > 1 ? "decode_splitkv_v0" : "decode_single_v0")
                       : PACK == 1 ? (sp.num_splits > 1 ? (FP8 ? "decode_splitkv_pack_fp8" : "decode_splitkv_pack") : (FP8 ? "decode_single_pack_fp8" : "decode_single_pack"))
                       : PACK == 2 ? (sp.num_splits > 1 ? (FP8 ? "decode_splitkv_pack2_fp8" : "decode_splitkv_pack2") : (FP8 ? "decode_single_pack2_fp8" : "decode_single_pack2"))
                       : sp.num_splits > 1 ? (FP8 ? "decode_splitkv_fp8" : "decode_splitkv") : (FP8 ? "decode_single_fp8" : "decode_single"));
  return rc;
}

// The PACK instantiations are built in a translation unit of their own (decode_splitkv_pack.hip: this source with
// DECODE_TU = 1), so that the library's objects keep compiling side by side in about a minute each.
template <typename T, typename KVT, int D>
int launch_decode_pack(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream, bool feat, int groups);

#if DECODE_TU == 1
template <typename T, typename KVT, int D>
int launch_decode_pack(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream, bool feat, int groups) {
  if (groups == 1)
    return feat ? launch_decode_t<T, KVT, D, true, false, false, 1>(p, ws, ws_bytes, stream) : launch_decode_t<T, KVT, D, false, false, false, 1>(p, ws, ws_bytes, stream);
  if constexpr (D <= 128)
    if (groups == 2 && !feat) return launch_decode_t<T, KVT, D, false, false, false, 2>(p, ws, ws_bytes, stream);
  set_error("decode: no packed kernel for %d column groups at head size %d", groups, D);
  return MI355_ERR_UNSUPPORTED;
}
#define MI355_PACK_INST(T, KVT) \
  template int launch_decode_pack<T, KVT, 64>(const mi355_attn_params&, void*, size_t, hipStream_t, bool, int); \
  template int launch_decode_pack<T, KVT, 128>(const mi355_attn_params&, void*, size_t, hipStream_t, bool, int); \
  template int launch_decode_pack<T, KVT, 256>(const mi355_attn_params&, void*, size_t, hipStream_t, bool, int);
MI355_PACK_INST(bf16_t, bf16_t) MI355_PACK_INST(bf16_t, e4m3_t) MI355_PACK_INST(bf16_t, e5m2_t)
MI355_PACK_INST(f16_t, f16_t) MI355_PACK_INST(f16_t, e4m3_t) MI355_PACK_INST(f16_t, e5m2_t)
#undef MI355_PACK_INST
#endif

#if DECODE_TU == 0
template <typename T, typename KVT, int D>
static int launch_decode_f(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  if constexpr (__is_same(T, KVT)) {
    if (!layout_is_flash(p))   // decode_supported admitted it: the legacy v0 layout
      return feat ? launch_decode_t<T, KVT, D, true, false, true>(p, ws, ws_bytes, stream) : launch_decode_t<T, KVT, D, false, false, true>(p, ws, ws_bytes, stream);
  }
  if (const int groups = decode_pack_groups(p)) return launch_decode_pack<T, KVT, D>(p, ws, ws_bytes, stream, feat, groups);
  if (p.head_size != D)
    return feat ? launch_decode_t<T, KVT, D, true, true, false>(p, ws, ws_bytes, stream) : launch_decode_t<T, KVT, D, false, true, false>(p, ws, ws_bytes, stream);
  return feat ? launch_decode_t<T, KVT, D, true, false, false>(p, ws, ws_bytes, stream) : launch_decode_t<T, KVT, D, false, false, false>(p, ws, ws_bytes, stream);
}

template <typename T, typename KVT>
static int launch_decode_d(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  switch (padded_head_size(p.head_size, is_fp8_dtype(p.kv_dtype))) {
    case 64: return launch_decode_f<T, KVT, 64>(p, ws, ws_bytes, stream);
    case 128: return launch_decode_f<T, KVT, 128>(p, ws, ws_bytes, stream);
    case 256: return launch_decode_f<T, KVT, 256>(p, ws, ws_bytes, stream);
  }
  set_error("decode: head_size %d not built", p.head_size);
  return MI355_ERR_UNSUPPORTED;
}

template <typename T>
static int launch_decode_kv(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (p.kv_dtype == MI355_FP8_E4M3) return launch_decode_d<T, e4m3_t>(p, ws, ws_bytes, stream);
  if (p.kv_dtype == MI355_FP8_E5M2) return launch_decode_d<T, e5m2_t>(p, ws, ws_bytes, stream);
  return launch_decode_d<T, T>(p, ws, ws_bytes, stream);
}

int launch_decode(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!decode_supported(p)) {
    set_error("decode kernel does not support this configuration");
    return MI355_ERR_UNSUPPORTED;
  }
  if (p.q_dtype == MI355_BF16) return launch_decode_kv<bf16_t>(p, ws, ws_bytes, stream);
  return launch_decode_kv<f16_t>(p, ws, ws_bytes, stream);
}
#endif

}  // namespace mi355
