// Short-prompt prefill for gfx950: the LATENCY form of kernel_unified_attention_2d
// (LIB/kernels/triton_unified_attention.py:275-523) - round 4.
//
// The reference's own headline protocol is a batch-1 prompt of 500 tokens (scripts/bench_vllm_latency_range.py:48-50,
// :98-103), where a launch is a few microseconds of work spread thin: at 1 x 512 tokens (Hq 32 / Hk 8 / D 128) the
// 128-row Q blocks of prefill_dma_kernel<4, 2> are 64 workgroups for 256 CUs, the heaviest walks its 8 key tiles one
// dependent Q.K^T -> softmax -> P.V chain of 1.4 us at a time (16.6 us per launch), and prefill_pw_kernel's 256-row
// blocks are 32 workgroups. This kernel deals the same work to every CU and halves the chain:
//   * Q block = 64 rows (64 / G tokens x G heads, rows ordered (token, head-in-group) like the reference, :343-346):
//     512 tokens of the 32 / 8 shape are 256 workgroups, one per CU; two fit a CU (64 KiB of LDS, < 128 VGPRs);
//   * the workgroup's four waves are 2 ROW halves x 2 KEY halves of every 64-key tile: a wave computes 32 rows (two
//     16-column groups of the matrix instruction) against 32 keys - 16 + 16 matrix instructions per tile instead of the
//     4-wave kernel's 16 + 16 of four times the size - and reads only its key half of the staged tile from LDS;
//   * both contractions on v_mfma_f32_16x16x32 in the decode kernel's orientation (S^T = K . Q^T, O^T += V^T . P^T): a lane
//     owns one query row per column group, the online softmax is lane-local (one exchange across the four lane groups
//     for the row maximum), the score registers are the P.V operand as they stand;
//   * K / V tiles HBM -> LDS by LDS-DMA, two 32 KiB stages, the 16-byte chunk index XOR-swizzled on the source side
//     (K: chunk ^= row & 15, V: chunk ^= 2 (row & 7): conflict-free for ds_read_b128 and ds_read_b64_tr_b16 as they
//     are used here), rows past the sequence redirected to its last row, one barrier per tile;
//   * the two key halves of a row half meet once, after the loop, through LDS: each wave hands the partner the column
//     group it does not finish and folds the partner's partial of the one it does, so all four waves normalise and store.
// The sequence of a Q block and its lengths come in one memory round trip (a ballot over the batch's words, common.h).
// Served: bf16 / f16, head size 128, 16-bit flash-layout cache, causal or not, no window / soft-cap / ALiBi; the host
// (launch_prefill, prefill_mfma.hip) picks it where the key range is short and the wider kernels' grids underfill the chip.
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 lbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 lf16x8_t;
typedef __attribute__((ext_vector_type(4))) short ls16x4_t;
typedef __attribute__((ext_vector_type(8))) short ls16x8_t;
typedef __attribute__((ext_vector_type(4))) float lf32x4_t;
typedef __attribute__((ext_vector_type(2))) float lf32x2_t;
typedef __attribute__((ext_vector_type(4))) unsigned int lu32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int lu32x2_t;

constexpr int kLatRows = 64;       // Q-block rows per workgroup
constexpr float kLatLog2e = 1.4426950408889634f;
constexpr float kLatLn2 = 0.6931471805599453f;
constexpr float kLatDefer = 8.0f;  // a row's reference moves when a tile's maximum exceeds it by 2^8

struct LatArgs {
  mi355_attn_params p;
  int group;       // G
  int block_q;     // tokens per Q block = 64 / G
  int page_shift;  // log2(page_size)
  int kv_same_strides;
  uint32_t k_page_stride, k_slot_stride, v_page_stride, v_slot_stride;   // elements; validated on the host
};

template <typename T> struct lmma;
template <> struct lmma<bf16_t> {
  static __device__ __forceinline__ lf32x4_t run(ls16x8_t a, ls16x8_t b, lf32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lbf16x8_t, a), __builtin_bit_cast(lbf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
};
template <> struct lmma<f16_t> {
  static __device__ __forceinline__ lf32x4_t run(ls16x8_t a, ls16x8_t b, lf32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(lf16x8_t, a), __builtin_bit_cast(lf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
};

// v_max3_f32 as it stands (fmaxf chains come out as canonicalising v_max pairs: 16 instructions more per tile, and this
// kernel's tile is what one wave per SIMD can issue)
__device__ __forceinline__ float lat_max3(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float lat_max4(float v) {      // over the four lanes {g, g+16, g+32, g+48}
  v = fmaxf(v, lane_xor32(v));
  return fmaxf(v, lane_xor16(v));
}
__device__ __forceinline__ float lat_sum4(float v) {
  v += lane_xor32(v);
  return v + lane_xor16(v);
}

// One LDS-DMA piece with a SCALAR base: lane l's 16 bytes from sbase + voff land at LDS byte address lds_dst + 16 l.
__device__ __forceinline__ void lat_glds16(uint32_t voff, uint64_t sbase, uint32_t lds_dst) {
  // (M0 is not saved: nothing else in this kernel uses it - no indirect register indexing, no GWS / message instructions)
  const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_dst);
  const uint32_t blo = __builtin_amdgcn_readfirstlane((uint32_t)sbase), bhi = __builtin_amdgcn_readfirstlane((uint32_t)(sbase >> 32));
  const uint64_t bu = ((uint64_t)bhi << 32) | blo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(bu), "s"(dst) : "memory");
}

// Diagnostic build only (-DMI355_PROFILE_WG, tools/wg_profile.py): life of a workgroup in s_memtime ticks.
#ifdef MI355_PROFILE_WG
#define LAT_WG_STAMP(var)                                                                  \
  unsigned long long var;                                                                  \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#define LAT_WG_REALTIME(var)                                                               \
  unsigned long long var;                                                                  \
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
#else
#define LAT_WG_STAMP(var) do { } while (0)
#define LAT_WG_REALTIME(var) do { } while (0)
#endif

// NKQ = key parts of a tile: the workgroup is 2 row halves x NKQ key parts = 2 NKQ waves, a tile 32 NKQ keys.
//   NKQ 2: four waves, 64-key tiles, 64 KiB of LDS - two workgroups per CU (many Q blocks: their prologues overlap);
//   NKQ 4: eight waves (two per SIMD), 128-key tiles, 128 KiB - one workgroup per CU with twice the waves on a Q block's
//          keys: the form for launches of at most one workgroup per CU, where the heaviest Q block IS the launch.
// WR: the instantiation that carries the fused cache write (write_new_kv) - apart, so that the plain one keeps its stream.
// KV8: an fp8 cache (1 = e4m3, 2 = e5m2; reference :434-455 dequantises on load), the scheme of prefill_pw_kernel's KV8
// instantiations: a wave's 16-key group of the next tile travels as fp8 - two 1 KiB pieces per matrix instead of four - into
// a 4 KiB staging area of the wave behind the stages; when it has landed (the wait the loop has anyway) the same wave widens
// it (v_cvt_scalef32_pk_*: exact) into the rows the 16-bit LDS-DMA would have filled, swizzle included, in front of the
// tile's barrier. k_scale rides in Q', v_scale in 1 / l.
template <typename T, int NKQ, bool WR = false, int KV8 = 0>
__global__ __launch_bounds__(128 * NKQ, NKQ == 2 ? 2 : 1) void prefill_lat_kernel(const LatArgs a) {
  static_assert(!(WR && KV8), "the fused cache write is built for 16-bit caches");
  constexpr int EB = KV8 ? 1 : 2;                   // bytes per cache element
  using KVT = std::conditional_t<KV8 == 2, e5m2_t, std::conditional_t<KV8 == 1, e4m3_t, T>>;
  constexpr int D = 128, ROWB = D * 2;              // 256-byte rows of 16 chunks of 16 bytes
  constexpr int NW = 2 * NKQ, TILE = 32 * NKQ;      // waves; keys per staged tile (one 16-key group per wave)
  constexpr int KBUF = TILE * ROWB, STAGE = 2 * KBUF;   // K tile then V tile
  constexpr int KSTEPS = D / 32, DBLK = D / 16;
  constexpr int NST = 2, IPT = 8, PD = NST - 1;     // stages; LDS-DMA instructions per tile and wave (4 K + 4 V); tiles in flight

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mi355_attn_params& p = a.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 1, kh = wave >> 1;          // row half, key part (0 .. NKQ-1)
  const int G = a.group, BQ = a.block_q;
  const int g = lane & 15, grp = lane >> 4;
  LAT_WG_STAMP(wg_t0);
  LAT_WG_REALTIME(wg_r0);

  const int head = (int)(blockIdx.x % p.num_kv_heads);
  const int qblock = (int)(gridDim.x / p.num_kv_heads - 1 - blockIdx.x / p.num_kv_heads);    // heaviest (latest) Q blocks first
  // the sequence and its three words in ONE round trip (a ballot over up to 63 sequences' words, common.h)
  int q_start, q_len, seq_len;
  const int seq = find_seq_and_lengths(p.cu_seqlens_q, p.seqused_k, p.num_seqs, qblock, BQ, lane, q_start, q_len, seq_len);
  if (seq < 0) return;
  // block-table entries 64 at a time in a VGPR (lane l = entry chunk * 64 + l), picked with v_readlane: requested as
  // soon as the sequence is known, bounded by max_seqlen_k's page count
  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;
  const int bt_last_any = ((max(p.max_seqlen_k, 1) + p.page_size - 1) >> a.page_shift) - 1;
  int bt_chunk = 0;
  int bt_cur = bt[min(lane, bt_last_any)];
  int bt_nxt = bt[min(64 + lane, bt_last_any)];
  const int qb_local = qblock - (q_start / BQ + seq);
  if (qb_local < 0 || qb_local * BQ >= q_len || q_len <= p.skip_decodes || (p.only_decodes && q_len > p.only_decodes)) return;
  const int ctx_len = p.non_causal ? seq_len : seq_len - q_len;     // non-causal: every row sees the whole sequence
  const int tok0 = qb_local * BQ;
  LAT_WG_STAMP(wg_ta);   // metadata known

  // ---- this lane's two query rows (column groups) --------------------------------------------------------------------
  int tok_local[2], hq[2], lim[2];
  bool row_ok[2];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) {
    const int m = rg * 32 + cg * 16 + g;            // row inside the Q block
    tok_local[cg] = tok0 + m / G;
    hq[cg] = head * G + m % G;
    row_ok[cg] = (m < BQ * G) && (tok_local[cg] < q_len);
    lim[cg] = row_ok[cg] ? min(p.non_causal ? seq_len - 1 : ctx_len + tok_local[cg], seq_len - 1) : -1;   // last visible key
  }
  const int wg_tok_hi = min(tok0 + BQ - 1, q_len - 1);
  const int n_keys_wg = max(0, min(p.non_causal ? seq_len : ctx_len + wg_tok_hi + 1, seq_len));
  const int w_tok_lo = tok0 + (rg * 32) / G;                                          // first token of this wave's rows
  const int tile_hi = (n_keys_wg + TILE - 1) / TILE;

  // ---- Q fragments (B operand of S^T = K . Q^T): lane (g, grp) holds Q[row][32c + 8grp .. +7], pre-scaled by scale * log2(e)
  // when they land (below): the matrix pipe then delivers scores in the log2 domain ---------------------------------------
  lu32x4_t qraw[2][KSTEPS];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) {
    // padding rows read the sequence's last query row (a valid address) and are masked through lim = -1
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)(q_start + min(tok_local[cg], q_len - 1)) * p.q_stride_token +
                         (int64_t)hq[cg] * p.q_stride_head + 8 * grp;
#pragma unroll
    for (int c = 0; c < KSTEPS; ++c) qraw[cg][c] = *(const lu32x4_t*)(qp + 32 * c);
  }

  // ---- LDS-DMA staging (the 4-wave form of prefill_dma_kernel): lane handles chunk (row = tid >> 4, c = tid & 15) of a
  // 16-row piece; a tile is four pieces of K and of V --------------------------------------------------------------------
  const char* kbase = (const char*)p.k_cache + (int64_t)head * p.k_stride_head * EB;
  const char* vbase = (const char*)p.v_cache + (int64_t)head * p.v_stride_head * EB;
  const int last_group = (max(n_keys_wg, 1) - 1) >> 4;
  const int page_mask = p.page_size - 1;
  // fused cache write (see issue_dma): the linear key / value rows of this KV head, and this Q block's own tokens stored
  // into their pages - by slot_mapping when the caller hands one in (a negative slot is a padding token: not stored,
  // triton_attn.py:149-151), else by position through the block table. 16 bytes per thread and matrix, sixteen rows a pass.
  constexpr bool fused = WR;
  const int new_st = (int)p.new_stride_token;
  const char* const knew = (const char*)p.k_new + (int64_t)head * p.new_stride_head * 2;
  const char* const vnew = (const char*)p.v_new + (int64_t)head * p.new_stride_head * 2;
  if constexpr (fused) {
    const int tok_end = min(tok0 + BQ, q_len);
    for (int tok = tok0 + (tid >> 4); tok < tok_end; tok += NW * 4) {
      int64_t slot;
      if (p.slot_mapping) slot = p.slot_mapping[q_start + tok];
      else if (p.slot_mapping_i32) slot = p.slot_mapping_i32[q_start + tok];
      else { const int pos = ctx_len + tok; slot = (int64_t)bt[pos >> a.page_shift] * p.page_size + (pos & page_mask); }
      if (slot < 0) continue;
      const int64_t pg = slot >> a.page_shift, sl = slot & page_mask;
      const int64_t src = (int64_t)(q_start + tok) * (new_st * 2) + (tid & 15) * 16;
      const lu32x4_t kk = *(const lu32x4_t*)(knew + src), vv = *(const lu32x4_t*)(vnew + src);
      *(lu32x4_t*)((char*)kbase + pg * ((int64_t)a.k_page_stride * 2) + sl * ((int64_t)a.k_slot_stride * 2) + (tid & 15) * 16) = kk;
      *(lu32x4_t*)((char*)vbase + pg * ((int64_t)a.v_page_stride * 2) + sl * ((int64_t)a.v_slot_stride * 2) + (tid & 15) * 16) = vv;
    }
  }
  // Wave w stages 16-key group w of every tile, both matrices: ONE block-table entry and one scalar base per tile and
  // wave, then four K and four V instructions of 1 KiB (rows 4j .. 4j+3 of the group), each with its own constant per-lane
  // byte offset (the source-side swizzle inside) and the base as an SGPR pair: ~50 scalar instructions per tile and wave
  // where per-piece addresses cost ~250 - and this kernel's tile is bound by what one wave per SIMD can ISSUE.
  const int rin = lane >> 4, ch = lane & 15;           // row within an instruction's four, 16-byte chunk
  uint32_t k_voff[4], v_voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int rig = 4 * j + rin;                        // row within the 16-key group
    k_voff[j] = (uint32_t)(rig * (int)a.k_slot_stride * 2 + ((ch ^ rig) << 4));
    v_voff[j] = (uint32_t)(rig * (int)a.v_slot_stride * 2 + ((ch ^ (2 * (rig & 7))) << 4));
  }
  const uint32_t k_page_bytes = a.k_page_stride * EB, v_page_bytes = a.v_page_stride * EB;
  const uint32_t smem_base = lds_addr(smem);
  // fp8 cache: a piece covers eight 128-byte key rows, unswizzled (the widening write applies the swizzle); lane = (row lane >> 3,
  // 16-byte piece lane & 7 = head dims 16 c8 .. + 15)
  const int r8 = lane >> 3, c8 = lane & 7;
  char* const stg = smem + NST * STAGE + wave * 4096;          // this wave's staging area: K group, then V group
  const uint32_t stg_lds = smem_base + NST * STAGE + (uint32_t)wave * 4096;
  auto issue_dma = [&](int tile, uint32_t stage_off) {
    const int e0 = (min(tile * NW, last_group) << 4) >> a.page_shift;
    if ((e0 >> 6) != bt_chunk) {            // wave-uniform; entries only ever move forward
      bt_chunk = e0 >> 6;
      bt_cur = bt_nxt;
      bt_nxt = bt[min((bt_chunk + 1) * 64 + lane, bt_last_any)];
    }
    const int gi = min(tile * NW + wave, last_group);
    const int key0 = gi << 4;
    const int slot0 = key0 & page_mask;
    const uint32_t page = (uint32_t)__builtin_amdgcn_readlane(bt_cur, (key0 >> a.page_shift) & 63);
    const uint64_t k_off = (uint64_t)page * k_page_bytes + (uint64_t)((uint32_t)slot0 * a.k_slot_stride) * EB;
    const uint64_t v_off = a.kv_same_strides ? k_off : (uint64_t)page * v_page_bytes + (uint64_t)((uint32_t)slot0 * a.v_slot_stride) * EB;
    const uint64_t kb = (uint64_t)kbase + k_off, vb = (uint64_t)vbase + v_off;
    const uint32_t dst = smem_base + stage_off + (uint32_t)wave * (16 * ROWB);
    if constexpr (KV8) {       // two pieces per matrix into the staging area; rows past the sequence fetch its last row
      const int maxr = max(seq_len - 1 - key0, 0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int r = min(8 * j + r8, maxr);
        lat_glds16((uint32_t)(r * (int)a.k_slot_stride + (c8 << 4)), kb, stg_lds + j * 1024);
        lat_glds16((uint32_t)(r * (int)a.v_slot_stride + (c8 << 4)), vb, stg_lds + 2048 + j * 1024);
      }
      return;
    }
    if (fused && key0 + 16 > ctx_len) {        // (fused: compile time)
      // FUSED CACHE WRITE (write_new_kv): keys at positions >= ctx_len are this call's own tokens and come from the linear
      // key / value tensors, never from the cache (whoever stores them - the Q block that owns the token, above - need not have
      // done so yet: no order between workgroups is needed). A group of sixteen such keys is sixteen consecutive rows of
      // [T, Hk, D]: the same shape as a cache page's slots, another scalar base and its own row stride. The one group of a
      // sequence that straddles ctx_len takes per-lane addresses.
      if (key0 >= ctx_len) {
        const uint64_t nb = (uint64_t)(uint32_t)(q_start + key0 - ctx_len) * (uint64_t)(new_st * 2);
        const uint64_t knb = (uint64_t)knew + nb, vnb = (uint64_t)vnew + nb;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rig = 4 * j + rin, r = min(rig, max(seq_len - 1 - key0, 0));        // (rows past the sequence: its last row)
          lat_glds16((uint32_t)(r * new_st * 2 + ((ch ^ rig) << 4)), knb, dst + j * 1024);
          lat_glds16((uint32_t)(r * new_st * 2 + ((ch ^ (2 * (rig & 7))) << 4)), vnb, dst + KBUF + j * 1024);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int rig = 4 * j + rin, pos = min(key0 + rig, seq_len - 1);
          const bool is_new = pos >= ctx_len;
          const char* ks = is_new ? knew + (int64_t)(q_start + pos - ctx_len) * (new_st * 2) : (const char*)kb + (int64_t)(pos - key0) * ((int)a.k_slot_stride * 2);
          const char* vs = is_new ? vnew + (int64_t)(q_start + pos - ctx_len) * (new_st * 2) : (const char*)vb + (int64_t)(pos - key0) * ((int)a.v_slot_stride * 2);
          glds16(ks + ((ch ^ rig) << 4), dst + j * 1024);
          glds16(vs + ((ch ^ (2 * (rig & 7))) << 4), dst + KBUF + j * 1024);
        }
      }
    } else
    if (key0 + 16 > seq_len) {              // wave-uniform: the sequence ends inside this group -> rows past it fetch its last row
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int rig = 4 * j + rin, r = min(rig, max(seq_len - 1 - key0, 0));
        lat_glds16((uint32_t)(r * (int)a.k_slot_stride * 2 + ((ch ^ rig) << 4)), kb, dst + j * 1024);
        lat_glds16((uint32_t)(r * (int)a.v_slot_stride * 2 + ((ch ^ (2 * (rig & 7))) << 4)), vb, dst + KBUF + j * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lat_glds16(k_voff[j], kb, dst + j * 1024);
        lat_glds16(v_voff[j], vb, dst + KBUF + j * 1024);
      }
    }
  };
  // fp8: this wave's staged group (landed: the caller waited) widened into its sixteen rows of stage `stage_off`
  auto widen_staged = [&](uint32_t stage_off) {
    if constexpr (KV8) {
      char* const kd = smem + stage_off + wave * (16 * ROWB);
      char* const vd = kd + KBUF;
#pragma unroll
      for (int m = 0; m < 2; ++m)            // K, V
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const lu32x4_t in = *(const lu32x4_t*)(stg + m * 2048 + j * 1024 + lane * 16);
          uint32_t o[8];
#pragma unroll
          for (int w = 0; w < 4; ++w) widen_fp8x4<T, KVT>(in[w], o[2 * w], o[2 * w + 1]);
          const int rig = 8 * j + r8, f = m ? 2 * (rig & 7) : rig;
          char* const row = (m ? vd : kd) + rig * ROWB;
          *(lu32x4_t*)(row + (((2 * c8) ^ f) << 4)) = lu32x4_t{o[0], o[1], o[2], o[3]};
          *(lu32x4_t*)(row + (((2 * c8 + 1) ^ f) << 4)) = lu32x4_t{o[4], o[5], o[6], o[7]};
        }
    }
  };
#pragma unroll
  for (int t = 0; t < PD; ++t)
    if (t < tile_hi) issue_dma(t, t * STAGE);
  // wait until this wave's pieces of the NEXT tile have landed: `newer` tiles behind it may stay in flight
  auto wait_next_tile = [&](int newer) {
    if (PD >= 3 && newer >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPT) : "memory");
    else if (PD >= 2 && newer == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(IPT) : "memory");
    else glds_wait_all();
  };

  // ---- per-lane LDS read offsets inside a stage (swizzles folded in) --------------------------------------------------
  // K fragment of 16-key group kg, k-step c: row kh*32 + kg*16 + g, logical chunk 4c + grp
  uint32_t k_rd[KSTEPS];
#pragma unroll
  for (int c = 0; c < KSTEPS; ++c) k_rd[c] = (uint32_t)((kh * 32 + g) * ROWB + (((4 * c + grp) ^ g) << 4));
  // V transposed read, output block b: rows kh*32 + 4grp + q4 (+16), byte columns 32b + 8pp: chunk 2b + (pp >> 1), + 8 (pp & 1)
  const int q4 = g >> 2, pp = g & 3;
  const int vrow = kh * 32 + 4 * grp + q4;
  const uint32_t v_row0 = (uint32_t)(KBUF + vrow * ROWB + 8 * (pp & 1)), v_row1 = v_row0 + 16 * ROWB;
  const int vsw = 2 * (vrow & 7);                    // (row + 16 has the same low bits)
  uint32_t v_ch[DBLK];
#pragma unroll
  for (int b = 0; b < DBLK; ++b) v_ch[b] = (uint32_t)(((2 * b + (pp >> 1)) ^ vsw) << 4);

  const float scale2 = p.scale * kLatLog2e * ((KV8 && p.k_scale) ? p.k_scale[0] : 1.0f);     // (fp8: K is widened unscaled)
  const float v_sc = (KV8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  // Softmax state per row (column group): a REFERENCE m_ref instead of a running maximum - P = 2^(s - m_ref) is the same
  // softmax for any reference as long as P stays in range - moved only when a row's tile maximum exceeds it by 2^kLatDefer
  // (the 4-wave kernel's scheme, prefill_mfma.hip): a calm tile costs no exchange across the lane groups and no rescaling
  // of O, and the reference enters through the matrix instruction's C operand, so the exponential takes the score as it comes.
  float m_ref[2] = {0.0f, 0.0f}, l_run[2] = {0.0f, 0.0f};
  bool started[2] = {false, false};
  lf32x4_t cin[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};      // -m_ref in every register
  lf32x4_t o_acc[2][DBLK];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg)
#pragma unroll
    for (int b = 0; b < DBLK; ++b) o_acc[cg][b] = lf32x4_t{0, 0, 0, 0};
  ls16x8_t qf[2][KSTEPS];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg)
#pragma unroll
    for (int c = 0; c < KSTEPS; ++c) {      // (the compiler's wait for the Q loads lands here: they are older than the DMA)
      lu32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const uint32_t v = qraw[cg][c][e];
        float lo, hi;
        if constexpr (__is_same(T, bf16_t)) { lo = bf16_to_f32((uint16_t)(v & 0xffff)); hi = bf16_to_f32((uint16_t)(v >> 16)); }
        else { lo = f16_to_f32((uint16_t)(v & 0xffff)); hi = f16_to_f32((uint16_t)(v >> 16)); }
        w[e] = lmma<T>::pack2(lo * scale2, hi * scale2);
      }
      qf[cg][c] = __builtin_bit_cast(ls16x8_t, w);
    }
#pragma unroll
  for (int cg = 0; cg < 2; ++cg)
#pragma unroll
    for (int c = 0; c < KSTEPS; ++c) asm volatile("" : "+v"(qf[cg][c]));
  LAT_WG_STAMP(wg_tb);   // Q landed, first tiles requested
  wait_next_tile(max(0, min(PD - 1, tile_hi - 1)));      // the first tile has landed
  if (tile_hi > 0) widen_staged(0);
  __syncthreads();
  LAT_WG_STAMP(wg_t1);

  auto compute_tile = [&](int tile, const char* stage) {
    const int key_base = tile * TILE + kh * 32;       // first key of this wave's part
    // ---- S^T - m_ref = K . Q'^T + cin: two 16-key groups x two column groups ---------------------------------------------
    lf32x4_t s[2][2];                                 // [cg][kg]
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      lu32x4_t kf[KSTEPS];
#pragma unroll
      for (int c = 0; c < KSTEPS; ++c) kf[c] = *(const lu32x4_t*)(stage + k_rd[c] + kg * 16 * ROWB);
      lf32x4_t acc0 = cin[0], acc1 = cin[1];
#pragma unroll
      for (int c = 0; c < KSTEPS; ++c) {
        acc0 = lmma<T>::run(__builtin_bit_cast(ls16x8_t, kf[c]), qf[0][c], acc0);
        acc1 = lmma<T>::run(__builtin_bit_cast(ls16x8_t, kf[c]), qf[1][c], acc1);
      }
      s[0][kg] = acc0;
      s[1][kg] = acc1;
    }
    // ---- softmax against the row's reference (log2 domain) -----------------------------------------------------------------
    const bool need_mask = p.non_causal ? (key_base + 32 > seq_len) : (key_base + 31 > ctx_len + w_tok_lo) || (key_base + 32 > seq_len);
    float sv[2][8], mx[2];
    bool calm = true;
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[cg][kg * 4 + r] = s[cg][kg][r];
      if (need_mask) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int key = key_base + (j >> 2) * 16 + grp * 4 + (j & 3);
          sv[cg][j] = key <= lim[cg] ? sv[cg][j] : -INFINITY;
        }
      }
      mx[cg] = lat_max3(lat_max3(sv[cg][0], sv[cg][1], sv[cg][2]), lat_max3(sv[cg][3], sv[cg][4], sv[cg][5]), fmaxf(sv[cg][6], sv[cg][7]));
      calm = calm && ((started[cg] && mx[cg] <= kLatDefer) || !row_ok[cg]);
    }
    if (!__all(calm)) {
      // some row's reference moves: its first visible key, or a score more than 2^kLatDefer above it. The decision is the
      // ROW's (four lanes hold its keys), so the tile maximum is exchanged across the lane groups first.
#pragma unroll
      for (int cg = 0; cg < 2; ++cg) {
        const float mxr = lat_max4(mx[cg]);
        const bool row_calm = started[cg] && mxr <= kLatDefer;
        const float upd = (!row_calm && mxr > -INFINITY) ? mxr : 0.0f;
        // (a row's FIRST reference may lie far below 0: 2^-upd would overflow; nothing has been accumulated yet, so its factor is 1)
        const float alpha = started[cg] ? __builtin_amdgcn_exp2f(-upd) : 1.0f;
        started[cg] = started[cg] || (mxr > -INFINITY);
        m_ref[cg] += upd;
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[cg][j] -= upd;
#pragma unroll
        for (int r = 0; r < 4; ++r) cin[cg][r] = -m_ref[cg];
        l_run[cg] *= alpha;
#pragma unroll
        for (int b = 0; b < DBLK; ++b)
#pragma unroll
          for (int r = 0; r < 4; ++r) o_acc[cg][b][r] *= alpha;
      }
    }
    ls16x8_t pf[2];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = __builtin_amdgcn_exp2f(sv[cg][j]);
      l_run[cg] += ((pv[0] + pv[1]) + (pv[2] + pv[3])) + ((pv[4] + pv[5]) + (pv[6] + pv[7]));
      // P^T fragment (B operand): k-slot j -> key 4 grp + (j & 3) of group (j >> 2); rounded to the cache type like the reference (:508)
      pf[cg] = __builtin_bit_cast(ls16x8_t, lu32x4_t{lmma<T>::pack2(pv[0], pv[1]), lmma<T>::pack2(pv[2], pv[3]),
                                                      lmma<T>::pack2(pv[4], pv[5]), lmma<T>::pack2(pv[6], pv[7])});
    }
    // ---- O^T += V^T . P^T ---------------------------------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < DBLK; ++b) {
      const ls16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ls16x4_t*)(stage + v_row0 + v_ch[b]));
      const ls16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ls16x4_t*)(stage + v_row1 + v_ch[b]));
      const ls16x8_t vf = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
      o_acc[0][b] = lmma<T>::run(vf, pf[0], o_acc[0][b]);
      o_acc[1][b] = lmma<T>::run(vf, pf[1], o_acc[1][b]);
    }
  };

  // ---- tile loop: two tiles per trip so that the stage is a compile-time offset ------------------------------------------
  // this wave's rows see keys up to wave_keys; its key half of a tile may lie past them (the diagonal tile's upper half)
  const int w_tok_hi = min(min(tok0 + (rg * 32 + 31) / G, tok0 + BQ - 1), q_len - 1);
  const int wave_keys = (w_tok_lo <= w_tok_hi) ? min(p.non_causal ? seq_len : ctx_len + w_tok_hi + 1, seq_len) : 0;
  for (int tile = 0; tile < tile_hi; tile += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int t = tile + u;
      if (t < tile_hi) {
        char* cur = smem + u * STAGE;
        // (tile t + PD goes to the stage tile t - 1 left, released by its barrier)
        if (t + PD < tile_hi) issue_dma(t + PD, ((u + PD) % NST) * STAGE);
        if (t * TILE + kh * 32 < wave_keys) compute_tile(t, cur);
        wait_next_tile(max(0, min(t + PD, tile_hi - 1) - (t + 1)));     // this wave's pieces of tile t+1 have landed ...
        if (t + 1 < tile_hi) widen_staged(((u + 1) % NST) * STAGE);     // (fp8: ... in its staging area; widened into tile t+1's stage, which tile t-1 left at the last barrier)
        __syncthreads();                                 // ... and so have everyone else's; stage `cur` is free
      }
    }
  }

  LAT_WG_STAMP(wg_t2);
  // ---- the NKQ key parts of a row half meet: every wave parks its partial (both column groups, lane-major: the waves of a
  // row half share one lane layout) in the idle stages, wave kh folds the 16-column blocks b = kh, kh + NKQ, .. of all of
  // them, normalises and stores ----------------------------------------------------------------------------------------
  float l_tot[2];
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) l_tot[cg] = lat_sum4(l_run[cg]);
  constexpr int XCH = 2 * DBLK * 1024 + 1024;        // per wave: O of two column groups + 64 x {m0, l0, m1, l1}
  constexpr int NB = DBLK / NKQ;
  float* const mine = (float*)(smem + wave * XCH);
#pragma unroll
  for (int cg = 0; cg < 2; ++cg)
#pragma unroll
    for (int b = 0; b < DBLK; ++b) *(lf32x4_t*)(mine + ((cg * DBLK + b) * 64 + lane) * 4) = o_acc[cg][b];
  *(lf32x4_t*)(mine + 2 * DBLK * 256 + 4 * lane) = lf32x4_t{l_tot[0] > 0.0f ? m_ref[0] : -INFINITY, l_tot[0], l_tot[1] > 0.0f ? m_ref[1] : -INFINITY, l_tot[1]};
  __syncthreads();
#pragma unroll
  for (int cg = 0; cg < 2; ++cg) {
    float m_all = -INFINITY, l_all = 0.0f, wgt[NKQ];
    lf32x4_t ml[NKQ];
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      ml[i] = *(const lf32x4_t*)((const float*)(smem + (rg + 2 * i) * XCH) + 2 * DBLK * 256 + 4 * lane);
      m_all = fmaxf(m_all, ml[i][2 * cg]);
    }
#pragma unroll
    for (int i = 0; i < NKQ; ++i) {
      wgt[i] = ml[i][2 * cg] == -INFINITY ? 0.0f : __builtin_amdgcn_exp2f(ml[i][2 * cg] - m_all);
      l_all += ml[i][2 * cg + 1] * wgt[i];
    }
    const bool ok = row_ok[cg];
    const int tok = q_start + tok_local[cg], hqx = hq[cg];
    if (p.lse && ok && grp == 0 && kh == 0)
      p.lse[(int64_t)tok * p.lse_stride_token + hqx] = l_all > 0.0f ? (m_all + __builtin_amdgcn_logf(l_all)) * kLatLn2 : -INFINITY;
    const float inv = l_all > 0.0f ? v_sc / l_all : 0.0f;        // a row that sees no key: 0 (:494)
    uint16_t* op = (uint16_t*)p.out + (int64_t)tok * p.out_stride_token + (int64_t)hqx * p.out_stride_head + 4 * grp;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int b = kh + NKQ * j;
      lf32x4_t o = {0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NKQ; ++i) o += *(const lf32x4_t*)((const float*)(smem + (rg + 2 * i) * XCH) + ((cg * DBLK + b) * 64 + lane) * 4) * (wgt[i] * inv);
      if (ok) *(lu32x2_t*)(op + 16 * b) = lu32x2_t{lmma<T>::pack2(o[0], o[1]), lmma<T>::pack2(o[2], o[3])};
    }
  }
#ifdef MI355_PROFILE_WG
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LAT_WG_STAMP(wg_t3);
    LAT_WG_REALTIME(wg_r3);
    unsigned long long* dbg = (unsigned long long*)(((unsigned long long)(unsigned)p.reserved1 << 32) | (unsigned)p.reserved0);
    if (dbg && tid == 0) {       // the record layout of prefill_dma_kernel's stamps (tools/wg_profile.py)
      atomicAdd(dbg + 8, wg_t1 - wg_t0);
      atomicAdd(dbg + 15, wg_ta - wg_t0);
      atomicAdd(dbg + 7, wg_tb - wg_ta);
      atomicAdd(dbg + 9, wg_t2 - wg_t1);
      atomicAdd(dbg + 10, wg_t3 - wg_t2);
      atomicAdd(dbg + 11, (unsigned long long)tile_hi);
      atomicAdd(dbg + 12, 1ull);
      atomicMin(dbg + 13, wg_t0);
      atomicMax(dbg + 14, wg_t3);
      const unsigned hw_id = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));
      const unsigned xcc_id = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));
      unsigned long long* rec = dbg + 16 + 4ull * blockIdx.x;
      rec[0] = wg_r0; rec[1] = wg_r3; rec[2] = ((unsigned long long)xcc_id << 32) | hw_id; rec[3] = (unsigned long long)tile_hi;
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Which calls launch_prefill gives this kernel (one box's measurements, see there): about one 64-row Q block per CU, or ONE
// sequence of 640 .. 2047 keys. MI355_PREFILL (lab switch) pins / excludes it.
bool prefill_lat_selected(const mi355_attn_params& p) {
  if (!prefill_lat_applicable(p) || p.max_seqlen_k >= 2048) return false;
  static const char* const variant = lab_env("MI355_PREFILL");
  if (variant) return variant[0] == 'l';
  const long wgs = ((long)p.num_tokens * (p.num_q_heads / p.num_kv_heads) / 64 + p.num_seqs) * p.num_kv_heads;   // its grid
  return wgs <= 288 || (p.num_seqs == 1 && p.max_seqlen_k >= 640 && wgs <= 1024);
}

// A step with a prefill whose cache write rides the attention launches (write_new_kv with max_seqlen_q > 1; SURVEY.md 8f-2, the
// pair of calls at LIB/backend/triton_attn.py:393-405 + :437 as ONE op): the prefill rows on this kernel or on one of the
// LDS-DMA kernels (prefill_mfma.hip) - they attend over the step's new keys straight from the linear tensors and the Q block
// that owns a token stores its rows - and, in a step that may carry one-token rows as well (several sequences of unequal
// query lengths: two launches, api.hip), those rows on the split-KV decode kernel's own fused write. Not the long-prefill
// kernel (prefill_pw_kernel: from 2048 keys on, 1536 / 1024 with several sequences), key-split launches, features, fp8 caches.
bool prefill_write_fusable(const mi355_attn_params& p) {
  mi355_attn_params q = p;
  q.write_new_kv = 1;
  if (!q.k_new || !q.v_new || q.max_seqlen_q <= 1 || q.non_causal || q.skip_decodes || q.only_decodes || q.new_kv_all_rows) return false;
  if (q.kernel_select != MI355_SELECT_AUTO) return false;
  if (q.softcap > 0.0f || q.alibi_slopes || q.sliding_window > 0) return false;
  if (q.kv_dtype != q.q_dtype || q.new_stride_token % 8 != 0 || q.new_stride_head % 8 != 0 || ((uintptr_t)q.k_new & 15) != 0 || ((uintptr_t)q.v_new & 15) != 0) return false;
  if (q.new_stride_token <= 0 || q.new_stride_token >= (1 << 22)) return false;
  if (!prefill_runs_without_key_splits(q) || !(prefill_lat_selected(q) || prefill_dma_selected(q))) return false;     // (a key-split launch writes partial outputs: no fused write)
  const bool single_launch = q.num_seqs == 1 || (int64_t)q.num_seqs * q.max_seqlen_q == q.num_tokens;
  if (single_launch) return true;
  mi355_attn_params d = q;            // the one-token rows' launch
  d.only_decodes = 1;
  return decode_supported(d);
}

bool prefill_lat_applicable(const mi355_attn_params& p) {
  if (!prefill_supported(p)) return false;
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  const int G = p.num_q_heads / p.num_kv_heads;
  const bool fp8 = p.kv_dtype == MI355_FP8_E4M3 || p.kv_dtype == MI355_FP8_E5M2;      // (round 4: its KV8 instantiations; no fused cache write there)
  return !feat && p.head_size == 128 && (p.kv_dtype == p.q_dtype || (fp8 && !p.write_new_kv)) && G <= kLatRows && ((uintptr_t)p.out & 7) == 0;
}

template <typename T, int NKQ, bool WR, int KV8 = 0>
static int launch_lat_t(const mi355_attn_params& p, hipStream_t stream) {
  LatArgs a;
  a.p = p;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.block_q = kLatRows / a.group;
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.kv_same_strides = (p.k_stride_page == p.v_stride_page && p.k_stride_slot == p.v_stride_slot && p.k_stride_head == p.v_stride_head) ? 1 : 0;
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  const int qblocks = p.num_tokens / a.block_q + p.num_seqs;    // static upper bound (:886-889,:935-943)
  // two stages of a K and a V tile; the waves' partials meet in the same bytes afterwards (a little more than the stages)
  constexpr size_t stages = (size_t)2 * 2 * (32 * NKQ) * 256, xch = (size_t)2 * NKQ * (2 * 8 * 1024 + 1024);
  constexpr size_t staging = KV8 ? (size_t)2 * NKQ * 4096 : 0;       // fp8: 4 KiB per wave behind the stages
  constexpr size_t lds = (stages + staging > xch ? stages + staging : xch);
  static_assert(lds <= 160 * 1024, "prefill_lat_kernel: LDS");
  static std::atomic<uint64_t> lds_opt_in{0};
  const int rc0 = ensure_dynamic_lds((const void*)prefill_lat_kernel<T, NKQ, WR, KV8>, (int)lds, lds_opt_in, "hipFuncSetAttribute(prefill_lat)");
  if (rc0 != MI355_OK) return rc0;
  hipLaunchKernelGGL((prefill_lat_kernel<T, NKQ, WR, KV8>), dim3(qblocks * p.num_kv_heads), dim3(128 * NKQ), lds, stream, a);
  const int rc = check_hip(hipGetLastError(), "prefill_lat_kernel launch");
  if (rc == MI355_OK) set_kernel_name(KV8 ? "prefill_mfma_lat_fp8" : "prefill_mfma_lat");
  return rc;
}

int launch_prefill_lat(const mi355_attn_params& p, hipStream_t stream) {
  if (!prefill_lat_applicable(p)) { set_error("prefill_lat_kernel does not serve this configuration"); return MI355_ERR_UNSUPPORTED; }
  // eight waves per Q block while the launch is at most ~one workgroup per CU (the heaviest Q block is the launch), four
  // (two workgroups per CU) beyond. MI355_LAT_WAVES=4 | 8 pins one (measurements).
  static const int pin = [] { const char* e = lab_env("MI355_LAT_WAVES"); return e ? atoi(e) : 0; }();
  const long wgs = ((long)p.num_tokens / (kLatRows / (p.num_q_heads / p.num_kv_heads)) + p.num_seqs) * p.num_kv_heads;
  const bool eight = pin ? pin == 8 : wgs <= 288;
  const bool bf = p.q_dtype == MI355_BF16;
  if (p.kv_dtype == MI355_FP8_E4M3) {
    if (eight) return bf ? launch_lat_t<bf16_t, 4, false, 1>(p, stream) : launch_lat_t<f16_t, 4, false, 1>(p, stream);
    return bf ? launch_lat_t<bf16_t, 2, false, 1>(p, stream) : launch_lat_t<f16_t, 2, false, 1>(p, stream);
  }
  if (p.kv_dtype == MI355_FP8_E5M2) {
    if (eight) return bf ? launch_lat_t<bf16_t, 4, false, 2>(p, stream) : launch_lat_t<f16_t, 4, false, 2>(p, stream);
    return bf ? launch_lat_t<bf16_t, 2, false, 2>(p, stream) : launch_lat_t<f16_t, 2, false, 2>(p, stream);
  }
  if (p.write_new_kv) {
    if (eight) return bf ? launch_lat_t<bf16_t, 4, true>(p, stream) : launch_lat_t<f16_t, 4, true>(p, stream);
    return bf ? launch_lat_t<bf16_t, 2, true>(p, stream) : launch_lat_t<f16_t, 2, true>(p, stream);
  }
  if (eight) return bf ? launch_lat_t<bf16_t, 4, false>(p, stream) : launch_lat_t<f16_t, 4, false>(p, stream);
  return bf ? launch_lat_t<bf16_t, 2, false>(p, stream) : launch_lat_t<f16_t, 2, false>(p, stream);
}

}  // namespace mi355
