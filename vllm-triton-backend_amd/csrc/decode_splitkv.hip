// Paged-KV decode for gfx950: split-KV partials + merge.
//
// Replaces kernel_unified_attention_3d + reduce_segments
// (LIB/kernels/triton_unified_attention.py:526-754, :757-836) and, through the same math, the legacy
// paged_attention_2d/3d kernels. HBM-bound: every K/V byte is read exactly once and only
// G = Hq/Hk query rows reuse it, so the design goal is bytes in flight, not FLOPs.
//
// Work decomposition: one WAVE per (query token, KV head, KV split); no inter-wave communication,
// no barriers. A wave walks its KV range in tiles of 32 keys (two 16-key groups; a group never
// straddles a page because page_size % 16 == 0):
//   * K and V of the NEXT tile are fetched HBM -> VGPR with row-shaped 16-byte loads (each load
//     instruction covers whole 2*D-byte key rows, so every 128-byte line is fetched once) while the
//     current tile is computed;
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
#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((ext_vector_type(8))) short s16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

constexpr int kTileKeys = 32;
constexpr float kLog2e = 1.4426950408889634f;

struct DecodeArgs {
  mi355_attn_params p;
  float* ws_acc;   // [T*Hq*num_splits][D]   un-normalised partial outputs
  float2* ws_ml;   // [T*Hq*num_splits]      (running max in log2 domain, partial sum)
  int num_splits;
  int tiles_per_split;
  int group;       // G = Hq / Hk
};

template <typename T> struct mma;
template <> struct mma<bf16_t> {
  static __device__ __forceinline__ f32x4_t run(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return pack_bf16x2(lo, hi);
  }
  static __device__ __forceinline__ u32x2_t pack4(float a, float b, float c, float d) { return u32x2_t{pack2(a, b), pack2(c, d)}; }
};
template <> struct mma<f16_t> {
  static __device__ __forceinline__ f32x4_t run(s16x8_t a, s16x8_t b, f32x4_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return pack_f16x2(lo, hi);
  }
  static __device__ __forceinline__ u32x2_t pack4(float a, float b, float c, float d) { return u32x2_t{pack2(a, b), pack2(c, d)}; }
};

// max / sum over the four lanes {g, g+16, g+32, g+48}
__device__ __forceinline__ float max_over_lane_groups(float v) {
  v = fmaxf(v, lane_xor32(v));
  return fmaxf(v, lane_xor16(v));
}
__device__ __forceinline__ float sum_over_lane_groups(float v) {
  v += lane_xor32(v);
  return v + lane_xor16(v);
}

template <typename T, int D, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void decode_splitkv_kernel(const DecodeArgs a) {
  constexpr int PPR = D / 8;            // 16-byte pieces per key row
  constexpr int NLD = PPR / 4;          // row-shaped loads per 16-key group per lane
  constexpr int RS = D * 2 + 32;        // LDS row stride in bytes
  constexpr int KSTEPS = D / 32;        // MFMA k-steps of Q.K^T
  constexpr int DBLK = D / 16;          // 16-wide output blocks of P.V
  constexpr int LDS_PER_WAVE = 48 * RS; // K: one 16-key group, V: two

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const mi355_attn_params& p = a.p;
  const int lane = threadIdx.x & 63;
  const int wave_in_wg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  char* k_lds = smem + wave_in_wg * LDS_PER_WAVE;
  char* v_lds = k_lds + 16 * RS;

  // ---- which (token, split, kv head) this wave owns (all wave-uniform) -------------------------
  const int Hk = p.num_kv_heads;
  const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * WAVES + wave_in_wg);
  const int head = item % Hk;
  const int rest = item / Hk;
  const int split = rest % a.num_splits;
  const int token = rest / a.num_splits;
  if (token >= p.num_tokens) return;

  const int seq = find_seq_by_token(p.cu_seqlens_q, p.num_seqs, token);
  const int q_start = p.cu_seqlens_q[seq];
  const int q_len = p.cu_seqlens_q[seq + 1] - q_start;
  if (p.skip_decodes && q_len == 1) return;
  if (p.only_decodes && q_len != 1) return;
  const int seq_len = p.seqused_k[seq];
  const int ctx_len = seq_len - q_len;
  const int q_pos = token - q_start;
  int n_keys = min(ctx_len + q_pos + 1, seq_len);   // causal
  if (n_keys < 0) n_keys = 0;
  int first_key = 0;                                // sliding window: keep j with q_abs - j < window
  if (p.sliding_window > 0) first_key = max(0, ctx_len + q_pos - p.sliding_window + 1);

  const int G = a.group;
  const int g = lane & 15, grp = lane >> 4;
  const bool g_ok = g < G;
  const int hq = head * G + g;

  const int tile_lo = first_key / kTileKeys;
  const int tile_hi = (n_keys + kTileKeys - 1) / kTileKeys;
  const int t0 = tile_lo + split * a.tiles_per_split;
  const int t1 = min(t0 + a.tiles_per_split, tile_hi);
  const bool direct = a.num_splits == 1;
  if (t0 >= t1) {
    if (direct && g_ok) {  // no visible key at all: the reference returns acc/L = 0/1 = 0
      const int64_t o = (int64_t)token * p.out_stride_token + (int64_t)hq * p.out_stride_head;
#pragma unroll
      for (int b = 0; b < DBLK; ++b) *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) = u32x2_t{0, 0};
    }
    return;
  }

  // ---- Q fragments: B operand of S^T = K.Q^T: lane (g, grp) holds Q[g][32c + 8grp .. +7] --------
  s16x8_t qf[KSTEPS];
  {
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)token * p.q_stride_token + (int64_t)hq * p.q_stride_head + 8 * grp;
#pragma unroll
    for (int c = 0; c < KSTEPS; ++c) {
      u32x4_t v = {0, 0, 0, 0};
      if (g_ok) v = *(const u32x4_t*)(qp + 32 * c);
      qf[c] = __builtin_bit_cast(s16x8_t, v);
    }
  }

  const float slope = (p.alibi_slopes && g_ok) ? p.alibi_slopes[hq] : 0.0f;
  const float scale2 = p.scale * kLog2e;
  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;
  const uint16_t* kbase = (const uint16_t*)p.k_cache + (int64_t)head * p.k_stride_head;
  const uint16_t* vbase = (const uint16_t*)p.v_cache + (int64_t)head * p.v_stride_head;
  const int last_group = (n_keys - 1) >> 4;  // last 16-key group that holds a visible key

  // per-lane constants of the row-shaped loads: piece idx = lane + 64*i -> (row = idx / PPR, piece = idx % PPR)
  int ld_row[NLD], ld_off[NLD];
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int idx = lane + 64 * i;
    ld_row[i] = idx / PPR;
    ld_off[i] = (idx % PPR) * 8;
  }

  auto group_ptrs = [&](int gi, const uint16_t*& kp, const uint16_t*& vp) {
    gi = min(gi, last_group);  // never index the block table past the sequence's pages
    const int key0 = gi << 4;
    const int page = bt[key0 / p.page_size];
    const int slot = key0 % p.page_size;
    kp = kbase + (int64_t)page * p.k_stride_page + (int64_t)slot * p.k_stride_slot;
    vp = vbase + (int64_t)page * p.v_stride_page + (int64_t)slot * p.v_stride_slot;
  };

  u32x4_t kreg[2][NLD], vreg[2][NLD];
  auto issue_loads = [&](int tile) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const uint16_t *kp, *vp;
      group_ptrs(tile * 2 + h, kp, vp);
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        kreg[h][i] = *(const u32x4_t*)(kp + (int64_t)ld_row[i] * p.k_stride_slot + ld_off[i]);
        vreg[h][i] = *(const u32x4_t*)(vp + (int64_t)ld_row[i] * p.v_stride_slot + ld_off[i]);
      }
    }
  };

  float m_run = -INFINITY, l_run = 0.0f;
  f32x4_t o_acc[DBLK];
#pragma unroll
  for (int b = 0; b < DBLK; ++b) o_acc[b] = f32x4_t{0, 0, 0, 0};

  issue_loads(t0);

  for (int tile = t0; tile < t1; ++tile) {
    // ---- park the current tile's rows in LDS, then refill the registers with the next tile ------
    u32x4_t kcur[2][NLD];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NLD; ++i) kcur[h][i] = kreg[h][i];
    const bool tail = (tile * kTileKeys + kTileKeys > n_keys);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NLD; ++i) {
        u32x4_t v = vreg[h][i];
        // rows past the sequence hold stale cache contents: keep NaN/Inf out of 0 * V
        if (tail && (tile * kTileKeys + h * 16 + ld_row[i] >= n_keys)) v = u32x4_t{0, 0, 0, 0};
        *(u32x4_t*)(v_lds + (h * 16 + ld_row[i]) * RS + ld_off[i] * 2) = v;
      }
    if (tile + 1 < t1) issue_loads(tile + 1);

    // ---- S^T = K . Q^T, one 16-key group at a time through the K buffer --------------------------
    f32x4_t s[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int i = 0; i < NLD; ++i) *(u32x4_t*)(k_lds + ld_row[i] * RS + ld_off[i] * 2) = kcur[h][i];
      f32x4_t acc = {0, 0, 0, 0};
#pragma unroll
      for (int c = 0; c < KSTEPS; ++c) {
        const u32x4_t kf = *(const u32x4_t*)(k_lds + g * RS + c * 64 + grp * 16);  // lane = key row g of the group
        acc = mma<T>::run(__builtin_bit_cast(s16x8_t, kf), qf[c], acc);
      }
      s[h] = acc;
    }

    // ---- scores -> log2 domain, masks (reference order: scale, softcap, causal, window, +alibi) ---
    float sv[8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[h * 4 + r] = s[h][r];
    const bool plain = !(p.softcap > 0.0f) && !p.alibi_slopes;
    const bool need_mask = tail || (tile * kTileKeys < first_key);
    if (plain && !need_mask) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sv[j] *= scale2;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = tile * kTileKeys + (j >> 2) * 16 + grp * 4 + (j & 3);
        float x = sv[j] * p.scale;
        if (p.softcap > 0.0f) x = softcap_fn(x, p.softcap);
        if (key >= n_keys || key < first_key) x = -INFINITY;
        if (p.alibi_slopes) x += slope * (float)(key - ctx_len);
        sv[j] = x * kLog2e;
      }
    }
    float mx = fmaxf(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])), fmaxf(fmaxf(sv[4], sv[5]), fmaxf(sv[6], sv[7])));
    mx = max_over_lane_groups(mx);
    float m_new = fmaxf(m_run, mx);
    if (!(m_new > -INFINITY)) m_new = 0.0f;   // fully masked so far (:486-489)
    const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
    float pv[8], psum = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      pv[j] = __builtin_amdgcn_exp2f(sv[j] - m_new);
      psum += pv[j];
    }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    // P^T fragment (B operand): k-slot j -> key 4*grp + (j&3) of group (j>>2); rounded to the KV
    // dtype before P.V like the reference (:508)
    const u32x2_t plo = mma<T>::pack4(pv[0], pv[1], pv[2], pv[3]);
    const u32x2_t phi = mma<T>::pack4(pv[4], pv[5], pv[6], pv[7]);
    const s16x8_t pf = __builtin_bit_cast(s16x8_t, u32x4_t{plo[0], plo[1], phi[0], phi[1]});

    // ---- O^T += V^T . P^T -----------------------------------------------------------------------
#pragma unroll
    for (int b = 0; b < DBLK; ++b) {
      // transposed read: lane 4q+pp of group grp addresses row 4*grp+q, columns 16b+4pp..+3 and
      // receives V[4*grp + e][16b + (lane&15)] in element e
      const int q4 = g >> 2, pp = g & 3;
      const char* va = v_lds + (grp * 4 + q4) * RS + (16 * b + 4 * pp) * 2;
      const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va));
      const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(va + 16 * RS));
      const s16x8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
      for (int r = 0; r < 4; ++r) o_acc[b][r] *= alpha;
      o_acc[b] = mma<T>::run(vf, pf, o_acc[b]);
    }
  }

  // ---- epilogue ----------------------------------------------------------------------------------
  const float l_tot = sum_over_lane_groups(l_run);
  if (!g_ok) return;
  if (direct) {
    const float inv = l_tot > 0.0f ? 1.0f / l_tot : 0.0f;
    const int64_t o = (int64_t)token * p.out_stride_token + (int64_t)hq * p.out_stride_head;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
      *(u32x2_t*)((uint16_t*)p.out + o + 16 * b + 4 * grp) =
          mma<T>::pack4(o_acc[b][0] * inv, o_acc[b][1] * inv, o_acc[b][2] * inv, o_acc[b][3] * inv);
  } else {
    const int64_t slot = ((int64_t)token * p.num_q_heads + hq) * a.num_splits + split;
    float* dst = a.ws_acc + slot * D;
#pragma unroll
    for (int b = 0; b < DBLK; ++b) *(f32x4_t*)(dst + 16 * b + 4 * grp) = o_acc[b];
    if (grp == 0) a.ws_ml[slot] = make_float2(m_run, l_tot);
  }
}

// Merge of the split partials (reference: reduce_segments, :757-836). One wave per (token, query
// head); lane d-strided over the head dimension.
template <typename T, int D>
__global__ __launch_bounds__(64) void reduce_splits_kernel(const DecodeArgs a) {
  const mi355_attn_params& p = a.p;
  const int token = blockIdx.x, hq = blockIdx.y, lane = threadIdx.x;
  const int seq = find_seq_by_token(p.cu_seqlens_q, p.num_seqs, token);
  const int q_start = p.cu_seqlens_q[seq];
  const int q_len = p.cu_seqlens_q[seq + 1] - q_start;
  if (p.skip_decodes && q_len == 1) return;
  if (p.only_decodes && q_len != 1) return;
  const int seq_len = p.seqused_k[seq];
  const int ctx_len = seq_len - q_len;
  const int q_pos = token - q_start;
  int n_keys = min(ctx_len + q_pos + 1, seq_len);
  if (n_keys < 0) n_keys = 0;
  int first_key = 0;
  if (p.sliding_window > 0) first_key = max(0, ctx_len + q_pos - p.sliding_window + 1);
  const int tile_lo = first_key / kTileKeys;
  const int tile_hi = (n_keys + kTileKeys - 1) / kTileKeys;
  const int n_tiles = max(0, tile_hi - tile_lo);
  const int active = min(a.num_splits, (n_tiles + a.tiles_per_split - 1) / a.tiles_per_split);

  const int64_t slot0 = ((int64_t)token * p.num_q_heads + hq) * a.num_splits;
  float m_all = -INFINITY;
  for (int s = 0; s < active; ++s) m_all = fmaxf(m_all, a.ws_ml[slot0 + s].x);
  float l_all = 0.0f;
  constexpr int PER = (D + 63) / 64;
  float acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = 0.0f;
  for (int s = 0; s < active; ++s) {
    const float2 ml = a.ws_ml[slot0 + s];
    const float w = __builtin_amdgcn_exp2f(ml.x - m_all);
    l_all += ml.y * w;
    const float* src = a.ws_acc + (slot0 + s) * D;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int d = lane + 64 * i;
      if (d < D) acc[i] += src[d] * w;
    }
  }
  const float inv = l_all > 0.0f ? 1.0f / l_all : 0.0f;  // "0 if the overall sum is 0" (:828)
  const int64_t o = (int64_t)token * p.out_stride_token + (int64_t)hq * p.out_stride_head;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int d = lane + 64 * i;
    if (d < D) elem<T>::store(p.out, o + d, acc[i] * inv);
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

bool decode_supported(const mi355_attn_params& p) {
  if (!(p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16)) return false;
  if (p.kv_dtype != p.q_dtype) return false;
  if (!(p.head_size == 64 || p.head_size == 128 || p.head_size == 256)) return false;
  if (p.k_new || p.v_new) return false;
  if (p.page_size % 16 != 0) return false;
  if (p.k_x != p.head_size || p.k_stride_d != 1 || p.v_stride_d != 1) return false;  // flash layout only
  const int G = p.num_q_heads / p.num_kv_heads;
  if (G > 16) return false;
  if (!aligned16(p.q) || !aligned16(p.k_cache) || !aligned16(p.v_cache)) return false;
  if (((uintptr_t)p.out & 7) != 0) return false;
  const int64_t strides[] = {p.q_stride_token, p.q_stride_head, p.k_stride_page, p.k_stride_slot, p.k_stride_head,
                             p.v_stride_page, p.v_stride_slot, p.v_stride_head};
  for (int64_t s : strides) if (s % 8 != 0) return false;
  if (p.out_stride_token % 4 != 0 || p.out_stride_head % 4 != 0) return false;
  return true;
}

struct SplitPlan { int num_splits, tiles_per_split; };

// Capture-stable split policy: depends only on host-known sizes (T, Hk, max_seqlen_k).
static SplitPlan plan_splits(const mi355_attn_params& p) {
  const int max_tiles = (p.max_seqlen_k + kTileKeys - 1) / kTileKeys;
  if (max_tiles <= 1) return {1, 1};
  int want;
  if (p.num_segments > 0) {
    want = p.num_segments;
  } else {
    const long base = (long)p.num_tokens * p.num_kv_heads;  // waves with one split each
    const long target = 256L * 8 * 2;                       // ~2 waves per resident slot (8 waves/CU)
    want = (int)((target + base - 1) / base);
    // keep each split at least 4 tiles (128 keys) long
    want = min(want, max(1, max_tiles / 4));
  }
  want = max(1, min(want, min(max_tiles, 64)));
  const int tps = (max_tiles + want - 1) / want;
  return {(max_tiles + tps - 1) / tps, tps};
}

size_t decode_workspace_bytes(const mi355_attn_params& p) {
  if (!decode_supported(p)) return 0;
  const SplitPlan sp = plan_splits(p);
  if (sp.num_splits == 1) return 0;
  const size_t slots = (size_t)p.num_tokens * p.num_q_heads * sp.num_splits;
  return slots * p.head_size * sizeof(float) + slots * sizeof(float2);
}

template <typename T, int D>
static int launch_decode_t(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  constexpr int WAVES = 4;
  DecodeArgs a;
  a.p = p;
  const SplitPlan sp = plan_splits(p);
  a.num_splits = sp.num_splits;
  a.tiles_per_split = sp.tiles_per_split;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.ws_acc = nullptr;
  a.ws_ml = nullptr;
  if (sp.num_splits > 1) {
    const size_t slots = (size_t)p.num_tokens * p.num_q_heads * sp.num_splits;
    const size_t need = slots * D * sizeof(float) + slots * sizeof(float2);
    if (!ws || ws_bytes < need) {
      set_error("decode needs a %zu-byte workspace, got %zu", need, ws_bytes);
      return MI355_ERR_WORKSPACE;
    }
    a.ws_acc = (float*)ws;
    a.ws_ml = (float2*)((char*)ws + slots * D * sizeof(float));
  }
  const long items = (long)p.num_tokens * sp.num_splits * p.num_kv_heads;
  const int grid = (int)((items + WAVES - 1) / WAVES);
  const size_t lds = (size_t)WAVES * 48 * (D * 2 + 32);
  hipLaunchKernelGGL((decode_splitkv_kernel<T, D, WAVES>), dim3(grid), dim3(WAVES * 64), lds, stream, a);
  int rc = check_hip(hipGetLastError(), "decode_splitkv_kernel launch");
  if (rc != MI355_OK) return rc;
  if (sp.num_splits > 1) {
    hipLaunchKernelGGL((reduce_splits_kernel<T, D>), dim3(p.num_tokens, p.num_q_heads), dim3(64), 0, stream, a);
    rc = check_hip(hipGetLastError(), "reduce_splits_kernel launch");
  }
  if (rc == MI355_OK) set_kernel_name(sp.num_splits > 1 ? "decode_splitkv" : "decode_single");
  return rc;
}

template <typename T>
static int launch_decode_d(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  switch (p.head_size) {
    case 64: return launch_decode_t<T, 64>(p, ws, ws_bytes, stream);
    case 128: return launch_decode_t<T, 128>(p, ws, ws_bytes, stream);
    case 256: return launch_decode_t<T, 256>(p, ws, ws_bytes, stream);
  }
  set_error("decode: head_size %d not built", p.head_size);
  return MI355_ERR_UNSUPPORTED;
}

int launch_decode(const mi355_attn_params& p, void* ws, size_t ws_bytes, hipStream_t stream) {
  if (!decode_supported(p)) {
    set_error("decode kernel does not support this configuration");
    return MI355_ERR_UNSUPPORTED;
  }
  if (p.q_dtype == MI355_BF16) return launch_decode_d<bf16_t>(p, ws, ws_bytes, stream);
  return launch_decode_d<f16_t>(p, ws, ws_bytes, stream);
}

}  // namespace mi355
