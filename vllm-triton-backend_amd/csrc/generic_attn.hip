// Shape-agnostic paged attention kernel: any dtype (f32/f16/bf16 Q, same or fp8 KV), any head
// size <= 512, any page size, flash or legacy-v0 cache layout, optional linear new-token K/V,
// ALiBi, soft-cap, sliding window. One wave64 per (query token, query head); fp32 arithmetic on
// the VALU. It is the correctness path for configurations the MFMA kernels do not cover
// (fp32 inputs, odd head sizes, legacy layouts), not the fast path.
//
// Semantics follow kernel_unified_attention_2d
// (LIB/kernels/triton_unified_attention.py:275-523): S = scale*q.k -> softcap -> causal mask ->
// sliding-window mask -> + alibi*(j - ctx) (:465-482); running max with the "-inf -> 0" guard
// (:486-489); P rounded to the V/Q dtype before P.V (:508); fp8 K/V dequantised as
// (fp8 -> f32) * scale -> Q dtype (:434-455).
#include "common.h"

namespace mi355 {

constexpr int kMaxHeadSize = 512;
constexpr int kDRegs = kMaxHeadSize / 64;

struct GenericArgs {
  mi355_attn_params p;
};

template <typename QT, typename KT>
__global__ __launch_bounds__(64) void generic_attn_kernel(const GenericArgs a) {
  const mi355_attn_params& p = a.p;
  const int token = blockIdx.x;
  const int head = blockIdx.y;
  const int lane = threadIdx.x;
  const int D = p.head_size;
  constexpr bool kKvIsFp8 = sizeof(typename KT::storage) == 1;

  __shared__ float q_s[kMaxHeadSize];
  __shared__ float p_s[64];

  const int seq = find_seq_by_token(p.cu_seqlens_q, p.num_seqs, token);
  if (seq < 0 || seq >= p.num_seqs) return;
  const int q_start = p.cu_seqlens_q[seq];
  const int q_len = p.cu_seqlens_q[seq + 1] - q_start;
  if (q_len <= p.skip_decodes) return;                       // (N: rows of sequences with up to N query tokens are another launch's)
  if (p.only_decodes && q_len > p.only_decodes) return;
  const int seq_len = p.seqused_k[seq];
  const int ctx_len = seq_len - q_len;
  const int q_pos = token - q_start;          // position inside the query
  int n_keys = p.non_causal ? seq_len : ctx_len + q_pos + 1;   // causal: keys j <= ctx + q_pos
  if (n_keys > seq_len) n_keys = seq_len;
  const int kv_head = head / (p.num_q_heads / p.num_kv_heads);
  const bool use_new = (p.k_new != nullptr) && (q_len > 1 || p.new_kv_all_rows);

  const float k_scale = (kKvIsFp8 && p.k_scale) ? p.k_scale[0] : 1.0f;
  const float v_scale = (kKvIsFp8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  const float slope = p.alibi_slopes ? p.alibi_slopes[head] : 0.0f;

  const int64_t q_off = (int64_t)token * p.q_stride_token + (int64_t)head * p.q_stride_head;
  for (int d = lane; d < D; d += 64) q_s[d] = elem<QT>::load(p.q, q_off + d);
  __syncthreads();

  float m = -INFINITY, l = 0.0f;
  float acc[kDRegs];
#pragma unroll
  for (int i = 0; i < kDRegs; ++i) acc[i] = 0.0f;

  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;

  // sliding window lower bound: keep j with (ctx + q_pos) - j < window
  int first_key = 0;
  if (p.sliding_window > 0) {
    first_key = ctx_len + q_pos - p.sliding_window + 1;
    if (first_key < 0) first_key = 0;
    first_key &= ~63;  // chunk aligned; the mask below is still applied per key
  }

  for (int base = first_key; base < n_keys; base += 64) {
    const int j = base + lane;
    float s = -INFINITY;
    if (j < n_keys) {
      float dot = 0.0f;
      if (use_new && j >= ctx_len) {
        const int64_t off = (int64_t)(q_start + j - ctx_len) * p.new_stride_token + (int64_t)kv_head * p.new_stride_head;
        for (int d = 0; d < D; ++d) dot = fmaf(q_s[d], elem<QT>::load(p.k_new, off + d), dot);
      } else {
        const int page = bt[j / p.page_size];
        const int slot = j % p.page_size;
        const int64_t off = (int64_t)page * p.k_stride_page + (int64_t)slot * p.k_stride_slot + (int64_t)kv_head * p.k_stride_head;
        for (int d = 0; d < D; ++d) {
          float kv = elem<KT>::load(p.k_cache, off + (int64_t)(d / p.k_x) * p.k_stride_dx + (int64_t)(d % p.k_x) * p.k_stride_d);
          if (kKvIsFp8) kv = elem<QT>::round(kv * k_scale);
          dot = fmaf(q_s[d], kv, dot);
        }
      }
      s = p.scale * dot;
      if (p.softcap > 0.0f) s = softcap_fn(s, p.softcap);
      if (p.sliding_window > 0 && (ctx_len + q_pos - j) >= p.sliding_window) s = -INFINITY;
      if (p.alibi_slopes) s += slope * (float)(j - ctx_len);
    }
    float m_new = fmaxf(m, wave_max(s));
    if (!(m_new > -INFINITY)) m_new = 0.0f;
    const float pj = expf(s - m_new);  // exp(-inf) = 0 for masked lanes
    const float alpha = expf(m - m_new);
    l = l * alpha + wave_sum(pj);
    m = m_new;
    __syncthreads();  // previous chunk's readers are done with p_s
    p_s[lane] = elem<QT>::round(pj);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kDRegs; ++i) acc[i] *= alpha;
    const int cnt = min(64, n_keys - base);
    for (int jj = 0; jj < cnt; ++jj) {
      const float pw = p_s[jj];
      if (pw == 0.0f) continue;  // masked (also keeps garbage V of masked slots out of the sum)
      const int jk = base + jj;
      if (use_new && jk >= ctx_len) {
        const int64_t off = (int64_t)(q_start + jk - ctx_len) * p.new_stride_token + (int64_t)kv_head * p.new_stride_head;
#pragma unroll
        for (int i = 0; i < kDRegs; ++i) {
          const int d = lane + 64 * i;
          if (d < D) acc[i] = fmaf(pw, elem<QT>::load(p.v_new, off + d), acc[i]);
        }
      } else {
        const int page = bt[jk / p.page_size];
        const int slot = jk % p.page_size;
        const int64_t off = (int64_t)page * p.v_stride_page + (int64_t)slot * p.v_stride_slot + (int64_t)kv_head * p.v_stride_head;
#pragma unroll
        for (int i = 0; i < kDRegs; ++i) {
          const int d = lane + 64 * i;
          if (d < D) {
            float vv = elem<KT>::load(p.v_cache, off + (int64_t)d * p.v_stride_d);
            if (kKvIsFp8) vv = elem<QT>::round(vv * v_scale);
            acc[i] = fmaf(pw, vv, acc[i]);
          }
        }
      }
    }
  }

  const float inv_l = (l > 0.0f) ? 1.0f / l : 0.0f;
  const int64_t o_off = (int64_t)token * p.out_stride_token + (int64_t)head * p.out_stride_head;
#pragma unroll
  for (int i = 0; i < kDRegs; ++i) {
    const int d = lane + 64 * i;
    if (d < D) elem<QT>::store(p.out, o_off + d, acc[i] * inv_l);
  }
  if (p.lse && lane == 0) p.lse[(int64_t)token * p.lse_stride_token + head] = l > 0.0f ? m + logf(l) : -INFINITY;
}

template <typename QT>
static int launch_q(const mi355_attn_params& p, hipStream_t stream) {
  GenericArgs a{p};
  dim3 grid(p.num_tokens, p.num_q_heads), block(64);
  if (p.kv_dtype == p.q_dtype) {
    hipLaunchKernelGGL((generic_attn_kernel<QT, QT>), grid, block, 0, stream, a);
  } else if (p.kv_dtype == MI355_FP8_E4M3) {
    hipLaunchKernelGGL((generic_attn_kernel<QT, e4m3_t>), grid, block, 0, stream, a);
  } else if (p.kv_dtype == MI355_FP8_E5M2) {
    hipLaunchKernelGGL((generic_attn_kernel<QT, e5m2_t>), grid, block, 0, stream, a);
  } else {
    set_error("generic attention: kv dtype %d with q dtype %d is not supported", p.kv_dtype, p.q_dtype);
    return MI355_ERR_UNSUPPORTED;
  }
  return check_hip(hipGetLastError(), "generic_attn_kernel launch");
}

int launch_generic(const mi355_attn_params& p, hipStream_t stream) {
  if (p.head_size > kMaxHeadSize) {
    set_error("head_size %d exceeds the supported maximum %d", p.head_size, kMaxHeadSize);
    return MI355_ERR_UNSUPPORTED;
  }
  if (p.num_tokens == 0) return MI355_OK;
  switch (p.q_dtype) {
    case MI355_F32: return launch_q<f32_t>(p, stream);
    case MI355_F16: return launch_q<f16_t>(p, stream);
    case MI355_BF16: return launch_q<bf16_t>(p, stream);
    default:
      set_error("generic attention: q dtype %d is not supported", p.q_dtype);
      return MI355_ERR_UNSUPPORTED;
  }
}

}  // namespace mi355
