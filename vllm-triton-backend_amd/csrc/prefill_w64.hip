// prefill_w64_kernel: the D = 128 prefill fast path, 64 query rows per wave.
//
// prefill_dma_kernel (32 rows per wave, two waves per SIMD) turned out LDS-bandwidth-bound: every
// wave re-reads the whole K and V tile from LDS for only 32 MFMAs, i.e. 1 KiB of LDS per MFMA, and
// eight such waves per CU ask for the LDS array's full 256 B/clk exactly when the matrix pipes are
// saturated. Here one wave owns TWO 32-row sub-blocks (A, B) of the Q block and every K fragment /
// transposed V fragment read from LDS feeds two MFMAs (one per sub-block): half the LDS bytes, half
// the DMA issue work and half the barrier crossings per MFMA. The price is registers: O (2 x 64),
// S (2 x 32), Q (2 x 32) ... ~400 per lane, so the kernel runs one wave per SIMD on the whole
// 512-entry register file (accumulators in the AGPR half).
//
// Everything else is as in prefill_dma_kernel: rows ordered (token, head-in-group) like the
// reference (LIB/kernels/triton_unified_attention.py:343-346); swapped product S^T = K.Q'^T on
// v_mfma_f32_32x32x16 so that a lane owns a query row; Q pre-scaled into the log2 domain; running
// max carried in the MFMA accumulator's initial value and moved only when a row's tile max exceeds
// it by more than 2^8 (softmax is shift-invariant); bf16-packed S accumulators reused directly as
// the B operand of O^T += V^T.P^T; K/V tiles HBM -> LDS by LDS-DMA into two XOR-swizzled stages;
// block-table entries held 64 at a time in a VGPR and picked with v_readlane (no scalar-cache
// round trip inside the loop); KV head fastest in the grid so that each XCD's L2 serves one head.
#include <cstdlib>

#include "common.h"

namespace mi355 {

typedef __attribute__((ext_vector_type(8))) __bf16 wbf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 wf16x8_t;
typedef __attribute__((ext_vector_type(4))) short ws16x4_t;
typedef __attribute__((ext_vector_type(8))) short ws16x8_t;
typedef __attribute__((ext_vector_type(16))) float wf32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int wu32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int wu32x2_t;

constexpr int kW64Tile = 64;        // keys per KV tile
constexpr float kW64Log2e = 1.4426950408889634f;
constexpr float kW64DeferThr = 8.0f;

struct W64Args {
  mi355_attn_params p;
  int group;       // G
  int block_q;     // tokens per Q block = 64*WAVES / G
  int page_shift;  // log2(page_size)
  uint32_t k_page_stride, k_slot_stride, v_page_stride, v_slot_stride;  // elements
};

template <typename T> struct wmma;
template <> struct wmma<bf16_t> {
  static __device__ __forceinline__ wf32x16_t run(ws16x8_t a, ws16x8_t b, wf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wbf16x8_t, a), __builtin_bit_cast(wbf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_bf16x2(lo, hi); }
  static __device__ __forceinline__ float lo(uint32_t w) { return bf16_to_f32((uint16_t)(w & 0xffff)); }
  static __device__ __forceinline__ float hi(uint32_t w) { return bf16_to_f32((uint16_t)(w >> 16)); }
};
template <> struct wmma<f16_t> {
  static __device__ __forceinline__ wf32x16_t run(ws16x8_t a, ws16x8_t b, wf32x16_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wf16x8_t, a), __builtin_bit_cast(wf16x8_t, b), c, 0, 0, 0);
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return pack_f16x2(lo, hi); }
  static __device__ __forceinline__ float lo(uint32_t w) { return f16_to_f32((uint16_t)(w & 0xffff)); }
  static __device__ __forceinline__ float hi(uint32_t w) { return f16_to_f32((uint16_t)(w >> 16)); }
};

// MFMA through inline asm so that the register CLASS of every operand is ours to choose: O
// accumulators and the Q fragments live in the AGPR half of the file for the whole kernel (they are
// only ever MFMA operands), S accumulators / K, V, P fragments in the VGPR half (the VALU touches
// them). Left to itself hipcc (ROCm 7.2) shuffles ~700 v_accvgpr_read/write per tile here.
// hipcc neither waits for nor pads an asm MFMA (cdna_hip_programming.md 5.7): mfma_settle() supplies
// the wait states between the last MFMA of a chain and the first non-MFMA reader of its result,
// valu_to_mfma_pad() those between a VALU write and an MFMA that reads it.
template <typename T> struct amma;
#define MI355_DEF_AMMA(TAG, MNEM)                                                                              \
  template <> struct amma<TAG> {                                                                               \
    /* o(AGPR) += a(VGPR) * b(VGPR) */                                                                          \
    static __device__ __forceinline__ void acc_o(wf32x16_t& o, ws16x8_t a, ws16x8_t b) {                       \
      asm volatile(MNEM " %0, %1, %2, %0" : "+a"(o) : "v"(a), "v"(b));                                          \
    }                                                                                                          \
    /* s(VGPR) = a(VGPR) * q(AGPR) + c(VGPR) */                                                                  \
    static __device__ __forceinline__ void first_s(wf32x16_t& s, wu32x4_t a, ws16x8_t q, const wf32x16_t& c) { \
      asm volatile(MNEM " %0, %1, %2, %3" : "=&v"(s) : "v"(a), "a"(q), "v"(c));                                  \
    }                                                                                                          \
    /* s(VGPR) += a(VGPR) * q(AGPR) */                                                                           \
    static __device__ __forceinline__ void acc_s(wf32x16_t& s, wu32x4_t a, ws16x8_t q) {                       \
      asm volatile(MNEM " %0, %1, %2, %0" : "+v"(s) : "v"(a), "a"(q));                                          \
    }                                                                                                          \
  };
MI355_DEF_AMMA(bf16_t, "v_mfma_f32_32x32x16_bf16")
MI355_DEF_AMMA(f16_t, "v_mfma_f32_32x32x16_f16")
#undef MI355_DEF_AMMA

__device__ __forceinline__ void mfma_settle_v(wf32x16_t& a, wf32x16_t& b, wf32x16_t& c, wf32x16_t& d) {
  asm volatile("s_nop 15\n\ts_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ void mfma_settle_a(wf32x16_t& a, wf32x16_t& b, wf32x16_t& c, wf32x16_t& d) {
  asm volatile("s_nop 15\n\ts_nop 7" : "+a"(a), "+a"(b), "+a"(c), "+a"(d));
}
__device__ __forceinline__ void valu_to_mfma_pad(ws16x8_t& a, ws16x8_t& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

__device__ __forceinline__ int w64_find_seq(const int32_t* __restrict__ cu, int num_seqs, int qblock, int block_q) {
  int left = 0, right = num_seqs;
  while (left < right) {
    const int mid = (left + right) >> 1;
    if (cu[mid] / block_q + mid <= qblock) left = mid + 1; else right = mid;
  }
  return left - 1;
}

template <typename T, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void prefill_w64_kernel(const W64Args a) {
  static_assert(WAVES == 4, "staging is written for 256 threads");
  constexpr int D = 128, ROWB = 256;
  constexpr int KBUF = kW64Tile * ROWB, STAGE = 2 * KBUF;
  constexpr int KSTEPS = D / 16, DBLK = D / 32;
  constexpr int ROWS = 64 * WAVES;

  extern __shared__ __attribute__((aligned(16))) char smem[];   // two stages of (K tile, V tile)
  const mi355_attn_params& p = a.p;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int G = a.group, BQ = a.block_q;

  const int head = (int)(blockIdx.x % p.num_kv_heads);
  const int qblock = (int)(gridDim.x / p.num_kv_heads - 1 - blockIdx.x / p.num_kv_heads);   // heaviest first
  const int seq = w64_find_seq(p.cu_seqlens_q, p.num_seqs, qblock, BQ);
  if (seq < 0) return;
  const int q_start = p.cu_seqlens_q[seq];
  const int q_len = p.cu_seqlens_q[seq + 1] - q_start;
  const int qb_local = qblock - (q_start / BQ + seq);
  if (qb_local * BQ >= q_len) return;
  if (p.skip_decodes && q_len == 1) return;
  if (p.only_decodes && q_len != 1) return;
  const int seq_len = p.seqused_k[seq];
  const int ctx_len = seq_len - q_len;
  const int tok0 = qb_local * BQ;

  // ---- this lane's two query rows (sub-blocks A = 0, B = 1) -----------------------------------------
  const int qr = lane & 31, half = lane >> 5;
  int tok_local[2], hq[2], lim[2];
  bool row_ok[2];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const int m_row = wave * 64 + sb * 32 + qr;
    tok_local[sb] = tok0 + m_row / G;
    hq[sb] = head * G + m_row % G;
    row_ok[sb] = (m_row < BQ * G) && (tok_local[sb] < q_len);
    lim[sb] = row_ok[sb] ? min(ctx_len + tok_local[sb], seq_len - 1) : -1;   // last visible key
  }
  const int w_tok_lo = tok0 + (wave * 64) / G;
  const int w_tok_hi = min(min(tok0 + (wave * 64 + 63) / G, tok0 + BQ - 1), q_len - 1);
  const int wg_tok_hi = min(tok0 + BQ - 1, q_len - 1);
  const int n_keys_wg = max(0, min(ctx_len + wg_tok_hi + 1, seq_len));
  const int wave_keys = min(ctx_len + w_tok_hi + 1, seq_len);
  const bool wave_has_rows = w_tok_lo <= w_tok_hi;
  const int tile_hi = (n_keys_wg + kW64Tile - 1) / kW64Tile;

  // ---- Q fragments, pre-scaled into the log2 domain -------------------------------------------------
  const float scale2 = p.scale * kW64Log2e;
  ws16x8_t qf[2][KSTEPS];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    const uint16_t* qp = (const uint16_t*)p.q + (int64_t)(q_start + tok_local[sb]) * p.q_stride_token + (int64_t)hq[sb] * p.q_stride_head + 8 * half;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      wu32x4_t v = {0, 0, 0, 0};
      if (row_ok[sb]) v = *(const wu32x4_t*)(qp + 16 * ks);
      wu32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = wmma<T>::pack2(wmma<T>::lo(v[e]) * scale2, wmma<T>::hi(v[e]) * scale2);
      // give the fragment an AGPR home once (a tied empty asm: the copy happens here, not per MFMA)
      ws16x8_t qv = __builtin_bit_cast(ws16x8_t, w), qa;
      asm volatile("" : "=a"(qa) : "0"(qv));
      qf[sb][ks] = qa;
    }
  }

  // ---- DMA staging ----------------------------------------------------------------------------------
  // thread handles LDS chunk (row = (tid>>4) + 16 i, c = tid & 15) of both tiles; load i <-> 16-key group i
  const int32_t* bt = p.block_table + (int64_t)seq * p.block_table_stride;
  const char* kbase = (const char*)p.k_cache + (int64_t)head * p.k_stride_head * 2;
  const char* vbase = (const char*)p.v_cache + (int64_t)head * p.v_stride_head * 2;
  const int last_group = (max(n_keys_wg, 1) - 1) >> 4;
  const int last_entry = (last_group << 4) >> a.page_shift;
  const int page_mask = p.page_size - 1;
  const int rowin = tid >> 4, ch = tid & 15;
  const int fk = rowin & 15;
  const int fv = ((rowin & 3) << 2) | ((rowin >> 2) & 3);
  const uint32_t k_voff = (uint32_t)(rowin * (int)a.k_slot_stride * 2 + ((ch ^ fk) << 4));
  const uint32_t v_voff = (uint32_t)(rowin * (int)a.v_slot_stride * 2 + ((ch ^ fv) << 4));
  const uint32_t lds_wave = (uint32_t)(wave * 64 * 16);        // + i*4096 (+KBUF for V) + stage
  const uint32_t k_page_bytes = a.k_page_stride * 2, v_page_bytes = a.v_page_stride * 2;

  // block-table entries, 64 at a time in a VGPR (lane l = entry chunk*64 + l), one chunk ahead
  int bt_chunk = 0;
  int bt_cur = bt[min(lane, last_entry)];
  int bt_nxt = bt[min(64 + lane, last_entry)];

  auto issue_dma = [&](int tile, char* stage) {
    const int e0 = (min(tile * 4, last_group) << 4) >> a.page_shift;
    if ((e0 >> 6) != bt_chunk) {            // wave-uniform; entries only ever move forward
      bt_chunk = e0 >> 6;
      bt_cur = bt_nxt;
      bt_nxt = bt[min((bt_chunk + 1) * 64 + lane, last_entry)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int gi = min(tile * 4 + i, last_group);
      const int key0 = gi << 4;
      const int slot0 = key0 & page_mask;
      const int page = __builtin_amdgcn_readlane(bt_cur, (key0 >> a.page_shift) & 63);
      uint32_t kvo = k_voff, vvo = v_voff;
      if (key0 + 16 > seq_len) {          // wave-uniform: the sequence ends inside this group -> rows past it
                                          // fetch its last row instead (never stale cache contents)
        const int r = min(rowin, max(seq_len - 1 - key0, 0));
        kvo = (uint32_t)(r * (int)a.k_slot_stride * 2 + ((ch ^ fk) << 4));
        vvo = (uint32_t)(r * (int)a.v_slot_stride * 2 + ((ch ^ fv) << 4));
      }
      const char* kp = kbase + ((uint64_t)(uint32_t)page * k_page_bytes + (uint64_t)((uint32_t)slot0 * a.k_slot_stride) * 2);
      const char* vp = vbase + ((uint64_t)(uint32_t)page * v_page_bytes + (uint64_t)((uint32_t)slot0 * a.v_slot_stride) * 2);
      glds16(kp + kvo, lds_addr(stage) + lds_wave + i * 4096);
      glds16(vp + vvo, lds_addr(stage) + KBUF + lds_wave + i * 4096);
    }
  };

  // ---- per-lane LDS read addresses (swizzle folded in) ----------------------------------------------
  uint32_t k_rd[KSTEPS];       // K fragment ks of 32-key block kb: row 32kb + qr, logical chunk 2ks + half
#pragma unroll
  for (int ks = 0; ks < KSTEPS; ++ks) k_rd[ks] = (uint32_t)(qr * ROWB + (((2 * ks + half) ^ (qr & 15)) << 4));
  // V transposed read of k-step sk (16 keys), output block b: row 16sk + 4half + q4 (+8), logical byte
  // column 64b + 32g1 + 8pp -> chunk 4b + 2g1 + (pp>>1), sub-offset 8(pp&1)
  const int gq1 = (lane >> 4) & 1, li = lane & 15, q4 = li >> 2, pp = li & 3;
  uint32_t v_rd0[DBLK], v_rd1[DBLK];
#pragma unroll
  for (int b = 0; b < DBLK; ++b) {
    const int lc = 4 * b + 2 * gq1 + (pp >> 1);
    const int r0 = 4 * half + q4, r1 = r0 + 8;
    const int f0 = ((r0 & 3) << 2) | ((r0 >> 2) & 3), f1 = ((r1 & 3) << 2) | ((r1 >> 2) & 3);
    v_rd0[b] = (uint32_t)(KBUF + r0 * ROWB + ((lc ^ f0) << 4) + 8 * (pp & 1));
    v_rd1[b] = (uint32_t)(KBUF + r1 * ROWB + ((lc ^ f1) << 4) + 8 * (pp & 1));
  }

  float m_ref[2] = {0.0f, 0.0f}, l_run[2] = {0.0f, 0.0f};
  bool started[2] = {!row_ok[0], !row_ok[1]};   // padding rows never see a key
  wf32x16_t cinit[2];                           // -m_ref in every register: the C operand that starts S^T
  wf32x16_t o_acc[2][DBLK];
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
    for (int r = 0; r < 16; ++r) cinit[sb][r] = 0.0f;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) o_acc[sb][b][r] = 0.0f;
  }

  if (tile_hi > 0) issue_dma(0, smem);
  glds_wait_all();
  __syncthreads();

  auto compute_tile = [&](int tile, const char* stage) {
    const int key_base = tile * kW64Tile;
    // ---- S^T - m_ref = K . Q'^T + cinit, both sub-blocks per K fragment ---------------------------------
    wf32x16_t s_acc[2][2];   // [sub-block][32-key block]
#ifdef MI355_ABLATE_QK
    s_acc[0][0] = cinit[0]; s_acc[0][1] = cinit[0]; s_acc[1][0] = cinit[1]; s_acc[1][1] = cinit[1];
#else
    wu32x4_t kf[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) kf[ks] = *(const wu32x4_t*)(stage + k_rd[ks]);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks == 0) {
        amma<T>::first_s(s_acc[0][0], kf[ks], qf[0][ks], cinit[0]);
        amma<T>::first_s(s_acc[1][0], kf[ks], qf[1][ks], cinit[1]);
      } else {
        amma<T>::acc_s(s_acc[0][0], kf[ks], qf[0][ks]);
        amma<T>::acc_s(s_acc[1][0], kf[ks], qf[1][ks]);
      }
      kf[ks] = *(const wu32x4_t*)(stage + 32 * ROWB + k_rd[ks]);
    }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if (ks == 0) {
        amma<T>::first_s(s_acc[0][1], kf[ks], qf[0][ks], cinit[0]);
        amma<T>::first_s(s_acc[1][1], kf[ks], qf[1][ks], cinit[1]);
      } else {
        amma<T>::acc_s(s_acc[0][1], kf[ks], qf[0][ks]);
        amma<T>::acc_s(s_acc[1][1], kf[ks], qf[1][ks]);
      }
    }
#endif
    mfma_settle_v(s_acc[0][0], s_acc[0][1], s_acc[1][0], s_acc[1][1]);
    // ---- all 16 transposed V fragments of the tile are requested now and kept in registers: they
    //      land while the VALU does sub-block A's softmax, feed P.V of A and are reused for B ------------
    ws16x8_t vfr[16];   // [4*b + sk]
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int b = j >> 2, sk = j & 3;
      const ws16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4_t*)(stage + sk * 16 * ROWB + v_rd0[b]));
      const ws16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) ws16x4_t*)(stage + sk * 16 * ROWB + v_rd1[b]));
      vfr[j] = __builtin_shufflevector(v0, v1, 0, 1, 2, 3, 4, 5, 6, 7);
    }
    // ---- softmax ----------------------------------------------------------------------------------------
    const bool need_mask = (key_base + kW64Tile - 1 > ctx_len + w_tok_lo) || (key_base + kW64Tile > seq_len);
    // mask, running-max check and (rarely) the move of the reference max for one sub-block
    auto settle_rows = [&](int sb) {
      if (need_mask) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int key = key_base + 32 * kb + (r & 3) + 8 * (r >> 2) + 4 * half;
            s_acc[sb][kb][r] = key <= lim[sb] ? s_acc[sb][kb][r] : -INFINITY;
          }
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[sb][kb][r]);
      mx = fmaxf(mx, lane_xor32(mx));                     // the other half-wave holds the other 32 keys
      const bool calm = started[sb] && mx <= kW64DeferThr;
      if (!__all(calm)) {
        const float upd = (!calm && mx > -INFINITY) ? mx : 0.0f;
        started[sb] = started[sb] || (mx > -INFINITY);
        m_ref[sb] += upd;
        const float alpha = __builtin_amdgcn_exp2f(-upd);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s_acc[sb][kb][r] -= upd;
#pragma unroll
        for (int r = 0; r < 16; ++r) cinit[sb][r] = -m_ref[sb];
        l_run[sb] *= alpha;
        mfma_settle_a(o_acc[sb][0], o_acc[sb][1], o_acc[sb][2], o_acc[sb][3]);
#pragma unroll
        for (int b = 0; b < DBLK; ++b)
#pragma unroll
          for (int r = 0; r < 16; ++r) o_acc[sb][b][r] *= alpha;
      }
    };
    // P dword j of a sub-block = exp2 of S registers (kb = j>>3, r = 2(j&7), r+1), bf16-packed
    uint32_t pw[2][16];
    float psum[2] = {0.0f, 0.0f};
    auto exp_step = [&](int sb, int j) {
      const int kb = j >> 3, r = 2 * (j & 7);
#ifdef MI355_ABLATE_SOFTMAX
      asm volatile("" :: "v"(s_acc[sb][kb][r]), "v"(s_acc[sb][kb][r + 1]));
      pw[sb][j] = 0x3c003c00u;
#else
      const float e0 = __builtin_amdgcn_exp2f(s_acc[sb][kb][r]), e1 = __builtin_amdgcn_exp2f(s_acc[sb][kb][r + 1]);
      psum[sb] += e0 + e1;
      pw[sb][j] = wmma<T>::pack2(e0, e1);
#endif
    };
    auto pfrag = [&](int sb, int sk) {   // B operand of k-step sk (16 keys) of P.V
      return __builtin_bit_cast(ws16x8_t, wu32x4_t{pw[sb][4 * sk], pw[sb][4 * sk + 1], pw[sb][4 * sk + 2], pw[sb][4 * sk + 3]});
    };

    // phase 2: sub-block A's softmax (VALU alone; the V reads above land underneath)
    settle_rows(0);
#pragma unroll
    for (int j = 0; j < 16; ++j) exp_step(0, j);
    l_run[0] += psum[0];
    ws16x8_t pfa[4] = {pfrag(0, 0), pfrag(0, 1), pfrag(0, 2), pfrag(0, 3)};
    // sub-block B: mask / max check now, its exponentials ride in the MFMA gaps of phase 3
    settle_rows(1);
    valu_to_mfma_pad(pfa[0], pfa[3]);
    __builtin_amdgcn_sched_barrier(0);
    // phase 3: O_A^T += V^T . P_A^T (16 MFMAs) with B's 32 exp2 + sums + packs between them
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#ifndef MI355_ABLATE_PV
      amma<T>::acc_o(o_acc[0][j >> 2], vfr[j], pfa[j & 3]);
#else
      asm volatile("" :: "v"(vfr[j]), "v"(pfa[j & 3]));
#endif
      exp_step(1, j);
      __builtin_amdgcn_sched_barrier(0);
    }
    l_run[1] += psum[1];
    ws16x8_t pfb[4] = {pfrag(1, 0), pfrag(1, 1), pfrag(1, 2), pfrag(1, 3)};
    valu_to_mfma_pad(pfb[0], pfb[3]);
    // phase 4: O_B^T += V^T . P_B^T from the same V fragments
#pragma unroll
    for (int j = 0; j < 16; ++j) {
#ifndef MI355_ABLATE_PV
      amma<T>::acc_o(o_acc[1][j >> 2], vfr[j], pfb[j & 3]);
#else
      asm volatile("" :: "v"(vfr[j]), "v"(pfb[j & 3]));
#endif
    }
  };

  // tile loop, two tiles per trip so that the LDS stage is a compile-time offset
  for (int tile = 0; tile < tile_hi; tile += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int t = tile + u;
      if (t < tile_hi) {
        char* cur = smem + u * STAGE;
        char* nxt = smem + (u ^ 1) * STAGE;
#ifndef MI355_ABLATE_DMA
        if (t + 1 < tile_hi) issue_dma(t + 1, nxt);
#endif
        if (wave_has_rows && t * kW64Tile < wave_keys) compute_tile(t, cur);
        glds_wait_all();     // this wave's pieces of tile t+1 have landed ...
#ifndef MI355_ABLATE_BARRIER
        __syncthreads();     // ... and so have everyone else's; stage `cur` is free
#endif
      }
    }
  }

  // ---- epilogue --------------------------------------------------------------------------------------
#pragma unroll
  for (int sb = 0; sb < 2; ++sb) {
    mfma_settle_a(o_acc[sb][0], o_acc[sb][1], o_acc[sb][2], o_acc[sb][3]);
    const float l = l_run[sb] + lane_xor32(l_run[sb]);
    if (!row_ok[sb]) continue;
    const float inv = l > 0.0f ? 1.0f / l : 0.0f;
    uint16_t* op = (uint16_t*)p.out + (int64_t)(q_start + tok_local[sb]) * p.out_stride_token + (int64_t)hq[sb] * p.out_stride_head + 4 * half;
#pragma unroll
    for (int b = 0; b < DBLK; ++b)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const wu32x2_t w = {wmma<T>::pack2(o_acc[sb][b][4 * c + 0] * inv, o_acc[sb][b][4 * c + 1] * inv),
                            wmma<T>::pack2(o_acc[sb][b][4 * c + 2] * inv, o_acc[sb][b][4 * c + 3] * inv)};
        *(wu32x2_t*)(op + 32 * b + 8 * c) = w;
      }
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <typename T>
static int launch_w64_t(const mi355_attn_params& p, hipStream_t stream) {
  constexpr int WAVES = 4;
  W64Args a;
  a.p = p;
  a.group = p.num_q_heads / p.num_kv_heads;
  a.block_q = (64 * WAVES) / a.group;
  a.page_shift = __builtin_ctz((unsigned)p.page_size);
  a.k_page_stride = (uint32_t)p.k_stride_page; a.k_slot_stride = (uint32_t)p.k_stride_slot;
  a.v_page_stride = (uint32_t)p.v_stride_page; a.v_slot_stride = (uint32_t)p.v_stride_slot;
  const int qblocks = p.num_tokens / a.block_q + p.num_seqs;   // static upper bound, as the reference (:886-889,:935-943)
  constexpr size_t lds = 2 * 2 * (size_t)kW64Tile * 256;
  static bool attr_set = false;
  if (!attr_set) {
    const int rc0 = check_hip(hipFuncSetAttribute((const void*)prefill_w64_kernel<T, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds),
                              "hipFuncSetAttribute(prefill_w64)");
    if (rc0 != MI355_OK) return rc0;
    attr_set = true;
  }
  hipLaunchKernelGGL((prefill_w64_kernel<T, WAVES>), dim3(qblocks * p.num_kv_heads), dim3(WAVES * 64), lds, stream, a);
  const int rc = check_hip(hipGetLastError(), "prefill_w64_kernel launch");
  if (rc == MI355_OK) set_kernel_name("prefill_mfma");
  return rc;
}

// Preconditions beyond prefill_supported(): head size 128, no soft-cap / ALiBi / sliding window,
// G <= 256 rows.
bool prefill_w64_applicable(const mi355_attn_params& p) {
  const bool feat = p.softcap > 0.0f || p.alibi_slopes != nullptr || p.sliding_window > 0;
  const int G = p.num_q_heads / p.num_kv_heads;
  return !feat && p.head_size == 128 && G <= 256 && p.kv_dtype == p.q_dtype;
}

int launch_prefill_w64(const mi355_attn_params& p, hipStream_t stream) {
  return p.q_dtype == MI355_BF16 ? launch_w64_t<bf16_t>(p, stream) : launch_w64_t<f16_t>(p, stream);
}

}  // namespace mi355
