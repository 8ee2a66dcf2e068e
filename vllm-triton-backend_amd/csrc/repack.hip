// Repack path for the legacy ops (context_attention_fwd, chunked_prefill_paged_decode;
// LIB/kernels/legacy/triton_prefix_prefill.py:588-765, triton_chunked_prefill_paged_decode.py:28-117).
//
// Those ops read context keys from the vLLM v0 cache layout (K [nb, Hk, D/x, page, x], V [nb, Hk, D, page]) and the
// keys of the tokens being prefilled from linear [T, Hk, D] tensors (the cache may be fp8, the linear tensors never are). The matrix-core prefill kernel stages whole
// flash-layout rows through LDS-DMA; rather than a second copy of that kernel for a d-major V and a two-source key
// stream, one pass gathers every sequence's keys - context pages from the cache (any layout the ABI's strides
// describe), new rows from the linear tensors - into a flash-layout scratch cache in the caller's workspace with an
// identity block table, and the prefill kernel runs on that. The pass moves 2x the K/V bytes once; prefill does
// O(query_len) more work per key than that, so it is noise next to the attention itself.
#include <cstdlib>

#include "common.h"

namespace mi355 {

namespace {

constexpr int kRepackPage = 16;   // page size of the scratch cache
constexpr size_t kRepackMaxBytes = (size_t)32 << 30;

struct RepackArgs {
  mi355_attn_params p;
  uint16_t* k_dst;
  uint16_t* v_dst;
  int32_t* bt_dst;
  int pages_per_seq;
  int total_pages;  // scratch pages in all (block-table entries past a sequence's pages are clamped to it: never dereferenced, but always valid)
  int tight;        // 1: a sequence's scratch pages start at cu_seqlens_q[seq] / 16 + seq (self-attention: keys == query tokens, see layout())
  int vec_k, vec_v, vec_new;   // 16-byte (fp8: 8-byte) loads of eight head dims are legal for that source
  int v_keys_contiguous;       // V is d-major with a page's keys contiguous (v0): eight KEYS per load, turned in LDS
  int skip_single;             // sequences with query_len == 1 are not repacked (left out or served from the cache)
};

typedef __attribute__((ext_vector_type(4))) unsigned int ru32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int ru32x2_t;

// cache pages are read once by this pass: streaming (nt) loads, see decode_splitkv.hip
__device__ inline uint4 load16_stream(const void* ptr) {
  const ru32x4_t v = __builtin_nontemporal_load((const ru32x4_t*)ptr);
  return uint4{v.x, v.y, v.z, v.w};
}
__device__ inline uint2 load8_stream(const void* ptr) {
  const ru32x2_t v = __builtin_nontemporal_load((const ru32x2_t*)ptr);
  return uint2{v.x, v.y};
}

__device__ inline uint4 pack8(const uint16_t (&e)[8]) {
  return uint4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
               (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
}

__device__ inline uint4 gather8(const uint16_t* base, int64_t off0, int64_t stride) {
  uint16_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = base[off0 + i * stride];
  return pack8(e);
}

// Eight consecutive head dims of one cached key as 16-bit elements of the query type. A 16-bit cache is moved as raw
// bits; an fp8 cache is dequantised the way the reference's kernels do on load, (fp8 -> f32) * scale -> query type
// (legacy/triton_prefix_prefill.py:154-155,:208-209; triton_unified_attention.py:434-455), so the scratch cache needs no scales.
template <typename QT, typename KVT>
struct CachePiece {
  static constexpr bool kFp8 = !__is_same(QT, KVT);
  // element (dim d0 + i) lives at off(i); `vec`: the eight are contiguous and aligned
  template <typename OffFn>
  static __device__ inline uint4 load(const void* cache, bool vec, float scale, OffFn off) {
    uint16_t e[8];
    if constexpr (!kFp8) {
      const uint16_t* c = (const uint16_t*)cache;
      if (vec) return load16_stream(c + off(0));
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = c[off(i)];
    } else {
      const uint8_t* c = (const uint8_t*)cache;
      uint2 w;
      if (vec) {
        w = load8_stream(c + off(0));
      } else {
        w = uint2{0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) { w.x |= (uint32_t)c[off(i)] << (8 * i); w.y |= (uint32_t)c[off(4 + i)] << (8 * i); }
      }
      // v_cvt_scalef32_pk_f32_{fp8,bf8} at scale 1: exact (every fp8 value is an f32); then the reference's f32
      // multiply and one rounding to the query type (v_cvt_pk_{bf16,f16}_f32, round to nearest even)
      typedef __attribute__((ext_vector_type(2))) float f2;
      auto widen = [](uint32_t word, f2& lo, f2& hi) {
        if constexpr (__is_same(KVT, e4m3_t)) {
          lo = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(word, 1.0f, false);
          hi = __builtin_amdgcn_cvt_scalef32_pk_f32_fp8(word, 1.0f, true);
        } else {
          lo = __builtin_amdgcn_cvt_scalef32_pk_f32_bf8(word, 1.0f, false);
          hi = __builtin_amdgcn_cvt_scalef32_pk_f32_bf8(word, 1.0f, true);
        }
      };
      f2 f[4];
      widen(w.x, f[0], f[1]);
      widen(w.y, f[2], f[3]);
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (__is_same(QT, bf16_t)) o[i] = pack_bf16x2(f[i].x * scale, f[i].y * scale); else o[i] = pack_f16x2(f[i].x * scale, f[i].y * scale);
      }
      return uint4{o[0], o[1], o[2], o[3]};
    }
    return pack8(e);
  }
};

// grid (ceil(pages_per_seq / kPagesPerWg), num_seqs, Hk), 256 threads: kPagesPerWg consecutive scratch pages (16 keys x D
// each) of one sequence and KV head per workgroup, moved as 16-byte pieces (8 head dims of one key). A d-major V (v0
// layout: the 16 keys of a page are contiguous per head dim) is read along the keys, 8 per load, and turned through LDS;
// reading it along d would be eight 2-byte loads per piece.
// (Round 3: four pages per workgroup instead of one. One page is a single piece of K and of V per thread behind three
// dependent loads - lengths, block table, data: latency bound at ~1.5 TB/s on the 25 MB of a 4096-token sequence. With
// four pages a thread has its block-table entries and then eight pieces in flight together.)
constexpr int kPagesPerWg = 4;

template <typename QT, typename KVT>
__global__ __launch_bounds__(256) void repack_kernel(RepackArgs a) {
  const mi355_attn_params& p = a.p;
  const int pg0 = blockIdx.x * kPagesPerWg, seq = blockIdx.y, h = blockIdx.z;
  const int seq_len = p.seqused_k[seq];
  const int q_start = p.cu_seqlens_q[seq], q_len = p.cu_seqlens_q[seq + 1] - q_start;
  // first scratch page of this sequence: its row of the padded [num_seqs][pages_per_seq] grid, or - tight - the reference's
  // Q-block numbering applied to pages (an upper bound of the pages of the sequences before it, :935-943)
  const int64_t page_base = a.tight ? (int64_t)(q_start / kRepackPage + seq) : (int64_t)seq * a.pages_per_seq;
  if (h == 0 && threadIdx.x < kPagesPerWg && pg0 + (int)threadIdx.x < a.pages_per_seq)
    a.bt_dst[(int64_t)seq * a.pages_per_seq + pg0 + threadIdx.x] = (int32_t)min(page_base + pg0 + (int64_t)threadIdx.x, (int64_t)a.total_pages - 1);
  if (q_len <= 0 || (a.skip_single && q_len == 1)) return;
  if (pg0 * kRepackPage >= seq_len) return;
  const int ctx = seq_len - q_len;
  const bool use_new = p.k_new != nullptr && (q_len > 1 || p.new_kv_all_rows);   // generic_attn.hip: same rule
  const int D = p.head_size, Hk = p.num_kv_heads, chunks = D >> 3;
  const int pieces = kRepackPage * chunks;
  const uint16_t* kn = (const uint16_t*)p.k_new;
  const uint16_t* vn = (const uint16_t*)p.v_new;
  constexpr bool kFp8 = CachePiece<QT, KVT>::kFp8;
  const float k_scale = (kFp8 && p.k_scale) ? p.k_scale[0] : 1.0f;
  const float v_scale = (kFp8 && p.v_scale) ? p.v_scale[0] : 1.0f;
  const int32_t* bt_row = p.block_table + (int64_t)seq * p.block_table_stride;

  __shared__ uint16_t vt[256][kRepackPage + 2];
  // the common case in one go: every page's pieces are one per thread (D = 128: 256 pieces), all from the cache or all
  // from the linear tensors, V row-major -> the block-table entries, then all loads, then all stores
  const bool simple = pieces == 256 && !a.v_keys_contiguous;
  if (simple) {
    const int c = threadIdx.x % chunks, slot = threadIdx.x / chunks, d0 = 8 * c;
    int page[kPagesPerWg];
#pragma unroll
    for (int u = 0; u < kPagesPerWg; ++u) {
      const int j = (pg0 + u) * kRepackPage + slot;
      page[u] = (j < seq_len && !(use_new && j >= ctx)) ? bt_row[j / p.page_size] : 0;
    }
    uint4 kk[kPagesPerWg], vv[kPagesPerWg];
#pragma unroll
    for (int u = 0; u < kPagesPerWg; ++u) {
      const int j = (pg0 + u) * kRepackPage + slot;
      kk[u] = uint4{0u, 0u, 0u, 0u}; vv[u] = uint4{0u, 0u, 0u, 0u};   // slots past the sequence end are zero-filled
      if (j < seq_len) {
        if (use_new && j >= ctx) {
          const int64_t off = (int64_t)(q_start + j - ctx) * p.new_stride_token + (int64_t)h * p.new_stride_head + d0;
          if (a.vec_new) { kk[u] = *(const uint4*)(kn + off); vv[u] = *(const uint4*)(vn + off); }
          else { kk[u] = gather8(kn, off, 1); vv[u] = gather8(vn, off, 1); }
        } else {
          const int o = j % p.page_size;
          const int64_t kb = (int64_t)page[u] * p.k_stride_page + (int64_t)o * p.k_stride_slot + (int64_t)h * p.k_stride_head;
          kk[u] = CachePiece<QT, KVT>::load(p.k_cache, a.vec_k, k_scale, [&](int i) {
            const int d = d0 + i;
            return kb + (int64_t)(d / p.k_x) * p.k_stride_dx + (int64_t)(d % p.k_x) * p.k_stride_d;
          });
          const int64_t vb = (int64_t)page[u] * p.v_stride_page + (int64_t)o * p.v_stride_slot + (int64_t)h * p.v_stride_head;
          vv[u] = CachePiece<QT, KVT>::load(p.v_cache, a.vec_v, v_scale, [&](int i) { return vb + (int64_t)(d0 + i) * p.v_stride_d; });
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kPagesPerWg; ++u) {
      if (pg0 + u >= a.pages_per_seq || (pg0 + u) * kRepackPage >= seq_len) break;   // (a page past the sequence is nobody's - packed: the next sequence's)
      const int64_t dst = (page_base + pg0 + u) * kRepackPage * Hk * D + ((int64_t)slot * Hk + h) * D + d0;
      *(uint4*)(a.k_dst + dst) = kk[u];
      *(uint4*)(a.v_dst + dst) = vv[u];
    }
    return;
  }

  for (int u = 0; u < kPagesPerWg; ++u) {
  const int pg = pg0 + u;
  if (pg >= a.pages_per_seq) break;
  const int j0 = pg * kRepackPage;
  if (j0 >= seq_len) break;
  const int64_t dst_page = (page_base + pg) * kRepackPage * Hk * D;
  if (u > 0) __syncthreads();                   // vt is reused
  // V of a page that comes from the cache as a whole (no new-token rows in it), d-major source
  const bool turn_v = a.v_keys_contiguous && !(use_new && j0 + kRepackPage > ctx);
  if (turn_v) {
    const int page = bt_row[j0 / p.page_size];
    const int o0 = j0 % p.page_size;
    const int64_t vb = (int64_t)page * p.v_stride_page + (int64_t)h * p.v_stride_head + o0;
    for (int idx = threadIdx.x; idx < 2 * D; idx += 256) {
      const int d = idx >> 1, k0 = (idx & 1) * 8;
      const uint4 w = CachePiece<QT, KVT>::load(p.v_cache, true, v_scale, [&](int i) { return vb + (int64_t)d * p.v_stride_d + k0 + i; });
      const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) { vt[d][k0 + 2 * i] = (uint16_t)ww[i]; vt[d][k0 + 2 * i + 1] = (uint16_t)(ww[i] >> 16); }
    }
    __syncthreads();
  }

  for (int idx = threadIdx.x; idx < pieces; idx += 256) {
    const int c = idx % chunks, slot = idx / chunks;
    const int j = j0 + slot, d0 = 8 * c;
    uint4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};   // slots past the sequence end are zero-filled
    if (j < seq_len) {
      if (use_new && j >= ctx) {
        const int64_t off = (int64_t)(q_start + j - ctx) * p.new_stride_token + (int64_t)h * p.new_stride_head + d0;
        if (a.vec_new) {
          kk = *(const uint4*)(kn + off);
          vv = *(const uint4*)(vn + off);
        } else {
          kk = gather8(kn, off, 1);
          vv = gather8(vn, off, 1);
        }
      } else {
        const int page = bt_row[j / p.page_size];
        const int o = j % p.page_size;
        const int64_t kb = (int64_t)page * p.k_stride_page + (int64_t)o * p.k_stride_slot + (int64_t)h * p.k_stride_head;
        kk = CachePiece<QT, KVT>::load(p.k_cache, a.vec_k, k_scale, [&](int i) {
          const int d = d0 + i;
          return kb + (int64_t)(d / p.k_x) * p.k_stride_dx + (int64_t)(d % p.k_x) * p.k_stride_d;
        });
        if (turn_v) {
          uint16_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = vt[d0 + i][slot];
          vv = pack8(e);
        } else {
          const int64_t vb = (int64_t)page * p.v_stride_page + (int64_t)o * p.v_stride_slot + (int64_t)h * p.v_stride_head;
          vv = CachePiece<QT, KVT>::load(p.v_cache, a.vec_v, v_scale, [&](int i) { return vb + (int64_t)(d0 + i) * p.v_stride_d; });
        }
      }
    }
    const int64_t dst = dst_page + ((int64_t)slot * Hk + h) * D + d0;
    *(uint4*)(a.k_dst + dst) = kk;
    *(uint4*)(a.v_dst + dst) = vv;
  }
  }
}

bool is_fp8(int d) { return d == MI355_FP8_E4M3 || d == MI355_FP8_E5M2; }

bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

int pages_per_seq(const mi355_attn_params& p) { return (p.max_seqlen_k + kRepackPage - 1) / kRepackPage; }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RepackLayout { size_t bt_off, k_off, v_off, total; };

// the scratch follows `head` bytes that the attention kernels on the scratch use themselves (split-KV partials)
// Self-attention from linear tensors (new_kv_all_rows: every key is one of the call's own query tokens, no context): the
// sequences' pages are packed - sequence i's start at cu_seqlens_q[i] / 16 + i, num_tokens / 16 + num_seqs pages in all, host
// arithmetic - instead of padded to the longest sequence: a skewed varlen batch (256 sequences, one of 8192 tokens) took
// 8.6 GB of scratch where its 2 x 130k keys need 0.5 GB. Block-table rows stay pages_per_seq wide.
static bool tight_scratch(const mi355_attn_params& p) { return p.new_kv_all_rows != 0 && p.k_new != nullptr; }
RepackLayout layout(const mi355_attn_params& p, size_t head) {
  const size_t grid_pages = (size_t)p.num_seqs * pages_per_seq(p);
  const size_t pages = tight_scratch(p) ? (size_t)p.num_tokens / kRepackPage + p.num_seqs + 1 : grid_pages;
  const size_t cache_bytes = pages * kRepackPage * p.num_kv_heads * p.head_size * 2;
  RepackLayout l;
  l.bt_off = align256(head);
  l.k_off = align256(l.bt_off + grid_pages * sizeof(int32_t));
  l.v_off = align256(l.k_off + cache_bytes);
  l.total = l.v_off + cache_bytes;
  return l;
}

}  // namespace

// The call as the matrix-core kernels see it: flash-layout scratch cache, identity block table, no linear source.
mi355_attn_params repacked_params(const mi355_attn_params& p, void* scratch, size_t head) {
  const RepackLayout l = layout(p, head);
  char* base = (char*)scratch;
  mi355_attn_params r = p;
  r.k_cache = base ? base + l.k_off : (const void*)(uintptr_t)256;   // size queries only look at the alignment
  r.v_cache = base ? base + l.v_off : (const void*)(uintptr_t)256;
  r.block_table = (const int32_t*)(base ? base + l.bt_off : (char*)(uintptr_t)256);
  r.block_table_stride = pages_per_seq(p);
  r.k_new = r.v_new = nullptr;
  r.new_kv_all_rows = 0;
  r.kv_dtype = p.q_dtype;                 // an fp8 cache is dequantised on the way in
  r.k_scale = r.v_scale = nullptr;
  r.page_size = kRepackPage;
  r.k_x = p.head_size;
  r.k_stride_d = r.v_stride_d = 1;
  r.k_stride_dx = 0;
  r.k_stride_head = r.v_stride_head = p.head_size;
  r.k_stride_slot = r.v_stride_slot = (int64_t)p.num_kv_heads * p.head_size;
  r.k_stride_page = r.v_stride_page = (int64_t)kRepackPage * p.num_kv_heads * p.head_size;
  return r;
}

// A LONG prefill over an fp8 flash-layout cache: dequantise the sequences' keys into the 16-bit scratch once - exactly the
// reference's (fp8 -> f32) * scale -> query type, :434-455 - and run the 64-rows-per-wave kernel on that, instead of the
// register-staged kernel that widens every tile in every Q block (670 / 896 TFLOP/s at 1 / 16 x 4096 tokens). The pass
// moves 3 bytes per cache element once, the attention does 2 q G flops per element: worth it when the sequences bring
// many query rows (avg query_len * G >= 4096: the pass is then <= 15 % of the attention's time).
static bool fp8_prefill_through_scratch(const mi355_attn_params& p) {
  static const char* const pin = lab_env("MI355_FP8_PREFILL_SCRATCH");   // A/B: 0 = never, 1 = also where the kernel reads fp8 itself
  const bool off = pin && pin[0] == '0', forced = pin && pin[0] == '1';
  if (off || !is_fp8(p.kv_dtype) || p.k_new || p.max_seqlen_q <= 1 || p.max_seqlen_k < 2048) return false;
  // Round 4: plain attention at head size 128 needs no scratch - prefill_pw_kernel's KV8 instantiations take the fp8 tiles by
  // LDS-DMA and widen them on their way into the rings. What is left for this route: the window / soft-cap / ALiBi / small-head
  // instantiations, which only read 16-bit tiles.
  if (!forced && prefill_pw_applicable(p)) return false;
  // (soft-cap, and ALiBi by itself: the 64-rows-per-wave kernel's SC / AL instantiations serve them)
  if (p.alibi_slopes && (p.softcap > 0.0f || p.sliding_window > 0)) return false;
  if (!(p.head_size == 128 || ((p.head_size == 64 || p.head_size == 80 || p.head_size == 96) && !p.alibi_slopes && p.softcap == 0.0f && (p.sliding_window <= 0 || p.head_size == 96)))) return false;   // (what prefill_pw_applicable serves)
  const int64_t G = p.num_q_heads / p.num_kv_heads;
  return (int64_t)p.num_tokens * G >= (int64_t)4096 * p.num_seqs;
}

bool repack_supported(const mi355_attn_params& p) {
  if (!(p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16) || (p.kv_dtype != p.q_dtype && !is_fp8(p.kv_dtype))) return false;
  const bool flash = p.k_x == p.head_size && p.k_stride_d == 1 && p.v_stride_d == 1;
  if (!p.k_new && flash && !fp8_prefill_through_scratch(p)) return false;   // nothing to repack: the kernels read that cache themselves
  if (p.only_decodes || p.max_seqlen_k <= 0) return false;
  if (p.head_size % 8 != 0 || p.page_size <= 0) return false;
  if (layout(p, 0).total > kRepackMaxBytes) return false;
  const mi355_attn_params r = repacked_params(p, nullptr, 0);
  // a decode-only call is repacked only when the split-KV kernel cannot read the cache itself (fp8 or 4-D v0 caches:
  // 1 + 2 + 2 bytes per element moved instead of 1, still several times faster than the shape-agnostic kernel)
  if (p.max_seqlen_q <= 1) return (p.new_kv_all_rows || !decode_supported(p)) && decode_supported(r);
  return prefill_supported(r);
}

size_t repack_scratch_bytes(const mi355_attn_params& p, size_t head) { return layout(p, head).total; }

int launch_repack(const mi355_attn_params& p, void* scratch, size_t head, bool skip_single, hipStream_t stream) {
  const RepackLayout l = layout(p, head);
  char* base = (char*)scratch;
  RepackArgs a;
  a.p = p;
  a.bt_dst = (int32_t*)(base + l.bt_off);
  a.k_dst = (uint16_t*)(base + l.k_off);
  a.v_dst = (uint16_t*)(base + l.v_off);
  a.pages_per_seq = pages_per_seq(p);
  a.tight = tight_scratch(p) ? 1 : 0;
  a.total_pages = (int)(tight_scratch(p) ? (size_t)p.num_tokens / kRepackPage + p.num_seqs + 1 : (size_t)p.num_seqs * pages_per_seq(p));
  a.skip_single = skip_single ? 1 : 0;
  // 16-byte (16-bit cache) / 8-byte (fp8 cache) loads of eight consecutive head dims
  const int64_t unit = is_fp8(p.kv_dtype) ? 8 : 16;
  auto aligned = [&](const void* ptr) { return ((uintptr_t)ptr & (uintptr_t)(unit - 1)) == 0; };
  const int64_t ks[] = {p.k_stride_page, p.k_stride_slot, p.k_stride_head, p.k_stride_dx};
  a.vec_k = p.k_stride_d == 1 && p.k_x % 8 == 0 && aligned(p.k_cache);
  for (int64_t s : ks) a.vec_k = a.vec_k && s % 8 == 0;
  const int64_t vs[] = {p.v_stride_page, p.v_stride_slot, p.v_stride_head};
  a.vec_v = p.v_stride_d == 1 && aligned(p.v_cache);
  for (int64_t s : vs) a.vec_v = a.vec_v && s % 8 == 0;
  a.vec_new = p.k_new && aligned16(p.k_new) && aligned16(p.v_new) && p.new_stride_token % 8 == 0 && p.new_stride_head % 8 == 0;
  a.v_keys_contiguous = p.v_stride_slot == 1 && p.v_stride_d % 8 == 0 && p.page_size % kRepackPage == 0 && p.head_size <= 256 && aligned(p.v_cache);
  for (int64_t s : {p.v_stride_page, p.v_stride_head}) a.v_keys_contiguous = a.v_keys_contiguous && s % 8 == 0;
  dim3 grid((a.pages_per_seq + kPagesPerWg - 1) / kPagesPerWg, p.num_seqs, p.num_kv_heads);
  const bool bf = p.q_dtype == MI355_BF16;
  if (p.kv_dtype == MI355_FP8_E4M3) {
    if (bf) hipLaunchKernelGGL((repack_kernel<bf16_t, e4m3_t>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((repack_kernel<f16_t, e4m3_t>), grid, dim3(256), 0, stream, a);
  } else if (p.kv_dtype == MI355_FP8_E5M2) {
    if (bf) hipLaunchKernelGGL((repack_kernel<bf16_t, e5m2_t>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((repack_kernel<f16_t, e5m2_t>), grid, dim3(256), 0, stream, a);
  } else {
    hipLaunchKernelGGL((repack_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, stream, a);   // 16-bit caches move as raw bits
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("repack launch failed: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
  return MI355_OK;
}

}  // namespace mi355
