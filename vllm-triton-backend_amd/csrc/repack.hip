// Repack path for the legacy ops (context_attention_fwd, chunked_prefill_paged_decode;
// LIB/kernels/legacy/triton_prefix_prefill.py:588-765, triton_chunked_prefill_paged_decode.py:28-117).
//
// Those ops read context keys from the vLLM v0 cache layout (K [nb, Hk, D/x, page, x], V [nb, Hk, D, page]) and the
// keys of the tokens being prefilled from linear [T, Hk, D] tensors. The matrix-core prefill kernel stages whole
// flash-layout rows through LDS-DMA; rather than a second copy of that kernel for a d-major V and a two-source key
// stream, one pass gathers every sequence's keys - context pages from the cache (any layout the ABI's strides
// describe), new rows from the linear tensors - into a flash-layout scratch cache in the caller's workspace with an
// identity block table, and the prefill kernel runs on that. The pass moves 2x the K/V bytes once; prefill does
// O(query_len) more work per key than that, so it is noise next to the attention itself.
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
  int vec_k, vec_v, vec_new;   // 16-byte loads are legal for that source
  int skip_single;             // sequences with query_len == 1 are not repacked (left out or served from the cache)
};

__device__ inline uint4 gather8(const uint16_t* base, int64_t off0, int64_t stride) {
  uint16_t e[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) e[i] = base[off0 + i * stride];
  return uint4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
               (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
}

// grid (pages_per_seq, num_seqs), 256 threads: one scratch page (16 keys x Hk x D) of one sequence per workgroup,
// moved as 16-byte pieces (8 head dims of one key and head). 16-bit elements are moved as raw bits.
__global__ __launch_bounds__(256) void repack_kernel(RepackArgs a) {
  const mi355_attn_params& p = a.p;
  const int pg = blockIdx.x, seq = blockIdx.y;
  if (threadIdx.x == 0) a.bt_dst[(int64_t)seq * a.pages_per_seq + pg] = seq * a.pages_per_seq + pg;
  const int seq_len = p.seqused_k[seq];
  const int q_start = p.cu_seqlens_q[seq], q_len = p.cu_seqlens_q[seq + 1] - q_start;
  if (q_len <= 0 || (a.skip_single && q_len == 1)) return;
  const int j0 = pg * kRepackPage;
  if (j0 >= seq_len) return;
  const int ctx = seq_len - q_len;
  const bool use_new = p.k_new != nullptr && q_len > 1;   // generic_attn.hip: same rule
  const int D = p.head_size, Hk = p.num_kv_heads, chunks = D >> 3;
  const int pieces = kRepackPage * Hk * chunks;
  const int64_t dst_page = ((int64_t)seq * a.pages_per_seq + pg) * kRepackPage * Hk * D;
  const uint16_t* kc = (const uint16_t*)p.k_cache;
  const uint16_t* vc = (const uint16_t*)p.v_cache;
  const uint16_t* kn = (const uint16_t*)p.k_new;
  const uint16_t* vn = (const uint16_t*)p.v_new;
  for (int idx = threadIdx.x; idx < pieces; idx += 256) {
    const int c = idx % chunks, h = (idx / chunks) % Hk, slot = idx / (chunks * Hk);
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
        const int page = p.block_table[(int64_t)seq * p.block_table_stride + j / p.page_size];
        const int o = j % p.page_size;
        const int64_t kb = (int64_t)page * p.k_stride_page + (int64_t)o * p.k_stride_slot + (int64_t)h * p.k_stride_head;
        const int64_t vb = (int64_t)page * p.v_stride_page + (int64_t)o * p.v_stride_slot + (int64_t)h * p.v_stride_head;
        if (a.vec_k) {
          kk = *(const uint4*)(kc + kb + (int64_t)(d0 / p.k_x) * p.k_stride_dx + (d0 % p.k_x));
        } else {
          uint16_t e[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int d = d0 + i;
            e[i] = kc[kb + (int64_t)(d / p.k_x) * p.k_stride_dx + (int64_t)(d % p.k_x) * p.k_stride_d];
          }
          kk = uint4{(uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16),
                     (uint32_t)e[4] | ((uint32_t)e[5] << 16), (uint32_t)e[6] | ((uint32_t)e[7] << 16)};
        }
        vv = a.vec_v ? *(const uint4*)(vc + vb + d0) : gather8(vc, vb + (int64_t)d0 * p.v_stride_d, p.v_stride_d);
      }
    }
    const int64_t dst = dst_page + ((int64_t)slot * Hk + h) * D + d0;
    *(uint4*)(a.k_dst + dst) = kk;
    *(uint4*)(a.v_dst + dst) = vv;
  }
}

bool aligned16(const void* ptr) { return ((uintptr_t)ptr & 15) == 0; }

int pages_per_seq(const mi355_attn_params& p) { return (p.max_seqlen_k + kRepackPage - 1) / kRepackPage; }

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct RepackLayout { size_t bt_off, k_off, v_off, total; };

// the scratch follows `head` bytes that the attention kernels on the scratch use themselves (split-KV partials)
RepackLayout layout(const mi355_attn_params& p, size_t head) {
  const size_t pages = (size_t)p.num_seqs * pages_per_seq(p);
  const size_t cache_bytes = pages * kRepackPage * p.num_kv_heads * p.head_size * 2;
  RepackLayout l;
  l.bt_off = align256(head);
  l.k_off = align256(l.bt_off + pages * sizeof(int32_t));
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
  r.page_size = kRepackPage;
  r.k_x = p.head_size;
  r.k_stride_d = r.v_stride_d = 1;
  r.k_stride_dx = 0;
  r.k_stride_head = r.v_stride_head = p.head_size;
  r.k_stride_slot = r.v_stride_slot = (int64_t)p.num_kv_heads * p.head_size;
  r.k_stride_page = r.v_stride_page = (int64_t)kRepackPage * p.num_kv_heads * p.head_size;
  return r;
}

bool repack_supported(const mi355_attn_params& p) {
  if (!(p.q_dtype == MI355_BF16 || p.q_dtype == MI355_F16) || p.kv_dtype != p.q_dtype) return false;
  const bool flash = p.k_x == p.head_size && p.k_stride_d == 1 && p.v_stride_d == 1;
  if (!p.k_new && flash) return false;                       // nothing to repack
  if (p.only_decodes || p.max_seqlen_q <= 1 || p.max_seqlen_k <= 0) return false;
  if (p.head_size % 8 != 0 || p.page_size <= 0) return false;
  if (layout(p, 0).total > kRepackMaxBytes) return false;
  return prefill_supported(repacked_params(p, nullptr, 0));
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
  a.skip_single = skip_single ? 1 : 0;
  const int64_t ks[] = {p.k_stride_page, p.k_stride_slot, p.k_stride_head, p.k_stride_dx};
  a.vec_k = p.k_stride_d == 1 && p.k_x % 8 == 0 && aligned16(p.k_cache);
  for (int64_t s : ks) a.vec_k = a.vec_k && s % 8 == 0;
  const int64_t vs[] = {p.v_stride_page, p.v_stride_slot, p.v_stride_head};
  a.vec_v = p.v_stride_d == 1 && aligned16(p.v_cache);
  for (int64_t s : vs) a.vec_v = a.vec_v && s % 8 == 0;
  a.vec_new = p.k_new && aligned16(p.k_new) && aligned16(p.v_new) && p.new_stride_token % 8 == 0 && p.new_stride_head % 8 == 0;
  dim3 grid(a.pages_per_seq, p.num_seqs);
  hipLaunchKernelGGL(repack_kernel, grid, dim3(256), 0, stream, a);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("repack launch failed: %s", hipGetErrorString(e)); return MI355_ERR_HIP; }
  return MI355_OK;
}

}  // namespace mi355
