// reshape_and_cache_flash: scatter new-token K/V rows into the paged cache.
//
// Replaces torch.ops._C_cache_ops.reshape_and_cache_flash as called at
// LIB/backend/triton_attn.py:396-405 (semantics restated in the reference by
// scripts/vllm_utils.py:377-401; negative slots are padding and skipped, triton_attn.py:149-151).
// HBM-bound copy: one workgroup per token, 16-byte vector loads/stores across the Hk*D row when
// source and cache have the same 2-byte type and the row is contiguous; element-wise otherwise
// (fp32, fp8 quantising store).
#include "common.h"

namespace mi355 {

struct CacheArgs {
  mi355_cache_params p;
};

__device__ __forceinline__ int64_t slot_of(const mi355_cache_params& p, int t) {
  return p.slot_mapping ? p.slot_mapping[t] : (int64_t)p.slot_mapping_i32[t];
}

// same 16-bit type, contiguous head rows: copy 8 elements (16 B) per lane. The token's slot and its K/V rows are
// independent loads: the rows are requested BEFORE the slot is looked at, so the kernel is two memory round trips
// (loads, stores), not three - it is latency, not bandwidth, at the sizes of a serving step (a 500-token prefill step of the
// e2e proxy: 19.6 -> 19.2 us per layer for cache write + attention).
// Threads 0..127 take K chunks, 128..255 V chunks when a token's row has at most 128 chunks (Hk * D <= 1024).
__global__ __launch_bounds__(256) void cache_write_vec16_kernel(const CacheArgs a) {
  const mi355_cache_params& p = a.p;
  const int t = blockIdx.x;
  const int chunks_per_head = p.head_size / 8;
  const int n = p.num_kv_heads * chunks_per_head;
  const uint16_t* ks = (const uint16_t*)p.key + (int64_t)t * p.key_stride_token;
  const uint16_t* vs = (const uint16_t*)p.value + (int64_t)t * p.value_stride_token;
  if (n <= 128) {
    const int i = threadIdx.x & 127;
    const bool is_v = threadIdx.x >= 128;
    const int h = i / chunks_per_head, c = i % chunks_per_head;
    uint4 x = uint4{0, 0, 0, 0};
    if (i < n) x = is_v ? *(const uint4*)(vs + (int64_t)h * p.value_stride_head + c * 8) : *(const uint4*)(ks + (int64_t)h * p.key_stride_head + c * 8);
    const int64_t slot = slot_of(p, t);
    if (slot < 0 || i >= n) return;
    const int64_t page = slot / p.page_size, off = slot % p.page_size;
    if (is_v) *(uint4*)((uint16_t*)p.v_cache + page * p.v_stride_page + off * p.v_stride_slot + (int64_t)h * p.v_stride_head + c * 8) = x;
    else *(uint4*)((uint16_t*)p.k_cache + page * p.k_stride_page + off * p.k_stride_slot + (int64_t)h * p.k_stride_head + c * 8) = x;
    return;
  }
  const int64_t slot = slot_of(p, t);
  if (slot < 0) return;
  const int64_t page = slot / p.page_size;
  const int64_t off = slot % p.page_size;
  uint16_t* kd = (uint16_t*)p.k_cache + page * p.k_stride_page + off * p.k_stride_slot;
  uint16_t* vd = (uint16_t*)p.v_cache + page * p.v_stride_page + off * p.v_stride_slot;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / chunks_per_head, c = i % chunks_per_head;
    const uint4 kx = *(const uint4*)(ks + (int64_t)h * p.key_stride_head + c * 8);
    const uint4 vx = *(const uint4*)(vs + (int64_t)h * p.value_stride_head + c * 8);
    *(uint4*)(kd + (int64_t)h * p.k_stride_head + c * 8) = kx;
    *(uint4*)(vd + (int64_t)h * p.v_stride_head + c * 8) = vx;
  }
}

template <typename ST, typename CT>
__global__ __launch_bounds__(256) void cache_write_elem_kernel(const CacheArgs a) {
  const mi355_cache_params& p = a.p;
  const int t = blockIdx.x;
  const int64_t slot = slot_of(p, t);
  if (slot < 0) return;
  const int64_t page = slot / p.page_size;
  const int64_t off = slot % p.page_size;
  constexpr bool kQuant = sizeof(typename CT::storage) == 1;
  // x / scale, a true division like the reference's store (scripts/vllm_utils.py:377-401): x * (1 / scale) rounds
  // differently for scales that are no powers of two
  const float k_div = (kQuant && p.k_scale) ? p.k_scale[0] : 1.0f;
  const float v_div = (kQuant && p.v_scale) ? p.v_scale[0] : 1.0f;
  const int n = p.num_kv_heads * p.head_size;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int h = i / p.head_size, d = i % p.head_size;
    float kx = elem<ST>::load(p.key, (int64_t)t * p.key_stride_token + (int64_t)h * p.key_stride_head + d);
    float vx = elem<ST>::load(p.value, (int64_t)t * p.value_stride_token + (int64_t)h * p.value_stride_head + d);
    if (kQuant) { kx /= k_div; vx /= v_div; }
    elem<CT>::store(p.k_cache, page * p.k_stride_page + off * p.k_stride_slot + (int64_t)h * p.k_stride_head + d, kx);
    elem<CT>::store(p.v_cache, page * p.v_stride_page + off * p.v_stride_slot + (int64_t)h * p.v_stride_head + d, vx);
  }
}

// 16-bit source -> fp8 cache, contiguous head rows: one lane quantises 16 elements of K and of V (two 16-byte loads,
// one 16-byte store each; the element kernel above stores single bytes: 4096 tokens took it ~40 us)
template <typename ST, typename CT>
__global__ __launch_bounds__(256) void cache_write_fp8_vec_kernel(const CacheArgs a) {
  const mi355_cache_params& p = a.p;
  const int ppt = p.num_kv_heads * (p.head_size / 16);          // pieces per token
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int t = (int)(gi / ppt), piece = (int)(gi % ppt);
  if (t >= p.num_tokens) return;
  const int64_t slot = slot_of(p, t);
  if (slot < 0) return;
  const int64_t page = slot / p.page_size, off = slot % p.page_size;
  const int h = piece / (p.head_size / 16), c = piece % (p.head_size / 16);
  const float k_div = p.k_scale ? p.k_scale[0] : 1.0f, v_div = p.v_scale ? p.v_scale[0] : 1.0f;
  const uint16_t* ks = (const uint16_t*)p.key + (int64_t)t * p.key_stride_token + (int64_t)h * p.key_stride_head + c * 16;
  const uint16_t* vs = (const uint16_t*)p.value + (int64_t)t * p.value_stride_token + (int64_t)h * p.value_stride_head + c * 16;
  const cw_u32x4_t k0 = *(const cw_u32x4_t*)ks, k1 = *(const cw_u32x4_t*)(ks + 8), v0 = *(const cw_u32x4_t*)vs, v1 = *(const cw_u32x4_t*)(vs + 8);
  uint8_t* kd = (uint8_t*)p.k_cache + page * p.k_stride_page + off * p.k_stride_slot + (int64_t)h * p.k_stride_head + c * 16;
  uint8_t* vd = (uint8_t*)p.v_cache + page * p.v_stride_page + off * p.v_stride_slot + (int64_t)h * p.v_stride_head + c * 16;
  *(cw_u32x4_t*)kd = quantise_fp8x16<ST, CT>(k0, k1, k_div);
  *(cw_u32x4_t*)vd = quantise_fp8x16<ST, CT>(v0, v1, v_div);
}

template <typename ST>
static int launch_src(const mi355_cache_params& p, hipStream_t stream) {
  CacheArgs a{p};
  dim3 grid(p.num_tokens), block(256);
  if (p.cache_dtype == p.src_dtype) {
    hipLaunchKernelGGL((cache_write_elem_kernel<ST, ST>), grid, block, 0, stream, a);
  } else if (p.cache_dtype == MI355_FP8_E4M3) {
    hipLaunchKernelGGL((cache_write_elem_kernel<ST, e4m3_t>), grid, block, 0, stream, a);
  } else if (p.cache_dtype == MI355_FP8_E5M2) {
    hipLaunchKernelGGL((cache_write_elem_kernel<ST, e5m2_t>), grid, block, 0, stream, a);
  } else {
    set_error("reshape_and_cache_flash: cache dtype %d with source dtype %d is not supported", p.cache_dtype, p.src_dtype);
    return MI355_ERR_UNSUPPORTED;
  }
  return check_hip(hipGetLastError(), "cache_write launch");
}

static bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

int launch_cache_write(const mi355_cache_params& p, hipStream_t stream) {
  if (p.num_tokens == 0) return MI355_OK;
  const bool two_byte = p.src_dtype == MI355_F16 || p.src_dtype == MI355_BF16;
  const bool vec_ok = two_byte && p.cache_dtype == p.src_dtype && p.head_size % 8 == 0 &&
                      aligned16(p.key) && aligned16(p.value) && aligned16(p.k_cache) && aligned16(p.v_cache) &&
                      p.key_stride_token % 8 == 0 && p.key_stride_head % 8 == 0 &&
                      p.value_stride_token % 8 == 0 && p.value_stride_head % 8 == 0 &&
                      p.k_stride_page % 8 == 0 && p.k_stride_slot % 8 == 0 && p.k_stride_head % 8 == 0 &&
                      p.v_stride_page % 8 == 0 && p.v_stride_slot % 8 == 0 && p.v_stride_head % 8 == 0;
  if (vec_ok) {
    CacheArgs a{p};
    hipLaunchKernelGGL(cache_write_vec16_kernel, dim3(p.num_tokens), dim3(256), 0, stream, a);
    return check_hip(hipGetLastError(), "cache_write_vec16 launch");
  }
  const bool fp8_cache = p.cache_dtype == MI355_FP8_E4M3 || p.cache_dtype == MI355_FP8_E5M2;
  const bool fp8_vec_ok = two_byte && fp8_cache && p.head_size % 16 == 0 &&
                          aligned16(p.key) && aligned16(p.value) && aligned16(p.k_cache) && aligned16(p.v_cache) &&
                          p.key_stride_token % 8 == 0 && p.key_stride_head % 8 == 0 &&
                          p.value_stride_token % 8 == 0 && p.value_stride_head % 8 == 0 &&
                          p.k_stride_page % 16 == 0 && p.k_stride_slot % 16 == 0 && p.k_stride_head % 16 == 0 &&
                          p.v_stride_page % 16 == 0 && p.v_stride_slot % 16 == 0 && p.v_stride_head % 16 == 0;
  if (fp8_vec_ok) {
    CacheArgs a{p};
    const int64_t pieces = (int64_t)p.num_tokens * p.num_kv_heads * (p.head_size / 16);
    const dim3 grid((unsigned)((pieces + 255) / 256)), block(256);
    const bool e4 = p.cache_dtype == MI355_FP8_E4M3;
    if (p.src_dtype == MI355_BF16) {
      if (e4) hipLaunchKernelGGL((cache_write_fp8_vec_kernel<bf16_t, e4m3_t>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((cache_write_fp8_vec_kernel<bf16_t, e5m2_t>), grid, block, 0, stream, a);
    } else {
      if (e4) hipLaunchKernelGGL((cache_write_fp8_vec_kernel<f16_t, e4m3_t>), grid, block, 0, stream, a);
      else hipLaunchKernelGGL((cache_write_fp8_vec_kernel<f16_t, e5m2_t>), grid, block, 0, stream, a);
    }
    return check_hip(hipGetLastError(), "cache_write_fp8_vec launch");
  }
  switch (p.src_dtype) {
    case MI355_F32: return launch_src<f32_t>(p, stream);
    case MI355_F16: return launch_src<f16_t>(p, stream);
    case MI355_BF16: return launch_src<bf16_t>(p, stream);
    default:
      set_error("reshape_and_cache_flash: source dtype %d is not supported", p.src_dtype);
      return MI355_ERR_UNSUPPORTED;
  }
}

}  // namespace mi355
