#include "common.h"
namespace mi355 {
bool decode_supported(const mi355_attn_params&) { return false; }
size_t decode_workspace_bytes(const mi355_attn_params&) { return 0; }
int launch_decode(const mi355_attn_params&, void*, size_t, hipStream_t) { set_error("decode kernel not built"); return MI355_ERR_UNSUPPORTED; }
}
