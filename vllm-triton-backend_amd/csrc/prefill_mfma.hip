#include "common.h"
namespace mi355 {
bool prefill_supported(const mi355_attn_params&) { return false; }
int launch_prefill(const mi355_attn_params&, hipStream_t) { set_error("prefill kernel not built"); return MI355_ERR_UNSUPPORTED; }
}
