// prefill_pw_kernel's fp8-cache instantiations (KV8), a translation unit of their own: see the host side of prefill_pw.hip
#define PW_TU 3
#include "prefill_pw.hip"
