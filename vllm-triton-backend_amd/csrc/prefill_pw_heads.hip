// prefill_pw_kernel's instantiations for head sizes 64 / 80 / 96 (a translation unit of their own: see "host side" in prefill_pw.hip).
#define PW_TU 2
#include "prefill_pw.hip"
