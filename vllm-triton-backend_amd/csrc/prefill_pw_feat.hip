// prefill_pw_kernel's soft-cap and ALiBi instantiations (a translation unit of their own: see "host side" in prefill_pw.hip).
#define PW_TU 1
#include "prefill_pw.hip"
