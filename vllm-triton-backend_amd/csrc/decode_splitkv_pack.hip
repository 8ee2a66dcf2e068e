// decode_splitkv_kernel's PACK instantiations - multi-token decode steps, several query tokens of a sequence in a wave's
// matrix columns - as a translation unit of their own (see launch_decode_pack in decode_splitkv.hip).
#define DECODE_TU 1
#include "decode_splitkv.hip"
