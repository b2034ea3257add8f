// csrc/decode_fused.hip -- kernels and C entry points -- compiled for the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <gfx950_prims.h>
#include "../../visualrwkv_amd/csrc/decode_fused.hip"
