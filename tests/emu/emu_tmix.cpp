// csrc/tmix_fused.hip -- kernels AND C entry points -- compiled for the host lockstep emulator: the same vrwkv_* symbols as the
// product library, computed on the CPU.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <gfx950_prims.h>
#include "../../visualrwkv_amd/csrc/tmix_fused.hip"
