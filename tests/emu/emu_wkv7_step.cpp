// csrc/wkv7_step.hip (single-token WKV7 step with carried state) compiled whole for the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <gfx950_prims.h>
#include "../../visualrwkv_amd/csrc/wkv7_step.hip"
