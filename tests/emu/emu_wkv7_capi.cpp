// csrc/wkv7_capi.hip -- the WKV7 launchers (variant selection, the two-workgroups-per-head forward, argument checks) and the kernels
// behind them -- compiled whole for the host lockstep emulator.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>
#include <gfx950_prims.h>
#include "../../visualrwkv_amd/csrc/wkv7_capi.hip"
