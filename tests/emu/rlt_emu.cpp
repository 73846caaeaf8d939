// CPU-only check of the kernels of kanzi-cpp_amd/csrc/rlt.hip (transform id 5) against the oracle, forward and inverse; see xf_harness.hpp.
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/rlt.hip"
#define XF_TTYPE 5
#define XF_FWD(st) launch_rlt_forward(nullptr,st)
#define XF_INV(st) launch_rlt_inverse(nullptr,st)
#define XF_SCRATCH_U32(nb, ml) ((size_t)0)
#include "xf_harness.hpp"
