// CPU-only check of the kernels of kanzi-cpp_amd/csrc/sbrt.hip (transform id 64) against the oracle, forward and inverse; see xf_harness.hpp.
#define KNZ_EMU 1
#include "hip/hip_runtime.h"
#include "../../kanzi-cpp_amd/csrc/sbrt.hip"
#define XF_TTYPE 64
#define XF_FWD(st) launch_sbrt_forward(nullptr,st,3)
#define XF_INV(st) launch_sbrt_inverse(nullptr,st,3)
#define XF_SCRATCH_U32(nb, ml) ((size_t)0)
#include "xf_harness.hpp"
