// TEST-ONLY stand-in for <hip/hip_runtime.h> (see hip_runtime_api.h next to it): lets csrc/sm_device.h - the REAL hysteresis
// state machine the kernels run - compile as host code for the sanitizer harness.
#pragma once
#include "hip_runtime_api.h"
#include <algorithm>
#define __device__
#define __forceinline__ inline
using std::min;
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
