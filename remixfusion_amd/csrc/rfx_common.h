// rfx_common.h -- shared host/device helpers for librfx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/rfx.h"

#define RFX_WAVE 64

namespace rfx {

extern thread_local int g_last_hip_error;

inline int hip_fail(hipError_t e) { g_last_hip_error = (int)e; return RFX_ERR_HIP; }

#define RFX_HIP_TRY(expr)                                   \
    do {                                                    \
        hipError_t _e = (expr);                             \
        if (_e != hipSuccess) return ::rfx::hip_fail(_e);   \
    } while (0)

#define RFX_LAUNCH_CHECK()                                  \
    do {                                                    \
        hipError_t _e = hipGetLastError();                  \
        if (_e != hipSuccess) return ::rfx::hip_fail(_e);   \
    } while (0)

inline hipStream_t as_stream(rfx_stream s) { return reinterpret_cast<hipStream_t>(s); }

// The whole library is compiled with -ffp-contract=off: a*b+c fuses only where fmaf() is
// written.  madd() marks the places where the reference's nvcc (-fmad=true) contracts.
__device__ __forceinline__ float madd(float a, float b, float c) { return fmaf(a, b, c); }

// CUDA __float2int_rn: round-half-even.
__device__ __forceinline__ int f2i_rn(float v) { return (int)rintf(v); }

}  // namespace rfx
