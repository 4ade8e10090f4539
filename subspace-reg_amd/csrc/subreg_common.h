// Shared definitions of libsubreg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/subreg_hip.h"

namespace subreg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Launch-error to ABI-error translation: never throw across the C ABI.
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SUBREG_OK : -(1000 + (int)e);
}

#define SUBREG_CHECK_ARG(cond)            \
    do {                                  \
        if (!(cond)) return SUBREG_EINVAL; \
    } while (0)

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int DT = SUBREG_F32;
    static __device__ __forceinline__ float to_float(float v) { return v; }
    static __device__ __forceinline__ float from_float(float v) { return v; }
};
template <> struct ElemTraits<__bf16> {
    static constexpr int DT = SUBREG_BF16;
    static __device__ __forceinline__ float to_float(__bf16 v) { return (float)v; }
    static __device__ __forceinline__ __bf16 from_float(float v) { return (__bf16)v; }
};

__device__ __forceinline__ float lrelu(float v) { return v >= 0.f ? v : v * 0.1f; }   // nn.LeakyReLU(0.1)

}  // namespace subreg
