// Shared definitions of libsubreg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/subreg_hip.h"

namespace subreg {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// Launch-error to ABI-error translation: never throw across the C ABI.
inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? SUBREG_OK : -(1000 + (int)e);
}

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE attribute of a kernel: set it once per (kernel, device).  `done` is
// a per-kernel bitmask of the devices already configured (one process may drive several GPUs; host threads may race).
inline int ensure_dynamic_lds(const void* kern, size_t bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SUBREG_EHIP;
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return SUBREG_OK;
    if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return SUBREG_EHIP;
    done.fetch_or(bit, std::memory_order_release);
    return SUBREG_OK;
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

// one 1-KiB LDS-DMA piece: 64 lanes x 16 B from `base + voff` (wave-uniform 64-bit base in SGPRs, per-lane 32-bit byte
// offset), LDS image lane-linear from the wave-uniform LDS byte address `lds_addr`.
// Issued from inline asm ON PURPOSE: for the builtin form hipcc (ROCm 7.2) inserts `s_waitcnt vmcnt(0)` in front of
// the next ds_read of ANY LDS address, which drains the prefetch every step; asm DMAs are invisible to that pass, so
// the counted `s_waitcnt vmcnt(N)` + `s_barrier` at the end of each step are the only (hand-placed) waits on them.
// M0 (the DMA's LDS base) is set inside the statement and declared clobbered.  Rounds 1-3 saved and restored it around every DMA
// (four scalar instructions per piece); hipcc (ROCm 7.2) never uses M0 in any kernel of this library - every mention of m0 in the
// generated ISA is inside these statements (tools/kernel_resources.sh has the one-line check) - and with the clobber it knows the
// register does not survive, so it would re-materialise a value of its own.  SUBREG_DMA_KEEP_M0=1 restores the old form.
#ifndef SUBREG_DMA_KEEP_M0
#define SUBREG_DMA_KEEP_M0 0
#endif
__device__ __forceinline__ void dma16(const char* base_in, unsigned voff, unsigned lds_addr) {
    // the base is wave-uniform by construction; say so where the compiler's uniformity analysis cannot see it
    const unsigned long long bu = (unsigned long long)(size_t)base_in;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bu);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bu >> 32));
    const char* base = (const char*)(size_t)(((unsigned long long)bhi << 32) | blo);
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);   // wave-uniform too (M0 is a scalar register)
#if SUBREG_DMA_KEEP_M0
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_u)
        : "memory");
#else
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(base), "s"(lds_u)
        : "memory", "m0");
#endif
}

__device__ __forceinline__ float lrelu(float v) { return v >= 0.f ? v : v * 0.1f; }   // nn.LeakyReLU(0.1)

}  // namespace subreg
