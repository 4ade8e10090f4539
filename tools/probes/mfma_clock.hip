// Calibration: how many s_memtime ticks does one v_mfma_f32_32x32x16_bf16 take (one wave per SIMD, back-to-back, 4 accumulators),
// alone and with one ds_read_b128 + counted wait per MFMA; and s_memtime ticks per microsecond (s_memrealtime = 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_clock tools/probes/mfma_clock.hip && ./mfma_clock
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read16(unsigned a) { u32x4 d; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); return d; }
template <int N> __device__ __forceinline__ u32x4 lds_wait(u32x4 f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); return f; }

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) char smem[32768];
    const int lane = threadIdx.x & 63;
    for (int o = threadIdx.x * 16; o < 32768; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
    f32x16 acc[4];
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    u32x4 b0 = {0x3f803f80u + lane, 0x3f803f80, 0x3f803f80, 0x3f803f80}, a0 = b0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < reps; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j)
                acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, b0), acc[j & 3], 0, 0, 0);
        } else {
            u32x4 ring[8];
#pragma unroll
            for (int j = 0; j < 7; ++j) ring[j] = lds_read16(base + j * 1024);
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if (j + 7 < 64) ring[(j + 7) % 8] = lds_read16(base + ((j + 7) % 32) * 1024);
                u32x4 f = ring[j % 8];
                if (j + 7 < 64) f = lds_wait<7>(f); else f = lds_wait<0>(f);
                if (MODE == 1) {
                    acc[j & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, b0), acc[j & 3], 0, 0, 0);
                } else {   // one fragment, two MFMAs
                    acc[(2 * j) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, b0), acc[(2 * j) & 3], 0, 0, 0);
                    acc[(2 * j + 1) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, a0), acc[(2 * j + 1) & 3], 0, 0, 0);
                }
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int mfma_per_rep, int blocks) {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&ticks, blocks * 2 * sizeof(unsigned long long));
    const int reps = 2000;
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, ticks, reps);
    hipDeviceSynchronize();
    unsigned long long h[2];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double per = (double)h[0] / ((double)reps * mfma_per_rep), us = (double)h[1] / 100.0;
    printf("%-58s blocks %4d: %.1f s_memtime ticks per MFMA, %.1f ticks/us, %.2f ns per MFMA -> %.0f TFLOP/s per chip if all 1024 SIMDs did this\n",
           name, blocks, per, (double)h[0] / us, us * 1e3 / ((double)reps * mfma_per_rep), 32768.0 * 1024 / (us * 1e3 / ((double)reps * mfma_per_rep)) / 1e3);
    hipFree(out); hipFree(ticks);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0>("MFMA back to back (operands in registers)", 64, blocks);
        run<1>("ds_read_b128 + counted wait + 1 MFMA per fragment", 64, blocks);
        run<2>("ds_read_b128 + counted wait + 2 MFMAs per fragment", 128, blocks);
    }
    return 0;
}
