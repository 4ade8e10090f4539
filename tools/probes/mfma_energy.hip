// Probe (round 5): what the chip's clock does under the INGREDIENTS of the conv loop, one at a time.  One wave per SIMD, every CU,
// 30 v_mfma_f32_32x32x16_bf16 per step on random bf16 data (960 matrix-pipe cycles), plus per step and wave:
//   NR ds_read_b128 (operand fragments from LDS; the conv loops issue 16 per 30 MFMAs at 96x160 wave tiles, 21 at 64x160)
//   ND LDS-DMA pieces of 1 KiB from an L2-resident 2 MB region (weight staging; 3.5 per wave and step in conv_wide.hip)
// The matrix pipe is kept saturated in every variant, so cycles per step stay ~970 and the in-kernel clock (s_memtime / s_memrealtime)
// is the measurement: TFLOP/s = 256 CUs x 4 SIMDs x 30 MFMAs x 32768 FLOP / step time.  The chip is power-limited on this loop: what
// lowers the clock is what costs energy.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -o mfma_energy tools/probes/mfma_energy.hip && ./mfma_energy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma_m0(const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}

template <int NR, int ND, int ZERO>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* ticks, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    for (int o = threadIdx.x; o < 65536 / 4; o += 256) {
        unsigned h = (unsigned)o * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned lo = (h & 0x80ffu) | 0x3f00u, hi = ((h >> 16) & 0x80ffu) | 0x3f00u;
        reinterpret_cast<unsigned*>(smem)[o] = ZERO ? 0u : (lo | (hi << 16));
    }
    __syncthreads();
    const unsigned rbase = (unsigned)lane * 16;
    f32x16 acc[15];
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    uint4 f0[8], f1[8];
    for (int q = 0; q < 8; ++q) { f0[q] = *reinterpret_cast<const uint4*>(smem + rbase + q * 1024); f1[q] = *reinterpret_cast<const uint4*>(smem + rbase + 8192 + q * 1024); }
    const unsigned voff = (unsigned)lane * 16;
    auto half = [&](uint4(&c)[8], uint4(&n)[8], unsigned roff, int h, unsigned soff, unsigned lbase) {
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            // reads of this half: NR / 2, spread over the five groups, into the other set
#pragma unroll
            for (int r = 0; r < (NR / 2 + 4 - g) / 5; ++r) {
                const int idx = (g + 5 * r) % 8;
                n[idx] = *reinterpret_cast<const uint4*>(smem + rbase + roff + (g + 5 * r) * 1024);
            }
            if (h == 1 && g < ND) dma_m0(src, voff + soff + g * 1024, lbase + g * 1024);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[3 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[j]), __builtin_bit_cast(bf16x8, c[3 + g]), acc[3 * g + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
        unsigned soff = (unsigned)(s & 63) * 32768u + (unsigned)wid * 5120u;
        asm volatile("" : "+s"(soff));
        const unsigned lbase = lds_base + 40960u + (unsigned)(s % 3) * 0u + (unsigned)wid * 5120u;   // (a region the fragment reads do not touch)
        const unsigned roff = (unsigned)(s & 1) * 16384u;
        half(f0, f1, roff, 0, soff, lbase);
        if (ND > 0) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        half(f1, f0, roff + 256, 1, soff, lbase);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.f;
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum + smem[40960 + lane];
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int NR, int ND, int ZERO>
static void run(const char* name, const char* src, float* out, unsigned long long* ticks) {
    const int blocks = 256, steps = 20000;
    auto kern = k<NR, ND, ZERO>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best_tf = 0, cycs = 0, clk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, out, ticks, steps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, out, ticks, steps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[b * 2] / steps); ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double tf = 2.0 * 96 * 160 * 32 * 4.0 * blocks * steps / (ms * 1e-3) * 1e-12;
        if (tf > best_tf) { best_tf = tf; cycs = cyc[cyc.size() / 2]; clk = ghz[ghz.size() / 2]; }
    }
    printf("%-64s %6.0f cycles/step  %.2f GHz  %5.0f TFLOP/s (%.3f of 2500)\n", name, cycs, clk, best_tf, best_tf / 2500.0);
}

int main() {
    char* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 4 << 20); hipMemset(src, 0x3f, 4 << 20);
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&ticks, 256 * 2 * sizeof(unsigned long long));
    for (int round = 0; round < 2; ++round) {
        run<0, 0, 0>("MFMAs only (operands stay in registers)", src, out, ticks);
        run<8, 0, 0>("+  8 ds_read_b128 per step", src, out, ticks);
        run<16, 0, 0>("+ 16 ds_read_b128 per step (96x160 wave tile)", src, out, ticks);
        run<20, 0, 0>("+ 20 ds_read_b128 per step (~64x160 wave tile per 30 MFMAs)", src, out, ticks);
        run<16, 2, 0>("+ 16 reads + 2 LDS-DMA KiB per wave and step", src, out, ticks);
        run<16, 4, 0>("+ 16 reads + 4 LDS-DMA KiB per wave and step (conv_wide.hip)", src, out, ticks);
        run<20, 5, 0>("+ 20 reads + 5 LDS-DMA KiB (general kernel's bytes per MAC)", src, out, ticks);
        run<16, 0, 1>("16 reads, ALL-ZERO operands", src, out, ticks);
    }
    return 0;
}
