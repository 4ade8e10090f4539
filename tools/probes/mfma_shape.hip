// Probe (round 5): v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 in the loop shape of conv_wide.hip - one wave per SIMD, a
// 96 x 160 output tile per wave (240 accumulator registers either way), 16 ds_read_b128 per K = 32 step, random bf16 operands in LDS,
// every CU busy.  The two shapes cost the same matrix-pipe cycles per FLOP; what the chip's clock does under each is the question
// (MI355X_MICROARCH.md, DVFS give-back item 7).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_shape tools/probes/mfma_shape.hip && ./mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    // random bf16 in [-2, 2): sign, exponent 126..127, 7 mantissa bits
    for (int o = threadIdx.x; o < 65536 / 4; o += 256) {
        unsigned h = (unsigned)o * 2654435761u + blockIdx.x * 40503u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned lo = (h & 0x80ffu) | 0x3f00u, hi = ((h >> 16) & 0x80ffu) | 0x3f00u;
        reinterpret_cast<unsigned*>(smem)[o] = lo | (hi << 16);
    }
    __syncthreads();
    const unsigned rbase = (unsigned)lane * 16;
    unsigned long long t0, t1, r0, r1;
    float sum = 0.f;
    if constexpr (SHAPE == 32) {
        f32x16 acc[15];
        for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        uint4 f0[8], f1[8];
        for (int q = 0; q < 8; ++q) { f0[q] = *reinterpret_cast<const uint4*>(smem + rbase + q * 1024); f1[q] = f0[q]; }
        auto group = [&](uint4(&c)[8], uint4(&n)[8], int g, unsigned roff) {
            n[(2 * g) % 8] = *reinterpret_cast<const uint4*>(smem + rbase + roff + g * 2048);
            if (g < 3) n[(2 * g + 1) % 8] = *reinterpret_cast<const uint4*>(smem + rbase + roff + g * 2048 + 1024);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                acc[3 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[j]), __builtin_bit_cast(bf16x8, c[3 + g]), acc[3 * g + j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        for (int s = 0; s < steps; ++s) {
            const unsigned roff = (unsigned)(s & 3) * 12288u;
#pragma unroll
            for (int g = 0; g < 5; ++g) group(f0, f1, g, roff);
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int g = 0; g < 5; ++g) group(f1, f0, g, roff + 256);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    } else {
        // 16x16x32: 6 x 10 tiles; a step = all six A fragments (held), B fragments in two halves of five
        f32x4 acc[6][10];
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 10; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        uint4 fa[2][6], fb[2][5];
        for (int q = 0; q < 6; ++q) { fa[0][q] = *reinterpret_cast<const uint4*>(smem + rbase + q * 1024); fa[1][q] = fa[0][q]; }
        for (int q = 0; q < 5; ++q) { fb[0][q] = *reinterpret_cast<const uint4*>(smem + rbase + 8192 + q * 1024); fb[1][q] = fb[0][q]; }
        t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime();
        auto step16 = [&](uint4(&ac)[6], uint4(&an)[6], unsigned roff) {
            // H1: tiles j = 0..4 on (ac, fb[0]) | reads: fb[1] (five) + three of the next step's A
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                fb[1][g] = *reinterpret_cast<const uint4*>(smem + rbase + roff + g * 1024);
                if (g < 3) an[g] = *reinterpret_cast<const uint4*>(smem + rbase + roff + 6144 + g * 1024);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    acc[i][g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ac[i]), __builtin_bit_cast(bf16x8, fb[0][g]), acc[i][g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                fb[0][g] = *reinterpret_cast<const uint4*>(smem + rbase + roff + 256 + g * 1024);
                if (g < 3) an[3 + g] = *reinterpret_cast<const uint4*>(smem + rbase + roff + 9216 + g * 1024);
#pragma unroll
                for (int i = 0; i < 6; ++i)
                    acc[i][5 + g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, ac[i]), __builtin_bit_cast(bf16x8, fb[1][g]), acc[i][5 + g], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int s = 0; s < steps; s += 2) {
            const unsigned roff = (unsigned)(s & 2) * 12288u;
            step16(fa[0], fa[1], roff);
            step16(fa[1], fa[0], roff + 12288u);
        }
        t1 = __builtin_amdgcn_s_memtime(); r1 = __builtin_amdgcn_s_memrealtime();
        for (int i = 0; i < 6; ++i) for (int j = 0; j < 10; ++j) for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, float* out, unsigned long long* ticks) {
    const int blocks = 256, steps = 20000;
    auto kern = k<SHAPE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, ticks, steps);      // warm-up: lets the clock settle under this load
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, out, ticks, steps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[b * 2] / steps); ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double flop = 2.0 * 96 * 160 * 32 * 4.0 * blocks * steps;
        printf("%-26s %7.1f cycles per K=32 step (960 = matrix pipe), in-kernel clock %.2f GHz, kernel %.2f ms -> %.0f TFLOP/s for the chip\n", name,
               cyc[cyc.size() / 2], ghz[ghz.size() / 2], ms, flop / (ms * 1e-3) * 1e-12);
    }
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&ticks, 256 * 2 * sizeof(unsigned long long));
    for (int round = 0; round < 2; ++round) {
        run<32>("v_mfma_f32_32x32x16_bf16", out, ticks);
        run<16>("v_mfma_f32_16x16x32_bf16", out, ticks);
    }
    return 0;
}
