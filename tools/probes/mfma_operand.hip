// Probe (round 5): does the ORDER in which a wave walks its (A fragment, B fragment) pairs change what the matrix pipe costs?  One wave per
// SIMD, every CU, 15 accumulator tiles (3 A x 5 B fragments of random bf16, operands in registers), 30 v_mfma_f32_32x32x16_bf16 per step.
// The chip is power-limited under this load (mfma_energy.hip): TFLOP/s follows the clock it can hold, i.e. energy per MFMA.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mfma_operand tools/probes/mfma_operand.hip && ./mfma_operand
// ORDER 0: B-major (B_g with A_0, A_1, A_2; then B_g+1 ...): both operands change at every group boundary  (conv_wide.hip, conv_fwd.hip)
//       1: serpentine (A_0 A_1 A_2 | A_2 A_1 A_0 | ...): exactly one operand changes between consecutive MFMAs
//       2: A-major (A_i with B_0 .. B_4)
//       3: ONE pair for all 30 MFMAs (no operand ever changes; the accumulators still do)
//       4: A fixed, B walks its five fragments
//       5: as 0 with all-zero operands (the floor)
//       6: as 0, operands with the low 4 mantissa bits cleared (3-bit mantissas)
//       7: as 0, every second K element zero
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ORDER>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int steps) {
    const int lane = threadIdx.x & 63;
    uint4 f[8];
    for (int q = 0; q < 8; ++q) {
        unsigned w[4];
        for (int e = 0; e < 4; ++e) {
            unsigned h = (unsigned)(threadIdx.x * 32 + q * 4 + e) * 2654435761u + blockIdx.x * 40503u;
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            unsigned lo = (h & 0x80ffu) | 0x3f00u, hi = ((h >> 16) & 0x80ffu) | 0x3f00u;
            if (ORDER == 6) { lo &= 0xfff0u; hi &= 0xfff0u; }
            if (ORDER == 7) hi = 0;
            w[e] = ORDER == 5 ? 0u : (lo | (hi << 16));
        }
        f[q] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    f32x16 acc[15];
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    auto mm = [&](int i, int j) {
        acc[3 * j + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[i]), __builtin_bit_cast(bf16x8, f[3 + j]), acc[3 * j + i], 0, 0, 0);
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if (ORDER == 0 || ORDER >= 5) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) mm(i, j);
            } else if (ORDER == 1) {
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int i = 0; i < 3; ++i) mm((j & 1) ? 2 - i : i, j);
            } else if (ORDER == 2) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 5; ++j) mm(i, j);
            } else if (ORDER == 3) {
#pragma unroll
                for (int e = 0; e < 15; ++e)
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[3]), acc[e], 0, 0, 0);
            } else {
#pragma unroll
                for (int e = 0; e < 15; ++e)
                    acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[0]), __builtin_bit_cast(bf16x8, f[3 + e % 5]), acc[e], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int ORDER>
static void run(const char* name, float* out, unsigned long long* ticks) {
    const int blocks = 256, steps = 20000;
    auto kern = k<ORDER>;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best_tf = 0, cycs = 0, clk = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[b * 2] / steps); ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double tf = 2.0 * 32 * 32 * 16 * 30 * 4.0 * blocks * steps / (ms * 1e-3) * 1e-12;
        if (tf > best_tf) { best_tf = tf; cycs = cyc[cyc.size() / 2]; clk = ghz[ghz.size() / 2]; }
    }
    printf("%-72s %6.0f cycles/step  %.2f GHz  %5.0f TFLOP/s (%.3f of 2500)\n", name, cycs, clk, best_tf, best_tf / 2500.0);
}

int main() {
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&ticks, 256 * 2 * sizeof(unsigned long long));
    for (int round = 0; round < 2; ++round) {
        run<0>("B-major: B_g x (A_0, A_1, A_2), the kernels' order", out, ticks);
        run<1>("serpentine: one operand changes per MFMA", out, ticks);
        run<2>("A-major: A_i x (B_0 .. B_4)", out, ticks);
        run<4>("A fixed, B walks five fragments", out, ticks);
        run<3>("one (A, B) pair for every MFMA", out, ticks);
        run<6>("B-major, 3-bit mantissas", out, ticks);
        run<7>("B-major, every second K element zero", out, ticks);
        run<5>("B-major, all-zero operands", out, ticks);
    }
    return 0;
}
