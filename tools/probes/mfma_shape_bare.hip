// Probe (round 5): bare MFMA loops (operands in registers, random bf16), every SIMD busy: 32x32x16 against 16x16x32 at equal FLOPs per step.
// The in-kernel clock under each shape is the measurement (MI355X_MICROARCH.md, DVFS give-back item 7).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int steps) {
    const int lane = threadIdx.x & 63;
    uint4 a[4], b[4];
    for (int q = 0; q < 4; ++q) {
        unsigned h = (unsigned)(lane * 4 + q) * 2654435761u + blockIdx.x * 40503u + threadIdx.x * 977u;
        unsigned v[8];
        for (int e = 0; e < 8; ++e) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; v[e] = (h & 0x80ff80ffu) | 0x3f003f00u; }
        a[q] = make_uint4(v[0], v[1], v[2], v[3]); b[q] = make_uint4(v[4], v[5], v[6], v[7]);
    }
    float sum = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if constexpr (SHAPE == 32) {
        f32x16 acc[8];
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 32; ++j)
                acc[j & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[(j >> 2) & 3]), acc[j & 7], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    } else {
        f32x4 acc[16];
        for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) acc[q][r] = 0.f;
        for (int s = 0; s < steps; ++s) {
#pragma unroll
            for (int j = 0; j < 64; ++j)
                acc[j & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[j & 3]), __builtin_bit_cast(bf16x8, b[(j >> 2) & 3]), acc[j & 15], 0, 0, 0);
        }
        for (int q = 0; q < 16; ++q) for (int r = 0; r < 4; ++r) sum += acc[q][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, float* out, unsigned long long* ticks) {
    const int blocks = 256, steps = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::vector<double> cyc, ghz;
        for (int b = 0; b < blocks; ++b) { cyc.push_back((double)h[b * 2] / steps); ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); }
        std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
        const double tf = 32.0 * 32768.0 * 4.0 * blocks * steps / (ms * 1e-3) * 1e-12;
        printf("%-26s %7.1f cycles per 1.05 MFLOP-per-wave step (1024 = matrix pipe), clock %.2f GHz, %.0f TFLOP/s (%.3f of 2500)\n", name,
               cyc[cyc.size() / 2], ghz[ghz.size() / 2], tf, tf / 2500);
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
