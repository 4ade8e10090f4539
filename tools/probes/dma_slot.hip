// Probe (round 5): what does a conditional LDS-DMA SLOT cost inside the step structure of conv_wide.hip?  One 4-wave workgroup per
// CU; a step = H1 (5 groups of 2 ds_read_b128 + 3 MFMA), MID (vmcnt wait + barrier), H2 (5 groups of [slot] + 2 reads + 3 MFMA);
// waves 0-1 stage 5 pieces per step (count = 5), waves 2-3 none (count = 0).
//   hipcc --offload-arch=gfx950 -O3 -o dma_slot tools/probes/dma_slot.hip && ./dma_slot
// FORM 0: no slot code at all; 1: unconditional DMA on waves 0-1 via a C++ branch on the role OUTSIDE the step loop (two loops);
// 2: s_cmp + s_cbranch inside the asm (conv_wide.hip's dma16_slot); 3: EXEC mask; 4: slot scalar code only (test + branch, never a DMA);
// 5: as 2 but ALL waves stage (count 5 everywhere: 20 pieces per step).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int G>
__device__ __forceinline__ void slot_branch(int count, const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_cmp_le_i32 %3, %4\n\ts_cbranch_scc1 1f\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n1:"
                 : : "v"(voff), "s"(base), "s"(lds), "s"(count), "n"(G) : "memory", "m0", "scc");
}
template <int G>
__device__ __forceinline__ void slot_exec(int count, const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_cmp_gt_i32 %3, %4\n\ts_cselect_b64 exec, -1, 0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\ts_mov_b64 exec, -1"
                 : : "v"(voff), "s"(base), "s"(lds), "s"(count), "n"(G) : "memory", "m0", "scc");
}
__device__ __forceinline__ void slot_plain(const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
template <int G>
__device__ __forceinline__ void slot_test_only(int count) {
    asm volatile("s_cmp_le_i32 %0, %1\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" : : "s"(count), "n"(G) : "memory", "scc");
}

template <int FORM>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* ticks, int steps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    f32x16 acc[15];
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    uint4 a0 = make_uint4(0x3f803f80u + lane, 0x3f803f80, 0x3f813f80, 0x3f803f82);
    uint4 f0[8], f1[8];
    for (int q = 0; q < 8; ++q) { f0[q] = a0; f1[q] = a0; }
    const unsigned voff = (unsigned)lane * 16;
    const unsigned rbase = 40960u + (unsigned)lane * 16;
    const bool stager = FORM == 5 || wid < 2;
    const int count = __builtin_amdgcn_readfirstlane((FORM == 0 || FORM == 4) ? 0 : (stager ? 5 : 0));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    auto group = [&](uint4(&c)[8], uint4(&n)[8], int g, unsigned roff) {
        n[(2 * g) % 8] = *reinterpret_cast<const uint4*>(smem + rbase + roff + g * 1024);
        n[(2 * g + 1) % 8] = *reinterpret_cast<const uint4*>(smem + rbase + roff + g * 1024 + 8192);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            acc[3 * g + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, c[j]), __builtin_bit_cast(bf16x8, c[3 + g]), acc[3 * g + j], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto body = [&](auto plain_tag) {
        constexpr bool PLAIN = decltype(plain_tag)::value;
        for (int s = 0; s < steps; ++s) {
            unsigned soff = (unsigned)(s & 63) * 32768u + (unsigned)wid * 5120u;      // 2 MB shared by every workgroup: L2 hits
            asm volatile("" : "+s"(soff));
            const unsigned lbase = lds_base + (unsigned)(s % 3) * 10240u + (unsigned)(wid & 1) * 5120u;
            const unsigned roff = (unsigned)(s & 1) * 512u;
#pragma unroll
            for (int g = 0; g < 5; ++g) group(f0, f1, g, roff);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                if (FORM == 1) { if (PLAIN) slot_plain(src, voff + soff + g * 1024, lbase + g * 1024); }
                if (FORM == 2 || FORM == 5) {
                    if (g == 0) slot_branch<0>(count, src, voff + soff, lbase);
                    if (g == 1) slot_branch<1>(count, src, voff + soff + 1024, lbase + 1024);
                    if (g == 2) slot_branch<2>(count, src, voff + soff + 2048, lbase + 2048);
                    if (g == 3) slot_branch<3>(count, src, voff + soff + 3072, lbase + 3072);
                    if (g == 4) slot_branch<4>(count, src, voff + soff + 4096, lbase + 4096);
                }
                if (FORM == 3) {
                    if (g == 0) slot_exec<0>(count, src, voff + soff, lbase);
                    if (g == 1) slot_exec<1>(count, src, voff + soff + 1024, lbase + 1024);
                    if (g == 2) slot_exec<2>(count, src, voff + soff + 2048, lbase + 2048);
                    if (g == 3) slot_exec<3>(count, src, voff + soff + 3072, lbase + 3072);
                    if (g == 4) slot_exec<4>(count, src, voff + soff + 4096, lbase + 4096);
                }
                if (FORM == 4) {
                    if (g == 0) slot_test_only<0>(count);
                    if (g == 1) slot_test_only<1>(count);
                    if (g == 2) slot_test_only<2>(count);
                    if (g == 3) slot_test_only<3>(count);
                    if (g == 4) slot_test_only<4>(count);
                }
                group(f1, f0, g, roff + 256);
            }
        }
    };
    if (FORM == 1) { if (stager) body(std::true_type{}); else body(std::false_type{}); }
    else body(std::false_type{});
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int q = 0; q < 15; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum + smem[lane];
    if (lane == 0) ticks[blockIdx.x * 4 + wid] = t1 - t0;
}

template <int FORM>
static void run(const char* name, const char* src, float* out, unsigned long long* ticks) {
    const int blocks = 256, steps = 400;
    auto kern = k<FORM>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, out, ticks, steps);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < blocks; ++b) v.push_back((double)h[b * 4] / steps);
    std::sort(v.begin(), v.end());
    printf("%-78s %7.0f cycles per step (30 MFMAs = 960)\n", name, v[v.size() / 2]);
}

int main() {
    char* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 8 << 20); hipMemset(src, 0x3f, 8 << 20);
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&ticks, 256 * 4 * sizeof(unsigned long long));
    run<0>("no slot code", src, out, ticks);
    run<4>("slot scalar code only (s_cmp + taken s_cbranch per slot)", src, out, ticks);
    run<1>("waves 0-1: unconditional DMA per slot, role branch outside the loop", src, out, ticks);
    run<2>("s_cmp + s_cbranch inside the asm; waves 0-1 stage 5, waves 2-3 skip", src, out, ticks);
    run<3>("EXEC mask inside the asm; waves 0-1 stage 5, waves 2-3 masked", src, out, ticks);
    run<5>("s_cmp + s_cbranch; ALL waves stage 5 (20 pieces per step)", src, out, ticks);
    return 0;
}
