// Probe (round 5): what does one LDS-DMA piece (global_load_lds_dwordx4, 1 KiB per wave-instruction) cost a wave that runs ALONE on its
// SIMD (one 4-wave workgroup per CU, 30 back-to-back v_mfma_f32_32x32x16_bf16 per "step" = 960 matrix-pipe cycles)?
//   hipcc --offload-arch=gfx950 -O3 -o dma_issue tools/probes/dma_issue.hip && ./dma_issue
// Variants per step: ND pieces per wave (0..5), placed one per MFMA group of 6; all four waves issuing in the same group (lockstep
// behind the step's barrier) or each wave in its own group (STAG); the m0 form (s_mov m0 + s_nop per piece) or m0 written once
// per step with the piece's offset in the instruction's immediate... (not encodable for LDS-DMA: the LDS address is M0 + inst_offset,
// so pieces 4 KiB apart CAN share one M0: OFF form); waves per CU 4 or 1 (ACTIVE).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void dma_m0(const char* base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds) : "memory", "m0");
}
template <int OFF> __device__ __forceinline__ void dma_off(const char* base, unsigned voff) {     // M0 set by the caller
    asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" : : "v"(voff), "s"(base), "n"(OFF) : "memory");
}

// MODE bit 3: two ds_read_b128 per MFMA group, consumed by the NEXT group's MFMAs (the real kernel's fragment pipeline);
// MODE bit 4: with bit 3, the step's ten reads sit in groups 0-1 (five each) and the pieces in groups 2.. (reads and pieces in different groups)
// MODE bit 0: stagger (wave w issues piece k in group (k + w) % 5); bit 1: one M0 per step + immediate offsets; bit 2: only wave 0 issues and computes
template <int ND, int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* src, float* out, unsigned long long* ticks, int steps, unsigned wg_stride, unsigned step_mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    f32x16 acc[6];
    for (int q = 0; q < 6; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    uint4 a0 = make_uint4(0x3f803f80u + lane, 0x3f803f80, 0x3f813f80, 0x3f803f82), b0 = make_uint4(0x3f803f80, 0x3f803f81u + lane, 0x3f803f80, 0x3f803f80);
    const bool active = !(MODE & 4) || wid == 0;
    const unsigned voff = (unsigned)lane * 16;
    const unsigned rbase = 61440u + (unsigned)lane * 16;      // fragment reads: a region the pieces do not write
    uint4 fr[10];
    for (int q = 0; q < 10; ++q) fr[q] = a0;
    unsigned long long t0 = 0, t1 = 0;
    __syncthreads();
    t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
        // a 16 KiB weight-tile-like region per step, shared by all workgroups (L2 / L1 hits), pieces contiguous
        const unsigned soff = (unsigned)(s & step_mask) * 32768u + (unsigned)wid * 5120u + blockIdx.x * wg_stride;
        const unsigned lbase = lds_base + (unsigned)(s % 3) * 20480u + (unsigned)wid * 5120u;
        if (MODE & 2) asm volatile("s_mov_b32 m0, %0" : : "s"(lbase) : "m0");
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            if (active) {
                if (MODE & 8) {
                    if (MODE & 16) {
                        if (g < 2) {
#pragma unroll
                            for (int q = 0; q < 5; ++q) fr[5 * g + q] = *reinterpret_cast<const uint4*>(smem + rbase + ((s + q) & 1) * 1024);
                        }
                    } else {
                        fr[2 * g] = *reinterpret_cast<const uint4*>(smem + rbase + (s & 1) * 1024);
                        fr[2 * g + 1] = *reinterpret_cast<const uint4*>(smem + rbase + 2048 - (s & 1) * 1024);
                    }
                }
#pragma unroll
                for (int kk = 0; kk < ND; ++kk) {
                    const int gg = (MODE & 16) ? (kk < 3 ? kk + 2 : kk - 1) : kk;
                    const bool here = (MODE & 1) ? ((kk + wid) % 5 == g) : (gg == g);
                    if (here) {
                        if (MODE & 2) {                                    // (the immediate is added to the global AND the LDS address; 13-bit signed)
                            if (kk == 0) dma_off<0>(src, voff + soff);
                            if (kk == 1) dma_off<1024>(src, voff + soff);
                            if (kk == 2) dma_off<2048>(src, voff + soff);
                            if (kk == 3) dma_off<3072>(src, voff + soff);
                            if (kk == 4) dma_m0(src, voff + soff + 4096, lbase + 4096);
                        } else {
                            dma_m0(src, voff + soff + kk * 1024, lbase + kk * 1024);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    const uint4 fa = (MODE & 8) ? fr[(2 * g + 8 + (j & 1)) % 10] : a0;      // (a fragment read one group earlier)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, b0), acc[j], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE & 32) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(ND) : "memory");   // this step's pieces stay in flight: waited for one step later
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.f;
    for (int q = 0; q < 6; ++q) for (int r = 0; r < 16; ++r) sum += acc[q][r];
    out[blockIdx.x * 256 + threadIdx.x] = sum + smem[lane];
    if (lane == 0) ticks[blockIdx.x * 4 + wid] = t1 - t0;
}

static unsigned g_wg_stride = 0, g_step_mask = 63;
template <int ND, int MODE>
static double run(const char* src, float* out, unsigned long long* ticks, int blocks);
template <int ND, int MODE>
static double run(const char* src, float* out, unsigned long long* ticks, int blocks) {
    const int steps = 400;
    auto kern = k<ND, MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 65536, 0, src, out, ticks, steps, g_wg_stride, g_step_mask);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> v;
    for (int b = 0; b < blocks; ++b) v.push_back((double)h[b * 4] / steps);
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
}

int main() {
    const int blocks = 256;
    char* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, (size_t)1100 << 20); hipMemset(src, 0x3f, (size_t)1100 << 20);
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipMalloc(&ticks, blocks * 4 * sizeof(unsigned long long));
    printf("cycles per step of 30 MFMAs (960 matrix-pipe cycles) + ND LDS-DMA pieces per wave + vmcnt(0) + s_barrier; median over 256 workgroups\n");
    printf("%-44s %7s %7s %7s %7s %7s %7s\n", "variant", "ND=0", "1", "2", "3", "4", "5");
#define ROW(name, M) printf("%-44s %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f\n", name, run<0, M>(src, out, ticks, blocks), run<1, M>(src, out, ticks, blocks), \
    run<2, M>(src, out, ticks, blocks), run<3, M>(src, out, ticks, blocks), run<4, M>(src, out, ticks, blocks), run<5, M>(src, out, ticks, blocks))
    ROW("4 waves, lockstep groups, m0 per piece", 0);
    ROW("4 waves, staggered groups, m0 per piece", 1);
    ROW("4 waves, lockstep, one m0 + immediate offsets", 2);
    ROW("4 waves, staggered, one m0 + immediate offsets", 3);
    ROW("4 waves, lockstep, m0/piece, 2 ds_read per group", 8);
    ROW("4 waves, lockstep, one m0, 2 ds_read per group", 10);
    ROW("4 waves, reads in groups 0-1, pieces in 2-4", 24);
    ROW("pipelined (waited for one step later), 2 MB shared", 40);
    g_wg_stride = 0; g_step_mask = 2047;
    ROW("pipelined, 64 MB shared by all workgroups (MALL)", 40);
    g_wg_stride = 4u << 20; g_step_mask = 127;
    ROW("pipelined, 4 MB per workgroup (HBM)", 40);
    g_wg_stride = 0; g_step_mask = 2047;      // every workgroup the same 64 MB: beyond the XCD's L2, MALL-resident
    ROW("same, 64 MB shared by all workgroups (MALL)", 8);
    g_wg_stride = 4u << 20; g_step_mask = 127;   // 4 MB per workgroup, 1 GB in all: HBM
    ROW("same, 4 MB per workgroup (HBM)", 8);
    g_wg_stride = 0; g_step_mask = 63;
    ROW("1 wave per CU, m0 per piece", 4);
    ROW("1 wave per CU, one m0 + immediate offsets", 6);
    return 0;
}
