// Probe: issue cost of back-to-back LDS-DMA pieces (global_load_lds_dwordx4, 1 KB each) of one wave, by how the LDS destination is given:
//   mode 0: s_mov m0 before every piece (a new LDS base each);   mode 1: one s_mov m0, every piece to the same base;
//   mode 2: one s_mov m0, the pieces' instruction offsets 0 / 1024 / 2048 / 3072 (does inst_offset move the LDS side as well? where do they land?)
//   mode 3 / 4 / 5: as mode 0 with STRIDED sources - a piece is 16 rows of 64 bytes at a pitch of 128 / 320 / 1280 bytes (one 32-channel chunk of
//   64- / 160- / 640-channel pixels: sixteen half cache lines instead of eight whole ones)
//   hipcc --offload-arch=gfx950 -O3 -o dma_m0 tools/probes/dma_m0.hip && ./dma_m0
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const char* src, unsigned* landed, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) char smem[65536];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    for (int o = threadIdx.x * 16; o < 65536; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    __syncthreads();
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + wid * 16384;
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds0);
    const unsigned voff = lane * 16 + wid * 65536;
    unsigned long long t0 = 0, t1 = 0;
    for (int it = 0; it < reps; ++it) {
        if (it == 1) t0 = __builtin_amdgcn_s_memtime();
        if (MODE == 0) {
#pragma unroll
            for (int p = 0; p < 16; ++p)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff + p * 1024), "s"(src), "s"(lds_u + p * 1024) : "memory", "m0");
        } else if (MODE >= 3) {
            constexpr unsigned PITCH = MODE == 3 ? 128 : MODE == 4 ? 320 : 1280;
            const unsigned sv = (unsigned)(lane >> 2) * PITCH + (unsigned)(lane & 3) * 16 + wid * 65536 * 4;
#pragma unroll
            for (int p = 0; p < 16; ++p)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(sv + p * 16 * PITCH), "s"(src), "s"(lds_u + p * 1024) : "memory", "m0");
        } else if (MODE == 1) {
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_u) : "memory", "m0");
#pragma unroll
            for (int p = 0; p < 16; ++p) asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"(voff + p * 1024), "s"(src) : "memory");
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" :: "s"(lds_u + g * 4096) : "memory", "m0");
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:0" :: "v"(voff + g * 4096), "s"(src) : "memory");
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" :: "v"(voff + g * 4096), "s"(src) : "memory");
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" :: "v"(voff + g * 4096), "s"(src) : "memory");
                asm volatile("global_load_lds_dwordx4 %0, %1 offset:3072" :: "v"(voff + g * 4096), "s"(src) : "memory");
            }
        }
        if (it == reps - 1) t1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    if (lane == 0) ticks[blockIdx.x * 4 + wid] = t1 - t0;
    if (blockIdx.x == 0) for (int o = threadIdx.x; o < 16384; o += 256) landed[o] = reinterpret_cast<unsigned*>(smem)[o];
}
int main() {
    const int reps = 200, nb = 256;
    char* src; unsigned* landed; unsigned long long* ticks;
    hipMalloc(&src, 4 << 20); hipMalloc(&landed, 65536); hipMalloc(&ticks, nb * 4 * 8);
    std::vector<unsigned> h(1 << 20);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)i;          // dword i of the source holds i
    hipMemcpy(src, h.data(), 4 << 20, hipMemcpyHostToDevice);
    std::vector<unsigned long long> t(nb * 4);
    std::vector<unsigned> l(16384);
    for (int mode = 0; mode < 6; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, nb, 256, 0, 0, src, landed, ticks, reps);
        if (mode == 1) hipLaunchKernelGGL(k<1>, nb, 256, 0, 0, src, landed, ticks, reps);
        if (mode == 2) hipLaunchKernelGGL(k<2>, nb, 256, 0, 0, src, landed, ticks, reps);
        if (mode == 3) hipLaunchKernelGGL(k<3>, nb, 256, 0, 0, src, landed, ticks, reps);
        if (mode == 4) hipLaunchKernelGGL(k<4>, nb, 256, 0, 0, src, landed, ticks, reps);
        if (mode == 5) hipLaunchKernelGGL(k<5>, nb, 256, 0, 0, src, landed, ticks, reps);
        hipDeviceSynchronize();
        hipMemcpy(t.data(), ticks, nb * 4 * 8, hipMemcpyDeviceToHost);
        hipMemcpy(l.data(), landed, 65536, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : t) s += (double)v;
        // per piece: (reps - 1) iterations x 16 pieces, incl. the vmcnt(0) drain of every iteration but the last
        printf("mode %d: %.1f s_memtime ticks per piece (issue + drain per 16)\n", mode, s / t.size() / ((reps - 1) * 16.0));
        // wave 0: where did source dword d land?  expected LDS dword == source dword for a straight copy of 16 KB
        int ok = 0; for (int d = 0; d < 4096; ++d) ok += l[d] == (unsigned)d;
        printf("        wave 0: %d of 4096 LDS dwords hold the source dword of the same index; dword 256 holds %u, dword 1024 holds %u\n", ok, l[256], l[1024]);
    }
    return 0;
}
