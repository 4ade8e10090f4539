// Calibration: cycles per v_mfma_f32_32x32x16_bf16 (one wave per SIMD, fragment from LDS through an 8-deep ring, 2 MFMAs per
// fragment) with N filler instructions of one kind per FRAGMENT: which fillers are free, which stretch the MFMA stream.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_fillers tools/probes/mfma_fillers.hip && ./mfma_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read16(unsigned a) { u32x4 d; asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(a)); return d; }
template <int N> __device__ __forceinline__ u32x4 lds_wait(u32x4 f) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "n"(N)); return f; }

// KIND 0: none; 1: v_mul_f32 (dependent chain of its own); 3: ds_write_b16; 4: v_accvgpr_read of a finished accumulator;
// 5: v_cvt_pk_bf16_f32; 6: s_mov exec pair (exec write + restore); 7: global_store_dwordx4 (1 KB contiguous) once per 8 fragments;
// 8: LDS-DMA (global_load_lds_dwordx4, 1 KB, M0 saved / set / restored, exec set / restored) once per 4 fragments, N = 1: every
//    DMA a new 1 KB of a large buffer (HBM stream), N = 2: always the same 1 KB (L2 hit), N = 3 / 4: strided pieces (16 rows x 64 bytes at a pitch of 128 / 320 bytes), N = 5: strided, always the same 2 KB;
template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* ticks, int reps) {
    __shared__ __attribute__((aligned(16))) char smem[40960];
    const int lane = threadIdx.x & 63;
    for (int o = threadIdx.x * 16; o < 40960; o += 256 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0x3f803f80, 0x3f803f80, 0x3f803f80, 0x3f803f80);
    __syncthreads();
    const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + lane * 16;
    const unsigned wbase = base + 32768 + (threadIdx.x >> 6) * 1024;
    f32x16 acc[2], done;
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) { acc[q][r] = 0.f; done[r] = (float)r; }
    asm volatile("" : "+a"(done));
    u32x4 b0 = {0x3f803f80u + lane, 0x3f803f80, 0x3f803f80, 0x3f803f80}, a0 = b0;
    float fv = 1.0f + lane;
    unsigned sv = 1;
    uint4 st = make_uint4(lane, 1, 2, 3);
    if (KIND == 9) asm volatile("s_mov_b32 m0, %0" :: "s"((unsigned)__builtin_amdgcn_readfirstlane((int)(wbase - lane * 16))));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < reps; ++it) {
        u32x4 ring[8];
#pragma unroll
        for (int j = 0; j < 7; ++j) ring[j] = lds_read16(base + j * 1024);
#pragma unroll
        for (int j = 0; j < 36; ++j) {
            if (j + 7 < 36) ring[(j + 7) % 8] = lds_read16(base + ((j + 7) % 32) * 1024);
#pragma unroll
            for (int n = 0; n < N; ++n) {
                if (KIND == 1) asm volatile("v_mul_f32 %0, 0x3f800000, %0" : "+v"(fv));
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv));
                if (KIND == 3) asm volatile("ds_write_b16 %0, %1" :: "v"(wbase + n * 2), "v"(fv) : "memory");
                if (KIND == 4) { float t; asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t) : "a"(done[(j + n) & 15])); fv += t; }
                if (KIND == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %0" : "+v"(fv));
                if (KIND == 6) asm volatile("s_mov_b64 exec, -1\n\ts_mov_b64 exec, -1" ::: "memory");
            }
            if ((KIND == 8 || KIND == 9) && (j & 3) == 2) {
                const char* src = reinterpret_cast<const char*>(out) + ((N == 2 || N == 5) ? 0 : ((size_t)(((blockIdx.x * 4 + (threadIdx.x >> 6)) * reps + it) * 9 + (j >> 2)) * (N == 4 ? 5120 : N == 3 ? 2048 : 1024)) % (512u << 20));
                const unsigned long long bu = (unsigned long long)(size_t)src;
                const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bu), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bu >> 32));
                const unsigned long long bs = ((unsigned long long)hi << 32) | lo;
                const unsigned ldsd = (unsigned)__builtin_amdgcn_readfirstlane((int)(wbase - lane * 16 + 4096 * 0));
                if (KIND == 8) {
                    unsigned keep;
                    // N = 3 / 4: the piece is 16 rows of 64 bytes at a row pitch of 128 / 320 bytes (a 32-channel chunk of 64- / 160-channel pixels)
                    const unsigned voff = (N == 3 || N == 5) ? (unsigned)(lane >> 2) * 128u + (unsigned)(lane & 3) * 16u
                                        : N == 4 ? (unsigned)(lane >> 2) * 320u + (unsigned)(lane & 3) * 16u : (unsigned)lane * 16u;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 exec, -1\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0\n\ts_mov_b64 exec, -1"
                                 : "=&s"(keep) : "s"(ldsd), "v"(voff), "s"(bs) : "memory");
                } else {
                    asm volatile("global_load_lds_dwordx4 %0, %1" :: "v"((unsigned)lane * 16u), "s"(bs) : "memory");
                }
            }
            if (KIND == 7 && (j & 7) == 0) *reinterpret_cast<uint4*>(reinterpret_cast<char*>(out) + (size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (j >> 3)) * 1024 + lane * 16) = st;
            u32x4 f = ring[j % 8];
            if (j + 7 < 36) f = lds_wait<7>(f); else f = lds_wait<0>(f);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, b0), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, a0), acc[1], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = fv + (float)sv;
    for (int q = 0; q < 2; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x + (140u << 20)] = s;
    if (threadIdx.x == 0) { ticks[blockIdx.x * 2] = t1 - t0; ticks[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int KIND, int N>
static void run(const char* name) {
    const int blocks = 256, reps = 500;
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, (size_t)(600u << 20));
    (void)hipMalloc(&ticks, blocks * 2 * sizeof(unsigned long long));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<KIND, N>), dim3(blocks), dim3(256), 0, 0, out, ticks, reps);
    (void)hipDeviceSynchronize();
    unsigned long long h[2];
    (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    printf("%-34s x %d per fragment (2 MFMAs): %5.1f cycles per MFMA\n", name, N, (double)h[0] / ((double)reps * 72)); fflush(stdout);
    (void)hipFree(out); (void)hipFree(ticks);
}

int main() {
    run<0, 0>("no filler");
    run<1, 2>("v_mul_f32"); run<1, 4>("v_mul_f32"); run<1, 6>("v_mul_f32"); run<1, 8>("v_mul_f32"); run<1, 12>("v_mul_f32");
    run<3, 1>("ds_write_b16"); run<3, 2>("ds_write_b16"); run<3, 4>("ds_write_b16");
    run<4, 2>("v_accvgpr_read_b32 + v_add"); run<4, 4>("v_accvgpr_read_b32 + v_add");
    run<5, 2>("v_cvt_pk_bf16_f32"); run<5, 4>("v_cvt_pk_bf16_f32");
    run<6, 1>("exec write pair"); run<6, 2>("exec write pair");
    run<7, 1>("1 KB global store per 8 fragments");
    run<8, 1>("LDS-DMA 1 KB / 4 fragments, stream"); run<8, 2>("LDS-DMA 1 KB / 4 fragments, same KB"); run<8, 3>("LDS-DMA 16 rows x 64 B, pitch 128 B"); run<8, 4>("LDS-DMA 16 rows x 64 B, pitch 320 B"); run<8, 5>("... pitch 128 B, same 2 KB (cache hit)");
    return 0;
}
