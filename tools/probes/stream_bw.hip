// HBM streaming probe: what a write-only, a read-only and a copy stream sustain on this MI355X (16 B per lane, fully
// coalesced, every CU busy) - the rooflines the elementwise / first-layer kernels of profiles/rNN_hbm_kernels.txt are
// priced against (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy = read + write bytes).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_bw.hip -o tools/probes/stream_bw && tools/probes/stream_bw [MiB]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(256) void k_write(uint4* __restrict__ dst, size_t n) {
    const uint4 v = make_uint4(threadIdx.x, blockIdx.x, 3u, 4u);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = v;
}
__global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ src, size_t n, unsigned* out) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint4 v = src[i];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;          // (keeps the loads alive)
}
__global__ __launch_bounds__(256) void k_copy(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

int main(int argc, char** argv) {
    const size_t mib = argc > 1 ? (size_t)atol(argv[1]) : 1024;
    const size_t bytes = mib << 20, n = bytes / 16;
    uint4 *a, *b;
    unsigned* o;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    (void)hipMemset(b, 2, bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int grid : {2048, 8192}) {
        for (int what = 0; what < 3; ++what) {
            float best = 1e30f;
            for (int it = 0; it < 7; ++it) {
                (void)hipEventRecord(e0, 0);
                if (what == 0) k_write<<<grid, 256>>>(a, n);
                else if (what == 1) k_read<<<grid, 256>>>(a, n, o);
                else k_copy<<<grid, 256>>>(a, b, n);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                if (it > 1 && ms < best) best = ms;
            }
            const double moved = what == 2 ? 2.0 * bytes : (double)bytes;
            printf("%-5s %5zu MiB, grid %5d x 256: %8.1f us  %6.2f TB/s%s\n", what == 0 ? "write" : what == 1 ? "read" : "copy", mib, grid,
                   best * 1e3, moved / best * 1e-9, what == 2 ? "  (read + write bytes)" : "");
        }
    }
    return 0;
}
