// Device health probe: one trivial kernel + copy back, timed on the host.  Used by tools/round_end_sequence.sh between the
// steps of the driver's round-end sequence (pytest -m gpu, smoke(), bench.py) to show that the GPU still answers promptly.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/health.hip -o tools/probes/health
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>

__global__ void fill(int* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i ^ 0x5a5a;
}

int main() {
    auto t0 = std::chrono::steady_clock::now();
    const int n = 1 << 20;
    int* d = nullptr;
    if (hipMalloc(&d, n * sizeof(int)) != hipSuccess) { printf("health: hipMalloc FAILED\n"); return 2; }
    fill<<<n / 256, 256>>>(d, n);
    int h[4] = {0, 0, 0, 0};
    if (hipMemcpy(h, d + n - 4, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("health: copy FAILED\n"); return 3; }
    size_t fr = 0, tot = 0;
    (void)hipMemGetInfo(&fr, &tot);
    (void)hipFree(d);
    auto t1 = std::chrono::steady_clock::now();
    bool ok = h[3] == ((n - 1) ^ 0x5a5a);
    printf("health: %s  %.3f s  free %.1f / %.1f GiB\n", ok ? "OK" : "WRONG", std::chrono::duration<double>(t1 - t0).count(),
           fr / 1073741824.0, tot / 1073741824.0);
    return ok ? 0 : 1;
}
