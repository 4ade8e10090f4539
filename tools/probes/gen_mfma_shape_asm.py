#!/usr/bin/env python3
"""Generator of tools/probes/mfma_shape_asm.hip (round 6).

The round-5 probe (mfma_shape_bare.hip) ran v_mfma_f32_16x16x32_bf16 at 73 % matrix-pipe duty: hipcc un-ties the 4-register
accumulators of that shape (v_accvgpr_mov + s_nop between the MFMAs).  Here every MFMA is inline asm whose C and D are the SAME
AGPR range, the whole loop is one asm statement with hard-coded registers, so nothing can be inserted between the MFMAs.

Both arms own the same 64 x 160 output tile per wave (160 accumulator AGPRs) and one loop step is K = 32 of it:
  32x32x16: 2 x 5 tiles x 2 k-halves = 20 MFMAs of 32 cycles = 640 matrix-pipe cycles, fragments: (2 + 5) x 2 of 16 B per lane
  16x16x32: 4 x 10 tiles           = 40 MFMAs of 16 cycles = 640 matrix-pipe cycles, fragments: 4 + 10 of 16 B per lane
so both read the same 14 fragments (56 VGPRs) per step.  Variant LDS = 0 loads the fragments once; LDS = 1 re-reads all 14 by
ds_read_b128 every step into the other register set (double-buffered, spread evenly between the MFMAs, one lgkmcnt(0) per step:
21 reads per 960 pipe cycles - the conv loops issue 16-20).
"""
import sys

A0, A1 = 100, 160          # VGPR bases of fragment set 0 / 1 (14 x 4 registers each)
NFRAG = 14


def frag(setbase, i):
    return f"v[{setbase + 4 * i}:{setbase + 4 * i + 3}]"


def mfmas(shape, setbase):
    """MFMA list of one K = 32 step on fragment set at setbase, serpentine-free plain order."""
    out = []
    if shape == 32:
        # fragments: A(i, kh) = i * 2 + kh (i < 2), B(j, kh) = 4 + j * 2 + kh (j < 5); acc tile (i, j) = a[(i * 5 + j) * 16 ...]
        for kh in range(2):
            for i in range(2):
                for j in range(5):
                    t = (i * 5 + j) * 16
                    out.append(f"v_mfma_f32_32x32x16_bf16 a[{t}:{t + 15}], {frag(setbase, i * 2 + kh)}, {frag(setbase, 4 + j * 2 + kh)}, a[{t}:{t + 15}]")
    else:
        # fragments: A(i) = i (i < 4), B(j) = 4 + j (j < 10); acc tile (i, j) = a[(i * 10 + j) * 4 ...]
        for i in range(4):
            for j in range(10):
                t = (i * 10 + j) * 4
                out.append(f"v_mfma_f32_16x16x32_bf16 a[{t}:{t + 3}], {frag(setbase, i)}, {frag(setbase, 4 + j)}, a[{t}:{t + 3}]")
    return out


def step(shape, lds, use, fill):
    ms = mfmas(shape, use)
    lines = []
    if lds:
        every = len(ms) // NFRAG           # 20 // 14 = 1, 40 // 14 = 2
        nxt = 0
        for n, m in enumerate(ms):
            lines.append(m)
            if nxt < NFRAG and (n % every == every - 1 or every == 1):
                lines.append(f"ds_read_b128 {frag(fill, nxt)}, %[addr] offset:{nxt * 1024}")
                nxt += 1
        while nxt < NFRAG:
            lines.append(f"ds_read_b128 {frag(fill, nxt)}, %[addr] offset:{nxt * 1024}")
            nxt += 1
        lines.append("s_waitcnt lgkmcnt(0)")
    else:
        lines += ms
    return lines


def kernel(shape, lds):
    name = f"k{shape}_{'lds' if lds else 'reg'}"
    body = []
    for i in range(NFRAG):
        body.append(f"ds_read_b128 {frag(A0, i)}, %[addr] offset:{i * 1024}")
        body.append(f"ds_read_b128 {frag(A1, i)}, %[addr] offset:{i * 1024}")
    for r in range(160):
        body.append(f"v_accvgpr_write_b32 a{r}, 0")
    body.append("s_waitcnt lgkmcnt(0)")
    body.append("s_memtime %[t0]")
    body.append("s_memrealtime %[r0]")
    body.append("s_waitcnt lgkmcnt(0)")
    body.append("L_loop_%=:")
    body += step(shape, lds, A0, A1)
    body += step(shape, lds, A1, A0)
    body.append("s_sub_u32 %[cnt], %[cnt], 1")
    body.append("s_cmp_lg_u32 %[cnt], 0")
    body.append("s_cbranch_scc1 L_loop_%=")
    body.append("s_nop 7")
    body.append("s_nop 7")
    body.append("s_nop 7")
    body.append("s_memtime %[t1]")
    body.append("s_memrealtime %[r1]")
    body.append("v_mov_b32 %[sum], 0")
    for r in range(160):
        body.append(f"v_accvgpr_read_b32 %[tmp], a{r}")
        body.append("s_nop 0")
        body.append("v_add_f32 %[sum], %[sum], %[tmp]")
    body.append("s_waitcnt lgkmcnt(0)")
    text = "\n".join(f'        "{l}\\n"' for l in body)
    clob = ", ".join([f'"a{r}"' for r in range(160)] + [f'"v{r}"' for r in range(A0, A1 + 4 * NFRAG)] + ['"scc"', '"memory"'])
    return f"""
__global__ __launch_bounds__(256, 1) void {name}(float* out, unsigned long long* ticks, int steps) {{
    __shared__ uint4 img[4][{NFRAG}][64];
    fill_image(&img[0][0][0]);
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)&img[threadIdx.x >> 6][0][threadIdx.x & 63];
    unsigned long long t0, t1, r0, r1;
    float sum, tmp;
    int cnt = steps;
    asm volatile(
{text}
        : [t0] "=&s"(t0), [t1] "=&s"(t1), [r0] "=&s"(r0), [r1] "=&s"(r1), [sum] "=&v"(sum), [tmp] "=&v"(tmp), [cnt] "+s"(cnt)
        : [addr] "v"(addr)
        : {clob});
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0) {{ ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2] = t1 - t0; ticks[(blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + 1] = r1 - r0; }}
}}
"""


HEAD = r"""// GENERATED by tools/probes/gen_mfma_shape_asm.py - do not edit.  Probe (round 6): inline-asm MFMA loops with C/D tied in AGPRs,
// v_mfma_f32_32x32x16_bf16 against v_mfma_f32_16x16x32_bf16 on the same 64 x 160 output tile per wave, random bf16, one wave per SIMD on
// every CU, fragments in registers (reg) or re-read from LDS every step (lds).  Reports cycles per K = 32 step (640 = matrix pipe),
// the in-kernel clock (s_memtime / s_memrealtime) and TFLOP/s by HIP events, arms interleaved in one process.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

__device__ int g_mode = 0;

__device__ __forceinline__ void fill_image(uint4* img) {
    // random bf16: random sign, exponent 0x3f, 7 random mantissa bits (the distribution of mfma_shape_bare.hip), or N(0,1)-like / zeros
    for (int i = threadIdx.x; i < 4 * 14 * 64; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        unsigned v[4];
        for (int e = 0; e < 4; ++e) {
            h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            if (g_mode == 0) v[e] = (h & 0x80ff80ffu) | 0x3f003f00u;
            else if (g_mode == 1) {   // wide: exponents 0x3c..0x3f (|x| in [0.125, 2)), random sign and mantissa
                const unsigned e0 = 0x3c00u + ((h >> 3) & 0x180u) * 2u, e1 = 0x3c00u + ((h >> 19) & 0x180u) * 2u;
                v[e] = (h & 0x807f807fu) | e0 | (e1 << 16);
            } else v[e] = 0u;
        }
        img[i] = make_uint4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
}
"""

TAIL = r"""
typedef void (*kern_t)(float*, unsigned long long*, int);
struct Arm { const char* name; kern_t k; };

static void run(const Arm& arm, float* out, unsigned long long* ticks, double seconds, const char* data) {
    const int blocks = 256, steps = 8000;            // 2 K-steps per loop trip: 16000 steps of 640 pipe cycles ~ 4.5-6 ms per launch
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // warm: back-to-back launches for `seconds`, then time the last quarter
    const int warm = (int)(seconds / 0.006) + 1, timed = warm / 3 + 1;
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL(arm.k, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
    hipEventRecord(e0);
    for (int i = 0; i < timed; ++i) hipLaunchKernelGGL(arm.k, dim3(blocks), dim3(256), 0, 0, out, ticks, steps);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= timed;
    std::vector<unsigned long long> h(blocks * 4 * 2);
    hipMemcpy(h.data(), ticks, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> cyc, ghz;
    for (int b = 0; b < blocks * 4; ++b) { cyc.push_back((double)h[b * 2] / (2.0 * steps)); ghz.push_back((double)h[b * 2] / (double)h[b * 2 + 1] * 0.1); }
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double flop = 2.0 * 64 * 160 * 32 * 2.0 * steps * 4.0 * blocks;
    const double tf = flop / (ms * 1e-3) * 1e-12;
    printf("%-8s %-6s %7.1f cycles per K=32 step (640 = matrix pipe, duty %5.1f %%), clock %.3f GHz, %8.3f ms per launch, %6.0f TFLOP/s (%.3f of 2500)\n",
           arm.name, data, cyc[cyc.size() / 2], 64000.0 / cyc[cyc.size() / 2], ghz[ghz.size() / 2], ms, tf, tf / 2500);
    fflush(stdout);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const int rounds = argc > 2 ? atoi(argv[2]) : 3;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, 256 * 256 * sizeof(float));
    hipMalloc(&ticks, 256 * 4 * 2 * sizeof(unsigned long long));
    const Arm arms[4] = {{"32x32x16 reg", k32_reg}, {"16x16x32 reg", k16_reg}, {"32x32x16 lds", k32_lds}, {"16x16x32 lds", k16_lds}};
    const char* names[3] = {"narrow", "wide", "zeros"};
    for (int mode = 0; mode < 3; ++mode) {
        hipMemcpyToSymbol(HIP_SYMBOL(g_mode), &mode, sizeof(int));
        for (int round = 0; round < (mode == 2 ? 1 : rounds); ++round)
            for (int a = 0; a < 4; ++a) run(arms[a], out, ticks, mode == 2 ? 1.0 : seconds, names[mode]);
    }
    return 0;
}
"""


def main():
    path = sys.argv[1] if len(sys.argv) > 1 else "tools/probes/mfma_shape_asm.hip"
    with open(path, "w") as f:
        f.write(HEAD)
        for shape in (32, 16):
            for lds in (0, 1):
                f.write(kernel(shape, lds))
        f.write(TAIL)


if __name__ == "__main__":
    main()
