// First convolution of the backbone (3 -> 64, 3x3, stride 1, pad 1: models/resnet_language.py:249 `conv1` of layer1.0 with
// :250-251 eval-mode BatchNorm + LeakyReLU(0.1) fused) straight from the fp32 NCHW image the reference's loader delivers
// (eval/language_eval.py:252: `net(support_xs)`), bf16, gfx950.
//
// Why its own kernel: the layer has 1728 MACs per pixel against 12 B read + 128 B written - it is HBM-bound by construction
// (ridge ~26 flop/B against 312).  The general kernel ran it as a K = 32 GEMM over an im2col buffer that an extra kernel
// wrote (64 B per pixel written, 64 B read back: 1.9x the layer's own traffic, plus one launch).  Here the image is read once:
//   * a workgroup takes R whole image rows of one image; their R + 2 input rows x 3 channel planes (contiguous runs of the
//     NCHW image) are converted to bf16 and laid out in LDS as [row][column + zero border][4 channels] (8 B per pixel, the
//     fourth channel the constant 1), so a 3x3 tap is a constant LDS offset and the zero padding needs no selects;
//   * GEMM K = 4 channels x 9 taps = 36, padded to 48 = three v_mfma_f32_32x32x16_bf16 k-steps; a lane's 8 k-values of a
//     k-step are two taps x 4 channels = two ds_read_b64 (no im2col row is ever materialised);
//   * operands SWAPPED (A = weights [cout][k], B = pixels): a lane then holds, for ONE pixel, 4 consecutive output channels per
//     4 accumulator registers; LeakyReLU, bf16 pairs, v_permlane32_swap between the two lanes of a pixel, and every lane
//     stores 16 bytes = 8 consecutive channels of its pixel's NHWC row (no LDS round trip in the epilogue); the BN shift is
//     two more weight columns against the constant-1 channel (bf16 hi + lo parts);
//   * the 64 x 27 weight matrix arrives in the backbone's packed layout ([cout][32], k = 3 tap + c: subreg_pack_conv_weight
//     mode 1, BN scale folded in) and is re-ordered into the six A fragments once per workgroup, through LDS.
// Algorithmic traffic 140 B per pixel; one launch, no workspace.
#include "subreg_common.h"

namespace subreg {

struct ConvFirstArgs {
    const float* img;    // [B][3][H][W] fp32
    const char* w;       // [64][32] bf16, k = 3 tap + c (27 used)
    char* y;             // [B H W][64] bf16
    const float* shift;  // [64]
    int B, H, W, R, tpi, ntiles, act;
    int dr, dc;          // 128 / W, 128 % W: a wave's next 32-pixel group is 128 pixels further
    int w4, nitems;      // VEC4 staging: float4 columns per row, (R + 2) * w4 items per tile
};

constexpr int CF_WAVES = 4, CF_NIT = 2;     // waves per workgroup; staging items per thread (VEC4 path)

typedef float cf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cf_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cf_pack(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(cf_f32x2{a, b}, cf_bf16x2));
}

// LDS patch: [(R + 2) rows][W + 4 columns] x 8 B = (c0, c1, c2, ONE) in bf16; image column x sits at column x + 2 (so that a
// float4 group of pixels starts on a 16-byte boundary), columns 1 and W + 2 are the zero border, rows outside the image are
// zeros.  The fourth channel is the constant 1 EVERYWHERE: the BN shift rides in the GEMM as weight columns (centre tap,
// channel 3) = bf16 hi part and (tap 3, channel 3) = lo part - no per-channel adds (and no 32 shift registers) in the epilogue.
// VEC4 (W % 4 == 0): a staging item is 4 consecutive pixels of one row - three float4 loads (one per channel plane), two
// 16-byte LDS writes - and the NEXT tile's items are loaded into registers before this tile's MFMAs, so the global latency
// hides behind the compute.  Otherwise: pixel-by-pixel staging, no prefetch (odd widths of tests only).
template <bool VEC4>
__global__ __launch_bounds__(CF_WAVES * 64) void conv_first_kernel(const ConvFirstArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int WP = a.W + 4, prow_b = WP * 8;                          // padded patch row: pixels, bytes
    const int patch_b = (a.R + 2) * prow_b;
    char* const patch = smem;
    __bf16* const wl = reinterpret_cast<__bf16*>(smem + patch_b);    // [64][32] weights as packed, then [64] shift floats
    float* const shl = reinterpret_cast<float*>(smem + patch_b + 64 * 32 * 2);

    reinterpret_cast<uint4*>(wl)[tid] = reinterpret_cast<const uint4*>(a.w)[tid];      // 256 threads x 16 B = 4 KiB
    if (tid < 64) shl[tid] = a.shift[tid];
    for (int o = tid * 8; o < patch_b; o += CF_WAVES * 64 * 8) *reinterpret_cast<uint2*>(patch + o) = make_uint2(0u, 0x3F800000u);
    __syncthreads();
    // ---- A fragments (weights).  Fragment (i, s): lane (r, h) holds k' = 16 s + 8 h + j, j = 0..7, of output channel
    //      32 i + r, with k' = 4 tap + c; c = 3 carries the shift (tap 4: hi, tap 3: lo), taps >= 9 are zero columns
    uint4 wf[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float shv = shl[32 * i + lr];
        const __bf16 sh_hi = (__bf16)shv, sh_lo = (__bf16)(shv - (float)sh_hi);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            unsigned short e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int tap = 4 * s + 2 * lh + (j >> 2), c = j & 3;
                const bool ok = c < 3 && tap < 9;
                const unsigned short v = __builtin_bit_cast(unsigned short, wl[(32 * i + lr) * 32 + (ok ? 3 * tap + c : 0)]);
                e[j] = ok ? v : (unsigned short)0;
                if (c == 3 && tap == 4) e[j] = __builtin_bit_cast(unsigned short, sh_hi);
                if (c == 3 && tap == 3) e[j] = __builtin_bit_cast(unsigned short, sh_lo);
            }
            wf[i][s] = make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16),
                                  e[6] | ((unsigned)e[7] << 16));
        }
    }
    // LDS byte offsets of this lane's two taps per k-step, relative to the pixel's top-left neighbour (dy = dx = -1)
    int toff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int tap = 4 * s + 2 * lh + u;
            tap = tap < 9 ? tap : 8;                                   // zero weight columns: any finite data will do
            toff[s][u] = (tap / 3) * prow_b + (tap % 3) * 8;
        }

    const size_t plane = (size_t)a.H * a.W;
    // ---- VEC4 staging: item = (patch row pr, float4 column j); loaded into registers, written to LDS a phase later
    float4 pv[CF_NIT][3];
    auto load_tile = [&](int t) {
        const int b = t / a.tpi, h0 = (t - b * a.tpi) * a.R;
        const float* const ib = a.img + (size_t)b * 3 * plane;
#pragma unroll
        for (int n = 0; n < CF_NIT; ++n) {
            const int item = tid + n * CF_WAVES * 64, pr = item / a.w4, j = item - pr * a.w4, h = h0 - 1 + pr;
            const bool ok = item < a.nitems && h >= 0 && h < a.H;
            const float4* p = reinterpret_cast<const float4*>(ib + (size_t)(ok ? h : 0) * a.W) + (ok ? j : 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float4 v = p[c * (plane / 4)];
                pv[n][c] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int n = 0; n < CF_NIT; ++n) {
            const int item = tid + n * CF_WAVES * 64, pr = item / a.w4, j = item - pr * a.w4;
            if (item < a.nitems) {
                const unsigned one = 0x3F800000u;                          // (c2, 1.0) -> high half = bf16 one
                uint4* dst = reinterpret_cast<uint4*>(patch + pr * prow_b + (4 * j + 2) * 8);
                dst[0] = make_uint4(cf_pack(pv[n][0].x, pv[n][1].x), (cf_pack(pv[n][2].x, 0.f) & 0xffffu) | one,
                                    cf_pack(pv[n][0].y, pv[n][1].y), (cf_pack(pv[n][2].y, 0.f) & 0xffffu) | one);
                dst[1] = make_uint4(cf_pack(pv[n][0].z, pv[n][1].z), (cf_pack(pv[n][2].z, 0.f) & 0xffffu) | one,
                                    cf_pack(pv[n][0].w, pv[n][1].w), (cf_pack(pv[n][2].w, 0.f) & 0xffffu) | one);
            }
        }
    };
    if (VEC4 && (int)blockIdx.x < a.ntiles) load_tile(blockIdx.x);

    for (int t = blockIdx.x; t < a.ntiles; t += gridDim.x) {
        const int b = t / a.tpi, h0 = (t - b * a.tpi) * a.R;
        const int rows = a.H - h0 < a.R ? a.H - h0 : a.R;
        __syncthreads();                                               // the previous tile's patch has been consumed
        if (VEC4) {
            store_tile();
        } else {
            // pixel-by-pixel: patch row pr = image row h0 - 1 + pr; wave w takes rows w, w + 4, ...; lanes walk the columns
            const float* const ib = a.img + (size_t)b * 3 * plane;
            for (int pr = wid; pr < a.R + 2; pr += CF_WAVES) {
                const int h = h0 - 1 + pr;
                const bool hin = h >= 0 && h < a.H;
                for (int x = lane; x < a.W; x += 64) {
                    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
                    if (hin) {
                        const float* p = ib + (size_t)h * a.W + x;
                        v0 = p[0]; v1 = p[plane]; v2 = p[2 * plane];
                    }
                    *reinterpret_cast<uint2*>(patch + pr * prow_b + (x + 2) * 8) =
                        make_uint2(cf_pack(v0, v1), (cf_pack(v2, 0.f) & 0xffffu) | 0x3F800000u);
                }
            }
        }
        __syncthreads();
        if (VEC4 && t + (int)gridDim.x < a.ntiles) load_tile(t + gridDim.x);   // in flight during this tile's MFMAs
        // ---- compute: 32-pixel groups of the tile's rows x W pixels (row-major); wave w takes groups w, w + 4, ...
        const int npx = rows * a.W;
        const size_t pix0 = ((size_t)b * a.H + h0) * a.W;
        int p = wid * 32 + lr, py = p / a.W, px = p - py * a.W;
        for (int g0 = wid * 32; g0 < npx; g0 += CF_WAVES * 32) {
            const bool valid = p < npx;
            const char* const base = patch + (valid ? py * prow_b + (px + 1) * 8 : 8);
            f32x16 acc[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const uint2 t0 = *reinterpret_cast<const uint2*>(base + toff[s][0]);
                const uint2 t1 = *reinterpret_cast<const uint2*>(base + toff[s][1]);
                const uint4 xf = make_uint4(t0.x, t0.y, t1.x, t1.y);
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[i][s]), __builtin_bit_cast(bf16x8, xf),
                                                                      acc[i], 0, 0, 0);
            }
            // ---- epilogue: lane (pixel lr, half lh) holds channels 32 i + 8 q + 4 lh + {0..3} in registers 4q..4q+3 (shift
            //      already inside); LeakyReLU, bf16 pairs, halves exchanged with the partner lane, 16-byte stores
            char* const yrow = a.y + (pix0 + (valid ? p : 0)) * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    unsigned pk[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            float v0 = acc[i][4 * (q + u) + 2 * e], v1 = acc[i][4 * (q + u) + 2 * e + 1];
                            if (a.act) { v0 = fmaxf(v0, v0 * 0.1f); v1 = fmaxf(v1, v1 * 0.1f); }
                            pk[u][e] = cf_pack(v0, v1);
                        }
                    // lower lanes keep group q and receive its upper half from the partner; upper lanes keep group q + 1
                    const auto s0 = __builtin_amdgcn_permlane32_swap(pk[0][0], pk[1][0], false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(pk[0][1], pk[1][1], false, false);
                    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
                    const u32x4v vec = {s0[0], s1[0], s0[1], s1[1]};
                    if (valid) *reinterpret_cast<u32x4v*>(yrow + (32 * i + 8 * (q + lh)) * 2) = vec;
                }
            p += CF_WAVES * 32;
            py += a.dr;
            px += a.dc;
            if (px >= a.W) { px -= a.W; ++py; }
        }
    }
}

// shapes the direct first layer takes
bool conv_first_supported(int B, int H, int W) {
    return B > 0 && H >= 1 && W >= 8 && W <= 1024 && (long long)B * H * W < (1LL << 31) - 4096;
}

int conv_first(const float* img, const void* w, void* y, const float* shift, int B, int H, int W, int act, hipStream_t stream) {
    if (!conv_first_supported(B, H, W)) return SUBREG_EUNSUPPORTED;
    ConvFirstArgs a;
    a.img = img; a.w = (const char*)w; a.y = (char*)y; a.shift = shift;
    a.B = B; a.H = H; a.W = W; a.act = act;
    const bool vec4 = W % 4 == 0 && ((size_t)img % 16) == 0 && ((size_t)H * W) % 4 == 0;
    int R = 1024 / W;                                                  // <= 1024 pixels (32 groups) per tile ...
    if (R > H) R = H;
    if (R < 1) R = 1;
    for (int cand = R; cand >= (R + 1) / 2; --cand)                    // ... preferring a divisor of H (no short last tile)
        if (H % cand == 0) { R = cand; break; }
    while (vec4 && R > 1 && (R + 2) * (W / 4) > CF_NIT * CF_WAVES * 64) --R;   // the staging items must fit the prefetch registers
    if (vec4 && (R + 2) * (W / 4) > CF_NIT * CF_WAVES * 64) return SUBREG_EUNSUPPORTED;
    a.R = R;
    a.tpi = (H + R - 1) / R;
    a.ntiles = B * a.tpi;
    a.dr = (CF_WAVES * 32) / W;
    a.dc = (CF_WAVES * 32) % W;
    a.w4 = W / 4 > 0 ? W / 4 : 1;
    a.nitems = (R + 2) * a.w4;
    const size_t lds = (size_t)(R + 2) * (W + 4) * 8 + 64 * 32 * 2 + 64 * 4;
    auto kern = vec4 ? conv_first_kernel<true> : conv_first_kernel<false>;
    static std::atomic<unsigned long long> lds_set[2] = {{0}, {0}};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set[vec4 ? 1 : 0])) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = a.ntiles < 4 * cus ? a.ntiles : 4 * cus;          // grid-stride over tiles: set-up paid <= 4 x per CU, tiles pipelined
    hipLaunchKernelGGL(kern, dim3(grid), dim3(CF_WAVES * 64), lds, stream, a);
    return launch_status();
}

}  // namespace subreg

using namespace subreg;

extern "C" int subreg_conv_first_fwd(const float* x_nchw, const void* w_packed, void* y, const float* shift, int B, int H, int W,
                                     int Cout, int flags, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x_nchw && w_packed && y && shift && B > 0 && H > 0 && W > 0);
    if (dtype != SUBREG_BF16 || Cout != 64) return SUBREG_EUNSUPPORTED;
    return conv_first(x_nchw, w_packed, y, shift, B, H, W, (flags & SUBREG_CONV_LRELU) ? 1 : 0, (hipStream_t)stream);
}
