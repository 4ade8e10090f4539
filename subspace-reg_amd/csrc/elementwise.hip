// HBM-bound helper kernels around the convolution (gfx950): layout packing, BN
// folding / train-mode statistics, the train-mode normalise+residual+activation+
// pool+mask pass, average pooling.  Reference call sites in
// models/resnet_language.py: nn.BatchNorm2d :148,250,253,255; LeakyReLU :251;
// residual add :288; MaxPool2d :256,290; F.dropout :299; DropBlock :311-325;
// AdaptiveAvgPool2d(1) :125,179-181.
#include "subreg_common.h"

namespace subreg {

constexpr int EW_THREADS = 256;

static inline int ew_blocks(size_t n, int per_thread = 1) {
    size_t b = (n + (size_t)EW_THREADS * per_thread - 1) / ((size_t)EW_THREADS * per_thread);
    return (int)(b < 1 ? 1 : b);
}

// ---------------------------------------------------------------- first-layer im2col (Cin = 3 -> K = 32)
// x NCHW f32 [B,3,H,W] -> col [B*H*W][32] T with k = 3*tap + c (tap = 3*ky+kx), k >= 27 zero.
// One workgroup = PACK_ROWS image rows of one image: the PACK_ROWS + 2 input rows of the three channels go through LDS with
// coalesced loads (zero-padded at the borders), then one thread = one 16-byte piece of an im2col row (8 bf16 / 4 f32 values
// of k), so consecutive lanes store consecutive addresses.  Several rows per workgroup so that one load -> barrier ->
// store latency chain moves ~38 KB (one row per workgroup left the kernel latency-bound at 2 TB/s; 7 rows: 4.2 TB/s,
// 44 us for 350 images where the per-pixel gather took 88).
constexpr int PACK_ROWS = 7;
template <typename T>
__global__ __launch_bounds__(1024) void pack_input_kernel(const float* __restrict__ x, T* __restrict__ col, int B, int H, int W) {
    constexpr int VEC = 16 / sizeof(T), PER = 32 / VEC;             // values per thread, threads per pixel
    extern __shared__ float s_in[];                                  // [PACK_ROWS + 2 rows][3 channels][W + 2]
    const int hblocks = (H + PACK_ROWS - 1) / PACK_ROWS;
    const int b = blockIdx.x / hblocks, h0 = (blockIdx.x % hblocks) * PACK_ROWS, W2 = W + 2, hw = H * W;
    const float* xb = x + (size_t)b * 3 * hw;
    for (int i = threadIdx.x; i < (PACK_ROWS + 2) * 3 * W2; i += blockDim.x) {
        const int dy = i / (3 * W2), c = (i / W2) % 3, wc = i % W2;  // wc = w + 1
        const int hh = h0 + dy - 1, ww = wc - 1;
        s_in[i] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? xb[(size_t)c * hw + hh * W + ww] : 0.f;
    }
    __syncthreads();
    const int rows = H - h0 < PACK_ROWS ? H - h0 : PACK_ROWS;
    T* const out0 = col + ((size_t)b * hw + (size_t)h0 * W) * 32;   // the rows of one image are contiguous
    // a thread keeps its (pixel column w, 16-byte piece q) and walks down the rows: the tap offsets are computed once
    for (int item = threadIdx.x; item < W * PER; item += blockDim.x) {
        const int w = item / PER, q = item % PER;
        int off[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int k = q * VEC + j, t = k / 3, c = k % 3;         // k = 3*tap + c
            off[j] = k < 27 ? ((t / 3) * 3 + c) * W2 + w + t % 3 : -1;
        }
        for (int r = 0; r < rows; ++r) {
            uint4 v;
            T* out = reinterpret_cast<T*>(&v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) out[j] = ElemTraits<T>::from_float(off[j] >= 0 ? s_in[off[j] + r * 3 * W2] : 0.f);
            *reinterpret_cast<uint4*>(out0 + ((size_t)r * W + w) * 32 + q * VEC) = v;
        }
    }
}

// ---------------------------------------------------------------- weight packing
// mode 0: OIHW f32 -> [k*k][Cin/32][Cout][32] T: one (tap, 32-channel chunk) weight tile is Cout contiguous 64/128-byte
//         rows, so every 1-KiB LDS-DMA piece of conv_fwd.hip (16 or 8 rows) reads whole cache lines.
// mode 1 (Cin == 3): -> [Cout][32] T over the im2col K axis (3x3: k = 3*tap + c; 1x1: centre tap 4).
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, const float* __restrict__ fold, T* __restrict__ out, int Cout,
                                   int Cin, int ks, int mode) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int taps = ks * ks;
    if (mode == 0) {
        const size_t total = (size_t)Cout * taps * Cin;
        if (i >= total) return;
        const int nch = Cin / 32;
        const int c32 = i % 32, o = (i / 32) % Cout, ch = (i / ((size_t)32 * Cout)) % nch, t = i / ((size_t)32 * Cout * nch);
        out[i] = ElemTraits<T>::from_float(w[((size_t)o * Cin + ch * 32 + c32) * taps + t] * (fold ? fold[o] : 1.f));
    } else {
        const size_t total = (size_t)Cout * 32;
        if (i >= total) return;
        const int k = i % 32, o = i / 32;
        float v = 0.f;
        if (k < 27) {
            const int t = k / 3, c = k % 3;
            if (ks == 3) v = w[((size_t)o * 3 + c) * 9 + t];
            else if (t == 4) v = w[(size_t)o * 3 + c];
        }
        out[i] = ElemTraits<T>::from_float(v * (fold ? fold[o] : 1.f));
    }
}

template <typename T>
__global__ void pack_identity_kernel(T* __restrict__ out, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)C * C) return;
    const int c32 = i % 32, o = (i / 32) % C, ch = i / ((size_t)32 * C);     // [C/32][C][32], as pack_weight_kernel
    out[i] = ElemTraits<T>::from_float(o == ch * 32 + c32 ? 1.f : 0.f);
}

__global__ void vec_add_kernel(float* __restrict__ dst, const float* __restrict__ a, const float* __restrict__ b, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a[i] + b[i];
}

// ---------------------------------------------------------------- layout conversion
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int B, int C, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over NHWC output
    if (i >= (size_t)B * C * HW) return;
    const int c = i % C;
    const size_t pb = i / C;
    const int p = pb % HW, b = pb / HW;
    y[i] = ElemTraits<T>::from_float(x[((size_t)b * C + c) * HW + p]);
}

template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over NCHW output
    if (i >= (size_t)B * C * HW) return;
    const int p = i % HW;
    const size_t cb = i / HW;
    const int c = cb % C, b = cb / C;
    y[i] = ElemTraits<T>::to_float(x[((size_t)b * HW + p) * C + c]);
}

// ---------------------------------------------------------------- BN folding (eval mode)
__global__ void bn_fold_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                               const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ scale,
                               float* __restrict__ shift, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float s = weight[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = bias[c] - rm[c] * s;
}

// eval-mode BatchNorm of a forward that keeps a stash for the backward (whole-network fine-tuning before freeze_backbone_at,
// eval/language_eval.py:243-249 with the model in eval mode from the first validate() on): scale / shift from the RUNNING
// statistics, which also take the place of the batch mean / invstd the backward normalises with
__global__ void bn_eval_stash_kernel(const float* __restrict__ weight, const float* __restrict__ bias,
                                     const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ scale,
                                     float* __restrict__ shift, float* __restrict__ mean, float* __restrict__ invstd, int C, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float is = 1.f / sqrtf(rv[c] + eps), s = weight[c] * is;
    scale[c] = s;
    shift[c] = bias[c] - rm[c] * s;
    mean[c] = rm[c];
    invstd[c] = is;
}

// ---------------------------------------------------------------- BN train-mode statistics
// partial [rows][C][2] (sum, sumsq) -> batch mean / biased var -> scale/shift for this batch, and the
// running-stat update (momentum, unbiased variance) in place.  One 256-thread block per channel; per-thread
// strided sums and a fixed-shape tree in double => bitwise reproducible.
__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ partial, int rows, int C, double count,
                                                          const float* __restrict__ weight, const float* __restrict__ bias,
                                                          float* __restrict__ rm, float* __restrict__ rv, float momentum,
                                                          float eps, float* __restrict__ scale, float* __restrict__ shift,
                                                          float* __restrict__ save_mean, float* __restrict__ save_invstd) {
    __shared__ double r1[256], r2[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    // eight loads in flight per thread: at 7056 rows (64 x 84 x 84 pixels) the rolled loop was 28 dependent round trips
    constexpr int UN = 8;
    for (int r0 = tid; r0 < rows; r0 += 256 * UN) {
        float2 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int r = r0 + u * 256;
            v[u] = r < rows ? *reinterpret_cast<const float2*>(partial + ((size_t)r * C + c) * 2) : make_float2(0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) { s1 += (double)v[u].x; s2 += (double)v[u].y; }
    }
    r1[tid] = s1;
    r2[tid] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { r1[tid] += r1[tid + o]; r2[tid] += r2[tid + o]; }
        __syncthreads();
    }
    if (tid != 0) return;
    s1 = r1[0]; s2 = r2[0];
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double inv = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)weight[c] * inv;
    scale[c] = (float)sc;
    shift[c] = (float)((double)bias[c] - mean * sc);
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = (float)inv;
    const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
    rm[c] = (float)((1.0 - (double)momentum) * (double)rm[c] + (double)momentum * mean);
    rv[c] = (float)((1.0 - (double)momentum) * (double)rv[c] + (double)momentum * unbiased);
}

// ---------------------------------------------------------------- train-mode apply pass
// y = mask * mask_scale * pool( act( x*scale+shift + (res*rscale+rshift | res) ) ), all NHWC.  HBM-bound pass.
// Thread = (channel group of 16 bytes, pixel lane), like the BN-backward reduce: its 8 / 4 channels' coefficients are
// loaded ONCE and it then walks output pixels with stride `lanes`, four pixels in flight per step.  (One thread per
// 16 output bytes with its coefficients re-loaded every time - 32 scalar loads beside two vector loads - ran at
// 1.0-2.0 TB/s algorithmic, profiles/r03_hbm_kernels.txt.)
template <typename T, bool POOL, bool RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, const T* __restrict__ res,
                                                       const float* __restrict__ rscale, const float* __restrict__ rshift,
                                                       const unsigned char* __restrict__ keep, float mask_scale_host,
                                                       const float* __restrict__ mask_scale_dev,
                                                       T* __restrict__ y, int H, int W, int C, int act, long long npo,
                                                       int ppb) {
    constexpr int VEC = 16 / sizeof(T), UN = POOL ? 1 : 4;
    const float mask_scale = mask_scale_dev ? *mask_scale_dev : mask_scale_host;   // DropBlock: numel / count, still on the device
    const int ngrp = C / VEC, lanes = 256 / ngrp;
    const int cg = threadIdx.x % ngrp, pl = threadIdx.x / ngrp;
    if (pl >= lanes) return;
    const int c0 = cg * VEC;
    float sc[VEC], sh[VEC], rsc[VEC], rsh[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        sc[k] = scale[c0 + k]; sh[k] = shift[c0 + k];
        rsc[k] = (RES && rscale) ? rscale[c0 + k] : 1.f; rsh[k] = (RES && rshift) ? rshift[c0 + k] : 0.f;
    }
    const long long p0 = (long long)blockIdx.x * ppb;
    long long p1 = p0 + ppb;
    if (p1 > npo) p1 = npo;
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W;
    auto finish = [&](float (&v)[VEC], long long po) {
        uint4 vo;
        T* to = reinterpret_cast<T*>(&vo);
        const size_t e = (size_t)po * C + c0;
        unsigned long long kb = 0x0101010101010101ull;
        if (keep) kb = VEC == 8 ? *reinterpret_cast<const unsigned long long*>(keep + e) : (unsigned long long)*reinterpret_cast<const unsigned*>(keep + e);
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float r = v[k];
            if (act) r = lrelu(r);
            if (keep) r = ((kb >> (8 * k)) & 0xff) ? r * mask_scale : 0.f;
            to[k] = ElemTraits<T>::from_float(r);
        }
        *reinterpret_cast<uint4*>(y + e) = vo;
    };
    if constexpr (!POOL) {
        for (long long pb = p0 + pl; pb < p1; pb += (long long)lanes * UN) {
            uint4 vx[UN], vr[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long p = pb + (long long)u * lanes;
                if (p < p1) {
                    vx[u] = *reinterpret_cast<const uint4*>(x + (size_t)p * C + c0);
                    if (RES) vr[u] = *reinterpret_cast<const uint4*>(res + (size_t)p * C + c0);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long p = pb + (long long)u * lanes;
                if (p < p1) {
                    const T* tx = reinterpret_cast<const T*>(&vx[u]);
                    const T* tr = reinterpret_cast<const T*>(&vr[u]);
                    float v[VEC];
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        v[k] = ElemTraits<T>::to_float(tx[k]) * sc[k] + sh[k];
                        if (RES) v[k] += ElemTraits<T>::to_float(tr[k]) * rsc[k] + rsh[k];
                    }
                    finish(v, p);
                }
            }
        }
    } else {
        for (long long po = p0 + pl; po < p1; po += lanes) {
            const int wo = (int)(po % Wo), ho = (int)((po / Wo) % Ho);
            const long long b = po / ((long long)Wo * Ho);
            const size_t pin = ((size_t)b * H + 2 * ho) * W + 2 * wo;        // top-left pixel of the 2x2 window
            uint4 vx[4], vr[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const size_t e = (pin + (q >> 1) * W + (q & 1)) * C + c0;
                vx[q] = *reinterpret_cast<const uint4*>(x + e);
                if (RES) vr[q] = *reinterpret_cast<const uint4*>(res + e);
            }
            float best[VEC];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const T* tx = reinterpret_cast<const T*>(&vx[q]);
                const T* tr = reinterpret_cast<const T*>(&vr[q]);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float v = ElemTraits<T>::to_float(tx[k]) * sc[k] + sh[k];
                    if (RES) v += ElemTraits<T>::to_float(tr[k]) * rsc[k] + rsh[k];
                    best[k] = q == 0 ? v : fmaxf(best[k], v);
                }
            }
            finish(best, po);
        }
    }
}

// ---------------------------------------------------------------- keep-mask helpers
// NCHW f32 {0,1} -> NHWC u8 (masks are drawn in the reference's NCHW order).
__global__ void mask_nchw_to_nhwc_kernel(const float* __restrict__ m, unsigned char* __restrict__ out, int B, int C, int HW,
                                         int invert) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * C * HW) return;
    const int c = i % C;
    const size_t pb = i / C;
    const int p = pb % HW, b = pb / HW;
    const bool one = m[((size_t)b * C + c) * HW + p] != 0.f;
    out[i] = (one != (invert != 0)) ? 1 : 0;
}

// counter-based Bernoulli keep mask for free-running (non-injected) train-mode forwards
__device__ __forceinline__ unsigned mix32(unsigned long long z) {
    z ^= z >> 33; z *= 0xff51afd7ed558ccdULL; z ^= z >> 33; z *= 0xc4ceb9fe1a85ec53ULL; z ^= z >> 33;
    return (unsigned)(z >> 11);
}
// element i keeps iff uniform(seed, i) >= p_drop.  16 elements per thread (one 16-byte store); the kept count goes through
// a block reduction and ONE atomic per block (an atomic per wave on a single address serialises the whole kernel).
__global__ __launch_bounds__(256) void random_keep_kernel(unsigned char* __restrict__ out, size_t n, unsigned long long seed,
                                                          float p_drop, unsigned int* __restrict__ kept_count) {
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    unsigned kept = 0;
    if (i0 < n) {
        unsigned char b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float u = (mix32(seed * 0x9E3779B97F4A7C15ULL + i0 + k) & 0xFFFFFF) * (1.0f / 16777216.0f);
            b[k] = (i0 + k < n && u >= p_drop) ? 1 : 0;
            kept += b[k];
        }
        if (i0 + 16 <= n && (reinterpret_cast<size_t>(out) & 15) == 0) {
            *reinterpret_cast<uint4*>(out + i0) = *reinterpret_cast<const uint4*>(b);
        } else {
            for (int k = 0; k < 16 && i0 + k < n; ++k) out[i0 + k] = b[k];
        }
    }
    if (kept_count) {
        __shared__ unsigned wsum[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = kept;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kept_count, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    }
}

// ---------------------------------------------------------------- DropBlock block mask (block_size > 1)
// DropBlock._compute_block_mask (models/resnet_language.py:327-357) on the device: sample [B][C][H-bs+1][W-bs+1] (u8, 1 =
// seed) -> keep mask NHWC u8 (pre-filled with ones by the caller).  The reference pairs seed i % n with offset i % bs^2 for
// i < n bs^2 (nz.repeat vs offsets.repeat, :345-346), so the seed of rank r (row-major order of Tensor.nonzero()) receives
// the offset o iff (r - o) % gcd(n, bs^2) == 0 - plus its own padded position (the padded sample itself).  The ranks need a
// prefix sum over the sample in row-major order: ONE workgroup, every thread counts the seeds of its contiguous chunk, an
// LDS scan gives each chunk its first rank and the total n, then every thread walks its chunk again and clears the covered
// mask elements (plain byte stores of 0: writers never disagree).  The reference's scripts all pass --no_dropblock (block
// size 1, which never comes here), so this is a correctness path, not a throughput one.
__global__ __launch_bounds__(1024) void dropblock_mask_kernel(const unsigned char* __restrict__ sample, unsigned char* __restrict__ keep,
                                                              int B, int C, int H, int W, int bs) {
    __shared__ unsigned cnt[1024];
    __shared__ unsigned total;
    const int hs = H - bs + 1, ws = W - bs + 1, lp = (bs - 1) / 2, tid = threadIdx.x;
    const size_t n_el = (size_t)B * C * hs * ws;
    const size_t per = (n_el + 1023) / 1024, lo = (size_t)tid * per, hi = lo + per < n_el ? lo + per : n_el;
    unsigned c = 0;
    for (size_t i = lo; i < hi; ++i) c += sample[i] != 0;
    cnt[tid] = c;
    __syncthreads();
    if (tid == 0) {                                   // exclusive scan of 1024 chunk counts (serial: 1024 adds)
        unsigned run = 0;
        for (int k = 0; k < 1024; ++k) { const unsigned v = cnt[k]; cnt[k] = run; run += v; }
        total = run;
    }
    __syncthreads();
    const unsigned n = total;
    if (n == 0) return;
    unsigned g = n, b2 = (unsigned)(bs * bs);
    while (b2) { const unsigned t = g % b2; g = b2; b2 = t; }       // g = gcd(n, bs^2)
    unsigned rank = cnt[tid];
    for (size_t i = lo; i < hi; ++i) {
        if (sample[i] == 0) continue;
        const int j = (int)(i % ws), ii = (int)((i / ws) % hs), cc = (int)((i / ((size_t)ws * hs)) % C);
        const size_t b = i / ((size_t)ws * hs * C);
        for (int o = 0; o < bs * bs; ++o) {
            const int oy = o / bs, ox = o % bs;
            const bool own = oy == lp && ox == lp;                   // the seed itself in the padded sample
            if (!own && (int)((rank + (unsigned)(bs * bs) * g - (unsigned)o) % g) != 0) continue;
            keep[((b * H + ii + oy) * W + j + ox) * C + cc] = 0;
        }
        ++rank;
    }
}

__global__ __launch_bounds__(256) void count_ones_kernel(const unsigned char* __restrict__ m, size_t n, unsigned int* __restrict__ out) {
    unsigned c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += m[i] != 0;
    __shared__ unsigned wsum[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// ---------------------------------------------------------------- AdaptiveAvgPool2d(1) + view
template <typename T>
__global__ void avgpool_kernel(const T* __restrict__ x, float* __restrict__ feat, int B, int HW, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [B][C]
    if (i >= (size_t)B * C) return;
    const int c = i % C, b = i / C;
    float s = 0.f;
    for (int p = 0; p < HW; ++p) s += ElemTraits<T>::to_float(x[((size_t)b * HW + p) * C + c]);
    feat[i] = s / (float)HW;
}

}  // namespace subreg

using namespace subreg;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)      \
    if ((dtype) == SUBREG_F32) { CALL_F32; }        \
    else if ((dtype) == SUBREG_BF16) { CALL_BF16; } \
    else return SUBREG_EINVAL;

extern "C" int subreg_pack_input(const float* x_nchw, void* col, int B, int H, int W, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x_nchw && col && B > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    SUBREG_CHECK_ARG(W <= 1024 && (long long)B * H < (1LL << 31));
    const size_t lds = (size_t)(PACK_ROWS + 2) * 3 * (W + 2) * sizeof(float);
    const int grid = B * ((H + PACK_ROWS - 1) / PACK_ROWS);
    const int per = dtype == SUBREG_BF16 ? 4 : 8;                    // 16-byte pieces per im2col row
    int threads = ((W * per + 63) / 64) * 64;
    if (threads > 1024) threads = 1024;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_input_kernel<float>, grid, threads, lds, s, x_nchw, (float*)col, B, H, W),
               hipLaunchKernelGGL(pack_input_kernel<__bf16>, grid, threads, lds, s, x_nchw, (__bf16*)col, B, H, W));
    return launch_status();
}

extern "C" int subreg_pack_identity(void* out, int Cc, int dtype, void* stream) {
    SUBREG_CHECK_ARG(out && Cc > 0);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)Cc * Cc;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_identity_kernel<float>, ew_blocks(n), EW_THREADS, 0, s, (float*)out, Cc),
               hipLaunchKernelGGL(pack_identity_kernel<__bf16>, ew_blocks(n), EW_THREADS, 0, s, (__bf16*)out, Cc));
    return launch_status();
}

extern "C" int subreg_vec_add(float* dst, const float* a, const float* b, int n, void* stream) {
    SUBREG_CHECK_ARG(dst && a && b && n > 0);
    hipLaunchKernelGGL(vec_add_kernel, ew_blocks(n), EW_THREADS, 0, (hipStream_t)stream, dst, a, b, n);
    return launch_status();
}

extern "C" int subreg_pack_conv_weight(const float* w_oihw, const float* fold_scale, void* out, int Cout, int Cin, int ksize,
                                       int mode, int dtype, void* stream) {
    SUBREG_CHECK_ARG(w_oihw && out && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3));
    SUBREG_CHECK_ARG(mode == 0 || (mode == 1 && Cin == 3));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = mode == 0 ? (size_t)Cout * ksize * ksize * Cin : (size_t)Cout * 32;
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(pack_weight_kernel<float>, ew_blocks(n), EW_THREADS, 0, s, w_oihw, fold_scale, (float*)out, Cout, Cin, ksize, mode),
               hipLaunchKernelGGL(pack_weight_kernel<__bf16>, ew_blocks(n), EW_THREADS, 0, s, w_oihw, fold_scale, (__bf16*)out, Cout, Cin, ksize, mode));
    return launch_status();
}

extern "C" int subreg_nchw_to_nhwc(const float* x, void* y, int B, int C, int H, int W, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * C * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, ew_blocks(n), EW_THREADS, 0, s, x, (float*)y, B, C, H * W),
               hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, ew_blocks(n), EW_THREADS, 0, s, x, (__bf16*)y, B, C, H * W));
    return launch_status();
}

extern "C" int subreg_nhwc_to_nchw(const void* x, float* y, int B, int C, int H, int W, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && y && B > 0 && C > 0 && H > 0 && W > 0);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * C * H * W;
    DISPATCH_T(dtype, hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, ew_blocks(n), EW_THREADS, 0, s, (const float*)x, y, B, C, H * W),
               hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, ew_blocks(n), EW_THREADS, 0, s, (const __bf16*)x, y, B, C, H * W));
    return launch_status();
}

extern "C" int subreg_bn_fold(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                              float* scale, float* shift, int C, float eps, void* stream) {
    SUBREG_CHECK_ARG(weight && bias && running_mean && running_var && scale && shift && C > 0);
    hipLaunchKernelGGL(bn_fold_kernel, ew_blocks(C), EW_THREADS, 0, (hipStream_t)stream, weight, bias, running_mean,
                       running_var, scale, shift, C, eps);
    return launch_status();
}

extern "C" int subreg_bn_eval_stash(const float* weight, const float* bias, const float* running_mean, const float* running_var,
                                    float eps, int C, float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
    SUBREG_CHECK_ARG(weight && bias && running_mean && running_var && scale && shift && save_mean && save_invstd && C > 0);
    hipLaunchKernelGGL(bn_eval_stash_kernel, ew_blocks(C), EW_THREADS, 0, (hipStream_t)stream, weight, bias, running_mean, running_var,
                       scale, shift, save_mean, save_invstd, C, eps);
    return launch_status();
}

extern "C" int subreg_bn_train_finalize(const float* stats_partial, int rows, int C, long long count, const float* weight,
                                        const float* bias, float* running_mean, float* running_var, float momentum,
                                        float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                        void* stream) {
    SUBREG_CHECK_ARG(stats_partial && weight && bias && running_mean && running_var && scale && shift);
    SUBREG_CHECK_ARG(rows > 0 && C > 0 && count > 0);
    hipLaunchKernelGGL(bn_finalize_kernel, C, 256, 0, (hipStream_t)stream, stats_partial, rows, C,
                       (double)count, weight, bias, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd);
    return launch_status();
}

extern "C" int subreg_bn_apply(const void* x, const float* scale, const float* shift, const void* residual,
                               const float* res_scale, const float* res_shift, const unsigned char* keep_mask,
                               float mask_scale, const float* mask_scale_dev, void* y, int B, int H, int W, int C, int flags,
                               int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && scale && shift && y && B > 0 && H > 0 && W > 0 && C > 0);
    const int act = (flags & SUBREG_CONV_LRELU) ? 1 : 0, pool = (flags & SUBREG_CONV_POOL2) ? 1 : 0;
    SUBREG_CHECK_ARG(!pool || (H >= 2 && W >= 2));
    SUBREG_CHECK_ARG(C % (dtype == SUBREG_BF16 ? 8 : 4) == 0);      // 16-byte vectors along the channel axis
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == SUBREG_BF16 ? 8 : 4;
    SUBREG_CHECK_ARG(C / vec <= 256);
    const long long npo = (long long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const int lanes = 256 / (C / vec);
    // pixels per block: ~4096 blocks, at least one unrolled step of every pixel lane
    long long ppb = (npo + 4095) / 4096;
    const int step = lanes * (pool ? 1 : 4);
    ppb = (ppb + step - 1) / step * step;
    if (ppb < step) ppb = step;
    const int grid = (int)((npo + ppb - 1) / ppb);
#define BNA(TT, P, R) hipLaunchKernelGGL((bn_apply_kernel<TT, P, R>), grid, 256, 0, s, (const TT*)x, scale, shift, (const TT*)residual, \
                                         res_scale, res_shift, keep_mask, mask_scale, mask_scale_dev, (TT*)y, H, W, C, act, npo, (int)ppb)
#define BNA_T(TT) do { if (pool) { if (residual) BNA(TT, true, true); else BNA(TT, true, false); } \
                       else { if (residual) BNA(TT, false, true); else BNA(TT, false, false); } } while (0)
    DISPATCH_T(dtype, BNA_T(float), BNA_T(__bf16));
#undef BNA_T
#undef BNA
    return launch_status();
}

extern "C" int subreg_mask_nchw_to_nhwc(const float* mask_nchw, unsigned char* keep_nhwc, int B, int C, int H, int W,
                                        int invert, void* stream) {
    SUBREG_CHECK_ARG(mask_nchw && keep_nhwc && B > 0 && C > 0 && H > 0 && W > 0);
    const size_t n = (size_t)B * C * H * W;
    hipLaunchKernelGGL(mask_nchw_to_nhwc_kernel, ew_blocks(n), EW_THREADS, 0, (hipStream_t)stream, mask_nchw, keep_nhwc,
                       B, C, H * W, invert);
    return launch_status();
}

extern "C" int subreg_random_keep_mask(unsigned char* keep, long long n, unsigned long long seed, float p_drop,
                                       unsigned int* kept_count, void* stream) {
    SUBREG_CHECK_ARG(keep && n > 0 && p_drop >= 0.f && p_drop < 1.f);
    hipLaunchKernelGGL(random_keep_kernel, dim3((unsigned)(((size_t)n + 4095) / 4096)), dim3(256), 0, (hipStream_t)stream, keep,
                       (size_t)n, seed, p_drop, kept_count);
    return launch_status();
}

// The same kernel with its seed and drop probability read from DEVICE memory when the launch RUNS: a captured hipGraph replays the launch with
// fresh randomness (and DropBlock's step-dependent gamma, :294-296) after the host has refreshed *param with subreg_mask_params_set.
__global__ __launch_bounds__(256) void random_keep_dev_kernel(unsigned char* __restrict__ out, size_t n, const subreg_mask_param* __restrict__ param,
                                                              unsigned int* __restrict__ kept_count) {
    const unsigned long long seed = param->seed;
    const float p_drop = param->p_drop;
    const size_t i0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
    unsigned kept = 0;
    if (i0 < n) {
        unsigned char b[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float u = (mix32(seed * 0x9E3779B97F4A7C15ULL + i0 + k) & 0xFFFFFF) * (1.0f / 16777216.0f);
            b[k] = (i0 + k < n && u >= p_drop) ? 1 : 0;
            kept += b[k];
        }
        if (i0 + 16 <= n && (reinterpret_cast<size_t>(out) & 15) == 0) {
            *reinterpret_cast<uint4*>(out + i0) = *reinterpret_cast<const uint4*>(b);
        } else {
            for (int k = 0; k < 16 && i0 + k < n; ++k) out[i0 + k] = b[k];
        }
    }
    if (kept_count) {
        __shared__ unsigned wsum[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = kept;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(kept_count, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
    }
}

extern "C" int subreg_random_keep_mask_dev(unsigned char* keep, long long n, const subreg_mask_param* param, unsigned int* kept_count,
                                           void* stream) {
    SUBREG_CHECK_ARG(keep && n > 0 && param);
    hipLaunchKernelGGL(random_keep_dev_kernel, dim3((unsigned)(((size_t)n + 4095) / 4096)), dim3(256), 0, (hipStream_t)stream, keep,
                       (size_t)n, param, kept_count);
    return launch_status();
}

struct MaskParamPack { subreg_mask_param v[SUBREG_MASK_PARAMS_MAX]; };
__global__ void mask_params_set_kernel(subreg_mask_param* __restrict__ dst, int n, const MaskParamPack pack) {
    const int i = threadIdx.x;
    if (i < n) dst[i] = pack.v[i];
}

// host values travel as a kernel ARGUMENT (no pinned staging buffer whose reuse would have to be ordered against the stream)
extern "C" int subreg_mask_params_set(subreg_mask_param* params_dev, int n, const subreg_mask_param* host_values, void* stream) {
    SUBREG_CHECK_ARG(params_dev && host_values && n > 0 && n <= SUBREG_MASK_PARAMS_MAX);
    MaskParamPack pack;
    for (int i = 0; i < SUBREG_MASK_PARAMS_MAX; ++i) pack.v[i] = host_values[i < n ? i : n - 1];
    for (int i = 0; i < n; ++i) SUBREG_CHECK_ARG(pack.v[i].p_drop >= 0.f && pack.v[i].p_drop < 1.f);
    hipLaunchKernelGGL(mask_params_set_kernel, 1, 64, 0, (hipStream_t)stream, params_dev, n, pack);
    return launch_status();
}

// DropBlock's rescale factor countM / count_ones (resnet_language.py:318-323) from the device-side counter of the mask kernels:
// (float)(numel / max(count, 1)) in double like the host expression it replaces - without the host reading the counter
__global__ void mask_scale_kernel(const unsigned int* __restrict__ count, double numel, float* __restrict__ scale) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { const unsigned int c = *count; *scale = (float)(numel / (double)(c > 0 ? c : 1)); }
}

extern "C" int subreg_mask_scale(const unsigned int* kept_count, long long numel, float* scale, void* stream) {
    SUBREG_CHECK_ARG(kept_count && scale && numel > 0);
    hipLaunchKernelGGL(mask_scale_kernel, 1, 64, 0, (hipStream_t)stream, kept_count, (double)numel, scale);
    return launch_status();
}

extern "C" int subreg_dropblock_mask(const unsigned char* sample, unsigned char* keep_nhwc, int B, int C, int H, int W, int block_size,
                                     unsigned int* kept_count, void* stream) {
    SUBREG_CHECK_ARG(sample && keep_nhwc && kept_count && B > 0 && C > 0 && block_size >= 1 && H >= block_size && W >= block_size);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * C * H * W;
    if (hipMemsetAsync(keep_nhwc, 1, n, s) != hipSuccess || hipMemsetAsync(kept_count, 0, sizeof(unsigned int), s) != hipSuccess)
        return SUBREG_EHIP;
    hipLaunchKernelGGL(dropblock_mask_kernel, 1, 1024, 0, s, sample, keep_nhwc, B, C, H, W, block_size);
    size_t blocks = (n + 256 * 64 - 1) / (256 * 64);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_ones_kernel, dim3((unsigned)blocks), dim3(256), 0, s, keep_nhwc, n, kept_count);
    return launch_status();
}

extern "C" int subreg_avgpool(const void* x, float* feat, int B, int H, int W, int C, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && feat && B > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(avgpool_kernel<float>, ew_blocks(n), EW_THREADS, 0, s, (const float*)x, feat, B, H * W, C),
               hipLaunchKernelGGL(avgpool_kernel<__bf16>, ew_blocks(n), EW_THREADS, 0, s, (const __bf16*)x, feat, B, H * W, C));
    return launch_status();
}
