// Backward kernels of one pretraining step (train_supervised.py:205-268: loss.backward() through
// models/resnet_language.py BasicBlock.forward :268-301), gfx950.
//
//   block_tail_bwd   dropout/DropBlock mask, MaxPool2d(2) (first maximum of the window), LeakyReLU' of the block output
//   bn_bwd_*         BatchNorm2d training-mode backward (d gamma, d beta, d x) with the preceding LeakyReLU' fused in
//   conv_wgrad       dW[o][tap][c] = sum_p dY[p][o] * X[p+off(tap)][c]   (exact-f32 MFMA, fp32 atomics over K splits)
//   (dX of a conv is the forward kernel run on dY with flipped/transposed weights: pack_conv_weight_dgrad)
//   avgpool_bwd, unpack_wgrad, sgd_momentum
// Round 1: correctness first.  conv_wgrad runs on the exact-f32 MFMA (v_mfma_f32_32x32x2_f32) for BOTH dtypes with
// operands fetched straight from global memory (bf16 operands are widened on the fly); a bf16-MFMA version with LDS
// transposing reads is the next step.
#include "conv_index.h"
#include <stdlib.h>

#include <math.h>

#include "subreg_common.h"

namespace subreg {

constexpr int BW_THREADS = 256;
static inline int bw_blocks(size_t n) { return (int)((n + BW_THREADS - 1) / BW_THREADS); }

__device__ __forceinline__ float lrelu_grad(float pre) { return pre > 0.f ? 1.f : 0.1f; }

// ---------------------------------------------------------------- block tail
// out = mask*scale * pool(lrelu(v)), v = raw3*sc3+sh3 + (res*rsc+rsh | res).  Thread = (channel group of 16 bytes, pixel
// lane) walking OUTPUT pixels (coefficients loaded once, 16-byte loads and stores).  A pooled output writes all four pixels
// of its window - dV at the first maximum in scan order, zeros at the other three - and the windows at the right / bottom
// edge also zero the column / row that floor pooling drops, so dV needs no separate zero-fill pass.
// STATS: the kernel also emits the reduce pass of the two BatchNorms that consume dV (bn3 over raw3; the shortcut's BatchNorm over
// `res` when stat.mean_d is set): partial[block][C][2] = (sum dV, sum dV * xhat) in fp64 across the block's pixel lanes, exactly what
// bn_bwd_reduce_kernel would compute from dV and the raw tensors - which this kernel has in registers already (dV is non-zero at the
// window's first maximum only, so one (raw3, res) pair per output element is all the sums need).  Two launches and four tensor reads
// less per block of the backward.
struct TailStats {
    const float* mean3; const float* invstd3; double* partial3;
    const float* mean_d; const float* invstd_d; double* partial_d;      // mean_d == NULL: no shortcut BatchNorm
};
template <typename T, bool POOL, bool STATS>
__global__ __launch_bounds__(256) void block_tail_bwd_kernel(const T* __restrict__ gout, const unsigned char* __restrict__ keep,
                                                             float mask_scale_host, const float* __restrict__ mask_scale_dev,
                                                             const T* __restrict__ raw3,
                                                             const float* __restrict__ sc3, const float* __restrict__ sh3,
                                                             const T* __restrict__ res, const float* __restrict__ rsc,
                                                             const float* __restrict__ rsh, T* __restrict__ dv, int H, int W, int C,
                                                             long long npo, int ppb, const TailStats st) {
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char tsm[];
    const float mask_scale = mask_scale_dev ? *mask_scale_dev : mask_scale_host;
    const int ngrp = C / VEC, lanes = 256 / ngrp;
    const int cg = threadIdx.x % ngrp, pl = threadIdx.x / ngrp;
    if (!STATS && pl >= lanes) return;
    const int c0 = cg * VEC;
    float a[VEC], sft[VEC], ra[VEC], rs[VEC];
    float s_g[VEC], s_x3[VEC], s_xd[VEC], m3[VEC], i3[VEC], md[VEC], id[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { a[k] = sc3[c0 + k]; sft[k] = sh3[c0 + k]; ra[k] = rsc ? rsc[c0 + k] : 1.f; rs[k] = rsh ? rsh[c0 + k] : 0.f; }
    if (STATS) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            s_g[k] = s_x3[k] = s_xd[k] = 0.f;
            m3[k] = st.mean3[c0 + k]; i3[k] = st.invstd3[c0 + k];
            md[k] = st.mean_d ? st.mean_d[c0 + k] : 0.f; id[k] = st.mean_d ? st.invstd_d[c0 + k] : 0.f;
        }
    }
    const int Ho = POOL ? H / 2 : H, Wo = POOL ? W / 2 : W, NQ = POOL ? 4 : 1;
    const long long p0 = (long long)blockIdx.x * ppb;
    long long p1 = p0 + ppb;
    if (p1 > npo) p1 = npo;
    for (long long po = p0 + pl; pl < lanes && po < p1; po += lanes) {
        const int wo = (int)(po % Wo), ho = (int)((po / Wo) % Ho);
        const long long b = po / ((long long)Wo * Ho);
        const size_t pin = POOL ? ((size_t)b * H + 2 * ho) * W + 2 * wo : (size_t)po;
        const size_t eo = (size_t)po * C + c0;
        const uint4 vg = *reinterpret_cast<const uint4*>(gout + eo);
        unsigned long long kb = 0x0101010101010101ull;
        if (keep) kb = VEC == 8 ? *reinterpret_cast<const unsigned long long*>(keep + eo) : (unsigned long long)*reinterpret_cast<const unsigned*>(keep + eo);
        uint4 vx[NQ], vr[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const size_t e = (pin + (q >> 1) * W + (q & 1)) * C + c0;
            vx[q] = *reinterpret_cast<const uint4*>(raw3 + e);
            vr[q] = *reinterpret_cast<const uint4*>(res + e);
        }
        const T* tg = reinterpret_cast<const T*>(&vg);
        float best[VEC], bx[VEC], br[VEC];
        int arg[VEC];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const T* tx = reinterpret_cast<const T*>(&vx[q]);
            const T* tr = reinterpret_cast<const T*>(&vr[q]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float x3 = ElemTraits<T>::to_float(tx[k]), xr = ElemTraits<T>::to_float(tr[k]);
                const float v = x3 * a[k] + sft[k] + xr * ra[k] + rs[k];
                if (q == 0 || v > best[k]) { best[k] = v; arg[k] = q; if (STATS) { bx[k] = x3; br[k] = xr; } }   // first maximum in scan order
            }
        }
        float d[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            float g = ElemTraits<T>::to_float(tg[k]);
            if (keep) g = ((kb >> (8 * k)) & 0xff) ? g * mask_scale : 0.f;
            d[k] = g * lrelu_grad(best[k]);
            if (STATS) {
                const float dr = ElemTraits<T>::to_float(ElemTraits<T>::from_float(d[k]));     // the value the BatchNorm passes will read back
                s_g[k] += dr;
                s_x3[k] += dr * (bx[k] - m3[k]) * i3[k];
                s_xd[k] += dr * (br[k] - md[k]) * id[k];
            }
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            uint4 vo;
            T* to = reinterpret_cast<T*>(&vo);
#pragma unroll
            for (int k = 0; k < VEC; ++k) to[k] = ElemTraits<T>::from_float((!POOL || arg[k] == q) ? d[k] : 0.f);
            *reinterpret_cast<uint4*>(dv + (pin + (q >> 1) * W + (q & 1)) * C + c0) = vo;
        }
        if (POOL) {
            // floor pooling: an odd last column / row belongs to no window and receives no gradient
            const uint4 z = make_uint4(0, 0, 0, 0);
            const bool ec = (W & 1) && wo == Wo - 1, er = (H & 1) && ho == Ho - 1;
            if (ec) {
                *reinterpret_cast<uint4*>(dv + (pin + 2) * C + c0) = z;
                *reinterpret_cast<uint4*>(dv + (pin + W + 2) * C + c0) = z;
            }
            if (er) {
                *reinterpret_cast<uint4*>(dv + (pin + 2 * (size_t)W) * C + c0) = z;
                *reinterpret_cast<uint4*>(dv + (pin + 2 * (size_t)W + 1) * C + c0) = z;
                if (ec) *reinterpret_cast<uint4*>(dv + (pin + 2 * (size_t)W + 2) * C + c0) = z;
            }
        }
    }
    if (STATS) {
        float* const l1 = reinterpret_cast<float*>(tsm);           // [lanes][C] x 3
        float* const l2 = l1 + (size_t)lanes * C;
        float* const l3 = l2 + (size_t)lanes * C;
        if (pl < lanes) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) { l1[pl * C + c0 + k] = s_g[k]; l2[pl * C + c0 + k] = s_x3[k]; l3[pl * C + c0 + k] = s_xd[k]; }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
            for (int l = 0; l < lanes; ++l) { t1 += (double)l1[l * C + c]; t2 += (double)l2[l * C + c]; t3 += (double)l3[l * C + c]; }
            st.partial3[((size_t)blockIdx.x * C + c) * 2] = t1;
            st.partial3[((size_t)blockIdx.x * C + c) * 2 + 1] = t2;
            if (st.mean_d) {
                st.partial_d[((size_t)blockIdx.x * C + c) * 2] = t1;
                st.partial_d[((size_t)blockIdx.x * C + c) * 2 + 1] = t3;
            }
        }
    }
}

// ---------------------------------------------------------------- BN backward
// g = dy * lrelu'(act) (act == NULL: g = dy); xhat = (raw - mean) * invstd.
// reduce: partial[slice][C][2] = (sum g, sum g*xhat) over the slice's pixels.  block (32 channels x 8 pixel lanes).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dy, const T* __restrict__ act,
                                                            const T* __restrict__ raw, const float* __restrict__ mean,
                                                            const float* __restrict__ invstd, double* __restrict__ partial,
                                                            long long npix, int C, int pix_per_slice) {
    // 16-byte vector loads: thread = (channel group of VEC channels, pixel lane); per-thread fp32 partials over <= ~100
    // pixels, then fp64 across the pixel lanes and slices like the reference's CPU batch_norm backward
    // (acc_type<float> = double): d beta of a BN that feeds another BN is a sum with near-total cancellation.
    constexpr int VEC = 16 / sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char bsm[];
    const int ngrp = C / VEC;                      // channel groups per pixel row
    const int lanes = 256 / ngrp;                  // pixel lanes of this block (threads beyond lanes*ngrp idle)
    const int tid = threadIdx.x, cg = tid % ngrp, pl = tid / ngrp;
    const long long p0 = (long long)blockIdx.x * pix_per_slice;
    long long p1 = p0 + pix_per_slice;
    if (p1 > npix) p1 = npix;
    float a1[VEC], a2[VEC], m[VEC], is[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) { a1[k] = 0.f; a2[k] = 0.f; m[k] = mean[cg * VEC + k]; is[k] = invstd[cg * VEC + k]; }
    if (pl < lanes) {
        constexpr int UN = 4;                      // pixels in flight per thread (all loads of a step before its arithmetic)
        for (long long pb = p0 + pl; pb < p1; pb += (long long)lanes * UN) {
            uint4 vd[UN], vr[UN], va[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const long long p = pb + (long long)u * lanes;
                if (p < p1) {
                    const size_t e = (size_t)p * C + cg * VEC;
                    vd[u] = *reinterpret_cast<const uint4*>(dy + e);
                    vr[u] = *reinterpret_cast<const uint4*>(raw + e);
                    if (act) va[u] = *reinterpret_cast<const uint4*>(act + e);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                if (pb + (long long)u * lanes < p1) {
                    const T* td = reinterpret_cast<const T*>(&vd[u]);
                    const T* tr = reinterpret_cast<const T*>(&vr[u]);
                    const T* ta = reinterpret_cast<const T*>(&va[u]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        float g = ElemTraits<T>::to_float(td[k]);
                        if (act) g *= lrelu_grad(ElemTraits<T>::to_float(ta[k]));
                        a1[k] += g;
                        a2[k] += g * (ElemTraits<T>::to_float(tr[k]) - m[k]) * is[k];
                    }
                }
            }
        }
    }
    float* s1 = reinterpret_cast<float*>(bsm);     // [lanes][C] x 2
    float* s2 = s1 + (size_t)lanes * C;
    if (pl < lanes) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) { s1[pl * C + cg * VEC + k] = a1[k]; s2[pl * C + cg * VEC + k] = a2[k]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        double t1 = 0.0, t2 = 0.0;
        for (int l = 0; l < lanes; ++l) { t1 += (double)s1[l * C + c]; t2 += (double)s2[l * C + c]; }
        partial[((size_t)blockIdx.x * C + c) * 2] = t1;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = t2;
    }
}

// block = 8 channels x 32 slice lanes: the slices of a channel are summed by 32 threads (fixed order, fp64), eight 16-byte
// loads in flight each - 512 slices are two round trips (the 16 x 16 rolled form was 32 dependent ones, 11 us per BatchNorm).
// Also emits the three per-channel coefficients of the apply pass:
//   dx = gamma*invstd*(g - dbeta/N - xhat*dgamma/N) = k1*g + k2*raw + k3,
//   k1 = gamma*invstd, k2 = -k1*invstd*dgamma/N, k3 = k1*(mean*invstd*dgamma - dbeta)/N.
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ partial, int slices, int C,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma, double inv_n,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              float* __restrict__ coef) {
    __shared__ double red[2][32][9];
    constexpr int UN = 8;
    const int cl = threadIdx.x & 7, sl = threadIdx.x >> 3, c = blockIdx.x * 8 + cl;
    double t1 = 0.0, t2 = 0.0;
    if (c < C)
        for (int s0 = sl; s0 < slices; s0 += 32 * UN) {
            double2 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int s = s0 + u * 32;
                v[u] = s < slices ? *reinterpret_cast<const double2*>(partial + ((size_t)s * C + c) * 2) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) { t1 += v[u].x; t2 += v[u].y; }
        }
    red[0][sl][cl] = t1;
    red[1][sl][cl] = t2;
    __syncthreads();
    if (sl == 0 && c < C) {
        t1 = 0.0; t2 = 0.0;
        for (int l = 0; l < 32; ++l) { t1 += red[0][l][cl]; t2 += red[1][l][cl]; }
        dbeta[c] = (float)t1;
        dgamma[c] = (float)t2;
        const double is = invstd[c], k1 = (double)gamma[c] * is;
        coef[c] = (float)k1;
        coef[C + c] = (float)(-k1 * is * t2 * inv_n);
        coef[2 * C + c] = (float)(k1 * ((double)mean[c] * is * t2 - t1) * inv_n);
    }
}

// dx = k1*g + k2*raw + k3 with g = dy*lrelu'(act).  Thread = (channel group of 16 bytes, pixel lane): its 8 / 4 channels'
// coefficients are loaded once, then it walks pixels with stride `lanes`, four in flight.  HBM-bound (3 tensors read, 1 written).
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ act,
                                                           const T* __restrict__ raw, const float* __restrict__ coef,
                                                           T* __restrict__ dx, long long npix, int C, int ppb) {
    constexpr int VEC = 16 / sizeof(T), UN = 4;
    const int ngrp = C / VEC, lanes = 256 / ngrp;
    const int cg = threadIdx.x % ngrp, pl = threadIdx.x / ngrp;
    if (pl >= lanes) return;
    const int c0 = cg * VEC;
    float k1[VEC], k2[VEC], k3[VEC];
#pragma unroll
    for (int k = 0; k < VEC; k += 4) {
        *reinterpret_cast<float4*>(k1 + k) = *reinterpret_cast<const float4*>(coef + c0 + k);
        *reinterpret_cast<float4*>(k2 + k) = *reinterpret_cast<const float4*>(coef + C + c0 + k);
        *reinterpret_cast<float4*>(k3 + k) = *reinterpret_cast<const float4*>(coef + 2 * C + c0 + k);
    }
    const long long p0 = (long long)blockIdx.x * ppb;
    long long p1 = p0 + ppb;
    if (p1 > npix) p1 = npix;
    for (long long pb = p0 + pl; pb < p1; pb += (long long)lanes * UN) {
        uint4 vd[UN], vr[UN], va[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = pb + (long long)u * lanes;
            if (p < p1) {
                const size_t e = (size_t)p * C + c0;
                vd[u] = *reinterpret_cast<const uint4*>(dy + e);
                vr[u] = *reinterpret_cast<const uint4*>(raw + e);
                if (act) va[u] = *reinterpret_cast<const uint4*>(act + e);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const long long p = pb + (long long)u * lanes;
            if (p < p1) {
                const T* td = reinterpret_cast<const T*>(&vd[u]);
                const T* tr = reinterpret_cast<const T*>(&vr[u]);
                const T* ta = reinterpret_cast<const T*>(&va[u]);
                uint4 vo;
                T* to = reinterpret_cast<T*>(&vo);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    float g = ElemTraits<T>::to_float(td[k]);
                    if (act) g *= lrelu_grad(ElemTraits<T>::to_float(ta[k]));
                    to[k] = ElemTraits<T>::from_float(k1[k] * g + (k2[k] * ElemTraits<T>::to_float(tr[k]) + k3[k]));
                }
                *reinterpret_cast<uint4*>(dx + (size_t)p * C + c0) = vo;
            }
        }
    }
}

// ---------------------------------------------------------------- AdaptiveAvgPool2d(1) backward
template <typename T>
__global__ void avgpool_bwd_kernel(const float* __restrict__ dfeat, T* __restrict__ dx, int B, int HW, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * HW * C) return;
    const int c = i % C, b = i / ((size_t)HW * C);
    dx[i] = ElemTraits<T>::from_float(dfeat[(size_t)b * C + c] / (float)HW);
}

// ---------------------------------------------------------------- weight gradient (exact-f32 MFMA)
// One wave: output tile 32 (o) x 32 (c) for all taps over a pixel range; C[o][c] += A[o][k] B[k][c] with k = pixel:
// lane (lr, lh) supplies A = dY[p+lh][o0+lr] and B = X[p+lh+off(tap)][c0+lr] (0 outside the image) - both coalesced.
template <typename T, int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                         float* __restrict__ gw, ConvGeom g, int Cin, int Cout,
                                                         int pix_per_wave) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, lr = lane & 31, lh = lane >> 5;
    const int n_ct = Cin / 32;
    const int o0 = (blockIdx.x / n_ct) * 32, c0 = (blockIdx.x % n_ct) * 32;
    const long long pstart = ((long long)blockIdx.y * 4 + wid) * pix_per_wave;
    if (pstart >= g.npix) return;
    long long pend = pstart + pix_per_wave;
    if (pend > g.npix) pend = g.npix;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // this lane walks pixels pstart+lh, +2, +4, ...
    long long p = pstart + lh;
    int rem = (int)(p % ((long long)g.H * g.W));
    int h = rem / g.W, w = rem % g.W;
    for (; p - lh < pend; p += 2) {
        const bool pv = p < pend;
        const float a = pv ? ElemTraits<T>::to_float(dy[(size_t)p * Cout + o0 + lr]) : 0.f;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int ddy = TAPS == 9 ? t / 3 - 1 : 0, ddx = TAPS == 9 ? t % 3 - 1 : 0;
            const bool ok = pv && tap_valid(g, h, w, ddy, ddx);
            const float b = ok ? ElemTraits<T>::to_float(x[(size_t)(p + ddy * g.W + ddx) * Cin + c0 + lr]) : 0.f;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        }
        w += 2;
        while (w >= g.W) { w -= g.W; if (++h >= g.H) h = 0; }
    }
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&gw[((size_t)o * TAPS + t) * Cin + c0 + lr], acc[t][r]);
        }
}

// ---------------------------------------------------------------- weight gradient, bf16 MFMA
// Zero-bordered copies: compact [B][H][W][C] -> padded [B][H+2][W+2][C]; in padded coordinates q a tap is the constant
// row offset dy*(W+2)+dx and every out-of-image product vanishes because one of its factors is a border zero.
__global__ void pad_copy_kernel(const __bf16* __restrict__ x, __bf16* __restrict__ y, int B, int H, int W, int C) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // over padded elements / 8 (16-byte vectors)
    const int C8 = C / 8;
    if (i >= (size_t)B * (H + 2) * (W + 2) * C8) return;
    const int c8 = i % C8;
    const size_t q = i / C8;
    const int w = q % (W + 2), h = (q / (W + 2)) % (H + 2), b = q / ((size_t)(W + 2) * (H + 2));
    uint4 v = make_uint4(0, 0, 0, 0);
    if (h >= 1 && h <= H && w >= 1 && w <= W)
        v = *reinterpret_cast<const uint4*>(x + (((size_t)b * H + (h - 1)) * W + (w - 1)) * C + c8 * 8);
    *reinterpret_cast<uint4*>(y + q * C + c8 * 8) = v;
}

// one LDS-DMA piece whose 64 lanes read from arbitrary 64-bit addresses (the address lives in a VGPR pair, no scalar base)
__device__ __forceinline__ void dma16_far(const char* p, unsigned lds_addr) {
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(p), "s"(lds_u)
        : "memory");
}

// ds_read_b64_tr_b16: per 16-lane group a 4 (rows = pixels) x 16 (columns = channels) block is delivered transposed:
// lane i of the group gets column i, element e = row e.  Lane 4*qr+pc supplies the address of row qr, columns 4pc..4pc+3.
__device__ __forceinline__ uint2 tr_read(unsigned addr) {
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(r) : "v"(addr));
    return r;
}
// the same with a compile-time byte offset in the instruction (saves the v_add of "four rows further")
template <int OFF>
__device__ __forceinline__ uint2 tr_read_off(unsigned addr) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit DS offset");
    uint2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
    return r;
}

// One workgroup = NCW waves: output channels [o0, o0+32) x input channels [c0, c0+32*NCW), all TAPS, over a range of padded
// pixel rows.  Both operands are pixel-major in memory and in LDS; the MFMA wants 8 consecutive K (= pixels) per lane and
// channel, which the transposing LDS read provides.  dW accumulated with fp32 atomics over the K splits.
template <int NCW, int TAPS>
__global__ __launch_bounds__(NCW * 64, 2) void conv_wgrad_bf16_kernel(const __bf16* __restrict__ xq, const __bf16* __restrict__ dyq,
                                                                    float* __restrict__ gw, long long Q, int Wrow, int Cin,
                                                                    int Cout, int rows_per_block) {
    constexpr int KCH = 64;                              // pixel rows per chunk (4 MFMA k-steps)
    constexpr int NT = NCW * 64;
    constexpr int XS = NCW * 64 + 16, DS = 64 + 16;      // LDS row strides (bytes): 32 channels per wave / tile + 16 B pad
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int halo = TAPS == 9 ? Wrow + 1 : 0;
    const int xrows = KCH + 2 * halo;
    char* const sD = smem;
    char* const sX = smem + KCH * DS;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ncg = Cin / (32 * NCW);
    const int o0 = (blockIdx.x / ncg) * 32, c0 = (blockIdx.x % ncg) * 32 * NCW;
    const long long qb = (long long)blockIdx.y * rows_per_block;
    long long qe = qb + rows_per_block;
    if (qe > Q) qe = Q;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // per-lane part of the transposing-read addresses
    const int li = lane & 15, grp = lane >> 4, qr = li >> 2, pc = li & 3;
    const unsigned a_lane = (unsigned)((8 * (grp >> 1) + qr) * DS + (16 * (grp & 1) + 4 * pc) * 2);
    const unsigned b_lane = (unsigned)((8 * (grp >> 1) + qr) * XS + wid * 64 + (16 * (grp & 1) + 4 * pc) * 2);
    const unsigned sD_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sD;
    const unsigned sX_addr = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sX;
    for (long long q0 = qb; q0 < qe; q0 += KCH) {
        __syncthreads();
        for (int idx = tid; idx < KCH * 4; idx += NT) {                 // dY tile: KCH rows x 64 B
            const int row = idx >> 2, sl = idx & 3;
            const long long q = q0 + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q < qe) v = *reinterpret_cast<const uint4*>(dyq + (size_t)q * Cout + o0 + sl * 8);
            *reinterpret_cast<uint4*>(sD + row * DS + sl * 16) = v;
        }
        for (int idx = tid; idx < xrows * NCW * 4; idx += NT) {         // X tile: (KCH + 2 halo) rows x NCW*64 B
            const int row = idx / (NCW * 4), sl = idx % (NCW * 4);
            const long long q = q0 - halo + row;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q >= 0 && q < Q) v = *reinterpret_cast<const uint4*>(xq + (size_t)q * Cin + c0 + sl * 8);
            *reinterpret_cast<uint4*>(sX + row * XS + sl * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < KCH / 16; ++ks) {
            uint2 a0 = tr_read(sD_addr + a_lane + (16 * ks) * DS), a1 = tr_read(sD_addr + a_lane + (16 * ks + 4) * DS);
            uint2 b0[TAPS], b1[TAPS];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int off = TAPS == 9 ? (t / 3 - 1) * Wrow + (t % 3 - 1) : 0;
                const unsigned base = sX_addr + b_lane + (unsigned)((16 * ks + halo + off) * XS);
                b0[t] = tr_read(base);
                b1[t] = tr_read(base + 4 * XS);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            const uint4 av = make_uint4(a0.x, a0.y, a1.x, a1.y);
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const uint4 bv = make_uint4(b0[t].x, b0[t].y, b1[t].x, b1[t].y);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[t], 0, 0, 0);
            }
        }
    }
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            atomicAdd(&gw[((size_t)o * TAPS + t) * Cin + c0 + wid * 32 + lr], acc[t][r]);
        }
}

template <int NCW, int TAPS>
static int launch_wgrad_bf16(const __bf16* xq, const __bf16* dyq, float* gw, long long Q, int Wrow, int Cin, int Cout,
                             hipStream_t s) {
    const int halo = TAPS == 9 ? Wrow + 1 : 0;
    const size_t lds = (size_t)64 * (64 + 16) + (size_t)(64 + 2 * halo) * (NCW * 64 + 16);
    if (lds > 160 * 1024) return SUBREG_EUNSUPPORTED;
    auto kern = conv_wgrad_bf16_kernel<NCW, TAPS>;
    static std::atomic<unsigned long long> lds_set{0};   // per instantiation
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), 160 * 1024, lds_set)) return rc;
    const int tiles = (Cout / 32) * (Cin / (32 * NCW));
    long long splits = (1536 + tiles - 1) / tiles;                        // ~6 workgroups per CU in total
    long long rpb = ((Q + splits - 1) / splits + 63) / 64 * 64;
    if (rpb < 64) rpb = 64;
    dim3 grid(tiles, (unsigned)((Q + rpb - 1) / rpb));
    hipLaunchKernelGGL(kern, grid, dim3(NCW * 64), lds, s, xq, dyq, gw, Q, Wrow, Cin, Cout, (int)rpb);
    return launch_status();
}

// ---------------------------------------------------------------- weight gradient, bf16 MFMA, 3x3, streaming version
// One WAVE per workgroup: dW tile 32 (o) x 32 (c) x 9 taps (nine 32x32 accumulators) over the padded pixel rows
// [qb, qe) of one K split.  X rows live in an LDS RING (64-byte rows, slot = (q - qbase) mod R): a chunk of 64 pixels
// only stages its 64 NEW rows (the 2*halo rows the taps reach back/forward to are still resident), where the tiled
// kernel above re-reads 64 + 2*halo rows per chunk (3.7x at 84x84).  The first 32 ring rows are mirrored behind the
// ring's end so that a lane's row offset (< 20 rows) never needs a per-lane wrap: one v_add per transposing read.
// Staging is LDS-DMA one chunk ahead (X: 4 pieces, dY: 4 pieces into a double buffer); a single wave needs no barrier,
// only s_waitcnt vmcnt(0) before it reads what it staged.  Each split writes its own partial dW with plain stores
// (float atomics run at 1.3 TB/s chip-wide and were 40 % of this kernel at 1536+ workgroups); unpack sums the splits.
__global__ __launch_bounds__(64, 2) void conv_wgrad3x3_stream_kernel(const __bf16* __restrict__ xq, const __bf16* __restrict__ dyq,
                                                                     float* __restrict__ gw_part, long long Q, int Wrow, int Cin,
                                                                     int Cout, int rows_per_block, int ring_rows) {
    constexpr int KCH = 64, MIRROR = 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int halo = Wrow + 1;
    const int ncg = Cin / 32;
    const int o0 = (blockIdx.x / ncg) * 32, c0 = (blockIdx.x % ncg) * 32;
    const long long qb = (long long)blockIdx.y * rows_per_block;
    long long qe = qb + rows_per_block;
    if (qe > Q) qe = Q;
    const unsigned sX = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned sD = sX + (unsigned)(ring_rows + MIRROR) * 64;
    const int R = ring_rows;
    // ---- staging
    const int prow = lane >> 2, pslot = (lane & 3) * 16;
    const char* const xbase = reinterpret_cast<const char*>(xq + c0);
    const char* const dbase = reinterpret_cast<const char*>(dyq + o0);
    const long long qbase = ((qb - halo) >> 4) << 4;                   // floor to a piece boundary (may be negative)
    auto stage_x = [&](long long q16, int slot16) {                    // piece of rows q16..q16+15 into ring slot slot16
        long long row = q16 + prow;
        row = row < 0 ? 0 : (row >= Q ? Q - 1 : row);                   // outside the tensor: any finite row (its dY factor is a border zero)
        const unsigned voff = (unsigned)row * (unsigned)(Cin * 2) + pslot;
        dma16(xbase, voff, sX + (unsigned)slot16 * 64);
        if (slot16 < MIRROR) dma16(xbase, voff, sX + (unsigned)(R + slot16) * 64);
    };
    auto stage_d = [&](long long q0, int buf) {
#pragma unroll
        for (int j = 0; j < KCH / 16; ++j) {
            const long long row = q0 + 16 * j + prow;
            const unsigned srow = row < qe ? (unsigned)row : 0u;       // beyond the split: row 0 of padded dY, a zero border row
            dma16(dbase, srow * (unsigned)(Cout * 2) + pslot, sD + (unsigned)buf * (KCH * 64) + j * 1024);
        }
    };
    long long fq = qbase;                                               // staging frontier (absolute row) ...
    int fslot = 0;                                                      // ... and its ring slot
    const long long f0 = ((qb + KCH + halo + 15) >> 4) << 4;
    for (; fq < f0; fq += 16) {
        stage_x(fq, fslot);
        fslot += 16;
        if (fslot >= R) fslot -= R;
    }
    stage_d(qb, 0);
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // per-lane part of the transposing-read addresses (see tr_read): 64-byte rows
    const int li = lane & 15, grp = lane >> 4, qr = li >> 2, pc = li & 3;
    const unsigned lane_part = (unsigned)((8 * (grp >> 1) + qr) * 64 + (16 * (grp & 1) + 4 * pc) * 2);
    int base_slot = (int)(qb - halo - qbase);                           // ring slot of row q0 - halo (0..15 at the start)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- software-pipelined fragment reads.  A chunk is 36 "elements" e = 9*ks + t (k-step ks of 16 pixels, tap t): one MFMA each,
    //      fed by two transposing reads of X (+ two of dY when t == 0).  Element e + 4 is read while element e multiplies: its
    //      registers are ring slot (e + 4) % 6 - those of element e - 2, whose MFMA issued two MFMAs ago - and the dY fragments
    //      alternate between two sets.  LDS returns in order, so "at most 8 reads outstanding" = element e has landed.  (Reading
    //      all 20 fragments of a k-step, waiting, then issuing its 9 MFMAs left the matrix pipe idle for every LDS round trip:
    //      at the 4 waves per CU the two-stream schedule runs this kernel with nothing else hides them.)
    constexpr int NE = (KCH / 16) * 9, AHEAD = 4, RING = 6;
    static_assert(NE % RING == 0, "ring slots must repeat per chunk");
    uint2 fa[2][2], fb0[RING], fb1[RING];
    auto issue = [&](int e, unsigned dbuf_l, int bslot) {             // e: compile-time after unrolling
        const int ks = e / 9, t = e % 9;
        if (t == 0) {
            fa[ks & 1][0] = tr_read(dbuf_l + (16 * ks) * 64);
            fa[ks & 1][1] = tr_read(dbuf_l + (16 * ks + 4) * 64);
        }
        int slot = bslot + 16 * ks + halo + (t / 3 - 1) * Wrow + (t % 3 - 1);       // wave-uniform: scalar ALU
        if (slot >= R) slot -= R;
        const unsigned ad = sX + (unsigned)slot * 64 + lane_part;
        fb0[e % RING] = tr_read(ad);
        fb1[e % RING] = tr_read(ad + 4 * 64);
    };
    auto wait_lgkm = [&](int n) {
        switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
        default: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
        }
    };
    {
        const unsigned dbuf0 = sD + lane_part;
#pragma unroll
        for (int e = 0; e < AHEAD; ++e) issue(e, dbuf0, base_slot);
    }
    int it = 0;
    for (long long q0 = qb; q0 < qe; q0 += KCH, ++it) {
        const bool more = q0 + KCH < qe;
        if (more) {                                                     // stage the next chunk while this one computes
#pragma unroll
            for (int k = 0; k < KCH / 16; ++k) {
                stage_x(fq, fslot);
                fq += 16;
                fslot += 16;
                if (fslot >= R) fslot -= R;
            }
            stage_d(q0 + KCH, (it + 1) & 1);
        }
        const unsigned dbuf = sD + (unsigned)(it & 1) * (KCH * 64) + lane_part;
        const unsigned dbuf_n = sD + (unsigned)((it + 1) & 1) * (KCH * 64) + lane_part;
        int bslot_n = base_slot + KCH;
        if (bslot_n >= R) bslot_n -= R;
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (i + AHEAD < NE) {
                issue(i + AHEAD, dbuf, base_slot);
                wait_lgkm(2 * AHEAD);
            } else if (more) {
                if (i + AHEAD == NE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next chunk landed (staged NE - AHEAD MFMAs ago)
                issue(i + AHEAD - NE, dbuf_n, bslot_n);
                wait_lgkm(2 * AHEAD);
            } else {
                wait_lgkm(2 * (NE - 1 - i));                            // last chunk: nothing left to read ahead
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ks = i / 9;
            const uint4 av = make_uint4(fa[ks & 1][0].x, fa[ks & 1][0].y, fa[ks & 1][1].x, fa[ks & 1][1].y);
            const uint4 bv = make_uint4(fb0[i % RING].x, fb0[i % RING].y, fb1[i % RING].x, fb1[i % RING].y);
            acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[i % 9], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        base_slot = bslot_n;
    }
    float* const out = gw_part + (size_t)blockIdx.y * ((size_t)Cout * 9 * Cin);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((size_t)o * 9 + t) * Cin + c0 + lr] = acc[t][r];
        }
}

// ---------------------------------------------------------------- the same, FOUR waves per workgroup sharing the staged rows
// The one-wave kernel above streams 128 bytes per pixel (32 channels of X, 32 of dY) for its 32 x 32 x 9 tile, and every 32-channel
// slice of X / dY is re-read by Cout/32 / Cin/32 workgroups: 472 MB for one 640 x 640 layer at B = 64, arriving at 3-4.7 TB/s on
// every layer shape - the kernel is bound by that operand stream (tools/bench_wgrad.py: 30 % fewer LDS reads or software-pipelined
// reads change nothing; profiles/r03_ab_wgrad_tile.txt).  Here a workgroup of 2 x 2 waves owns 64 output x 64 input channels: X rows
// are 128 bytes (both input-channel halves) in ONE ring, dY rows 128 bytes in one double buffer, each staged once per workgroup by
// LDS-DMA (8-row pieces, two X + two dY pieces per wave and chunk) - half the bytes per MFMA.  One barrier per 64-pixel chunk,
// placed where the register pipeline crosses into the next chunk.  Ragged channel counts (160 = 64 + 64 + 32): the waves beyond
// Cout / Cin stage and synchronise but neither read fragments nor multiply; their half of a row is a duplicate of a valid half.
#ifndef WGRAD_DIAG
#define WGRAD_DIAG 0                    // timing experiments only (wrong results): 1 no MFMAs, 2 no fragment reads, 3 no LDS-DMA staging
#endif
#ifndef WGRAD_TILE_DEPTH
#define WGRAD_TILE_DEPTH 1
#endif
#ifndef WGRAD_TILE_AHEAD
#define WGRAD_TILE_AHEAD 4              // fragment elements read ahead of the MFMA that uses them (ring = AHEAD + 2; 36 % ring == 0)
#endif
// DIRECT: X and dY are the COMPACT [B][H][W][C] tensors; the staging lanes map their padded row (image, h + 1, w + 1) to the compact
// row or, on the one-pixel border, to a line of zeros - the two pad_copy launches per convolution (5 % of the step's kernel time,
// 2 x the tensor in traffic) are gone.  Two divisions per staged row by multiply-shift with host-made reciprocals (exact for
// Q < 2^26): ~14 VALU instructions per LDS-DMA, four to six DMAs per wave and 36-MFMA chunk.
struct WgradGeom {
    int H, W, HpWp, Wp;                   // image size, padded plane and row lengths
    unsigned long long m_plane, m_row;    // ceil(2^40 / HpWp), ceil(2^40 / Wp)
};
__device__ __attribute__((aligned(16))) const unsigned char wgrad_zero_line[128] = {0};

template <bool DIRECT>
__global__ __launch_bounds__(256, 2) void conv_wgrad3x3_tile_kernel(const __bf16* __restrict__ xq, const __bf16* __restrict__ dyq,
                                                                    float* __restrict__ gw_part, long long Q, int Wrow, int Cin,
                                                                    int Cout, int rows_per_block, int ring_rows, const WgradGeom gm) {
    constexpr int KCH = 64, MIRROR = 32, XS = 128, DS = 128, NWAVE = 4;
    constexpr int DEPTH = WGRAD_TILE_DEPTH, NDBUF = DEPTH + 1;           // chunks staged ahead of the one being multiplied; dY buffers
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wave_o = wid >> 1, wave_c = wid & 1;
    const int halo = Wrow + 1;
    const int ncg = (Cin + 63) / 64;
    const int ob = (blockIdx.x / ncg) * 64, cb = (blockIdx.x % ncg) * 64;       // the workgroup's channel origin
    const int o0 = ob + 32 * wave_o, c0 = cb + 32 * wave_c;
    const bool active = o0 < Cout && c0 < Cin;
    const long long qb = (long long)blockIdx.y * rows_per_block;
    long long qe = qb + rows_per_block;
    if (qe > Q) qe = Q;
    const unsigned sX = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned sD = sX + (unsigned)(ring_rows + MIRROR) * XS;
    const int R = ring_rows;
    // ---- staging: a piece = 8 rows x 128 bytes; lane l supplies row l / 8, bytes 16 * (l % 8) ...: the second 64-byte half of a
    //      row falls back to the first where the tensor has no such channels
    const int prow = lane >> 3, pbyte = (lane & 7) * 16;
    const int xhalf = (cb + 32 + (pbyte >= 64 ? 32 : 0) <= Cin || pbyte < 64) ? pbyte : pbyte - 64;
    const int dhalf = (ob + 32 + (pbyte >= 64 ? 32 : 0) <= Cout || pbyte < 64) ? pbyte : pbyte - 64;
    const char* const xbase = reinterpret_cast<const char*>(xq + cb);
    const char* const dbase = reinterpret_cast<const char*>(dyq + ob);
    const long long qbase = ((qb - halo) >> 4) << 4;                   // floor to a 16-row boundary (may be negative)
    // DIRECT: byte address of padded row q of a compact [B][H][W][C] tensor (+ this lane's 16 bytes), or the zero line on the border
    auto compact_src = [&](unsigned q, const char* base, int C, int halfbyte) -> const char* {
        const unsigned img = (unsigned)(((unsigned long long)q * gm.m_plane) >> 40);
        const unsigned r = q - img * (unsigned)gm.HpWp;
        const unsigned hp = (unsigned)(((unsigned long long)r * gm.m_row) >> 40);
        const unsigned wp = r - hp * (unsigned)gm.Wp;
        const bool inside = hp - 1u < (unsigned)gm.H && wp - 1u < (unsigned)gm.W;
        const size_t crow = ((size_t)img * gm.H + (hp - 1u)) * gm.W + (wp - 1u);
        return inside ? base + crow * (size_t)(C * 2) + halfbyte : reinterpret_cast<const char*>(wgrad_zero_line) + (halfbyte & 63);
    };
    auto stage_x8 = [&](long long q8, int slot8) {                     // rows q8 .. q8+7 into ring slots slot8 .. slot8+7
        long long row = q8 + prow;
        row = row < 0 ? 0 : (row >= Q ? Q - 1 : row);                   // outside the tensor: any finite row (its dY factor is a border zero)
        if (WGRAD_DIAG == 3) return;
        if (DIRECT) {
            const char* src = compact_src((unsigned)row, xbase, Cin, xhalf);
            dma16_far(src, sX + (unsigned)slot8 * XS);
            if (slot8 < MIRROR) dma16_far(src, sX + (unsigned)(R + slot8) * XS);
            return;
        }
        const unsigned voff = (unsigned)row * (unsigned)(Cin * 2) + xhalf;
        dma16(xbase, voff, sX + (unsigned)slot8 * XS);
        if (slot8 < MIRROR) dma16(xbase, voff, sX + (unsigned)(R + slot8) * XS);
    };
    auto stage_d8 = [&](long long q8, int buf, int piece) {
        const long long row = q8 + prow;
        const unsigned srow = row < qe ? (unsigned)row : 0u;           // beyond the split: row 0 of padded dY, a zero border row
        if (WGRAD_DIAG == 3) return;
        if (DIRECT) {
            dma16_far(compact_src(srow, dbase, Cout, dhalf), sD + (unsigned)buf * (KCH * DS) + piece * 1024);
            return;
        }
        dma16(dbase, srow * (unsigned)(Cout * 2) + dhalf, sD + (unsigned)buf * (KCH * DS) + piece * 1024);
    };
    // this wave's share of 64 new X rows starting at absolute row fq (ring slot fslot) and of the dY chunk at q0
    auto stage_chunk = [&](long long fq, int fslot, long long q0, int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int piece = wid + NWAVE * k;                            // 0..7
            int slot = fslot + 8 * piece;
            if (slot >= R) slot -= R;
            stage_x8(fq + 8 * piece, slot);
            stage_d8(q0 + 8 * piece, buf, piece);
        }
    };
    long long fq = qbase;                                               // staging frontier (absolute row) ...
    int fslot = 0;                                                      // ... and its ring slot
    const long long f0 = ((qb + DEPTH * KCH + halo + 15) >> 4) << 4;
    for (; fq < f0; fq += 16) {                                         // prologue: 16 rows per step, two pieces, waves take turns
        const int turn = (int)((fq - qbase) >> 4) & 1;
        if ((wid >> 1) == turn) {
            int slot = fslot + 8 * (wid & 1);
            if (slot >= R) slot -= R;
            stage_x8(fq + 8 * (wid & 1), slot);
        }
        fslot += 16;
        if (fslot >= R) fslot -= R;
    }
#pragma unroll
    for (int dch = 0; dch < DEPTH; ++dch)                                // dY chunks 0 .. DEPTH-1 (rows beyond the split read a zero row)
#pragma unroll
        for (int k = 0; k < 2; ++k) stage_d8(qb + dch * KCH + 8 * (wid + NWAVE * k), dch, wid + NWAVE * k);
    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    // per-lane part of the transposing-read addresses (see tr_read), for 128-byte rows; + this wave's 64-byte half
    const int li = lane & 15, grp = lane >> 4, qr = li >> 2, pc = li & 3;
    const unsigned lane_x = (unsigned)((8 * (grp >> 1) + qr) * XS + (16 * (grp & 1) + 4 * pc) * 2 + 64 * wave_c);
    const unsigned lane_d = (unsigned)((8 * (grp >> 1) + qr) * DS + (16 * (grp & 1) + 4 * pc) * 2 + 64 * wave_o);
    int base_slot = (int)(qb - halo - qbase);                           // ring slot of row q0 - halo (0..15 at the start)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int NE = (KCH / 16) * 9, AHEAD = WGRAD_TILE_AHEAD, RING = AHEAD + 2;   // the register pipeline of the one-wave kernel
    static_assert(NE % RING == 0, "ring slots must repeat per chunk");
    uint2 fa[2][2], fb0[RING], fb1[RING];
    auto issue = [&](int e, unsigned dbuf_l, int bslot) {
        const int ks = e / 9, t = e % 9;
        if (WGRAD_DIAG == 2) return;
        if (t == 0) {
            const unsigned da = dbuf_l + (16 * ks) * DS;
            fa[ks & 1][0] = tr_read(da);
            fa[ks & 1][1] = tr_read_off<4 * DS>(da);
        }
        int slot = bslot + 16 * ks + halo + (t / 3 - 1) * Wrow + (t % 3 - 1);
        if (slot >= R) slot -= R;
        const unsigned ad = sX + (unsigned)slot * XS + lane_x;
        fb0[e % RING] = tr_read(ad);
        fb1[e % RING] = tr_read_off<4 * XS>(ad);
    };
    auto wait_lgkm = [&](int n) {
        switch (n) {
        case 0: asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt lgkmcnt(14)" ::: "memory"); break;
        }
    };
    if (active) {
#pragma unroll
        for (int e = 0; e < AHEAD; ++e) issue(e, sD + lane_d, base_slot);
    }
    int it = 0;
    for (long long q0 = qb; q0 < qe; q0 += KCH, ++it) {
        const bool more = q0 + KCH < qe;
        // stage the chunk DEPTH ahead while this one computes (its DMAs are the youngest: they may stay in flight at the barrier)
        int n_dma = 0;
        if (q0 + DEPTH * KCH < qe) {
            n_dma = 4 + (fslot + 8 * wid < MIRROR || (fslot + 8 * wid >= R && fslot + 8 * wid - R < MIRROR) ? 1 : 0)
                      + (fslot + 8 * (wid + NWAVE) < MIRROR || (fslot + 8 * (wid + NWAVE) >= R && fslot + 8 * (wid + NWAVE) - R < MIRROR) ? 1 : 0);
            stage_chunk(fq, fslot, q0 + DEPTH * KCH, (it + DEPTH) % NDBUF);
            fq += KCH;
            fslot += KCH;
            if (fslot >= R) fslot -= R;
        }
        const unsigned dbuf = sD + (unsigned)(it % NDBUF) * (KCH * DS) + lane_d;
        const unsigned dbuf_n = sD + (unsigned)((it + 1) % NDBUF) * (KCH * DS) + lane_d;
        int bslot_n = base_slot + KCH;
        if (bslot_n >= R) bslot_n -= R;
        auto next_chunk_landed = [&]() {                                // everybody's share of the NEXT chunk has landed (LDS-DMA completes
            switch (DEPTH > 1 ? n_dma : 0) {                            // in issue order: the n_dma youngest belong to the chunk after it)
            case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
            }
            __syncthreads();
        };
        if (!active) {                                                  // (one branch per chunk, not one per element)
            if (more) next_chunk_landed();
            base_slot = bslot_n;
            continue;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            if (i + AHEAD == NE && more) next_chunk_landed();
            if (i + AHEAD < NE) {
                issue(i + AHEAD, dbuf, base_slot);
                wait_lgkm(2 * AHEAD);
            } else if (more) {
                issue(i + AHEAD - NE, dbuf_n, bslot_n);
                wait_lgkm(2 * AHEAD);
            } else {
                wait_lgkm(2 * (NE - 1 - i));
            }
            __builtin_amdgcn_sched_barrier(0);
            const int ks = i / 9;
            const uint4 av = make_uint4(fa[ks & 1][0].x, fa[ks & 1][0].y, fa[ks & 1][1].x, fa[ks & 1][1].y);
            const uint4 bv = make_uint4(fb0[i % RING].x, fb0[i % RING].y, fb1[i % RING].x, fb1[i % RING].y);
            if (WGRAD_DIAG != 1) acc[i % 9] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), acc[i % 9], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        base_slot = bslot_n;
    }
    if (!active) return;
    float* const out = gw_part + (size_t)blockIdx.y * ((size_t)Cout * 9 * Cin);
    const int lr = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((size_t)o * 9 + t) * Cin + c0 + lr] = acc[t][r];
        }
}

// K splits of the streaming kernel: one wave per workgroup, ~1024 waves on the chip (4 per CU).  Every split writes its own
// partial dW, which unpack_wgrad re-reads: at the 2048 waves that fill all 8 wave slots per CU the extra partials cost more
// than the occupancy buys (B = 64 train step on ONE stream, same box: 6.08 ms at 2048, 5.85 at 1536, 5.87 at 1024, 6.11 at
// 3072; profiles/r03_train_step.txt).  With the dW chains on the side stream they share the CUs with the dX convolutions and
// fewer, longer waves win: 4.97 / 4.80 / 4.80 / 4.87 / 5.00 ms at 512 / 768 / 1024 / 1536 / 2048, and 376 / 391 / 402 / 397 / 393
// TFLOP/s at B = 128 (profiles/r03_ab_train_two_streams.txt).  SUBREG_WGRAD_WAVES overrides (measurements).
// SUBREG_WGRAD_1WAVE=1: the one-wave kernel (A/B runs)
static bool wgrad_tile4() {
    static const bool on = [] { const char* e = getenv("SUBREG_WGRAD_1WAVE"); return !(e && e[0] == '1'); }();
    return on;
}

static void wgrad_stream_plan(long long Q, int Cin, int Cout, int* splits, int* rows_per_block) {
    static const int target = [] { const char* e = getenv("SUBREG_WGRAD_WAVES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 1024; }();
    // four-wave kernel: 64 x 64-channel tiles, a quarter of the workgroups for the same number of waves
    const int tiles = wgrad_tile4() ? ((Cout + 63) / 64) * ((Cin + 63) / 64) * 4 : (Cout / 32) * (Cin / 32);
    long long sp = (target + tiles - 1) / tiles;
    long long rpb = ((Q + sp - 1) / sp + 63) / 64 * 64;
    if (rpb < 64) rpb = 64;
    *splits = (int)((Q + rpb - 1) / rpb);
    *rows_per_block = (int)rpb;
}

static bool wgrad_stream_ok(long long Q, int W, int Cin, int Cout) {
    const int ahead = wgrad_tile4() ? WGRAD_TILE_DEPTH : 1;
    const int ring = (64 * (ahead + 1) + 2 * (W + 3) + 32 + 15) / 16 * 16;
    const size_t row = wgrad_tile4() ? 128 : 64, cap = wgrad_tile4() ? 80 * 1024 : 64 * 1024;    // two workgroups per CU either way
    return (size_t)(ring + 32) * row + (size_t)(ahead + 1) * 64 * row <= cap && Q * Cin * 2 < (1LL << 32) && Q * Cout * 2 < (1LL << 32);
}

// packed fp32 [splits][Cout][taps][Cin_k] -> Conv2d.weight.grad OIHW, summing the K splits (mode 1: first-layer K=32
// layout, see pack_weight_kernel).  Block = 32 consecutive PACKED elements x 8 split lanes: coalesced 128-byte reads,
// every element's splits summed by 8 threads in a fixed order (deterministic), one scattered write per element.
__global__ __launch_bounds__(256) void unpack_wgrad_kernel(const float* __restrict__ gw, float* __restrict__ grad, int Cout, int Cin,
                                                           int ks, int mode, int splits) {
    __shared__ float red[8][33];
    const int el = threadIdx.x & 31, sl = threadIdx.x >> 5;
    const size_t i = (size_t)blockIdx.x * 32 + el;
    const int taps = ks * ks;
    const size_t n = mode == 0 ? (size_t)Cout * taps * Cin : (size_t)Cout * 32;
    float v = 0.f;
    if (i < n)
        for (int sp = sl; sp < splits; sp += 8) v += gw[(size_t)sp * n + i];
    red[sl][el] = v;
    __syncthreads();
    if (sl != 0 || i >= n) return;
    v = 0.f;
#pragma unroll
    for (int l = 0; l < 8; ++l) v += red[l][el];
    if (mode == 0) {
        const int c = i % Cin, t = (i / Cin) % taps, o = i / ((size_t)Cin * taps);
        grad[((size_t)o * Cin + c) * taps + t] = v;
    } else {
        const int k = i % 32, o = i / 32;
        if (ks == 3) { if (k < 27) grad[((size_t)o * 3 + k % 3) * 9 + k / 3] = v; }
        else if (k >= 12 && k < 15) grad[(size_t)o * 3 + (k - 12)] = v;
    }
}

// 3x3 form of the same: for one output channel the packed [9][Cin] block and the OIHW [Cin][9] block are the same contiguous
// 9*Cin floats, transposed.  Block = (output channel, chunk of CK input channels): coalesced reads of CK floats per tap and
// split (fixed split order => deterministic), transpose through LDS, ONE contiguous 36*CK-byte write - the element-wise
// kernel above writes 4 bytes every 36 (1.5 TB/s on the 640 x 640 layers, a third of what the streams sustain).
template <int CK>
__global__ __launch_bounds__(256) void unpack_wgrad3x3_kernel(const float* __restrict__ gw, float* __restrict__ grad, int Cout, int Cin,
                                                              int splits) {
    __shared__ float t[9][CK + 1];
    const int o = blockIdx.x, c0 = blockIdx.y * CK;
    const size_t n = (size_t)Cout * 9 * Cin;
    const float* src = gw + ((size_t)o * 9) * Cin + c0;
    for (int e = threadIdx.x; e < 9 * CK; e += 256) {
        const int tap = e / CK, c = e % CK;
        const float* q = src + (size_t)tap * Cin + c;
        float v = 0.f;
#pragma unroll 4
        for (int sp = 0; sp < splits; ++sp) v += q[(size_t)sp * n];
        t[tap][c] = v;
    }
    __syncthreads();
    float* dst = grad + ((size_t)o * Cin + c0) * 9;
    for (int e = threadIdx.x; e < 9 * CK; e += 256) dst[e] = t[e % 9][e / 9];
}

// OIHW fp32 -> dgrad operand: the forward weight layout [taps][Cout/32][Cin][32] T of the transposed conv (output
// channels = Cin, reduction over Cout) with the taps flipped: dX = conv(dY, this)
template <typename T>
__global__ void pack_weight_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int ks) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // over [Cin][taps][Cout]
    const int taps = ks * ks;
    if (i >= (size_t)Cin * taps * Cout) return;
    const int nch = Cout / 32;
    const int o32 = i % 32, c = (i / 32) % Cin, ch = (i / ((size_t)32 * Cin)) % nch, t = i / ((size_t)32 * Cin * nch);
    out[i] = ElemTraits<T>::from_float(w[((size_t)(ch * 32 + o32) * Cin + c) * taps + (taps - 1 - t)]);
}

// Every conv of the backbone in one launch: raw forward layout (pack_weight_kernel, elementwise.hip) and dX layout (above).
// The job table travels as a kernel argument (no device-side table to build or keep).
constexpr int PACK_MAX_JOBS = 32;
struct PackJobs {
    const float* src[PACK_MAX_JOBS];      // OIHW fp32
    void* raw[PACK_MAX_JOBS];             // [taps][Cin/32][Cout][32] (first conv: [Cout][32] im2col rows)
    void* dgrad[PACK_MAX_JOBS];           // [taps][Cout/32][Cin][32], taps flipped; may be null
    int cout[PACK_MAX_JOBS], cin[PACK_MAX_JOBS], ks[PACK_MAX_JOBS], mode[PACK_MAX_JOBS];
    unsigned begin[PACK_MAX_JOBS + 1];    // element ranges of the jobs in the 1-D grid
    int n;
};
template <typename T>
__global__ void pack_train_kernel(const PackJobs jobs) {
    const unsigned g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= jobs.begin[jobs.n]) return;
    int j = 0;
    while (j + 1 < jobs.n && g >= jobs.begin[j + 1]) ++j;
    const size_t i = g - jobs.begin[j];
    const float* __restrict__ w = jobs.src[j];
    const int Cout = jobs.cout[j], Cin = jobs.cin[j], ks = jobs.ks[j], taps = ks * ks;
    if (jobs.mode[j] == 1) {                                   // the 3-channel first layer as K=32 im2col rows (k = 3*tap + c)
        if (i >= (size_t)Cout * 32) return;
        const int k = i % 32, o = i / 32;
        float v = 0.f;
        if (k < 27) {
            const int t = k / 3, c = k % 3;
            if (ks == 3) v = w[((size_t)o * 3 + c) * 9 + t];
            else if (t == 4) v = w[(size_t)o * 3 + c];
        }
        reinterpret_cast<T*>(jobs.raw[j])[i] = ElemTraits<T>::from_float(v);
        return;
    }
    {
        const int nch = Cin / 32;
        const int c32 = i % 32, o = (i / 32) % Cout, ch = (i / ((size_t)32 * Cout)) % nch, t = i / ((size_t)32 * Cout * nch);
        reinterpret_cast<T*>(jobs.raw[j])[i] = ElemTraits<T>::from_float(w[((size_t)o * Cin + ch * 32 + c32) * taps + t]);
    }
    if (jobs.dgrad[j]) {
        const int nch = Cout / 32;
        const int o32 = i % 32, c = (i / 32) % Cin, ch = (i / ((size_t)32 * Cin)) % nch, t = i / ((size_t)32 * Cin * nch);
        reinterpret_cast<T*>(jobs.dgrad[j])[i] = ElemTraits<T>::from_float(w[((size_t)(ch * 32 + o32) * Cin + c) * taps + (taps - 1 - t)]);
    }
}

// SGD(momentum, weight decay) + both re-packings of every conv weight, one 32 x 32-channel tile (all taps) per workgroup.
struct SgdPackJobs {
    float* p[PACK_MAX_JOBS];              // OIHW fp32 master weight (updated in place)
    const float* g[PACK_MAX_JOBS];        // gradient, OIHW
    float* m[PACK_MAX_JOBS];              // momentum buffer, OIHW
    void* raw[PACK_MAX_JOBS];
    void* dgrad[PACK_MAX_JOBS];           // may be null
    int cout[PACK_MAX_JOBS], cin[PACK_MAX_JOBS], ks[PACK_MAX_JOBS], mode[PACK_MAX_JOBS];
    unsigned begin[PACK_MAX_JOBS + 1];    // first workgroup of every job
    int n;
    float lr, momentum, wd;
    int first;
};
template <typename T>
__global__ __launch_bounds__(256) void sgd_pack_train_kernel(const SgdPackJobs jobs) {
    __shared__ float tile[32][32 * 9 + 1];                      // [o][c * taps + t]
    int j = 0;
    while (j + 1 < jobs.n && blockIdx.x >= jobs.begin[j + 1]) ++j;
    const int blk = blockIdx.x - jobs.begin[j], tid = threadIdx.x;
    const int Cout = jobs.cout[j], Cin = jobs.cin[j], ks = jobs.ks[j], taps = ks * ks;
    float* __restrict__ p = jobs.p[j];
    const float* __restrict__ g = jobs.g[j];
    float* __restrict__ m = jobs.m[j];
    auto update = [&](size_t i) -> float {
        const float w = p[i], d = g[i] + jobs.wd * w;
        const float b = jobs.first ? d : jobs.momentum * m[i] + d;
        m[i] = b;
        const float nw = w - jobs.lr * b;
        p[i] = nw;
        return nw;
    };
    if (jobs.mode[j] == 1) {
        // the 3-channel first layer (one workgroup): update, then its K=32 im2col rows (k = 3*tap + c; 1x1: centre tap only)
        float* flat = &tile[0][0];                              // Cout * 3 * taps <= 32 * 289 floats
        const int n = Cout * 3 * taps;
        for (int i = tid; i < n; i += 256) flat[i] = update((size_t)i);
        __syncthreads();
        for (int i = tid; i < Cout * 32; i += 256) {
            const int k = i % 32, o = i / 32;
            float v = 0.f;
            if (k < 27) {
                const int t = k / 3, c = k % 3;
                if (ks == 3) v = flat[(o * 3 + c) * 9 + t];
                else if (t == 4) v = flat[o * 3 + c];
            }
            reinterpret_cast<T*>(jobs.raw[j])[i] = ElemTraits<T>::from_float(v);
        }
        return;
    }
    const int ncb = Cin / 32, o0 = (blk / ncb) * 32, c0 = (blk % ncb) * 32, run = 32 * taps;
    // the update streams 16 bytes per lane where the three tensors allow it (their flat-buffer offsets are multiples of 4 floats for every
    // resnet parameter; checked here, not assumed), and the two packed copies leave as 16-byte vectors: 8 bf16 / 4 f32 consecutive
    // channels of one (tap, row) per lane - 64 lanes = 1 KB contiguous
    const bool vec = (((size_t)p | (size_t)g | (size_t)m) & 15) == 0;
    if (vec) {
        const int run4 = run >> 2;
        for (int e = tid; e < 32 * run4; e += 256) {
            const int o = e / run4, r = (e - o * run4) << 2;        // r = c * taps + t: contiguous in OIHW for a fixed o
            const size_t i = ((size_t)(o0 + o) * Cin + c0) * taps + r;
            const float4 w = *reinterpret_cast<const float4*>(p + i), gr = *reinterpret_cast<const float4*>(g + i);
            float4 b;
            if (jobs.first) {
                b = make_float4(gr.x + jobs.wd * w.x, gr.y + jobs.wd * w.y, gr.z + jobs.wd * w.z, gr.w + jobs.wd * w.w);
            } else {
                const float4 mo = *reinterpret_cast<const float4*>(m + i);
                b = make_float4(jobs.momentum * mo.x + (gr.x + jobs.wd * w.x), jobs.momentum * mo.y + (gr.y + jobs.wd * w.y),
                                jobs.momentum * mo.z + (gr.z + jobs.wd * w.z), jobs.momentum * mo.w + (gr.w + jobs.wd * w.w));
            }
            const float4 nw = make_float4(w.x - jobs.lr * b.x, w.y - jobs.lr * b.y, w.z - jobs.lr * b.z, w.w - jobs.lr * b.w);
            *reinterpret_cast<float4*>(m + i) = b;
            *reinterpret_cast<float4*>(p + i) = nw;
            tile[o][r] = nw.x; tile[o][r + 1] = nw.y; tile[o][r + 2] = nw.z; tile[o][r + 3] = nw.w;
        }
    } else {
        for (int e = tid; e < 32 * run; e += 256) {
            const int o = e / run, r = e - o * run;
            tile[o][r] = update(((size_t)(o0 + o) * Cin + c0) * taps + r);
        }
    }
    __syncthreads();
    T* const raw = reinterpret_cast<T*>(jobs.raw[j]);
    T* const dg = reinterpret_cast<T*>(jobs.dgrad[j]);
    constexpr int EPT = 16 / (int)sizeof(T), GPR = 32 / EPT;        // elements per 16-byte vector, vectors per 32-channel row
    for (int e = tid; e < taps * 32 * GPR; e += 256) {
        const int t = e / (32 * GPR), a = (e / GPR) & 31, b0 = (e % GPR) * EPT;
        alignas(16) T v[EPT];
        alignas(16) T u[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            v[k] = ElemTraits<T>::from_float(tile[a][(b0 + k) * taps + t]);     // raw: [tap][Cin/32][Cout][32]: a = o, b = c32
            u[k] = ElemTraits<T>::from_float(tile[b0 + k][a * taps + t]);       // dX: [flipped tap][Cout/32][Cin][32]: a = c, b = o32
        }
        *reinterpret_cast<uint4*>(raw + (((size_t)t * ncb + c0 / 32) * Cout + o0 + a) * 32 + b0) = *reinterpret_cast<const uint4*>(v);
        if (dg) *reinterpret_cast<uint4*>(dg + (((size_t)(taps - 1 - t) * (Cout / 32) + o0 / 32) * Cin + c0 + a) * 32 + b0) = *reinterpret_cast<const uint4*>(u);
    }
}

// torch.optim.SGD (dampening 0, no nesterov): d = g + wd*p; buf = first ? d : m*buf + d; p -= lr*buf
__global__ void sgd_momentum_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf, size_t n,
                                    float lr, float momentum, float wd, int first) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = g[i] + wd * p[i];
    const float b = first ? d : momentum * buf[i] + d;
    buf[i] = b;
    p[i] -= lr * b;
}

// torch.optim.Adam (amsgrad off, L2 weight decay added to the gradient: train_supervised.py:128-131 builds Adam(lr, weight_decay=5e-4)):
// g' = g + wd p; m = b1 m + (1 - b1) g'; v = b2 v + (1 - b2) g'^2; p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float lr, float b1, float b2, float eps, float wd, float step_size, float inv_sqrt_bc2) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float d = g[i] + wd * p[i];
    const float mi = b1 * m[i] + (1.f - b1) * d;
    const float vi = b2 * v[i] + (1.f - b2) * d * d;
    m[i] = mi;
    v[i] = vi;
    p[i] -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
}

// n tensors in one launch: thread i finds its tensor by binary search in the prefix sums
__global__ void sgd_momentum_multi_kernel(float* const* __restrict__ params, float* const* __restrict__ bufs,
                                          const float* __restrict__ gbase, const long long* __restrict__ goff,
                                          const long long* __restrict__ ends, int n, float lr, float momentum, float wd, int first) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ends[n - 1]) return;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (i < ends[mid]) hi = mid; else lo = mid + 1;
    }
    const long long j = i - (lo ? ends[lo - 1] : 0);
    float* p = params[lo];
    float* b = bufs[lo];
    const float d = gbase[goff[lo] + j] + wd * p[j];
    const float m = first ? d : momentum * b[j] + d;
    b[j] = m;
    p[j] -= lr * m;
}

}  // namespace subreg

using namespace subreg;

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16)      \
    if ((dtype) == SUBREG_F32) { CALL_F32; }        \
    else if ((dtype) == SUBREG_BF16) { CALL_BF16; } \
    else return SUBREG_EINVAL;

static int block_tail_launch(const void* grad_out, const unsigned char* keep_mask, float mask_scale,
                             const float* mask_scale_dev, const void* raw3,
                             const float* scale3, const float* shift3, const void* residual, const float* res_scale,
                             const float* res_shift, void* dv, int B, int H, int W, int C, int pool, int dtype,
                             const TailStats* st, int* slices, void* stream) {
    SUBREG_CHECK_ARG(grad_out && raw3 && scale3 && shift3 && residual && dv && B > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t s = (hipStream_t)stream;
    const int vec = dtype == SUBREG_BF16 ? 8 : 4;
    SUBREG_CHECK_ARG(C % vec == 0 && C / vec <= 256 && (!pool || (H >= 2 && W >= 2)));
    const long long npo = (long long)B * (pool ? H / 2 : H) * (pool ? W / 2 : W);
    const int lanes = 256 / (C / vec);
    // ~4096 blocks, whole rounds of the pixel lanes; with the fused statistics at most 512 (a block is a slice of the finalize pass,
    // whose scratch holds subreg_bn_bwd_slices() of them)
    long long ppb = (npo + (st ? 511 : 4095)) / (st ? 512 : 4096);
    ppb = (ppb + lanes - 1) / lanes * lanes;
    if (ppb < lanes) ppb = lanes;
    const int grid = (int)((npo + ppb - 1) / ppb);
    const TailStats none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const size_t lds = st ? (size_t)lanes * C * 3 * sizeof(float) : 0;
#define BTB(TT, P, S) hipLaunchKernelGGL((block_tail_bwd_kernel<TT, P, S>), grid, 256, lds, s, (const TT*)grad_out, keep_mask, mask_scale, mask_scale_dev, (const TT*)raw3, \
                                      scale3, shift3, (const TT*)residual, res_scale, res_shift, (TT*)dv, H, W, C, npo, (int)ppb, st ? *st : none)
#define BTB2(TT) if (st) { if (pool) BTB(TT, true, true); else BTB(TT, false, true); } else { if (pool) BTB(TT, true, false); else BTB(TT, false, false); }
    DISPATCH_T(dtype, BTB2(float), BTB2(__bf16));
#undef BTB2
#undef BTB
    if (slices) *slices = grid;
    return launch_status();
}

extern "C" int subreg_block_tail_bwd(const void* grad_out, const unsigned char* keep_mask, float mask_scale,
                                     const float* mask_scale_dev, const void* raw3,
                                     const float* scale3, const float* shift3, const void* residual, const float* res_scale,
                                     const float* res_shift, void* dv, int B, int H, int W, int C, int pool, int dtype,
                                     void* stream) {
    return block_tail_launch(grad_out, keep_mask, mask_scale, mask_scale_dev, raw3, scale3, shift3, residual, res_scale, res_shift, dv, B, H, W, C,
                             pool, dtype, nullptr, nullptr, stream);
}

extern "C" int subreg_block_tail_bwd_stats(const void* grad_out, const unsigned char* keep_mask, float mask_scale,
                                           const float* mask_scale_dev, const void* raw3,
                                           const float* scale3, const float* shift3, const void* residual, const float* res_scale,
                                           const float* res_shift, void* dv, int B, int H, int W, int C, int pool, int dtype,
                                           const float* mean3, const float* invstd3, double* partial3, const float* mean_d,
                                           const float* invstd_d, double* partial_d, int* slices, void* stream) {
    SUBREG_CHECK_ARG(mean3 && invstd3 && partial3 && slices && ((mean_d != nullptr) == (invstd_d != nullptr)) && (!mean_d || partial_d));
    const TailStats st = {mean3, invstd3, partial3, mean_d, invstd_d, partial_d};
    return block_tail_launch(grad_out, keep_mask, mask_scale, mask_scale_dev, raw3, scale3, shift3, residual, res_scale, res_shift, dv, B, H, W, C,
                             pool, dtype, &st, slices, stream);
}

// slices of the reduce pass + one more slice-sized region of `partial` that holds the apply pass's coefficients
// (at most 512 slices, whatever the problem: see the slice plan in subreg_bn_bwd)
extern "C" int subreg_bn_bwd_slices(long long npix) { (void)npix; return 512 + 1; }

// eval_mode: the statistics are constants (running mean / variance), so d(x) = gamma * invstd * g: the two batch-statistics terms of
// the train-mode formula carry a factor 1 / N, and passing 1 / N = 0 to the finalize pass drops exactly them (k2 = k3 = 0); d(gamma)
// and d(beta) are the same sums over x_hat = (raw - mean) * invstd in both modes
static int bn_bwd_impl(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                       const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C,
                       int dtype, bool eval_mode, void* stream, int pre_slices = 0);

extern "C" int subreg_bn_bwd(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                             const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C,
                             int dtype, void* stream) {
    return bn_bwd_impl(dy, act, raw, mean, invstd, gamma, partial, dgamma, dbeta, dx, npix, C, dtype, false, stream);
}

extern "C" int subreg_bn_bwd_eval(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                                  const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C,
                                  int dtype, void* stream) {
    return bn_bwd_impl(dy, act, raw, mean, invstd, gamma, partial, dgamma, dbeta, dx, npix, C, dtype, true, stream);
}

extern "C" int subreg_bn_bwd_partials(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                                      const float* gamma, double* partial, int slices, float* dgamma, float* dbeta, void* dx,
                                      long long npix, int C, int dtype, int eval_mode, void* stream) {
    SUBREG_CHECK_ARG(slices > 0 && slices < subreg_bn_bwd_slices(npix));
    return bn_bwd_impl(dy, act, raw, mean, invstd, gamma, partial, dgamma, dbeta, dx, npix, C, dtype, eval_mode != 0, stream, slices);
}

static int bn_bwd_impl(const void* dy, const void* act, const void* raw, const float* mean, const float* invstd,
                       const float* gamma, double* partial, float* dgamma, float* dbeta, void* dx, long long npix, int C,
                       int dtype, bool eval_mode, void* stream, int pre_slices) {
    SUBREG_CHECK_ARG(dy && raw && mean && invstd && gamma && partial && dgamma && dbeta && dx && npix > 0 && C > 0);
    hipStream_t s = (hipStream_t)stream;
    // at most 512 slices (two per CU): the finalize pass walks every slice of a channel, and at 1764 slices (64 x 84 x 84
    // pixels) it cost 11 us of dependent fp64 loads per BatchNorm.  At least one unrolled step (4 pixels) per pixel lane and
    // slice: the small maps (1600 pixels x 640 channels in layer 4.1) then still spread over ~130 workgroups instead of 7 -
    // with 256-pixel slices those layers ran at 0.4-1.0 TB/s (profiles/r03_hbm_kernels.txt).
    const int vec_ = dtype == SUBREG_BF16 ? 8 : 4;
    const int lanes_ = C / vec_ > 0 && C / vec_ <= 256 ? 256 / (C / vec_) : 1;
    long long pps = (npix + 511) / 512;
    if (pps < 4LL * lanes_) pps = 4LL * lanes_;
    const int slices = pre_slices > 0 ? pre_slices : (int)((npix + pps - 1) / pps);     // <= 512 (pre_slices: the reduce pass was done by the producer of dy)
    float* const coef = reinterpret_cast<float*>(partial + (size_t)slices * C * 2);     // 3*C floats in the extra slice (4*C)
    const int vec = dtype == SUBREG_BF16 ? 8 : 4;
    SUBREG_CHECK_ARG(C % vec == 0 && C / vec <= 256);
    const size_t lds = (size_t)(256 / (C / vec)) * C * 2 * sizeof(float);
    if (pre_slices <= 0) {
        DISPATCH_T(dtype,
                   hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, slices, 256, lds, s, (const float*)dy, (const float*)act, (const float*)raw, mean, invstd, partial, npix, C, (int)pps),
                   hipLaunchKernelGGL(bn_bwd_reduce_kernel<__bf16>, slices, 256, lds, s, (const __bf16*)dy, (const __bf16*)act, (const __bf16*)raw, mean, invstd, partial, npix, C, (int)pps));
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, (C + 7) / 8, 256, 0, s, partial, slices, C, mean, invstd, gamma,
                       eval_mode ? 0.0 : 1.0 / (double)npix, dgamma, dbeta, coef);
    long long ppb = (npix + 4095) / 4096;                   // apply pass: ~4096 blocks, whole unrolled rounds of the pixel lanes
    ppb = (ppb + 4 * lanes_ - 1) / (4 * lanes_) * (4 * lanes_);
    const int agrid = (int)((npix + ppb - 1) / ppb);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, agrid, 256, 0, s, (const float*)dy, (const float*)act, (const float*)raw, coef, (float*)dx, npix, C, (int)ppb),
               hipLaunchKernelGGL(bn_bwd_apply_kernel<__bf16>, agrid, 256, 0, s, (const __bf16*)dy, (const __bf16*)act, (const __bf16*)raw, coef, (__bf16*)dx, npix, C, (int)ppb));
    return launch_status();
}

extern "C" int subreg_avgpool_bwd(const float* dfeat, void* dx, int B, int H, int W, int C, int dtype, void* stream) {
    SUBREG_CHECK_ARG(dfeat && dx && B > 0 && H > 0 && W > 0 && C > 0);
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * H * W * C;
    DISPATCH_T(dtype, hipLaunchKernelGGL(avgpool_bwd_kernel<float>, bw_blocks(n), BW_THREADS, 0, s, dfeat, (float*)dx, B, H * W, C),
               hipLaunchKernelGGL(avgpool_bwd_kernel<__bf16>, bw_blocks(n), BW_THREADS, 0, s, dfeat, (__bf16*)dx, B, H * W, C));
    return launch_status();
}

extern "C" int subreg_conv_wgrad_splits(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    if (dtype != SUBREG_BF16 || ksize != 3 || B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return 1;
    const long long Q = (long long)B * (H + 2) * (W + 2);
    if (!wgrad_stream_ok(Q, W, Cin, Cout)) return 1;
    int splits, rpb;
    wgrad_stream_plan(Q, Cin, Cout, &splits, &rpb);
    return splits > 1 ? splits : 1;
}

extern "C" int subreg_conv_wgrad(const void* x, const void* dy, float* gw_packed, void* pad_x, void* pad_dy, int B, int H, int W,
                                 int Cin, int Cout, int ksize, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && dy && gw_packed && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    SUBREG_CHECK_ARG((ksize == 1 || ksize == 3) && Cin % 32 == 0 && Cout % 32 == 0);
    hipStream_t s = (hipStream_t)stream;
    const int taps = ksize * ksize;
    const ConvGeom g = make_geom(B, H, W, taps, false);
    const int nsplit = subreg_conv_wgrad_splits(B, H, W, Cin, Cout, ksize, dtype);
    // SUBREG_WGRAD_PADDED=1: the four-wave kernel on zero-bordered copies, as the one-wave kernel needs them (A/B runs)
    static const bool direct_on = [] { const char* e = getenv("SUBREG_WGRAD_PADDED"); return !(e && e[0] == '1'); }();
    const long long Qp = (long long)B * (H + 2) * (W + 2);
    const bool direct = nsplit > 1 && wgrad_tile4() && direct_on && Qp < (1LL << 26);
    if (nsplit > 1 && (direct || (pad_x && pad_dy))) {
        // streaming bf16 3x3 kernel: per-split partial dW, no zero-fill, no atomics
        const long long Q = Qp;
        int splits, rpb;
        wgrad_stream_plan(Q, Cin, Cout, &splits, &rpb);
        if (wgrad_tile4()) {
            const int ring = (64 * (WGRAD_TILE_DEPTH + 1) + 2 * (W + 3) + 32 + 15) / 16 * 16;
            const size_t lds = (size_t)(ring + 32) * 128 + (size_t)(WGRAD_TILE_DEPTH + 1) * 64 * 128;
            dim3 grid(((Cout + 63) / 64) * ((Cin + 63) / 64), splits);
            WgradGeom gm;
            gm.H = H; gm.W = W; gm.HpWp = (H + 2) * (W + 2); gm.Wp = W + 2;
            gm.m_plane = ((1ULL << 40) + gm.HpWp - 1) / gm.HpWp;
            gm.m_row = ((1ULL << 40) + gm.Wp - 1) / gm.Wp;
            if (direct) {
                static std::atomic<unsigned long long> lds_set{0};
                if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad3x3_tile_kernel<true>), 80 * 1024, lds_set)) return rc;
                hipLaunchKernelGGL(conv_wgrad3x3_tile_kernel<true>, grid, dim3(256), lds, s, (const __bf16*)x, (const __bf16*)dy, gw_packed,
                                   Q, W + 2, Cin, Cout, rpb, ring, gm);
                return launch_status();
            }
            hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cin / 8), BW_THREADS, 0, s, (const __bf16*)x, (__bf16*)pad_x, B, H, W, Cin);
            hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cout / 8), BW_THREADS, 0, s, (const __bf16*)dy, (__bf16*)pad_dy, B, H, W, Cout);
            static std::atomic<unsigned long long> lds_set{0};
            if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv_wgrad3x3_tile_kernel<false>), 80 * 1024, lds_set)) return rc;
            hipLaunchKernelGGL(conv_wgrad3x3_tile_kernel<false>, grid, dim3(256), lds, s, (const __bf16*)pad_x, (const __bf16*)pad_dy, gw_packed,
                               Q, W + 2, Cin, Cout, rpb, ring, gm);
            return launch_status();
        }
        hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cin / 8), BW_THREADS, 0, s, (const __bf16*)x, (__bf16*)pad_x, B, H, W, Cin);
        hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cout / 8), BW_THREADS, 0, s, (const __bf16*)dy, (__bf16*)pad_dy, B, H, W, Cout);
        const int ring = (128 + 2 * (W + 3) + 32 + 15) / 16 * 16;
        const size_t lds = (size_t)(ring + 32) * 64 + 2 * 64 * 64;
        dim3 grid((Cout / 32) * (Cin / 32), splits);
        hipLaunchKernelGGL(conv_wgrad3x3_stream_kernel, grid, dim3(64), lds, s, (const __bf16*)pad_x, (const __bf16*)pad_dy, gw_packed,
                           Q, W + 2, Cin, Cout, rpb, ring);
        return launch_status();
    }
    // the kernels below accumulate into copy 0 with atomics; the caller adds all nsplit copies up
    if (hipMemsetAsync(gw_packed, 0, sizeof(float) * (size_t)nsplit * Cout * taps * Cin, s) != hipSuccess) return SUBREG_EHIP;
    if (dtype == SUBREG_BF16 && (taps == 1 || (pad_x && pad_dy))) {
        // bf16 MFMA path: 3x3 on zero-bordered copies (padded pixel coordinates), 1x1 directly on the compact tensors
        const __bf16 *xq = (const __bf16*)x, *dq = (const __bf16*)dy;
        long long Q = g.npix;
        if (taps == 9) {
            Q = (long long)B * (H + 2) * (W + 2);
            hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cin / 8), BW_THREADS, 0, s, (const __bf16*)x, (__bf16*)pad_x, B, H, W, Cin);
            hipLaunchKernelGGL(pad_copy_kernel, bw_blocks((size_t)Q * Cout / 8), BW_THREADS, 0, s, (const __bf16*)dy, (__bf16*)pad_dy, B, H, W, Cout);
            xq = (const __bf16*)pad_x; dq = (const __bf16*)pad_dy;
        }
#define WGB(NCW) (taps == 9 ? launch_wgrad_bf16<NCW, 9>(xq, dq, gw_packed, Q, W + 2, Cin, Cout, s) \
                            : launch_wgrad_bf16<NCW, 1>(xq, dq, gw_packed, Q, W + 2, Cin, Cout, s))
        int rc;
        if (taps == 9) rc = WGB(1);           // 144 accumulator registers per wave: one-wave workgroups keep 8 waves per CU
        else if (Cin % 160 == 0) rc = WGB(5);
        else if (Cin % 64 == 0) rc = WGB(2);
        else rc = WGB(1);
#undef WGB
        if (rc != SUBREG_EUNSUPPORTED) return rc;
    }
    const int tiles = (Cout / 32) * (Cin / 32);
    // enough waves to fill the chip, but at least 64 pixels per wave
    long long ppw = ((long long)g.npix * tiles + 8191) / 8192;
    ppw = ((ppw + 63) / 64) * 64;
    if (ppw < 64) ppw = 64;
    const int ky = (int)((g.npix + ppw * 4 - 1) / (ppw * 4));
    dim3 grid(tiles, ky);
#define WG(TT, TAPS) hipLaunchKernelGGL((conv_wgrad_kernel<TT, TAPS>), grid, 256, 0, s, (const TT*)x, (const TT*)dy, gw_packed, g, Cin, Cout, (int)ppw)
    if (dtype == SUBREG_F32) { if (taps == 9) WG(float, 9); else WG(float, 1); }
    else if (dtype == SUBREG_BF16) { if (taps == 9) WG(__bf16, 9); else WG(__bf16, 1); }
    else return SUBREG_EINVAL;
#undef WG
    return launch_status();
}

extern "C" int subreg_unpack_wgrad(const float* gw_packed, float* grad_oihw, int Cout, int Cin, int ksize, int mode, int splits,
                                   void* stream) {
    SUBREG_CHECK_ARG(gw_packed && grad_oihw && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3) && splits >= 1);
    SUBREG_CHECK_ARG(mode == 0 || (mode == 1 && Cin == 3));
    const size_t n = mode == 0 ? (size_t)Cout * Cin * ksize * ksize : (size_t)Cout * 32;
    // few splits: the scattered 4-byte writes dominate -> transposing kernel; many splits (layers 1-3: 16...370 partial copies):
    // the coalesced split reads dominate and the element-wise kernel spreads them over 8 split lanes
    if (mode == 0 && ksize == 3 && Cin % 32 == 0 && splits <= 8) {
        if (Cin % 160 == 0) hipLaunchKernelGGL(unpack_wgrad3x3_kernel<160>, dim3(Cout, Cin / 160), dim3(256), 0, (hipStream_t)stream, gw_packed, grad_oihw, Cout, Cin, splits);
        else if (Cin % 64 == 0) hipLaunchKernelGGL(unpack_wgrad3x3_kernel<64>, dim3(Cout, Cin / 64), dim3(256), 0, (hipStream_t)stream, gw_packed, grad_oihw, Cout, Cin, splits);
        else hipLaunchKernelGGL(unpack_wgrad3x3_kernel<32>, dim3(Cout, Cin / 32), dim3(256), 0, (hipStream_t)stream, gw_packed, grad_oihw, Cout, Cin, splits);
        return launch_status();
    }
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((unsigned)((n + 31) / 32)), dim3(256), 0, (hipStream_t)stream, gw_packed, grad_oihw,
                       Cout, Cin, ksize, mode, splits);
    return launch_status();
}

extern "C" int subreg_pack_conv_weight_dgrad(const float* w_oihw, void* out, int Cout, int Cin, int ksize, int dtype,
                                             void* stream) {
    SUBREG_CHECK_ARG(w_oihw && out && Cout > 0 && Cin > 0 && (ksize == 1 || ksize == 3));
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)Cout * Cin * ksize * ksize;
    DISPATCH_T(dtype, hipLaunchKernelGGL(pack_weight_dgrad_kernel<float>, bw_blocks(n), BW_THREADS, 0, s, w_oihw, (float*)out, Cout, Cin, ksize),
               hipLaunchKernelGGL(pack_weight_dgrad_kernel<__bf16>, bw_blocks(n), BW_THREADS, 0, s, w_oihw, (__bf16*)out, Cout, Cin, ksize));
    return launch_status();
}

extern "C" int subreg_backbone_pack_train(const subreg_backbone_desc* d, const subreg_train_desc* t, void* stream) {
    SUBREG_CHECK_ARG(d && t && d->blocks && t->blocks && d->n_blocks > 0);
    PackJobs jobs;
    jobs.n = 0;
    unsigned long long total = 0;
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const subreg_conv_desc* cs[4] = {&b.conv1, &b.conv2, &b.conv3, &b.down};
        const subreg_conv_train* ts[4] = {&tb.conv1, &tb.conv2, &tb.conv3, &tb.down};
        for (int k = 0; k < 4; ++k) {
            const subreg_conv_desc& c = *cs[k];
            if (!c.w) continue;
            SUBREG_CHECK_ARG(jobs.n < PACK_MAX_JOBS && c.w_oihw);
            const int j = jobs.n++;
            const bool first = c.cin_raw == 3;
            jobs.src[j] = c.w_oihw; jobs.raw[j] = const_cast<void*>(c.w); jobs.dgrad[j] = first ? nullptr : const_cast<void*>(ts[k]->w_dgrad);
            jobs.cout[j] = c.cout; jobs.cin[j] = c.cin_raw; jobs.ks[j] = c.ksize_raw; jobs.mode[j] = first ? 1 : 0;
            SUBREG_CHECK_ARG(first || (c.cin_raw % 32 == 0 && c.cout % 32 == 0));
            jobs.begin[j] = (unsigned)total;
            total += first ? (unsigned long long)c.cout * 32 : (unsigned long long)c.cout * c.ksize_raw * c.ksize_raw * c.cin_raw;
            SUBREG_CHECK_ARG(total < (1ull << 32));
        }
    }
    jobs.begin[jobs.n] = (unsigned)total;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(d->dtype, hipLaunchKernelGGL(pack_train_kernel<float>, bw_blocks((size_t)total), BW_THREADS, 0, s, jobs),
               hipLaunchKernelGGL(pack_train_kernel<__bf16>, bw_blocks((size_t)total), BW_THREADS, 0, s, jobs));
    return launch_status();
}

extern "C" int subreg_sgd_pack_train(const subreg_backbone_desc* d, const subreg_train_desc* t, const float* grad_base,
                                     const float* grad_origin, float* mom_base, float lr, float momentum, float weight_decay,
                                     int first_step, void* stream) {
    SUBREG_CHECK_ARG(d && t && d->blocks && t->blocks && d->n_blocks > 0 && grad_base && grad_origin && mom_base);
    SgdPackJobs jobs;
    jobs.n = 0;
    jobs.lr = lr; jobs.momentum = momentum; jobs.wd = weight_decay; jobs.first = first_step;
    unsigned blocks = 0;
    for (int i = 0; i < d->n_blocks; ++i) {
        const subreg_block_desc& b = d->blocks[i];
        const subreg_block_train& tb = t->blocks[i];
        const subreg_conv_desc* cs[4] = {&b.conv1, &b.conv2, &b.conv3, &b.down};
        const subreg_conv_train* ts[4] = {&tb.conv1, &tb.conv2, &tb.conv3, &tb.down};
        for (int k = 0; k < 4; ++k) {
            const subreg_conv_desc& c = *cs[k];
            if (!c.w) continue;
            SUBREG_CHECK_ARG(jobs.n < PACK_MAX_JOBS && c.w_oihw && ts[k]->grad_w);
            const int j = jobs.n++;
            const bool first = c.cin_raw == 3;
            const ptrdiff_t off = ts[k]->grad_w - grad_origin;
            jobs.p[j] = const_cast<float*>(c.w_oihw); jobs.g[j] = grad_base + off; jobs.m[j] = mom_base + off;
            jobs.raw[j] = const_cast<void*>(c.w); jobs.dgrad[j] = first ? nullptr : const_cast<void*>(ts[k]->w_dgrad);
            jobs.cout[j] = c.cout; jobs.cin[j] = c.cin_raw; jobs.ks[j] = c.ksize_raw; jobs.mode[j] = first ? 1 : 0;
            SUBREG_CHECK_ARG(first ? c.cout * 27 <= 32 * 289 : (c.cin_raw % 32 == 0 && c.cout % 32 == 0));
            jobs.begin[j] = blocks;
            blocks += first ? 1u : (unsigned)((c.cout / 32) * (c.cin_raw / 32));
        }
    }
    jobs.begin[jobs.n] = blocks;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_T(d->dtype, hipLaunchKernelGGL(sgd_pack_train_kernel<float>, dim3(blocks), dim3(256), 0, s, jobs),
               hipLaunchKernelGGL(sgd_pack_train_kernel<__bf16>, dim3(blocks), dim3(256), 0, s, jobs));
    return launch_status();
}

extern "C" int subreg_sgd_momentum_multi(float* const* params, float* const* momentum_bufs, const float* grad_base,
                                         const long long* grad_offsets, const long long* ends, int n, long long total, float lr,
                                         float momentum, float weight_decay, int first_step, void* stream) {
    SUBREG_CHECK_ARG(params && momentum_bufs && grad_base && grad_offsets && ends && n > 0 && total > 0);
    hipLaunchKernelGGL(sgd_momentum_multi_kernel, bw_blocks((size_t)total), BW_THREADS, 0, (hipStream_t)stream, params, momentum_bufs,
                       grad_base, grad_offsets, ends, n, lr, momentum, weight_decay, first_step);
    return launch_status();
}

extern "C" int subreg_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long long n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, void* stream) {
    SUBREG_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1);
    // bias corrections on the host in double, as torch does (1 - beta^t)
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, bw_blocks((size_t)n), BW_THREADS, 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq, (size_t)n,
                       lr, beta1, beta2, eps, weight_decay, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)));
    return launch_status();
}

extern "C" int subreg_sgd_momentum(float* param, const float* grad, float* momentum_buf, long long n, float lr, float momentum,
                                   float weight_decay, int first_step, void* stream) {
    SUBREG_CHECK_ARG(param && grad && momentum_buf && n > 0);
    hipLaunchKernelGGL(sgd_momentum_kernel, bw_blocks((size_t)n), BW_THREADS, 0, (hipStream_t)stream, param, grad, momentum_buf,
                       (size_t)n, lr, momentum, weight_decay, first_step);
    return launch_status();
}
