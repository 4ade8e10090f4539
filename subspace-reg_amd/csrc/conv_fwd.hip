// Implicit-GEMM convolution forward for gfx950 (MI355X), bf16 and exact-f32 MFMA.
//
// Replaces the cuDNN calls behind models/resnet_language.py:402-405 (conv3x3),
// :146-147 (1x1 shortcut conv) and fuses what follows them in
// BasicBlock.forward (:268-301): eval-mode BatchNorm (:250,253,255 as a folded
// per-channel scale/shift), the residual add (:288), LeakyReLU(0.1) (:251,289)
// and MaxPool2d(2) (:256,290).  In train mode (epoch 1 of every session,
// eval/language_eval.py:211) it writes the raw convolution and per-channel
// partial sums for the batch statistics instead (bn_train.hip finishes the job).
//
// Data layout: activations compact NHWC [B*H*W][C]; weights [Cout][tap][Cin]
// (tap = 3*ky+kx), both in the compute type T (bf16 or f32).  One workgroup
// computes TM rows x TN channels.  Per 32-channel chunk of Cin it stages the
// CONTIGUOUS pixel range that the tile touches through all nine taps (the
// "patch": tile rows +- (W+1) pixels) into LDS once and reuses it for the 9 taps;
// a tap is a constant row offset, image borders are handled by pointing the
// lane's LDS address at a zero row.  LDS rows are 32 channels (64 B bf16 / 128 B
// f32), XOR-swizzled per conv_index.h so that ds_read_b128 is conflict-free.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate      (throughput mode)
//   f32 : v_mfma_f32_32x32x2_f32, bitwise an fmaf chain  (parity mode, 1e-4 gate)
#include "conv_index.h"
#include "subreg_common.h"

namespace subreg {

template <typename T> struct KT;
template <> struct KT<__bf16> {
    static constexpr int ELEM = 2, SLOTS = 4, KSTEPS = 2, ROWB = 64;
};
template <> struct KT<float> {
    static constexpr int ELEM = 4, SLOTS = 8, KSTEPS = 4, ROWB = 128;
};

struct ConvArgs {
    const char* x;       // [npix][Cin] T
    const char* w;       // [Cout][taps][Cin] T
    char* y;             // LINEAR [npix][Cout] T ; POOL [B*Hp*Wp][Cout] T
    const float* scale;  // [Cout] folded BN scale (null when raw)
    const float* shift;  // [Cout]
    const char* res;     // [npix][Cout] T residual or null
    float* stats;        // raw: [gridDim.x*WAVES_M][Cout][2] partial (sum, sumsq)
    ConvGeom g;
    int Cin, Cout;
    int act;             // LeakyReLU(0.1) after scale/shift/residual
    int raw;             // write the un-normalised conv + stats partials
};

template <typename T>
__device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16& acc);

template <>
__device__ __forceinline__ void mma_step<__bf16>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<float>(const uint4& a, const uint4& b, f32x16& acc) {
    // lane (r, h) holds k = 8s + 4h + q, q = 0..3 of row r for both operands: four K=2 steps
    const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0], fb[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1], fb[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2], fb[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3], fb[3], acc, 0, 0, 0);
}

// NI x NJ 32x32 accumulator tiles per wave; WAVES_M x WAVES_N waves per workgroup.
template <typename T, int NI, int NJ, int WAVES_M, int WAVES_N, int TAPS, bool POOL, int AROWS>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, 2 * WAVES_M * WAVES_N / 4) void conv_fwd_kernel(const ConvArgs a) {
    using K = KT<T>;
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int TM = WAVES_M * NI * 32, TN = WAVES_N * NJ * 32;
    constexpr int SLOTS = K::SLOTS, ROWB = K::ROWB, ELEM = K::ELEM;
    constexpr int A_BYTES = (AROWS + 1) * ROWB;        // + one zero row for padded taps
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const sA = smem;
    char* const sB = smem + A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wave_m = wid / WAVES_N, wave_n = wid % WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const ConvGeom g = a.g;
    const int m0 = blockIdx.x * TM, n0 = blockIdx.y * TN;

    int plo, phi;
    patch_range<POOL>(g, m0, TM, &plo, &phi);
    const int prow = phi - plo;

    // zero row (index AROWS) for taps that fall outside the image / rows beyond M
    if (tid < ROWB / 16) *reinterpret_cast<uint4*>(sA + AROWS * ROWB + tid * 16) = make_uint4(0, 0, 0, 0);

    // per-lane LDS addresses of this lane's A rows for every tap (k-step 0; k-step s is addr ^ 32*s)
    int aaddr[NI][TAPS];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int m = m0 + (wave_m * NI + i) * 32 + lr;
        const bool mv = m < g.M;
        const Pix px = row_to_pixel<POOL>(g, mv ? m : 0);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dy = TAPS == 9 ? t / 3 - 1 : 0, dx = TAPS == 9 ? t % 3 - 1 : 0;
            const bool ok = mv && tap_valid(g, px.h, px.w, dy, dx);
            const int row = px.p + dy * g.W + dx - plo;
            const int f = swz<SLOTS>(row);
            aaddr[i][t] = ok ? row * ROWB + 32 * (f >> 1) + 16 * (lh ^ (f & 1)) : AROWS * ROWB + 16 * lh;
        }
    }
    int baddr[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nl = (wave_n * NJ + j) * 32 + lr;
        const int f = swz<SLOTS>(nl);
        baddr[j] = A_BYTES + nl * ROWB + 32 * (f >> 1) + 16 * (lh ^ (f & 1));
    }

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const size_t xrow = (size_t)a.Cin * ELEM;             // bytes per pixel row of x
    const size_t wrow = (size_t)TAPS * a.Cin * ELEM;      // bytes per output channel of w
    const int nchunks = a.Cin / 32;
    for (int c = 0; c < nchunks; ++c) {
        __syncthreads();   // everyone is done reading the previous chunk's patch / last tap's weights
        {
            const char* src = a.x + (size_t)plo * xrow + (size_t)c * 32 * ELEM;
#pragma unroll 4
            for (int idx = tid; idx < prow * SLOTS; idx += NT) {
                const int row = idx / SLOTS, slot = idx % SLOTS;
                const uint4 v = *reinterpret_cast<const uint4*>(src + (size_t)row * xrow + slot * 16);
                *reinterpret_cast<uint4*>(sA + row * ROWB + ((slot ^ swz<SLOTS>(row)) << 4)) = v;
            }
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            if (t > 0) __syncthreads();   // previous tap's weights fully consumed
            {
                const char* src = a.w + (size_t)t * a.Cin * ELEM + (size_t)c * 32 * ELEM;
#pragma unroll
                for (int idx = tid; idx < TN * SLOTS; idx += NT) {
                    const int row = idx / SLOTS, slot = idx % SLOTS;
                    uint4 v = make_uint4(0, 0, 0, 0);
                    if (n0 + row < a.Cout) v = *reinterpret_cast<const uint4*>(src + (size_t)(n0 + row) * wrow + slot * 16);
                    *reinterpret_cast<uint4*>(sB + row * ROWB + ((slot ^ swz<SLOTS>(row)) << 4)) = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int s = 0; s < K::KSTEPS; ++s) {
                uint4 fa[NI], fb[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) fa[i] = *reinterpret_cast<const uint4*>(smem + (aaddr[i][t] ^ (32 * s)));
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[j] = *reinterpret_cast<const uint4*>(smem + (baddr[j] ^ (32 * s)));
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) mma_step<T>(fa[i], fb[j], acc[i][j]);
            }
        }
    }

    // ------------------------------------------------------------------ epilogue
    // C layout of a 32x32 tile: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    T* const y = reinterpret_cast<T*>(a.y);
    const T* const res = reinterpret_cast<const T*>(a.res);
    if (a.raw) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wave_n * NJ + j) * 32 + lr;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int mb = m0 + (wave_m * NI + i) * 32 + 4 * lh;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < g.M && n < a.Cout) {
                        const float v = acc[i][j][r];
                        y[(size_t)m * a.Cout + n] = ElemTraits<T>::from_float(v);
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0 && n < a.Cout) {
                float* dst = a.stats + ((size_t)(blockIdx.x * WAVES_M + wave_m) * a.Cout + n) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + (wave_n * NJ + j) * 32 + lr;
        const bool nv = n < a.Cout;
        const float sc = nv ? a.scale[n] : 0.f, sh = nv ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int mb = m0 + (wave_m * NI + i) * 32 + 4 * lh;
            if (!POOL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < g.M && nv) {
                        float v = acc[i][j][r] * sc + sh;
                        if (res) v += ElemTraits<T>::to_float(res[(size_t)m * a.Cout + n]);
                        if (a.act) v = lrelu(v);
                        y[(size_t)m * a.Cout + n] = ElemTraits<T>::from_float(v);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // register group q: rows mb + 8q + {0,1,2,3} == one 2x2 window
                    const int m = mb + 8 * q;
                    if (m < g.M && nv) {
                        const Pix px = row_to_pixel<true>(g, m);      // top-left pixel of the window
                        float best = -3.0e38f;
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub) {
                            float v = acc[i][j][4 * q + sub] * sc + sh;
                            if (res) {
                                const size_t p = (size_t)px.p + (sub >> 1) * g.W + (sub & 1);
                                v += ElemTraits<T>::to_float(res[p * a.Cout + n]);
                            }
                            best = fmaxf(best, v);
                        }
                        if (a.act) best = lrelu(best);     // monotone => lrelu(max) == max(lrelu)
                        y[(size_t)(m >> 2) * a.Cout + n] = ElemTraits<T>::from_float(best);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- host side
template <typename T, int NI, int NJ, int WM, int WN, int TAPS, bool POOL, int AROWS>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    using K = KT<T>;
    constexpr int TM = WM * NI * 32, TN = WN * NJ * 32;
    // worst-case patch rows over all tiles must fit the LDS patch
    int worst = 0;
    for (int m0 = 0; m0 < a.g.M; m0 += TM) {
        int lo, hi;
        patch_range<POOL>(a.g, m0, TM, &lo, &hi);
        if (hi - lo > worst) worst = hi - lo;
    }
    if (worst > AROWS) return SUBREG_EUNSUPPORTED;
    const size_t lds = (size_t)(AROWS + 1) * K::ROWB + (size_t)TN * K::ROWB;
    auto kern = conv_fwd_kernel<T, NI, NJ, WM, WN, TAPS, POOL, AROWS>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return SUBREG_EHIP;
        attr_done = true;
    }
    dim3 grid((a.g.M + TM - 1) / TM, (a.Cout + TN - 1) / TN);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, a);
    return launch_status();
}

template <typename T, int NI, int NJ, int WM, int WN, int AROWS>
static int launch_shape(const ConvArgs& a, bool pool, hipStream_t s) {
    if (a.g.taps == 9) {
        return pool ? launch_cfg<T, NI, NJ, WM, WN, 9, true, AROWS>(a, s) : launch_cfg<T, NI, NJ, WM, WN, 9, false, AROWS>(a, s);
    }
    return pool ? launch_cfg<T, NI, NJ, WM, WN, 1, true, AROWS>(a, s) : launch_cfg<T, NI, NJ, WM, WN, 1, false, AROWS>(a, s);
}

// rows of stats partials the raw mode writes for a given problem (caller sizes the buffer with this)
static int stats_rows_for(int dtype, int Cout, int M) {
    const int wm = dtype == SUBREG_BF16 ? 4 : 2;
    const int tm = wm * 2 * 32;
    (void)Cout;
    return ((M + tm - 1) / tm) * wm;
}

}  // namespace subreg

using namespace subreg;

extern "C" int subreg_conv_stats_rows(int dtype, int B, int H, int W, int Cout) {
    return stats_rows_for(dtype, Cout, B * H * W);
}

extern "C" int subreg_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* shift,
                               const void* residual, float* stats_partial, int B, int H, int W, int Cin, int Cout,
                               int ksize, int flags, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && w && y);
    SUBREG_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    SUBREG_CHECK_ARG(ksize == 1 || ksize == 3);
    SUBREG_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0);
    SUBREG_CHECK_ARG(dtype == SUBREG_F32 || dtype == SUBREG_BF16);
    const bool raw = flags & SUBREG_CONV_RAW_STATS, pool = flags & SUBREG_CONV_POOL2;
    SUBREG_CHECK_ARG(!(raw && (pool || residual)));
    SUBREG_CHECK_ARG(raw ? stats_partial != nullptr : (scale && shift));
    SUBREG_CHECK_ARG(!pool || (H >= 2 && W >= 2));
    ConvArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.y = (char*)y;
    a.scale = scale; a.shift = shift; a.res = (const char*)residual; a.stats = stats_partial;
    a.g = make_geom(B, H, W, ksize * ksize, pool);
    a.Cin = Cin; a.Cout = Cout;
    a.act = (flags & SUBREG_CONV_LRELU) ? 1 : 0;
    a.raw = raw ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (Cout % 160 == 0);
    if (dtype == SUBREG_BF16) {
        return wide ? launch_shape<__bf16, 2, 5, 4, 1, 704>(a, pool, s) : launch_shape<__bf16, 2, 2, 4, 1, 704>(a, pool, s);
    }
    return wide ? launch_shape<float, 2, 5, 2, 1, 448>(a, pool, s) : launch_shape<float, 2, 2, 2, 1, 448>(a, pool, s);
}
