// Layer-1 3x3 convolutions (Cin = Cout = 64, 84x84 maps: models/resnet_language.py:249-256 conv2 / conv3 of layer1.0) as a
// PERSISTENT implicit GEMM with REGISTER-RESIDENT weights, bf16, eval mode (BN scale folded into the weights, shift +
// LeakyReLU(0.1) + optional MaxPool2d(2) + the fused K=32 shortcut GEMM of conv3 in the epilogue, :268-301).
//
// Why a second kernel for this shape (measurements of the general kernel in profiles/r01_conv_stamps.txt): with K = 576
// and N = 64 a 256-row tile is only 4.6 k MFMA cycles per wave, but it stages 74 KB of weights + 55 KB of activation
// patch and pays a 5.5 k-cycle prologue and a 4.3 k-cycle epilogue per tile - the layer ran at 23-25 % of the MFMA peak
// and is 20 % of the backbone's time.  Here:
//   * one workgroup of 8 waves per CU, looping over m-tiles (persistent): no per-tile prologue, the next tile's patch is
//     prefetched by LDS-DMA while this tile computes;
//   * the WHOLE weight matrix lives in registers: wave (wm, h) owns output rows [64 wm, 64 wm + 64) x columns
//     [32 h, 32 h + 32) of the 256 x 64 tile and keeps the 36 B fragments (9 taps x 2 channel chunks x 2 k-steps) of its 32
//     columns in 144 VGPRs for the whole kernel - no weight staging, no weight LDS reads, ever;
//   * LDS holds only activation planes ([row][32 channels], 64-byte rows, XOR-swizzled like conv_index.h::swz): three
//     planes in rotation (chunk 0 / chunk 1 of this tile, chunk 0 of the next), plus the K=32 shortcut rows of conv3;
//     two workgroup barriers per tile.
// Same data layouts, same row orders (LINEAR / POOL window-major) and the same numerics as conv_fwd.hip: fp32
// accumulation over taps in the same tap order per k-step.
#include "conv_index.h"
#include "subreg_common.h"

namespace subreg {

// n / d for n * d < 2^40 (checked on the host): (n * ceil(2^40 / d)) >> 40
struct FastDiv {
    unsigned long long mul;
    unsigned d;
};
static FastDiv make_fastdiv(int d) {
    FastDiv f;
    f.d = (unsigned)d;
    f.mul = ((1ull << 40) + (unsigned)d - 1) / (unsigned)d;
    return f;
}
__device__ __forceinline__ unsigned fdiv(unsigned n, const FastDiv& f) { return (unsigned)(((unsigned long long)n * f.mul) >> 40); }

struct Conv64Args {
    const char* x;       // [npix][64] bf16
    const char* w;       // [9][2][64][32] bf16, BN scale folded in
    const char* x2;      // fused shortcut GEMM: [npix][32] bf16 or null
    const char* w2;      // [64][32] bf16
    char* y;             // LINEAR [npix][64] ; POOL [B*Hp*Wp][64]
    const float* shift;  // [64]
    int H, W, Hp, Wp, npix, M, act, ntiles;
    FastDiv d_hw, d_w, d_pp, d_wp;   // H*W, W, Hp*Wp, Wp
};

constexpr int R64_TM = 256, R64_ROWB = 64, R64_NW = 8;

template <bool POOL>
__device__ __forceinline__ void r64_pixel(const Conv64Args& a, int m, int& p, int& h, int& w) {
    if (!POOL) {
        const unsigned img = fdiv((unsigned)m, a.d_hw), rem = (unsigned)m - img * a.d_hw.d;
        h = (int)fdiv(rem, a.d_w);
        w = (int)(rem - (unsigned)h * a.d_w.d);
        p = m;
    } else {
        const unsigned win = (unsigned)m >> 2, sub = (unsigned)m & 3;
        const unsigned b = fdiv(win, a.d_pp), rem = win - b * a.d_pp.d;
        const unsigned hp = fdiv(rem, a.d_wp), wp = rem - hp * a.d_wp.d;
        h = (int)(2 * hp + (sub >> 1));
        w = (int)(2 * wp + (sub & 1));
        p = ((int)b * a.H + h) * a.W + w;
    }
}

// patch of tile rows [m0, m0 + 256): contiguous pixel range through all 9 taps ([lo, hi)) and without halo ([cf, cl])
template <bool POOL>
__device__ __forceinline__ void r64_range(const Conv64Args& a, int m0, int& lo, int& hi, int& cf, int& cl) {
    int m1 = m0 + R64_TM;
    if (m1 > a.M) m1 = a.M;
    int h, w;
    if (!POOL) {
        cf = m0; cl = m1 - 1;
    } else {
        r64_pixel<true>(a, m0, cf, h, w);
        r64_pixel<true>(a, (m1 - 1) | 3, cl, h, w);
    }
    lo = cf - (a.W + 1);
    hi = cl + (a.W + 1) + 1;
    if (lo < 0) lo = 0;
    if (hi > a.npix) hi = a.npix;
    // wave-uniform by construction (functions of the tile index); say so: they steer DMA loops and scalar operands
    lo = __builtin_amdgcn_readfirstlane(lo); hi = __builtin_amdgcn_readfirstlane(hi);
    cf = __builtin_amdgcn_readfirstlane(cf); cl = __builtin_amdgcn_readfirstlane(cl);
}

// AROWS: patch rows an LDS plane holds (+ one zero row); XROWS: rows of the shortcut plane (tile rows without halo)
template <bool POOL, bool SC, int AROWS, int XROWS>
__global__ __launch_bounds__(R64_NW * 64, 2) void conv64_resident_kernel(const Conv64Args a) {
    constexpr int PLANE = (AROWS + 1) * R64_ROWB;                  // + zero row
    constexpr int XPLANE = SC ? (XROWS + 1) * R64_ROWB : 0;
    constexpr int SLAB_ROWS = POOL ? 8 : 32, SLAB_RS = 32 * 2 + 16;   // one wave's slab: rows x (32 bf16 + pad)
    constexpr int SLAB = SLAB_ROWS * SLAB_RS;
    constexpr int X_BASE = 3 * PLANE, SLAB_BASE = X_BASE + XPLANE, SHIFT_BASE = SLAB_BASE + R64_NW * SLAB;
    static_assert(AROWS % 16 == 0 && (!SC || XROWS % 16 == 0), "DMA pieces are 16 rows");
    static_assert(PLANE < 65536 && XPLANE < 65536, "packed A addresses are 16-bit");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wh = wid & 1;                          // row quarter, column half of the 256 x 64 tile
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    // ---- tile schedule: XCD x (= blockIdx % 8: the workgroups that share an L2) owns one contiguous range of tiles, its
    //      workgroups walk it interleaved, so the halo rows of neighbouring tiles are fetched into ONE L2
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (nwg + 7 - xcd) >> 3;
    const int per = (a.ntiles + 7) >> 3;
    const int t_begin = xcd * per + slot, t_end = min((xcd + 1) * per, a.ntiles);

    // ---- resident weights: B fragments of this wave's 32 output columns, all taps / chunks / k-steps (144 VGPRs)
    uint4 bw[2][9][2];
    {
        const char* wl = a.w + (size_t)(32 * wh + lr) * R64_ROWB + lh * 16;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    bw[c][t][s] = *reinterpret_cast<const uint4*>(wl + (size_t)((t * 2 + c) * 64) * R64_ROWB + s * 32);
    }
    uint4 bw2[2];
    if (SC) {
        const char* wl = a.w2 + (size_t)(32 * wh + lr) * R64_ROWB + lh * 16;
        bw2[0] = *reinterpret_cast<const uint4*>(wl);
        bw2[1] = *reinterpret_cast<const uint4*>(wl + 32);
    }
    // zero rows (row AROWS of every plane, row XROWS of the shortcut plane) and this wave's shift values
    if (tid < 16) {
        const int pl = tid >> 2, q = tid & 3;
        if (pl < 3) *reinterpret_cast<uint4*>(smem + pl * PLANE + AROWS * R64_ROWB + q * 16) = make_uint4(0, 0, 0, 0);
        else if (SC) *reinterpret_cast<uint4*>(smem + X_BASE + XROWS * R64_ROWB + q * 16) = make_uint4(0, 0, 0, 0);
    }
    float* const s_shift = reinterpret_cast<float*>(smem + SHIFT_BASE);
    if (tid < 64) s_shift[tid] = a.shift[tid];

    const int prl = lane >> 2, psl = lane & 3;                      // row within a DMA piece, physical 16-byte slot
    // stage channel chunk `c` of the patch rows [lo, lo + rows) into plane `pl` (pieces dealt round-robin over the waves)
    auto stage_plane = [&](int pl, int c, int lo, int rows) {
        const char* base = a.x + (size_t)lo * 128 + c * 64;         // wave-uniform
        const int pieces = (rows + 15) >> 4;
        for (int q = wid; q < pieces; q += R64_NW) {
            const int row = q * 16 + prl;
            const int srow = row < rows ? row : rows - 1;
            dma16(base, (unsigned)srow * 128u + ((psl ^ swz<4>(row)) << 4), lds_base + pl * PLANE + q * 1024);
        }
    };
    auto stage_x2 = [&](int cf, int rows) {
        const char* base = a.x2 + (size_t)cf * 64;
        const int pieces = (rows + 15) >> 4;
        for (int q = wid; q < pieces; q += R64_NW) {
            const int row = q * 16 + prl;
            const int srow = row < rows ? row : rows - 1;
            dma16(base, (unsigned)srow * 64u + ((psl ^ swz<4>(row)) << 4), lds_base + X_BASE + q * 1024);
        }
    };

    int t = t_begin;
    if (t >= t_end) return;
    int lo, hi, cf, cl;
    r64_range<POOL>(a, t * R64_TM, lo, hi, cf, cl);
    stage_plane(0, 0, lo, hi - lo);
    stage_plane(1, 1, lo, hi - lo);
    if (SC) stage_x2(cf, cl - cf + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int m0 = t * R64_TM;
        const int p0 = (2 * it) % 3, p1 = (2 * it + 1) % 3, pf = (2 * it + 2) % 3;   // planes: chunk 0, chunk 1, free
        const int tn = t + nslot;
        const bool more = tn < t_end;
        int nlo = 0, nhi = 0, ncf = 0, ncl = 0;
        if (more) {
            r64_range<POOL>(a, tn * R64_TM, nlo, nhi, ncf, ncl);
            stage_plane(pf, 0, nlo, nhi - nlo);                    // next tile's chunk 0 -> the free plane
        }
        // per-lane LDS addresses (relative to a plane) of this lane's two A rows for the nine taps; k-step s is addr ^ 32 s
        unsigned apk[9], axs = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) apk[k] = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wm * 64 + i * 32 + lr;
            const bool mv = m < a.M;
            int p, h, w;
            r64_pixel<POOL>(a, mv ? m : 0, p, h, w);
            const bool up = h > 0, dn = h < a.H - 1, lf = w > 0, rt = w < a.W - 1;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt) {
                const int dy = tt / 3 - 1, dx = tt % 3 - 1;
                const bool ok = mv && (dy < 0 ? up : dy > 0 ? dn : true) && (dx < 0 ? lf : dx > 0 ? rt : true);
                const int row = p + dy * a.W + dx - lo;
                const unsigned ad = ok ? (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row)) : (unsigned)AROWS * R64_ROWB + 16u * lh;
                apk[(i * 9 + tt) >> 1] |= ad << (16 * ((i * 9 + tt) & 1));
            }
            if (SC) {
                const int row = p - cf;
                const unsigned ad = mv ? (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row)) : (unsigned)XROWS * R64_ROWB + 16u * lh;
                axs |= ad << (16 * i);
            }
        }
        auto aaddr = [&](int i, int tt) -> unsigned {
            const int idx = i * 9 + tt;
            return (idx & 1) ? (apk[idx >> 1] >> 16) : (apk[idx >> 1] & 0xffffu);
        };

        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

        auto mma = [&](const uint4& av, const uint4& bv, f32x16& c) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
        };
        auto chunk = [&](int pl, int c) {
            const char* pb = smem + pl * PLANE;
#pragma unroll
            for (int tt = 0; tt < 9; ++tt)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const uint4 a0 = *reinterpret_cast<const uint4*>(pb + (aaddr(0, tt) ^ (32u * s)));
                    const uint4 a1 = *reinterpret_cast<const uint4*>(pb + (aaddr(1, tt) ^ (32u * s)));
                    mma(a0, bw[c][tt][s], acc[0]);
                    mma(a1, bw[c][tt][s], acc[1]);
                }
        };
        if (SC) {                                                   // shortcut GEMM first: its plane is re-staged at the mid barrier
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const uint4 a0 = *reinterpret_cast<const uint4*>(smem + X_BASE + ((axs & 0xffffu) ^ (32u * s)));
                const uint4 a1 = *reinterpret_cast<const uint4*>(smem + X_BASE + ((axs >> 16) ^ (32u * s)));
                mma(a0, bw2[s], acc[0]);
                mma(a1, bw2[s], acc[1]);
            }
        }
        chunk(p0, 0);
        // every wave has finished reading plane p0 (and the shortcut plane): their data fed MFMAs already issued
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            stage_plane(p0, 1, nlo, nhi - nlo);                    // next tile's chunk 1 -> the plane chunk 0 just left
            if (SC) stage_x2(ncf, ncl - ncf + 1);
        }
        chunk(p1, 1);

        // ---- epilogue: + shift, LeakyReLU, (2x2 max), bf16, through this wave's LDS slab, 16-byte stores
        // C layout of a 32x32 tile: column = lane % 32; register r holds row (r & 3) + 8 (r >> 2) + 4 (lane / 32)
        char* const slab = smem + SLAB_BASE + wid * SLAB;
        const float sh = s_shift[32 * wh + lr];
        constexpr int NST = POOL ? 1 : 2;                           // global-store instructions per 32-row MFMA tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mrow0 = m0 + wm * 64 + i * 32;
            if (!POOL) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][r] + sh;
                    if (a.act) v = fmaxf(v, v * 0.1f);
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    *reinterpret_cast<__bf16*>(slab + row * SLAB_RS + lr * 2) = (__bf16)v;
                }
#pragma unroll
                for (int v0 = 0; v0 < 128; v0 += 64) {             // 32 rows x 4 vectors of 16 bytes
                    const int v = v0 + lane, row = v >> 2, c16 = v & 3;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * SLAB_RS + c16 * 16);
                    if (mrow0 + row < a.M)
                        *reinterpret_cast<uint4*>(a.y + (size_t)(mrow0 + row) * 128 + wh * 64 + c16 * 16) = val;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // registers 4q .. 4q+3 = rows 8q + 4 lh + {0..3} = one window
                    float best = fmaxf(fmaxf(acc[i][4 * q], acc[i][4 * q + 1]), fmaxf(acc[i][4 * q + 2], acc[i][4 * q + 3])) + sh;
                    if (a.act) best = fmaxf(best, best * 0.1f);     // monotone => lrelu(max) == max(lrelu)
                    *reinterpret_cast<__bf16*>(slab + (2 * q + lh) * SLAB_RS + lr * 2) = (__bf16)best;
                }
                const int row = lane >> 2, c16 = lane & 3;          // 8 pooled rows x 4 vectors: lanes 0..31
                if (lane < 32) {
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * SLAB_RS + c16 * 16);
                    if (mrow0 + 4 * row < a.M)
                        *reinterpret_cast<uint4*>(a.y + (size_t)((mrow0 >> 2) + row) * 128 + wh * 64 + c16 * 16) = val;
                }
            }
        }
        // the next tile's DMAs are older than this epilogue's stores: wait for all but the 2 NST youngest operations
        // (a ragged last tile may skip store instructions: wait for everything there)
        if (m0 + R64_TM > a.M) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        lo = nlo; cf = ncf;
    }
}

template <bool POOL, bool SC, int AROWS, int XROWS>
static int launch_r64(const Conv64Args& a, hipStream_t stream) {
    constexpr int PLANE = (AROWS + 1) * R64_ROWB, XPLANE = SC ? (XROWS + 1) * R64_ROWB : 0;
    constexpr int SLAB = (POOL ? 8 : 32) * (32 * 2 + 16);
    const size_t lds = 3 * (size_t)PLANE + XPLANE + R64_NW * SLAB + 64 * sizeof(float);
    static_assert(3 * PLANE + XPLANE + R64_NW * SLAB + 256 <= 160 * 1024, "LDS budget");
    auto kern = conv64_resident_kernel<POOL, SC, AROWS, XROWS>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int grid = cus < a.ntiles ? cus : a.ntiles;                    // one persistent workgroup per CU
    grid = (grid + 7) / 8 * 8;                                     // whole XCD groups (surplus workgroups exit at once)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(R64_NW * 64), lds, stream, a);
    return launch_status();
}

// Returns SUBREG_EUNSUPPORTED when the shape is not this kernel's (the caller then uses the general kernel).
int conv64_resident(const void* x, const void* w, void* y, const float* shift, const void* x2, const void* w2, int Cin2, int B,
                    int H, int W, bool pool, int act, hipStream_t stream) {
    if ((x2 != nullptr) != (w2 != nullptr) || (x2 && Cin2 != 32)) return SUBREG_EUNSUPPORTED;
    const ConvGeom g = make_geom(B, H, W, 9, pool);
    if ((long long)g.npix * (H * W) >= (1LL << 40) || g.npix >= (1 << 27)) return SUBREG_EUNSUPPORTED;   // FastDiv range, 32-bit offsets
    // rows a patch plane (with halo) and the shortcut plane (without) must hold: scan one period of tile starts (tile t starts
    // at window 64 t; the pattern repeats after Hp*Wp tiles) plus the first and last tile
    int worst = 1, xworst = 1;
    {
        const int ntiles = (g.M + R64_TM - 1) / R64_TM;
        const int period = pool ? g.Hp * g.Wp + 1 : 3;
        for (int k = 0; k <= period; ++k) {
            const int t = k < period ? k : ntiles - 1;
            if (t >= ntiles) continue;
            int lo, hi;
            int m1 = t * R64_TM + R64_TM;
            if (m1 > g.M) m1 = g.M;
            int core;
            if (pool) {
                patch_range<true>(g, t * R64_TM, R64_TM, &lo, &hi);
                core = row_to_pixel<true>(g, (m1 - 1) | 3).p - row_to_pixel<true>(g, t * R64_TM).p + 1;
            } else {
                patch_range<false>(g, t * R64_TM, R64_TM, &lo, &hi);
                core = m1 - t * R64_TM;
            }
            if (hi - lo > worst) worst = hi - lo;
            if (core > xworst) xworst = core;
        }
    }
    Conv64Args a;
    a.x = (const char*)x; a.w = (const char*)w; a.x2 = (const char*)x2; a.w2 = (const char*)w2; a.y = (char*)y; a.shift = shift;
    a.H = H; a.W = W; a.Hp = g.Hp; a.Wp = g.Wp; a.npix = g.npix; a.M = g.M; a.act = act;
    a.ntiles = (g.M + R64_TM - 1) / R64_TM;
    a.d_hw = make_fastdiv(H * W); a.d_w = make_fastdiv(W);
    a.d_pp = make_fastdiv(g.Hp * g.Wp > 0 ? g.Hp * g.Wp : 1); a.d_wp = make_fastdiv(g.Wp > 0 ? g.Wp : 1);
    if (!pool) {
        if (worst > 432 || xworst > 256) return SUBREG_EUNSUPPORTED;
        return x2 ? launch_r64<false, true, 432, 256>(a, stream) : launch_r64<false, false, 432, 256>(a, stream);
    }
    if (worst > 560 || xworst > 384) return SUBREG_EUNSUPPORTED;
    return x2 ? launch_r64<true, true, 560, 384>(a, stream) : launch_r64<true, false, 560, 384>(a, stream);
}

}  // namespace subreg
