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

constexpr int R64_ROWB = 64;
#ifndef R64_TWO_WG
#define R64_TWO_WG 1    // unpooled conv: 128-row tiles, two 4-wave workgroups per CU (0: one 8-wave workgroup, 256-row tiles)
#endif
constexpr int R64_TM_SMALL = 128;
#ifndef R64_DIAG
#define R64_DIAG 0      // 1: per-wave s_memtime stamps of the tile loop's phases into r64_diag (measurement builds only)
#endif
#if R64_DIAG
__device__ float r64_diag[4096 * 8];   // [workgroup * waves + wave][8]: tiles, addr, chunk0, barrier1, chunk1, epilogue, dma wait, barrier2
#endif
#ifndef R64_DEPTH
#define R64_DEPTH 4     // A-fragment reads in flight ahead of the MFMA that consumes them (+1)
#endif

// LDS fragment read whose completion the compiler must not guess: issued and waited for by hand (see `chunk` below)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 lds_read16(unsigned lds_addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(lds_addr));
    return d;
}
template <int N>
__device__ __forceinline__ u32x4 lds_wait(u32x4 frag) {      // all but the N youngest LDS operations are done; `frag` is one of them
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
    return frag;
}

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
template <bool POOL, int TM>
__device__ __forceinline__ void r64_range(const Conv64Args& a, int m0, int& lo, int& hi, int& cf, int& cl) {
    int m1 = m0 + TM;
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

// AROWS: patch rows an LDS plane holds (+ one zero row); XROWS: rows of the shortcut plane (tile rows without halo);
// WM: waves along M (tile = 64 WM rows, 2 WM waves).  WM = 4: one 8-wave workgroup per CU; WM = 2: two independent 4-wave
// workgroups per CU, whose phases (address arithmetic / MFMA / epilogue) drift apart and overlap - the 8 waves of one
// workgroup run in lock step between the two barriers of a tile and leave the MFMA pipe idle during their common epilogue.
template <bool POOL, bool SC, int AROWS, int XROWS, int WM>
__global__ __launch_bounds__(2 * WM * 64, 2) void conv64_resident_kernel(const Conv64Args a) {
    constexpr int R64_TM = 64 * WM, R64_NW = 2 * WM;
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
    r64_range<POOL, R64_TM>(a, t * R64_TM, lo, hi, cf, cl);
    stage_plane(0, 0, lo, hi - lo);
    stage_plane(1, 1, lo, hi - lo);
    if (SC) stage_x2(cf, cl - cf + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define R64_STAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dt[k] += n_ - dq; dq = n_; } while (0)
#else
#define R64_STAMP(k) do { } while (0)
#endif
    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int m0 = t * R64_TM;
        const int p0 = (2 * it) % 3, p1 = (2 * it + 1) % 3, pf = (2 * it + 2) % 3;   // planes: chunk 0, chunk 1, free
        const int tn = t + nslot;
        const bool more = tn < t_end;
        int nlo = 0, nhi = 0, ncf = 0, ncl = 0;
        if (more) {
            r64_range<POOL, R64_TM>(a, tn * R64_TM, nlo, nhi, ncf, ncl);
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
#if R64_DIAG
        asm volatile("" : "+v"(apk[0]), "+v"(apk[8]));                // the address arithmetic ends here
        dt[0] += 1;
#endif
        R64_STAMP(1);

        auto mma = [&](const uint4& av, const uint4& bv, f32x16& c) {
            c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c, 0, 0, 0);
        };
        // one channel chunk = 36 A-fragment reads (tap-major, then k-step, then row tile), each feeding one MFMA.  An LDS read
        // takes ~100+ cycles and an MFMA 32, so the reads run RD - 1 fragments ahead of their use through a register ring.
        // hipcc serialises read -> wait -> MFMA on one register when left alone (it minimises pressure), so the reads and
        // their counted waits are inline asm: the wait statement names the fragment it completes ("+v"), which orders the
        // MFMA behind it; no other LDS / scalar-memory operation is in flight inside a chunk (drained on entry).
        auto chunk = [&](int pl, int c) {
            const unsigned pb = lds_base + pl * PLANE;
            constexpr int NRD = 36, RD = R64_DEPTH;
            u32x4 ring[RD];
            auto rd = [&](int j) {
                const int tt = j >> 2, s = (j >> 1) & 1, i = j & 1;
                ring[j % RD] = lds_read16(pb + (aaddr(i, tt) ^ (32u * s)));
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int j = 0; j < RD - 1; ++j) rd(j);
#pragma unroll
            for (int j = 0; j < NRD; ++j) {
                if (j + RD - 1 < NRD) rd(j + RD - 1);
                const int left = NRD - 1 - j;                          // reads issued after fragment j
                u32x4 f = ring[j % RD];
                if (left >= RD - 1) f = lds_wait<RD - 1>(f);
                else if (left == 3) f = lds_wait<3>(f);
                else if (left == 2) f = lds_wait<2>(f);
                else if (left == 1) f = lds_wait<1>(f);
                else f = lds_wait<0>(f);
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f), __builtin_bit_cast(bf16x8, bw[c][j >> 2][(j >> 1) & 1]),
                                                                      acc[j & 1], 0, 0, 0);
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
        R64_STAMP(2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(3);
        if (more) {
            stage_plane(p0, 1, nlo, nhi - nlo);                    // next tile's chunk 1 -> the plane chunk 0 just left
            if (SC) stage_x2(ncf, ncl - ncf + 1);
        }
        chunk(p1, 1);
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(4);

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
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(5);
        // the next tile's DMAs are older than this epilogue's stores: wait for all but the 2 NST youngest operations
        // (a ragged last tile may skip store instructions: wait for everything there)
        if (m0 + R64_TM > a.M) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NST == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        R64_STAMP(6);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(7);
        lo = nlo; cf = ncf;
    }
#if R64_DIAG
    if (lane == 0 && blockIdx.x * R64_NW + wid < 4096) {
        float* d = r64_diag + (size_t)(blockIdx.x * R64_NW + wid) * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (float)dt[k];
    }
#endif
}

template <bool POOL, bool SC, int AROWS, int XROWS, int WM>
static int launch_r64(const Conv64Args& a, hipStream_t stream) {
    constexpr int NW = 2 * WM;
    constexpr int PLANE = (AROWS + 1) * R64_ROWB, XPLANE = SC ? (XROWS + 1) * R64_ROWB : 0;
    constexpr int SLAB = (POOL ? 8 : 32) * (32 * 2 + 16);
    const size_t lds = 3 * (size_t)PLANE + XPLANE + NW * SLAB + 64 * sizeof(float);
    static_assert((3 * PLANE + XPLANE + NW * SLAB + 256) * (WM == 2 ? 2 : 1) <= 160 * 1024, "LDS budget (two workgroups per CU when WM = 2)");
    auto kern = conv64_resident_kernel<POOL, SC, AROWS, XROWS, WM>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int slots = cus * (WM == 2 ? 2 : 1);                     // persistent workgroups the chip holds at once
    int grid = slots < a.ntiles ? slots : a.ntiles;
    grid = (grid + 7) / 8 * 8;                                     // whole XCD groups (surplus workgroups exit at once)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(NW * 64), lds, stream, a);
    return launch_status();
}

// rows a patch plane (with halo) and the shortcut plane (without) must hold for TM-row tiles: scan one period of tile starts
// (tile t starts at window TM/4 * t; the pattern repeats after Hp*Wp tiles) plus the last tile
static void r64_worst_rows(const ConvGeom& g, bool pool, int TM, int* worst_out, int* xworst_out) {
    int worst = 1, xworst = 1;
    const int ntiles = (g.M + TM - 1) / TM;
    const int period = pool ? g.Hp * g.Wp + 1 : 3;
    for (int k = 0; k <= period; ++k) {
        const int t = k < period ? k : ntiles - 1;
        if (t >= ntiles) continue;
        int lo, hi;
        int m1 = t * TM + TM;
        if (m1 > g.M) m1 = g.M;
        int core;
        if (pool) {
            patch_range<true>(g, t * TM, TM, &lo, &hi);
            core = row_to_pixel<true>(g, (m1 - 1) | 3).p - row_to_pixel<true>(g, t * TM).p + 1;
        } else {
            patch_range<false>(g, t * TM, TM, &lo, &hi);
            core = m1 - t * TM;
        }
        if (hi - lo > worst) worst = hi - lo;
        if (core > xworst) xworst = core;
    }
    *worst_out = worst; *xworst_out = xworst;
}

// Returns SUBREG_EUNSUPPORTED when the shape is not this kernel's (the caller then uses the general kernel).
int conv64_resident(const void* x, const void* w, void* y, const float* shift, const void* x2, const void* w2, int Cin2, int B,
                    int H, int W, bool pool, int act, hipStream_t stream) {
    if ((x2 != nullptr) != (w2 != nullptr) || (x2 && Cin2 != 32)) return SUBREG_EUNSUPPORTED;
    const ConvGeom g = make_geom(B, H, W, 9, pool);
    if ((long long)g.npix * (H * W) >= (1LL << 40) || g.npix >= (1 << 27)) return SUBREG_EUNSUPPORTED;   // FastDiv range, 32-bit offsets
    Conv64Args a;
    a.x = (const char*)x; a.w = (const char*)w; a.x2 = (const char*)x2; a.w2 = (const char*)w2; a.y = (char*)y; a.shift = shift;
    a.H = H; a.W = W; a.Hp = g.Hp; a.Wp = g.Wp; a.npix = g.npix; a.M = g.M; a.act = act;
    a.d_hw = make_fastdiv(H * W); a.d_w = make_fastdiv(W);
    a.d_pp = make_fastdiv(g.Hp * g.Wp > 0 ? g.Hp * g.Wp : 1); a.d_wp = make_fastdiv(g.Wp > 0 ? g.Wp : 1);
    int worst, xworst;
    if (!pool) {
        r64_worst_rows(g, false, R64_TM_SMALL, &worst, &xworst);
        if (R64_TWO_WG && worst <= 304) {                              // 128-row tiles, two workgroups per CU
            a.ntiles = (g.M + R64_TM_SMALL - 1) / R64_TM_SMALL;
            return x2 ? launch_r64<false, true, 304, 128, 2>(a, stream) : launch_r64<false, false, 304, 128, 2>(a, stream);
        }
        r64_worst_rows(g, false, 256, &worst, &xworst);
        if (worst > 432) return SUBREG_EUNSUPPORTED;
        a.ntiles = (g.M + 255) / 256;
        return x2 ? launch_r64<false, true, 432, 256, 4>(a, stream) : launch_r64<false, false, 432, 256, 4>(a, stream);
    }
    r64_worst_rows(g, true, 256, &worst, &xworst);
    if (worst > 560 || xworst > 384) return SUBREG_EUNSUPPORTED;
    a.ntiles = (g.M + 255) / 256;
    return x2 ? launch_r64<true, true, 560, 384, 4>(a, stream) : launch_r64<true, false, 560, 384, 4>(a, stream);
}

#if R64_DIAG
extern "C" int subreg_r64_diag_read(float* host_out, int n_floats) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(r64_diag), sizeof(float) * n_floats) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace subreg
