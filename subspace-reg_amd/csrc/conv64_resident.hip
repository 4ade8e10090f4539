// Layer-1 3x3 convolutions (Cin = Cout = 64 on 84x84 maps: models/resnet_language.py:249-256, conv2 / conv3 of layer1.0) as a
// PERSISTENT implicit GEMM with REGISTER-RESIDENT weights, bf16, eval mode (BN scale folded into the weights; shift +
// LeakyReLU(0.1) + optional MaxPool2d(2) + the fused K=32 shortcut GEMM of conv3 in the epilogue, :268-301).
//
// Why a second kernel for this shape: with K = 576 and N = 64 a 256-row tile is only 4.6 k MFMA cycles per wave; the general
// kernel (conv_fwd.hip) stages 74 KB of weights + 55 KB of patch per tile and pays a 5.5 k-cycle prologue and a 4.3 k-cycle
// epilogue around them - 23-25 % of the MFMA peak on 20 % of the backbone's time (profiles/r01_conv_stamps.txt).  A first
// persistent version with per-tile address arithmetic turned out VALU-ISSUE bound (in-kernel stamps: ~900 instructions per
// 72 MFMAs per wave and tile, a quarter of them address arithmetic).  This version removes the per-tile arithmetic:
//   * one workgroup of 8 waves per CU loops over tiles (persistent); the next tile's patch is prefetched by LDS-DMA;
//   * the WHOLE weight matrix lives in registers: wave (wm, wh) owns rows [64 wm, +64) x columns [32 wh, +32) of the
//     256 x 64 tile and keeps the 36 B fragments (9 taps x 2 channel chunks x 2 k-steps) of its columns in 144 VGPRs;
//   * LDS holds activation planes ([row][32 channels], 64-byte rows, XOR-swizzled like conv_index.h::swz) in a PADDED image
//     layout: every image row of the patch is a block of P = 96 LDS rows (one zero row, the W pixels, zero rows), rows
//     outside the image are blocks of zeros.  A tap is then a CONSTANT offset: dy = +-1 is +-P rows (an immediate of the
//     ds_read), dx = +-1 a second / third base register; no validity selects (the zero padding does it), and because a
//     LINEAR tile is a whole number of image rows (3 rows = 252 of 256 GEMM rows) its per-lane addresses are the same
//     for every tile - computed once per kernel.  POOL tiles are 63 windows (1.5 row pairs): two alternating geometries.
// Same data layouts and row orders (LINEAR / POOL window-major) as conv_fwd.hip; fp32 accumulation.
#include <type_traits>

#include "conv_index.h"
#include "subreg_common.h"

namespace subreg {

#ifndef R64_DEPTH
#define R64_DEPTH 4     // A-fragment reads in flight ahead of the MFMA that consumes them (+1)
#endif
#ifndef R64_FUSED_STAGGER
#define R64_FUSED_STAGGER 1     // conv64_fused_first_kernel: waves 4-7 run conv1 of the next tile AFTER their conv2 chunks (0: all waves first)
#endif
#ifndef R64_PI_STAGGER
#define R64_PI_STAGGER 1        // conv64_pool_img_kernel: waves 4-7 stage the next tile between their chunks and their epilogue (0: all waves first)
#endif
#ifndef R64_PI_PRIO
#define R64_PI_PRIO 1           // conv64_pool_img_kernel: s_setprio around the chunks
#endif
#ifndef R64_SPLIT_STAGE
#define R64_SPLIT_STAGE 0       // conv64_resident_kernel: 1 = waves 4-7 stage their chunk-0 pieces of the next tile behind the mid barrier (measured: no effect, profiles/r04_ab_l1_young_prio.txt)
#endif
#ifndef R64_YOUNG_PRIO
#define R64_YOUNG_PRIO 0        // conv64_resident_kernel: s_setprio of waves 4-7 for the whole kernel (measured: see profiles/r04_ab_l1_young_prio.txt)
#endif
#ifndef R64_FUSED_CONV1_HALF
#define R64_FUSED_CONV1_HALF 0  // conv64_fused_first_kernel, experiment: 1 = waves 0-3 compute ALL of conv1 (before their chunks), waves 4-7 none
#endif
#ifndef R64_FUSED_ROLL
#define R64_FUSED_ROLL 1        // conv64_fused_first_kernel: conv1 computes 3 new rows per tile and copies the 2 it shares with the tile above
#endif
#ifndef R64_FUSED_PRIO
#define R64_FUSED_PRIO 1        // conv64_fused_first_kernel: s_setprio around the conv2 chunks (see there)
#endif
#ifndef R64_FAST_STAGE
#define R64_FAST_STAGE 1        // conv64_resident_kernel: branch-free patch staging for tiles whose blocks all lie inside the image
#endif
#ifndef R64_CUT
#define R64_CUT 0       // conv64_fused_first_kernel, timing experiments only (WRONG results): leave one component out -
                        // 1 conv2's MFMAs, 2 conv2's A-fragment reads, 3 conv1, 4 the epilogue, 5 image DMA + patch conversion
#endif
#ifndef R64WF_CUT
#define R64WF_CUT 0     // conv64_wide_fused_kernel, timing experiments only (WRONG results): 1 no conv1 fillers, 2 no row-tile epilogue fillers, 3 no roll copy,
                        // 4 no conv1 MFMAs (reads, LeakyReLU and stores stay), 5 no conv1 LeakyReLU / stores (reads and MFMAs stay)
#endif
#ifndef R64_DIAG
#define R64_DIAG 0      // 1: per-wave s_memtime stamps of the tile loop's phases into r64_diag (measurement builds only)
#endif
#if R64_DIAG
__device__ float r64_diag[4096 * 12];   // [workgroup * 8 + wave][8]: tiles, setup, chunk0, barrier1, chunk1, epilogue, dma wait, barrier2
#endif

// 128 zero bytes: the DMA source of patch blocks that lie outside the image (rows above the top / below the bottom)
__device__ __attribute__((aligned(128))) unsigned r64_zero_line[32];

constexpr int R64_P = 96;            // LDS rows per image row of the patch (W + 1 <= P, multiple of 16)
constexpr int R64_PF = R64_P;        // ... in conv64_fused_first_kernel.  (Its planes are written by conv1, not by 16-row DMA pieces, but P must
                                     // still be a multiple of 16: a tap's dy is an IMMEDIATE offset of dy * P rows on an address whose swizzle
                                     // was computed for the dy = 0 row, so swz(row + P) has to equal swz(row).  P = 100 would keep a tile that
                                     // crosses an image-row boundary conflict-free - and read the wrong slots; no 4-valued row swizzle with a
                                     // period dividing P serves both, tests/test_host_cpu.py::test_layer1_plane_swizzle_emulation.)
constexpr int R64_ROWB = 64;         // bytes per LDS row (32 bf16 channels)
constexpr int R64_NW = 8;            // waves per workgroup

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
    const char* x2;      // fused shortcut GEMM: [npix][32] bf16 (im2col rows, k = 3 tap + c) or null
    const float* img;    // ... or the fp32 NCHW image itself [B][3][H][W] (IMG kernels: the 1x1 shortcut reads its 3 channels)
    const char* w2;      // [64][32] bf16
    char* y;             // LINEAR [npix][64] ; POOL [B*Hp*Wp][64]
    const float* shift;  // [64]
    int H, W, act, ntiles, tpi;      // tpi: tiles per image
    int R;                           // LINEAR: image rows per tile
    int Hp, Wp, WT, nwin;            // POOL: windows per tile, windows per image
    FastDiv d_w, d_wp, d_tpi;        // W, Wp, tiles per image
};

template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// LDS fragment read whose completion the compiler must not guess: issued and waited for by hand (see `chunk` below)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ u32x4 lds_read16(unsigned lds_addr) {
    u32x4 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "n"(OFF));
    return d;
}
template <int N>
__device__ __forceinline__ u32x4 lds_wait(u32x4 frag) {      // all but the N youngest LDS operations are done; `frag` is one of them
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(frag) : "n"(N));
    return frag;
}

typedef float r64_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 r64_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned r64_pack(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(r64_f32x2{lo, hi}, r64_bf16x2));
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

// one LDS-DMA piece under a lane mask held in an SGPR pair (exec is put back inside the statement: the compiler never sees it move; no
// s_and_saveexec / s_cbranch_execz pair around the piece)
__device__ __forceinline__ void dma16_masked(const char* base, unsigned voff, unsigned lds_addr, unsigned long long mask) {
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    unsigned long long keep;
    asm volatile(
        "s_mov_b64 %0, exec\n\t"
        "s_mov_b64 exec, %4\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b64 exec, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_u), "s"(mask)
        : "memory", "m0");
}

// BLOCKS: image-row blocks per plane (LINEAR R + 2 = 5, POOL 2 row pairs + halo = 6)
// IMG (with SC): the shortcut's input is the fp32 NCHW image (layer1.0's downsample conv, models/resnet_language.py:146-147,286):
// each tile's own pixels are loaded from the three channel planes, converted to bf16 and written into logical slot 0 of
// their LDS rows ([c0 c1 c2 0 0 0 0 0]; the other slots stay zero), and the shortcut GEMM is ONE k-step - no im2col buffer.
template <bool POOL, bool SC, int BLOCKS, bool IMG = false>
__global__ __launch_bounds__(R64_NW * 64, 2) void conv64_resident_kernel(const Conv64Args a) {
    static_assert(!IMG || (SC && BLOCKS == 6), "the image-fed shortcut exists for the pooled conv3 of layer 1");
    constexpr int P = R64_P, PROWS = BLOCKS * P, PLANE = PROWS * R64_ROWB, PIECES = PROWS / 16, PPB = P / 16;
    constexpr int NPK = (PIECES + R64_NW - 1) / R64_NW;              // DMA pieces per wave and plane
    constexpr int SLAB_ROWS = POOL ? 8 : 32, SLAB_RS = 32 * 2 + 16;   // one wave's slab: rows x (32 bf16 + pad)
    constexpr int SLAB = SLAB_ROWS * SLAB_RS;
    constexpr int X_BASE = 3 * PLANE, SLAB_BASE = X_BASE + (SC ? PLANE : 0), SHIFT_BASE = SLAB_BASE + R64_NW * SLAB;
    static_assert(2 * P * R64_ROWB < 65536, "tap offsets are 16-bit immediates");
    extern __shared__ __attribute__((aligned(16))) char smem[];

#if R64_DIAG
    const unsigned long long d_entry = __builtin_amdgcn_s_memtime(), d_entry_r = __builtin_amdgcn_s_memrealtime();
#endif
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
    if (t_begin >= t_end) return;

#if R64_YOUNG_PRIO
    if (wid >= 4) __builtin_amdgcn_s_setprio(R64_YOUNG_PRIO);       // static priority for the later-dispatched half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#endif
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
    if (SC && !IMG) {
        const char* wl = a.w2 + (size_t)(32 * wh + lr) * R64_ROWB + lh * 16;
        bw2[0] = *reinterpret_cast<const uint4*>(wl);
        bw2[1] = *reinterpret_cast<const uint4*>(wl + 32);
    }
    if (IMG) {
        // packed 1x1 weights of the first layer sit at k = 3 * (centre tap 4) + c = 12..14 of their row; the LDS rows carry
        // the channels at k = 0..2, so the B fragment of the one k-step is [w12 w13 w14 0 ...] in the lanes holding k 0..7
        const unsigned short* wr = reinterpret_cast<const unsigned short*>(a.w2 + (size_t)(32 * wh + lr) * R64_ROWB);
        const unsigned w12 = wr[12], w13 = wr[13], w14 = wr[14];
        bw2[0] = lh == 0 ? make_uint4(w12 | (w13 << 16), w14, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
        bw2[1] = make_uint4(0u, 0u, 0u, 0u);
    }
    // ---- LDS starts as zeros: the pad rows of every block are never written again (the DMAs mask those lanes off)
    for (int o = tid * 16; o < SLAB_BASE; o += R64_NW * 64 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    float* const s_shift = reinterpret_cast<float*>(smem + SHIFT_BASE);
    if (tid < 64) s_shift[tid] = a.shift[tid];
    __syncthreads();                                                // zeros are in LDS before any DMA may land on them

    // ---- DMA geometry, the same for every tile: piece q = wid + 8 k covers LDS rows 16 q .. 16 q + 15 of a plane = block
    //      q / 6, columns 16 (q % 6) .. + 15 of the padded image row (column c holds pixel c - 1; 0 and W+1.. are pads)
    const int prl = lane >> 2, psl = lane & 3;                      // row within a piece, physical 16-byte slot
    unsigned goff[NPK], goff2[NPK];                                 // this lane's byte offset from the patch origin in x / in x2
    unsigned long long dmask[NPK];                                  // the piece's data lanes (pads are never written) as an exec mask
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
        const int q = wid + R64_NW * k, idx = q * 16 + prl, rb = q / PPB, c = idx - rb * P;
        const unsigned slot16 = (unsigned)(psl ^ swz<4>(idx)) << 4;
        dmask[k] = __ballot(q < PIECES && c >= 1 && c <= a.W);
        goff[k] = (unsigned)(rb * a.W + c - 1) * 128u + slot16;
        goff2[k] = (unsigned)(rb * a.W + c - 1) * 64u + slot16;
    }
    // stage chunk `c` (c = 2: the shortcut rows of x2) of the patch whose block 0 is image row `h_first` of image `b`
    auto stage = [&](unsigned dst, int c, int b, int h_first) {
        const long long origin = ((long long)b * a.H + h_first) * a.W;          // pixel index of (block 0, column 1); may lie before x
        const char* const zero = reinterpret_cast<const char*>(r64_zero_line);
        if (R64_FAST_STAGE && h_first >= 0 && h_first + BLOCKS <= a.H) {
            // every block of the patch lies inside the image (all tiles but the first and the last of an image): no per-piece
            // decisions, one masked DMA statement per piece (measured: profiles/r04_ab_l1_conv3_stage.txt)
            const char* const src = c == 2 ? a.x2 + origin * 64 : a.x + origin * 128 + c * 64;
#pragma unroll
            for (int k = 0; k < NPK; ++k) {
                const int q = wid + R64_NW * k;
                if (q < PIECES) {                                               // wave-uniform (compile-time true but for the last k)
                    const int rb = q / PPB;
                    if (!(c == 2 && (rb == 0 || rb == BLOCKS - 1)) && dmask[k])
                        dma16_masked(src, c == 2 ? goff2[k] : goff[k], dst + q * 1024, dmask[k]);
                }
            }
            return;
        }
#pragma unroll
        for (int k = 0; k < NPK; ++k) {
            const int q = wid + R64_NW * k;
            if (q < PIECES) {                                                   // wave-uniform
                const int rb = q / PPB, h = h_first + rb;
                const bool halo = rb == 0 || rb == BLOCKS - 1;                  // the shortcut GEMM reads the centre tap only
                const bool inside = h >= 0 && h < a.H;                          // wave-uniform: a block is in or out of the image
                if (!(c == 2 && halo) && ((dmask[k] >> lane) & 1ull)) {
                    if (!inside) dma16(zero, (unsigned)psl << 4, dst + q * 1024);
                    else if (c == 2) dma16(a.x2 + origin * 64, goff2[k], dst + q * 1024);
                    else dma16(a.x + origin * 128 + c * 64, goff[k], dst + q * 1024);
                }
            }
        }
    };
    // IMG: thread (block ib = tid / 128 of the 4 non-halo blocks, column x = tid % 128) loads its pixel's three channels ...
    const int ib = tid >> 7, ix = tid & 127;
    float iv[3] = {0.f, 0.f, 0.f};
    auto img_load = [&](int b, int h_first) {
        const int h = h_first + 1 + ib;
        const bool ok = ix < a.W && h >= 0 && h < a.H;
        const size_t plane = (size_t)a.H * a.W;
        const float* p = a.img + (size_t)b * 3 * plane + (size_t)(ok ? h : 0) * a.W + (ok ? ix : 0);
        const float v0 = p[0], v1 = p[plane], v2 = p[2 * plane];
        iv[0] = ok ? v0 : 0.f; iv[1] = ok ? v1 : 0.f; iv[2] = ok ? v2 : 0.f;
    };
    // ... and writes them (bf16, logical slot 0 of LDS row (1 + ib) P + 1 + x) once every wave is done with the shortcut plane
    auto img_store = [&]() {
        if (ix < a.W) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
            const int row = (1 + ib) * P + 1 + ix;
            const uint4 v = make_uint4(__builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{iv[0], iv[1]}, bf16x2)),
                                       __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{iv[2], 0.f}, bf16x2)), 0u, 0u);
            *reinterpret_cast<uint4*>(smem + X_BASE + row * R64_ROWB + 16 * (0 ^ swz<4>(row))) = v;
        }
    };

    // ---- per-lane A addresses, plane-relative, of the dy = -1 row: [row tile i][dx + 1][k-step]; tap (dy, dx) of a chunk =
    //      areg[i][dx+1][s] + plane base + immediate (dy + 1) * P * 64
    unsigned areg[2][3][2];
    auto set_addresses = [&](int i, int hrel, int w) {              // hrel: block of the pixel's row, minus one
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = hrel * P + w + dx;                      // column (w + 1) + (dx - 1)
            const unsigned ad = (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row));
            areg[i][dx][0] = ad;
            areg[i][dx][1] = ad ^ 32u;
        }
    };
    if (!POOL) {
        // LINEAR: tile = R whole image rows; GEMM row j = (image row j / W, column j % W); rows >= R W are padding
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wm * 64 + i * 32 + lr, jv = j < a.R * a.W ? j : 0;
            const int ir = (int)fdiv((unsigned)jv, a.d_w), w = jv - ir * a.W;
            set_addresses(i, ir, w);
        }
    }

    // ---- tile -> (image, image row of block 0, first window within its row pair).  LINEAR: tile k of an image starts at
    //      image row R k.  POOL: tile k covers windows [WT k, WT k + WT) of the image (row pairs 2 rp, 2 rp + 1)
    auto tile_geom = [&](int t, int& b, int& k_img, int& h_first, int& s0) {
        const int bb = (int)fdiv((unsigned)t, a.d_tpi), k = t - bb * a.tpi;
        b = __builtin_amdgcn_readfirstlane(bb);
        k_img = __builtin_amdgcn_readfirstlane(k);
        if (!POOL) {
            h_first = k_img * a.R - 1;
            s0 = 0;
        } else {
            const int win0 = k_img * a.WT, rp0 = (int)fdiv((unsigned)win0, a.d_wp);
            h_first = __builtin_amdgcn_readfirstlane(2 * rp0 - 1);
            s0 = __builtin_amdgcn_readfirstlane(win0 - rp0 * a.Wp);
        }
    };

    int t = t_begin, b, k_img, h_first, s0;
    tile_geom(t, b, k_img, h_first, s0);
    stage(lds_base + 0 * PLANE, 0, b, h_first);
    stage(lds_base + 1 * PLANE, 1, b, h_first);
    if (SC && !IMG) stage(lds_base + X_BASE, 2, b, h_first);
    if (IMG) { img_load(b, h_first); img_store(); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long d_c0 = dq, d_r0 = __builtin_amdgcn_s_memrealtime();
#define R64_STAMP(k) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dt[k] += n_ - dq; dq = n_; } while (0)
#else
#define R64_STAMP(k) do { } while (0)
#endif
    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int p0 = (2 * it) % 3, p1 = (2 * it + 1) % 3, pf = (2 * it + 2) % 3;   // planes: chunk 0, chunk 1, free
        const int tn = t + nslot;
        const bool more = tn < t_end;
        int nb = 0, nk = 0, nh = 0, ns0 = 0;
        if (more) {
            tile_geom(tn, nb, nk, nh, ns0);
            // next tile's chunk 0 -> the free plane: waves 0-3 stage their pieces here, waves 4-7 theirs behind the mid barrier
            // (R64_SPLIT_STAGE): a wave stalls ~100 cycles per piece while eight waves stage at once, and with both waves of a SIMD
            // in the same phase nothing runs under that; split, one half's MFMAs run under the other half's DMA issue
            if (!R64_SPLIT_STAGE || wid < 4) stage(lds_base + pf * PLANE, 0, nb, nh);
        }
        if (POOL) {
            // GEMM row j = window s0 + j / 4 counted from the start of the tile's first row pair, pixel j % 4 of that window
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int j = wm * 64 + i * 32 + lr, wq = s0 + (j >> 2), sub = j & 3;
                // (a window beyond the tile's WT can reach a third row pair; it is never stored: fold it back into the second)
                const int rp = wq >= a.Wp ? 1 : 0, wp = wq - (wq >= 2 * a.Wp ? 2 : rp) * a.Wp;
                set_addresses(i, 2 * rp + (sub >> 1), 2 * wp + (sub & 1));
            }
        }
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#if R64_DIAG
        dt[0] += 1;
#endif
        R64_STAMP(1);

        // one channel chunk = 36 A-fragment reads (tap-major, then k-step, then row tile), each feeding one MFMA.  An LDS read
        // takes ~100+ cycles and an MFMA 32, so the reads run RD - 1 fragments ahead of their use through a register ring.
        // hipcc serialises read -> wait -> MFMA on one register when left alone (it minimises pressure), so the reads and
        // their counted waits are inline asm: the wait statement names the fragment it completes ("+v"), which orders the
        // MFMA behind it; no other LDS / scalar-memory operation is in flight inside a chunk (drained on entry).
        auto chunk = [&](int pl, auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int NRD = 36, RD = R64_DEPTH;
            unsigned ta[2][3][2];
            const unsigned pb = lds_base + pl * PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < 2; ++s) ta[i][dx][s] = areg[i][dx][s] + pb;
            u32x4 ring[RD];
            auto rd = [&](auto jc) {
                constexpr int j = decltype(jc)::value, tt = j >> 2, s = (j >> 1) & 1, i = j & 1, dy = tt / 3, dx = tt % 3;
                ring[j % RD] = lds_read16<dy * P * R64_ROWB>(ta[i][dx][s]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<0, RD - 1>(rd);
            static_for<0, NRD>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + RD - 1 < NRD) rd(std::integral_constant<int, j + RD - 1>{});
                constexpr int left = NRD - 1 - j;                     // reads issued after fragment j
                u32x4 f = ring[j % RD];
                f = lds_wait<(left >= RD - 1 ? RD - 1 : left)>(f);
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f),
                                                                      __builtin_bit_cast(bf16x8, bw[c][j >> 2][(j >> 1) & 1]), acc[j & 1], 0, 0, 0);
            });
        };
        if (SC) {                                                   // shortcut GEMM first: its plane is re-staged at the mid barrier
#pragma unroll
            for (int s = 0; s < (IMG ? 1 : 2); ++s) {
                const uint4 a0 = *reinterpret_cast<const uint4*>(smem + X_BASE + P * R64_ROWB + areg[0][1][s]);
                const uint4 a1 = *reinterpret_cast<const uint4*>(smem + X_BASE + P * R64_ROWB + areg[1][1][s]);
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a0), __builtin_bit_cast(bf16x8, bw2[s]), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1), __builtin_bit_cast(bf16x8, bw2[s]), acc[1], 0, 0, 0);
            }
        }
        chunk(p0, std::integral_constant<int, 0>{});
        // every wave has finished reading plane p0 (and the shortcut plane): their data fed MFMAs already issued
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(2);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(3);
        if (more) {
            if (IMG) img_load(nb, nh);                              // (ordinary loads first: the compiler's wait for them must not
                                                                    // cover the DMAs below, which are issued AFTER them)
            if (R64_SPLIT_STAGE && wid >= 4) stage(lds_base + pf * PLANE, 0, nb, nh);
            stage(lds_base + p0 * PLANE, 1, nb, nh);                // next tile's chunk 1 -> the plane chunk 0 just left
            if (SC && !IMG) stage(lds_base + X_BASE, 2, nb, nh);
        }
        chunk(p1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(4);

        // ---- epilogue: + shift, LeakyReLU, (2x2 max), bf16, through this wave's LDS slab, 16-byte stores
        // C layout of a 32x32 tile: column = lane % 32; register r holds row (r & 3) + 8 (r >> 2) + 4 (lane / 32)
        char* const slab = smem + SLAB_BASE + wid * SLAB;
        const float sh = s_shift[32 * wh + lr];
        bool ragged;
        if (!POOL) {
            // output rows of the tile = pixels [pix0, pix0 + nvalid): contiguous (whole image rows)
            const int rows_left = a.H - k_img * a.R, nvalid = (rows_left < a.R ? rows_left : a.R) * a.W;
            const long long pix0 = ((long long)b * a.H + (long long)k_img * a.R) * a.W;
            ragged = nvalid <= 224;                                 // the last 32-row slab's stores may be skipped altogether
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int jrow0 = wm * 64 + i * 32;
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
                    if (jrow0 + row < nvalid)
                        *reinterpret_cast<uint4*>(a.y + (size_t)(pix0 + jrow0 + row) * 128 + wh * 64 + c16 * 16) = val;
                }
            }
        } else {
            // pooled output rows = windows [win0, win0 + nv) of image b: contiguous in the [B Hp Wp][64] output
            const int win0 = k_img * a.WT, left = a.nwin - win0, nv = left < a.WT ? left : a.WT;
            const long long out0 = (long long)b * a.nwin + win0;
            ragged = nv <= 56;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int wrow0 = (wm * 64 + i * 32) >> 2;          // first window of this 32-row MFMA tile
#pragma unroll
                for (int q = 0; q < 4; ++q) {                       // registers 4q .. 4q+3 = rows 8q + 4 lh + {0..3} = one window
                    float best = fmaxf(fmaxf(acc[i][4 * q], acc[i][4 * q + 1]), fmaxf(acc[i][4 * q + 2], acc[i][4 * q + 3])) + sh;
                    if (a.act) best = fmaxf(best, best * 0.1f);     // monotone => lrelu(max) == max(lrelu)
                    *reinterpret_cast<__bf16*>(slab + (2 * q + lh) * SLAB_RS + lr * 2) = (__bf16)best;
                }
                const int row = lane >> 2, c16 = lane & 3;          // 8 pooled rows x 4 vectors: lanes 0..31
                if (lane < 32) {
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * SLAB_RS + c16 * 16);
                    if (wrow0 + row < nv)
                        *reinterpret_cast<uint4*>(a.y + (size_t)(out0 + wrow0 + row) * 128 + wh * 64 + c16 * 16) = val;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(5);
        // the next tile's DMAs are older than this epilogue's stores: wait for all but the youngest store instructions (4 LINEAR,
        // 2 POOL per wave); where store instructions may have been skipped altogether, wait for everything
        if (ragged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (POOL) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        R64_STAMP(6);
        // IMG: the next tile's shortcut pixels (loaded at the mid barrier, older than the DMAs just waited for) go to LDS now -
        // every wave left the shortcut GEMM before the mid barrier, the next one starts behind the barrier below
        if (IMG && more) img_store();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(7);
        b = nb; k_img = nk; h_first = nh; s0 = ns0;
    }
#if R64_DIAG
    if (lane == 0 && blockIdx.x * R64_NW + wid < 4096) {
        float* d = r64_diag + (size_t)(blockIdx.x * R64_NW + wid) * 12;
        d[10] = (float)(__builtin_amdgcn_s_memrealtime() - d_entry_r);   // resident time of this wave, 100 MHz ticks
        d[11] = 0.f;
        d[8] = (float)(d_c0 - d_entry);                             // prologue cycles
        d[9] = (float)(d_entry_r % 100000000ull);                   // kernel entry on the 100 MHz reference clock (mod 1 s)
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (float)dt[k];
        // clock of the loop: shader cycles per 100 MHz reference tick, x 1000 (reported in the "barrier2" slot's fraction: slot 7 keeps
        // its cycle count, the clock goes to the last float of the NEXT wave's unused padding - simpler: overwrite slot 6 + 7 sum)
        const float ghz = (float)(__builtin_amdgcn_s_memtime() - d_c0) / (float)(__builtin_amdgcn_s_memrealtime() - d_r0) * 0.1f;
        d[6] = dt[6] + dt[7];
        d[7] = ghz;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv3 of layer1.0 (3x3, 64 -> 64, + the 1x1 shortcut from the image + MaxPool2d(2), eval mode) re-cut after the fused kernel below:
// conv64_resident_kernel<true, true, 6, true> runs its eight waves in lockstep - patch staging (22 LDS-DMA pieces per wave and tile at ~100
// cycles of issue each), two barriers, epilogue: 51 % of a tile with the matrix pipe idle (in-kernel stamps, profiles/r04_ab_l1_conv3_stage.txt).
// Here:  * the image's three channels live in a compact [block][column][c0 c1 c2 0] bf16 patch (8 B per pixel, double-buffered, 6 KB) instead
//          of slot 0 of a fourth 36 KB plane; the freed LDS holds a FOURTH conv plane: two plane PAIRS (this tile's, the next tile's), so
//          the next tile's two chunks are staged any time during this tile and the tile has ONE barrier;
//        * the two waves of a SIMD stage at opposite ends of the tile: waves 0-3 issue their pieces first, waves 4-7 between their chunks
//          and their epilogue - one half's MFMAs run under the other half's DMA issue; the chunks run at a raised issue priority.
// Same tiles (63 windows = 1.5 row pairs, two alternating geometries), planes, swizzle, register-resident weights, chunk loop and pooled
// slab epilogue as conv64_resident_kernel.
constexpr int R64_PI_PLANE = 6 * R64_P * R64_ROWB, R64_PI_IMGB = 4 * R64_P * 8, R64_PI_SLAB = 8 * (32 * 2 + 16);
constexpr int R64_PI_LDS = 4 * R64_PI_PLANE + 2 * R64_PI_IMGB + R64_NW * R64_PI_SLAB + 256;
static_assert(R64_PI_LDS <= 160 * 1024, "LDS budget");
__global__ __launch_bounds__(R64_NW * 64, 2) void conv64_pool_img_kernel(const Conv64Args a) {
    constexpr int P = R64_P, BLOCKS = 6, PROWS = BLOCKS * P, PLANE = R64_PI_PLANE, PIECES = PROWS / 16, PPB = P / 16;
    constexpr int NPK = (PIECES + R64_NW - 1) / R64_NW;
    constexpr int SLAB_RS = 32 * 2 + 16, SLAB = R64_PI_SLAB, IMGB = R64_PI_IMGB;
    constexpr int IMG_BASE = 4 * PLANE, SLAB_BASE = IMG_BASE + 2 * IMGB, SHIFT_BASE = SLAB_BASE + R64_NW * SLAB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wh = wid & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (nwg + 7 - xcd) >> 3;
    const int per = (a.ntiles + 7) >> 3;
    const int t_begin = xcd * per + slot, t_end = min((xcd + 1) * per, a.ntiles);
    if (t_begin >= t_end) return;

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
    // shortcut: the packed 1x1 weights of the first layer sit at k = 12..14 of their row; the compact patch carries the channels at k = 0..2
    uint4 bw2;
    {
        const unsigned short* wr = reinterpret_cast<const unsigned short*>(a.w2 + (size_t)(32 * wh + lr) * R64_ROWB);
        const unsigned w12 = wr[12], w13 = wr[13], w14 = wr[14];
        bw2 = lh == 0 ? make_uint4(w12 | (w13 << 16), w14, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
    }
    for (int o = tid * 16; o < SLAB_BASE; o += R64_NW * 64 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    float* const s_shift = reinterpret_cast<float*>(smem + SHIFT_BASE);
    if (tid < 64) s_shift[tid] = a.shift[tid];
    __syncthreads();

    const int prl = lane >> 2, psl = lane & 3;
    unsigned goff[NPK];
    unsigned long long dmask[NPK];
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
        const int q = wid + R64_NW * k, idx = q * 16 + prl, rb = q / PPB, c = idx - rb * P;
        dmask[k] = __ballot(q < PIECES && c >= 1 && c <= a.W);
        goff[k] = (unsigned)(rb * a.W + c - 1) * 128u + ((unsigned)(psl ^ swz<4>(idx)) << 4);
    }
    // both 32-channel chunks of the patch whose block 0 is image row h_first of image b -> plane pair `pair`: one masked DMA statement per
    // piece and chunk; a block outside the image (first / last tile of an image) reads the zero line instead - a scalar select, no second
    // code path (two paths at three call sites cost 52 spilled SGPRs, read back by v_readlane inside the staging code)
    const unsigned zoff = (unsigned)psl << 4;
    auto stage_pair = [&](int pair, int b, int h_first) {
        const long long origin = ((long long)b * a.H + h_first) * a.W;
        const char* const src = a.x + origin * 128;
        const char* const zero = reinterpret_cast<const char*>(r64_zero_line);
        const unsigned dst = lds_base + 2 * pair * PLANE;
#pragma unroll
        for (int k = 0; k < NPK; ++k) {
            const int q = wid + R64_NW * k;
            if (q < PIECES && dmask[k]) {                               // wave-uniform
                const int h = h_first + q / PPB;
                const bool inside = h >= 0 && h < a.H;                  // wave-uniform: a block is in or out of the image
                const char* const s0 = inside ? src : zero;
                const unsigned vo = inside ? goff[k] : zoff;
                dma16_masked(s0, vo, dst + q * 1024, dmask[k]);
                dma16_masked(inside ? s0 + 64 : s0, vo, dst + PLANE + q * 1024, dmask[k]);
            }
        }
    };
    // the image under the tile's four non-halo blocks: thread (block ib, column ix) loads its pixel's three channels ...
    const int ib = tid >> 7, ix = tid & 127;
    float iv[3] = {0.f, 0.f, 0.f};
    auto img_load = [&](int b, int h_first) {
        const int h = h_first + 1 + ib;
        const bool ok = ix < a.W && h >= 0 && h < a.H;
        const size_t plane = (size_t)a.H * a.W;
        const float* p = a.img + (size_t)b * 3 * plane + (size_t)(ok ? h : 0) * a.W + (ok ? ix : 0);
        const float v0 = p[0], v1 = p[plane], v2 = p[2 * plane];
        iv[0] = ok ? v0 : 0.f; iv[1] = ok ? v1 : 0.f; iv[2] = ok ? v2 : 0.f;
    };
    // ... and writes them as [c0 c1 c2 0] bf16 into compact patch `buf`
    auto img_store = [&](int buf) {
        if (ix < a.W)
            *reinterpret_cast<uint2*>(smem + IMG_BASE + buf * IMGB + (ib * P + ix) * 8) = make_uint2(r64_pack(iv[0], iv[1]), r64_pack(iv[2], 0.f));
    };

    unsigned areg[2][3][2], ioff[2];
    auto set_addresses = [&](int i, int hrel, int w) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = hrel * P + w + dx;
            const unsigned ad = (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row));
            areg[i][dx][0] = ad;
            areg[i][dx][1] = ad ^ 32u;
        }
        ioff[i] = (unsigned)(hrel * P + w) * 8u;                      // the pixel itself in the compact image patch
    };
    auto tile_geom = [&](int t, int& b, int& k_img, int& h_first, int& s0) {
        const int bb = (int)fdiv((unsigned)t, a.d_tpi), k = t - bb * a.tpi;
        b = __builtin_amdgcn_readfirstlane(bb);
        k_img = __builtin_amdgcn_readfirstlane(k);
        const int win0 = k_img * a.WT, rp0 = (int)fdiv((unsigned)win0, a.d_wp);
        h_first = __builtin_amdgcn_readfirstlane(2 * rp0 - 1);
        s0 = __builtin_amdgcn_readfirstlane(win0 - rp0 * a.Wp);
    };

    int t = t_begin, b, k_img, h_first, s0;
    tile_geom(t, b, k_img, h_first, s0);
    img_load(b, h_first);
    stage_pair(0, b, h_first);
    img_store(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long d_c0 = dq, d_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int pp = it & 1;
        const int tn = t + nslot;
#if R64_DIAG
        dt[0] += 1;
#endif
        const bool more = tn < t_end;
        int nb = 0, nk = 0, nh = 0, ns0 = 0;
        if (more) {
            tile_geom(tn, nb, nk, nh, ns0);
            if (wid < 4 || !R64_PI_STAGGER) {                           // (ordinary loads first: the compiler's wait for them must not
                img_load(nb, nh);                                       //  cover the DMAs, which are issued AFTER them)
                stage_pair(pp ^ 1, nb, nh);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = wm * 64 + i * 32 + lr, wq = s0 + (j >> 2), sub = j & 3;
            const int rp = wq >= a.Wp ? 1 : 0, wp = wq - (wq >= 2 * a.Wp ? 2 : rp) * a.Wp;
            set_addresses(i, 2 * rp + (sub >> 1), 2 * wp + (sub & 1));
        }
        R64_STAMP(1);
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        auto chunk = [&](int pl, auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int NRD = 36, RD = R64_DEPTH;
            unsigned ta[2][3][2];
            const unsigned pb = lds_base + pl * PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < 2; ++s) ta[i][dx][s] = areg[i][dx][s] + pb;
            u32x4 ring[RD];
            auto rd = [&](auto jc) {
                constexpr int j = decltype(jc)::value, tt = j >> 2, s = (j >> 1) & 1, i = j & 1, dy = tt / 3, dx = tt % 3;
                ring[j % RD] = lds_read16<dy * P * R64_ROWB>(ta[i][dx][s]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<0, RD - 1>(rd);
            static_for<0, NRD>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + RD - 1 < NRD) rd(std::integral_constant<int, j + RD - 1>{});
                constexpr int left = NRD - 1 - j;
                u32x4 f = ring[j % RD];
                f = lds_wait<(left >= RD - 1 ? RD - 1 : left)>(f);
                acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f),
                                                                      __builtin_bit_cast(bf16x8, bw[c][j >> 2][(j >> 1) & 1]), acc[j & 1], 0, 0, 0);
            });
        };
        {   // shortcut GEMM: one k-step from the compact patch (every lane reads its pixel: the k = 8..15 lanes meet zero weights)
            const uint2 p0 = *reinterpret_cast<const uint2*>(smem + IMG_BASE + pp * IMGB + ioff[0]);
            const uint2 p1 = *reinterpret_cast<const uint2*>(smem + IMG_BASE + pp * IMGB + ioff[1]);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(p0.x, p0.y, 0u, 0u)), __builtin_bit_cast(bf16x8, bw2), acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, make_uint4(p1.x, p1.y, 0u, 0u)), __builtin_bit_cast(bf16x8, bw2), acc[1], 0, 0, 0);
        }
        if (R64_PI_PRIO) { if (wid >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
        chunk(2 * pp, std::integral_constant<int, 0>{});
        R64_STAMP(2);
        chunk(2 * pp + 1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        if (R64_PI_PRIO) __builtin_amdgcn_s_setprio(0);
        R64_STAMP(3);
        if (R64_PI_STAGGER && more && wid >= 4) {                                         // the later half stages here: its DMAs fly under its epilogue
            img_load(nb, nh);
            stage_pair(pp ^ 1, nb, nh);
        }
        R64_STAMP(4);
        // ---- epilogue: + shift, (2x2 max), LeakyReLU, bf16, through this wave's LDS slab, 16-byte stores (as conv64_resident_kernel)
        char* const slab = smem + SLAB_BASE + wid * SLAB;
        const float sh = s_shift[32 * wh + lr];
        const int win0 = k_img * a.WT, left = a.nwin - win0, nv = left < a.WT ? left : a.WT;
        const long long out0 = (long long)b * a.nwin + win0;
        const bool ragged = nv <= 56;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int wrow0 = (wm * 64 + i * 32) >> 2;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float best = fmaxf(fmaxf(acc[i][4 * q], acc[i][4 * q + 1]), fmaxf(acc[i][4 * q + 2], acc[i][4 * q + 3])) + sh;
                if (a.act) best = fmaxf(best, best * 0.1f);
                *reinterpret_cast<__bf16*>(slab + (2 * q + lh) * SLAB_RS + lr * 2) = (__bf16)best;
            }
            const int row = lane >> 2, c16 = lane & 3;
            if (lane < 32) {
                const uint4 val = *reinterpret_cast<const uint4*>(slab + row * SLAB_RS + c16 * 16);
                if (wrow0 + row < nv)
                    *reinterpret_cast<uint4*>(a.y + (size_t)(out0 + wrow0 + row) * 128 + wh * 64 + c16 * 16) = val;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // this wave's DMAs are older than the epilogue's two store instructions (fewer where a ragged tile skipped them: wait for all)
        R64_STAMP(5);
        if (ragged) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        R64_STAMP(6);
        if (more) img_store(pp ^ 1);
        __syncthreads();                                                // the next tile's planes and image patch are complete; this tile's are free
        R64_STAMP(7);
        b = nb; k_img = nk; h_first = nh; s0 = ns0;
    }
#if R64_DIAG
    if (lane == 0 && blockIdx.x * R64_NW + wid < 4096) {
        // [tiles, staging (waves 0-3) + addresses, chunk 0, chunk 1, staging (waves 4-7), epilogue, DMA wait, barrier, -, -, -, GHz]
        float* d = r64_diag + (size_t)(blockIdx.x * R64_NW + wid) * 12;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (float)dt[k];
        d[8] = d[9] = d[10] = 0.f;
        d[11] = (float)(__builtin_amdgcn_s_memtime() - d_c0) / (float)(__builtin_amdgcn_s_memrealtime() - d_r0) * 0.1f;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 -> conv2 of layer1.0 in ONE kernel (models/resnet_language.py:249-253: conv3x3(3 -> 64) + BN + LeakyReLU, then
// conv3x3(64 -> 64) + BN + LeakyReLU, eval mode): the 64-channel intermediate never leaves the chip.  The resident kernel above
// stages, per tile of 3 image rows, the 5 rows x W pixels x 64 channels of conv1's OUTPUT from HBM (128 B per pixel written by
// the conv1 kernel, 128 B read back: the two largest streams of the whole backbone).  Here the same planes are COMPUTED from the
// fp32 NCHW image instead: per tile 7 image rows x 3 channels arrive by LDS-DMA (12 B per pixel), are converted to the bf16
// [row][column][c0 c1 c2 1] patch of conv_first.hip, and a K = 48 MFMA pass (weights = conv1's packed matrix with its BN shift
// riding on the constant-1 channel, fragments kept in LDS) produces the 5 x W x 64 block straight into the swizzled, zero-bordered
// plane layout the register-resident conv2 reads.  Extra MFMA work: 48 / 576 x 5 / 3 = 14 % of conv2's; HBM traffic of the pair:
// 12 B + 128 B per pixel instead of 12 + 128 + 128 + 128.
// Pipeline per iteration (tile T): [DMA fp32 patch rows of T+2] -> [conv1 of T+1 from its bf16 patch into the other plane pair] ->
// [conv2 chunks of T from this pair, epilogue] -> [every wave converts the patch row it fetched itself into the other bf16 patch
// buffer] -> ONE barrier.  Four planes (two pairs) and two bf16 patch buffers, so conv1 of the next tile has no dependency on this
// tile's reads.  LINEAR tiles only (conv2 is not pooled).
constexpr int R64_FUSED_LDS = 4 * (5 * R64_PF * R64_ROWB) + 7 * 1024 + 2 * 7 * (96 + 4) * 8 + 6 * 64 * 16 + (4096 + 256) + 256;
struct Conv64FusedArgs {
    const float* img;    // [B][3][H][W] fp32
    const char* w1;      // [64][32] bf16, k = 3 tap + c, BN scale folded (subreg_pack_conv_weight mode 1)
    const float* shift1; // [64]
    const char* w;       // conv2: [9][2][64][32] bf16, BN scale folded
    const float* shift;  // [64]
    char* y;             // [npix][64] bf16
    int H, W, act, ntiles, tpi, R;
    FastDiv d_w, d_tpi;
};

__global__ __launch_bounds__(R64_NW * 64, 2) void conv64_fused_first_kernel(const Conv64FusedArgs a) {
    constexpr int P = R64_PF, BLOCKS = 5, PROWS = BLOCKS * P, PLANE = PROWS * R64_ROWB;
    constexpr int XROWS = BLOCKS + 2;                                   // image rows under a tile's 5 conv1 rows
    constexpr int XF_BASE = 4 * PLANE, XF_BYTES = XROWS * 1024;         // fp32 patch: ONE 1 KB DMA piece per image row, [c][x] (3 W <= 256 floats)
    constexpr int X4_BASE = XF_BASE + XF_BYTES, X4_ROW = (96 + 4) * 8, X4_BUF = XROWS * X4_ROW, X4_BYTES = 2 * X4_BUF;   // bf16 patch [2][row][x + 2][4]
    constexpr int WF_BASE = X4_BASE + X4_BYTES, WF_BYTES = 6 * 64 * 16; // conv1 A fragments [i][s][lane]
    constexpr int SCR_BASE = WF_BASE + WF_BYTES, SCR_BYTES = 4096 + 256; // set-up scratch: conv1's packed weights + its BN shift
    constexpr int SHIFT_BASE = SCR_BASE + SCR_BYTES;
    static_assert(SHIFT_BASE + 256 <= 160 * 1024 && SHIFT_BASE + 256 == R64_FUSED_LDS, "LDS budget");
    static_assert(PLANE % 16 == 0 && 2 * P * R64_ROWB < 65536, "plane bases are 16-byte aligned; tap offsets are 16-bit immediates");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wid >> 1, wh = wid & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int W = a.W, W4 = W >> 2;

    // ---- tile schedule: every workgroup walks ONE contiguous range of tiles, i.e. down its images (tile t + 1 is the three rows below
    //      tile t until the image ends), so that conv1 can ROLL: of the 5 conv1 rows under tile t + 1 the first two are the last two of
    //      tile t - copied from its planes, not recomputed (R64_FUSED_ROLL)
    const int nwg = gridDim.x, nslot = 1;
    const int per = a.ntiles / nwg, extra = a.ntiles - per * nwg;     // the first `extra` workgroups take one tile more
    const int t_begin = blockIdx.x * per + min((int)blockIdx.x, extra), t_end = t_begin + per + ((int)blockIdx.x < extra ? 1 : 0);
    if (t_begin >= t_end) return;

    // ---- conv2's resident weights (as in conv64_resident_kernel)
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
    // ---- LDS: planes and fp32 patch zero; bf16 patch = (0, 0, 0, 1) everywhere (its border columns are never rewritten)
    for (int o = tid * 16; o < X4_BASE; o += R64_NW * 64 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    for (int o = tid * 8; o < X4_BYTES; o += R64_NW * 64 * 8) *reinterpret_cast<uint2*>(smem + X4_BASE + o) = make_uint2(0u, 0x3F800000u);
    float* const s_shift = reinterpret_cast<float*>(smem + SHIFT_BASE);
    if (tid < 64) s_shift[tid] = a.shift[tid];
    // conv1's A fragments through the set-up scratch: fragment (i, s), lane (r, h) holds k' = 16 s + 8 h + j of output
    // channel 32 i + r, k' = 4 tap + c; c = 3: the BN shift (tap 4: bf16 hi part, tap 3: lo part) against the constant-1 channel
    {
        __bf16* const wl = reinterpret_cast<__bf16*>(smem + SCR_BASE);
        float* const sh1 = reinterpret_cast<float*>(smem + SCR_BASE + 4096);
        if (tid < 256) reinterpret_cast<uint4*>(wl)[tid] = reinterpret_cast<const uint4*>(a.w1)[tid];
        if (tid < 64) sh1[tid] = a.shift1[tid];
        __syncthreads();
        if (wid < 6) {                                                 // wave w builds fragment (i, s) = (w / 3, w % 3)
            const int i = wid / 3, s = wid % 3;
            const float shv = sh1[32 * i + lr];
            const __bf16 sh_hi = (__bf16)shv, sh_lo = (__bf16)(shv - (float)sh_hi);
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
            *reinterpret_cast<uint4*>(smem + WF_BASE + (wid * 64 + lane) * 16) =
                make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
        }
    }
    __syncthreads();

    // ---- tile -> (image, first tile row)
    auto tile_geom = [&](int t, int& b, int& k_img) {
        const int bb = (int)fdiv((unsigned)t, a.d_tpi);
        b = __builtin_amdgcn_readfirstlane(bb);
        k_img = __builtin_amdgcn_readfirstlane(t - bb * a.tpi);
    };
    // ---- fp32 patch by LDS-DMA: XF = [c][row r][x] floats, r = 0..6 <-> image row R k - 2 + r; piece q = floats [256 q, +256)
    const size_t plane_px = (size_t)a.H * W;
    // One piece per image row r of the patch, issued by wave r: lane l brings floats 4 l .. 4 l + 3 of [c][x] (c = 4 l / W; W % 4 == 0, so a
    // float4 never straddles two channel rows).  The wave that issued a row's piece is also the one that converts it (convert_row, after its
    // own vmcnt wait): no workgroup barrier stands between the DMA and the conversion.
    const int dq_c = (4 * lane) / W, dq_x = 4 * lane - dq_c * W;
    const bool dq_in = 4 * lane < 3 * W;
    const unsigned dq_off = (unsigned)((size_t)dq_c * plane_px + (size_t)dq_x);                      // < 3 B H W < 2^28 floats
    auto dma_patch = [&](int b, int k_img) {
        if (wid < XROWS) {
            const int h = a.R * k_img - 2 + wid;
            const bool ok = dq_in && h >= 0 && h < a.H;
            const char* src = ok ? reinterpret_cast<const char*>(a.img + (size_t)b * 3 * plane_px + dq_off + (unsigned)(h * W))
                                 : reinterpret_cast<const char*>(r64_zero_line);
            dma16_far(src, lds_base + XF_BASE + wid * 1024);
        }
    };
    // ---- fp32 row -> bf16 [row][x + 2][c0 c1 c2 1] of patch buffer `buf`: wave r, lane j = float4 column of its own row
    auto convert_row = [&](int buf) {
        if (wid < XROWS && lane < W4) {
            const char* const xr = smem + XF_BASE + wid * 1024;
            const float4 v0 = *reinterpret_cast<const float4*>(xr + (0 * W + 4 * lane) * 4);
            const float4 v1 = *reinterpret_cast<const float4*>(xr + (1 * W + 4 * lane) * 4);
            const float4 v2 = *reinterpret_cast<const float4*>(xr + (2 * W + 4 * lane) * 4);
            const unsigned one = 0x3F800000u;
            uint4* dst = reinterpret_cast<uint4*>(smem + X4_BASE + buf * X4_BUF + wid * X4_ROW + (4 * lane + 2) * 8);
            dst[0] = make_uint4(r64_pack(v0.x, v1.x), (r64_pack(v2.x, 0.f) & 0xffffu) | one, r64_pack(v0.y, v1.y), (r64_pack(v2.y, 0.f) & 0xffffu) | one);
            dst[1] = make_uint4(r64_pack(v0.z, v1.z), (r64_pack(v2.z, 0.f) & 0xffffu) | one, r64_pack(v0.w, v1.w), (r64_pack(v2.w, 0.f) & 0xffffu) | one);
        }
    };
    // ---- conv1 of the tile whose bf16 patch is in X4, into plane pair `pp`: 32-pixel groups of its 5 rows x W pixels; wave w
    //      takes groups w, w + 8.  Rows outside the image become zero rows (conv2's zero padding, not conv1 of zeros).
    int toff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int tap = 4 * s + 2 * lh + u;
            tap = tap < 9 ? tap : 8;
            toff[s][u] = (tap / 3) * X4_ROW + (tap % 3) * 8;
        }
    auto lrelu_pk = [&](float x0, float x1) -> unsigned {             // LeakyReLU(0.1) of two values as a bf16 pair
        float y0, y1;
        asm("v_max_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x0 * 0.1f));   // (fmaxf would add a canonicalising v_max per value)
        asm("v_max_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x1 * 0.1f));
        return r64_pack(y0, y1);
    };
    // EDGE (wave-uniform): some of the tile's 5 rows lie outside the image - only then do the stores need the per-lane select
    auto conv1_groups = [&](int k_img, int pp, int buf, int rb0, auto edge_c) {
        constexpr bool EDGE = decltype(edge_c)::value;
        const int npx = (BLOCKS - rb0) * W;                             // blocks rb0 .. 4 (rb0 = 2: the three rows a rolling tile adds)
        for (int g0 = (R64_FUSED_CONV1_HALF ? wid & 3 : wid) * 32; g0 < npx; g0 += (R64_FUSED_CONV1_HALF ? 4 : R64_NW) * 32) {
            const int p = g0 + lr;
            const bool valid = p < npx;
            const int rbl = valid ? (int)fdiv((unsigned)p, a.d_w) : 0, x = valid ? p - rbl * W : 0, rb = rb0 + rbl;
            const int h = a.R * k_img - 1 + rb;
            const bool inside = h >= 0 && h < a.H;
            const char* const base = smem + X4_BASE + buf * X4_BUF + rb * X4_ROW + (x + 1) * 8;
            f32x16 acc1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const uint2 t0 = *reinterpret_cast<const uint2*>(base + toff[s][0]);
                const uint2 t1 = *reinterpret_cast<const uint2*>(base + toff[s][1]);
                const uint4 xf = make_uint4(t0.x, t0.y, t1.x, t1.y);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 wfr = *reinterpret_cast<const uint4*>(smem + WF_BASE + ((i * 3 + s) * 64 + lane) * 16);
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr), __builtin_bit_cast(bf16x8, xf), acc1[i], 0, 0, 0);
                }
            }
            // lane (pixel lr, half lh) holds channels 32 i + 8 q + 4 lh + {0..3} in registers 4q..4q+3: 8 bytes of slot q of plane
            // (pair, chunk i), row rb P + 1 + x
            const int row = rb * P + 1 + x;
            const unsigned rbase = (unsigned)row * R64_ROWB + 8u * lh;
            const int sw = swz<4>(row);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned d0 = lrelu_pk(acc1[i][4 * q], acc1[i][4 * q + 1]), d1 = lrelu_pk(acc1[i][4 * q + 2], acc1[i][4 * q + 3]);
                    if (EDGE) { d0 = inside ? d0 : 0u; d1 = inside ? d1 : 0u; }
                    if (valid) *reinterpret_cast<uint2*>(smem + (2 * pp + i) * PLANE + rbase + 16u * (unsigned)(q ^ sw)) = make_uint2(d0, d1);
                }
        }
    };
    // roll: the tile is the one below the tile whose planes are pair pp ^ 1 - its blocks 0, 1 are that pair's blocks 3, 4 (a straight
    // copy of 16-byte slots: a block is 96 rows, a multiple of the swizzle period), only blocks 2 .. 4 are computed
    auto conv1_tile = [&](int k_img, int pp, int buf, bool roll) {
        const int h0 = a.R * k_img - 1;                                 // first of the 5 rows
        const int rb0 = roll ? 2 : 0;
        if (roll) {
            constexpr int SPAN = 2 * P * R64_ROWB;                      // two blocks of one plane
            constexpr int CPT = R64_FUSED_CONV1_HALF ? 256 : R64_NW * 64;   // threads that take part (the waves that run conv1)
            for (int o = (tid & (CPT - 1)) * 16; o < 2 * SPAN; o += CPT * 16) {
                const int c = o >= SPAN ? 1 : 0, off = o - c * SPAN;
                *reinterpret_cast<uint4*>(smem + (2 * pp + c) * PLANE + off) =
                    *reinterpret_cast<const uint4*>(smem + (2 * (pp ^ 1) + c) * PLANE + 3 * P * R64_ROWB + off);
            }
        }
        if (h0 < 0 || h0 + BLOCKS > a.H) conv1_groups(k_img, pp, buf, rb0, std::true_type{});
        else conv1_groups(k_img, pp, buf, rb0, std::false_type{});
    };

    // ---- conv2's per-lane A addresses (LINEAR: the same for every tile)
    unsigned areg[2][3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wm * 64 + i * 32 + lr, jv = j < a.R * W ? j : 0;
        const int ir = (int)fdiv((unsigned)jv, a.d_w), w = jv - ir * W;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = ir * P + w + dx;
            const unsigned ad = (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row));
            areg[i][dx][0] = ad;
            areg[i][dx][1] = ad ^ 32u;
        }
    }

    // ---- prologue: patch (buffer 0) + conv1 of the first tile, patch of the second (buffer 1)
    int t = t_begin, b, k_img;
    tile_geom(t, b, k_img);
    dma_patch(b, k_img);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_row(0);
    __syncthreads();
    conv1_tile(k_img, 0, 0, false);
    int nb = 0, nk = 0;
    if (t + nslot < t_end) {
        tile_geom(t + nslot, nb, nk);
        dma_patch(nb, nk);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        convert_row(1);
    }
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long d_c0 = dq, d_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int pp = it & 1;
        const bool more = t + nslot < t_end, more2 = t + 2 * nslot < t_end;
#if R64_DIAG
        dt[0] += 1;
#endif
        int b2 = 0, k2 = 0;
        if (more2) {
            tile_geom(t + 2 * nslot, b2, k2);
            if (R64_CUT != 5) dma_patch(b2, k2);                       // lands while this tile computes
        }
        // conv1 of the next tile needs nothing from this tile (its own plane pair, its own patch): the two waves of a SIMD (w and
        // w + 4) run it at opposite ends of the tile, so one wave's VALU / LDS-heavy conv1 phase meets its partner's MFMA chunks
        const bool roll = R64_FUSED_ROLL && nb == b && nk == k_img + 1;  // (wave-uniform) the next tile lies directly below this one
        if (R64_CUT != 3 && more && (wid < 4 || !R64_FUSED_STAGGER)) conv1_tile(nk, pp ^ 1, pp ^ 1, roll);
        R64_STAMP(1);
        // conv2 runs with SWAPPED MFMA operands (A = the resident weights, B = the pixel fragments): lane (lr, lh) then holds pixel lr
        // of a row tile and register r holds channel 32 wh + (r & 3) + 8 (r >> 2) + 4 lh - four consecutive channels per four
        // registers, which is what the LDS-free epilogue below needs.  The accumulators start from the BN shift (LDS reads that
        // complete under the chunk's own entry wait) instead of zero.
        f32x16 acc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = *reinterpret_cast<const float4*>(smem + SHIFT_BASE + (32 * wh + 8 * q + 4 * lh) * 4);
                acc[i][4 * q] = v.x; acc[i][4 * q + 1] = v.y; acc[i][4 * q + 2] = v.z; acc[i][4 * q + 3] = v.w;
            }
        asm volatile("" : "+v"(acc[0]), "+v"(acc[1]));                 // (the loads are complete before the hand-counted LDS reads start)
        auto chunk = [&](int pl, auto cc) {
            constexpr int c = decltype(cc)::value;
            constexpr int NRD = 36, RD = R64_DEPTH;
            unsigned ta[2][3][2];
            const unsigned pb = lds_base + pl * PLANE;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < 2; ++s) ta[i][dx][s] = areg[i][dx][s] + pb;
            u32x4 ring[RD];
            auto rd = [&](auto jc) {
                constexpr int j = decltype(jc)::value, tt = j >> 2, s = (j >> 1) & 1, i = j & 1, dy = tt / 3, dx = tt % 3;
                if (R64_CUT != 2) ring[j % RD] = lds_read16<dy * P * R64_ROWB>(ta[i][dx][s]);
            };
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            static_for<0, RD - 1>(rd);
            static_for<0, NRD>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (j + RD - 1 < NRD) rd(std::integral_constant<int, j + RD - 1>{});
                constexpr int left = NRD - 1 - j;
                u32x4 f = ring[j % RD];
                if (R64_CUT != 2) f = lds_wait<(left >= RD - 1 ? RD - 1 : left)>(f);
                if (R64_CUT != 1) acc[j & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[c][j >> 2][(j >> 1) & 1]),
                                                                      __builtin_bit_cast(bf16x8, f), acc[j & 1], 0, 0, 0);
            });
        };
#if R64_FUSED_PRIO
        // the MFMA phase runs at a raised issue priority, waves 4-7 (whose chunks come first in the tile) above waves 0-3: when wave w
        // leaves conv1 while its SIMD partner w + 4 is still in its second chunk, the partner keeps the matrix pipe, finishes, and
        // starts its long conv1 / epilogue phase under this wave's chunks - instead of being starved to the end of them (in-kernel
        // stamps, profiles/r04_l1_fused_stamps.txt: partner's chunk 1 3.9 k cycles, then 3 k cycles of conv1 with this wave idle)
        if (R64_FUSED_PRIO == 2) __builtin_amdgcn_s_setprio(0);          // (2: the other way round - the VALU-heavy phases run raised)
        else if (wid >= 4) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2);
#endif
        chunk(2 * pp, std::integral_constant<int, 0>{});
        R64_STAMP(2);
        chunk(2 * pp + 1, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
#if R64_FUSED_PRIO
        __builtin_amdgcn_s_setprio(R64_FUSED_PRIO == 2 ? 2 : 0);
#endif
        R64_STAMP(3);
        // ---- epilogue (LINEAR), no LDS: LeakyReLU, bf16 pairs, then v_permlane32_swap hands the upper lane's channels +4..7 of each
        //      group of 16 to the lower lane and the lower lane's +8..11 to the upper one: every lane stores 16 bytes = 8 consecutive
        //      channels of its pixel (a store instruction = 32 pixel rows x 32 contiguous bytes)
        const int rows_left = a.H - k_img * a.R, nvalid = (rows_left < a.R ? rows_left : a.R) * W;
        const long long pix0 = ((long long)b * a.H + (long long)k_img * a.R) * W;
        const float slope = a.act ? 0.1f : 1.f;
        auto lrelu2 = [&](float x0, float x1) -> unsigned {
            float y0, y1;
            asm("v_max_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x0 * slope));   // (fmaxf would add a canonicalising v_max per value)
            asm("v_max_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x1 * slope));
            return r64_pack(y0, y1);
        };
#pragma unroll
        for (int i = 0; i < (R64_CUT == 4 ? 0 : 2); ++i) {
            const int jrow = wm * 64 + i * 32 + lr;
            char* const yrow = a.y + (size_t)(pix0 + jrow) * 128 + wh * 64 + lh * 16;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const unsigned p0 = lrelu2(acc[i][8 * g2], acc[i][8 * g2 + 1]), p1 = lrelu2(acc[i][8 * g2 + 2], acc[i][8 * g2 + 3]);
                const unsigned p2 = lrelu2(acc[i][8 * g2 + 4], acc[i][8 * g2 + 5]), p3 = lrelu2(acc[i][8 * g2 + 6], acc[i][8 * g2 + 7]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(p0, p2, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p1, p3, false, false);
                const u32x4 vec = {s0[0], s1[0], s0[1], s1[1]};
                if (jrow < nvalid) *reinterpret_cast<u32x4*>(yrow + g2 * 32) = vec;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        R64_STAMP(4);
        if (R64_CUT != 3 && R64_FUSED_STAGGER && !R64_FUSED_CONV1_HALF && more && wid >= 4) conv1_tile(nk, pp ^ 1, pp ^ 1, roll);
        R64_STAMP(5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's patch row of tile t + 2 landed (and its stores are out)
        R64_STAMP(6);
        // the bf16 patch of t + 2 goes into the buffer tile t's patch had: its last readers (conv1 of t, one iteration ago) are behind
        // the previous barrier, and conv1 of t + 1 (this iteration) reads the other buffer - ONE barrier per tile
        if (more2 && R64_CUT != 5) convert_row(pp);
        __syncthreads();                                               // bf16 patch of t + 2 and the planes of t + 1 are complete
        R64_STAMP(7);
        b = nb; k_img = nk; nb = b2; nk = k2;
    }
#if R64_DIAG
    if (lane == 0 && blockIdx.x * R64_NW + wid < 4096) {
        // [tiles, conv1 (waves 0-3), chunk 0, chunk 1, epilogue, conv1 (waves 4-7), DMA wait + barrier, convert + barrier, -, -, -, GHz]
        float* d = r64_diag + (size_t)(blockIdx.x * R64_NW + wid) * 12;
#pragma unroll
        for (int k = 0; k < 8; ++k) d[k] = (float)dt[k];
        d[8] = d[9] = d[10] = 0.f;
        d[11] = (float)(__builtin_amdgcn_s_memtime() - d_c0) / (float)(__builtin_amdgcn_s_memrealtime() - d_r0) * 0.1f;
    }
#endif
}

// all but the n youngest vector-memory operations (LDS-DMAs and stores, in issue order) of this wave are done; n wave-uniform
__device__ __forceinline__ void vm_wait(int n) {
#define R64_VMW(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
    switch (n) {
        R64_VMW(0) R64_VMW(1) R64_VMW(2) R64_VMW(3) R64_VMW(4) R64_VMW(5) R64_VMW(6) R64_VMW(7) R64_VMW(8) R64_VMW(9) R64_VMW(10) R64_VMW(11)
        R64_VMW(12) R64_VMW(13) R64_VMW(14) R64_VMW(15) R64_VMW(16)
        default: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    }
#undef R64_VMW
}

// 2 KB of zeros: DMA source of patch blocks outside the image for the kernel below (a whole piece of 16 pixels x 128 bytes,
// so the lane offsets of a real piece work on it unchanged)
__device__ __attribute__((aligned(128))) unsigned r64_zero_piece[512];

// ---------------------------------------------------------------------------------------------------------------------
// The same convolution with ONE wave per SIMD (4 waves per workgroup, one workgroup per CU, 512 registers per lane).
// Why: the 8-wave kernel above is bound by the instructions AROUND its MFMAs - ~600 per wave and tile for 72 MFMAs, twice per
// SIMD (profiles/r02_ab_conv64_phase_shift.txt) - and its 256-register waves have no room left.  Here wave w owns rows
// [64 w, +64) x ALL 64 output channels:
//   * one A fragment (LDS read) feeds two MFMAs (both column halves): half the LDS reads and waits per MFMA;
//   * 288 registers of weights + 64 of accumulators live mostly in the accumulation registers (hipcc feeds MFMA operands
//     from a[...]), leaving room for an 8-deep fragment ring: LDS latency is covered without a partner wave;
//   * the tile is computed as two half-phases - row tile 0 (72 MFMAs over all of K), then row tile 1 - and everything else
//     is a FILLER between their MFMAs: the epilogue of the row tile finished one phase earlier (one accumulator register
//     per fragment) and the DMA pieces of the next tile's two planes (one per four fragments, in three small steps).  No
//     extra accumulators are needed: while row tile 0 of tile t+1 accumulates, row tile 1 of tile t is still in registers;
//   * the epilogue never touches LDS (an LDS write per fragment costs the MFMA stream ~5 cycles per MFMA,
//     tools/probes/mfma_fillers.hip): the MFMA operands are SWAPPED (A = weights, B = pixels), so a lane holds 4 consecutive
//     channels of ONE pixel per 4 registers; pairs are packed to bf16, v_permlane32_swap exchanges halves between the two
//     lanes of a pixel, and each lane stores 16 bytes = 8 consecutive channels (a store instruction = 32 pixel rows x 32
//     contiguous bytes).  The BN shift enters as one more k-step (A = shift split into bf16 hi + lo, B = ones): the first
//     MFMA of a phase, with C = 0;
//   * both planes of a tile stay live for the whole tile, so the ring has four planes (this tile's pair, the next tile's
//     pair, staged a whole tile ahead) and ONE barrier per tile.
// LINEAR only (layer1.0 conv2); same LDS image layout, tile schedule and data layouts as above.
template <int BLOCKS, bool ACT>
__global__ __launch_bounds__(256, 1) void conv64_wide_kernel(const Conv64Args a) {
    constexpr int P = R64_P, PROWS = BLOCKS * P, PLANE = PROWS * R64_ROWB, PIECES = PROWS / 16, PPB = P / 16, NW = 4;
    constexpr int NPK = (PIECES + NW - 1) / NW;                       // DMA pieces per wave and plane (8)
    constexpr int LDS_PLANES = 4 * PLANE;
    constexpr int RD = 8, NF = 36;                                    // fragment reads in flight (+1); fragments per half-phase
    static_assert(PLANE + 2 * P * R64_ROWB < 65536, "chunk + tap offsets are 16-bit immediates");
    static_assert(4 * (NPK - 1) + 2 <= 30 && NPK <= 8, "filler schedule");
    extern __shared__ __attribute__((aligned(16))) char smem[];
#if R64_DIAG
    const unsigned long long d_entry = __builtin_amdgcn_s_memtime(), d_entry_r = __builtin_amdgcn_s_memrealtime();
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslot = (nwg + 7 - xcd) >> 3;
    const int per = (a.ntiles + 7) >> 3;
    const int t_begin = xcd * per + slot, t_end = min((xcd + 1) * per, a.ntiles);
    if (t_begin >= t_end) return;

    // ---- resident weights: B fragments of all 64 output columns (two halves), all taps / chunks / k-steps: 288 registers.
    //      The first NWA fragments are moved into accumulation registers by hand ("=a": MFMA reads its B operand from
    //      a[...] directly); left to itself hipcc keeps what fits in v[...] and SPILLS the rest to a[...], copying four
    //      registers back in front of every MFMA that uses them (136 v_accvgpr_read per tile).
    constexpr int NWA = 48;
    u32x4 bw[72];
    static_for<0, 3>([&](auto bc) {                                   // three batches of 24 loads in flight, then their moves
        constexpr int b0 = decltype(bc)::value * 24;
        static_for<b0, b0 + 24>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, h = idx & 1, s = (idx >> 1) & 1, t = (idx >> 2) % 9, c = idx / 36;
            bw[idx] = *reinterpret_cast<const u32x4*>(a.w + (size_t)(32 * h + lr) * R64_ROWB + lh * 16 + (size_t)((t * 2 + c) * 64) * R64_ROWB + s * 32);
        });
        static_for<b0, b0 + 24>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            if constexpr (idx < NWA) {                                // from here on the fragment lives in a[...] (the compiler makes the move, after its own wait)
                u32x4 v = bw[idx];
                asm volatile("" : "+a"(v));
                bw[idx] = v;
            }
        });
    });
    // the BN shift as a k-step of its own: A rows = output channels, k = 0 / 1 hold the shift's bf16 hi / lo parts; B = ones
    u32x4 bias_a[2], ones_b;
    {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float sh = a.shift[32 * h + lr];
            const __bf16 hi = (__bf16)sh, lo = (__bf16)(sh - (float)hi);
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
            bias_a[h] = u32x4{lh == 0 ? pk : 0u, 0u, 0u, 0u};
        }
        ones_b = u32x4{lh == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u};
    }
    for (int o = tid * 16; o < LDS_PLANES; o += NW * 64 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    __syncthreads();

    // ---- DMA pieces of this wave (q = wid + 4 k): everything but the plane's base pointer is computed once
    const int prl = lane >> 2, psl = lane & 3;
    const unsigned glane = (unsigned)prl * 128u + ((unsigned)(psl ^ swz<4>(prl)) << 4);
    unsigned poff[NPK];
    unsigned long long emask[NPK];
    int prb[NPK];
#pragma unroll
    for (int k = 0; k < NPK; ++k) {
        const int q = wid + NW * k, rb = q / PPB, c0 = (q - rb * PPB) * 16, col = c0 + prl;
        poff[k] = (unsigned)__builtin_amdgcn_readfirstlane((rb * a.W + c0) * 128);
        emask[k] = __builtin_amdgcn_ballot_w64(col >= 1 && col <= a.W);
        prb[k] = __builtin_amdgcn_readfirstlane(rb);
    }
    int vm_issued = 0;                                               // vector-memory instructions of this wave so far (in-order completion)
    struct Src { const char* base; unsigned dst; int h_first; };    // base: one pixel BEFORE (block 0, column 1)
    auto make_src = [&](unsigned dst, int c, int b, int h_first) {
        Src r;
        r.base = a.x + (((long long)b * a.H + h_first) * a.W - 1) * 128 + c * 64;
        r.dst = dst + wid * 1024;
        r.h_first = h_first;
        return r;
    };
    const char* const zeros = reinterpret_cast<const char*>(r64_zero_piece);
    // a DMA piece in three filler steps (a: in the image?  b: source address  c: issue), values pinned where they are computed
    int dm_in[NPK];
    unsigned long long dm_src[NPK];
    auto dma_a = [&](const Src& src, auto kc) {
        constexpr int k = decltype(kc)::value;
        int in = __builtin_amdgcn_readfirstlane((unsigned)(src.h_first + prb[k]) < (unsigned)a.H ? 1 : 0);
        asm volatile("" : "+s"(in));
        dm_in[k] = in;
    };
    auto dma_b = [&](const Src& src, auto kc) {
        constexpr int k = decltype(kc)::value;
        const unsigned long long b64 = (unsigned long long)(size_t)(dm_in[k] ? src.base + poff[k] : zeros);
        unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b64 >> 32));
        asm volatile("" : "+s"(lo), "+s"(hi));
        dm_src[k] = ((unsigned long long)hi << 32) | lo;
    };
    auto dma_c = [&](const Src& src, auto kc) {
        constexpr int k = decltype(kc)::value;
        if (k < PIECES / NW || wid + NW * k < PIECES) {
            const unsigned lds_u = src.dst + k * (NW * 1024);
            unsigned keep_m0;
            asm volatile(
                "s_mov_b32 %0, m0\n\t"
                "s_mov_b64 exec, %1\n\t"
                "s_mov_b32 m0, %2\n\t"
                "s_nop 0\n\t"
                "global_load_lds_dwordx4 %3, %4\n\t"
                "s_mov_b32 m0, %0\n\t"
                "s_mov_b64 exec, -1"
                : "=&s"(keep_m0)
                : "s"(emask[k]), "s"(lds_u), "v"(glane), "s"(dm_src[k])
                : "memory");
            ++vm_issued;
        }
    };
    auto stage_all = [&](const Src& src) {
        static_for<0, NPK>([&](auto kc) { dma_a(src, kc); dma_b(src, kc); dma_c(src, kc); });
    };

    // ---- per-lane A addresses (absolute, plane pair a_set, chunk 0) of the dy = -1 row: [row tile][dx + 1][k-step];
    //      chunk 1 and the tap row are immediates (c PLANE + (dy + 1) P 64)
    unsigned areg[2][3][2];
    int a_set = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wid * 64 + i * 32 + lr, jv = j < a.R * a.W ? j : 0;
        const int ir = (int)fdiv((unsigned)jv, a.d_w), w = jv - ir * a.W;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = ir * P + w + dx;
            const unsigned ad = lds_base + (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row));
            areg[i][dx][0] = ad;
            areg[i][dx][1] = ad ^ 32u;
        }
    }
    struct Geom { int b, k_img, h_first; };
    auto tile_geom = [&](int t, Geom& g) {
        const int bb = (int)fdiv((unsigned)t, a.d_tpi), k = t - bb * a.tpi;
        g.b = __builtin_amdgcn_readfirstlane(bb);
        g.k_img = __builtin_amdgcn_readfirstlane(k);
        g.h_first = g.k_img * a.R - 1;
    };

    f32x16 acc[2][2];                                                 // [row tile][column half]
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][h][r] = 0.f;
    struct Pend { char* y; int nv; };                                 // where the pending row tile goes: its first output row, valid rows
    // ---- the epilogue of row tile O, one accumulator register per step v = 16 h + r.  C layout with swapped operands: lane =
    //      (pixel lr, half lh); register r of column half h = channel 32 h + (r & 3) + 8 (r >> 2) + 4 lh.  Registers 8 g2 .. + 7
    //      are packed to four bf16 pairs P0..P3 (channels +0,1 | +2,3 | +8,9 | +10,11 of 16 g2 + 4 lh); two half-swaps give the
    //      lower lane channels 16 g2 + 0..7 and the upper lane 16 g2 + 8..15 of the pixel: one 16-byte store each
    auto lrelu1 = [&](float x) {
        if (!ACT) return x;
        float y;
        asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(x * 0.1f));      // (fmaxf would add a canonicalising v_max per value)
        return y;
    };
    float ex[8];
    unsigned epk[4];
    const unsigned st_lane = (unsigned)lr * 128u + (unsigned)lh * 16u;    // byte offset of this lane's vector in a row tile's output
    auto epi_step = [&](auto oc, auto vc, const Pend& pe) {
        constexpr int O = decltype(oc)::value, v = decltype(vc)::value, h = v >> 4, r = v & 15, e = r & 7;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        ex[e] = lrelu1(acc[O][h][r]);
        if constexpr (e & 1) epk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ex[e - 1], ex[e]}, bf16x2));
        if constexpr (e == 7) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(epk[0], epk[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(epk[1], epk[3], false, false);
            const u32x4 vec = {s0[0], s1[0], s0[1], s1[1]};
            if (pe.nv > 0) {                                          // wave-uniform: the instruction exists
                if (lr < pe.nv) *reinterpret_cast<u32x4*>(pe.y + (h * 64 + (r >> 3) * 32) + st_lane) = vec;
                ++vm_issued;
            }
        }
    };

    // one half-phase: row tile I over all of K (36 fragments x 2 MFMAs) from the plane pair at areg; fillers: the pending
    // epilogue of row tile 1 - I, the DMA pieces of plane `src`
    int vm_mark = 0;                                                  // vm_issued after the last DMA piece of a phase
    auto phase = [&](auto ic, const Src& src, const Pend& pe) {
        constexpr int I = decltype(ic)::value, O = 1 - I;
        u32x4 ring[RD];
        auto rd = [&](auto jc) {
            constexpr int j = decltype(jc)::value, c = j / 18, tt = (j % 18) >> 1, s = j & 1, dy = tt / 3, dx = tt % 3;
            ring[j % RD] = lds_read16<c * PLANE + dy * P * R64_ROWB>(areg[I][dx][s]);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<0, RD - 1>(rd);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias_a[0]), __builtin_bit_cast(bf16x8, ones_b), zero16, 0, 0, 0);
        acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias_a[1]), __builtin_bit_cast(bf16x8, ones_b), zero16, 0, 0, 0);
        static_for<0, NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value, c = f / 18, tt = (f % 18) >> 1, s = f & 1;
            if constexpr (f + RD - 1 < NF) rd(std::integral_constant<int, f + RD - 1>{});
            // ---- fillers
            if constexpr (f < 32) epi_step(std::integral_constant<int, O>{}, fc, pe);
            if constexpr (f % 4 == 0 && f / 4 < NPK) dma_a(src, std::integral_constant<int, f / 4>{});
            if constexpr (f % 4 == 1 && f / 4 < NPK) dma_b(src, std::integral_constant<int, f / 4>{});
            if constexpr (f % 4 == 2 && f / 4 < NPK) dma_c(src, std::integral_constant<int, f / 4>{});
            if constexpr (f == 4 * (NPK - 1) + 2) vm_mark = vm_issued;
            // ---- fragment f
            constexpr int younger = NF - 1 - f < RD - 1 ? NF - 1 - f : RD - 1;
            u32x4 fr = ring[f % RD];
            fr = lds_wait<younger>(fr);
            acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[((c * 9 + tt) * 2 + s) * 2 + 0]), __builtin_bit_cast(bf16x8, fr), acc[I][0], 0, 0, 0);
            acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[((c * 9 + tt) * 2 + s) * 2 + 1]), __builtin_bit_cast(bf16x8, fr), acc[I][1], 0, 0, 0);
        });
    };

    Geom cur, nxt = {0, 0, 0};
    int t = t_begin;
    tile_geom(t, cur);
    stage_all(make_src(lds_base + 0 * PLANE, 0, cur.b, cur.h_first));
    stage_all(make_src(lds_base + 1 * PLANE, 1, cur.b, cur.h_first));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long d_c0 = dq, d_r0 = __builtin_amdgcn_s_memrealtime();
#define R64W_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dt[k] += n_ - dq; dq = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define R64W_STAMP(k) do { } while (0)
#endif
    Pend pend = {a.y, 0};                                             // nothing pending before the first tile (no valid rows)
    for (int it = 0; t < t_end; ++it, t += nslot) {
        const int set = it & 1;                                       // this tile's plane pair: planes 2 set, 2 set + 1
        if (set != a_set) {
            const unsigned delta = (unsigned)((set - a_set) * 2 * PLANE);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < 2; ++s) areg[i][dx][s] += delta;
            a_set = set;
        }
        // the next tile's planes are staged during this one; after the last tile: this tile's patch once more (never read)
        if (t + nslot < t_end) tile_geom(t + nslot, nxt); else nxt = cur;
        const Src src0 = make_src(lds_base + (2 * (set ^ 1)) * PLANE, 0, nxt.b, nxt.h_first);
        const Src src1 = make_src(lds_base + (2 * (set ^ 1) + 1) * PLANE, 1, nxt.b, nxt.h_first);
        const int rows_left = a.H - cur.k_img * a.R, nvalid = (rows_left < a.R ? rows_left : a.R) * a.W;
        char* const ytile = a.y + (((long long)cur.b * a.H + (long long)cur.k_img * a.R) * a.W + wid * 64) * 128;   // wave-uniform
#if R64_DIAG
        dt[0] += 1;
#endif
        R64W_STAMP(1);
        phase(std::integral_constant<int, 0>{}, src0, pend);           // row tile 0 | epilogue of the previous tile's row tile 1
        R64W_STAMP(2);
        pend.y = ytile;
        pend.nv = nvalid - wid * 64;
        phase(std::integral_constant<int, 1>{}, src1, pend);           // row tile 1 | epilogue of this tile's row tile 0
        R64W_STAMP(4);
        pend.y = ytile + 32 * 128;
        pend.nv = nvalid - wid * 64 - 32;
        // the next tile's planes have landed (this wave's pieces); the stores behind them may stay in flight
        vm_wait(vm_issued - vm_mark);
        R64W_STAMP(5);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        R64W_STAMP(6);
        cur = nxt;
    }
    // ---- the last tile's row tile 1
    static_for<0, 32>([&](auto vc) { epi_step(std::integral_constant<int, 1>{}, vc, pend); });
#if R64_DIAG
    if (lane == 0 && blockIdx.x * 8 + wid < 4096) {                  // (same table as the 8-wave kernel: rows wid 4..7 stay empty)
        float* d = r64_diag + (size_t)(blockIdx.x * 8 + wid) * 12;
        d[10] = (float)(__builtin_amdgcn_s_memrealtime() - d_entry_r);
        d[11] = 0.f;
        d[8] = (float)(d_c0 - d_entry);
        d[9] = (float)(d_entry_r % 100000000ull);
#pragma unroll
        for (int q = 0; q < 7; ++q) d[q] = (float)dt[q];
        d[7] = (float)(__builtin_amdgcn_s_memtime() - d_c0) / (float)(__builtin_amdgcn_s_memrealtime() - d_r0) * 0.1f;
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// conv1 -> conv2 of layer1.0 with ONE wave per SIMD: conv64_wide_kernel's filler structure with conv64_fused_first_kernel's
// data flow.  The 8-wave fused kernel spends 36 % of a tile with the matrix pipe idle: a conv1 wave beside a partner in its
// conv2 chunks gets neither the matrix pipe nor the VALU port (profiles/r04_l1_fused_stamps.txt).  Here a wave's conv1 work for the
// NEXT tile is a filler between its own conv2 MFMAs: per half-phase (72 + 2 MFMAs of one 32-row tile) one 32-pixel group of
// conv1 - 6 pixel-fragment reads (ds_read_b64 from the bf16 patch), 6 weight-fragment reads, 6 MFMAs in two half-groups, 16
// LeakyReLU pairs, 8 ds_write_b64 into the next tile's planes - beside the LDS-free epilogue of the row tile finished a phase
// earlier.  conv1's accumulators (2 x 16 registers) take the registers the pending row tile's epilogue has just freed.
// All filler LDS operations are asm statements counted into the fragment ring's hand-made lgkmcnt waits (fill_lds below);
// a filler read issued in slot r is complete after the ring wait of slot r + RD (LDS returns in order).
// Tiles: contiguous ranges per workgroup (rolling conv1: 3 new image rows per tile, the two shared rows copied from the tile
// above); the first tile of an image computes its two extra rows in a plain loop.
constexpr int R64_WF_LDS = 4 * (5 * R64_P * R64_ROWB) + 7 * 1024 + 2 * 7 * (96 + 4) * 8 + 6 * 64 * 16 + (4096 + 256) + 2048;
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int OFF>
__device__ __forceinline__ u32x2 lds_read8(unsigned lds_addr) {
    u32x2 d;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "n"(OFF));
    return d;
}
__device__ __forceinline__ void lds_write8(unsigned lds_addr, unsigned lo, unsigned hi) {
    const u32x2 v = {lo, hi};
    asm volatile("ds_write_b64 %0, %1" : : "v"(lds_addr), "v"(v) : "memory");
}
// filler LDS operations of slot f of a half-phase (conv64_wide_fused_kernel's schedule: one per slot at most)
__host__ __device__ constexpr int r64wf_fill_lds(int f) {
    return ((f >= 0 && f < 12) || (f >= 19 && f < 35 && ((f - 19) & 1) == 0) ? 1 : 0) + (f >= 28 && f < 34 && ((f - 28) & 1) == 0 ? 1 : 0);
}
__host__ __device__ constexpr int r64wf_fill_window(int f, int rd) {
    int n = 0;
    for (int j = f - rd + 1; j <= f; ++j) n += r64wf_fill_lds(j);
    return n;
}

template <bool ACT>
__global__ __launch_bounds__(256, 1) void conv64_wide_fused_kernel(const Conv64FusedArgs a) {
    constexpr int P = R64_P, BLOCKS = 5, PROWS = BLOCKS * P, PLANE = PROWS * R64_ROWB, NW = 4;
    constexpr int XROWS = BLOCKS + 2;
    constexpr int XF_BASE = 4 * PLANE, XF_BYTES = XROWS * 1024;
    constexpr int X4_BASE = XF_BASE + XF_BYTES, X4_ROW = (96 + 4) * 8, X4_BUF = XROWS * X4_ROW, X4_BYTES = 2 * X4_BUF;
    constexpr int WF_BASE = X4_BASE + X4_BYTES, WF_BYTES = 6 * 64 * 16;
    constexpr int SCR_BASE = WF_BASE + WF_BYTES, SCR_BYTES = 4096 + 256;      // set-up scratch; afterwards the dump for the stores of pixels beyond the tile
    constexpr int ZR_BASE = SCR_BASE + SCR_BYTES, ZR_BYTES = 2048;          // zeros: what conv1 reads for a row outside the image (its output is then LeakyReLU(0) = 0: conv2's padding)
    static_assert(ZR_BASE + ZR_BYTES == R64_WF_LDS && R64_WF_LDS <= 160 * 1024 && 2 * X4_ROW + 16 + 8 <= ZR_BYTES, "LDS budget");
    static_assert(PLANE + 2 * P * R64_ROWB < 65536 && 5 * 1024 + 1024 < 65536, "immediate offsets");
    constexpr int RD = 8, NF = 36;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const int W = a.W, W4 = W >> 2;

    const int nwg = gridDim.x;
    const int per = a.ntiles / nwg, extra = a.ntiles - per * nwg;
    const int t_begin = blockIdx.x * per + min((int)blockIdx.x, extra), t_end = t_begin + per + ((int)blockIdx.x < extra ? 1 : 0);
    if (t_begin >= t_end) return;

    // ---- conv2's resident weights (conv64_wide_kernel)
    constexpr int NWA = 48;
    u32x4 bw[72];
    static_for<0, 3>([&](auto bc) {
        constexpr int b0 = decltype(bc)::value * 24;
        static_for<b0, b0 + 24>([&](auto ic) {
            constexpr int idx = decltype(ic)::value, h = idx & 1, s = (idx >> 1) & 1, t = (idx >> 2) % 9, c = idx / 36;
            bw[idx] = *reinterpret_cast<const u32x4*>(a.w + (size_t)(32 * h + lr) * R64_ROWB + lh * 16 + (size_t)((t * 2 + c) * 64) * R64_ROWB + s * 32);
        });
        static_for<b0, b0 + 24>([&](auto ic) {
            constexpr int idx = decltype(ic)::value;
            if constexpr (idx < NWA) {
                u32x4 v = bw[idx];
                asm volatile("" : "+a"(v));
                bw[idx] = v;
            }
        });
    });
    u32x4 bias_a[2], ones_b;
    {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float sh = a.shift[32 * h + lr];
            const __bf16 hi = (__bf16)sh, lo = (__bf16)(sh - (float)hi);
            const unsigned pk = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
            bias_a[h] = u32x4{lh == 0 ? pk : 0u, 0u, 0u, 0u};
        }
        ones_b = u32x4{lh == 0 ? 0x3f803f80u : 0u, 0u, 0u, 0u};
    }
    // ---- LDS: planes and fp32 patch zero; bf16 patch = (0, 0, 0, 1) everywhere; conv1's A fragments (conv64_fused_first_kernel)
    for (int o = tid * 16; o < X4_BASE; o += NW * 64 * 16) *reinterpret_cast<uint4*>(smem + o) = make_uint4(0, 0, 0, 0);
    for (int o = tid * 8; o < X4_BYTES; o += NW * 64 * 8) *reinterpret_cast<uint2*>(smem + X4_BASE + o) = make_uint2(0u, 0x3F800000u);
    {
        __bf16* const wl = reinterpret_cast<__bf16*>(smem + SCR_BASE);
        float* const sh1 = reinterpret_cast<float*>(smem + SCR_BASE + 4096);
        reinterpret_cast<uint4*>(wl)[tid] = reinterpret_cast<const uint4*>(a.w1)[tid];
        if (tid < 64) sh1[tid] = a.shift1[tid];
        __syncthreads();
        for (int fi = wid; fi < 6; fi += NW) {
            const int i = fi / 3, s = fi % 3;
            const float shv = sh1[32 * i + lr];
            const __bf16 sh_hi = (__bf16)shv, sh_lo = (__bf16)(shv - (float)sh_hi);
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
            *reinterpret_cast<uint4*>(smem + WF_BASE + (fi * 64 + lane) * 16) =
                make_uint4(e[0] | ((unsigned)e[1] << 16), e[2] | ((unsigned)e[3] << 16), e[4] | ((unsigned)e[5] << 16), e[6] | ((unsigned)e[7] << 16));
        }
    }
    for (int o = tid * 8; o < ZR_BYTES; o += NW * 64 * 8) *reinterpret_cast<uint2*>(smem + ZR_BASE + o) = make_uint2(0u, 0u);
    __syncthreads();

    struct Geom { int b, k_img; };
    auto tile_geom = [&](int t, Geom& g) {
        const int bb = (int)fdiv((unsigned)t, a.d_tpi);
        g.b = __builtin_amdgcn_readfirstlane(bb);
        g.k_img = __builtin_amdgcn_readfirstlane(t - bb * a.tpi);
    };
    // ---- fp32 patch rows by LDS-DMA (wave w: rows w and w + 4), converted by the wave that fetched them
    int vm_issued = 0;
    const size_t plane_px = (size_t)a.H * W;
    // lane l brings floats 4 l .. 4 l + 3 of a row's [c][x] block (3 W floats; the lanes beyond it repeat the last one: never converted)
    const int dq_f = 4 * lane < 3 * W ? 4 * lane : 3 * W - 4, dq_c = dq_f / W, dq_x = dq_f - dq_c * W;
    const unsigned dq_voff = (unsigned)(((size_t)dq_c * plane_px + (size_t)dq_x) * 4);                  // < 3 H W floats
    const unsigned zq_voff = (unsigned)(lane & 7) * 16u;
    auto dma_patch = [&](const Geom& g) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wid + NW * rr;
            if (r < XROWS) {
                const int h = a.R * g.k_img - 2 + r;                     // wave-uniform
                const bool ok = h >= 0 && h < a.H;
                const char* base = ok ? reinterpret_cast<const char*>(a.img + ((size_t)g.b * 3 * plane_px + (size_t)h * W))
                                      : reinterpret_cast<const char*>(r64_zero_line);
                dma16(base, ok ? dq_voff : zq_voff, lds_base + XF_BASE + r * 1024);
                ++vm_issued;
            }
        }
    };
    auto convert_rows = [&](int buf) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int r = wid + NW * rr;
            if (r < XROWS && lane < W4) {
                const char* const xr = smem + XF_BASE + r * 1024;
                const float4 v0 = *reinterpret_cast<const float4*>(xr + (0 * W + 4 * lane) * 4);
                const float4 v1 = *reinterpret_cast<const float4*>(xr + (1 * W + 4 * lane) * 4);
                const float4 v2 = *reinterpret_cast<const float4*>(xr + (2 * W + 4 * lane) * 4);
                const unsigned one = 0x3F800000u;
                uint4* dst = reinterpret_cast<uint4*>(smem + X4_BASE + buf * X4_BUF + r * X4_ROW + (4 * lane + 2) * 8);
                dst[0] = make_uint4(r64_pack(v0.x, v1.x), (r64_pack(v2.x, 0.f) & 0xffffu) | one, r64_pack(v0.y, v1.y), (r64_pack(v2.y, 0.f) & 0xffffu) | one);
                dst[1] = make_uint4(r64_pack(v0.z, v1.z), (r64_pack(v2.z, 0.f) & 0xffffu) | one, r64_pack(v0.w, v1.w), (r64_pack(v2.w, 0.f) & 0xffffu) | one);
            }
        }
    };
    // ---- conv1, plain form (prologue; blocks 0-1 of a tile that starts an image): 32-pixel groups of blocks rb0 .. rb0 + nblk - 1
    int toff[3][2];
#pragma unroll
    for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            int tap = 4 * s + 2 * lh + u;
            tap = tap < 9 ? tap : 8;
            toff[s][u] = (tap / 3) * X4_ROW + (tap % 3) * 8;
        }
    auto lrelu_pk = [&](float x0, float x1) -> unsigned {
        float y0, y1;
        asm("v_max_f32 %0, %1, %2" : "=v"(y0) : "v"(x0), "v"(x0 * 0.1f));
        asm("v_max_f32 %0, %1, %2" : "=v"(y1) : "v"(x1), "v"(x1 * 0.1f));
        return r64_pack(y0, y1);
    };
    // (acc1: two accumulators that are FREE where this runs - between tiles the row tile 0 pair; a fresh pair would make hipcc park the
    //  pending row tile 1 in VGPRs at every loop head)
    auto conv1_plain = [&](int k_img, int pp, int buf, int rb0, int nblk, f32x16 (&acc1)[2]) {
        const int npx = nblk * W;
        for (int g0 = wid * 32; g0 < npx; g0 += NW * 32) {
            const int p = g0 + lr;
            const bool valid = p < npx;
            const int rbl = valid ? (int)fdiv((unsigned)p, a.d_w) : 0, x = valid ? p - rbl * W : 0, rb = rb0 + rbl;
            const int h = a.R * k_img - 1 + rb;
            const bool inside = h >= 0 && h < a.H;
            const char* const base = smem + X4_BASE + buf * X4_BUF + rb * X4_ROW + (x + 1) * 8;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const uint2 t0 = *reinterpret_cast<const uint2*>(base + toff[s][0]);
                const uint2 t1 = *reinterpret_cast<const uint2*>(base + toff[s][1]);
                const uint4 xf = make_uint4(t0.x, t0.y, t1.x, t1.y);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const uint4 wfr = *reinterpret_cast<const uint4*>(smem + WF_BASE + ((i * 3 + s) * 64 + lane) * 16);
                    acc1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfr), __builtin_bit_cast(bf16x8, xf), acc1[i], 0, 0, 0);
                }
            }
            const int row = rb * P + 1 + x;
            const unsigned rbase = (unsigned)row * R64_ROWB + 8u * lh;
            const int sw = swz<4>(row);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned d0 = lrelu_pk(acc1[i][4 * q], acc1[i][4 * q + 1]), d1 = lrelu_pk(acc1[i][4 * q + 2], acc1[i][4 * q + 3]);
                    d0 = inside ? d0 : 0u; d1 = inside ? d1 : 0u;
                    if (valid) *reinterpret_cast<uint2*>(smem + (2 * pp + i) * PLANE + rbase + 16u * (unsigned)(q ^ sw)) = make_uint2(d0, d1);
                }
        }
    };
    // ---- conv1, filler form: this wave's two groups of blocks 2 .. 4 (group wid in phase 0, wid + 4 in phase 1); everything that does
    //      not depend on the tile is computed here
    unsigned c1_src[2], c1_dst[2];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
        const int p = 32 * (wid + NW * gi) + lr;
        const bool valid = p < 3 * W;
        const int rbl = valid ? (int)fdiv((unsigned)p, a.d_w) : 0, x = valid ? p - rbl * W : 0, rb = 2 + rbl;
        const int row = rb * P + 1 + x;
        c1_src[gi] = lds_base + X4_BASE + (unsigned)(rb * X4_ROW + (x + 1) * 8) + ((unsigned)rb << 27);     // (block index rb in bits 27..29: see c1_keep)
        // stores: slot q of the row goes to byte 16 (q ^ swz(row)); pixels beyond the tile write into the dump
        c1_dst[gi] = valid ? (unsigned)row * R64_ROWB + 8u * lh + 16u * (unsigned)swz<4>(row) : 0x80000000u + (unsigned)lane * 64u + 8u * lh;
    }
    const unsigned wf_addr = lds_base + WF_BASE + (unsigned)lane * 16u;
    // roll copy: 2 planes x 2 blocks = 1536 slots of 16 bytes, six per thread (three per half-phase): offsets inside a plane pair
    unsigned cp_off[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        constexpr int SPAN = 2 * P * R64_ROWB;
        const int o = (tid + k * NW * 64) * 16, c = o >= SPAN ? 1 : 0;
        cp_off[k] = (unsigned)(c * PLANE + (o - c * SPAN));
    }

    // ---- conv2's per-lane A addresses (conv64_wide_kernel)
    unsigned areg[2][3][2];
    int a_set = 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = wid * 64 + i * 32 + lr, jv = j < a.R * W ? j : 0;
        const int ir = (int)fdiv((unsigned)jv, a.d_w), w = jv - ir * W;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int row = ir * P + w + dx;
            const unsigned ad = lds_base + (unsigned)row * R64_ROWB + 16u * (lh ^ swz<4>(row));
            areg[i][dx][0] = ad;
            areg[i][dx][1] = ad ^ 32u;
        }
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[1][h][r] = 0.f;
    struct Pend { char* y; int nv; };
    auto lrelu1 = [&](float x) {
        if (!ACT) return x;
        float y;
        asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(x), "v"(x * 0.1f));
        return y;
    };
    float ex[8];
    unsigned epk[4];
    const unsigned st_lane = (unsigned)lr * 128u + (unsigned)lh * 16u;
    auto epi_step = [&](auto oc, auto vc, const Pend& pe) {
        constexpr int O = decltype(oc)::value, v = decltype(vc)::value, h = v >> 4, r = v & 15, e = r & 7;
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        // ONE v_accvgpr_read per value (left alone hipcc reads it again for the max).  The compiler hoists the 32 reads of a pending row
        // tile to the loop head; pinning them to their slots (asm volatile v_accvgpr_read) costs more: phases 3.9 k cycles against 3.5 k
        float xv = acc[O][h][r];
        asm("" : "+v"(xv));
        ex[e] = lrelu1(xv);
        if constexpr (e & 1) epk[e >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{ex[e - 1], ex[e]}, bf16x2));
        if constexpr (e == 7) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(epk[0], epk[2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(epk[1], epk[3], false, false);
            const u32x4 vec = {s0[0], s1[0], s0[1], s1[1]};
            if (pe.nv > 0) {
                if (lr < pe.nv) *reinterpret_cast<u32x4*>(pe.y + (h * 64 + (r >> 3) * 32) + st_lane) = vec;
                ++vm_issued;
            }
        }
    };

    // per-tile values of the conv1 fillers: byte offset of the next tile's bf16 patch buffer, its plane pair's base, and per group the
    // mask that zeroes rows outside the image (conv2's zero padding is NOT conv1 of zeros)
    struct C1Tile { unsigned psrc[2], pbase, dump, cp_src, cp_dst; };
    // one half-phase: row tile I over all of K from the plane pair at areg | fillers: the pending row tile's epilogue (two registers per
    // slot, slots 0-15), conv1 group I of the next tile (reads slots 0-11, MFMAs slots 14-19, LeakyReLU + stores slots 19-34)
    auto phase = [&](auto ic, const Pend& pe, const C1Tile& c1) {
        constexpr int I = decltype(ic)::value, O = 1 - I;
        u32x4 ring[RD];
        auto rd = [&](auto jc) {
            constexpr int j = decltype(jc)::value, c = j / 18, tt = (j % 18) >> 1, s = j & 1, dy = tt / 3, dx = tt % 3;
            ring[j % RD] = lds_read16<c * PLANE + dy * P * R64_ROWB>(areg[I][dx][s]);
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        static_for<0, RD - 1>(rd);
        const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias_a[0]), __builtin_bit_cast(bf16x8, ones_b), zero16, 0, 0, 0);
        acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bias_a[1]), __builtin_bit_cast(bf16x8, ones_b), zero16, 0, 0, 0);
        u32x2 px[6];
        u32x4 wf[6];
        // conv1's accumulators ARE the pending row tile's: acc[O][i] is free once its epilogue steps (slots 8 i .. 8 i + 7) have read it, and
        // is given back (everything stored) before the next half-phase starts accumulating into it
        f32x16 (&c1a)[2] = acc[O];
        if (R64WF_CUT == 4) { c1a[0] = zero16; c1a[1] = zero16; }
        u32x4 cp[3];                                                   // roll copy: three 16-byte slots per thread and half-phase
        const unsigned psrc = c1.psrc[I];
        const bool dumped = (c1_dst[I] & 0x80000000u) != 0;
        unsigned pdst[2];
        pdst[0] = dumped ? c1.dump + (c1_dst[I] & 0x7fffffffu) : c1.pbase + c1_dst[I];
        pdst[1] = pdst[0] + (dumped ? 0u : (unsigned)PLANE);           // channel chunk 1 lives one plane further on
        static_for<0, NF>([&](auto fc) {
            constexpr int f = decltype(fc)::value, c = f / 18, tt = (f % 18) >> 1, s = f & 1;
            if constexpr (f + RD - 1 < NF) rd(std::integral_constant<int, f + RD - 1>{});
            // ---- fillers (LDS operations: exactly r64wf_fill_lds(f) of them, before the ring wait)
            if constexpr (f < 16 && R64WF_CUT != 2) {
                epi_step(std::integral_constant<int, O>{}, std::integral_constant<int, 2 * f>{}, pe);
                epi_step(std::integral_constant<int, O>{}, std::integral_constant<int, 2 * f + 1>{}, pe);
            }
            if constexpr (f < 6) px[f] = lds_read8<0>(psrc + (unsigned)toff[f >> 1][f & 1]);
            if constexpr (f >= 6 && f < 12) wf[f - 6] = lds_read16<(f - 6) * 1024>(wf_addr);
            if constexpr (f >= 19 && f < 35 && ((f - 19) & 1) == 0 && R64WF_CUT != 1 && R64WF_CUT != 5) {
                constexpr int e = (f - 19) >> 1, i = e >> 2, q = e & 3;
                float v0 = c1a[i][4 * q], v1 = c1a[i][4 * q + 1], v2 = c1a[i][4 * q + 2], v3 = c1a[i][4 * q + 3];
                asm("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
                const unsigned d0 = lrelu_pk(v0, v1), d1 = lrelu_pk(v2, v3);
                lds_write8(pdst[i] ^ (16u * q), d0, d1);
            }
            if constexpr (f >= 28 && f < 34 && ((f - 28) & 1) == 0 && R64WF_CUT != 3) cp[(f - 28) >> 1] = lds_read16<0>(cp_off[I * 3 + ((f - 28) >> 1)] + c1.cp_src);
            // ---- fragment f
            constexpr int ring_younger = NF - 1 - f < RD - 1 ? NF - 1 - f : RD - 1;
            constexpr int younger = ring_younger + r64wf_fill_window(f, RD);
            static_assert(younger <= 15, "lgkmcnt is a 4-bit counter");
            u32x4 fr = ring[f % RD];
            fr = lds_wait<younger>(fr);
            if constexpr (f >= 14 && f < 20 && R64WF_CUT != 1 && R64WF_CUT != 4) {   // conv1 MFMA (i, s): its operands were read >= RD slots ago
                constexpr int m = f - 14, i = m / 3, sx = m % 3;
                u32x4 xf = {px[2 * sx][0], px[2 * sx][1], px[2 * sx + 1][0], px[2 * sx + 1][1]};
                u32x4 wv = wf[i * 3 + sx];
                asm volatile("" : "+v"(xf), "+v"(wv));
                if constexpr (sx == 0) c1a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xf), zero16, 0, 0, 0);
                else c1a[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wv), __builtin_bit_cast(bf16x8, xf), c1a[i], 0, 0, 0);
            }
            acc[I][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[((c * 9 + tt) * 2 + s) * 2 + 0]), __builtin_bit_cast(bf16x8, fr), acc[I][0], 0, 0, 0);
            acc[I][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[((c * 9 + tt) * 2 + s) * 2 + 1]), __builtin_bit_cast(bf16x8, fr), acc[I][1], 0, 0, 0);
        });
        // the roll copy's three slots: read in the phase's last slots, written here (nothing else is in flight)
        if (R64WF_CUT != 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cp[0]), "+v"(cp[1]), "+v"(cp[2]));
#pragma unroll
            for (int k = 0; k < 3; ++k)
                asm volatile("ds_write_b128 %0, %1" : : "v"(cp_off[I * 3 + k] + c1.cp_dst), "v"(cp[k]) : "memory");
        }
    };

    // ---- prologue: the first tile's planes (all five rows, plain conv1), the second tile's patch
    Geom cur, nxt, nx2;
    int t = t_begin;
    tile_geom(t, cur);
    dma_patch(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_rows(0);
    __syncthreads();
    conv1_plain(cur.k_img, 0, 0, 0, BLOCKS, acc[0]);
    tile_geom(t + 1 < t_end ? t + 1 : t, nxt);
    dma_patch(nxt);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    convert_rows(1);
    __syncthreads();

#if R64_DIAG
    unsigned long long dq = __builtin_amdgcn_s_memtime(), dt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long d_c0 = dq, d_r0 = __builtin_amdgcn_s_memrealtime();
#define R64F_STAMP(k) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long n_ = __builtin_amdgcn_s_memtime(); dt[k] += n_ - dq; dq = n_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define R64F_STAMP(k) do { } while (0)
#endif
    Pend pend = {a.y, 0};
    for (int it = 0; t < t_end; ++it, ++t) {
        const int set = it & 1;
        if (set != a_set) {
            const unsigned delta = (unsigned)((set - a_set) * 2 * PLANE);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                    for (int s = 0; s < 2; ++s) areg[i][dx][s] += delta;
            a_set = set;
        }
        const bool more = t + 1 < t_end;
        tile_geom(t + 2 < t_end ? t + 2 : t_end - 1, nx2);
        dma_patch(nx2);                                               // the fp32 patch of tile t + 2 lands while this tile computes
        const int vm_mark = vm_issued;
        const int rows_left = a.H - cur.k_img * a.R, nvalid = (rows_left < a.R ? rows_left : a.R) * W;
        char* const ytile = a.y + (((long long)cur.b * a.H + (long long)cur.k_img * a.R) * W + wid * 64) * 128;
#if R64_DIAG
        dt[0] += 1;
#endif
        // ---- rows the next tile shares with this one (its blocks 0, 1 = this pair's blocks 3, 4), or - a new image - computed
        const bool roll = more && nxt.b == cur.b && nxt.k_img == cur.k_img + 1;
        // roll: 2 planes x 2 blocks = 1536 slots of 16 bytes, six per thread, copied by the half-phases (three each); a tile that starts an
        // image computes the two rows instead and the copy moves a block onto itself (harmless: same bytes)
        if (!roll) conv1_plain(nxt.k_img, set ^ 1, set ^ 1, 0, 2, acc[0]);
        C1Tile c1;
        c1.cp_dst = lds_base + (unsigned)(2 * (set ^ 1) * PLANE);
        c1.cp_src = roll ? lds_base + (unsigned)(2 * set * PLANE + 3 * P * R64_ROWB) : c1.cp_dst;
        c1.pbase = c1.cp_dst;
        c1.dump = lds_base + SCR_BASE;
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
            const int h = a.R * nxt.k_img - 1 + (int)(c1_src[gi] >> 27);
            c1.psrc[gi] = (h >= 0 && h < a.H) ? (c1_src[gi] & 0x07ffffffu) + (unsigned)((set ^ 1) * X4_BUF) : lds_base + ZR_BASE;
        }
        R64F_STAMP(1);
        phase(std::integral_constant<int, 0>{}, pend, c1);
        R64F_STAMP(2);
        pend.y = ytile;
        pend.nv = nvalid - wid * 64;
        phase(std::integral_constant<int, 1>{}, pend, c1);
        R64F_STAMP(3);
        pend.y = ytile + 32 * 128;
        pend.nv = nvalid - wid * 64 - 32;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // this wave's plane writes are done
        vm_wait(vm_issued - vm_mark);                                  // its patch rows of tile t + 2 have landed (stores may stay in flight)
        R64F_STAMP(4);
        convert_rows(set);                                            // patch of t + 2 -> the buffer tile t's patch had
        __syncthreads();
        R64F_STAMP(5);
        cur = nxt; nxt = nx2;
    }
    static_for<0, 32>([&](auto vc) { epi_step(std::integral_constant<int, 1>{}, vc, pend); });
#if R64_DIAG
    if (lane == 0 && blockIdx.x * 8 + wid < 4096) {
        float* d = r64_diag + (size_t)(blockIdx.x * 8 + wid) * 12;
#pragma unroll
        for (int q = 0; q < 8; ++q) d[q] = (float)dt[q];
        d[8] = d[9] = d[10] = 0.f;
        d[11] = (float)(__builtin_amdgcn_s_memtime() - d_c0) / (float)(__builtin_amdgcn_s_memrealtime() - d_r0) * 0.1f;
    }
#endif
}

template <int BLOCKS, bool ACT>
static int launch_r64_wide(const Conv64Args& a, hipStream_t stream) {
    constexpr int PLANE = BLOCKS * R64_P * R64_ROWB;
    constexpr size_t lds = 4 * (size_t)PLANE;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv64_wide_kernel<BLOCKS, ACT>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int grid = cus < a.ntiles ? cus : a.ntiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    return launch_status();
}

template <bool POOL, bool SC, int BLOCKS, bool IMG = false>
static int launch_r64(const Conv64Args& a, hipStream_t stream) {
    constexpr int PLANE = BLOCKS * R64_P * R64_ROWB;
    constexpr int SLAB = (POOL ? 8 : 32) * (32 * 2 + 16);
    constexpr size_t lds = 3 * (size_t)PLANE + (SC ? PLANE : 0) + R64_NW * SLAB + 256;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv64_resident_kernel<POOL, SC, BLOCKS, IMG>;
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

// conv1 + conv2 of layer1.0 fused (eval mode, bf16): SUBREG_EUNSUPPORTED for shapes outside the kernel's LDS plan.
bool conv64_fused_first_supported(int B, int H, int W) {
    return W % 4 == 0 && W >= 80 && W + 1 <= R64_PF && W <= 96 && H >= 2 && 256 / W == 3 && (long long)B * H * W < (1LL << 26);
}
int conv64_fused_first(const float* img, const void* w1, const float* shift1, const void* w2, const float* shift2, void* y, int B, int H,
                       int W, int act, hipStream_t stream, int kernel) {
    if (!conv64_fused_first_supported(B, H, W) || ((size_t)img & 15)) return SUBREG_EUNSUPPORTED;
    Conv64FusedArgs a;
    a.img = img; a.w1 = (const char*)w1; a.shift1 = shift1; a.w = (const char*)w2; a.shift = shift2; a.y = (char*)y;
    a.H = H; a.W = W; a.act = act;
    a.R = 3;
    a.tpi = (H + a.R - 1) / a.R;
    a.ntiles = B * a.tpi;
    a.d_w = make_fastdiv(W);
    a.d_tpi = make_fastdiv(a.tpi);
    // the one-wave-per-SIMD form (conv64_wide_fused_kernel): on request (SUBREG_CONV_KERNEL_WIDE in the flags, or SUBREG_L1_WIDE_FUSED=1 for
    // whole runs).  Measured equal to the 8-wave form at batch 700 (478-481 us against 474-479; profiles/r05_l1_wide_fused.txt), so the
    // rule keeps the 8-wave kernel
    static const bool wide_env = [] { const char* e = getenv("SUBREG_L1_WIDE_FUSED"); return e && e[0] == '1'; }();
    if (kernel == 2 || (kernel == 0 && wide_env)) {
        auto kern = act ? conv64_wide_fused_kernel<true> : conv64_wide_fused_kernel<false>;
        static std::atomic<unsigned long long> lds_set_w[2] = {{0}, {0}};
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), R64_WF_LDS, lds_set_w[act ? 1 : 0])) return rc;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        int grid = cus < a.ntiles ? cus : a.ntiles;
        grid = (grid + 7) / 8 * 8;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), R64_WF_LDS, stream, a);
        return launch_status();
    }
    constexpr size_t lds = R64_FUSED_LDS;
    static_assert(lds <= 160 * 1024, "LDS budget");
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv64_fused_first_kernel), lds, lds_set)) return rc;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    int grid = cus < a.ntiles ? cus : a.ntiles;
    grid = (grid + 7) / 8 * 8;
    hipLaunchKernelGGL(conv64_fused_first_kernel, dim3(grid), dim3(R64_NW * 64), lds, stream, a);
    return launch_status();
}

// Returns SUBREG_EUNSUPPORTED when the shape is not this kernel's (the caller then uses the general kernel).
// img != null: the shortcut's input is the fp32 NCHW image (x2 must be null, w2 the packed first-layer 1x1 weights).
int conv64_resident(const void* x, const void* w, void* y, const float* shift, const void* x2, const void* w2, int Cin2, int B,
                    int H, int W, bool pool, int act, hipStream_t stream, const float* img) {
    if (img && (x2 || !w2 || !pool || W > 127)) return SUBREG_EUNSUPPORTED;
    if (!img && ((x2 != nullptr) != (w2 != nullptr) || (x2 && Cin2 != 32))) return SUBREG_EUNSUPPORTED;
    // padded row pitch 96 (and not wastefully narrow images); FastDiv ranges; 32-bit byte offsets inside a patch
    if (W + 1 > R64_P || W < 64 || H < 2 || (long long)B * H * W >= (1LL << 26)) return SUBREG_EUNSUPPORTED;
    Conv64Args a;
    a.x = (const char*)x; a.w = (const char*)w; a.x2 = (const char*)x2; a.w2 = (const char*)w2; a.y = (char*)y; a.shift = shift;
    a.img = img;
    a.H = H; a.W = W; a.act = act;
    a.R = 0; a.Hp = H / 2; a.Wp = W / 2; a.WT = 0; a.nwin = a.Hp * a.Wp;
    a.d_w = make_fastdiv(W);
    a.d_wp = make_fastdiv(a.Wp > 0 ? a.Wp : 1);
    if (!pool) {
        if (x2) return SUBREG_EUNSUPPORTED;                         // (no caller: conv2 has no shortcut, conv3 is pooled)
        a.R = 256 / W;                                              // whole image rows per 256-row tile
        if (a.R < 1 || a.R > 3) return SUBREG_EUNSUPPORTED;         // 5 blocks hold R + 2 <= 5 image rows
        a.tpi = (H + a.R - 1) / a.R;
        a.ntiles = B * a.tpi;
        a.d_tpi = make_fastdiv(a.tpi);
        // one wave per SIMD (conv64_wide_kernel) where its DMA accounting holds: every 16-column piece of a block has pixels
        static const bool wide_on = [] { const char* e = getenv("SUBREG_NO_WIDE64"); return !(e && e[0] == '1'); }();
        if (wide_on && W >= R64_P - 16) return act ? launch_r64_wide<5, true>(a, stream) : launch_r64_wide<5, false>(a, stream);
        return launch_r64<false, false, 5>(a, stream);
    }
    // windows per tile: the largest WT <= 64 for which no tile touches more than two row pairs (6 blocks = 2 pairs + halo)
    int wt = 0;
    for (int cand = 64; cand >= 48 && !wt; --cand) {
        bool ok = true;
        for (int k = 0; k * cand < a.nwin && ok; ++k) {
            const int first = k * cand, last = (first + cand < a.nwin ? first + cand : a.nwin) - 1;
            if (last / a.Wp - first / a.Wp > 1) ok = false;
        }
        if (ok) wt = cand;
    }
    if (!wt) return SUBREG_EUNSUPPORTED;
    a.WT = wt;
    a.tpi = (a.nwin + wt - 1) / wt;
    a.ntiles = B * a.tpi;
    a.d_tpi = make_fastdiv(a.tpi);
    if (img) {
        static const bool old_on = [] { const char* e = getenv("SUBREG_L1_CONV3_OLD"); return e && e[0] == '1'; }();   // A/B switch
        if (old_on) return launch_r64<true, true, 6, true>(a, stream);
        static std::atomic<unsigned long long> lds_set{0};
        if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(conv64_pool_img_kernel), R64_PI_LDS, lds_set)) return rc;
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) {
            int v = 0;
            if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
        }
        int grid = cus < a.ntiles ? cus : a.ntiles;
        grid = (grid + 7) / 8 * 8;
        hipLaunchKernelGGL(conv64_pool_img_kernel, dim3(grid), dim3(R64_NW * 64), R64_PI_LDS, stream, a);
        return launch_status();
    }
    return x2 ? launch_r64<true, true, 6>(a, stream) : launch_r64<true, false, 6>(a, stream);
}

// the shapes for which the pooled conv3 of layer 1 can take its shortcut from the image (what the function above checks, without
// launching): the caller decides BEFORE the forward whether it needs the im2col buffer at all
bool conv64_image_shortcut_supported(int B, int H, int W) {
    if (W + 1 > R64_P || W < 64 || H < 2 || (long long)B * H * W >= (1LL << 26)) return false;
    const int Hp = H / 2, Wp = W / 2, nwin = Hp * Wp;
    for (int cand = 64; cand >= 48; --cand) {
        bool ok = true;
        for (int k = 0; k * cand < nwin && ok; ++k) {
            const int first = k * cand, last = (first + cand < nwin ? first + cand : nwin) - 1;
            if (last / Wp - first / Wp > 1) ok = false;
        }
        if (ok) return true;
    }
    return false;
}

#if R64_DIAG
extern "C" int subreg_r64_diag_read(float* host_out, int n_floats) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(r64_diag), sizeof(float) * n_floats) == hipSuccess ? 0 : -1;
}
#endif

}  // namespace subreg
