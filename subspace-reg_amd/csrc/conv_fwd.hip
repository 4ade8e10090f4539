// Implicit-GEMM convolution forward for gfx950 (MI355X), bf16 and exact-f32 MFMA.
//
// Replaces the cuDNN calls behind models/resnet_language.py:402-405 (conv3x3),
// :146-147 (1x1 shortcut conv) and fuses what follows them in
// BasicBlock.forward (:268-301): eval-mode BatchNorm (:250,253,255; the scale is
// folded into the packed weights, the shift is added in the epilogue), the
// shortcut branch (:286-288; the 1x1 conv+BN or the identity is a SECOND GEMM
// accumulated into the same MFMA tiles), LeakyReLU(0.1) (:251,289) and
// MaxPool2d(2) (:256,290).  In train mode (epoch 1 of every session,
// eval/language_eval.py:211) it writes the raw convolution and per-channel
// partial sums for the batch statistics instead (elementwise.hip finishes it).
//
// Data layout: activations compact NHWC [B*H*W][C]; weights [Cout][tap][Cin]
// (tap = 3*ky+kx), both in the compute type T (bf16 or f32).  One workgroup
// computes TM rows x TN channels.  Per 32-channel chunk of Cin it stages the
// CONTIGUOUS pixel range that the tile touches through all nine taps (the
// "patch": tile rows +- (W+1) pixels) into LDS once and reuses it for the 9 taps;
// a tap is a constant row offset, image borders are handled by pointing the
// lane's LDS address at a zero row.  LDS rows are 32 channels (64 B bf16 / 128 B
// f32), XOR-swizzled per conv_index.h so that ds_read_b128 is conflict-free.
// Staging is LDS-DMA (global_load_lds_dwordx4: lane-linear LDS image, the swizzle
// is applied to the per-lane SOURCE address): the patch is double-buffered one
// chunk ahead, the per-tap weight tile one step ahead, one barrier per step.
//   bf16: v_mfma_f32_32x32x16_bf16, fp32 accumulate      (throughput mode)
//   f32 : v_mfma_f32_32x32x2_f32, bitwise an fmaf chain  (parity mode, 1e-4 gate)
#include <stdlib.h>

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
    const char* x2;      // fused shortcut GEMM: [npix][Cin2] T (centre tap only) or null
    const char* w2;      // [Cout][Cin2] T
    char* y;             // LINEAR [npix][Cout] T ; POOL [B*Hp*Wp][Cout] T
    const float* scale;  // [Cout] or null (scale folded into the weights)
    const float* shift;  // [Cout]
    const char* res;     // [npix][Cout] T residual or null
    float* stats;        // raw: [m_tiles*WAVES_M][Cout][2] partial (sum, sumsq)
    ConvGeom g;
    int Cin, Cin2, Cout;
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

// one 1-KiB LDS-DMA piece: 64 lanes x 16 B from `base + voff` (wave-uniform 64-bit base in SGPRs, per-lane 32-bit byte
// offset), LDS image lane-linear from the wave-uniform LDS byte address `lds_addr`.
// Issued from inline asm ON PURPOSE: for the builtin form hipcc (ROCm 7.2) inserts `s_waitcnt vmcnt(0)` in front of
// the next ds_read of ANY LDS address, which drains the prefetch every step; asm DMAs are invisible to that pass, so
// the counted `s_waitcnt vmcnt(N)` + `s_barrier` at the end of each step are the only (hand-placed) waits on them.
// M0 (the DMA's LDS base) is compiler-reserved: save / set / restore inside the one statement.
__device__ __forceinline__ void dma16(const char* base, unsigned voff, unsigned lds_addr) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, %2\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(base), "s"(lds_addr)
        : "memory");
}

// NI x NJ 32x32 accumulator tiles per wave; WAVES_M x WAVES_N waves per workgroup; TPS taps staged per step
// (one barrier per step); MINW = minimum waves per SIMD the register allocation must allow.
template <typename T, int NI, int NJ, int WAVES_M, int WAVES_N, int TAPS, int TPS, bool POOL, int AROWS, int MINW>
__global__ __launch_bounds__(WAVES_M* WAVES_N * 64, MINW) void conv_fwd_kernel(const ConvArgs a) {
    using K = KT<T>;
    constexpr int NW = WAVES_M * WAVES_N;
    constexpr int TM = WAVES_M * NI * 32, TN = WAVES_N * NJ * 32;
    constexpr int SLOTS = K::SLOTS, ROWB = K::ROWB, ELEM = K::ELEM;
    constexpr int RPP = 1024 / ROWB;                     // rows per 1-KiB DMA piece (16 bf16 / 8 f32)
    constexpr int ABUF = (AROWS + 1) * ROWB;             // one patch buffer + its zero row
    constexpr int BTAP = TN * ROWB;                      // one tap's weight tile
    constexpr int BBUF = TPS * BTAP;
    constexpr int B_BASE = 2 * ABUF;
    constexpr int CENTER = TAPS / 2;
    constexpr int NG = TAPS / TPS;                       // tap groups (= steps) per chunk
    static_assert(AROWS % RPP == 0 && TN % RPP == 0 && TAPS % TPS == 0, "DMA pieces must tile the buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WAVES_N, wave_n = wid % WAVES_N;
    const int lr = lane & 31, lh = lane >> 5;
    const ConvGeom g = a.g;
    // XCD-aware tile order (MI355X: 8 XCDs with private L2s, workgroups dealt round-robin): give each XCD one
    // CONTIGUOUS range of tiles, n-tile fastest, so that the halo rows two neighbouring m-tiles share and the whole
    // patch the n-tiles of one m-tile share are fetched into one L2 once.  Pure placement: bijective for any grid.
    const int ntn = (a.Cout + TN - 1) / TN;
    int vtile;
    {
        const int nwg = gridDim.x, lid = blockIdx.x, q8 = nwg / 8, r8 = nwg % 8, xcd = lid % 8, slot = lid / 8;
        vtile = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int mtile = vtile / ntn;
    const int m0 = mtile * TM, n0 = (vtile % ntn) * TN;

    int plo, phi;
    patch_range<POOL>(g, m0, TM, &plo, &phi);
    const int prow = phi - plo;
    const int apieces = (prow + RPP - 1) / RPP;

    // zero rows (index AROWS of each patch buffer) for taps outside the image / rows beyond M
    if (tid < 2 * (ROWB / 16)) {
        const int b = tid / (ROWB / 16), q = tid % (ROWB / 16);
        *reinterpret_cast<uint4*>(smem + b * ABUF + AROWS * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
    }

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
        baddr[j] = B_BASE + nl * ROWB + 32 * (f >> 1) + 16 * (lh ^ (f & 1));
    }

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // ---- staging (LDS-DMA).  Lane l of a piece writes LDS row 16q + l/SLOTS, physical slot l%SLOTS, so it must
    //      FETCH logical slot (l%SLOTS) ^ swz(row): the swizzle lives on the source address (rule 21).
    const int prl = lane / SLOTS, psl = lane % SLOTS;            // row within a piece, physical slot
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem
    // piece group `grp` of a patch = pieces grp*NW .. grp*NW+NW-1, one per wave (the last group re-loads the last
    // piece on the surplus waves so that every wave issues the same number of DMAs: counted vmcnt waits rely on it)
    auto stage_patch_group = [&](const char* xsrc, int cin, int chunk, int buf, int grp) {
        const unsigned xrow = (unsigned)cin * ELEM;
        const char* base = xsrc + (size_t)plo * xrow + (size_t)chunk * 32 * ELEM;     // wave-uniform
        int q = grp * NW + wid;
        q = q < apieces ? q : apieces - 1;
        const int row = q * RPP + prl;
        const int srow = row < prow ? row : prow - 1;                // tail rows of the last piece: any valid source
        dma16(base, (unsigned)srow * xrow + ((psl ^ swz<SLOTS>(row)) << 4), lds_base + buf * ABUF + q * 1024);
    };
    const int agroups = (apieces + NW - 1) / NW;                     // piece groups of one patch
    auto stage_patch = [&](const char* xsrc, int cin, int chunk, int buf) {
        for (int grp = 0; grp < agroups; ++grp) stage_patch_group(xsrc, cin, chunk, buf, grp);
    };
    // weight tiles of `ntaps` consecutive taps starting at `tap` into weight buffer `buf`.  The per-lane part of the
    // source address (output channel row, swizzled slot) does not depend on the step: precomputed once.
    constexpr int PW = (TN / RPP + NW - 1) / NW;                     // weight pieces per wave per tap
    int wn[PW], wsl[PW];
#pragma unroll
    for (int k = 0; k < PW; ++k) {
        const int q = k * NW + wid, row = q * RPP + prl;
        const int n = n0 + row;
        wn[k] = n < a.Cout ? n : a.Cout - 1;                         // N tail: those output columns are never stored
        wsl[k] = (psl ^ swz<SLOTS>(row)) << 4;
    }
    auto stage_w = [&](const char* wsrc, int cin, int taps, int chunk, int tap, int ntaps, int buf) {
        const unsigned wrow = (unsigned)(taps * cin) * ELEM;
        for (int tt = 0; tt < ntaps; ++tt) {
            const char* base = wsrc + ((size_t)(tap + tt) * cin + (size_t)chunk * 32) * ELEM;   // wave-uniform
#pragma unroll
            for (int k = 0; k < PW; ++k) {
                const int q = k * NW + wid;
                if (q < TN / RPP)
                    dma16(base, (unsigned)wn[k] * wrow + wsl[k], lds_base + B_BASE + buf * BBUF + tt * BTAP + q * 1024);
            }
        }
    };

    // ---- step list: phase 0 = the convolution (nch0 chunks x NG tap groups); phase 1 = the fused shortcut GEMM
    //      (nch1 chunks, centre tap only).  Patch of chunk c lives in buffer c&1, weights of step s in buffer s&1.
    const int nch0 = a.Cin / 32, nch1 = a.x2 ? a.Cin2 / 32 : 0;
    const int nchunks = nch0 + nch1;
    stage_patch(a.x, a.Cin, 0, 0);
    stage_w(a.w, a.Cin, TAPS, 0, 0, TPS, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's DMA landed ...
    __syncthreads();                                                  // ... everyone's did; zero rows visible
    // Patch of chunk c+1 is prefetched during chunk c, PA piece groups per step, issued AFTER the step's weight
    // prefetch: LDS-DMA completes in issue order, so the end-of-step wait `vmcnt(groups issued this step)` covers
    // the L2-resident weights of the next step but leaves the (HBM) patch pieces in flight for one more step.
    constexpr int PA = (AROWS / RPP + NW * NG - 1) / (NW * NG);
    int step = 0;
    for (int c = 0; c < nchunks; ++c) {
        const bool ph1 = c >= nch0;
        const bool more = c + 1 < nchunks;
        const char* nx = (c + 1 < nch0) ? a.x : a.x2;
        const int ncin = (c + 1 < nch0) ? a.Cin : a.Cin2, nck = (c + 1 < nch0) ? c + 1 : c + 1 - nch0;
        const int aoff = (c & 1) * ABUF;
#pragma unroll
        for (int tg = 0; tg < NG; ++tg) {
            if (ph1 && tg != 0) continue;
            const bool last_grp = ph1 || tg == NG - 1;
            // prefetch the NEXT step's weight tiles
            {
                const int nc = last_grp ? c + 1 : c;
                if (nc < nchunks) {
                    if (nc < nch0) stage_w(a.w, a.Cin, TAPS, nc, last_grp ? 0 : (tg + 1) * TPS, TPS, (step + 1) & 1);
                    else stage_w(a.w2, a.Cin2, 1, nc - nch0, 0, 1, (step + 1) & 1);
                }
            }
            // ... then this step's share of the next chunk's patch
            int issued = 0;
            if (more) {
                if (ph1) {
                    for (int grp = 0; grp < agroups; ++grp) stage_patch_group(nx, ncin, nck, (c + 1) & 1, grp);
                } else {
#pragma unroll
                    for (int k = 0; k < PA; ++k) {
                        const int grp = tg * PA + k;
                        if (grp < agroups) { stage_patch_group(nx, ncin, nck, (c + 1) & 1, grp); ++issued; }
                    }
                }
            }
            const int boff = (step & 1) * BBUF;
            // k-steps of this step (TPS taps x KSTEPS).  Where the register budget allows two fragment sets (small
            // wave tiles, which also run at low occupancy), software-pipeline: the LDS reads of k-step kk+1 are issued
            // between the MFMAs of k-step kk.  The 64x160 wave tile (160 accumulator registers) cannot afford it.
            constexpr int NK = TPS * K::KSTEPS;
            constexpr bool SWP = NI * NJ * 16 + 2 * (NI + NJ) * 4 + 40 <= 200;
            auto load_frags = [&](int kk, uint4(&xa)[NI], uint4(&xb)[NJ]) {
                const int tt = kk / K::KSTEPS, s = kk % K::KSTEPS;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const int ad = ph1 ? aaddr[i][CENTER] : aaddr[i][tg * TPS + tt];
                    xa[i] = *reinterpret_cast<const uint4*>(smem + aoff + (ad ^ (32 * s)));
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    xb[j] = *reinterpret_cast<const uint4*>(smem + boff + tt * BTAP + (baddr[j] ^ (32 * s)));
            };
            if constexpr (SWP) {
                constexpr int NRD = NI + NJ, NMM = NI * NJ * (sizeof(T) == 2 ? 1 : 4), PER = NMM / NRD > 0 ? NMM / NRD : 1;
                uint4 fa[2][NI], fb[2][NJ];
                load_frags(0, fa[0], fb[0]);
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    if (ph1 && kk >= K::KSTEPS) continue;                 // the shortcut GEMM has a single tap
                    const bool more_k = kk + 1 < NK && !(ph1 && kk + 1 >= K::KSTEPS);
                    if (more_k) load_frags(kk + 1, fa[(kk + 1) & 1], fb[(kk + 1) & 1]);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) mma_step<T>(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
                    if (kk + 1 < NK) {                                    // interleave: PER MFMAs, 1 ds_read, ...
#pragma unroll
                        for (int n = 0; n < NRD; ++n) {
                            __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    if (ph1 && kk >= K::KSTEPS) continue;
                    uint4 fa[NI], fb[NJ];
                    load_frags(kk, fa, fb);
#pragma unroll
                    for (int i = 0; i < NI; ++i)
#pragma unroll
                        for (int j = 0; j < NJ; ++j) mma_step<T>(fa[i], fb[j], acc[i][j]);
                }
            }
            ++step;
            // end of step: next step's weights landed (and, at a chunk boundary, the whole next patch); this wave's
            // LDS reads are complete (their results fed the MFMAs); then the workgroup barrier
            if (last_grp || issued == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (issued == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (issued == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (issued == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ------------------------------------------------------------------ epilogue
    // C layout of a 32x32 tile: column = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
    T* const y = reinterpret_cast<T*>(a.y);
    const T* const res = reinterpret_cast<const T*>(a.res);
    const bool full = m0 + TM <= g.M && n0 + TN <= a.Cout;            // no ragged edge in this tile
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
                    if (full || (m < g.M && n < a.Cout)) {
                        const float v = acc[i][j][r];
                        y[(unsigned)(m * a.Cout + n)] = ElemTraits<T>::from_float(v);
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
            s1 += __shfl_xor(s1, 32);
            s2 += __shfl_xor(s2, 32);
            if (lh == 0 && n < a.Cout) {
                float* dst = a.stats + ((size_t)(mtile * WAVES_M + wave_m) * a.Cout + n) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
        return;
    }
    constexpr bool SLAB_FITS = NW * 32 * (NJ * 32 * ELEM + 16) <= 2 * ABUF + 2 * BBUF;   // epilogue slabs reuse the staging LDS
    if constexpr (!POOL && SLAB_FITS) if (full && !res) {
        // Full linear tile: stage each 32-row slab of this wave's tile through LDS ([row][channel], +16 B row pad) and
        // write it back as whole 16-byte vectors, consecutive lanes on consecutive addresses of a pixel row.  (The
        // direct path below needs one 2/4-byte store per accumulator register and dominated short-K layers.)
        constexpr int TNW = NJ * 32, RS = TNW * ELEM + 16, VPR = TNW * ELEM / 16, NV = 32 * VPR;
        char* const slab = smem + wid * (32 * RS);               // all waves are past the last step's barrier
        float shj[NJ], scj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + (wave_n * NJ + j) * 32 + lr;
            shj[j] = a.shift[n];
            scj[j] = a.scale ? a.scale[n] : 1.f;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][j][r] * scj[j] + shj[j];
                    if (a.act) v = fmaxf(v, v * 0.1f);            // LeakyReLU(0.1)
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
                    *reinterpret_cast<T*>(slab + row * RS + (j * 32 + lr) * ELEM) = ElemTraits<T>::from_float(v);
                }
            const int mrow0 = m0 + (wave_m * NI + i) * 32;
            char* const ybase = a.y + ((size_t)mrow0 * a.Cout + n0 + wave_n * TNW) * ELEM;
#pragma unroll
            for (int v0 = 0; v0 < NV; v0 += 64) {
                const int v = v0 + lane;
                if (NV % 64 == 0 || v < NV) {
                    const int row = v / VPR, c16 = v % VPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                }
            }
        }
        return;
    }
    if constexpr (POOL && SLAB_FITS) if (full && !res) {
        // Full pooled tile: 2x2 max in registers (4 consecutive accumulator registers = one window), the 8 pooled rows
        // of every 32-row slab staged through LDS and written as whole 16-byte vectors.
        constexpr int TNW = NJ * 32, RS = TNW * ELEM + 16, VPR = TNW * ELEM / 16, NV = 8 * VPR;
        char* const slab = smem + wid * (32 * RS);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + (wave_n * NJ + j) * 32 + lr;
                const float sh = a.shift[n], sc = a.scale ? a.scale[n] : 1.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float best = fmaxf(fmaxf(acc[i][j][4 * q] * sc + sh, acc[i][j][4 * q + 1] * sc + sh),
                                       fmaxf(acc[i][j][4 * q + 2] * sc + sh, acc[i][j][4 * q + 3] * sc + sh));
                    if (a.act) best = fmaxf(best, best * 0.1f);       // monotone => lrelu(max) == max(lrelu)
                    *reinterpret_cast<T*>(slab + (2 * q + lh) * RS + (j * 32 + lr) * ELEM) = ElemTraits<T>::from_float(best);
                }
            }
            const int win0 = (m0 + (wave_m * NI + i) * 32) >> 2;      // first pooled pixel of this slab
            char* const ybase = a.y + ((size_t)win0 * a.Cout + n0 + wave_n * TNW) * ELEM;
#pragma unroll
            for (int v0 = 0; v0 < NV; v0 += 64) {
                const int v = v0 + lane;
                if (NV % 64 == 0 || v < NV) {
                    const int row = v / VPR, c16 = v % VPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + (wave_n * NJ + j) * 32 + lr;
        const bool nv = full || n < a.Cout;
        const float sc = (a.scale && nv) ? a.scale[n] : 1.f, sh = nv ? a.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int mb = m0 + (wave_m * NI + i) * 32 + 4 * lh;
            if (!POOL) {
                const unsigned obase = (unsigned)(mb * a.Cout + n);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dr = (r & 3) + 8 * (r >> 2);
                    if (full || (mb + dr < g.M && nv)) {
                        float v = acc[i][j][r] * sc + sh;
                        if (res) v += ElemTraits<T>::to_float(res[obase + (unsigned)(dr * a.Cout)]);
                        if (a.act) v = fmaxf(v, v * 0.1f);            // LeakyReLU(0.1)
                        y[obase + (unsigned)(dr * a.Cout)] = ElemTraits<T>::from_float(v);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {          // register group q: rows mb + 8q + {0,1,2,3} == one 2x2 window
                    const int m = mb + 8 * q;
                    if (full || (m < g.M && nv)) {
                        float v0 = acc[i][j][4 * q] * sc + sh, v1 = acc[i][j][4 * q + 1] * sc + sh;
                        float v2 = acc[i][j][4 * q + 2] * sc + sh, v3 = acc[i][j][4 * q + 3] * sc + sh;
                        if (res) {
                            const Pix px = row_to_pixel<true>(g, m);  // top-left pixel of the window
                            const T* rp = res + (size_t)px.p * a.Cout + n;
                            v0 += ElemTraits<T>::to_float(rp[0]);
                            v1 += ElemTraits<T>::to_float(rp[a.Cout]);
                            v2 += ElemTraits<T>::to_float(rp[(size_t)g.W * a.Cout]);
                            v3 += ElemTraits<T>::to_float(rp[(size_t)(g.W + 1) * a.Cout]);
                        }
                        float best = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
                        if (a.act) best = fmaxf(best, best * 0.1f);  // monotone => lrelu(max) == max(lrelu)
                        y[(unsigned)((m >> 2) * a.Cout + n)] = ElemTraits<T>::from_float(best);
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- host side
template <bool POOL>
static int worst_patch_rows(const ConvGeom& g, int TM) {
    int worst = 0;
    for (int m0 = 0; m0 < g.M; m0 += TM) {
        int lo, hi;
        patch_range<POOL>(g, m0, TM, &lo, &hi);
        if (hi - lo > worst) worst = hi - lo;
    }
    return worst;
}

template <typename T, int NI, int NJ, int WM, int WN, int TAPS, int TPS, bool POOL, int AROWS, int MINW>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    using K = KT<T>;
    constexpr int TM = WM * NI * 32, TN = WN * NJ * 32;
    const size_t lds = 2 * (size_t)(AROWS + 1) * K::ROWB + 2 * (size_t)TPS * TN * K::ROWB;
    auto kern = conv_fwd_kernel<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AROWS, MINW>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return SUBREG_EHIP;
        attr_done = true;
    }
    dim3 grid(((a.g.M + TM - 1) / TM) * ((a.Cout + TN - 1) / TN));      // 1-D: the kernel decodes (m-tile, n-tile) itself
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, a);
    return launch_status();
}

// AR_S / AR_L: small and large LDS patch capacities (rows); the small one allows more workgroups per CU
template <typename T, int NI, int NJ, int WM, int WN, int TAPS, int TPS, bool POOL, int AR_S, int AR_L, int MINW>
static int launch_rows(const ConvArgs& a, hipStream_t s) {
    if ((long long)a.g.M * a.Cout >= (1LL << 31) || (long long)a.g.npix * a.Cout >= (1LL << 31)) return SUBREG_EUNSUPPORTED;
    const int worst = worst_patch_rows<POOL>(a.g, WM * NI * 32);
    if (worst <= AR_S) return launch_cfg<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AR_S, MINW>(a, s);
    if (worst <= AR_L) return launch_cfg<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AR_L, MINW>(a, s);
    return SUBREG_EUNSUPPORTED;      // image too wide for the LDS patch
}

// TPS3: taps staged per step for the 3x3 case (1x1 convs always stage their single tap)
template <typename T, int NI, int NJ, int WM, int WN, int TPS3, int AR_S, int AR_L, int MINW>
static int launch_shape(const ConvArgs& a, bool pool, hipStream_t s) {
    if (a.g.taps == 9) {
        return pool ? launch_rows<T, NI, NJ, WM, WN, 9, TPS3, true, AR_S, AR_L, MINW>(a, s)
                    : launch_rows<T, NI, NJ, WM, WN, 9, TPS3, false, AR_S, AR_L, MINW>(a, s);
    }
    // 1x1: the patch is exactly the tile's own rows (no halo) => small patch buffers, more workgroups per CU
    constexpr int TM = WM * NI * 32;
    return pool ? launch_rows<T, NI, NJ, WM, WN, 1, 1, true, AR_S, AR_L, MINW>(a, s)
                : launch_rows<T, NI, NJ, WM, WN, 1, 1, false, TM, AR_L, MINW>(a, s);
}

// rows of stats partials the raw mode writes for a given problem (caller sizes the buffer with this)
static int stats_rows_for(int dtype, int M) {
    const int wm = dtype == SUBREG_BF16 ? 4 : 2;
    const int tm = wm * 2 * 32;
    return ((M + tm - 1) / tm) * wm;
}

}  // namespace subreg

using namespace subreg;

extern "C" int subreg_conv_stats_rows(int dtype, int B, int H, int W, int Cout) {
    (void)Cout;
    return stats_rows_for(dtype, B * H * W);
}

extern "C" int subreg_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* shift,
                               const void* residual, float* stats_partial, const void* x2, const void* w2, int Cin2, int B,
                               int H, int W, int Cin, int Cout, int ksize, int flags, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x && w && y);
    SUBREG_CHECK_ARG(B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0);
    SUBREG_CHECK_ARG(ksize == 1 || ksize == 3);
    SUBREG_CHECK_ARG(Cin % 32 == 0 && Cout % 32 == 0);
    SUBREG_CHECK_ARG(dtype == SUBREG_F32 || dtype == SUBREG_BF16);
    const bool raw = flags & SUBREG_CONV_RAW_STATS, pool = flags & SUBREG_CONV_POOL2;
    SUBREG_CHECK_ARG(!(raw && (pool || residual || x2)));
    SUBREG_CHECK_ARG(raw ? stats_partial != nullptr : shift != nullptr);
    SUBREG_CHECK_ARG(!pool || (H >= 2 && W >= 2));
    SUBREG_CHECK_ARG(!x2 || (w2 && Cin2 > 0 && Cin2 % 32 == 0));
    ConvArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.y = (char*)y;
    a.x2 = (const char*)x2; a.w2 = (const char*)w2; a.Cin2 = x2 ? Cin2 : 0;
    a.scale = scale; a.shift = shift; a.res = (const char*)residual; a.stats = stats_partial;
    a.g = make_geom(B, H, W, ksize * ksize, pool);
    a.Cin = Cin; a.Cout = Cout;
    a.act = (flags & SUBREG_CONV_LRELU) ? 1 : 0;
    a.raw = raw ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (Cout % 160 == 0);
    // LDS per block = 2 patch buffers + 2 weight buffers, sized so that >= 2 workgroups fit a CU (160 KiB).
    // Tile height by problem size: the chip has 256 CUs x 2 resident workgroups, so small-M layers (10x10, 5x5
    // feature maps) take 128- or 64-row tiles to fill it.
    if (dtype == SUBREG_BF16) {
        if (!wide) {
            // Cout = 64 (layer 1): 64x64 wave tiles are barrier-bound at one tap per step => stage 3 taps per step;
            // the pooled conv3 takes 128-row tiles so that two patch buffers + 3-tap weight buffers still fit twice per CU
            if (a.g.taps == 1) return launch_shape<__bf16, 2, 2, 4, 1, 1, 432, 560, 2>(a, pool, s);
            if (!pool || raw) return launch_rows<__bf16, 2, 2, 4, 1, 9, 3, false, 432, 560, 2>(a, s);
            return launch_rows<__bf16, 1, 2, 4, 1, 9, 3, true, 416, 560, 2>(a, s);
        }
        const long long nt = Cout / 160;
        if (raw || ((a.g.M + 255) / 256) * nt >= 384) return launch_shape<__bf16, 2, 5, 4, 1, 1, 432, 560, 2>(a, pool, s);
        // small maps (10x10, 5x5): 128-row tiles.  If they all fit one per CU (<= 256 workgroups) stage 3 taps per step
        // (84 KB LDS, covers the LDS-DMA latency at that occupancy); otherwise 1 tap per step and 3 workgroups per CU.
        if (((a.g.M + 127) / 128) * nt > 256) return launch_shape<__bf16, 1, 5, 4, 1, 1, 192, 432, 2>(a, pool, s);
        return launch_shape<__bf16, 1, 5, 4, 1, 3, 192, 432, 2>(a, pool, s);
    }
    return wide ? launch_shape<float, 2, 5, 2, 1, 1, 304, 408, 1>(a, pool, s) : launch_shape<float, 2, 2, 2, 1, 1, 304, 408, 1>(a, pool, s);
}
