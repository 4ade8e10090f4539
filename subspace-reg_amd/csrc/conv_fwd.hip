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
// Data layout: activations compact NHWC [B*H*W][C]; weights [tap][Cin/32][Cout][32]
// (tap = 3*ky+kx; one (tap, 32-channel chunk) tile = Cout contiguous rows), both in the compute type T (bf16 or f32).  One workgroup
// computes TM rows x TN channels.  Per 32-channel chunk of Cin it stages the
// CONTIGUOUS pixel range that the tile touches through all nine taps (the
// "patch": tile rows +- (W+1) pixels) into LDS once and reuses it for the 9 taps;
// a tap is a constant row offset, image borders are handled by pointing the
// lane's LDS address at a zero row.  LDS rows are 32 channels (64 B bf16 / 128 B
// f32), XOR-swizzled per conv_index.h so that ds_read_b128 is conflict-free.
// Staging is LDS-DMA (global_load_lds_dwordx4: lane-linear LDS image, the swizzle
// is applied to the per-lane SOURCE address): the patch is double-buffered one
// chunk ahead, the per-tap weight tile one step ahead, one barrier per step.
//   bf16: v_mfma_f32_32x32x16_bf16 / v_mfma_f32_16x16x32_bf16 by tile shape (mfma_tile below), fp32 accumulate
//   f32 : v_mfma_f32_32x32x2_f32, bitwise an fmaf chain  (parity mode, 1e-4 gate)
#include <stdlib.h>

#include <type_traits>

#include "conv_args.h"
#include "conv_index.h"
#include "subreg_common.h"

// Build-time switches.  DIAG is for measurements only.  Design alternatives that were built, verified and measured slower
// or equal (DMAs issued between MFMA groups, a 3-deep weight ring, a concurrent 256+128-row tail tiling, two m-tiles per
// workgroup, 512-row 8-wave workgroups) are recorded with their numbers in DESIGN.md section 4.1, not kept here.
#ifndef SUBREG_DIAG
#define SUBREG_DIAG 0            // 1 = no in-loop staging, 2 = no LDS reads / MFMAs (both: wrong results, timing only);
                                 // 3 = per-wave s_memtime stamps into `stats` of a non-raw call (tools/diag_conv.py)
#endif
#ifndef SUBREG_BF16_MFMA16
#define SUBREG_BF16_MFMA16 1     // 1: v_mfma_f32_16x16x32_bf16 where mfma_tile() says so, 0: v_mfma_f32_32x32x16_bf16 everywhere
#endif
#ifndef SUBREG_B_RING
#define SUBREG_B_RING 3          // depth of the B-fragment register ring of the big wave tiles (reads run SUBREG_B_RING - 1 column tiles ahead)
#endif
#ifndef SUBREG_EARLY_READS
#define SUBREG_EARLY_READS 1     // 1: a step's first fragment reads are issued before its DMAs (see EARLY in the kernel)
#endif
#ifndef SUBREG_STAGING_ROLES
#define SUBREG_STAGING_ROLES 0   // 1: in 3x3 launches the waves take ONE staging role each (3 weight waves + 1 patch wave of 4): measured
                                 // 1-17 % SLOWER (profiles/r04_ab_staging_roles.txt): a step is as long as the wave with the most DMAs
                                 // takes to issue them, and 10 weight pieces over 3 waves is 4 on one of them at every step
#endif
#ifndef SUBREG_MMA_PRIO
#define SUBREG_MMA_PRIO 0        // > 0: s_setprio of a step's fragment-read + MFMA block (0 again behind it)
#endif
#ifndef SUBREG_WAVES_K2
#define SUBREG_WAVES_K2 1        // 1: grids of <= 256 workgroups (128-row tiles, 3 taps per step) put two waves on every tile
#endif

namespace subreg {

template <typename T> struct KT;
template <> struct KT<__bf16> {
    static constexpr int ELEM = 2, SLOTS = 4, ROWB = 64;
};
template <> struct KT<float> {
    static constexpr int ELEM = 4, SLOTS = 8, ROWB = 128;
};
// MFMA tile edge per configuration.  bf16: 16 (v_mfma_f32_16x16x32_bf16) for the 32x160 wave tiles of the small maps, where
// its finer tiles let the B fragments ring-pipeline (+15..30 % on the 10x10 / 5x5 layers); 32 (v_mfma_f32_32x32x16_bf16)
// for the 64-row wave tiles (+5..10 % at batch 350-500 over the 16x16 shape).  f32: 32 (v_mfma_f32_32x32x2_f32).
template <typename T> constexpr int mfma_tile(int NI, int NJ) {
    return (SUBREG_BF16_MFMA16 && sizeof(T) == 2 && (NI == 1 || SUBREG_BF16_MFMA16 >= 2) && NJ == 5) ? 16 : 32;   // (2: measurement builds)
}


template <int TR> struct AccT;                       // accumulator registers of one TR x TR tile (TR*TR/64 per lane)
template <> struct AccT<32> { typedef f32x16 type; };
template <> struct AccT<16> { typedef f32x4 type; };

template <typename T>
__device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x16& acc);

// 16x16x32: lane l holds k = 8*(l/16) .. +7 of row (column) l%16; D[4*(l/16) + r][l%16] in register r
__device__ __forceinline__ void mma_step16(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}
template <>
__device__ __forceinline__ void mma_step<__bf16>(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc,
                                                  0, 0, 0);
}
template <typename T>
__device__ __forceinline__ void mma_step(const uint4& a, const uint4& b, f32x4& acc) {
    mma_step16(a, b, acc);
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

// NI x NJ 32x32 accumulator blocks per wave (held as TR x TR MFMA tiles); WAVES_M x WAVES_N waves per workgroup; TPS taps staged per step
// (one barrier per step); MINW = minimum waves per SIMD the register allocation must allow.
// WK = 2: every output tile is owned by TWO waves that take alternate k-steps of each step and add their accumulators
// through LDS at the end - twice the waves per CU for grids that cannot fill the chip (one 4-wave workgroup per CU leaves
// one wave per SIMD and nothing to hide its DMA issue, waits and barriers behind).
// SPLITK: grid.y workgroups share one output tile, each takes a contiguous range of the 32-channel chunks of Cin and writes its
// fp32 accumulators to a.part; splitk_reduce_kernel adds them up and runs the epilogue (raw + statistics, or scale / shift / act).
// For the 10x10 / 5x5 maps at the pretraining batch: 52-100 tiles on 256 CUs, every tile a chain of 30-60 steps that cannot be
// shorter than one L2 / MALL round trip each (the next step's weights are staged one step ahead).
// SWAPC (eval mode, un-pooled, bf16): the MFMA operands are swapped (A = weights, B = activations), so a lane holds ONE pixel and
// its accumulator registers hold consecutive CHANNELS: the epilogue packs four channels per ds_write_b64 into the store slab
// (the un-swapped form needs one ds_write_b16 and six VALU instructions per element).  Not for raw mode: its per-channel
// statistics want the channel in the lane.
//
// Staging roles (round 4).  In a 3x3 launch (NG > 1 steps per chunk) the waves of a workgroup have ONE staging role each: the
// first NWW waves stage weight tiles (the next step's, waited for at the end of every step), the last NPW waves stage the
// next chunk's patch (waited for only at the chunk's last step).  vmcnt counts a wave's DMAs in issue order, so a wave that
// issued both kinds could not wait for its weights without also waiting for the patch pieces it had issued a step earlier -
// an HBM / MALL round trip exposed every step.  The main loop is two loops (convolution chunks, shortcut chunks) with no
// per-piece phase decisions left in them.
template <typename T, int NI, int NJ, int WAVES_M, int WAVES_N, int TAPS, int TPS, bool POOL, int AROWS, int MINW, int WK = 1, bool SPLITK = false,
          bool SWAPC = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N* WK * 64, MINW) void conv_fwd_kernel(const ConvArgs a) {
    using K = KT<T>;
    constexpr int NWMN = WAVES_M * WAVES_N, NW = NWMN * WK;
    static_assert(WK == 1 || WK == 2, "one or two waves per output tile");
    static_assert(!SWAPC || (!POOL && !SPLITK && sizeof(T) == 2), "swapped accumulators: eval-mode un-pooled bf16 tiles only");
    constexpr int TM = WAVES_M * NI * 32, TN = WAVES_N * NJ * 32;
    constexpr int SLOTS = K::SLOTS, ROWB = K::ROWB, ELEM = K::ELEM;
    constexpr int RPP = 1024 / ROWB;                     // rows per 1-KiB DMA piece (16 bf16 / 8 f32)
    constexpr int ABUF = (AROWS + 1) * ROWB;             // one patch buffer + its zero row
    constexpr int BTAP = TN * ROWB;                      // one tap's weight tile
    constexpr int BBUF = TPS * BTAP;
    constexpr int B_BASE = 2 * ABUF;
    constexpr int NWB = 2;                               // weight buffers: a step's weights are staged one step ahead
    constexpr int CENTER = TAPS / 2;
    constexpr int TR = mfma_tile<T>(NI, NJ), LG = 64 / TR, NR = TR * TR / 64;   // MFMA tile edge, lane groups per tile, regs per tile
    constexpr int KSTEPS = SLOTS / LG;                   // MFMA k-steps per 32-channel chunk
    constexpr int MI = NI * 32 / TR, MJ = NJ * 32 / TR;          // MFMA tiles per wave
    typedef typename AccT<TR>::type acc_t;
    constexpr int NG = TAPS / TPS;                       // tap groups (= steps) per chunk
    static_assert(AROWS % RPP == 0 && TN % RPP == 0 && TAPS % TPS == 0, "DMA pieces must tile the buffers");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_k = wid / NWMN, wmn = wid % NWMN;
    const int wave_m = wmn / WAVES_N, wave_n = wmn % WAVES_N;
    const int lr = lane % TR, lh = lane / TR;            // column (A: row) within an MFMA tile, k-slot / row group
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
    const int m0 = mtile * TM;
    const int n0 = (vtile % ntn) * TN;

    int plo, phi;
    patch_range<POOL>(g, m0, TM, &plo, &phi);
    const int prow = phi - plo;
    const int apieces = (prow + RPP - 1) / RPP;

    // this tile's BN shift / scale into LDS now: fetched at the start of the epilogue they cost a full global-load
    // latency (several thousand cycles with the memory pipe busy) on every workgroup's critical path
    float* const s_shift = reinterpret_cast<float*>(smem + 2 * ABUF + NWB * BBUF);
    float* const s_scale = s_shift + TN;
    for (int t = tid; t < TN; t += NW * 64) {
        const int n = n0 + t < a.Cout ? n0 + t : a.Cout - 1;
        s_shift[t] = (a.raw || !a.shift) ? 0.f : a.shift[n];
        s_scale[t] = (a.raw || !a.scale) ? 1.f : a.scale[n];
    }
    // zero rows (index AROWS of each patch buffer) for taps outside the image / rows beyond M
    if (tid < 2 * (ROWB / 16)) {
        const int b = tid / (ROWB / 16), q = tid % (ROWB / 16);
        *reinterpret_cast<uint4*>(smem + b * ABUF + AROWS * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
    }

    // Accumulators start at zero - or, with swapped accumulators and the BatchNorm scale folded into the weights (every eval-mode
    // call of the backbone), at the BN SHIFT of the register's channel: the loads are issued behind the prologue's DMAs, land under
    // their wait, and the epilogue has no affine step left (no per-channel constants to fetch between its stores).
    const bool shift_in_acc = SWAPC && !a.scale;                     // kernel-uniform
    acc_t acc[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int r = 0; r < NR; ++r) acc[i][j][r] = 0.f;
    // ---- staging (LDS-DMA).  Lane l of a piece writes LDS row RPP*q + l/SLOTS, physical slot l%SLOTS, so it must
    //      FETCH logical slot (l%SLOTS) ^ swz(row): the swizzle lives on the source address (rule 21).
    //      All DMA source addresses are (kernel-argument base pointer) + 32-bit byte offset.
    const int prl = lane / SLOTS, psl = lane % SLOTS;            // row within a piece, physical slot
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;   // LDS byte address of smem
    const unsigned xrow0 = (unsigned)a.Cin * ELEM, xrow1 = (unsigned)a.Cin2 * ELEM;
    const unsigned porg0 = (unsigned)plo * xrow0, porg1 = (unsigned)plo * xrow1;       // byte offset of the patch origin in x / x2
    auto swz_off = [&](int row) -> unsigned { return (unsigned)((psl ^ swz_tr<SLOTS, TR>(row)) << 4); };
    auto rfl = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    // piece q (rows RPP*q ..) of the patch whose 32-channel chunk starts at byte `uoff` of `src` (row pitch xrow) into patch buffer buf;
    // the tail rows of the last piece read any valid source row
    auto patch_piece = [&](const char* src, unsigned uoff, unsigned xrow, int q, int buf) {
        const int row = q * RPP + prl;
        const int srow = row < prow ? row : prow - 1;
        dma16(src, uoff + (unsigned)srow * xrow + swz_off(row), lds_base + buf * ABUF + q * 1024);
    };
    // the same inside the main loop: the per-lane part of the address is one of two registers made once per chunk (pv: every
    // piece but the last; pv_last: the last piece with its tail rows clamped), the rest is scalar.  (Written as patch_piece
    // above, the compiler hoists one clamped row per unrolled piece out of the chunk loop - 24 registers it then spills.)
    static_assert(RPP % 16 == 0 || sizeof(T) == 4, "bf16: the swizzle of a piece row does not depend on the piece");
    auto patch_piece_fast = [&](const char* src, unsigned uoff, unsigned xrow, int q, int buf, unsigned pv, unsigned pv_last) {
        const unsigned so = rfl(uoff + (unsigned)(q * RPP) * xrow);
        if constexpr (MINW >= 3) {
            // three waves per SIMD (168 registers): the two per-lane values are re-made per piece (4 VALU instructions) instead of
            // being held through the chunk - held, the kernel spills five dwords to scratch
            int prl_o = lane / SLOTS;
            asm volatile("" : "+v"(prl_o));
            const int lim = q == apieces - 1 ? prow - 1 - (apieces - 1) * RPP : RPP;
            dma16(src, (unsigned)(prl_o < lim ? prl_o : lim) * xrow + swz_off(prl_o) + so, lds_base + buf * ABUF + q * 1024);
        } else {
            dma16(src, (q == apieces - 1 ? pv_last : pv) + so, lds_base + buf * ABUF + q * 1024);
        }
    };
    // per-lane source offset of weight piece q of a (tap, chunk) tile: output channel row, swizzled slot (N tail: those output
    // columns are never stored)
    auto wvoff_of = [&](int q) -> unsigned {
        const int row = q * RPP + prl;
        const int n = n0 + row;
        const int wn = n < a.Cout ? n : a.Cout - 1;
        return (unsigned)wn * ROWB + swz_off(row);
    };
    // ---- step list: phase 0 = the convolution (nch0 chunks x NG tap groups); phase 1 = the fused shortcut GEMM
    //      (nch1 chunks, centre tap only).  Patch of chunk c lives in buffer c&1, weights of step s in buffer s%NWB.
    const int nch0 = a.Cin / 32, nch1 = a.x2 ? a.Cin2 / 32 : 0;
    const int nchunks = nch0 + nch1;
    // chunk range of this workgroup: everything, or its share of the K split (no shortcut GEMM in split launches)
    const int c_beg = SPLITK ? (int)((long long)nch0 * blockIdx.y / a.ksplit) : 0;
    const int c_end = SPLITK ? (int)((long long)nch0 * (blockIdx.y + 1) / a.ksplit) : nchunks;
    const int c_main_end = c_end < nch0 ? c_end : nch0;               // end of the convolution chunks of this workgroup
    constexpr bool STAMPS = SUBREG_DIAG == 3;
    unsigned long long t_begin = 0, t_loop = 0, t_issue = 0, t_wait = 0, t_bar = 0, t_mma = 0, tq = 0, r_begin = 0;
    if (STAMPS) { t_begin = __builtin_amdgcn_s_memtime(); r_begin = __builtin_amdgcn_s_memrealtime(); }
    const unsigned wtile = (unsigned)a.Cout * ROWB, wtap = (unsigned)nch0 * wtile;     // bytes of one (tap, chunk) tile / of one tap

    // staging roles (see the header): with one step per chunk (1x1 launches) every step is a chunk end and all waves do both
    constexpr bool ROLES = SUBREG_STAGING_ROLES && NG > 1;
    constexpr int NPW = ROLES ? (NW >= 4 ? NW / 4 : 1) : NW;          // waves that stage patch pieces inside the loop
    constexpr int NWW = ROLES ? NW - NPW : NW;                        // waves that stage weight pieces
    const bool has_w = !ROLES || wid < NWW, has_p = !ROLES || wid >= NWW;          // wave-uniform
    // (without roles the in-loop patch pieces are dealt from the LAST wave down: the weight pieces are dealt from the first wave up,
    // TN / RPP = 10 of them over 4 waves is 3, 3, 2, 2, so a partial round of patch pieces lands on the waves with fewer weight DMAs)
    const int ww = has_w ? wid : 0, pw = ROLES ? (has_p ? wid - NWW : 0) : NW - 1 - wid;
    constexpr int WPT = TN / RPP;                                      // weight pieces per tap
    constexpr int PWW = (WPT + NWW - 1) / NWW;                         // ... per weight wave (the last one only on some waves)
    const bool w_last_ok = (PWW - 1) * NWW + ww < WPT;
    unsigned wvoff[PWW];
#pragma unroll
    for (int k = 0; k < PWW; ++k) wvoff[k] = wvoff_of(k * NWW + ww);
    // weight tiles of `ntaps` consecutive taps (byte offset soff of the first, tapstride between them) into weight buffer wb
    auto stage_weights = [&](const char* wsrc, unsigned soff, unsigned tapstride, int ntaps, int wb) {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
            if (tt >= ntaps) continue;
#pragma unroll
            for (int k = 0; k < PWW; ++k) {
                if (k == PWW - 1 && !w_last_ok) continue;
                // (the scalar part goes through readfirstlane as ONE value: split into a per-step and a per-chunk term the
                // compiler hoists wvoff + per-step term out of the chunk loop, one register per unrolled piece)
                dma16(wsrc, wvoff[k] + rfl(soff + (unsigned)tt * tapstride),
                      lds_base + B_BASE + wb * BBUF + tt * BTAP + (k * NWW + ww) * 1024);
            }
        }
    };
    // prologue: the first chunk's patch and the first step's weights, spread over all waves (every wave issues the same number of
    // patch DMAs: the surplus waves of the last round re-load the last piece)
    {
        const int rounds = (apieces + NW - 1) / NW;
        for (int r = 0; r < rounds; ++r) {
            int q = r * NW + wid;
            q = q < apieces ? q : apieces - 1;
            patch_piece(a.x, porg0 + (unsigned)c_beg * (32 * ELEM), xrow0, q, c_beg & 1);
        }
#pragma unroll
        for (int k = 0; k < (TPS * WPT + NW - 1) / NW; ++k) {
            const int idx = k * NW + wid;
            if (idx < TPS * WPT) {
                const int tt = idx / WPT, q = idx % WPT;
                dma16(a.w, wvoff_of(q) + (unsigned)tt * wtap + (unsigned)c_beg * wtile, lds_base + B_BASE + tt * BTAP + q * 1024);
            }
        }
    }
    // per-lane LDS addresses of this lane's A rows for every tap (k-step 0; k-step s is addr ^ 16*LG*s): logical slot
    // LG*s + lh of the row, physical slot = logical ^ swz.  Kept as 16-bit halves (every patch buffer is < 64 KiB):
    // the 16x16 MFMA shape needs MI = 4 row addresses per tap and the accumulators leave no room for 36 registers.
    static_assert(ABUF < 65536, "packed A addresses are 16-bit");
    constexpr int NAP = (MI * TAPS + 1) / 2;
    unsigned apk[NAP];
    {
        // Branch-free on purpose (round 6): written with && / ?: hipcc turned every entry into a divergent branch (exec save, branch,
        // restore) and this table into ~60 % of a prologue that two or three waves per SIMD have to issue before their first MFMA.
        // Validity per row offset dy and column offset dx once per MFMA row tile (unsigned compares), entries as selects.
#pragma unroll
        for (int k = 0; k < NAP; ++k) apk[k] = 0;
        const unsigned zad = (unsigned)(AROWS * ROWB + 16 * lh);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + (wave_m * MI + i) * TR + lr;
            const unsigned mv = m < g.M ? 1u : 0u;
            const Pix px = row_to_pixel<POOL>(g, mv ? m : 0);
            unsigned okh[3], okw[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                okh[d] = mv & ((unsigned)(px.h + d - 1) < (unsigned)g.H ? 1u : 0u);
                okw[d] = (unsigned)(px.w + d - 1) < (unsigned)g.W ? 1u : 0u;
            }
            const int base = px.p - plo;
#pragma unroll
            for (int t = 0; t < TAPS; ++t) {
                const int dy = TAPS == 9 ? t / 3 - 1 : 0, dx = TAPS == 9 ? t % 3 - 1 : 0;
                const int row = base + dy * g.W + dx;
                const unsigned adv = (unsigned)row * ROWB + 16u * ((unsigned)lh ^ (unsigned)swz_tr<SLOTS, TR>(row));
                const unsigned mask = 0u - (okh[dy + 1] & okw[dx + 1]);
                const unsigned ad = ((adv & mask) | (zad & ~mask)) & 0xffffu;
                apk[(i * TAPS + t) >> 1] |= ad << (16 * ((i * TAPS + t) & 1));
            }
        }
    }
    auto aaddr = [&](int i, int t) -> int {
        const int idx = i * TAPS + t;
        return (idx & 1) ? (int)(apk[idx >> 1] >> 16) : (int)(apk[idx >> 1] & 0xffffu);
    };
    // B rows are the tile's consecutive output channels: tile j sits j*TR rows further (a multiple of the swizzle period)
    const int baddr0 = B_BASE + (wave_n * MJ * TR + lr) * ROWB + 16 * (lh ^ swz_tr<SLOTS, TR>(lr));
    static_assert(TR % 16 == 0, "swizzle period");
    // (computed HERE, between issuing the prologue's DMAs and waiting for them: ~1000 cycles of integer divisions that would
    // otherwise precede the first global access of a short-K workgroup)
    // (after the tap addresses: loaded earlier, 80-160 accumulator registers are live across that register-hungry computation)
    if constexpr (SWAPC) {
        if (shift_in_acc && wave_k == 0) {                           // (WK = 2: the pair's sums are added at the end - one shift only)
#pragma unroll
            for (int j = 0; j < MJ; ++j)
#pragma unroll
                for (int q = 0; q < NR / 4; ++q) {
                    int n = n0 + wave_n * (NJ * 32) + j * TR + 8 * q + 4 * lh;
                    n = n < a.Cout ? n : a.Cout - 4;
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
                    for (int i = 0; i < MI; ++i)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = sh[e];
                }
        }
    }

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's DMA landed ...
    __syncthreads();                                                  // ... everyone's did; zero rows visible
    // one MFMA of this kernel: C[pixel][channel] tiles, or (SWAPC) C[channel][pixel] with the operands swapped
    auto mma_ab = [&](const uint4& xa, const uint4& wb, acc_t& c) {
        if constexpr (SWAPC) mma_step<T>(wb, xa, c); else mma_step<T>(xa, wb, c);
    };
    // ---- the k-steps of one step: TPS taps x KSTEPS (a shortcut step: the centre tap only) on patch buffer offset aoff, weight
    //      buffer offset boff.  Where the register budget allows two fragment sets (small wave tiles, which also run at low
    //      occupancy), software-pipeline: the LDS reads of k-step kk+1 are issued between the MFMAs of k-step kk.  The 64x160
    //      wave tile (160 accumulator registers) cannot afford it: per k-step the A fragments, then the B fragments streamed
    //      column tile by column tile through a 3-deep register ring, each read issued two tiles (2*MI MFMAs) ahead of its use.
    //      The order is pinned with sched_group_barriers: left alone, the scheduler sinks every read next to its use and waits
    //      lgkmcnt(0) on it (one exposed LDS latency per MI MFMAs).
    constexpr int NK = TPS * KSTEPS;
    constexpr int KX = 16 * LG;                               // address XOR per k-step
    constexpr bool SWP = NI * NJ * 16 + 2 * (MI + MJ) * 4 + 40 <= 200;
    static_assert(WK == 1 || !SWP, "the k-step split is implemented on the ring-pipelined path");
    // EARLY: the step's first LDS reads (k-step 0: the A fragments and the first two B fragments) are issued BEFORE the step's DMAs,
    // right behind the barrier that made them readable - their latency then runs under the DMA issue instead of in front of the
    // step's first MFMA.  (Ring-pipelined path, one wave per tile.)
    // Measured (profiles/r04_ab_early_reads.txt): -1...-3 % on the 32x32x16 / 256-row tiles, +1 % on the 16x16x32 / 128-row tiles
    // (three workgroups per CU: another wave's MFMAs already cover that latency) - so only where TR == 32.
    constexpr bool EARLY = SUBREG_EARLY_READS && !SWP && WK == 1 && TR == 32 && SUBREG_DIAG != 2;
    uint4 e_fa[MI], e_fb[2];
    auto early_reads = [&](auto ph1_tag, int aoff, int boff, int tg) {
        constexpr bool PH1 = decltype(ph1_tag)::value;
        const int tap = PH1 ? CENTER : tg * TPS;
#pragma unroll
        for (int i = 0; i < MI; ++i) e_fa[i] = *reinterpret_cast<const uint4*>(smem + aoff + aaddr(i, tap));
        e_fb[0] = *reinterpret_cast<const uint4*>(smem + boff + baddr0);
        if (MJ > 1) e_fb[1] = *reinterpret_cast<const uint4*>(smem + boff + (TR * ROWB) + baddr0);
    };
    auto mma_block = [&](auto ph1_tag, int aoff, int boff, int tg, int step) {
        constexpr bool PH1 = decltype(ph1_tag)::value;
        constexpr int NKE = PH1 ? KSTEPS : NK;                // k-steps of this step
        auto a_tap = [&](int kk) -> int { return PH1 ? CENTER : tg * TPS + kk / KSTEPS; };
        auto load_b1 = [&](int kk, int j) -> uint4 {
            const int tt = PH1 ? 0 : kk / KSTEPS, s = kk % KSTEPS;
            return *reinterpret_cast<const uint4*>(smem + boff + tt * BTAP + j * (TR * ROWB) + (baddr0 ^ (KX * s)));
        };
        if constexpr (SUBREG_DIAG == 2) {
        } else if constexpr (SWP) {
            constexpr int NRD = MI + MJ, NMM = MI * MJ * (sizeof(T) == 2 ? 1 : 4), PER = NMM / NRD > 0 ? NMM / NRD : 1;
            uint4 fa[2][MI], fb[2][MJ];
            auto load_a = [&](int kk, uint4(&xa)[MI]) {
                const int s = kk % KSTEPS;
#pragma unroll
                for (int i = 0; i < MI; ++i) xa[i] = *reinterpret_cast<const uint4*>(smem + aoff + (aaddr(i, a_tap(kk)) ^ (KX * s)));
            };
            load_a(0, fa[0]);
#pragma unroll
            for (int j = 0; j < MJ; ++j) fb[0][j] = load_b1(0, j);
#pragma unroll
            for (int kk = 0; kk < NKE; ++kk) {
                if (kk + 1 < NKE) {
                    load_a(kk + 1, fa[(kk + 1) & 1]);
#pragma unroll
                    for (int j = 0; j < MJ; ++j) fb[(kk + 1) & 1][j] = load_b1(kk + 1, j);
                }
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < MJ; ++j) mma_ab(fa[kk & 1][i], fb[kk & 1][j], acc[i][j]);
                if (kk + 1 < NKE) {                                   // interleave: PER MFMAs, 1 ds_read, ...
#pragma unroll
                    for (int n = 0; n < NRD; ++n) {
                        __builtin_amdgcn_sched_group_barrier(0x008, PER, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                }
            }
        } else {
            constexpr int MPG = MI * (sizeof(T) == 2 ? 1 : 4);    // MFMAs per column tile
#pragma unroll
            for (int kk = 0; kk < NKE; ++kk) {
                if (WK > 1 && ((step * NK + kk) % WK) != wave_k) continue;    // the partner wave's k-step
                const int s = kk % KSTEPS;
                constexpr int RD = SUBREG_B_RING;                 // B-fragment ring: each read is issued RD - 1 column tiles ahead of its use
                uint4 fa[MI], fb[RD];
                auto rd_a = [&](int i) { fa[i] = *reinterpret_cast<const uint4*>(smem + aoff + (aaddr(i, a_tap(kk)) ^ (KX * s))); };
                if (EARLY && kk == 0) {
#pragma unroll
                    for (int i = 0; i < MI; ++i) fa[i] = e_fa[i];
                    fb[0] = e_fb[0];
                    if (MJ > 1) fb[1] = e_fb[1];
#pragma unroll
                    for (int j = 2; j < RD - 1 && j < MJ; ++j) fb[j] = load_b1(kk, j);
                    if (RD > 3) __builtin_amdgcn_sched_group_barrier(0x100, RD - 3, 0);
                } else {
                    rd_a(0);
                    fb[0] = load_b1(kk, 0);
#pragma unroll
                    for (int i = 1; i < MI; ++i) rd_a(i);
#pragma unroll
                    for (int j = 1; j < RD - 1 && j < MJ; ++j) fb[j] = load_b1(kk, j);
                    __builtin_amdgcn_sched_group_barrier(0x100, MI + (MJ < RD - 1 ? MJ : RD - 1), 0);
                }
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    if (j + RD - 1 < MJ) {
                        fb[(j + RD - 1) % RD] = load_b1(kk, j + RD - 1);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
#pragma unroll
                    for (int i = 0; i < MI; ++i) mma_ab(fa[i], fb[j % RD], acc[i][j]);
                    __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
                }
            }
        }
    };
    // Patch of chunk c+1 is prefetched during chunk c, PA pieces per patch wave and step (none in the chunk's last step when
    // there are several, so that the chunk-end wait finds them landed).
    constexpr int PSTEPS = NG > 1 ? NG - 1 : 1;                       // steps of a chunk that carry patch pieces
    constexpr int PA = (AROWS / RPP + NPW * PSTEPS - 1) / (NPW * PSTEPS);
    int step = 0;
    if (STAMPS) t_loop = __builtin_amdgcn_s_memtime();
    int c = c_beg;
    // ---- phase 0: the convolution chunks
    for (; c < c_main_end; ++c) {
        const bool more = c + 1 < c_end;
        const bool nsecond = c + 1 >= nch0;                           // the next chunk belongs to the shortcut GEMM
        const char* const psrc = nsecond ? a.x2 : a.x;
        const unsigned pxrow = nsecond ? xrow1 : xrow0;
        const unsigned puoff = nsecond ? porg1 + (unsigned)(c + 1 - nch0) * (32 * ELEM) : porg0 + (unsigned)(c + 1) * (32 * ELEM);
        const int np = more ? apieces : 0;                            // patch pieces to stage during this chunk
        unsigned pv = 0, pv_last = 0;                                 // per-lane parts of the patch source addresses (patch waves)
        if (has_p && MINW < 3) {
            int prl_o = prl;
            asm volatile("" : "+v"(prl_o));                           // (opaque: keeps the two values out of the loop-invariant set)
            const int lim = prow - 1 - (apieces - 1) * RPP;           // last valid row of the last piece
            if constexpr (sizeof(T) == 2) {
                pv = (unsigned)prl_o * pxrow + swz_off(prl_o);
                pv_last = (unsigned)(prl_o < lim ? prl_o : lim) * pxrow + swz_off(prl_o);
            }
        }
        const int nbuf = (c + 1) & 1;                                 // patch buffer of the next chunk
        const int aoff = (c & 1) * ABUF;
#pragma unroll
        for (int tg = 0; tg < NG; ++tg) {
            if (STAMPS) tq = __builtin_amdgcn_s_memtime();
            const int wb = (step + 1) % NWB;
            int n_patch = 0;
            if constexpr (EARLY) early_reads(std::false_type{}, aoff, (step % NWB) * BBUF, tg);
            if (SUBREG_DIAG != 1) {
                if (has_w) {                                          // the next step's weights
                    if (tg < NG - 1) {
                        stage_weights(a.w, (unsigned)((tg + 1) * TPS) * wtap + (unsigned)c * wtile, wtap, TPS, wb);
                    } else if (c + 1 < c_main_end) {
                        stage_weights(a.w, (unsigned)(c + 1) * wtile, wtap, TPS, wb);
                    } else if (more) {
                        stage_weights(a.w2, (unsigned)(c + 1 - nch0) * wtile, 0, 1, wb);
                    }
                }
                if (has_p && (NG == 1 || tg < PSTEPS)) {              // PA pieces of the next chunk's patch
#pragma unroll
                    for (int k = 0; k < PA; ++k) {
                        const int q = (tg * PA + k) * NPW + pw;
                        if (q < np) {
                            ++n_patch;
                            if constexpr (sizeof(T) == 2) patch_piece_fast(psrc, puoff, pxrow, q, nbuf, pv, pv_last);
                            else patch_piece(psrc, puoff, pxrow, q, nbuf);
                        }
                    }
                }
            }
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_issue += n - tq; tq = n; }
            if (SUBREG_MMA_PRIO) __builtin_amdgcn_s_setprio(SUBREG_MMA_PRIO);
            mma_block(std::false_type{}, aoff, (step % NWB) * BBUF, tg, step);
            if (SUBREG_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
            ++step;
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_mma += n - tq; tq = n; }
            // end of step: the weight waves' DMAs (next step's weights) landed; at a chunk end the patch waves' too; this
            // wave's LDS reads are complete (their results fed the MFMAs); then the workgroup barrier
            if constexpr (ROLES) {
                if (has_w || tg == NG - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                // every wave issued [weights of the next step][n_patch patch pieces]: DMAs complete in issue order, so vmcnt(PA)
                // leaves exactly a full set of patch pieces - the youngest - in flight; a partial set (once per chunk) is waited for
                if (NG > 1 && tg < NG - 1 && n_patch == PA) {
                    static_assert(NG == 1 || PA <= 6, "vmcnt immediates below");
                    if constexpr (PA == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    if constexpr (PA == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                    if constexpr (PA == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    if constexpr (PA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    if constexpr (PA == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    if constexpr (PA == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
            }
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_wait += n - tq; tq = n; }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_bar += n - tq; tq = n; }
        }
    }
    // ---- phase 1: the shortcut GEMM's chunks, one step each (centre tap); every step consumes a whole patch, which all waves stage
    if constexpr (!SPLITK) {
        for (; c < c_end; ++c) {
            const bool more = c + 1 < c_end;
            const int aoff = (c & 1) * ABUF;
            if (STAMPS) tq = __builtin_amdgcn_s_memtime();
            if constexpr (EARLY) early_reads(std::true_type{}, aoff, (step % NWB) * BBUF, 0);
            if (SUBREG_DIAG != 1 && more) {
                if (has_w) stage_weights(a.w2, (unsigned)(c + 1 - nch0) * wtile, 0, 1, (step + 1) % NWB);
                const unsigned puoff = porg1 + (unsigned)(c + 1 - nch0) * (32 * ELEM);
                for (int q = wid; q < apieces; q += NW) patch_piece(a.x2, puoff, xrow1, q, (c + 1) & 1);
            }
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_issue += n - tq; tq = n; }
            if (SUBREG_MMA_PRIO) __builtin_amdgcn_s_setprio(SUBREG_MMA_PRIO);
            mma_block(std::true_type{}, aoff, (step % NWB) * BBUF, 0, step);
            if (SUBREG_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
            ++step;
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_mma += n - tq; tq = n; }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_wait += n - tq; tq = n; }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (STAMPS) { const unsigned long long n = __builtin_amdgcn_s_memtime(); t_bar += n - tq; tq = n; }
        }
    }

    if constexpr (WK > 1) {
        // add the partner wave's accumulators: [tile*NR + r][lane] floats through the (now idle) staging LDS, as many
        // tiles per round as fit; then only the first wave of each pair runs the epilogue
        constexpr int TILES = MI * MJ, AVAIL = (2 * ABUF + NWB * BBUF) / (NWMN * 64 * 4) / NR;
        constexpr int TPR = AVAIL < TILES ? AVAIL : TILES;
        static_assert(TPR >= 1, "no LDS for the accumulator exchange");
        float* const red = reinterpret_cast<float*>(smem) + (size_t)wmn * (TPR * NR * 64);
#pragma unroll
        for (int t0 = 0; t0 < TILES; t0 += TPR) {
            if (wave_k == 1) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < NR; ++r) red[((t - t0) * NR + r) * 64 + lane] = acc[t / MJ][t % MJ][r];
            }
            __syncthreads();
            if (wave_k == 0) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < NR; ++r) acc[t / MJ][t % MJ][r] += red[((t - t0) * NR + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (wave_k != 0) return;
    }
    if constexpr (SPLITK) {
        // fp32 partial sums of this K range: [split][m][n]; lanes of a row group write 64 / 128 contiguous bytes
        float* const part = a.part + (size_t)blockIdx.y * g.M * a.Cout;
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int n = n0 + (wave_n * MJ + j) * TR + lr;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int mb = m0 + (wave_m * MI + i) * TR + 4 * lh;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < g.M && n < a.Cout) part[(size_t)m * a.Cout + n] = acc[i][j][r];
                }
            }
        }
        return;
    }
    struct EpilogueStamp {                  // DIAG=3: cycles from the end of the main loop to the kernel's last instruction
        float* dst;
        unsigned long long t0;
        __device__ ~EpilogueStamp() { if (dst) *dst = (float)(__builtin_amdgcn_s_memtime() - t0); }
    } epi_stamp{nullptr, 0};
    if (STAMPS && a.stats && !a.raw && lane == 0) {
        const unsigned long long n = __builtin_amdgcn_s_memtime(), rn = __builtin_amdgcn_s_memrealtime();
        float* d = a.stats + ((size_t)blockIdx.x * NW + wid) * 8;
        d[0] = (float)(t_loop - t_begin); d[1] = (float)(n - t_loop); d[2] = (float)t_issue; d[3] = (float)t_mma;
        d[4] = (float)t_wait; d[5] = (float)t_bar; d[6] = (float)(rn - r_begin);
        epi_stamp.dst = d + 7;              // (slot 7 held the step count before)
        epi_stamp.t0 = n;
    }
    // ------------------------------------------------------------------ epilogue
    T* const y = reinterpret_cast<T*>(a.y);
    const T* const res = reinterpret_cast<const T*>(a.res);
    const bool full = m0 + TM <= g.M && n0 + TN <= a.Cout;            // no ragged edge in this tile
    constexpr int TPB = 32 / TR;                                      // MFMA tiles per 32-row slab
    constexpr bool SLAB_FITS = NWMN * 32 * (NJ * 32 * ELEM + 16) <= 2 * ABUF + NWB * BBUF;   // slabs reuse the staging LDS
    if constexpr (SWAPC) {
        // Swapped accumulators: lane = pixel (row lr of MFMA tile i), registers 4q..4q+3 of tile (i, j) = the four consecutive
        // channels j*TR + 8q + 4*lh + {0..3}.  Full tiles go through a [row][channel] LDS slab (+16 B row pad) with one
        // ds_write_b64 per four channels and leave as whole 16-byte vectors, consecutive lanes on consecutive addresses of a
        // pixel row; ragged tiles / a residual operand take 8-byte stores straight from the registers.
        constexpr int TNW = NJ * 32, RS = TNW * ELEM + 16, VPR = TNW * ELEM / 16, NV = 32 * VPR, NQ = NR / 4;
        const bool slab_ok = SLAB_FITS && full && !res;
        char* const slab = smem + wmn * (32 * RS);    // all waves are past the last step's barrier
        const float slope = a.act ? 0.1f : 1.f;       // LeakyReLU(0.1) as max(v, slope * v); slope 1 = no activation
        auto lrelu_pack = [&](float (&v)[4]) -> uint2 {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // (fmaxf would add a canonicalising v_max per element: the compiler does not see that a packed fma result is no sNaN)
                const float t = v[e] * slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(v[e]), "v"(t));
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            return __builtin_bit_cast(uint2, o);
        };
        // channels cb..cb+3 of tile registers 4q..4q+3: scale / shift (unless the shift is already in the accumulators: INACC),
        // residual, activation
        // (per-lane bases + compile-time offsets: spelled with the lane term inside the index the compiler keeps one address
        // register per store and spills)
        const float* const shb = s_shift + wave_n * TNW + 4 * lh;
        const float* const scb = s_scale + wave_n * TNW + 4 * lh;
        char* const wbase = slab + lr * RS + 4 * lh * ELEM;
        auto finish4 = [&](auto inacc_tag, const acc_t& cfr, int q, int cc, const uint2* rp) -> uint2 {   // cc = cb - 4*lh (compile-time)
            constexpr bool INACC = decltype(inacc_tag)::value;
            float v[4];
            if constexpr (INACC) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cfr[4 * q + e];
            } else {
                const f32x4 sh = *reinterpret_cast<const f32x4*>(shb + cc);
                const f32x4 sc = *reinterpret_cast<const f32x4*>(scb + cc);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = cfr[4 * q + e] * sc[e] + sh[e];
            }
            if (rp) {
                const uint2 rv = *rp;
                const __bf16* rb = reinterpret_cast<const __bf16*>(&rv);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += (float)rb[e];
            }
            return lrelu_pack(v);
        };
        auto slab_epilogue = [&](auto inacc_tag) {
#pragma unroll
            for (int ib = 0; ib < NI; ++ib) {
#pragma unroll
                for (int ii = 0; ii < TPB; ++ii)
#pragma unroll
                    for (int j = 0; j < MJ; ++j)
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            const int cc = j * TR + 8 * q;                       // channel (less 4*lh) within this wave's TNW columns
                            *reinterpret_cast<uint2*>(wbase + ii * TR * RS + cc * ELEM) = finish4(inacc_tag, acc[ib * TPB + ii][j], q, cc, nullptr);
                        }
                const int mrow0 = m0 + (wave_m * NI + ib) * 32;
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
        };
        auto direct_epilogue = [&](auto inacc_tag) {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + (wave_m * MI + i) * TR + lr;
#pragma unroll
                for (int j = 0; j < MJ; ++j)
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int cc = j * TR + 8 * q;
                        const int n = n0 + wave_n * TNW + cc + 4 * lh;
                        if (m < g.M && n < a.Cout) {
                            const size_t o = (size_t)m * a.Cout + n;
                            *reinterpret_cast<uint2*>(y + o) =
                                finish4(inacc_tag, acc[i][j], q, cc, res ? reinterpret_cast<const uint2*>(res + o) : nullptr);
                        }
                    }
            }
        };
        if (slab_ok) {
            if (shift_in_acc) slab_epilogue(std::true_type{}); else slab_epilogue(std::false_type{});
        } else {
            if (shift_in_acc) direct_epilogue(std::true_type{}); else direct_epilogue(std::false_type{});
        }
        return;
    }
    // C layout of a TR x TR tile: column = lane % TR; register r holds row (r&3) + 8*(r>>2) + 4*(lane / TR)
    // (32x32: 16 registers, 16x16: 4).  Registers 4q..4q+3 are 4 consecutive rows = one 2x2 pooling window.
    const bool raw_slab = !POOL && SLAB_FITS && full;                 // raw tile written by the slab path below
    if (a.raw) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int n = n0 + (wave_n * MJ + j) * TR + lr;
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int mb = m0 + (wave_m * MI + i) * TR + 4 * lh;
#pragma unroll
                for (int r = 0; r < NR; ++r) {
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (full || (m < g.M && n < a.Cout)) {
                        const float v = acc[i][j][r];
                        if (!raw_slab) y[(unsigned)(m * a.Cout + n)] = ElemTraits<T>::from_float(v);
                        s1 += v;
                        s2 += v * v;
                    }
                }
            }
#pragma unroll
            for (int o = TR; o < 64; o <<= 1) {                      // the LG lane groups hold the other rows
                s1 += __shfl_xor(s1, o);
                s2 += __shfl_xor(s2, o);
            }
            if (lh == 0 && n < a.Cout) {
                float* dst = a.stats + ((size_t)(mtile * WAVES_M + wave_m) * a.Cout + n) * 2;
                dst[0] = s1;
                dst[1] = s2;
            }
        }
        if (!raw_slab) return;
    }
    if constexpr (!POOL && SLAB_FITS) if (full && !res) {
        // Full linear tile: stage each 32-row slab of this wave's tile through LDS ([row][channel], +16 B row pad) and
        // write it back as whole 16-byte vectors, consecutive lanes on consecutive addresses of a pixel row.  (The
        // direct path below needs one 2/4-byte store per accumulator register and dominated short-K layers.)
        constexpr int TNW = NJ * 32, RS = TNW * ELEM + 16, VPR = TNW * ELEM / 16, NV = 32 * VPR;
        char* const slab = smem + wmn * (32 * RS);    // all waves are past the last step's barrier
        float shj[MJ], scj[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int nl = (wave_n * MJ + j) * TR + lr;
            shj[j] = s_shift[nl];
            scj[j] = s_scale[nl];
        }
#pragma unroll
        for (int ib = 0; ib < NI; ++ib) {
#pragma unroll
            for (int ii = 0; ii < TPB; ++ii)
#pragma unroll
                for (int j = 0; j < MJ; ++j)
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        float v = acc[ib * TPB + ii][j][r] * scj[j] + shj[j];
                        if (a.act && !a.raw) v = fmaxf(v, v * 0.1f);  // LeakyReLU(0.1)
                        const int row = ii * TR + (r & 3) + 8 * (r >> 2) + 4 * lh;
                        *reinterpret_cast<T*>(slab + row * RS + (j * TR + lr) * ELEM) = ElemTraits<T>::from_float(v);
                    }
            const int mrow0 = m0 + (wave_m * NI + ib) * 32;
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
        char* const slab = smem + wmn * (32 * RS);
#pragma unroll
        for (int ib = 0; ib < NI; ++ib) {
#pragma unroll
            for (int ii = 0; ii < TPB; ++ii)
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    const int nl = (wave_n * MJ + j) * TR + lr;
                    const float sh = s_shift[nl], sc = s_scale[nl];
                    const acc_t& c = acc[ib * TPB + ii][j];
#pragma unroll
                    for (int q = 0; q < NR / 4; ++q) {
                        float best = fmaxf(fmaxf(c[4 * q] * sc + sh, c[4 * q + 1] * sc + sh),
                                           fmaxf(c[4 * q + 2] * sc + sh, c[4 * q + 3] * sc + sh));
                        if (a.act) best = fmaxf(best, best * 0.1f);       // monotone => lrelu(max) == max(lrelu)
                        const int prow = ii * (TR / 4) + 2 * q + lh;     // window (ii*TR + 8q + 4lh) / 4 of the slab
                        *reinterpret_cast<T*>(slab + prow * RS + (j * TR + lr) * ELEM) = ElemTraits<T>::from_float(best);
                    }
                }
            const int win0 = (m0 + (wave_m * NI + ib) * 32) >> 2;     // first pooled pixel of this slab
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
    for (int j = 0; j < MJ; ++j) {
        const int n = n0 + (wave_n * MJ + j) * TR + lr;
        const bool nv = full || n < a.Cout;
        const float sc = s_scale[(wave_n * MJ + j) * TR + lr], sh = s_shift[(wave_n * MJ + j) * TR + lr];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int mb = m0 + (wave_m * MI + i) * TR + 4 * lh;
            if (!POOL) {
                const unsigned obase = (unsigned)(mb * a.Cout + n);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
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
                for (int q = 0; q < NR / 4; ++q) {     // register group q: rows mb + 8q + {0,1,2,3} == one 2x2 window
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

// Second pass of a SPLITK launch: y[m][n] = epilogue(sum over splits of part[s][m][n]).  Block = 32 rows x 32 channels
// (thread = one row, four channels: 16-byte loads, the splits added in a fixed order).  raw: y = the sum rounded to bf16 and
// stats[row_group][n] = (sum, sum of squares) of the UNROUNDED sums over the block's 32 rows - the same partition into
// 32-row groups (m-tile x 4 + wave) the 128-row-tile kernel writes, so subreg_conv_stats_rows and the finalize pass are unchanged.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, int ksplit, int M, int Cout,
                                                            __bf16* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const __bf16* __restrict__ res, int act, int raw,
                                                            float* __restrict__ stats) {
    __shared__ float r1[32][33], r2[32][33];
    const int cq = threadIdx.x & 7, row = threadIdx.x >> 3;
    const int n = blockIdx.x * 32 + cq * 4, m = blockIdx.y * 32 + row;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m < M) {
        const float* p = part + (size_t)m * Cout + n;
        const size_t stride = (size_t)M * Cout;
#pragma unroll 4
        for (int sp = 0; sp < ksplit; ++sp) {
            const float4 t = *reinterpret_cast<const float4*>(p + sp * stride);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
        if (!raw) {
            float r[4] = {0.f, 0.f, 0.f, 0.f};
            if (res) {                                                // the block's identity shortcut (eval mode): + x before the activation
                const uint2 rv = *reinterpret_cast<const uint2*>(res + (size_t)m * Cout + n);
                r[0] = __builtin_bit_cast(float, rv.x << 16); r[1] = __builtin_bit_cast(float, rv.x & 0xffff0000u);
                r[2] = __builtin_bit_cast(float, rv.y << 16); r[3] = __builtin_bit_cast(float, rv.y & 0xffff0000u);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                o[k] = o[k] * (scale ? scale[n + k] : 1.f) + (shift ? shift[n + k] : 0.f) + r[k];
                if (act) o[k] = fmaxf(o[k], o[k] * 0.1f);
            }
        }
        __bf16 ob[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) ob[k] = ElemTraits<__bf16>::from_float(o[k]);
        *reinterpret_cast<uint2*>(y + (size_t)m * Cout + n) = *reinterpret_cast<const uint2*>(ob);
    }
    if (!raw) return;
    const float vv[4] = {v.x, v.y, v.z, v.w};            // zeros for the rows beyond M
#pragma unroll
    for (int k = 0; k < 4; ++k) { r1[row][cq * 4 + k] = vv[k]; r2[row][cq * 4 + k] = vv[k] * vv[k]; }
    __syncthreads();
    if (threadIdx.x < 32) {
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < 32; ++r) { s1 += r1[r][threadIdx.x]; s2 += r2[r][threadIdx.x]; }
        float* dst = stats + ((size_t)blockIdx.y * Cout + blockIdx.x * 32 + threadIdx.x) * 2;
        dst[0] = s1;
        dst[1] = s2;
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

template <typename T, int NI, int NJ, int WM, int WN, int TAPS, int TPS, bool POOL, int AROWS, int MINW, int WK, bool SPLITK, bool SWAPC>
static int launch_kernel(const ConvArgs& a, hipStream_t stream) {
    using K = KT<T>;
    constexpr int TM = WM * NI * 32, TN = WN * NJ * 32;
    constexpr int ABUF = (AROWS + 1) * K::ROWB, BBUF = TPS * TN * K::ROWB;
    const size_t lds = 2 * (size_t)ABUF + 2 * (size_t)BBUF + 2 * TN * sizeof(float);   // patch x2, weights x2, shift/scale
    auto kern = conv_fwd_kernel<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AROWS, MINW, WK, SPLITK, SWAPC>;
    static std::atomic<unsigned long long> lds_set{0};   // per instantiation
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    dim3 grid(((a.g.M + TM - 1) / TM) * ((a.Cout + TN - 1) / TN), SPLITK ? a.ksplit : 1);   // x: the kernel decodes (m-tile, n-tile) itself
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * WK * 64), lds, stream, a);
    return launch_status();
}

// measurement switch: SUBREG_NO_SWAPC=1 in the environment keeps the un-swapped accumulator layout everywhere (A/B runs)
static bool swapc_enabled() {
    static const bool on = [] { const char* e = getenv("SUBREG_NO_SWAPC"); return !(e && e[0] == '1'); }();
    return on;
}

template <typename T, int NI, int NJ, int WM, int WN, int TAPS, int TPS, bool POOL, int AROWS, int MINW, int WK = 1, bool SPLITK = false>
static int launch_cfg(const ConvArgs& a, hipStream_t stream) {
    // eval-mode un-pooled bf16 tiles run with swapped accumulators (channels in the registers: packed epilogue); raw mode
    // (per-channel statistics in the epilogue), pooled tiles (window max over four registers) and K splits keep the pixel-major form
    if constexpr (!POOL && !SPLITK && sizeof(T) == 2) {
        if (!a.raw && swapc_enabled()) return launch_kernel<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AROWS, MINW, WK, SPLITK, true>(a, stream);
    }
    return launch_kernel<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AROWS, MINW, WK, SPLITK, false>(a, stream);
}

// AR_S / AR_L: small and large LDS patch capacities (rows); the small one allows more workgroups per CU
template <typename T, int NI, int NJ, int WM, int WN, int TAPS, int TPS, bool POOL, int AR_S, int AR_L, int MINW, int WK = 1, bool SPLITK = false>
static int launch_rows(const ConvArgs& a, hipStream_t s) {
    if ((long long)a.g.M * a.Cout >= (1LL << 31) || (long long)a.g.npix * a.Cout >= (1LL << 31)) return SUBREG_EUNSUPPORTED;
    // DMA sources are addressed as base pointer + 32-bit byte offset
    if ((long long)a.g.npix * a.Cin * KT<T>::ELEM >= (1LL << 32) || (long long)a.g.npix * a.Cin2 * KT<T>::ELEM >= (1LL << 32) ||
        (long long)a.g.taps * a.Cin * a.Cout * KT<T>::ELEM >= (1LL << 32))
        return SUBREG_EUNSUPPORTED;
    const int worst = worst_patch_rows<POOL>(a.g, WM * NI * 32);
    if (worst <= AR_S) return launch_cfg<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AR_S, MINW, WK, SPLITK>(a, s);
    if (worst <= AR_L) return launch_cfg<T, NI, NJ, WM, WN, TAPS, TPS, POOL, AR_L, MINW, WK, SPLITK>(a, s);
    return SUBREG_EUNSUPPORTED;      // image too wide for the LDS patch
}

// TPS3: taps staged per step for the 3x3 case (1x1 convs always stage their single tap)
template <typename T, int NI, int NJ, int WM, int WN, int TPS3, int AR_S, int AR_L, int MINW, int WK = 1>
static int launch_shape(const ConvArgs& a, bool pool, hipStream_t s) {
    if (a.g.taps == 9) {
        return pool ? launch_rows<T, NI, NJ, WM, WN, 9, TPS3, true, AR_S, AR_L, MINW, WK>(a, s)
                    : launch_rows<T, NI, NJ, WM, WN, 9, TPS3, false, AR_S, AR_L, MINW, WK>(a, s);
    }
    // 1x1: the patch is exactly the tile's own rows (no halo) => small patch buffers, more workgroups per CU
    constexpr int TM = WM * NI * 32;
    return pool ? launch_rows<T, NI, NJ, WM, WN, 1, 1, true, AR_S, AR_L, MINW, WK>(a, s)
                : launch_rows<T, NI, NJ, WM, WN, 1, 1, false, TM, AR_L, MINW, WK>(a, s);
}

// bf16, Cout % 160 == 0: 256-row tiles (2 workgroups per CU: 512 slots) or 128-row tiles (3 per CU: 768 slots).
// Measured over batch 64..1125 (tools/bench_conv.py with either tiling forced): on the 21x21 and smaller maps a round of
// 128-row tiles costs ~0.62 of a round of 256-row tiles, so the cheaper of ceil(T256/512) and 0.62*ceil(T128/768) wins
// (e.g. 548 tiles of 256 rows = 2 rounds lose to 1094 tiles of 128 rows = 2 x 0.62); on 42x42 maps the halo (2 x 43 rows
// per tile) makes the small tile as slow or slower (pooled: 1.2x), so those keep 256 rows whenever they fill the chip.
static bool wide_takes_256_rows(int M, int Cout, int W) {
    // measurement switch: SUBREG_FORCE_TILE=128 / 256 forces one tiling for every Cout % 160 == 0 launch (A/B runs of the rule below)
    static const int forced = [] { const char* e = getenv("SUBREG_FORCE_TILE"); return e ? atoi(e) : 0; }();
    if (forced == 128) return false;
    if (forced == 256) return true;
    const long long nt = Cout / 160, t256 = (long long)((M + 255) / 256) * nt, t128 = (long long)((M + 127) / 128) * nt;
    if (W > 21) return t256 >= 384;
    return (double)((t256 + 511) / 512) <= 0.62 * (double)((t128 + 767) / 768);
}

// rows of stats partials the raw mode writes for a given problem = m-tiles x waves along M of the configuration that
// subreg_conv_fwd picks for it (the caller sizes the buffer and calls subreg_bn_train_finalize with this)
// f32 (parity mode): 256-row tiles of four waves - one wave on EVERY SIMD of the CU (rounds 1-3 ran 128-row tiles of two waves at one
// workgroup per CU: half the SIMDs idle) - wherever the un-pooled 3x3 patch of such a tile fits the larger patch buffer
// (256 + 2 (W + 1) <= 432 rows); raw mode sizes its statistics rows from the same predicate
static bool f32_takes_256_rows(int W) {
    static const bool on = [] { const char* e = getenv("SUBREG_F32_TILE128"); return !(e && e[0] == '1'); }();   // A/B switch
    return on && 256 + 2 * (W + 1) <= 432;
}

static int stats_rows_for(int dtype, int M, int Cout, int W) {
    if (dtype != SUBREG_BF16) return f32_takes_256_rows(W) ? ((M + 255) / 256) * 4 : ((M + 127) / 128) * 2;   // f32: 4 or 2 waves along M
    const int tm = (Cout % 160 == 0 && !wide_takes_256_rows(M, Cout, W)) ? 128 : 256;
    return ((M + tm - 1) / tm) * 4;
}

int conv64_resident(const void* x, const void* w, void* y, const float* shift, const void* x2, const void* w2, int Cin2, int B,
                    int H, int W, bool pool, int act, hipStream_t stream, const float* img = nullptr);      // conv64_resident.hip

// measurement switch: SUBREG_NO_RESIDENT64=1 in the environment sends layer 1 through the general kernel (A/B runs)
static bool resident64_enabled() {
    static const bool on = [] { const char* e = getenv("SUBREG_NO_RESIDENT64"); return !(e && e[0] == '1'); }();
    return on;
}

}  // namespace subreg

using namespace subreg;

extern "C" int subreg_conv_stats_rows(int dtype, int B, int H, int W, int Cout) {
    return stats_rows_for(dtype, B * H * W, Cout, W);
}

namespace subreg {
bool conv64_image_shortcut_supported(int B, int H, int W);         // conv64_resident.hip
bool conv_first_supported(int B, int H, int W);                    // conv_first.hip
}

namespace subreg {
int conv64_fused_first(const float* img, const void* w1, const float* shift1, const void* w2, const float* shift2, void* y, int B, int H,
                       int W, int act, hipStream_t stream, int kernel);   // conv64_resident.hip; kernel: 0 rule, 1 the 8-wave form, 2 one wave per SIMD
}

extern "C" int subreg_conv12_first_fused(const float* x_nchw, const void* w1_packed, const float* shift1, const void* w2_packed,
                                         const float* shift2, void* y, int B, int H, int W, int flags, int dtype, void* stream) {
    SUBREG_CHECK_ARG(x_nchw && w1_packed && shift1 && w2_packed && shift2 && y && B > 0 && H > 0 && W > 0);
    static const bool on = [] { const char* e = getenv("SUBREG_NO_FUSED12"); return !(e && e[0] == '1'); }();   // A/B switch
    if (!on || dtype != SUBREG_BF16 || (flags & (SUBREG_CONV_POOL2 | SUBREG_CONV_RAW_STATS))) return SUBREG_EUNSUPPORTED;
    return conv64_fused_first(x_nchw, w1_packed, shift1, w2_packed, shift2, y, B, H, W, (flags & SUBREG_CONV_LRELU) ? 1 : 0,
                              (hipStream_t)stream, (flags & SUBREG_CONV_KERNEL_WIDE) ? 2 : (flags & SUBREG_CONV_KERNEL_GENERAL) ? 1 : 0);
}

extern "C" int subreg_layer1_direct_supported(int B, int H, int W, int dtype) {
    static const bool on = [] { const char* e = getenv("SUBREG_IM2COL_FIRST"); return !(e && e[0] == '1'); }();   // A/B switch
    return on && dtype == SUBREG_BF16 && resident64_enabled() && conv_first_supported(B, H, W) &&
           conv64_image_shortcut_supported(B, H, W) ? 1 : 0;
}

extern "C" int subreg_conv_fwd_image_shortcut(const void* x, const void* w, void* y, const float* shift, const float* img_nchw,
                                              const void* w2_first, int B, int H, int W, int Cin, int Cout, int flags, int dtype,
                                              void* stream) {
    SUBREG_CHECK_ARG(x && w && y && shift && img_nchw && w2_first && B > 0 && H > 0 && W > 0);
    if (dtype != SUBREG_BF16 || Cin != 64 || Cout != 64 || !(flags & SUBREG_CONV_POOL2) || (flags & SUBREG_CONV_RAW_STATS))
        return SUBREG_EUNSUPPORTED;
    return conv64_resident(x, w, y, shift, nullptr, w2_first, 32, B, H, W, true, (flags & SUBREG_CONV_LRELU) ? 1 : 0,
                           (hipStream_t)stream, img_nchw);
}

// K splits of the SPLITK path for this problem (1 = not used): bf16, 3x3, Cout % 160 == 0, un-pooled, 128-row tiles that
// leave at least half of the 256 CUs without a workgroup; every split keeps >= 2 chunks of 32 input channels
static int splitk_plan(int dtype, int B, int H, int W, int Cin, int Cout, int ksize) {
    static const bool on = [] { const char* e = getenv("SUBREG_NO_SPLITK"); return !(e && e[0] == '1'); }();   // A/B switch
    if (!on || dtype != SUBREG_BF16 || ksize != 3 || Cout % 160 != 0 || Cin % 32 != 0) return 1;
    const long long M = (long long)B * H * W;
    if (M * Cout >= (1LL << 31) || wide_takes_256_rows((int)M, Cout, W)) return 1;
    // SUBREG_SPLITK_MAXBLOCKS / SUBREG_SPLITK_SLOTS: the rule's two numbers (measurements)
    static const int max_blocks = [] { const char* e = getenv("SUBREG_SPLITK_MAXBLOCKS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 128; }();
    static const int slots = [] { const char* e = getenv("SUBREG_SPLITK_SLOTS"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 256; }();
    const long long blocks = ((M + 127) / 128) * (Cout / 160);
    if (blocks > max_blocks) return 1;
    long long ks = slots / blocks;
    if (ks > Cin / 64) ks = Cin / 64;
    if (ks > 8) ks = 8;
    return ks >= 2 ? (int)ks : 1;
}

extern "C" long long subreg_conv_splitk_floats(int B, int H, int W, int Cin, int Cout, int ksize, int dtype) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
    const int ks = splitk_plan(dtype, B, H, W, Cin, Cout, ksize);
    return ks > 1 ? (long long)ks * B * H * W * Cout : 0;
}

static int conv_fwd_impl(const void* x, const void* w, void* y, const float* scale, const float* shift, const void* residual,
                         float* stats_partial, const void* x2, const void* w2, int Cin2, int B, int H, int W, int Cin, int Cout,
                         int ksize, int flags, int dtype, float* ws, long long ws_floats, void* stream);

extern "C" int subreg_conv_fwd(const void* x, const void* w, void* y, const float* scale, const float* shift,
                               const void* residual, float* stats_partial, const void* x2, const void* w2, int Cin2, int B,
                               int H, int W, int Cin, int Cout, int ksize, int flags, int dtype, void* stream) {
    return conv_fwd_impl(x, w, y, scale, shift, residual, stats_partial, x2, w2, Cin2, B, H, W, Cin, Cout, ksize, flags, dtype,
                         nullptr, 0, stream);
}

extern "C" int subreg_conv_fwd_ws(const void* x, const void* w, void* y, const float* scale, const float* shift,
                                  const void* residual, float* stats_partial, const void* x2, const void* w2, int Cin2, int B,
                                  int H, int W, int Cin, int Cout, int ksize, int flags, int dtype, float* workspace,
                                  long long workspace_floats, void* stream) {
    return conv_fwd_impl(x, w, y, scale, shift, residual, stats_partial, x2, w2, Cin2, B, H, W, Cin, Cout, ksize, flags, dtype,
                         workspace, workspace_floats, stream);
}

static int conv_fwd_impl(const void* x, const void* w, void* y, const float* scale, const float* shift, const void* residual,
                         float* stats_partial, const void* x2, const void* w2, int Cin2, int B, int H, int W, int Cin, int Cout,
                         int ksize, int flags, int dtype, float* ws, long long ws_floats, void* stream) {
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
    a.part = nullptr; a.ksplit = 1;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = (Cout % 160 == 0);
    if (ws && !pool && !x2 && !(raw && residual)) {
        // small-M 3x3 layers with a caller-provided workspace: K split over workgroups + reduce pass (see conv_fwd_kernel)
        const int ks = splitk_plan(dtype, B, H, W, Cin, Cout, ksize);
        if (ks > 1 && ws_floats >= (long long)ks * a.g.M * Cout) {
            a.part = ws; a.ksplit = ks;
            const int rc = launch_rows<__bf16, 1, 5, 4, 1, 9, 3, false, 192, 432, 2, 2, true>(a, s);
            if (rc == SUBREG_OK) {
                const int srows = raw ? stats_rows_for(dtype, a.g.M, Cout, W) : (a.g.M + 31) / 32;
                hipLaunchKernelGGL(splitk_reduce_kernel, dim3(Cout / 32, srows), dim3(256), 0, s, ws, ks, a.g.M, Cout, (__bf16*)y,
                                   raw ? nullptr : scale, raw ? nullptr : shift, raw ? nullptr : (const __bf16*)residual, raw ? 0 : a.act,
                                   raw ? 1 : 0, stats_partial);
                return launch_status();
            }
            if (rc != SUBREG_EUNSUPPORTED) return rc;
            a.part = nullptr; a.ksplit = 1;
        }
    }
    // LDS per block = 2 patch buffers + 2 weight buffers, sized so that >= 2 workgroups fit a CU (160 KiB).
    // Tile height by problem size: the chip has 256 CUs x 2 resident workgroups, so small-M layers (10x10, 5x5
    // feature maps) take 128- or 64-row tiles to fill it.
    if (dtype == SUBREG_BF16) {
        // layer 1's 3x3 convs (Cin = Cout = 64, eval mode): persistent kernel with register-resident weights (conv64_resident.hip)
        if (Cin == 64 && Cout == 64 && a.g.taps == 9 && !raw && !residual && !scale && resident64_enabled()) {
            const int rc = conv64_resident(x, w, y, shift, x2, w2, Cin2, B, H, W, pool, a.act, s);
            if (rc != SUBREG_EUNSUPPORTED) return rc;
        }
        // wide layers, eval mode: the one-wave-per-SIMD kernel of conv_wide.hip where its measured rule prefers it
        // (SUBREG_CONV_KERNEL_WIDE / _GENERAL in `flags` force one of the two: parity tests, A/B runs)
        if (wide && !(flags & SUBREG_CONV_KERNEL_GENERAL)) {
            const bool force = flags & SUBREG_CONV_KERNEL_WIDE;
            if (force || conv_wide_preferred(a, pool)) {
                const int dtr = conv_wide_default_tr(pool);
                const int rows = (flags & SUBREG_CONV_KERNEL_WIDE_128) ? 128 : (flags & SUBREG_CONV_KERNEL_WIDE_256) ? 256 : 0;
                const int rc = conv_wide(a, pool, s, (flags & SUBREG_CONV_KERNEL_WIDE_ALT) ? 48 - dtr : dtr, rows);
                if (rc != SUBREG_EUNSUPPORTED || force) return rc;
            }
        } else if (flags & SUBREG_CONV_KERNEL_WIDE) {
            return SUBREG_EUNSUPPORTED;
        }
        if (!wide) {
            // Cout = 64 (layer 1): the unpooled convs stage 3 taps per step (64x64 wave tiles are barrier-bound at one)
            // 1x1 (the K=32 first layer): a streaming GEMM, 128-row tiles keep more workgroups in flight (-6 % vs 256 rows);
            // train mode keeps 256 rows (subreg_conv_stats_rows does not know the kernel size)
            if (a.g.taps == 1 && !raw) return launch_shape<__bf16, 1, 2, 4, 1, 1, 432, 560, 2>(a, pool, s);
            if (a.g.taps == 1) return launch_shape<__bf16, 2, 2, 4, 1, 1, 432, 560, 2>(a, pool, s);
            if (!pool || raw) return launch_rows<__bf16, 2, 2, 4, 1, 9, 3, false, 432, 560, 2>(a, s);
            // pooled conv3 (+ fused K=32 shortcut): 256-row tiles with ONE tap per step measured 16-19 % faster than the
            // 128-row / 3-tap tiling (and than 256-row / 3-tap) at batch 256 and 700
            return launch_rows<__bf16, 2, 2, 4, 1, 9, 1, true, 432, 560, 2>(a, s);
        }
        const long long nt = Cout / 160;
        auto tiles256 = [&](const ConvArgs& b, hipStream_t st) {
            // patches of <= 352 rows (W <= 42 unpooled) keep the LDS footprint at two workgroups per CU with room to spare
            const int rc = launch_shape<__bf16, 2, 5, 4, 1, 1, 352, 432, 2>(b, pool, st);
            return rc != SUBREG_EUNSUPPORTED ? rc : launch_shape<__bf16, 2, 5, 4, 1, 1, 560, 560, 2>(b, pool, st);
        };
        auto tiles128 = [&](const ConvArgs& b, hipStream_t st) {
            // 128-row tiles.  If they all fit one per CU (<= 256 workgroups) stage 3 taps per step (84 KB LDS, covers the
            // LDS-DMA latency at that occupancy); otherwise 1 tap per step and 3 workgroups per CU.
            if (((b.g.M + 127) / 128) * nt > 256) {
                const int rc = launch_shape<__bf16, 1, 5, 4, 1, 1, 192, 224, 3>(b, pool, st);    // 224: 42x42 maps, still 3 per CU
                return rc != SUBREG_EUNSUPPORTED ? rc : launch_shape<__bf16, 1, 5, 4, 1, 1, 432, 432, 2>(b, pool, st);
            }
#if SUBREG_WAVES_K2
            return launch_shape<__bf16, 1, 5, 4, 1, 3, 192, 432, 2, 2>(b, pool, st);      // two waves per tile (8 waves per CU)
#else
            return launch_shape<__bf16, 1, 5, 4, 1, 3, 192, 432, 2>(b, pool, st);
#endif
        };
        if (!wide_takes_256_rows(a.g.M, Cout, W)) return tiles128(a, s);
        return tiles256(a, s);
    }
    if (f32_takes_256_rows(W)) {
        const int rc = wide ? launch_shape<float, 2, 5, 4, 1, 1, 344, 432, 1>(a, pool, s) : launch_shape<float, 2, 2, 4, 1, 1, 344, 432, 1>(a, pool, s);
        if (rc != SUBREG_EUNSUPPORTED || raw) return rc;             // (a pooled tile's patch can be larger: the 128-row tiles below take it)
    }
    return wide ? launch_shape<float, 2, 5, 2, 1, 1, 304, 408, 1>(a, pool, s) : launch_shape<float, 2, 2, 2, 1, 1, 304, 408, 1>(a, pool, s);
}
