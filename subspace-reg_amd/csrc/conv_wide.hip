// Implicit-GEMM 3x3 convolution forward for the WIDE layers (Cout % 160 == 0: layers 2-4 of the ResNet18 of
// models/resnet_language.py:402-405, BasicBlock.forward :268-301), eval mode, bf16: ONE wave per SIMD.
//
// Why a second kernel (round 5).  conv_fwd.hip keeps two or three 4-wave workgroups per CU (64x160 / 32x160 wave tiles, 160 / 80
// accumulator registers) and lets them cover each other's DMA issue, waits and barriers.  Its measured limits:
// the 128-row tiles stage a 10 KB weight tile per 128 x 160 x 32 MACs and run at the chip's L2 -> LDS rate, every operand byte is
// re-read from LDS once per 64 (32) rows, and the chip holds 1.5-1.65 GHz under that LDS + MFMA load.  This kernel spends the
// register file differently: a wave owns the WHOLE 512-register budget of its SIMD (one 4-wave workgroup per CU) and a
// 128 x 160 output tile (320 accumulator registers), so
//   * a staged weight tile serves 512 rows (half the L2 -> LDS bytes per MAC of the 256-row tile, a quarter of the 128-row tile),
//   * an operand byte read from LDS feeds 128 / 160 instead of 64 / 160 (32 / 160) MACs (-36 % / -62 % LDS bytes per MAC),
//   * there is ONE barrier per 40 MFMAs (1280 matrix-pipe cycles) and the wave never waits for anything it issued less than
//     half a step earlier: fragments are double-buffered in registers one k-step (20 MFMAs) ahead, weight tiles arrive by
//     LDS-DMA into a ring of three 1.5 steps ahead, the next chunk's patch is spread over eight steps.
// With one wave per SIMD nothing covers a stall, so every wait sits behind at least 640 cycles of MFMAs by construction.
//
// Step t = (32-channel chunk c, tap): K = 32 = two k-steps of v_mfma_f32_32x32x16_bf16.
//   H1(t): MFMAs of k-step 0 (fragment set 0) | ds_reads of k-step 1 -> set 1
//   MID(t): s_waitcnt vmcnt(0) (the DMAs issued at MID(t-1): weights of step t+1, patch pieces) ; s_barrier ;
//           issue DMAs: weights of step t+2 -> ring slot (t+2) % 3, 1/8 of the next chunk's patch
//   H2(t): MFMAs of k-step 1 (set 1) | ds_reads of step t+1, k-step 0 -> set 0
// WAR: slot (t+2) % 3 held step t-1, whose last reads fed the MFMAs of H2(t-1); every wave is past those when it passes MID(t).
// The patch buffer of chunk c+1 held chunk c-1; it is written from MID(c, tap 0) on, when every wave has left chunk c-1.
// RAW: data issued at MID(t-1) is waited for (vmcnt) by its issuing wave before the barrier of MID(t) and read after it.
//
// Data layout, LDS image, swizzle, patch reuse over the nine taps, window-major rows for the pooled convolutions and the
// shortcut GEMM as a second phase: as conv_fwd.hip (conv_index.h).  Epilogues: un-pooled tiles run with swapped MFMA operands
// (lane = pixel, registers = consecutive channels; BN shift pre-loaded into the accumulators), pooled tiles pixel-major
// (2x2 max over four registers); both leave through a per-wave LDS slab as whole 16-byte vectors.
#include <stdlib.h>

#include <mutex>
#include <type_traits>

#include "conv_args.h"
#include "conv_index.h"
#include "subreg_common.h"

#ifndef SUBREG_W16_AHEAD
#define SUBREG_W16_AHEAD 2       // conv_wide16_kernel: B fragments read 2 or 3 groups of four MFMAs ahead of their use (see `group`); 3 measured
                                 // equal on the un-pooled layers and 7 % slower on the pooled ones (profiles/r06_wide16_ahead.txt)
#endif
#ifndef SUBREG_WIDE_DIAG
#define SUBREG_WIDE_DIAG 0       // 1 = no in-loop staging, 2 = no LDS reads / MFMAs (wrong results: timing only);
                                 // 3 = per-wave s_memtime stamps into a.stats (tools/diag_conv.py --kernel wide);
                                 // 4 / 5 = no in-loop patch / weight staging (timing only)
#endif

namespace subreg {

namespace {

__device__ __forceinline__ void mma32(const uint4& a, const uint4& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

template <int N> struct IC { static constexpr int value = N; };

// wave-uniform selects as instructions (written as C++ selects on a per-wave condition, hipcc turns them into branches around the
// DMAs: ~80 basic blocks per loop body, accumulators shuffled through scratch where they merge)
__device__ __forceinline__ unsigned ssel(int c, unsigned a, unsigned b) {
    unsigned r;
    asm("s_cmp_lg_u32 %1, 0\n\ts_cselect_b32 %0, %2, %3"
        : "=s"(r)
        : "s"(__builtin_amdgcn_readfirstlane(c)), "s"(__builtin_amdgcn_readfirstlane((int)a)), "s"(__builtin_amdgcn_readfirstlane((int)b))
        : "scc");
    return r;
}
__device__ __forceinline__ const char* ssel_ptr(int c, const char* a, const char* b) {
    const unsigned long long ua = (unsigned long long)(size_t)a, ub = (unsigned long long)(size_t)b;
    const unsigned lo = ssel(c, (unsigned)ua, (unsigned)ub), hi = ssel(c, (unsigned)(ua >> 32), (unsigned)(ub >> 32));
    return (const char*)(size_t)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned vsel(unsigned long long mask, unsigned a, unsigned b) {   // mask ? a : b
    unsigned r;
    const unsigned m = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)mask);
    const unsigned long long m2 = ((unsigned long long)m << 32) | m;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m2));
    return r;
}

// slot G of a step's DMA list: staged when G < count (wave-uniform, an SGPR) - the test, M0 and the DMA in ONE asm statement
template <int G>
__device__ __forceinline__ void dma16_slot(int count_sgpr, const char* base_in, unsigned voff, unsigned lds_addr) {
    const unsigned long long bu = (unsigned long long)(size_t)base_in;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bu);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bu >> 32));
    const char* base = (const char*)(size_t)(((unsigned long long)bhi << 32) | blo);
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
#if SUBREG_WIDE_DIAG == 6      // timing only: every slot issues (no test, no branch)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_u) : "memory", "m0");
#elif SUBREG_WIDE_DIAG == 7    // EXEC mask instead of a branch
    asm volatile(
        "s_cmp_gt_i32 %3, %4\n\t"
        "s_cselect_b64 exec, -1, 0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n\t"
        "s_mov_b64 exec, -1"
        :
        : "v"(voff), "s"(base), "s"(lds_u), "s"(count_sgpr), "n"(G)
        : "memory", "m0", "scc");
#else
    asm volatile(
        "s_cmp_le_i32 %3, %4\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n"
        "1:"
        :
        : "v"(voff), "s"(base), "s"(lds_u), "s"(count_sgpr), "n"(G)
        : "memory", "m0", "scc");
#endif
}

// dma16 (subreg_common.h) under a wave-uniform condition, as ONE asm statement: the branch stays inside it, so a step's code remains
// one basic block for the scheduler and the register allocator (with C++ branches around the DMAs every slot became two blocks,
// the loop body ~80, and hipcc shuffled accumulators through scratch where the blocks merged)
__device__ __forceinline__ void dma16_if(int ok, const char* base_in, unsigned voff, unsigned lds_addr) {
    const unsigned long long bu = (unsigned long long)(size_t)base_in;
    const unsigned blo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bu);
    const unsigned bhi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(bu >> 32));
    const char* base = (const char*)(size_t)(((unsigned long long)bhi << 32) | blo);
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_addr);
    const int ok_u = __builtin_amdgcn_readfirstlane(ok);
    asm volatile(
        "s_cmp_eq_u32 %3, 0\n\t"
        "s_cbranch_scc1 1f\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1\n"
        "1:"
        :
        : "v"(voff), "s"(base), "s"(lds_u), "s"(ok_u)
        : "memory", "m0", "scc");
}

}  // namespace

// WM x WN waves of (32 MI) x 160 output tiles; POOL: window-major rows + 2x2 max; AROWS: patch rows per LDS buffer.
// MI = 3 (96 rows, 240 accumulator registers): hipcc (ROCm 7.2) keeps MFMA accumulators in the 256 AGPRs only - with the 320 of a
// 128-row tile it parks 64 of them in VGPRs and copies them in and out around every MFMA (11 k v_accvgpr instructions, 1.8 KB of scratch).
template <int MI, int WM, int WN, bool POOL, int AROWS, int MINW>
__global__ __launch_bounds__(WM* WN * 64, MINW) void conv_wide_kernel(const ConvArgs a) {
    constexpr bool SWAPC = !POOL;
    constexpr int NW = WM * WN, MJ = 5, TM = WM * MI * 32, TN = WN * 160;
    constexpr int ROWB = 64, RPP = 16, SLOTS = 4, ELEM = 2, TAPS = 9, CENTER = 4;
    constexpr int BTAP = TN * ROWB, NWB = 3;             // one step's weight tile; ring of three
    constexpr int ABUF = (AROWS + 1) * ROWB;             // one patch buffer + its zero row
    constexpr int A_BASE = NWB * BTAP;                   // weights first: their fragment reads take immediate offsets
    constexpr int WPT = TN / RPP;                        // weight pieces (1 KiB) per step
    constexpr int WPW = (WPT + NW - 1) / NW;             // ... per wave
    // In-loop staging: ONE instruction stream for every wave, nothing conditional (with one wave per SIMD every instruction issued
    // between two MFMAs is matrix-pipe time, and a conditional DMA slot costs three times an unconditional one: tools/probes/
    // dma_slot.hip).  A step's list is its WPT = 10 weight pieces (the tile of step t+2) and NPP = 6 pieces of the next chunk's patch;
    // wave w has four slots, issued in this order:
    //   S2: weight piece w + 8 (w < 2) or patch piece w (w >= 2) - the one slot whose kind depends on the wave (selected as DATA)
    //   W0, W1: weight pieces w, w + 4
    //   S3: patch piece w (w < 2) or w + 2 (w >= 2); steps 0 .. PLAST only
    // Patch pieces 0 .. apieces - 2 go six per step from step 0 on (a slot beyond them re-stages piece apieces - 2: same bytes, same
    // place); the last piece (its tail rows are clamped to the patch's last row) is every wave's S3 of step PLAST.
    // MID(t) waits vmcnt(1) after a step with an S3: a wave's DMAs complete in issue order, so that leaves exactly its S3 patch
    // piece in flight for a second step (the patch of the 42x42 / 21x21 maps comes from HBM: 395 / 197 MB of activations; with
    // vmcnt(0) every step the wait was 155-170 cycles per 960-cycle step there, 8-20 on the MALL-resident small maps).
    constexpr int NPP = 6;
    constexpr int PLAST = (AROWS / RPP - 1 + NPP - 1) / NPP;   // first step without regular patch pieces: the last piece's step
    static_assert(NW == 4 && WPT == 10, "slot table above");
    static_assert(PLAST <= 6, "S3 exists in steps 0 .. 6; everything has landed at MID(8)");
    static_assert(MI >= 2 && MI <= 4, "read schedule below");
    static_assert(AROWS % RPP == 0 && ABUF < 65536, "patch buffer: whole pieces, 16-bit row addresses");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = __bf16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave_m = wid / WN, wave_n = wid % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const ConvGeom g = a.g;
    constexpr bool STAMPS = SUBREG_WIDE_DIAG == 3;
    constexpr bool ENDS = STAMPS || SUBREG_WIDE_DIAG == 8;   // 8: loop begin / end and the clock only
    unsigned long long t_begin = 0, t_loop = 0, t_wait = 0, t_bar = 0, r_begin = 0;
    if (ENDS) { t_begin = __builtin_amdgcn_s_memtime(); r_begin = __builtin_amdgcn_s_memrealtime(); }
    // XCD-aware tile order: each of the 8 XCDs owns a contiguous tile range, n-tile fastest (conv_fwd.hip)
    const int ntn = a.Cout / TN;
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

    float* const s_shift = reinterpret_cast<float*>(smem + A_BASE + 2 * ABUF);
    if constexpr (POOL) {
        for (int t = tid; t < TN; t += NW * 64) s_shift[t] = a.shift[n0 + t];
    }
    if (tid < 2 * (ROWB / 16)) {                         // zero rows (index AROWS of each patch buffer)
        const int b = tid / (ROWB / 16), q = tid % (ROWB / 16);
        *reinterpret_cast<uint4*>(smem + A_BASE + b * ABUF + AROWS * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
    }

    f32x16 acc[MI][MJ];
    // ---- staging (LDS-DMA): lane l of a piece writes LDS row 16 q + l / 4, physical slot l % 4, so it fetches logical slot
    //      (l % 4) ^ swz(row): the swizzle lives on the source address
    const int prl = lane / SLOTS, psl = lane % SLOTS;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned xrow0 = (unsigned)a.Cin * ELEM, xrow1 = (unsigned)a.Cin2 * ELEM;
    const unsigned porg0 = (unsigned)plo * xrow0, porg1 = (unsigned)plo * xrow1;
    const unsigned swzo = (unsigned)((psl ^ swz<SLOTS>(prl)) << 4);            // (RPP = 16: the swizzle of a piece row does not depend on the piece)
    auto rfl = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    // piece q of the patch whose 32-channel chunk starts at byte uoff of src (row pitch xrow) into patch buffer buf
    auto patch_piece = [&](const char* src, unsigned uoff, unsigned xrow, int q, int buf) {
        int row = q * RPP + prl;
        row = row < prow ? row : prow - 1;                                     // tail rows of the last piece: any valid row
        dma16(src, uoff + (unsigned)row * xrow + swzo, lds_base + A_BASE + buf * ABUF + q * 1024);
    };
    unsigned wvoff[WPW];                                                       // per-lane source offsets of this wave's weight pieces
#pragma unroll
    for (int k = 0; k < WPW; ++k) {
        const int i = wid + NW * k;
        const int row = (i < WPT ? i : WPT - 1) * RPP + prl;
        wvoff[k] = (unsigned)(n0 + row) * ROWB + swzo;
    }
    // in-loop slot table of this wave (see the kernel's head): S2 is a weight piece on waves 0-1, a patch piece on waves 2-3
    const int s2w = __builtin_amdgcn_readfirstlane(wid < 2 ? 1 : 0);
    const unsigned s2m = (unsigned)__builtin_amdgcn_readfirstlane(wid < 2 ? -1 : 0);
    const unsigned long long s2mask = ((unsigned long long)s2m << 32) | s2m;     // (v_cndmask mask of that select)
    const int p2 = wid, p3 = wid < 2 ? wid : wid + 2;                          // patch piece of S2 (waves 2-3) / S3 within a step's six
    // this wave's weight pieces of the tile at byte offset soff of wsrc into ring slot `slot`
    auto stage_weights = [&](const char* wsrc, unsigned soff, int slot) {
#pragma unroll
        for (int k = 0; k < WPW; ++k) {
            const int i = wid + NW * k;
            if (i < WPT) dma16(wsrc, wvoff[k] + rfl(soff), lds_base + slot * BTAP + i * 1024);
        }
    };
    const int nch0 = a.Cin / 32, nch1 = a.x2 ? a.Cin2 / 32 : 0;
    const unsigned wtile = (unsigned)a.Cout * ROWB, wtap = (unsigned)nch0 * wtile;   // bytes of one (tap, chunk) tile / of one tap
    // weight tile of global step index s (phase 0: s = 9 c + tap; phase 1: s = 9 nch0 + d): source and byte offset
    // ---- prologue: chunk 0's patch, the weights of steps 0 and 1
    for (int q = wid; q < apieces; q += NW) patch_piece(a.x, porg0, xrow0, q, 0);
    stage_weights(a.w, 0, 0);
    stage_weights(a.w, wtap, 1);

    // per-lane LDS offsets (within a patch buffer) of this lane's A rows for every tap, k-step 0 (k-step 1: ^ 32), as 16-bit halves
    constexpr int NAP = (MI * TAPS + 1) / 2;
    unsigned apk[NAP];
    {
        // branch-free (see conv_wide16_kernel: with && / ?: every entry became a divergent branch, the prologue 2.5 x the instructions)
#pragma unroll
        for (int k = 0; k < NAP; ++k) apk[k] = 0;
        const unsigned zad = (unsigned)(AROWS * ROWB + 16 * lh);
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int m = m0 + (wave_m * MI + i) * 32 + lr;
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
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const int row = base + dy * g.W + dx;
                const unsigned adv = (unsigned)row * ROWB + 16u * ((unsigned)lh ^ (unsigned)swz<SLOTS>(row));
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
    const int baddr0 = (wave_n * MJ * 32 + lr) * ROWB + 16 * (lh ^ swz<SLOTS>(lr));

    // accumulators: zero, or (swapped operands) the BN shift of the register's channel - the epilogue then has no affine step
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        if constexpr (SWAPC) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = n0 + wave_n * 160 + j * 32 + 8 * q + 4 * lh;
                const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n);
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] = sh[e];
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto mma_ab = [&](const uint4& xa, const uint4& wb, f32x16& c) {
        if constexpr (SWAPC) mma32(wb, xa, c); else mma32(xa, wb, c);
    };
    // fragment reads of one k-step: A rows of tap `tap` in the patch buffer at byte offset aoff, weight tile at byte offset boff
    // (tap -1: the step that follows a chunk's last one - tap 0 of the next chunk, or the centre tap of the first shortcut step)
    int after_last_is_center = 0;                                         // set per chunk (wave-uniform)
    auto rd_a = [&](int i, int aoff, int tap, int ks) -> uint4 {
        // (the select as an instruction: written in C++, hipcc turns it into a run-time index and moves the table to scratch)
        const int ad = tap >= 0 ? aaddr(i, tap)
                                : (int)vsel(after_last_is_center ? ~0ull : 0ull, (unsigned)aaddr(i, CENTER), (unsigned)aaddr(i, 0));
        return *reinterpret_cast<const uint4*>(smem + aoff + (ad ^ (32 * ks)));
    };
    auto rd_b = [&](int j, int boff, int ks) -> uint4 {
        return *reinterpret_cast<const uint4*>(smem + boff + j * (32 * ROWB) + (baddr0 ^ (32 * ks)));
    };
    // One half step: the 20 MFMAs of the fragment set (ca, cb) in five groups of four (column tile g); in front of group g the
    // wave issues DMA slot g (dma(g): a no-op where the step has none) and two of the nine fragment reads of the NEXT k-step into
    // (na, nb) (RD).  The order inside a group is pinned with sched_group_barriers, the groups with sched_barriers.
    auto half = [&](uint4(&ca)[MI], uint4(&cb)[MJ], uint4(&na)[MI], uint4(&nb)[MJ], auto rd_tag, int n_aoff, auto ntap_tag, auto nks_tag,
                    int n_boff, auto&& dma) {
        constexpr bool RD = decltype(rd_tag)::value;
        constexpr int NTAP = decltype(ntap_tag)::value, NKS = decltype(nks_tag)::value;
        auto group = [&](auto g_tag) {
            constexpr int gq = decltype(g_tag)::value;
            if (SUBREG_WIDE_DIAG != 1) dma(g_tag);
            if constexpr (SUBREG_WIDE_DIAG != 2) {
                // the MI + 5 fragment reads of the next k-step in the order b0 a0 .. a(MI-1) b1 .. b4 (the next half's first group
                // needs every a and b0), RPG(g) of them in front of group g
                constexpr int NRD = MI + MJ, R0 = gq * NRD / MJ, R1 = (gq + 1) * NRD / MJ;
                if constexpr (RD) {
#pragma unroll
                    for (int r = R0; r < R1; ++r) {
                        if (r == 0) nb[0] = rd_b(0, n_boff, NKS);
                        else if (r <= MI) na[r - 1] = rd_a(r - 1, n_aoff, NTAP, NKS);
                        else nb[r - MI] = rd_b(r - MI, n_boff, NKS);
                    }
                }
#pragma unroll
                for (int i = 0; i < MI; ++i) mma_ab(ca[i], cb[gq], acc[i][gq]);
                if constexpr (RD && R1 > R0) __builtin_amdgcn_sched_group_barrier(0x100, R1 - R0, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        group(IC<0>{});
        group(IC<1>{});
        group(IC<2>{});
        group(IC<3>{});
        group(IC<4>{});
    };
    auto no_dma = [](auto) {};
    uint4 fa0[MI], fb0[MJ], fa1[MI], fb1[MJ];
    // the first step's k-step 0
#pragma unroll
    for (int i = 0; i < MI; ++i) fa0[i] = rd_a(i, A_BASE, 0, 0);
#pragma unroll
    for (int j = 0; j < MJ; ++j) fb0[j] = rd_b(j, 0, 0);

    // MID of a step: wait for this wave's DMAs of the previous MID, barrier
    auto mid_sync = [&](bool drain_lds, int do_wait = 0) {   // do_wait: DMAs of this wave that may stay in flight (0 or 1)
        unsigned long long q0 = 0, q1 = 0;
        if (STAMPS) q0 = __builtin_amdgcn_s_memtime();
        if (drain_lds) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (do_wait == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STAMPS) q1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (STAMPS) { const unsigned long long q2 = __builtin_amdgcn_s_memtime(); t_wait += q1 - q0; t_bar += q2 - q1; }
    };
    if (ENDS) t_loop = __builtin_amdgcn_s_memtime();

    // ---- phase 0: the convolution chunks, nine steps each (unrolled: tap, ring slot and fragment addresses are compile-time)
    const unsigned wl0 = lds_base + (unsigned)wid * 1024u;                // LDS offset of weight piece `wid` within a ring slot
    for (int c = 0; c < nch0; ++c) {
        const int aoff = A_BASE + (c & 1) * ABUF;
        const int naoff = A_BASE + ((c + 1) & 1) * ABUF;
        const bool last_c = c + 1 >= nch0;
        // the chunk that follows (its patch is staged during this one); after the last one: this chunk again, into the idle buffer
        const bool to_x2 = last_c && nch1 > 0;
        const char* const psrc = to_x2 ? a.x2 : a.x;
        const unsigned pxrow = to_x2 ? xrow1 : xrow0;
        const unsigned puoff = to_x2 ? porg1 : porg0 + (unsigned)(last_c ? c : c + 1) * (32 * ELEM);
        const unsigned plds = lds_base + A_BASE + ((c + 1) & 1) * ABUF;
        after_last_is_center = last_c ? 1 : 0;
        unsigned cw = (unsigned)c * wtile;                               // byte offset of this chunk's weight tiles within a tap
        asm volatile("" : "+s"(cw));                                     // (opaque: per-step sums stay in the loop, not in 30 hoisted SGPRs)
        // per-lane source offset of a patch piece's rows (pv) / of the LAST piece's rows, tail clamped to the patch's last row
        const int lim = prow - 1 - (apieces - 1) * RPP;
        const unsigned pv = __umul24((unsigned)prl, pxrow) + swzo, pv_last = __umul24((unsigned)(prl < lim ? prl : lim), pxrow) + swzo;
        const unsigned gs = (unsigned)RPP * pxrow;                       // source offset step from piece to piece
        const unsigned l2 = vsel(s2mask, wvoff[2], pv);                  // S2's per-lane part
        auto step = [&](auto tap_tag) {
            constexpr int TAP = decltype(tap_tag)::value;
            constexpr int SL = TAP % NWB;
            // H1: k-step 0 on set 0 | reads of k-step 1 -> set 1
            half(fa0, fb0, fa1, fb1, std::true_type{}, aoff, IC<TAP>{}, IC<1>{}, SL * BTAP, no_dma);
            mid_sync(false, (TAP >= 1 && TAP - 1 <= PLAST) ? 1 : 0);     // (the previous step had an S3: it may stay in flight)
            // DMAs of this MID: the weight tile of step t+2 (ring slot (t+2) % 3; past the last step: tile 0 again, into a slot
            // nobody reads), six pieces of the next chunk's patch
            constexpr int WSL = (TAP + 2) % NWB;
            const char* wsrc = a.w;
            unsigned woff = cw;
            if constexpr (TAP + 2 <= 8) {
                woff += (unsigned)(TAP + 2) * wtap;
            } else {
                constexpr int T2 = TAP + 2 - 9;                           // step 0 or 1 of what follows this chunk
                if (!last_c) woff += (unsigned)T2 * wtap + wtile;
                else if (T2 < nch1) { wsrc = a.w2; woff = (unsigned)T2 * wtile; }
                else woff = 0;
            }
            int q0 = TAP * NPP;                                           // (opaque per step, as cw)
            asm volatile("" : "+s"(q0));
            auto dma = [&](auto g_tag) {
                constexpr int gq = decltype(g_tag)::value;
                if constexpr (gq == 0) {                                  // S2
                    int q = q0 + p2;
                    q = q < apieces - 2 ? q : apieces - 2;
                    const unsigned so = ssel(s2w, woff, puoff + (unsigned)q * gs);
                    const unsigned la = ssel(s2w, wl0 + WSL * BTAP + 8 * 1024, plds + (unsigned)q * 1024u);
                    dma16(ssel_ptr(s2w, wsrc, psrc), l2 + so, la);
                } else if constexpr (gq == 1 || gq == 2) {                // W0, W1
                    dma16(wsrc, wvoff[gq - 1] + rfl(woff), wl0 + WSL * BTAP + (gq - 1) * 4096);
                } else if constexpr (gq == 3 && TAP < PLAST) {            // S3: a regular patch piece
                    int q = q0 + p3;
                    q = q < apieces - 2 ? q : apieces - 2;
                    dma16(psrc, pv + rfl(puoff + (unsigned)q * gs), plds + (unsigned)q * 1024u);
                } else if constexpr (gq == 3 && TAP == PLAST) {           // S3: the last patch piece
                    dma16(psrc, pv_last + rfl(puoff + (unsigned)(apieces - 1) * gs), plds + (unsigned)(apieces - 1) * 1024u);
                }
            };
            // H2: k-step 1 on set 1 | reads of the next step's k-step 0 -> set 0.  The step after tap 8 is tap 0 of the next chunk or
            // the first shortcut step (centre tap): ONE code path with the tap's addresses selected (after the very last step the
            // reads fetch stale LDS into registers nobody uses)
            constexpr bool SAME = TAP < 8;                                // the next step belongs to this chunk
            half(fa1, fb1, fa0, fb0, std::true_type{}, SAME ? aoff : naoff, IC<(SAME ? TAP + 1 : -1)>{}, IC<0>{}, ((TAP + 1) % NWB) * BTAP, dma);
        };
        step(IC<0>{});
        step(IC<1>{});
        step(IC<2>{});
        step(IC<3>{});
        step(IC<4>{});
        step(IC<5>{});
        step(IC<6>{});
        step(IC<7>{});
        step(IC<8>{});
    }
    // ---- phase 1: the fused shortcut GEMM's chunks, ONE step each (centre tap).  A step consumes a whole patch, so the next
    //      chunk's patch is staged at MID and waited for at the end of the step (two patch buffers: not pipelined deeper).
    for (int d = 0; d < nch1; ++d) {
        const int s = 9 * nch0 + d;
        const int aoff = A_BASE + ((nch0 + d) & 1) * ABUF;
        const int naoff = A_BASE + ((nch0 + d + 1) & 1) * ABUF;
        const int boff = (s % NWB) * BTAP, nboff = ((s + 1) % NWB) * BTAP;
        const bool more = d + 1 < nch1;
        half(fa0, fb0, fa1, fb1, std::true_type{}, aoff, IC<CENTER>{}, IC<1>{}, boff, no_dma);
        mid_sync(true);                                                   // (the patch buffer written below is the one step s-1 read)
        if (SUBREG_WIDE_DIAG != 1) {
            if (d + 2 < nch1) stage_weights(a.w2, (unsigned)(d + 2) * wtile, (s + 2) % NWB);
            if (more) for (int q = wid; q < apieces; q += NW) patch_piece(a.x2, porg1 + (unsigned)(d + 1) * (32 * ELEM), xrow1, q, (nch0 + d + 1) & 1);
        }
        half(fa1, fb1, fa0, fb0, std::false_type{}, 0, IC<0>{}, IC<0>{}, 0, no_dma);
        if (more) {
            mid_sync(false);
#pragma unroll
            for (int i = 0; i < MI; ++i) fa0[i] = rd_a(i, naoff, CENTER, 0);
#pragma unroll
            for (int j = 0; j < MJ; ++j) fb0[j] = rd_b(j, nboff, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                      // every wave is done with the staging LDS: it becomes the slabs
    struct EpilogueStamp {                  // DIAG = 3: cycles from the end of the main loop to the kernel's last instruction
        float* dst;
        unsigned long long t0;
        __device__ ~EpilogueStamp() { if (dst) *dst = (float)(__builtin_amdgcn_s_memtime() - t0); }
    } epi_stamp{nullptr, 0};
    if (ENDS && a.stats && lane == 0) {
        const unsigned long long n = __builtin_amdgcn_s_memtime(), rn = __builtin_amdgcn_s_memrealtime();
        float* d = a.stats + ((size_t)blockIdx.x * NW + wid) * 8;
        d[0] = (float)(t_loop - t_begin); d[1] = (float)(n - t_loop); d[2] = 0.f; d[3] = (float)(n - t_loop) - (float)t_wait - (float)t_bar;
        d[4] = (float)t_wait; d[5] = (float)t_bar; d[6] = (float)(rn - r_begin);
        epi_stamp.dst = d + 7;
        epi_stamp.t0 = n;
    }

    // ------------------------------------------------------------------ epilogue
    T* const y = reinterpret_cast<T*>(a.y);
    const bool full = m0 + TM <= g.M;
    constexpr int TNW = 160, RS = TNW * ELEM + 16, VPR = TNW * ELEM / 16;
    char* const slab = smem + wid * (32 * RS);
    static_assert(NW * 32 * RS <= A_BASE + 2 * ABUF, "slabs reuse the staging LDS");
    const float slope = a.act ? 0.1f : 1.f;                               // LeakyReLU(0.1) as max(v, slope v)
    if constexpr (SWAPC) {
        // lane = pixel (row lr of block i), registers 4q..4q+3 of tile (i, j) = channels 32 j + 8 q + 4 lh + {0..3}
        auto lrelu_pack = [&](const f32x16& cfr, int q) -> uint2 {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = cfr[4 * q + e] * slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(cfr[4 * q + e]), "v"(t));
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            return __builtin_bit_cast(uint2, o);
        };
        if (full) {
            constexpr int NV = 32 * VPR;
            char* const wbase = slab + lr * RS + 4 * lh * ELEM;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < MJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<uint2*>(wbase + (j * 32 + 8 * q) * ELEM) = lrelu_pack(acc[i][j], q);
                const int mrow0 = m0 + (wave_m * MI + i) * 32;
                char* const ybase = a.y + ((size_t)mrow0 * a.Cout + n0 + wave_n * TNW) * ELEM;
#pragma unroll
                for (int v0 = 0; v0 < NV; v0 += 64) {
                    const int v = v0 + lane;
                    const int row = v / VPR, c16 = v % VPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int m = m0 + (wave_m * MI + i) * 32 + lr;
#pragma unroll
                for (int j = 0; j < MJ; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + wave_n * TNW + j * 32 + 8 * q + 4 * lh;
                        if (m < g.M) *reinterpret_cast<uint2*>(y + (size_t)m * a.Cout + n) = lrelu_pack(acc[i][j], q);
                    }
            }
        }
    } else {
        // pixel-major: column = lane % 32 (channel), register r = row (r & 3) + 8 (r >> 2) + 4 lh; registers 4q..4q+3 = one 2x2 window
        if (full) {
            constexpr int NV = 8 * VPR;
#pragma unroll
            for (int i = 0; i < MI; ++i) {
#pragma unroll
                for (int j = 0; j < MJ; ++j) {
                    const float sh = s_shift[wave_n * TNW + j * 32 + lr];
                    const f32x16& cfr = acc[i][j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float best = fmaxf(fmaxf(cfr[4 * q], cfr[4 * q + 1]), fmaxf(cfr[4 * q + 2], cfr[4 * q + 3])) + sh;
                        best = fmaxf(best, best * slope);                 // monotone: lrelu(max) == max(lrelu)
                        *reinterpret_cast<T*>(slab + (2 * q + lh) * RS + (j * 32 + lr) * ELEM) = (T)best;
                    }
                }
                const int win0 = (m0 + (wave_m * MI + i) * 32) >> 2;      // first pooled pixel of this block
                char* const ybase = a.y + ((size_t)win0 * a.Cout + n0 + wave_n * TNW) * ELEM;
#pragma unroll
                for (int v0 = 0; v0 < NV; v0 += 64) {
                    const int v = v0 + lane;
                    if (v < NV) {
                        const int row = v / VPR, c16 = v % VPR;
                        const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                        *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int n = n0 + wave_n * TNW + j * 32 + lr;
                const float sh = s_shift[wave_n * TNW + j * 32 + lr];
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    const int mb = m0 + (wave_m * MI + i) * 32 + 4 * lh;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int m = mb + 8 * q;
                        if (m < g.M) {
                            const f32x16& cfr = acc[i][j];
                            float best = fmaxf(fmaxf(cfr[4 * q], cfr[4 * q + 1]), fmaxf(cfr[4 * q + 2], cfr[4 * q + 3])) + sh;
                            best = fmaxf(best, best * slope);
                            y[(size_t)(m >> 2) * a.Cout + n] = (T)best;
                        }
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- round 6: the 256-row tiling on v_mfma_f32_16x16x32_bf16
// Why (profiles/r06_probe_mfma_shape.txt; MI355X_MICROARCH.md, DVFS give-back item 7): with inline-asm MFMAs at 98-99 % matrix-pipe duty
// on random bf16 data, same 64 x 160 wave tile, every SIMD busy, the 16x16x32 shape delivers 1.13-1.14 x the FLOP/s of 32x32x16
// (with all 14 fragments re-read from LDS every K = 32: 0.78-0.80 against 0.68-0.70 of 2.5 PFLOP/s) - the chip holds a higher clock
// on it.  This kernel is conv_wide_kernel<MI = 2> (64 x 160 wave tiles, 256 x 160 tiles, two workgroups per CU, the same LDS image
// up to the swizzle, the same staging slots, waits and barrier) with the step re-cut for that shape:
//   a step (32-channel chunk, tap) = ONE k-step = 4 A fragments (16 rows each) x 10 B fragments (16 channels each) = 40 MFMAs,
//   issued B-major (a0..a3 x b_g) except for the first and the last column pair, which run A-major: a fragment a_i is then dead a
//   quarter of a pair before the step ends and the NEXT step's a'_i is read into the same registers (see the step's schedule at
//   `pair0` below).  Register budget (160 accumulators of 256): ONE A set (16), a B ring of four (16) read two groups = 128
//   matrix-pipe cycles ahead.  (A first cut with the A fragments double-buffered a step ahead needed 20 registers more and hipcc
//   spilled four of the tap-address registers INSIDE the loop - scratch reloads count in vmcnt and drained the LDS-DMAs.)
//   MID (vmcnt wait, barrier) sits between groups 4 and 5, the DMA slots S2 W0 W1 S3 in front of groups 5, 6, 7 and the last pair.
// LDS swizzle of this shape: conv_index.h::swz_tr<4, 16> (lane l reads row l % 16 at logical slot l / 16).
namespace {
__device__ __forceinline__ void mma16(const uint4& a, const uint4& b, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}
}  // namespace

// NA = 4: the 256-row tiling described above.  NA = 2 (round 6, second half): the SAME kernel on 128 x 160 tiles (32 x 160 per wave, 80
// accumulator registers, three workgroups per CU where the patch is small: the 10x10 / 5x5 maps) - what the general kernel's 128-row /
// 16x16x32 tiles do, under this file's loop (unconditional DMA slots, reads running ahead through the steps, one barrier per step).
// Its step is B-major all the way (20 MFMAs = 10 groups of two): with 80 accumulators there are registers for the A fragments
// double-buffered a step ahead and a B ring of five read FOUR groups (128 cycles) ahead - 10 groups per step and five slots: b_g always
// sits in slot g % 5, in every step.
template <int NA, bool POOL, int AROWS, int MINW>
__global__ __launch_bounds__(256, MINW) void conv_wide16_kernel(const ConvArgs a) {
    constexpr bool SWAPC = !POOL;
    constexpr int NW = 4, NB = 10, RW = NA * 16, TM = NW * RW, TN = 160, TR = 16;    // RW rows per wave
    static_assert(NA == 4 || NA == 2, "step schedules below");
    constexpr int ROWB = 64, RPP = 16, SLOTS = 4, ELEM = 2, TAPS = 9, CENTER = 4;
    constexpr int BTAP = TN * ROWB, NWB = 3;
    constexpr int ABUF = (AROWS + 1) * ROWB;
    constexpr int A_BASE = NWB * BTAP;
    constexpr int WPT = TN / RPP, WPW = (WPT + NW - 1) / NW;
    constexpr int NPP = 6;
    constexpr int PLAST = (AROWS / RPP - 1 + NPP - 1) / NPP;
    static_assert(WPT == 10 && WPW == 3, "slot table of conv_wide_kernel");
    static_assert(PLAST <= 6, "S3 exists in steps 0 .. 6; everything has landed at MID(8)");
    static_assert(AROWS % RPP == 0 && ABUF < 65536, "patch buffer: whole pieces, 16-bit row addresses");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using T = __bf16;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lh = lane >> 4;             // row (column) within an MFMA tile, 8-channel k-slot (A, B) / row quad (C)
    const ConvGeom g = a.g;
    constexpr bool STAMPS = SUBREG_WIDE_DIAG == 3;           // per-MID stamps (they drain the LDS queue at every MID: this kernel's reads run through it)
    constexpr bool ENDS = STAMPS || SUBREG_WIDE_DIAG == 8;   // 8: loop begin / end and the clock only
    unsigned long long t_begin = 0, t_loop = 0, t_wait = 0, t_bar = 0, r_begin = 0;
    if (ENDS) { t_begin = __builtin_amdgcn_s_memtime(); r_begin = __builtin_amdgcn_s_memrealtime(); }
    const int ntn = a.Cout / TN;
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

    float* const s_shift = reinterpret_cast<float*>(smem + A_BASE + 2 * ABUF);
    if constexpr (POOL) {
        for (int t = tid; t < TN; t += NW * 64) s_shift[t] = a.shift[n0 + t];
    }
    if (tid < 2 * (ROWB / 16)) {
        const int b = tid / (ROWB / 16), q = tid % (ROWB / 16);
        *reinterpret_cast<uint4*>(smem + A_BASE + b * ABUF + AROWS * ROWB + q * 16) = make_uint4(0, 0, 0, 0);
    }

    f32x4 acc[NA][NB];
    const int prl = lane / SLOTS, psl = lane % SLOTS;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
    const unsigned xrow0 = (unsigned)a.Cin * ELEM, xrow1 = (unsigned)a.Cin2 * ELEM;
    const unsigned porg0 = (unsigned)plo * xrow0, porg1 = (unsigned)plo * xrow1;
    const unsigned swzo = (unsigned)((psl ^ swz_tr<SLOTS, TR>(prl)) << 4);     // (a piece is 16 rows: the swizzle of a piece row does not depend on the piece)
    auto rfl = [](unsigned v) -> unsigned { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
    auto patch_piece = [&](const char* src, unsigned uoff, unsigned xrow, int q, int buf) {
        int row = q * RPP + prl;
        row = row < prow ? row : prow - 1;
        dma16(src, uoff + (unsigned)row * xrow + swzo, lds_base + A_BASE + buf * ABUF + q * 1024);
    };
    // per-lane source offset of this wave's weight piece `wid`: (n0 + 16 wid) rows (scalar, wsoff) + this lane's row and swizzled slot,
    // made where it is used from prl and swzo (one v_lshl_add: the register budget of this kernel is counted in single registers);
    // pieces wid + 4, wid + 8 are 4096, 8192 bytes further (64 rows each)
    const unsigned wsoff = (unsigned)(n0 + wid * RPP) * ROWB;
    auto wlane = [&]() -> unsigned {
        unsigned p = (unsigned)prl;
        asm volatile("" : "+v"(p));                                        // (opaque: not hoisted into a loop-invariant register)
        return p * ROWB + swzo;
    };
    const int s2w = __builtin_amdgcn_readfirstlane(wid < 2 ? 1 : 0);
    const unsigned s2m = (unsigned)__builtin_amdgcn_readfirstlane(wid < 2 ? -1 : 0);
    const unsigned long long s2mask = ((unsigned long long)s2m << 32) | s2m;
    const int p2 = wid, p3 = wid < 2 ? wid : wid + 2;
    auto stage_weights = [&](const char* wsrc, unsigned soff, int slot) {
#pragma unroll
        for (int k = 0; k < WPW; ++k) {
            const int i = wid + NW * k;
            if (i < WPT) dma16(wsrc, wlane() + rfl(wsoff + soff + (unsigned)k * 4096u), lds_base + slot * BTAP + i * 1024);
        }
    };
    const int nch0 = a.Cin / 32, nch1 = a.x2 ? a.Cin2 / 32 : 0;
    const unsigned wtile = (unsigned)a.Cout * ROWB, wtap = (unsigned)nch0 * wtile;
    for (int q = wid; q < apieces; q += NW) patch_piece(a.x, porg0, xrow0, q, 0);
    stage_weights(a.w, 0, 0);
    stage_weights(a.w, wtap, 1);

    // per-lane LDS offsets (within a patch buffer) of this lane's four A rows for every tap, as 16-bit halves
    constexpr int NAP = (NA * TAPS + 1) / 2;
    unsigned apk[NAP];
    {
        // Branch-free on purpose: written with && / ?: hipcc turns every entry into a divergent branch (exec save, branch, restore: 103
        // exec saves and 68 branches in this prologue, ~1400 instructions for two waves per SIMD to issue = 11 k cycles per tile).
        // Validity per row offset dy and column offset dx once per fragment (unsigned compares: one per bound pair), entries as selects.
#pragma unroll
        for (int k = 0; k < NAP; ++k) apk[k] = 0;
        const unsigned zad = (unsigned)(AROWS * ROWB + 16 * lh);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int m = m0 + wid * RW + i * TR + lr;
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
                const int dy = t / 3 - 1, dx = t % 3 - 1;
                const int row = base + dy * g.W + dx;
                const unsigned adv = (unsigned)row * ROWB + 16u * ((unsigned)lh ^ (unsigned)swz_tr<SLOTS, TR>(row));
                const unsigned mask = 0u - (okh[dy + 1] & okw[dx + 1]);               // all ones where the tap is inside the image
                const unsigned ad = ((adv & mask) | (zad & ~mask)) & 0xffffu;
                apk[(i * TAPS + t) >> 1] |= ad << (16 * ((i * TAPS + t) & 1));
            }
        }
    }
    auto aaddr = [&](int i, int t) -> int {
        const int idx = i * TAPS + t;
        return (idx & 1) ? (int)(apk[idx >> 1] >> 16) : (int)(apk[idx >> 1] & 0xffffu);
    };
    const int baddr0 = lr * ROWB + 16 * (lh ^ swz_tr<SLOTS, TR>(lr));    // B tile j: + j * 16 rows (a multiple of the swizzle period)

    // accumulators: zero, or (swapped operands: lane = pixel, registers = four consecutive channels) the BN shift of the register's channel
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        if constexpr (SWAPC) {
            const f32x4 sh = *reinterpret_cast<const f32x4*>(a.shift + n0 + j * TR + 4 * lh);
#pragma unroll
            for (int i = 0; i < NA; ++i) acc[i][j] = sh;
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto mma_ab = [&](const uint4& xa, const uint4& wb, f32x4& c) {
        if constexpr (SWAPC) mma16(wb, xa, c); else mma16(xa, wb, c);
    };
    int after_last_is_center = 0;
    auto rd_a = [&](int i, int aoff, int tap) -> uint4 {
        const int ad = tap >= 0 ? aaddr(i, tap)
                                : (int)vsel(after_last_is_center ? ~0ull : 0ull, (unsigned)aaddr(i, CENTER), (unsigned)aaddr(i, 0));
        return *reinterpret_cast<const uint4*>(smem + aoff + ad);
    };
    auto rd_b = [&](int j, int boff) -> uint4 { return *reinterpret_cast<const uint4*>(smem + boff + j * (TR * ROWB) + baddr0); };

    uint4 fa[NA], ring[4];          // NA = 4
    uint4 fa2[2][2], ring5[5];      // NA = 2: A sets of two steps, B ring of five
    // One step on the fragments in registers (RO = 0 or 2: b_j of this step sits in ring slot (j + RO) % 4):
    //   P0   a0b0 a0b1 a1b0 a1b1 | a2b0 a2b1 a3b0 a3b1      reads: b2 in front, b3 in the middle
    //   G2 .. G7   a0..a3 x b_G                             reads: b_{G+2} in front of each           (MID between G4 and G5)
    //   P8   a0b8 a0b9 | a1b8 a1b9 | a2b8 a2b9 | a3b8 a3b9  reads (NEXT): b'_0 b'_1 in front, a'_i behind a_i's last MFMA, INTO a_i's registers
    // so ONE A set serves: a fragment of the next step lands >= 6 MFMAs (96 matrix-pipe cycles) before its first use, every B fragment
    // >= 8 (128).  The next step's b'_0 / b'_1 go to the slots of b6 / b7, i.e. the next step runs with RO ^ 2.
    // DMA slots: in front of G5, G6, G7 and P8.
    auto pair0 = [&](auto ro_tag, int boff) {
        constexpr int RO = decltype(ro_tag)::value;
        if constexpr (SUBREG_WIDE_DIAG != 2) {
            ring[(2 + RO) & 3] = rd_b(2, boff);
            mma_ab(fa[0], ring[(0 + RO) & 3], acc[0][0]); mma_ab(fa[0], ring[(1 + RO) & 3], acc[0][1]);
            mma_ab(fa[1], ring[(0 + RO) & 3], acc[1][0]); mma_ab(fa[1], ring[(1 + RO) & 3], acc[1][1]);
            ring[(3 + RO) & 3] = rd_b(3, boff);
            mma_ab(fa[2], ring[(0 + RO) & 3], acc[2][0]); mma_ab(fa[2], ring[(1 + RO) & 3], acc[2][1]);
            mma_ab(fa[3], ring[(0 + RO) & 3], acc[3][0]); mma_ab(fa[3], ring[(1 + RO) & 3], acc[3][1]);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // B reads of the B-major section (SUBREG_W16_AHEAD): 2 = b_{G+2} in front of group G (128 cycles ahead); 3 = the ring's fourth slot
    // used too: b4 b5 in front of G2, then b_{G+3} (192 cycles ahead), the next step's b'_0 in front of G7, b'_1 in front of the last pair
    auto group = [&](auto g_tag, auto ro_tag, auto next_tag, int boff, int n_boff, auto&& dma) {
        constexpr int G = decltype(g_tag)::value, RO = decltype(ro_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        if (SUBREG_WIDE_DIAG != 1) dma(g_tag);
        if constexpr (SUBREG_WIDE_DIAG != 2) {
            if constexpr (SUBREG_W16_AHEAD == 2) {
                ring[(G + 2 + RO) & 3] = rd_b(G + 2, boff);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            } else {
                constexpr int NRD = G == 2 ? 2 : (G <= 6 ? 1 : (NEXT ? 1 : 0));
                if constexpr (G == 2) { ring[(4 + RO) & 3] = rd_b(4, boff); ring[(5 + RO) & 3] = rd_b(5, boff); }
                else if constexpr (G <= 6) ring[(G + 3 + RO) & 3] = rd_b(G + 3, boff);
                else if constexpr (NEXT) ring[(2 + RO) & 3] = rd_b(0, n_boff);          // G = 7: b'_0 into the slot of b6
                if constexpr (NRD > 0) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) mma_ab(fa[i], ring[(G + RO) & 3], acc[i][G]);
            __builtin_amdgcn_sched_group_barrier(0x008, NA, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto pair8 = [&](auto ro_tag, auto next_tag, int n_aoff, auto ntap_tag, int n_boff, auto&& dma) {
        constexpr int RO = decltype(ro_tag)::value, NTAP = decltype(ntap_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        if (SUBREG_WIDE_DIAG != 1) dma(IC<8>{});
        if constexpr (SUBREG_WIDE_DIAG != 2) {
            if constexpr (NEXT) {
                if constexpr (SUBREG_W16_AHEAD == 2) ring[(2 + RO) & 3] = rd_b(0, n_boff);
                ring[(3 + RO) & 3] = rd_b(1, n_boff);
                __builtin_amdgcn_sched_group_barrier(0x100, SUBREG_W16_AHEAD == 2 ? 2 : 1, 0);
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                mma_ab(fa[i], ring[(0 + RO) & 3], acc[i][8]);             // b8: slot (8 + RO) % 4
                mma_ab(fa[i], ring[(1 + RO) & 3], acc[i][9]);
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if constexpr (NEXT) {
                    fa[i] = rd_a(i, n_aoff, NTAP);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // NA = 2: group G of a step on A set CUR: DMA slot, reads (b_{G+4}: G <= 5 this step's, G >= 6 the next step's b'_{G-6}; G = 5, 6: a'_0, a'_1 of
    // the next step into the other A set), two MFMAs
    auto sgroup = [&](auto g_tag, auto cur_tag, auto next_tag, int boff, int n_aoff, auto ntap_tag, int n_boff, auto&& dma) {
        constexpr int G = decltype(g_tag)::value, CUR = decltype(cur_tag)::value, NTAP = decltype(ntap_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        if (SUBREG_WIDE_DIAG != 1) dma(g_tag);
        if constexpr (SUBREG_WIDE_DIAG != 2) {
            constexpr int NRD = ((G <= 5 || NEXT) ? 1 : 0) + ((NEXT && (G == 5 || G == 6)) ? 1 : 0);
            if constexpr (G <= 5) ring5[(G + 4) % 5] = rd_b(G + 4, boff);
            else if constexpr (NEXT) ring5[(G + 4) % 5] = rd_b(G - 6, n_boff);
            if constexpr (NEXT && (G == 5 || G == 6)) fa2[CUR ^ 1][G - 5] = rd_a(G - 5, n_aoff, NTAP);
            mma_ab(fa2[CUR][0], ring5[G % 5], acc[0][G]);
            mma_ab(fa2[CUR][1], ring5[G % 5], acc[1][G]);
            if constexpr (NRD > 0) __builtin_amdgcn_sched_group_barrier(0x100, NRD, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto no_dma = [](auto) {};
    // the first step's fragments: a_0 .. a_3, b_0, b_1 (NA = 2: a_0, a_1, b_0 .. b_3)
    if constexpr (NA == 4) {
#pragma unroll
        for (int i = 0; i < NA; ++i) fa[i] = rd_a(i, A_BASE, 0);
        ring[0] = rd_b(0, 0);
        ring[1] = rd_b(1, 0);
    } else {
        fa2[0][0] = rd_a(0, A_BASE, 0);
        fa2[0][1] = rd_a(1, A_BASE, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) ring5[j] = rd_b(j, 0);
    }

    auto mid_sync = [&](bool drain_lds, int do_wait = 0) {
        unsigned long long q0 = 0, q1 = 0;
        if (STAMPS) q0 = __builtin_amdgcn_s_memtime();
        if (drain_lds) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        else if (do_wait == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (STAMPS) q1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (STAMPS) { const unsigned long long q2 = __builtin_amdgcn_s_memtime(); t_wait += q1 - q0; t_bar += q2 - q1; }
    };
    if (ENDS) t_loop = __builtin_amdgcn_s_memtime();

    // ---- phase 0: the convolution chunks, nine steps each
    const unsigned wl0 = lds_base + (unsigned)wid * 1024u;
    for (int c = 0; c < nch0; ++c) {
        const int aoff = A_BASE + (c & 1) * ABUF;
        const int naoff = A_BASE + ((c + 1) & 1) * ABUF;
        const bool last_c = c + 1 >= nch0;
        const bool to_x2 = last_c && nch1 > 0;
        const char* const psrc = to_x2 ? a.x2 : a.x;
        const unsigned pxrow = to_x2 ? xrow1 : xrow0;
        const unsigned puoff = to_x2 ? porg1 : porg0 + (unsigned)(last_c ? c : c + 1) * (32 * ELEM);
        const unsigned plds = lds_base + A_BASE + ((c + 1) & 1) * ABUF;
        after_last_is_center = last_c ? 1 : 0;
        unsigned cw = (unsigned)c * wtile;
        asm volatile("" : "+s"(cw));
        const int lim = prow - 1 - (apieces - 1) * RPP;
        // per-lane part of a patch piece's source offset: row prl of the piece + swizzled slot, made where it is used (see wlane)
        auto plane = [&]() -> unsigned {
            unsigned p = (unsigned)prl;
            asm volatile("" : "+v"(p));
            return __umul24(p, pxrow) + swzo;
        };
        const unsigned gs = (unsigned)RPP * pxrow;
        auto step = [&](auto tap_tag) {
            constexpr int TAP = decltype(tap_tag)::value;
            constexpr int SL = TAP % NWB, RO = 2 * (TAP & 1);            // (a chunk starts with b_0, b_1 in slots 0, 1)
            constexpr bool SAME = TAP < 8;
            constexpr int NT = SAME ? TAP + 1 : -1;
            const int boff = SL * BTAP, nboff = ((TAP + 1) % NWB) * BTAP, n_aoff = SAME ? aoff : naoff;
            constexpr int CUR = TAP & 1;                                  // (NA = 2: a chunk starts on A set 0)
            if constexpr (NA == 4) {
                pair0(IC<RO>{}, boff);
                group(IC<2>{}, IC<RO>{}, std::true_type{}, boff, nboff, no_dma);
                group(IC<3>{}, IC<RO>{}, std::true_type{}, boff, nboff, no_dma);
                group(IC<4>{}, IC<RO>{}, std::true_type{}, boff, nboff, no_dma);
            } else {
                sgroup(IC<0>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
                sgroup(IC<1>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
                sgroup(IC<2>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
                sgroup(IC<3>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
                sgroup(IC<4>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
            }
            mid_sync(false, (TAP >= 1 && TAP - 1 <= PLAST) ? 1 : 0);
            constexpr int WSL = (TAP + 2) % NWB;
            const char* wsrc = a.w;
            unsigned woff = cw;
            if constexpr (TAP + 2 <= 8) {
                woff += (unsigned)(TAP + 2) * wtap;
            } else {
                constexpr int T2 = TAP + 2 - 9;
                if (!last_c) woff += (unsigned)T2 * wtap + wtile;
                else if (T2 < nch1) { wsrc = a.w2; woff = (unsigned)T2 * wtile; }
                else woff = 0;
            }
            int q0 = TAP * NPP;
            asm volatile("" : "+s"(q0));
            auto dma = [&](auto g_tag) {
                constexpr int gq = decltype(g_tag)::value - 5;            // slots in front of groups 5 .. 8
                if constexpr (gq == 0) {                                  // S2
                    int q = q0 + p2;
                    q = q < apieces - 2 ? q : apieces - 2;
                    const unsigned so = ssel(s2w, wsoff + woff + 8192u, puoff + (unsigned)q * gs);   // (weight piece wid + 8 | patch piece q)
                    const unsigned la = ssel(s2w, wl0 + WSL * BTAP + 8 * 1024, plds + (unsigned)q * 1024u);
                    dma16(ssel_ptr(s2w, wsrc, psrc), vsel(s2mask, wlane(), plane()) + so, la);
                } else if constexpr (gq == 1 || gq == 2) {                // W0, W1
                    dma16(wsrc, wlane() + rfl(wsoff + woff + (unsigned)(gq - 1) * 4096u), wl0 + WSL * BTAP + (gq - 1) * 4096);
                } else if constexpr (gq == 3 && TAP < PLAST) {            // S3: a regular patch piece
                    int q = q0 + p3;
                    q = q < apieces - 2 ? q : apieces - 2;
                    dma16(psrc, plane() + rfl(puoff + (unsigned)q * gs), plds + (unsigned)q * 1024u);
                } else if constexpr (gq == 3 && TAP == PLAST) {           // S3: the last patch piece
                    // (tail rows clamped to the patch's last row; made here, once per chunk, instead of held in a register)
                    int prl_o = prl;
                    asm volatile("" : "+v"(prl_o));
                    const unsigned pv_last = __umul24((unsigned)(prl_o < lim ? prl_o : lim), pxrow) + swzo;
                    dma16(psrc, pv_last + rfl(puoff + (unsigned)(apieces - 1) * gs), plds + (unsigned)(apieces - 1) * 1024u);
                }
            };
            if constexpr (NA == 4) {
                group(IC<5>{}, IC<RO>{}, std::true_type{}, boff, nboff, dma);
                group(IC<6>{}, IC<RO>{}, std::true_type{}, boff, nboff, dma);
                group(IC<7>{}, IC<RO>{}, std::true_type{}, boff, nboff, dma);
                pair8(IC<RO>{}, std::true_type{}, n_aoff, IC<NT>{}, nboff, dma);
            } else {
                sgroup(IC<5>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, dma);
                sgroup(IC<6>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, dma);
                sgroup(IC<7>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, dma);
                sgroup(IC<8>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, dma);
                sgroup(IC<9>{}, IC<CUR>{}, std::true_type{}, boff, n_aoff, IC<NT>{}, nboff, no_dma);
            }
        };
        step(IC<0>{});
        step(IC<1>{});
        step(IC<2>{});
        step(IC<3>{});
        step(IC<4>{});
        step(IC<5>{});
        step(IC<6>{});
        step(IC<7>{});
        step(IC<8>{});
        if constexpr (NA == 4) {
            ring[0] = ring[2];                                            // step 8 (RO = 0) left b'_0, b'_1 in slots 2, 3: the next chunk
            ring[1] = ring[3];                                            // (or the first shortcut step) starts with RO = 0 again (8 moves per 360 MFMAs)
        } else {
            fa2[0][0] = fa2[1][0];                                        // step 8 ran on A set 0 and read the next step's into set 1
            fa2[0][1] = fa2[1][1];
        }
    }
    // ---- phase 1: the fused shortcut GEMM's chunks, ONE step each (centre tap, RO = 0); the next chunk's patch is staged
    //      at MID and waited for at the end of the step, where the next step's first fragments are read (not pipelined deeper)
    for (int d = 0; d < nch1; ++d) {
        const int s = 9 * nch0 + d;
        const int naoff = A_BASE + ((nch0 + d + 1) & 1) * ABUF;
        const int boff = (s % NWB) * BTAP, nboff = ((s + 1) % NWB) * BTAP;
        const bool more = d + 1 < nch1;
        if constexpr (NA == 4) {
            pair0(IC<0>{}, boff);
            group(IC<2>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
            group(IC<3>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
            group(IC<4>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
        } else {
            sgroup(IC<0>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<1>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<2>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<3>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<4>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
        }
        mid_sync(true);                                                   // (the patch buffer written below is the one step s-1 read)
        if (SUBREG_WIDE_DIAG != 1) {
            if (d + 2 < nch1) stage_weights(a.w2, (unsigned)(d + 2) * wtile, (s + 2) % NWB);
            if (more) for (int q = wid; q < apieces; q += NW) patch_piece(a.x2, porg1 + (unsigned)(d + 1) * (32 * ELEM), xrow1, q, (nch0 + d + 1) & 1);
        }
        if constexpr (NA == 4) {
            group(IC<5>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
            group(IC<6>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
            group(IC<7>{}, IC<0>{}, std::false_type{}, boff, 0, no_dma);
            pair8(IC<0>{}, std::false_type{}, 0, IC<0>{}, 0, no_dma);
        } else {
            sgroup(IC<5>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<6>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<7>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<8>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
            sgroup(IC<9>{}, IC<0>{}, std::false_type{}, boff, 0, IC<0>{}, 0, no_dma);
        }
        if (more) {
            mid_sync(false);
            if constexpr (NA == 4) {
#pragma unroll
                for (int i = 0; i < NA; ++i) fa[i] = rd_a(i, naoff, CENTER);
                ring[0] = rd_b(0, nboff);
                ring[1] = rd_b(1, nboff);
            } else {
                fa2[0][0] = rd_a(0, naoff, CENTER);
                fa2[0][1] = rd_a(1, naoff, CENTER);
#pragma unroll
                for (int j = 0; j < 4; ++j) ring5[j] = rd_b(j, nboff);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    struct EpilogueStamp {
        float* dst;
        unsigned long long t0;
        __device__ ~EpilogueStamp() { if (dst) *dst = (float)(__builtin_amdgcn_s_memtime() - t0); }
    } epi_stamp{nullptr, 0};
    if (ENDS && a.stats && lane == 0) {
        const unsigned long long n = __builtin_amdgcn_s_memtime(), rn = __builtin_amdgcn_s_memrealtime();
        float* d = a.stats + ((size_t)blockIdx.x * NW + wid) * 8;
        d[0] = (float)(t_loop - t_begin); d[1] = (float)(n - t_loop); d[2] = 0.f; d[3] = (float)(n - t_loop) - (float)t_wait - (float)t_bar;
        d[4] = (float)t_wait; d[5] = (float)t_bar; d[6] = (float)(rn - r_begin);
        epi_stamp.dst = d + 7;
        epi_stamp.t0 = n;
    }

    // ------------------------------------------------------------------ epilogue
    T* const y = reinterpret_cast<T*>(a.y);
    const bool full = m0 + TM <= g.M;
    constexpr int RS = TN * ELEM + 16, VPR = TN * ELEM / 16;
    char* const slab = smem + wid * (32 * RS);
    static_assert(NW * 32 * RS <= A_BASE + 2 * ABUF, "slabs reuse the staging LDS");
    const float slope = a.act ? 0.1f : 1.f;
    if constexpr (SWAPC) {
        // lane = pixel (row lr of A tile i), registers 0..3 of tile (i, j) = channels 16 j + 4 lh + {0..3}
        auto lrelu_pack = [&](const f32x4& cfr) -> uint2 {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = cfr[e] * slope;
                asm("v_max_f32 %0, %1, %2" : "=v"(v[e]) : "v"(cfr[e]), "v"(t));
            }
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (__bf16)v[e];
            return __builtin_bit_cast(uint2, o);
        };
        if (full) {
            constexpr int NV = 32 * VPR;
            char* const wbase = slab + lr * RS + 4 * lh * ELEM;
#pragma unroll
            for (int ib = 0; ib < NA / 2; ++ib) {                         // 32 rows at a time through the wave's slab
#pragma unroll
                for (int ii = 0; ii < 2; ++ii)
#pragma unroll
                    for (int j = 0; j < NB; ++j)
                        *reinterpret_cast<uint2*>(wbase + ii * TR * RS + j * TR * ELEM) = lrelu_pack(acc[2 * ib + ii][j]);
                const int mrow0 = m0 + wid * RW + ib * 32;
                char* const ybase = a.y + ((size_t)mrow0 * a.Cout + n0) * ELEM;
#pragma unroll
                for (int v0 = 0; v0 < NV; v0 += 64) {
                    const int v = v0 + lane;
                    const int row = v / VPR, c16 = v % VPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int m = m0 + wid * RW + i * TR + lr;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int n = n0 + j * TR + 4 * lh;
                    if (m < g.M) *reinterpret_cast<uint2*>(y + (size_t)m * a.Cout + n) = lrelu_pack(acc[i][j]);
                }
            }
        }
    } else {
        // pixel-major: column = lane % 16 (channel of tile j), registers 0..3 = rows 4 lh + {0..3} of A tile i = ONE 2x2 window
        if (full) {
            constexpr int NV = 4 * NA * VPR;                  // 4 NA pooled rows per wave
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float sh = s_shift[j * TR + lr];
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const f32x4& cfr = acc[i][j];
                    float best = fmaxf(fmaxf(cfr[0], cfr[1]), fmaxf(cfr[2], cfr[3])) + sh;
                    best = fmaxf(best, best * slope);                     // monotone: lrelu(max) == max(lrelu)
                    *reinterpret_cast<T*>(slab + (4 * i + lh) * RS + (j * TR + lr) * ELEM) = (T)best;
                }
            }
            const int win0 = (m0 + wid * RW) >> 2;                        // first pooled pixel of this wave
            char* const ybase = a.y + ((size_t)win0 * a.Cout + n0) * ELEM;
#pragma unroll
            for (int v0 = 0; v0 < NV; v0 += 64) {
                const int v = v0 + lane;
                if (NV % 64 == 0 || v < NV) {
                    const int row = v / VPR, c16 = v % VPR;
                    const uint4 val = *reinterpret_cast<const uint4*>(slab + row * RS + c16 * 16);
                    *reinterpret_cast<uint4*>(ybase + (size_t)row * a.Cout * ELEM + c16 * 16) = val;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const int n = n0 + j * TR + lr;
                const float sh = s_shift[j * TR + lr];
#pragma unroll
                for (int i = 0; i < NA; ++i) {
                    const int m = m0 + wid * RW + i * TR + 4 * lh;
                    if (m < g.M) {
                        const f32x4& cfr = acc[i][j];
                        float best = fmaxf(fmaxf(cfr[0], cfr[1]), fmaxf(cfr[2], cfr[3])) + sh;
                        best = fmaxf(best, best * slope);
                        y[(size_t)(m >> 2) * a.Cout + n] = (T)best;
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------- host side
namespace {

// largest patch (pixel rows) any m-tile of the problem stages.  The walk over the m-tiles is O(M / TM) host work (~8 k patch_range calls at
// 1125 images of 42x42) and the dispatcher asks twice per launch: the last answers are kept per (B, H, W, pool, TM).
template <bool POOL>
int worst_patch_rows_w(const ConvGeom& g, int TM) {
    struct Entry { int B, H, W, TM, pool, worst; };
    static std::mutex mu;
    static Entry cache[16];
    static int n_cached = 0, next_slot = 0;
    {
        std::lock_guard<std::mutex> lk(mu);
        for (int i = 0; i < n_cached; ++i)
            if (cache[i].B == g.B && cache[i].H == g.H && cache[i].W == g.W && cache[i].TM == TM && cache[i].pool == (POOL ? 1 : 0)) return cache[i].worst;
    }
    int worst = 0;
    for (int m0 = 0; m0 < g.M; m0 += TM) {
        int lo, hi;
        patch_range<POOL>(g, m0, TM, &lo, &hi);
        if (hi - lo > worst) worst = hi - lo;
    }
    std::lock_guard<std::mutex> lk(mu);
    cache[next_slot] = Entry{g.B, g.H, g.W, TM, POOL ? 1 : 0, worst};
    next_slot = (next_slot + 1) % 16;
    if (n_cached < 16) ++n_cached;
    return worst;
}

template <int MI, int WM, int WN, bool POOL, int AROWS, int MINW>
int launch_wide(const ConvArgs& a, hipStream_t stream) {
    constexpr int TM = WM * MI * 32, TN = WN * 160;
    const size_t lds = 3 * (size_t)TN * 64 + 2 * (size_t)(AROWS + 1) * 64 + TN * sizeof(float);
    auto kern = conv_wide_kernel<MI, WM, WN, POOL, AROWS, MINW>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    dim3 grid(((a.g.M + TM - 1) / TM) * (a.Cout / TN));
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, stream, a);
    return launch_status();
}

template <int NA, bool POOL, int AROWS, int MINW>
int launch_wide16(const ConvArgs& a, hipStream_t stream) {
    constexpr int TM = 64 * NA, TN = 160;
    const size_t lds = 3 * (size_t)TN * 64 + 2 * (size_t)(AROWS + 1) * 64 + TN * sizeof(float);
    auto kern = conv_wide16_kernel<NA, POOL, AROWS, MINW>;
    static std::atomic<unsigned long long> lds_set{0};
    if (const int rc = ensure_dynamic_lds(reinterpret_cast<const void*>(kern), lds, lds_set)) return rc;
    dim3 grid(((a.g.M + TM - 1) / TM) * (a.Cout / TN));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, stream, a);
    return launch_status();
}
// patch rows of the 128-row tiling of conv_wide16_kernel: 160 (the 10x10 / 5x5 maps: 52 KB of LDS, three workgroups per CU) or 224
// (21x21 / 42x42 un-pooled, 21x21 / 10x10 pooled: 60 KB, two)
constexpr int AR_S1 = 160, AR_S2 = 224;
// MFMA shape of the 256-row tiling.  Default: 16x16x32 (conv_wide16_kernel); SUBREG_WIDE_TR=32 selects conv_wide_kernel<2> (measurements),
// SUBREG_CONV_KERNEL_WIDE_ALT in a call's flags the other one than the default (parity tests).
int wide_tr_env() { static const int v = [] { const char* e = getenv("SUBREG_WIDE_TR"); return e && *e ? atoi(e) : 0; }(); return v == 16 || v == 32 ? v : 0; }

// Two tilings of the same kernel body (SUBREG_WIDE_MI picks; measurements):
//   MI = 3: 96 x 160 wave tiles, 384 x 160 tiles, ONE workgroup per CU (a wave owns its SIMD's 512 registers)
//   MI = 2: 64 x 160 wave tiles, 256 x 160 tiles, TWO workgroups per CU (256 registers per wave: 160 accumulators + two fragment
//           sets) - the general kernel's tile under this kernel's loop (reads a k-step ahead, weights 1.5 steps ahead, unconditional
//           slots), with a second workgroup to cover prologue, epilogue and the step's waits
int wide_mi() { static const int v = [] { const char* e = getenv("SUBREG_WIDE_MI"); return e && *e ? atoi(e) : 2; }(); return v == 3 ? 3 : 2; }
int wide_tm() { return 4 * wide_mi() * 32; }
constexpr int AR_LIN3 = 480, AR_POOL3 = 544;                         // patch rows: 384 + 2 (W + 1) at W <= 42; pooled window-major tiles
constexpr int AR_LIN2 = 352, AR_POOL2 = 384;                         // 256 + 2 (W + 1); pooled

}  // namespace

bool conv_wide_supported(const ConvArgs& a, bool pool) {
    if (a.g.taps != 9 || a.raw || a.res || a.scale || !a.shift || a.part) return false;
    if (a.Cout % 160 != 0 || a.Cin % 32 != 0 || a.Cin < 32) return false;
    if (a.x2 && (a.Cin2 % 32 != 0 || a.Cin2 < 32)) return false;
    if ((long long)a.g.M * a.Cout >= (1LL << 31) || (long long)a.g.npix * a.Cout >= (1LL << 31)) return false;
    if ((long long)a.g.npix * a.Cin * 2 >= (1LL << 32) || (long long)a.g.npix * a.Cin2 * 2 >= (1LL << 32) ||
        (long long)a.g.taps * a.Cin * a.Cout * 2 >= (1LL << 32))
        return false;
    if (a.g.npix < 16) return false;                                   // (a patch piece's tail rows re-read valid rows)
    const int worst = pool ? worst_patch_rows_w<true>(a.g, wide_tm()) : worst_patch_rows_w<false>(a.g, wide_tm());
    return worst <= (wide_mi() == 3 ? (pool ? AR_POOL3 : AR_LIN3) : (pool ? AR_POOL2 : AR_LIN2));
}

// Measured rule (profiles/r05_wide2_vs_general.txt: both kernels per layer at ten batch sizes, one box).  The 256-row / two-workgroup
// variant (MI = 2) is the general kernel's tile under this file's loop; it is faster wherever the general kernel ALSO runs 256-row
// tiles or nearly so - the 42x42 maps (layer 2: -3 ... -14 % on conv2 / conv3 at every batch from 125 to 1000 images; conv1, K = 576,
// only from ~460 images up) and the pooled 21x21 conv (layer 3.0's conv3: -2 ... -5 % from 250 images up) - and slower on the small
// maps, where the general kernel's 128-row tiles and its two-waves-per-tile form fill the chip better.
bool conv_wide_preferred(const ConvArgs& a, bool pool) {
    static const int mode = [] { const char* e = getenv("SUBREG_WIDE"); return e && *e ? atoi(e) : -1; }();   // 0 never, 1 always, -1 rule
    if (mode == 0) return false;
    if (!conv_wide_supported(a, pool)) return false;
    if (mode == 1) return true;
    if (wide_mi() != 2) return false;                                  // (the rule was measured for the 256-row variant)
    if (conv_wide_default_tr(pool) == 32) {
        // the 32x32x16 form (round 5, profiles/r05_wide2_vs_general.txt): the 42x42 maps and the pooled 21x21 conv
        if (a.g.M < 200000) return false;
        if (a.g.W >= 42) return a.Cin >= 160 || a.g.M >= 800000;
        if (pool && a.g.W == 21) return a.g.M >= 100000;
        return false;
    }
    // 16x16x32 form (profiles/r06_wide16_vs_general.txt: both kernels per layer at 250 ... 1125 images, one box, interleaved).  It wins
    // wherever the 256-row tiles fill the chip: -5 ... -13 % on layer 2 (conv1, K = 576, from ~350 images), -2 ... -9 % on layer 3.0 and
    // layer 4.0 from ~450 images; on layer 3.1 (10x10, 320 channels: 548 tiles at 700 images for 512 slots) the general kernel's
    // 128-row tiles win at every batch; the 5x5 maps only from ~1000 images (-7 %).
    // (SUBREG_WIDE_CLASSES: measurement switch - the layer classes that take this kernel whatever their size: A = 42x42 from 64 channels,
    // B = 42x42 from 160 (un-pooled and pooled), C = 21x21 un-pooled, D = 21x21 pooled, E = 10x10 to 640 channels, F = 10x10 to 320, G = 5x5)
    static const char* classes = getenv("SUBREG_WIDE_CLASSES");
    if (classes) {
        const char c = a.g.W >= 42 ? (a.Cin >= 160 ? 'B' : 'A') : a.g.W >= 21 ? (pool ? 'D' : 'C') : a.g.W >= 10 ? (a.Cout >= 640 ? 'E' : 'F') : 'G';
        for (const char* p = classes; *p; ++p)
            if (*p == c) return true;
        return false;
    }
    // (SUBREG_WIDE_SCALE: measurement switch, multiplies the thresholds - the table above is per launch on one stream, the forward runs two
    // lanes side by side, where a partial round of tiles costs less)
    static const double sc = [] { const char* e = getenv("SUBREG_WIDE_SCALE"); return e && *e ? atof(e) : 1.0; }();
    static const int rule = [] { const char* e = getenv("SUBREG_WIDE_RULE"); return e && *e ? atoi(e) : 3; }();
    const double M = (double)a.g.M;
    if (rule == 1) {           // the per-launch table alone (one stream, 20 launches back to back)
        if (a.g.W >= 42) return a.Cin >= 160 ? M >= 200000 * sc : M >= 600000 * sc;
        if (a.g.W >= 21) return M >= (pool ? 100000 : 200000) * sc;
        if (a.g.W >= 10) return a.Cout >= 640 && M >= 50000 * sc;
        return M >= 24000 * sc;
    }
    // rules 2 / 3 (3 = default): re-measured where the kernels RUN - in the two-lane forward, one layer class at a time at eight batches
    // (profiles/r06_forward_classes.txt), then whole rules against each other (r06_forward_rules.txt: rule 2 -0.4 %, rule 3 -0.9 % over
    // the eight batches, -1.3 % at 1000 / 1125 images).  Beside another lane's kernels a partial round of 256-row tiles costs less than
    // alone, so layer 3.0's un-pooled convolutions pay from ~190 images per lane, not ~450; and with every other wide layer on this
    // kernel layers 3.1 / 4.1 pay from 500 images per lane although each alone does not (the gains are not additive: the chip holds a
    // higher clock the more of the forward runs on the cheaper MFMA shape).
    if (a.g.W >= 42) return a.Cin >= 160 ? M >= 200000 * sc : M >= 550000 * sc;
    if (a.g.W >= 21) return M >= (pool ? 120000 : 80000) * sc;
    if (a.g.W >= 10) return a.Cout >= 640 ? M >= 50000 * sc : (rule >= 3 && M >= 50000 * sc);
    return rule >= 3 && M >= 12500 * sc;
}

int conv_wide_default_tr(bool pool) { (void)pool; return wide_tr_env() ? wide_tr_env() : 16; }

// measured rule of the 128-row tiling against the 256-row one (filled in from profiles/r06_wide16s.txt)
bool wide16s_preferred(const ConvArgs& a, bool pool);

// worst patch of the 128-row tiling, or 0 when it does not take the problem
int wide16s_arows(const ConvArgs& a, bool pool) {
    if (!conv_wide_supported(a, pool)) return 0;                       // (the argument checks are shared)
    const int worst = pool ? worst_patch_rows_w<true>(a.g, 128) : worst_patch_rows_w<false>(a.g, 128);
    return worst <= AR_S1 ? AR_S1 : (worst <= AR_S2 ? AR_S2 : 0);
}

// Tile height of the 16x16x32 form for this problem: 128 where the 128-row tiling is the faster one (profiles/r06_wide16s.txt), else 256.
int conv_wide_default_rows(const ConvArgs& a, bool pool) {
    static const int env = [] { const char* e = getenv("SUBREG_WIDE_ROWS"); return e && *e ? atoi(e) : 0; }();
    const int ar = wide16s_arows(a, pool);
    if (env == 128) return ar ? 128 : 256;
    if (env == 256 || !ar) return 256;
    return wide16s_preferred(a, pool) ? 128 : 256;
}

int conv_wide(const ConvArgs& a, bool pool, hipStream_t stream, int tr, int rows) {
    if (!conv_wide_supported(a, pool)) return SUBREG_EUNSUPPORTED;
    if (tr == 0) tr = conv_wide_default_tr(pool);
    if (wide_mi() == 2 && tr == 16) {
        if (rows == 0) rows = conv_wide_default_rows(a, pool);
        if (rows == 128) {
            const int ar = wide16s_arows(a, pool);
            if (ar == AR_S1) return pool ? launch_wide16<2, true, AR_S1, 3>(a, stream) : launch_wide16<2, false, AR_S1, 3>(a, stream);
            if (ar == AR_S2) return pool ? launch_wide16<2, true, AR_S2, 2>(a, stream) : launch_wide16<2, false, AR_S2, 2>(a, stream);
            return SUBREG_EUNSUPPORTED;
        }
        return pool ? launch_wide16<4, true, AR_POOL2, 2>(a, stream) : launch_wide16<4, false, AR_LIN2, 2>(a, stream);
    }
    if (wide_mi() == 2) return pool ? launch_wide<2, 4, 1, true, AR_POOL2, 2>(a, stream) : launch_wide<2, 4, 1, false, AR_LIN2, 2>(a, stream);
    return pool ? launch_wide<3, 4, 1, true, AR_POOL3, 1>(a, stream) : launch_wide<3, 4, 1, false, AR_LIN3, 1>(a, stream);
}

bool wide16s_preferred(const ConvArgs& a, bool pool) {
    // Measured (profiles/r06_wide16s.txt: general | 256-row | 128-row per layer at 125 ... 1000 images, one box, interleaved): the 128-row
    // tiling of this file's loop runs its steps at 94-98 % matrix-pipe duty on the 10x10 / 5x5 maps (stamps in the same file) and
    // still ends within +-3 % of the general kernel's 128-row tiles there (0.98-1.03 at 375-700 images; slower at 125-250 and 1000) and
    // 5-20 % behind on the larger maps (two workgroups per CU): what those layers lose is rounds of tiles, not loop cycles.  Never
    // selected; SUBREG_WIDE_ROWS=128 / SUBREG_CONV_KERNEL_WIDE_128 run it (parity tests, measurements).
    (void)a; (void)pool;
    return false;
}

}  // namespace subreg
