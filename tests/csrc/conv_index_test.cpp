// Host-side emulation of conv_fwd.hip's tiling with the SAME index functions (conv_index.h):
// every tile stages the contiguous pixel patch [lo,hi), every GEMM row reads its taps at
// (pixel + dy*W + dx - lo) or a zero row, POOL rows are window-major and pooled by groups of 4.
// Checks against a naive NHWC convolution (+2x2 floor max-pool).  Build: g++ -O2 -std=c++17.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../subspace-reg_amd/csrc/conv_index.h"

using namespace subreg;

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xFFFF) / 65536.0f - 0.5f; }

template <bool POOL>
static int run_case(int B, int H, int W, int C, int O, int taps, int TM, int AROWS) {
    ConvGeom g = make_geom(B, H, W, taps, POOL);
    unsigned seed = 1234u + B * 7 + H * 13 + W * 17 + C + O + taps + (POOL ? 99 : 0);
    std::vector<float> x((size_t)g.npix * C), w((size_t)O * taps * C);
    for (auto& v : x) v = frand(seed);
    for (auto& v : w) v = frand(seed);
    // naive reference: y[p][o]
    std::vector<float> ref((size_t)g.npix * O, 0.f);
    for (int b = 0; b < B; ++b) for (int h = 0; h < H; ++h) for (int ww = 0; ww < W; ++ww)
        for (int o = 0; o < O; ++o) {
            double acc = 0;
            for (int t = 0; t < taps; ++t) {
                const int dy = taps == 9 ? t / 3 - 1 : 0, dx = taps == 9 ? t % 3 - 1 : 0;
                const int hh = h + dy, w2 = ww + dx;
                if (hh < 0 || hh >= H || w2 < 0 || w2 >= W) continue;
                for (int c = 0; c < C; ++c) acc += (double)x[(((size_t)b * H + hh) * W + w2) * C + c] * w[((size_t)o * taps + t) * C + c];
            }
            ref[(((size_t)b * H + h) * W + ww) * O + o] = (float)acc;
        }
    const int Mout = POOL ? g.M / 4 : g.M;
    std::vector<float> got((size_t)Mout * O, NAN);
    int errors = 0, worst_rows = 0;
    for (int m0 = 0; m0 < g.M; m0 += TM) {
        int lo, hi;
        patch_range<POOL>(g, m0, TM, &lo, &hi);
        const int prow = hi - lo;
        if (prow > worst_rows) worst_rows = prow;
        if (prow > AROWS) { printf("patch %d rows > AROWS %d\n", prow, AROWS); return 1; }
        std::vector<float> acc((size_t)TM * O, 0.f);
        for (int r = 0; r < TM; ++r) {
            const int m = m0 + r;
            if (m >= g.M) continue;
            const Pix px = row_to_pixel<POOL>(g, m);
            for (int t = 0; t < taps; ++t) {
                const int dy = taps == 9 ? t / 3 - 1 : 0, dx = taps == 9 ? t % 3 - 1 : 0;
                if (!tap_valid(g, px.h, px.w, dy, dx)) continue;           // -> zero row
                const int row = px.p + dy * g.W + dx - lo;
                if (row < 0 || row >= prow) { if (errors++ < 5) printf("row %d outside patch [0,%d) m=%d t=%d\n", row, prow, m, t); continue; }
                for (int o = 0; o < O; ++o) {
                    double a = 0;
                    for (int c = 0; c < C; ++c) a += (double)x[((size_t)lo + row) * C + c] * w[((size_t)o * taps + t) * C + c];
                    acc[(size_t)r * O + o] += (float)a;
                }
            }
        }
        for (int r = 0; r < TM; ++r) {
            const int m = m0 + r;
            if (m >= g.M) continue;
            if (!POOL) { for (int o = 0; o < O; ++o) got[(size_t)m * O + o] = acc[(size_t)r * O + o]; }
            else if ((r & 3) == 0) {
                for (int o = 0; o < O; ++o) {
                    float best = -3e38f;
                    for (int s = 0; s < 4; ++s) best = std::fmax(best, acc[(size_t)(r + s) * O + o]);
                    got[(size_t)(m >> 2) * O + o] = best;
                }
            }
        }
    }
    // compare
    for (int b = 0; b < B; ++b)
        for (int ho = 0; ho < (POOL ? g.Hp : H); ++ho)
            for (int wo = 0; wo < (POOL ? g.Wp : W); ++wo)
                for (int o = 0; o < O; ++o) {
                    float want;
                    size_t idx;
                    if (!POOL) { idx = (((size_t)b * H + ho) * W + wo); want = ref[idx * O + o]; }
                    else {
                        idx = (((size_t)b * g.Hp + ho) * g.Wp + wo);
                        want = -3e38f;
                        for (int s = 0; s < 4; ++s)
                            want = std::fmax(want, ref[((((size_t)b * H + 2 * ho + (s >> 1)) * W + 2 * wo + (s & 1))) * O + o]);
                    }
                    const float gv = got[idx * O + o];
                    if (!(std::fabs(gv - want) <= 1e-4f + 1e-4f * std::fabs(want))) { if (errors++ < 5) printf("mismatch b%d %d,%d o%d got %g want %g\n", b, ho, wo, o, gv, want); }
                }
    printf("B%d %dx%d C%d O%d taps%d pool%d TM%d: worst patch rows %d, errors %d\n", B, H, W, C, O, taps, (int)POOL, TM, worst_rows, errors);
    return errors ? 1 : 0;
}

template <int SLOTS>
static int swizzle_check() {
    // (row, slot) -> physical slot must be a bijection per row, and conflict-free for ds_read_b128's 16-lane groups
    int bad = 0;
    const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int rowb = SLOTS * 16;
    for (int off = 0; off < 64; ++off)
        for (int slot = 0; slot < SLOTS; ++slot)
            for (int gi = 0; gi < 2; ++gi) {
                int used[16] = {0};
                for (int l = 0; l < 16; ++l) {
                    const int row = off + groups[gi][l];
                    const int addr = row * rowb + ((slot ^ swz<SLOTS>(row)) << 4);
                    const int bank_slot = (addr / 16) % 16;          // 16-byte units of the 256-byte bank row
                    if (used[bank_slot]++) bad++;
                }
            }
    printf("swizzle SLOTS=%d: %d conflicts\n", SLOTS, bad);
    return bad ? 1 : 0;
}

// 16x16x32 MFMA operand reads: lane l reads row off + l%16 at logical slot l/16 (bf16, 64-byte rows)
static int swizzle_check_tr16() {
    int bad = 0;
    const int g0[16] = {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27};
    const int g1[16] = {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31};
    for (int off = 0; off < 64; ++off)
        for (int half = 0; half < 2; ++half)
            for (int gi = 0; gi < 2; ++gi) {
                int used[16] = {0};
                for (int l = 0; l < 16; ++l) {
                    const int lane = (gi ? g1[l] : g0[l]) + 32 * half;
                    const int row = off + lane % 16, slot = lane / 16;
                    const int addr = row * 64 + ((slot ^ swz_tr<4, 16>(row)) << 4);
                    if (used[(addr / 16) % 16]++) bad++;
                }
            }
    for (int row = 0; row < 64; ++row) {                     // bijection per row
        int seen = 0;
        for (int slot = 0; slot < 4; ++slot) seen |= 1 << (slot ^ swz_tr<4, 16>(row));
        if (seen != 15) bad++;
    }
    printf("swizzle TR=16: %d conflicts\n", bad);
    return bad ? 1 : 0;
}

int main() {
    int rc = 0;
    rc |= swizzle_check<4>();
    rc |= swizzle_check<8>();
    rc |= swizzle_check_tr16();
    // bf16 config: TM 256, AROWS 704 ; f32 config: TM 128, AROWS 448
    rc |= run_case<false>(2, 84, 84, 4, 3, 9, 256, 704);
    rc |= run_case<true>(2, 84, 84, 4, 3, 9, 256, 704);
    rc |= run_case<true>(1, 84, 84, 4, 3, 9, 128, 448);
    rc |= run_case<false>(1, 84, 84, 4, 3, 9, 128, 448);
    rc |= run_case<true>(3, 42, 42, 4, 3, 9, 256, 704);
    rc |= run_case<true>(3, 21, 21, 4, 3, 9, 256, 704);
    rc |= run_case<true>(3, 21, 21, 4, 3, 9, 128, 448);
    rc |= run_case<false>(3, 21, 21, 4, 3, 1, 256, 704);
    rc |= run_case<true>(7, 10, 10, 4, 3, 9, 256, 704);
    rc |= run_case<false>(9, 5, 5, 4, 3, 9, 256, 704);
    rc |= run_case<true>(2, 9, 4, 4, 3, 1, 128, 448);
    rc |= run_case<false>(1, 5, 7, 4, 3, 9, 256, 704);
    rc |= run_case<true>(5, 6, 6, 4, 3, 9, 128, 448);
    rc |= run_case<true>(4, 32, 32, 4, 3, 9, 256, 704);
    printf(rc ? "FAILED\n" : "ALL OK\n");
    return rc;
}
