// Kernel arguments of the implicit-GEMM convolution kernels (conv_fwd.hip: the general kernel; conv_wide.hip: the
// one-wave-per-SIMD kernel of the wide layers).  See conv_fwd.hip for the data layout.
#pragma once
#include <hip/hip_runtime.h>

#include "conv_index.h"

namespace subreg {

struct ConvArgs {
    const char* x;       // [npix][Cin] T
    const char* w;       // [taps][Cin/32][Cout][32] T
    const char* x2;      // fused shortcut GEMM: [npix][Cin2] T (centre tap only) or null
    const char* w2;      // [Cin2/32][Cout][32] T
    char* y;             // LINEAR [npix][Cout] T ; POOL [B*Hp*Wp][Cout] T
    const float* scale;  // [Cout] or null (scale folded into the weights)
    const float* shift;  // [Cout]
    const char* res;     // [npix][Cout] T residual or null
    float* stats;        // raw: [m_tiles*WAVES_M][Cout][2] partial (sum, sumsq)
    ConvGeom g;
    int Cin, Cin2, Cout;
    int act;             // LeakyReLU(0.1) after scale/shift/residual
    int raw;             // write the un-normalised conv + stats partials
    float* part;         // SPLITK kernels: fp32 partial sums [ksplit][M][Cout] (grid.y = ksplit); splitk_reduce_kernel finishes
    int ksplit;
};

// conv_wide.hip: eval-mode bf16 3x3 convolutions with Cout % 160 == 0 (scale folded into the weights, optional fused shortcut
// GEMM, optional 2x2 max-pool) on 512-row x 160-channel tiles, one 4-wave workgroup per CU.  SUBREG_EUNSUPPORTED when the
// problem is outside that set (the caller then takes the general kernel).
// tr: MFMA tile edge of the 256-row tiling (16: conv_wide16_kernel, 32: conv_wide_kernel<2>); 0 = the default for the problem
// rows: tile height of the 16x16x32 form (256, or 128 where the problem's patch fits); 0 = the measured default for the problem
int conv_wide(const ConvArgs& a, bool pool, hipStream_t stream, int tr = 0, int rows = 0);
int conv_wide_default_tr(bool pool);
// does conv_wide take this problem, and should the dispatcher prefer it (measured rule)?
bool conv_wide_supported(const ConvArgs& a, bool pool);
bool conv_wide_preferred(const ConvArgs& a, bool pool);

}  // namespace subreg
