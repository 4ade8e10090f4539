// Classifier head, regularizers and the fused fine-tune step (gfx950).
//
// Reference call sites:
//   nn.Linear(640, n_cls, bias)          models/resnet_language.py:138-140,187
//   CrossEntropyLoss                     eval_incremental.py:118, eval/language_eval.py:252-258
//   LangPuller.get_projected_weight      models/resnet_language.py:92-97   (QR basis + projection)
//   LangPuller.loss1                     :89-90   pull * ||inspired - w||_F^2
//   ResNet.regloss / reglossnovel        :229-240 lmbd * ||W_rows - anchor||_F   (not squared, 0-subgradient at 0)
//   SGD(momentum, weight_decay)          eval/util.py:92-102 + loss.backward()/step, language_eval.py:293-295
//   stop rule                            language_eval.py:298-318
//   validate / accuracy                  language_eval.py:18-43, eval/util.py:26-40
// Everything here is tiny (<= 25 MFLOP, <= 1 MB) and launch-latency-bound, so the
// design goal is FEW launches with no host round trip: the per-epoch step is three
// launches (rows -> per-class-row update -> scalar finish) and the stop rule lives in
// a device-side state word so that epochs can be queued / graph-replayed ahead.
#include "subreg_common.h"

namespace subreg {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// block-wide sum (blockDim.x multiple of 64, <= 1024); result valid in every thread
template <typename F>
__device__ __forceinline__ F block_sum(F v, F* red /* >= 17 entries of LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (wid == 0) {
        F t = lane < nw ? red[lane] : (F)0;
        t = wave_sum(t);
        if (lane == 0) red[16] = t;
    }
    __syncthreads();
    return red[16];
}

__device__ __forceinline__ float block_max(float v, float* red /* >= 17 entries of LDS */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = red[0];
        for (int w = 1; w < nw; ++w) t = fmaxf(t, red[w]);
        red[16] = t;
    }
    __syncthreads();
    return red[16];
}

// ---------------------------------------------------------------- logits for one row b (block = 256 threads)
// wave w computes classes n = w, w+4, ...: lanes stride the feature dim (coalesced W rows), wave-reduce.
__device__ __forceinline__ void row_logits(const float* __restrict__ f, const float* __restrict__ W,
                                           const float* __restrict__ bias, int N, int D, float* s_logit) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // four class rows per pass: their loads are independent, so one memory latency covers four dot products (the rows are
    // L2-resident and a single chain of 10 loads + reduction per class was latency-bound: 40 us for 125 x 100 logits)
    constexpr int U = 4;
    for (int n0 = wid * U; n0 < N; n0 += nw * U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float fv = f[d];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int n = n0 + u < N ? n0 + u : N - 1;          // clamp: the surplus rows are computed and dropped
                acc[u] = fmaf(fv, W[(size_t)n * D + d], acc[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float t = wave_sum(acc[u]);
            if (lane == 0 && n0 + u < N) s_logit[n0 + u] = t + (bias ? bias[n0 + u] : 0.f);
        }
    }
}

constexpr int MAX_CLS = 1024;

__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ logits,
                                                          int N, int D) {
    __shared__ float s_logit[MAX_CLS];
    const int b = blockIdx.x;
    row_logits(feat + (size_t)b * D, W, bias, N, D, s_logit);
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) logits[(size_t)b * N + n] = s_logit[n];
}

// dW[n][d] = sum_b dlogits[b][n] * feat[b][d]      (block per class row n)
// Four independent partial sums per thread so that four feature loads are in flight (a single fmaf chain over the batch was
// 64 dependent round trips: 43 us for 60 x 640 outputs at B = 64); fixed summation order.
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ dlogits, const float* __restrict__ feat,
                                                            float* __restrict__ dW, int B, int N, int D) {
    const int n = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int b = 0;
        for (; b + 3 < B; b += 4) {
            const float f0 = feat[(size_t)b * D + d], f1 = feat[(size_t)(b + 1) * D + d];
            const float f2 = feat[(size_t)(b + 2) * D + d], f3 = feat[(size_t)(b + 3) * D + d];
            a0 = fmaf(dlogits[(size_t)b * N + n], f0, a0);
            a1 = fmaf(dlogits[(size_t)(b + 1) * N + n], f1, a1);
            a2 = fmaf(dlogits[(size_t)(b + 2) * N + n], f2, a2);
            a3 = fmaf(dlogits[(size_t)(b + 3) * N + n], f3, a3);
        }
        for (; b < B; ++b) a0 = fmaf(dlogits[(size_t)b * N + n], feat[(size_t)b * D + d], a0);
        dW[(size_t)n * D + d] = (a0 + a1) + (a2 + a3);
    }
}

// dfeat[b][d] = sum_n dlogits[b][n] * W[n][d]      (block per row b; four partial sums like above)
__global__ __launch_bounds__(256) void linear_bwd_x_kernel(const float* __restrict__ dlogits, const float* __restrict__ W,
                                                            float* __restrict__ dfeat, int N, int D) {
    const int b = blockIdx.x;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        int n = 0;
        for (; n + 3 < N; n += 4) {
            const float w0 = W[(size_t)n * D + d], w1 = W[(size_t)(n + 1) * D + d];
            const float w2 = W[(size_t)(n + 2) * D + d], w3 = W[(size_t)(n + 3) * D + d];
            a0 = fmaf(dlogits[(size_t)b * N + n], w0, a0);
            a1 = fmaf(dlogits[(size_t)b * N + n + 1], w1, a1);
            a2 = fmaf(dlogits[(size_t)b * N + n + 2], w2, a2);
            a3 = fmaf(dlogits[(size_t)b * N + n + 3], w3, a3);
        }
        for (; n < N; ++n) a0 = fmaf(dlogits[(size_t)b * N + n], W[(size_t)n * D + d], a0);
        dfeat[(size_t)b * D + d] = (a0 + a1) + (a2 + a3);
    }
}

// db[n] = sum_b dlogits[b][n]
__global__ void linear_bwd_b_kernel(const float* __restrict__ dlogits, float* __restrict__ db, int B, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float acc = 0.f;
    for (int b = 0; b < B; ++b) acc += dlogits[(size_t)b * N + n];
    db[n] = acc;
}

constexpr int MAX_BASE = 512;        // base classes (rows of W_base): 60 (miniImageNet) ... 351 (tieredImageNet); must be <= dim

// ---------------------------------------------------------------- orthonormal basis of span(W_base rows)
// One workgroup, classical Gram-Schmidt with re-orthogonalisation ("twice is enough") in fp64.
// Q [nb][D] row-major: row j = j-th orthonormal basis vector (== column j of the reference's Q up to sign;
// the projector Q Q^T that the regularizer uses is sign-invariant).  Returns rank deficiency through *info.
__global__ __launch_bounds__(1024) void subspace_basis_kernel(const float* __restrict__ Wb, float* __restrict__ Q,
                                                               double* __restrict__ scratch /* [nb][D] */, int nb, int D,
                                                               int* __restrict__ info) {
    __shared__ double red[17];
    __shared__ double coef[MAX_BASE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    int deficient = 0;
    for (int k = 0; k < nb; ++k) {
        for (int d = tid; d < D; d += blockDim.x) scratch[(size_t)k * D + d] = (double)Wb[(size_t)k * D + d];
        __syncthreads();
        double n0 = 0.0;
        for (int d = tid; d < D; d += blockDim.x) { const double v = scratch[(size_t)k * D + d]; n0 += v * v; }
        n0 = block_sum(n0, red);
        for (int pass = 0; pass < 2; ++pass) {
            for (int j = wid; j < k; j += nw) {          // all projections of the current vector in parallel
                double acc = 0.0;
                for (int d = lane; d < D; d += 64) acc += scratch[(size_t)j * D + d] * scratch[(size_t)k * D + d];
                acc = wave_sum(acc);
                if (lane == 0) coef[j] = acc;
            }
            __syncthreads();
            for (int d = tid; d < D; d += blockDim.x) {
                double v = scratch[(size_t)k * D + d];
                for (int j = 0; j < k; ++j) v -= coef[j] * scratch[(size_t)j * D + d];
                scratch[(size_t)k * D + d] = v;
            }
            __syncthreads();
        }
        double n1 = 0.0;
        for (int d = tid; d < D; d += blockDim.x) { const double v = scratch[(size_t)k * D + d]; n1 += v * v; }
        n1 = block_sum(n1, red);
        const bool bad = !(n1 > 1e-20 * (n0 > 0.0 ? n0 : 1.0));
        if (bad) deficient++;
        const double inv = bad ? 0.0 : 1.0 / sqrt(n1);
        for (int d = tid; d < D; d += blockDim.x) {
            const double v = scratch[(size_t)k * D + d] * inv;
            scratch[(size_t)k * D + d] = v;
            Q[(size_t)k * D + d] = (float)v;
        }
        __syncthreads();
    }
    if (tid == 0 && info) *info = deficient;
}

// c_j = w . q_j for all j, then P = sum_j c_j q_j   (one block per row of w; s_c has >= nb entries)
__device__ __forceinline__ void project_row(const float* __restrict__ w, const float* __restrict__ Q, int nb, int D,
                                            float* s_c) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    constexpr int U = 4;                                   // four basis rows per pass (independent loads), like row_logits
    for (int j0 = wid * U; j0 < nb; j0 += nw * U) {
        float acc[U];
#pragma unroll
        for (int u = 0; u < U; ++u) acc[u] = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float wv = w[d];
#pragma unroll
            for (int u = 0; u < U; ++u) acc[u] = fmaf(wv, Q[(size_t)(j0 + u < nb ? j0 + u : nb - 1) * D + d], acc[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float t = wave_sum(acc[u]);
            if (lane == 0 && j0 + u < nb) s_c[j0 + u] = t;
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void subspace_project_kernel(const float* __restrict__ w, const float* __restrict__ Q,
                                                                float* __restrict__ P, int nb, int D) {
    __shared__ float s_c[MAX_BASE];
    const float* wr = w + (size_t)blockIdx.x * D;
    project_row(wr, Q, nb, D, s_c);
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float p = 0.f;
        for (int j = 0; j < nb; ++j) p = fmaf(s_c[j], Q[(size_t)j * D + d], p);
        P[(size_t)blockIdx.x * D + d] = p;
    }
}

// loss = scale * sum (a-b)^2 ; optional grad_a = gscale * (a-b)   (single block; n <= a few 1e5)
__global__ __launch_bounds__(1024) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                       float scale, float* __restrict__ loss, const float* __restrict__ gout,
                                                       float gscale, float* __restrict__ grad_a, float* __restrict__ grad_b) {
    __shared__ double red[17];
    double s = 0.0;
    const float go = gout ? gout[0] : 1.f;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = a[i] - b[i];
        s += (double)d * d;
        if (grad_a) grad_a[i] = gscale * go * d;
        if (grad_b) grad_b[i] = -gscale * go * d;
    }
    if (loss) {
        s = block_sum(s, red);
        if (threadIdx.x == 0) loss[0] = (float)((double)scale * s);
    }
}

// Frobenius (not squared) regulariser: loss = lmbd*||a-b||, grad_a = gout*lmbd*(a-b)/||a-b|| (0 at 0)
__global__ __launch_bounds__(1024) void frob_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                     float lmbd, float* __restrict__ loss, const float* __restrict__ gout,
                                                     float* __restrict__ grad_a) {
    __shared__ double red[17];
    double s = 0.0;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) {
        const float d = a[i] - b[i];
        s += (double)d * d;
    }
    s = block_sum(s, red);
    const double nrm = sqrt(s);
    if (loss && threadIdx.x == 0) loss[0] = (float)((double)lmbd * nrm);
    if (grad_a) {
        const float k = nrm > 0.0 ? (float)((double)lmbd * (double)(gout ? gout[0] : 1.f) / nrm) : 0.f;
        for (long long i = threadIdx.x; i < n; i += blockDim.x) grad_a[i] = k * (a[i] - b[i]);
    }
}

// ================================================================ fused fine-tune step
// Device-resident loop state (one per session), see include/subreg_hip.h: subreg_loop_state.
struct StepArgs {
    const float* feat;      // [Bs+Bm][D] support rows then memory rows
    const long long* labels;// [Bs+Bm]
    int Bs, Bm, N, D;
    float* W;               // [N][D] live classifier.weight (updated in place)
    float* mom;             // [N][D] SGD momentum buffer
    const float* Wbase;     // [n_base][D]
    const float* Wprev;     // [n_prev][D] reserved novel rows or null
    const float* Q;         // [n_base][D] orthonormal basis rows (null => no subspace term)
    int n_base, n_prev, n_old;   // rows [n_old, N) are this session's novel rows
    float lr, momentum, wd, lmbd_base, lmbd_prev, pull;
    int use_base, use_prev, use_pull;
    float* dlogits;         // [Bs+Bm][N]
    float* rowloss;         // [Bs+Bm] un-scaled CE per row
    int* rowcorrect;        // [Bs+Bm] argmax == label
    float* norms;           // [3] ||W[:nb]-Wbase||, ||W[nb:nb+np]-Wprev||, ||b[:nb]-b_base|| (classifier with bias)
    float* rowl1;           // [N] pull*||P_n - w_n||^2 for the novel rows, 0 elsewhere
    const float* target;    // [N - n_old][D] constant pullers (semantic / mapping variants) or null (projection onto Q)
    subreg_loop_state* st;
    float* losses;          // [max_epochs] per-epoch loss
    float* train_acc;       // [max_epochs]
    int max_epochs, min_epochs, stable_epochs, stable_mode;
    float target_loss, eps;
    int adam;               // 0: SGD(momentum, wd); 1: torch.optim.Adam (mom = exp_avg, mom2 = exp_avg_sq)
    float beta1, beta2, adam_eps;
    float* mom2;
    // classifier WITH bias (nn.Linear(640, n, bias=opt.linear_bias), resnet_language.py:140): all null without
    float* bias;            // [N] live classifier.bias (updated in place)
    float* bmom;            // [N] its momentum buffer / exp_avg
    float* bmom2;           // [N] Adam exp_avg_sq
    const float* bias_base; // [n_base] base bias of regloss (:231-232)
};

// phase A: grid = Bs+Bm row blocks (+2 norm blocks): logits, softmax-CE, dlogits, argmax
__global__ __launch_bounds__(512) void step_rows_kernel(const StepArgs a) {
    __shared__ float s_logit[MAX_CLS];
    __shared__ double red[17];
    if (a.st->stop) return;
    const int Bt = a.Bs + a.Bm;
    const int b = blockIdx.x;
    if (b >= Bt) {            // the two Frobenius norms (read by phase B before any row is updated)
        const int which = b - Bt;
        const float* x = which == 0 ? a.W : a.W + (size_t)a.n_base * a.D;
        const float* y = which == 0 ? a.Wbase : a.Wprev;
        const long long n = which == 0 ? (long long)a.n_base * a.D : (long long)a.n_prev * a.D;
        double s = 0.0;
        if (y && ((which == 0 && a.use_base) || (which == 1 && a.use_prev))) {
            // these two blocks are the launch's critical path (38400 elements on 256 threads): four independent strided
            // chains per thread keep four load pairs in flight; fixed summation order
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
            const long long st = blockDim.x;
            long long i = threadIdx.x;
            for (; i + 3 * st < n; i += 4 * st) {
                const float d0 = x[i] - y[i], d1 = x[i + st] - y[i + st], d2 = x[i + 2 * st] - y[i + 2 * st], d3 = x[i + 3 * st] - y[i + 3 * st];
                s0 += (double)d0 * d0; s1 += (double)d1 * d1; s2 += (double)d2 * d2; s3 += (double)d3 * d3;
            }
            for (; i < n; i += st) { const float d = x[i] - y[i]; s0 += (double)d * d; }
            s = (s0 + s1) + (s2 + s3);
        }
        s = block_sum(s, red);
        if (threadIdx.x == 0) a.norms[which] = (float)sqrt(s);
        if (which == 0 && a.bias) {            // regloss' bias term: lmbd * ||b[:nb] - b_base||**2 (resnet_language.py:231-232)
            double sb = 0.0;
            if (a.use_base && a.bias_base)
                for (int i = threadIdx.x; i < a.n_base; i += blockDim.x) { const float d = a.bias[i] - a.bias_base[i]; sb += (double)d * d; }
            __syncthreads();
            sb = block_sum(sb, red);
            if (threadIdx.x == 0) a.norms[2] = (float)sqrt(sb);
        }
        return;
    }
    row_logits(a.feat + (size_t)b * a.D, a.W, a.bias, a.N, a.D, s_logit);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float mx = -3.0e38f;
        int arg = 0;
        for (int n = lane; n < a.N; n += 64) if (s_logit[n] > mx) { mx = s_logit[n]; arg = n; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {       // max with lowest-index tie-break (torch.topk/argmax order)
            const float om = __shfl_xor(mx, o);
            const int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        float se = 0.f;
        for (int n = lane; n < a.N; n += 64) se += expf(s_logit[n] - mx);
        se = wave_sum(se);
        const float lse = logf(se);
        const int y = (int)a.labels[b];
        const float inv = 1.f / (float)(b < a.Bs ? a.Bs : a.Bm);
        for (int n = lane; n < a.N; n += 64) {
            const float p = expf(s_logit[n] - mx - lse);
            a.dlogits[(size_t)b * a.N + n] = (p - (n == y ? 1.f : 0.f)) * inv;
        }
        if (lane == 0) {
            a.rowloss[b] = -(s_logit[y] - mx - lse);
            a.rowcorrect[b] = arg == y ? 1 : 0;
        }
    }
}

// phase B: grid = N class-row blocks: dW row, regulariser gradients, SGD(momentum, wd) update in place
__global__ __launch_bounds__(1024) void step_update_kernel(const StepArgs a) {
    __shared__ float s_dl[2048];
    __shared__ float s_c[MAX_BASE];
    __shared__ double red[17];
    if (a.st->stop) return;
    const int n = blockIdx.x, Bt = a.Bs + a.Bm, D = a.D;
    const bool first = a.st->epoch == 0;       // fresh optimiser: momentum buffer = first gradient
    for (int b = threadIdx.x; b < Bt; b += blockDim.x) s_dl[b] = a.dlogits[(size_t)b * a.N + n];
    __syncthreads();
    float* wr = a.W + (size_t)n * D;
    const bool novel = a.use_pull && (a.Q || a.target) && n >= a.n_old;
    if (novel && !a.target) project_row(wr, a.Q, a.n_base, D, s_c);
    float kb = 0.f, kp = 0.f;
    if (a.use_base && n < a.n_base) kb = a.norms[0] > 0.f ? a.lmbd_base / a.norms[0] : 0.f;
    if (a.use_prev && n >= a.n_base && n < a.n_base + a.n_prev) kp = a.norms[1] > 0.f ? a.lmbd_prev / a.norms[1] : 0.f;
    double l1 = 0.0;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        // dW[n][d] = sum_b dlogits[b][n] * feat[b][d]: four rows per pass so that four loads are in flight; the partial sums
        // are added in a fixed order (deterministic)
        float g0 = 0.f, g1 = 0.f, g2 = 0.f, g3 = 0.f;
        int b = 0;
        for (; b + 3 < Bt; b += 4) {
            const float f0 = a.feat[(size_t)b * D + d], f1 = a.feat[(size_t)(b + 1) * D + d];
            const float f2 = a.feat[(size_t)(b + 2) * D + d], f3 = a.feat[(size_t)(b + 3) * D + d];
            g0 = fmaf(s_dl[b], f0, g0);
            g1 = fmaf(s_dl[b + 1], f1, g1);
            g2 = fmaf(s_dl[b + 2], f2, g2);
            g3 = fmaf(s_dl[b + 3], f3, g3);
        }
        for (; b < Bt; ++b) g0 = fmaf(s_dl[b], a.feat[(size_t)b * D + d], g0);
        float g = (g0 + g1) + (g2 + g3);
        const float w = wr[d];
        if (kb != 0.f) g += kb * (w - a.Wbase[(size_t)n * D + d]);
        if (kp != 0.f) g += kp * (w - a.Wprev[(size_t)(n - a.n_base) * D + d]);
        if (novel) {
            float r;
            if (a.target) {                            // constant puller (semantic / linear-mapping variants): only d/dw
                r = a.target[(size_t)(n - a.n_old) * D + d] - w;
            } else {                                   // P(w) - w with P the projection onto span(W_base): both paths
                float p = 0.f;
                for (int j = 0; j < a.n_base; ++j) p = fmaf(s_c[j], a.Q[(size_t)j * D + d], p);
                r = p - w;
            }
            l1 += (double)r * r;
            g -= 2.f * a.pull * r;
        }
        g += a.wd * w;
        if (!a.adam) {
            const float m = first ? g : a.momentum * a.mom[(size_t)n * D + d] + g;
            a.mom[(size_t)n * D + d] = m;
            wr[d] = w - a.lr * m;
        } else {
            // torch.optim.Adam, single-tensor form: exp_avg.lerp_(g, 1 - b1); exp_avg_sq = b2 * v + (1 - b2) g^2;
            // denom = sqrt(v) / sqrt(1 - b2^t) + eps; p -= (lr / (1 - b1^t)) * exp_avg / denom   (scalars in double, like Python)
            const size_t e = (size_t)n * D + d;
            const float m0 = first ? 0.f : a.mom[e], v0 = first ? 0.f : a.mom2[e];
            const float m = m0 + (g - m0) * (1.f - a.beta1);
            const float v = v0 * a.beta2 + (1.f - a.beta2) * g * g;
            a.mom[e] = m;
            a.mom2[e] = v;
            const double t = (double)(a.st->epoch + 1);
            const double bc1 = 1.0 - pow((double)a.beta1, t), bc2 = 1.0 - pow((double)a.beta2, t);
            const float denom = sqrtf(v) / (float)sqrt(bc2) + a.adam_eps;
            wr[d] = w - (float)((double)a.lr / bc1) * (m / denom);
        }
    }
    l1 = block_sum(l1, red);
    if (threadIdx.x == 0) a.rowl1[n] = novel ? (float)((double)a.pull * l1) : 0.f;
    if (!a.bias) return;
    // the bias element of this class: d loss / d b[n] = sum_b dlogits[b][n] (+ 2*lmbd*(b - b_base), regloss :231-232), then the
    // same optimiser update as a weight element (the bias is in net.parameters(): weight decay applies to it too)
    double gs = 0.0;
    for (int b = threadIdx.x; b < Bt; b += blockDim.x) gs += (double)s_dl[b];
    __syncthreads();
    gs = block_sum(gs, red);
    if (threadIdx.x != 0) return;
    const float bw = a.bias[n];
    float g = (float)gs;
    if (a.use_base && a.bias_base && n < a.n_base) g += 2.f * a.lmbd_base * (bw - a.bias_base[n]);
    g += a.wd * bw;
    if (!a.adam) {
        const float m = first ? g : a.momentum * a.bmom[n] + g;
        a.bmom[n] = m;
        a.bias[n] = bw - a.lr * m;
    } else {
        const float m0 = first ? 0.f : a.bmom[n], v0 = first ? 0.f : a.bmom2[n];
        const float m = m0 + (g - m0) * (1.f - a.beta1);
        const float v = v0 * a.beta2 + (1.f - a.beta2) * g * g;
        a.bmom[n] = m;
        a.bmom2[n] = v;
        const double t = (double)(a.st->epoch + 1);
        const double bc1 = 1.0 - pow((double)a.beta1, t), bc2 = 1.0 - pow((double)a.beta2, t);
        const float denom = sqrtf(v) / (float)sqrt(bc2) + a.adam_eps;
        a.bias[n] = bw - (float)((double)a.lr / bc1) * (m / denom);
    }
}

// phase C: one block: assemble the loss, train accuracy, stop rule
__global__ __launch_bounds__(256) void step_finish_kernel(const StepArgs a) {
    __shared__ double red[17];
    subreg_loop_state* st = a.st;
    if (st->stop) return;
    const int Bt = a.Bs + a.Bm;
    double ces = 0.0, cem = 0.0, l1 = 0.0, corr = 0.0;
    for (int b = threadIdx.x; b < Bt; b += blockDim.x) {
        if (b < a.Bs) { ces += (double)a.rowloss[b]; corr += (double)a.rowcorrect[b]; }
        else cem += (double)a.rowloss[b];
    }
    for (int n = threadIdx.x; n < a.N; n += blockDim.x) l1 += (double)a.rowl1[n];
    ces = block_sum(ces, red);
    cem = block_sum(cem, red);
    l1 = block_sum(l1, red);
    corr = block_sum(corr, red);
    if (threadIdx.x != 0) return;
    // same fp32 accumulation order as the reference: CE_s (+ CE_m) (+ regloss) (+ reglossnovel) (+ loss1)
    float loss = (float)(ces / (double)a.Bs);
    if (a.Bm > 0) loss += (float)(cem / (double)a.Bm);
    if (a.use_base) {                          // reg = lmbd*norm(dW); reg += lmbd*norm(db)**2; loss += reg   (resnet_language.py:229-233)
        float reg = a.lmbd_base * a.norms[0];
        if (a.bias) reg += a.lmbd_base * (a.norms[2] * a.norms[2]);
        loss += reg;
    }
    if (a.use_prev) loss += a.lmbd_prev * a.norms[1];
    if (a.use_pull) loss += (float)l1;
    const int epoch = st->epoch + 1;           // 1-based epoch that just ran
    if (epoch - 1 < a.max_epochs) {
        a.losses[epoch - 1] = loss;
        a.train_acc[epoch - 1] = (float)(corr * (100.0 / (double)a.Bs));
    }
    int stop = 0;
    if (a.stable_mode) {                        // language_eval.py:300-305
        const double dlt = fabs((double)loss - (double)st->train_loss);
        st->stable = dlt < (double)a.eps ? st->stable + 1 : 0;
        if (st->stable == a.stable_epochs) stop = 1;
    }
    st->train_loss = loss;
    if (epoch >= a.max_epochs || (loss <= a.target_loss && epoch >= a.min_epochs + 1)) stop = 1;   // :317-318
    st->epoch = epoch;
    st->stop = stop;
}

// validation: one block per query row: argmax == label -> integer counter for (epoch slot, set)
__global__ __launch_bounds__(512) void validate_kernel(const float* __restrict__ feat, const long long* __restrict__ labels,
                                                        const float* __restrict__ W, const float* __restrict__ bias, int N, int D,
                                                        subreg_loop_state* st, int* __restrict__ correct, int set_index,
                                                        int n_sets_max, int is_last_set) {
    __shared__ float s_logit[MAX_CLS];
    int slot = 0;
    if (st) {
        if (st->val_epoch == st->epoch) return;      // this epoch's validation is already recorded (loop has stopped)
        slot = st->epoch;
    }
    const int b = blockIdx.x;
    row_logits(feat + (size_t)b * D, W, bias, N, D, s_logit);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float mx = -3.0e38f;
        int arg = 0;
        for (int n = lane; n < N; n += 64) if (s_logit[n] > mx) { mx = s_logit[n]; arg = n; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o);
            const int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        if (lane == 0 && arg == (int)labels[b]) atomicAdd(&correct[(size_t)slot * n_sets_max + set_index], 1);
    }
    (void)is_last_set;
}

// the same for ALL query sets of a session in one launch (the rows of set j follow those of set j-1; one launch per set
// cost ~19 us each for 125-row sets, i.e. up to 8 x per epoch): block b finds its set from the prefix sums
struct ValidateSets {
    int n_sets;
    int end[SUBREG_MAX_QUERY_SETS];      // exclusive prefix sums of the sets' row counts
};
__global__ __launch_bounds__(512) void validate_sets_kernel(const float* __restrict__ feat, const long long* __restrict__ labels,
                                                             const float* __restrict__ W, const float* __restrict__ bias, int N, int D,
                                                             subreg_loop_state* st, int* __restrict__ correct, int* __restrict__ correct5, int n_sets_max,
                                                             const ValidateSets vs) {
    __shared__ float s_logit[MAX_CLS];
    int slot = 0;
    if (st) {
        if (st->val_epoch == st->epoch) return;
        slot = st->epoch;
    }
    const int b = blockIdx.x;
    int set = 0;
    while (set + 1 < vs.n_sets && b >= vs.end[set]) ++set;
    row_logits(feat + (size_t)b * D, W, bias, N, D, s_logit);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        float mx = -3.0e38f;
        int arg = 0;
        for (int n = lane; n < N; n += 64) if (s_logit[n] > mx) { mx = s_logit[n]; arg = n; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float om = __shfl_xor(mx, o);
            const int oa = __shfl_xor(arg, o);
            if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
        }
        if (lane == 0 && arg == (int)labels[b]) atomicAdd(&correct[(size_t)slot * n_sets_max + set], 1);
        if (correct5) {
            // top-5 (eval/util.py:26-40 `accuracy(..., topk=(1, 5))`): the label is among the five largest logits <=> fewer than
            // five classes rank above it (strictly greater, ties towards the lower index like torch.topk)
            const int y = (int)labels[b];
            const bool yv = y >= 0 && y < N;                 // a label that is not (yet) a classifier row is a miss, like top-1
            const float ty = s_logit[yv ? y : 0];
            int above = yv ? 0 : 5;
            for (int n = lane; n < N; n += 64) above += (s_logit[n] > ty || (s_logit[n] == ty && n < y)) ? 1 : 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) above += __shfl_xor(above, o);
            if (lane == 0 && above < 5) atomicAdd(&correct5[(size_t)slot * n_sets_max + set], 1);
        }
    }
}

__global__ void validate_mark_kernel(subreg_loop_state* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) st->val_epoch = st->epoch;
}

// ---------------------------------------------------------------- CrossEntropyLoss(mean) + top-1 / top-k counters
// One wave per row: row loss, dlogits = (softmax - onehot) / B, and whether the label is the argmax / within the k largest
// (eval/util.py:26-40 `accuracy`: rank = logits strictly greater, ties broken towards the lower index).
__global__ __launch_bounds__(64) void softmax_ce_kernel(const float* __restrict__ logits, const long long* __restrict__ labels, int B,
                                                        int N, int k, float* __restrict__ rowloss, float* __restrict__ dlogits,
                                                        int* __restrict__ correct) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* z = logits + (size_t)b * N;
    const int y = (int)labels[b];
    float mx = -3.0e38f;
    for (int n = lane; n < N; n += 64) mx = fmaxf(mx, z[n]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.f;
    int rank = 0;
    const float zy = z[y];
    for (int n = lane; n < N; n += 64) {
        se += expf(z[n] - mx);
        rank += (z[n] > zy || (z[n] == zy && n < y)) ? 1 : 0;
    }
    se = wave_sum(se);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rank += __shfl_xor(rank, o);
    const float lse = logf(se);
    if (dlogits) {
        const float inv = 1.f / (float)B;
        for (int n = lane; n < N; n += 64) dlogits[(size_t)b * N + n] = (expf(z[n] - mx - lse) - (n == y ? 1.f : 0.f)) * inv;
    }
    if (lane == 0) {
        if (rowloss) rowloss[b] = -(zy - mx - lse);
        if (correct) {
            if (rank == 0) atomicAdd(&correct[0], 1);
            if (rank < k) atomicAdd(&correct[1], 1);
        }
    }
}

__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ double red[17];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += (double)x[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (float)(s / (double)n);
}

// ---------------------------------------------------------------- semantic subspace regularizer target
// LangPuller.forward, models/resnet_language.py:75-83: scores = E_novel E_base^T (optionally diagonal := -9999),
// probs = softmax(scores / temp, dim=1), target = probs @ W_base.  One block per novel row; n_base <= 1024.
__global__ __launch_bounds__(256) void semantic_target_kernel(const float* __restrict__ en, const float* __restrict__ eb,
                                                              const float* __restrict__ wb, int n_base, int edim, int D,
                                                              float temp, int mask_diag, float* __restrict__ probs,
                                                              float* __restrict__ target) {
    __shared__ float s_p[1024];
    __shared__ float red[17];
    const int r = blockIdx.x;
    const float* e = en + (size_t)r * edim;
    for (int j = threadIdx.x; j < n_base; j += blockDim.x) {
        float acc = 0.f;
        for (int k = 0; k < edim; ++k) acc = fmaf(e[k], eb[(size_t)j * edim + k], acc);
        if (mask_diag && j == r) acc = -9999.f;                      // scores.fill_diagonal_(-9999), :80-81
        s_p[j] = acc / temp;
    }
    __syncthreads();
    float mx = -3.0e38f;
    for (int j = threadIdx.x; j < n_base; j += blockDim.x) mx = fmaxf(mx, s_p[j]);
    mx = block_max(mx, red);
    float sum = 0.f;
    for (int j = threadIdx.x; j < n_base; j += blockDim.x) {
        const float ex = expf(s_p[j] - mx);
        s_p[j] = ex;
        sum += ex;
    }
    sum = block_sum(sum, red);
    __syncthreads();
    for (int j = threadIdx.x; j < n_base; j += blockDim.x) {
        s_p[j] /= sum;
        if (probs) probs[(size_t)r * n_base + j] = s_p[j];
    }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float acc = 0.f;
        for (int j = 0; j < n_base; ++j) acc = fmaf(s_p[j], wb[(size_t)j * D + d], acc);
        target[(size_t)r * D + d] = acc;
    }
}

// d W_base = probs^T @ d target (the scores do not depend on W_base)
__global__ void semantic_target_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ dt, int n_novel, int n_base,
                                           int D, float* __restrict__ dwb) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)n_base * D) return;
    const int d = i % D, j = i / D;
    float acc = 0.f;
    for (int r = 0; r < n_novel; ++r) acc = fmaf(probs[(size_t)r * n_base + j], dt[(size_t)r * D + d], acc);
    dwb[i] = acc;
}

__global__ void loop_state_init_kernel(subreg_loop_state* st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st->epoch = 0; st->stop = 0; st->stable = 0; st->val_epoch = -1; st->train_loss = 15.f;   // language_eval.py:234-239
    }
}

}  // namespace subreg

using namespace subreg;

extern "C" int subreg_linear_fwd(const float* feat, const float* weight, const float* bias, float* logits, int B, int N,
                                 int D, void* stream) {
    SUBREG_CHECK_ARG(feat && weight && logits && B > 0 && N > 0 && N <= MAX_CLS && D > 0);
    hipLaunchKernelGGL(linear_fwd_kernel, B, 256, 0, (hipStream_t)stream, feat, weight, bias, logits, N, D);
    return launch_status();
}

extern "C" int subreg_linear_bwd(const float* dlogits, const float* feat, const float* weight, float* dweight, float* dbias,
                                 float* dfeat, int B, int N, int D, void* stream) {
    SUBREG_CHECK_ARG(dlogits && B > 0 && N > 0 && D > 0);
    hipStream_t s = (hipStream_t)stream;
    if (dweight) {
        SUBREG_CHECK_ARG(feat != nullptr);
        hipLaunchKernelGGL(linear_bwd_w_kernel, N, 256, 0, s, dlogits, feat, dweight, B, N, D);
    }
    if (dbias) hipLaunchKernelGGL(linear_bwd_b_kernel, (N + 255) / 256, 256, 0, s, dlogits, dbias, B, N);
    if (dfeat) {
        SUBREG_CHECK_ARG(weight != nullptr);
        hipLaunchKernelGGL(linear_bwd_x_kernel, B, 256, 0, s, dlogits, weight, dfeat, N, D);
    }
    return launch_status();
}

extern "C" int subreg_subspace_basis(const float* w_base, float* q, double* scratch, int n_base, int D, int* info,
                                     void* stream) {
    SUBREG_CHECK_ARG(w_base && q && scratch && n_base > 0 && n_base <= MAX_BASE && D > 0);
    hipLaunchKernelGGL(subspace_basis_kernel, 1, 1024, 0, (hipStream_t)stream, w_base, q, scratch, n_base, D, info);
    return launch_status();
}

extern "C" int subreg_subspace_project(const float* w, const float* q, float* p, int k, int n_base, int D, void* stream) {
    SUBREG_CHECK_ARG(w && q && p && k > 0 && n_base > 0 && n_base <= MAX_BASE && D > 0);
    hipLaunchKernelGGL(subspace_project_kernel, k, 256, 0, (hipStream_t)stream, w, q, p, n_base, D);
    return launch_status();
}

extern "C" int subreg_sqdiff(const float* a, const float* b, long long n, float scale, float* loss, const float* grad_out,
                             float gscale, float* grad_a, float* grad_b, void* stream) {
    SUBREG_CHECK_ARG(a && b && n > 0);
    hipLaunchKernelGGL(sqdiff_kernel, 1, 1024, 0, (hipStream_t)stream, a, b, n, scale, loss, grad_out, gscale, grad_a, grad_b);
    return launch_status();
}

extern "C" int subreg_frob(const float* a, const float* b, long long n, float lmbd, float* loss, const float* grad_out,
                           float* grad_a, void* stream) {
    SUBREG_CHECK_ARG(a && b && n > 0);
    hipLaunchKernelGGL(frob_kernel, 1, 1024, 0, (hipStream_t)stream, a, b, n, lmbd, loss, grad_out, grad_a);
    return launch_status();
}

extern "C" int subreg_validate_sets(const float* feat, const long long* labels, const float* weight, const float* bias,
                                    const int* set_rows, int n_sets, int N, int D, subreg_loop_state* state, int* correct, int* correct_top5, int n_sets_max,
                                    int mark_done, void* stream) {
    SUBREG_CHECK_ARG(feat && labels && weight && correct && set_rows && N > 0 && N <= MAX_CLS && D > 0);
    SUBREG_CHECK_ARG(n_sets >= 1 && n_sets <= SUBREG_MAX_QUERY_SETS && n_sets <= n_sets_max);
    ValidateSets vs;
    vs.n_sets = n_sets;
    int total = 0;
    for (int j = 0; j < n_sets; ++j) {
        SUBREG_CHECK_ARG(set_rows[j] > 0);
        total += set_rows[j];
        vs.end[j] = total;
    }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(validate_sets_kernel, total, 512, 0, s, feat, labels, weight, bias, N, D, state, correct, correct_top5, n_sets_max, vs);
    if (mark_done && state) hipLaunchKernelGGL(validate_mark_kernel, 1, 64, 0, s, state);
    return launch_status();
}

extern "C" int subreg_softmax_ce(const float* logits, const long long* labels, int B, int N, int topk, float* rowloss, float* loss,
                                 float* dlogits, int* correct, void* stream) {
    SUBREG_CHECK_ARG(logits && labels && B > 0 && N > 0 && topk >= 1 && (!loss || rowloss));
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(softmax_ce_kernel, B, 64, 0, s, logits, labels, B, N, topk, rowloss, dlogits, correct);
    if (loss) hipLaunchKernelGGL(mean_kernel, 1, 256, 0, s, rowloss, B, loss);
    return launch_status();
}

extern "C" int subreg_semantic_target(const float* novel_embeds, const float* base_embeds, const float* base_weight, int n_novel,
                                      int n_base, int embed_dim, int dim, float temperature, int mask_diagonal, float* probs,
                                      float* target, void* stream) {
    SUBREG_CHECK_ARG(novel_embeds && base_embeds && base_weight && target);
    SUBREG_CHECK_ARG(n_novel > 0 && n_base > 0 && n_base <= 1024 && embed_dim > 0 && dim > 0 && temperature != 0.f);
    hipLaunchKernelGGL(semantic_target_kernel, n_novel, 256, 0, (hipStream_t)stream, novel_embeds, base_embeds, base_weight, n_base,
                       embed_dim, dim, temperature, mask_diagonal, probs, target);
    return launch_status();
}

extern "C" int subreg_semantic_target_bwd(const float* probs, const float* grad_target, int n_novel, int n_base, int dim,
                                          float* grad_base_weight, void* stream) {
    SUBREG_CHECK_ARG(probs && grad_target && grad_base_weight && n_novel > 0 && n_base > 0 && dim > 0);
    const size_t n = (size_t)n_base * dim;
    hipLaunchKernelGGL(semantic_target_bwd_kernel, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream, probs, grad_target,
                       n_novel, n_base, dim, grad_base_weight);
    return launch_status();
}

extern "C" int subreg_loop_state_init(subreg_loop_state* state, void* stream) {
    SUBREG_CHECK_ARG(state);
    hipLaunchKernelGGL(loop_state_init_kernel, 1, 64, 0, (hipStream_t)stream, state);
    return launch_status();
}

extern "C" int subreg_finetune_step(const subreg_step_desc* d, void* stream) {
    SUBREG_CHECK_ARG(d && d->feat && d->labels && d->weight && d->momentum_buf && d->state && d->dlogits && d->rowloss &&
                     d->rowcorrect && d->norms && d->rowl1 && d->losses && d->train_acc);
    SUBREG_CHECK_ARG(d->n_support > 0 && d->n_memory >= 0 && d->n_classes > 0 && d->n_classes <= MAX_CLS && d->dim > 0);
    SUBREG_CHECK_ARG(d->n_support + d->n_memory <= 2048 && d->n_base <= MAX_BASE);
    SUBREG_CHECK_ARG(!d->use_base_reg || d->w_base);
    SUBREG_CHECK_ARG(!d->use_prev_reg || (d->w_prev && d->n_prev > 0));
    SUBREG_CHECK_ARG(!d->use_pull || d->basis || d->pull_target);
    StepArgs a;
    a.feat = d->feat; a.labels = d->labels; a.Bs = d->n_support; a.Bm = d->n_memory; a.N = d->n_classes; a.D = d->dim;
    a.W = d->weight; a.mom = d->momentum_buf; a.Wbase = d->w_base; a.Wprev = d->w_prev; a.Q = d->basis; a.target = d->pull_target;
    a.n_base = d->n_base; a.n_prev = d->n_prev; a.n_old = d->n_old;
    a.lr = d->lr; a.momentum = d->momentum; a.wd = d->weight_decay;
    a.lmbd_base = d->lmbd_base; a.lmbd_prev = d->lmbd_prev; a.pull = d->pull;
    a.use_base = d->use_base_reg; a.use_prev = d->use_prev_reg; a.use_pull = d->use_pull;
    a.dlogits = d->dlogits; a.rowloss = d->rowloss; a.rowcorrect = d->rowcorrect; a.norms = d->norms; a.rowl1 = d->rowl1;
    a.st = d->state; a.losses = d->losses; a.train_acc = d->train_acc;
    a.max_epochs = d->max_epochs; a.min_epochs = d->min_epochs; a.stable_epochs = d->stable_epochs;
    a.stable_mode = d->stable_mode; a.target_loss = d->target_loss; a.eps = d->convergence_eps;
    a.adam = d->adam; a.beta1 = d->beta1; a.beta2 = d->beta2; a.adam_eps = d->adam_eps; a.mom2 = d->exp_avg_sq;
    SUBREG_CHECK_ARG(!d->adam || d->exp_avg_sq);
    a.bias = d->bias; a.bmom = d->bias_momentum_buf; a.bmom2 = d->bias_exp_avg_sq; a.bias_base = d->bias_base;
    SUBREG_CHECK_ARG(!d->bias || (d->bias_momentum_buf && (!d->adam || d->bias_exp_avg_sq) && (!d->use_base_reg || d->bias_base)));
    // reglossnovel indexes the 1-D bias with two indices (resnet_language.py:238): the reference raises there, so there is
    // nothing to compute - the host mirror raises the same IndexError before it gets here
    SUBREG_CHECK_ARG(!(d->bias && d->use_prev_reg));
    hipStream_t s = (hipStream_t)stream;
    const int Bt = a.Bs + a.Bm;
    // These launches are latency-bound (one short dependent chain per wave): more waves per block shorten the chains without
    // changing any per-element arithmetic - 8 waves share a row's classes (4 passes of 4 classes instead of 7 at 100 classes),
    // and one thread per feature dimension in the update (640 threads: one pass over the support rows instead of 2.5).
    hipLaunchKernelGGL(step_rows_kernel, Bt + 2, 512, 0, s, a);
    int ut = (a.D + 63) / 64 * 64;
    ut = ut < 256 ? 256 : (ut > 1024 ? 1024 : ut);
    hipLaunchKernelGGL(step_update_kernel, a.N, ut, 0, s, a);
    hipLaunchKernelGGL(step_finish_kernel, 1, 256, 0, s, a);
    return launch_status();
}

extern "C" int subreg_validate(const float* feat, const long long* labels, const float* weight, const float* bias, int B, int N, int D,
                               subreg_loop_state* state, int* correct, int set_index, int n_sets_max, int mark_done,
                               void* stream) {
    SUBREG_CHECK_ARG(feat && labels && weight && correct && B > 0 && N > 0 && N <= MAX_CLS && D > 0);
    SUBREG_CHECK_ARG(set_index >= 0 && set_index < n_sets_max);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(validate_kernel, B, 512, 0, s, feat, labels, weight, bias, N, D, state, correct, set_index, n_sets_max, mark_done);
    if (mark_done && state) hipLaunchKernelGGL(validate_mark_kernel, 1, 64, 0, s, state);
    return launch_status();
}
