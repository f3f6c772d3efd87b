// assemble.hip -- the weight re-association of a point-deconvolution block (deconv.py) as
// one gather kernel and its adjoint.  From the reference-shaped parameters
//   Wi = inte_conv_hk.0.weight (4F, 2F, 1, T),  W2 = conv2.conv.weight (2Fo, 2F, 1, 2k),  Wf = conv_fea.0.weight (16, 2F, 1, 1)
// it builds the per-point GEMM operand Wcat (Mw x F), split by input column into the `const` part (first Fc
// columns: the channels that are constant over a sample's points) and the varying part, and conv2's dense operand
//   Wb[o, p*4F + 2c + j] = W2[o, c, k + j*P + p]      (P = k - T + 1).
// Rows of Wcat (edge features are [centre, neighbour - centre], so a tap sees W[:, F:] and the centre the difference):
//   [0, T*4F)          taps of inte_conv_hk      Wi[o, F+c, t]                     row t*4F + o
//   [.., +4F)          its centre                sum_t Wi[o, c, t] - Wi[o, F+c, t]
//   [.., +k*2Fo)       taps of conv2[..., :k]    W2[o, F+c, s]                     row s*2Fo + o
//   [.., +2Fo)         its centre                sum_{s<k} W2[o, c, s] - W2[o, F+c, s]
//   [.., +16), [.., +16)  conv_fea tap / centre  Wf[q, F+c],  Wf[q, c] - Wf[q, F+c]         (bilateral blocks only)
// In torch ops this is ~15 launches per forward and ~25 per backward of every block (squeeze / slice / permute /
// sub / sum / cat and their adjoints) on tensors of a few MB: here it is one launch each way.
#include "common.h"

struct AsmDims {
    int F, Fo, k, T, P, bilateral, Fc, Fv;
    int R1, R2, R3, R4, R5, Mw;          // row offsets of the regions above, total rows
};

__device__ __forceinline__ AsmDims asm_dims(int F, int Fo, int k, int T, int bilateral, int Fc) {
    AsmDims d;
    d.F = F; d.Fo = Fo; d.k = k; d.T = T; d.P = k - T + 1; d.bilateral = bilateral; d.Fc = Fc; d.Fv = F - Fc;
    d.R1 = T * 4 * F; d.R2 = d.R1 + 4 * F; d.R3 = d.R2 + k * 2 * Fo; d.R4 = d.R3 + 2 * Fo; d.R5 = d.R4 + 16;
    d.Mw = bilateral ? d.R5 + 16 : d.R4;
    return d;
}

__global__ void assemble_fwd_kernel(int F, int Fo, int k, int T, int bilateral, int Fc, const float *__restrict__ Wi,
                                    const float *__restrict__ W2, const float *__restrict__ Wf, float *__restrict__ WcatC,
                                    float *__restrict__ WcatV, float *__restrict__ Wb) {
    const AsmDims d = asm_dims(F, Fo, k, T, bilateral, Fc);
    const long long ncat = (long long)d.Mw * F, nb = (long long)2 * Fo * d.P * 4 * F;
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ncat) {
        const int r = (int)(e / F), c = (int)(e % F);
        float v;
        if (r < d.R1) {
            const int t = r / (4 * F), o = r % (4 * F);
            v = Wi[((size_t)o * 2 * F + F + c) * T + t];
        } else if (r < d.R2) {
            const int o = r - d.R1;
            const float *a = Wi + ((size_t)o * 2 * F + c) * T, *b = Wi + ((size_t)o * 2 * F + F + c) * T;
            v = 0.f;
            for (int t = 0; t < T; ++t) v += a[t] - b[t];
        } else if (r < d.R3) {
            const int s = (r - d.R2) / (2 * Fo), o = (r - d.R2) % (2 * Fo);
            v = W2[((size_t)o * 2 * F + F + c) * 2 * k + s];
        } else if (r < d.R4) {
            const int o = r - d.R3;
            const float *a = W2 + ((size_t)o * 2 * F + c) * 2 * k, *b = W2 + ((size_t)o * 2 * F + F + c) * 2 * k;
            v = 0.f;
            for (int s = 0; s < k; ++s) v += a[s] - b[s];
        } else if (r < d.R5) {
            v = Wf[(size_t)(r - d.R4) * 2 * F + F + c];
        } else {
            const int q = r - d.R5;
            v = Wf[(size_t)q * 2 * F + c] - Wf[(size_t)q * 2 * F + F + c];
        }
        if (c < Fc) WcatC[(size_t)r * Fc + c] = v;
        else WcatV[(size_t)r * d.Fv + (c - Fc)] = v;
    } else if (e < ncat + nb) {
        const long long e2 = e - ncat;
        const int row = d.P * 4 * F;
        const int o = (int)(e2 / row), rem = (int)(e2 % row);
        const int p = rem / (4 * F), cj = rem % (4 * F), c = cj >> 1, j = cj & 1;
        Wb[e2] = W2[((size_t)o * 2 * F + c) * 2 * k + k + j * d.P + p];
    }
}

__device__ __forceinline__ float asm_g(const AsmDims &d, const float *__restrict__ gC, const float *__restrict__ gV, int r,
                                       int c) {
    if (c < d.Fc) return gC ? gC[(size_t)r * d.Fc + c] : 0.f;
    return gV ? gV[(size_t)r * d.Fv + (c - d.Fc)] : 0.f;
}

__global__ void assemble_bwd_kernel(int F, int Fo, int k, int T, int bilateral, int Fc, const float *__restrict__ gC,
                                    const float *__restrict__ gV, const float *__restrict__ gB, float *__restrict__ dWi,
                                    float *__restrict__ dW2, float *__restrict__ dWf) {
    const AsmDims d = asm_dims(F, Fo, k, T, bilateral, Fc);
    const long long ni = (long long)4 * F * 2 * F * T, n2 = (long long)2 * Fo * 2 * F * 2 * k, nf = bilateral ? 16LL * 2 * F : 0;
    long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < ni) {
        const int t = (int)(e % T), cp = (int)((e / T) % (2 * F)), o = (int)(e / ((long long)T * 2 * F));
        dWi[e] = cp >= F ? asm_g(d, gC, gV, t * 4 * F + o, cp - F) - asm_g(d, gC, gV, d.R1 + o, cp - F)
                         : asm_g(d, gC, gV, d.R1 + o, cp);
        return;
    }
    e -= ni;
    if (e < n2) {
        const int s = (int)(e % (2 * k)), cp = (int)((e / (2 * k)) % (2 * F)), o = (int)(e / ((long long)2 * k * 2 * F));
        float v;
        if (s < k) {
            v = cp >= F ? asm_g(d, gC, gV, d.R2 + s * 2 * Fo + o, cp - F) - asm_g(d, gC, gV, d.R3 + o, cp - F)
                        : asm_g(d, gC, gV, d.R3 + o, cp);
        } else {
            const int j = (s - k) / d.P, p = (s - k) % d.P;
            v = gB ? gB[(size_t)o * d.P * 4 * F + (size_t)p * 4 * F + cp * 2 + j] : 0.f;
        }
        dW2[e] = v;
        return;
    }
    e -= n2;
    if (e < nf) {
        const int cp = (int)(e % (2 * F)), q = (int)(e / (2 * F));
        dWf[e] = cp >= F ? asm_g(d, gC, gV, d.R4 + q, cp - F) - asm_g(d, gC, gV, d.R5 + q, cp - F)
                         : asm_g(d, gC, gV, d.R5 + q, cp);
    }
}

static bool asm_ok(int F, int Fo, int k, int T, int Fc) {
    return F >= 1 && Fo >= 1 && k >= 2 && T >= 1 && T <= k && Fc >= 0 && Fc < F;
}

// Wi (4F,2F,T), W2 (2Fo,2F,2k), Wf (16,2F) or NULL -> WcatC (Mw,Fc) (NULL when fc == 0), WcatV (Mw,F-fc), Wb (2Fo, P*4F);
// Mw = T*4F + 4F + k*2Fo + 2Fo (+ 32 when Wf is given), P = k - T + 1.
extern "C" int pdgn_deconv_assemble(int F, int Fo, int k, int T, int fc, const float *Wi, const float *W2,
                                    const float *Wf, float *WcatC, float *WcatV, float *Wb, pdgn_stream_t stream) {
    if (!asm_ok(F, Fo, k, T, fc) || (fc > 0 && !WcatC)) return PDGN_ERR_INVALID;
    const int bilateral = Wf != nullptr;
    const long long Mw = (long long)T * 4 * F + 4 * F + (long long)k * 2 * Fo + 2 * Fo + (bilateral ? 32 : 0);
    const long long total = Mw * F + (long long)2 * Fo * (k - T + 1) * 4 * F;
    hipLaunchKernelGGL(assemble_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, F, Fo, k, T, bilateral, fc,
                       Wi, W2, Wf, WcatC, WcatV, Wb);
    return pdgn_launch_status();
}

// Adjoint: gradients of (WcatC, WcatV, Wb) (any may be NULL = zero) -> dWi, dW2, dWf (dWf NULL for plain blocks).
extern "C" int pdgn_deconv_assemble_backward(int F, int Fo, int k, int T, int fc, const float *gWcatC,
                                             const float *gWcatV, const float *gWb, float *dWi, float *dW2, float *dWf,
                                             pdgn_stream_t stream) {
    if (!asm_ok(F, Fo, k, T, fc)) return PDGN_ERR_INVALID;
    const int bilateral = dWf != nullptr;
    const long long total = (long long)4 * F * 2 * F * T + (long long)2 * Fo * 2 * F * 2 * k + (bilateral ? 16LL * 2 * F : 0);
    hipLaunchKernelGGL(assemble_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, F, Fo, k, T, bilateral, fc,
                       gWcatC, gWcatV, gWb, dWi, dW2, dWf);
    return pdgn_launch_status();
}

// ---------------------------------------------------------------------------- per-sample bias of the gather-sums
// With the constant-channel split every gather-sum spec i = (T_i, C_i, off_i, offc_i) gets a
// per-sample bias   bb[b, o_i + c] = bias_i[c] + Yc[b, offc_i + c] + sum_{t < T_i} Yc[b, off_i + t*C_i + c]
// (Yc (B, ldy) = const @ WcatC^T; o_i = packed column offset).  One launch for all specs, one for the adjoint:
//   dYc[b, offc_i + c] = dYc[b, off_i + t*C_i + c] = g[b, o_i + c],   dbias_i[c] = sum_b g[b, o_i + c].
#define SB_MAXSPEC 4
struct SbSpecs {
    int n, T[SB_MAXSPEC], C[SB_MAXSPEC], off[SB_MAXSPEC], offc[SB_MAXSPEC], o[SB_MAXSPEC];
    const float *bias[SB_MAXSPEC];
    float *dbias[SB_MAXSPEC];
};

__global__ void sample_bias_fwd_kernel(int B, int ldy, int ctot, SbSpecs sp, const float *__restrict__ Yc,
                                       float *__restrict__ bb) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= B * ctot) return;
    const int b = e / ctot, col = e % ctot;
    int i = 0;
    while (i + 1 < sp.n && col >= sp.o[i + 1]) ++i;
    const int c = col - sp.o[i];
    const float *row = Yc + (size_t)b * ldy;
    float v = (sp.bias[i] ? sp.bias[i][c] : 0.f) + row[sp.offc[i] + c];
    for (int t = 0; t < sp.T[i]; ++t) v += row[sp.off[i] + t * sp.C[i] + c];
    bb[e] = v;
}

__global__ void sample_bias_bwd_kernel(int B, int ldy, int ctot, SbSpecs sp, const float *__restrict__ g,
                                       float *__restrict__ dYc) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < B * ctot) {
        const int b = e / ctot, col = e % ctot;
        int i = 0;
        while (i + 1 < sp.n && col >= sp.o[i + 1]) ++i;
        const int c = col - sp.o[i];
        const float v = g[e];
        float *row = dYc + (size_t)b * ldy;
        row[sp.offc[i] + c] = v;
        for (int t = 0; t < sp.T[i]; ++t) row[sp.off[i] + t * sp.C[i] + c] = v;
    } else if (e < B * ctot + ctot) {
        const int col = e - B * ctot;
        int i = 0;
        while (i + 1 < sp.n && col >= sp.o[i + 1]) ++i;
        if (sp.dbias[i]) {
            float s = 0.f;
            for (int b = 0; b < B; ++b) s += g[(size_t)b * ctot + col];
            sp.dbias[i][col - sp.o[i]] = s;
        }
    }
}

static bool sb_fill(SbSpecs *sp, int nspec, const int *T, const int *C, const int *off, const int *offc, int *ctot) {
    if (nspec < 1 || nspec > SB_MAXSPEC) return false;
    sp->n = nspec;
    int o = 0;
    for (int i = 0; i < nspec; ++i) {
        if (T[i] < 1 || C[i] < 1 || off[i] < 0 || offc[i] < 0) return false;
        sp->T[i] = T[i]; sp->C[i] = C[i]; sp->off[i] = off[i]; sp->offc[i] = offc[i]; sp->o[i] = o;
        sp->bias[i] = nullptr; sp->dbias[i] = nullptr;
        o += C[i];
    }
    *ctot = o;
    return true;
}

// bb (B, sum C_i) packed per-sample biases; bias[i] may be NULL.  The columns written / read in Yc must lie in [0, ldy).
extern "C" int pdgn_sample_bias(int b, int ldy, int nspec, const int *T, const int *C, const int *off, const int *offc,
                                const float *const *bias, const float *Yc, float *bb, pdgn_stream_t stream) {
    SbSpecs sp;
    int ctot;
    if (b < 1 || !sb_fill(&sp, nspec, T, C, off, offc, &ctot)) return PDGN_ERR_INVALID;
    for (int i = 0; i < nspec; ++i) sp.bias[i] = bias[i];
    hipLaunchKernelGGL(sample_bias_fwd_kernel, dim3(cdiv((long long)b * ctot, 256)), dim3(256), 0, (hipStream_t)stream, b, ldy, ctot,
                       sp, Yc, bb);
    return pdgn_launch_status();
}

// g (B, sum C_i) -> dYc (B, ldy) (every column covered by the specs is written; the caller zero-fills if they do not
// tile [0, ldy)), dbias[i] (C_i) or NULL.
extern "C" int pdgn_sample_bias_backward(int b, int ldy, int nspec, const int *T, const int *C, const int *off,
                                         const int *offc, const float *g, float *dYc, float *const *dbias,
                                         pdgn_stream_t stream) {
    SbSpecs sp;
    int ctot;
    if (b < 1 || !sb_fill(&sp, nspec, T, C, off, offc, &ctot)) return PDGN_ERR_INVALID;
    for (int i = 0; i < nspec; ++i) sp.dbias[i] = dbias[i];
    hipLaunchKernelGGL(sample_bias_bwd_kernel, dim3(cdiv((long long)b * ctot + ctot, 256)), dim3(256), 0, (hipStream_t)stream, b, ldy,
                       ctot, sp, g, dYc);
    return pdgn_launch_status();
}
