// lpbox_policy_f32_kernels.hip -- the early-fixing policy's attention encoder in FLOAT32 on the matrix cores.
//
// The reference evaluates GraphAttentionEncoder (LP/mha.py:202-249) in float32.  The fused fp16 kernel (lpbox_policy_kernels.hip) is the fast
// path and the plain-FMA kernel there (policy_f32_kernel, one workgroup per variable, 4.2 MB of weights per variable from L2) is the
// fp32 check; this file is the reference's arithmetic AT USABLE SPEED: the same fused structure on v_mfma_f32_16x16x4_f32 -- f32 in, f32
// accumulate, bit for bit a k-ordered fmaf chain (no reduced-precision step anywhere), 1/16 of the fp16 MFMA rate.
//
// One workgroup of 4 wavefronts (one per SIMD, 512 registers each) owns 80 tokens = 4 LP variables x 20 tokens or 16 SEG variables x 5:
//   * residual stream H (80 x 128, f32) in registers in the accumulator layout of the 16x16 tiles: wave w owns all 80 tokens x features
//     [32 w, 32 w + 32) = 5 x 2 tiles (40 VGPRs);
//   * f32 images of the GEMM inputs in LDS (H, attention output, Q / K / V of four heads, one 128-wide chunk of the FF hidden layer),
//     rows of 136 floats: a stride of 32 bytes mod 256 puts the 16 lanes of every ds_read_b128 group on 16 different 16-byte slots;
//   * GEMMs transposed (weights = A operand, activations = B operand): a lane ends up with 4 consecutive features of one token, every
//     epilogue store is 16 bytes.  One ds_read_b128 of an activation row feeds FOUR k-steps (the lane's four floats are k = 4 kq + s,
//     s = 0..3; the weight fragments are packed in the same k order), i.e. 8 MFMAs of 32 cycles each per LDS read: MFMA-bound by design;
//   * weights in fragment order from L2 (packed once on the host: lpbox_hip/policy.py), the next GEMM's fragments requested before the
//     current GEMM;
//   * attention per (variable, head, query) on the VALU from the f32 Q / K / V images (6 % of the GEMM time at this MFMA rate).
// Input straight from the solver's fp64 x_iters buffer; output = flattened f32 activations (rows x tokens*128) for the MLP head.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpbox_policy.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FM = 80;                 // tokens per workgroup
constexpr int FT = 256;                // threads
constexpr int E = 128;
constexpr int LDF = 136;               // floats per row of a 128-wide image (544 B = 32 mod 256: conflict-free fragment reads)
constexpr int LDH = 68;                // floats per row of a 64-wide image (Q, K, V of four heads)
constexpr int MT = 5;                  // 16-token tiles per wave

struct LdsF {
    float h[FM * LDF];                 // residual stream (B operand of the QKV and FF-up GEMMs)
    float ao[FM * LDF];                // attention output (B operand of the output projection); start of kernel: staging of x
    union {
        struct { float q[FM * LDH], k[FM * LDH], v[FM * LDH]; } a;
        float ff[FM * LDF];            // one 128-wide chunk of relu(FF-up)
    } u;
};
static_assert(sizeof(LdsF) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// weight fragments of one GEMM for this wave: 8 k-blocks of 16 x NT feature tiles, one float4 per lane each (element s: k = 16 kb + 4 kq + s)
template <int NT>
struct WF { f32x4 v[8][NT]; };

// fragment (tile, k-block) at (tile * 8 + kb) * 64 + lane
template <int NT>
__device__ __forceinline__ void load_wf(WF<NT> &w, const f32x4 *wp, int tile0, int tstride, int lane) {
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int kb = 0; kb < 8; kb++) w.v[kb][nt] = wp[((size_t)(tile0 + nt * tstride) * 8 + kb) * 64 + lane];
}

// acc[mt][nt] += (W^T)[this wave's NT feature tiles][0,128) * (ACT^T)[0,128)[all 80 tokens]; element r of acc[mt][nt] is token
// 16 mt + (lane & 15), feature 16 tile + 4 (lane >> 4) + r.  Activation fragments of k-block kb + 1 are requested before the MFMAs of kb.
template <int NT>
__device__ __forceinline__ void gemm_f32(const float *act, int lda, const WF<NT> &w, f32x4 (&acc)[MT][NT], int lane) {
    const float *arow = act + (size_t)(lane & 15) * lda + 4 * (lane >> 4);
    f32x4 a[2][MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) a[0][mt] = *(const f32x4 *)(arow + (size_t)mt * 16 * lda);
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        if (kb < 7) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) a[(kb + 1) & 1][mt] = *(const f32x4 *)(arow + (size_t)mt * 16 * lda + (kb + 1) * 16);
        }
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma4(w.v[kb][nt][s], a[kb & 1][mt][s], acc[mt][nt]);
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int TOK>
__global__ void __launch_bounds__(FT) policy_body_f32_kernel(PolicyArgs pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsF &S = *reinterpret_cast<LdsF *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l15 = lane & 15, g4 = (lane >> 4) * 4;
    constexpr int VARS = FM / TOK;
    const long var0 = (long)blockIdx.x * VARS;
    const int nvar = (int)min((long)VARS, pa.rows - var0);
    const f32x4 *wbase = reinterpret_cast<const f32x4 *>(pa.weights);

    WF<3> w3a, w3b;
    WF<2> w2a, w2b;
    // layer 0, heads 0-3: this wave's Q, K and V tile (head wn): tiles wn, 4 + wn, 8 + wn of the 12
    load_wf<3>(w3a, wbase, wn, 4, lane);

    // ---- stage x (fp64 in the solver's buffer) as float [80][5] in the `ao` region ----
    float *xs = S.ao;
    for (int e = tid; e < FM * 5; e += FT) {
        const int tok = e / 5, c = e - tok * 5;
        const int v = tok / TOK, t = tok - v * TOK;
        float val = 0.f;
        if (v < nvar) val = (float)pa.x[pa.row_off[var0 + v] + (long)t * pa.tok_stride + c];
        xs[e] = val;
    }
    __syncthreads();

    // ---- embedding: H = x W_in + (position code W_pos + bias), straight into the accumulator layout ----
    f32x4 H[MT][2];
    {
        const float *win = pa.consts + POLICY_OFF_WIN, *bin = pa.consts + POLICY_OFF_BIN;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int f0 = (wn * 2 + nt) * 16 + g4;
            f32x4 w[5];
#pragma unroll
            for (int c = 0; c < 5; c++) w[c] = *(const f32x4 *)(win + c * E + f0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int tok = mt * 16 + l15;
                f32x4 a = *(const f32x4 *)(bin + (tok % TOK) * E + f0);
#pragma unroll
                for (int c = 0; c < 5; c++) a += xs[tok * 5 + c] * w[c];
                H[mt][nt] = a;
            }
        }
    }
    __syncthreads();                                  // xs (aliasing ao) fully read
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++) *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + (wn * 2 + nt) * 16 + g4) = H[mt][nt];
    __syncthreads();

    constexpr int FRAGS_PER_LAYER = 96 * 2 + 64 + 4 * 64 + 4 * 64;     // float4 fragments x 64 lanes each
#pragma unroll
    for (int layer = 0; layer < 2; layer++) {
        const f32x4 *wl = wbase + (size_t)layer * FRAGS_PER_LAYER * 64;
        const float *cl = pa.consts + POLICY_OFF_LAYER(TOK) + layer * POLICY_LAYER_CONSTS;

        // ================= self-attention, four heads at a time =================
#pragma unroll
        for (int half = 0; half < 2; half++) {
            {
                f32x4 acc[MT][3];
                zero_acc<3>(acc);
                if (half == 0) { load_wf<3>(w3b, wl + (size_t)96 * 64, wn, 4, lane); gemm_f32<3>(S.h, LDF, w3a, acc, lane); }
                else           { load_wf<2>(w2a, wl + (size_t)192 * 64, wn * 2, 1, lane); gemm_f32<3>(S.h, LDF, w3b, acc, lane); }
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int tok = mt * 16 + l15;
                    *(f32x4 *)(S.u.a.q + tok * LDH + wn * 16 + g4) = acc[mt][0];
                    *(f32x4 *)(S.u.a.k + tok * LDH + wn * 16 + g4) = acc[mt][1];
                    *(f32x4 *)(S.u.a.v + tok * LDH + wn * 16 + g4) = acc[mt][2];
                }
            }
            __syncthreads();
            // ---- per (variable, head, query): softmax(q . k) v  (1/sqrt(16) folded into W_q on the host, exact; mha.py:42, :86-104) ----
            for (int p = tid; p < 4 * FM; p += FT) {
                const int v = p / (4 * TOK), rem = p - v * (4 * TOK);
                const int hh = rem / TOK, qi = rem - hh * TOK;
                const int tok0 = v * TOK;
                f32x4 q[4];
#pragma unroll
                for (int e = 0; e < 4; e++) q[e] = *(const f32x4 *)(S.u.a.q + (tok0 + qi) * LDH + hh * 16 + 4 * e);
                float s[TOK], mx = -3.0e38f;
#pragma unroll
                for (int j = 0; j < TOK; j++) {
                    float d = 0.f;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const f32x4 kv = *(const f32x4 *)(S.u.a.k + (tok0 + j) * LDH + hh * 16 + 4 * e);
                        d = __builtin_fmaf(q[e][0], kv[0], d); d = __builtin_fmaf(q[e][1], kv[1], d);       // (the library is built with
                        d = __builtin_fmaf(q[e][2], kv[2], d); d = __builtin_fmaf(q[e][3], kv[3], d);       //  -ffp-contract=off for the solver's sake)
                    }
                    s[j] = d;
                    mx = fmaxf(mx, d);
                }
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < TOK; j++) { s[j] = __expf(s[j] - mx); sum += s[j]; }      // v_exp_f32 on (x - max) log2(e): argument <= 0, 2 ulp
                const float inv = 1.f / sum;
                f32x4 o[4];
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < TOK; j++) {
                    const float pj = s[j] * inv;
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const f32x4 vv = *(const f32x4 *)(S.u.a.v + (tok0 + j) * LDH + hh * 16 + 4 * e);
                        o[e] = f32x4{__builtin_fmaf(pj, vv[0], o[e][0]), __builtin_fmaf(pj, vv[1], o[e][1]), __builtin_fmaf(pj, vv[2], o[e][2]),
                                     __builtin_fmaf(pj, vv[3], o[e][3])};
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) *(f32x4 *)(S.ao + (tok0 + qi) * LDF + (half * 4 + hh) * 16 + 4 * e) = o[e];
            }
            __syncthreads();
        }

        // ================= output projection + residual + BatchNorm (eval) =================
        {
            f32x4 acc[MT][2];
            zero_acc<2>(acc);
            load_wf<2>(w2b, wl + (size_t)256 * 64, wn * 2, 1, lane);              // FF-up chunk 0
            gemm_f32<2>(S.ao, LDF, w2a, acc, lane);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int f0 = (wn * 2 + nt) * 16 + g4;
                const f32x4 s1 = *(const f32x4 *)(cl + POLICY_LC_S1 + f0), t1 = *(const f32x4 *)(cl + POLICY_LC_T1 + f0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 hv = (H[mt][nt] + acc[mt][nt]) * s1 + t1;
                    H[mt][nt] = hv;
                    *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + f0) = hv;
                }
            }
        }
        __syncthreads();

        // ================= feed-forward 128 -> 512 -> 128, hidden layer in four 128-wide chunks =================
        {
            f32x4 acc2[MT][2];
            zero_acc<2>(acc2);
#pragma unroll
            for (int c = 0; c < 4; c++) {
                {
                    f32x4 acc[MT][2];
                    zero_acc<2>(acc);
                    load_wf<2>(w2a, wl + (size_t)(512 + c * 64) * 64, wn * 2, 1, lane);     // FF-down chunk c
                    gemm_f32<2>(S.h, LDF, w2b, acc, lane);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const int f0 = (wn * 2 + nt) * 16 + g4;
                        const f32x4 b1 = *(const f32x4 *)(cl + POLICY_LC_B1 + c * E + f0);
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {
                            const f32x4 hv = acc[mt][nt] + b1;
                            *(f32x4 *)(S.u.ff + (mt * 16 + l15) * LDF + f0) = f32x4{fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f)};
                        }
                    }
                }
                __syncthreads();
                if (c < 3) load_wf<2>(w2b, wl + (size_t)(256 + (c + 1) * 64) * 64, wn * 2, 1, lane);               // next FF-up chunk
                else if (layer == 0) load_wf<3>(w3a, wbase + (size_t)FRAGS_PER_LAYER * 64, wn, 4, lane);           // next layer's Q|K|V
                gemm_f32<2>(S.u.ff, LDF, w2a, acc2, lane);
                __syncthreads();
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int f0 = (wn * 2 + nt) * 16 + g4;
                const f32x4 b2 = *(const f32x4 *)(cl + POLICY_LC_B2 + f0), s2 = *(const f32x4 *)(cl + POLICY_LC_S2 + f0),
                            t2 = *(const f32x4 *)(cl + POLICY_LC_T2 + f0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 hv = (H[mt][nt] + acc2[mt][nt] + b2) * s2 + t2;
                    H[mt][nt] = hv;
                    *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + f0) = hv;
                }
            }
        }
        __syncthreads();
    }

    // ---- flattened activations (variables x TOK*128, f32), coalesced 16-byte stores ----
    {
        float *out = reinterpret_cast<float *>(pa.out) + var0 * (long)(TOK * E);
        const int ntok = nvar * TOK;
        for (int e = tid; e < FM * (E / 4); e += FT) {
            const int tok = e >> 5, c4 = e & 31;
            if (tok < ntok) *(f32x4 *)(out + (long)tok * E + c4 * 4) = *(const f32x4 *)(S.h + tok * LDF + c4 * 4);
        }
    }
}

}  // namespace

long policy_f32frag_floats() { return 2L * (96 * 2 + 64 + 4 * 64 + 4 * 64) * 64 * 4; }

hipError_t policy_launch_body_f32(const PolicyArgs &pa, int tokens, hipStream_t s) {
    if (pa.rows <= 0) return hipSuccess;
    if (tokens != 20 && tokens != 5) return hipErrorInvalidValue;
    const size_t lds = sizeof(LdsF);
    const long groups = (pa.rows + (FM / tokens) - 1) / (FM / tokens);
    auto go = [&](auto kernel) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(FT), lds, s, pa);
        return hipGetLastError();
    };
    return tokens == 20 ? go(policy_body_f32_kernel<20>) : go(policy_body_f32_kernel<5>);
}
