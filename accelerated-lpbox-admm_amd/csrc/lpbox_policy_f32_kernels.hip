// lpbox_policy_f32_kernels.hip -- the early-fixing policy's attention encoder in FLOAT32 on the matrix cores.
//
// The reference evaluates GraphAttentionEncoder (LP/mha.py:202-249) in float32.  The fused fp16 kernel (lpbox_policy_kernels.hip) is the fast
// path and the plain-FMA kernel there (policy_f32_kernel, one workgroup per variable, 4.2 MB of weights per variable from L2) is the
// fp32 check; this file is the reference's arithmetic AT USABLE SPEED: the same fused structure on v_mfma_f32_16x16x4_f32 -- f32 in, f32
// accumulate, bit for bit a k-ordered fmaf chain (no reduced-precision step anywhere), 1/16 of the fp16 MFMA rate.
//
// One workgroup of 8 wavefronts (two per SIMD, 168 registers each) owns 80 tokens = 4 LP variables x 20 tokens or 16 SEG variables x 5:
//   * residual stream H (80 x 128, f32) in registers in the accumulator layout of the 16x16 tiles: wave w owns all 80 tokens x features
//     [16 w, 16 w + 16) = 5 tiles (20 VGPRs);
//   * f32 images of the GEMM inputs in LDS (H, attention output, Q / K / V of four heads, one 128-wide chunk of the FF hidden layer),
//     rows of 136 floats: a stride of 32 bytes mod 256 puts the 16 lanes of every ds_read_b128 group on 16 different 16-byte slots;
//   * GEMMs transposed (weights = A operand, activations = B operand): a lane ends up with 4 consecutive features of one token, every
//     epilogue store is 16 bytes.  One ds_read_b128 of an activation row feeds FOUR k-steps (the lane's four floats are k = 4 kq + s,
//     s = 0..3; the weight fragments are packed in the same k order), i.e. 4 MFMAs of 32 cycles each per LDS read: MFMA-bound by design;
//   * the Q|K|V block of four heads has 12 tiles: pass 1 = Q (waves 0-3) and K (waves 4-7), pass 2 = V with the token tiles split between
//     the two waves of a SIMD (waves w and w + 4 share one: tiles 0-2 and 3-4), so every SIMD does 5 tile-products in each pass;
//   * weights in fragment order from L2 (packed once on the host: lpbox_hip/policy.py); two register sets alternate, the next GEMM's
//     fragments are requested before the current GEMM;
//   * attention per (variable, head, query) on the VALU from the f32 Q / K / V images: 320 items in one round of the 512 lanes, keys in
//     chunks of four with a running maximum (explicit fmaf: the library is built with -ffp-contract=off for the solver's sake).
//   (A four-wave form -- one wave per SIMD, 80 x 32 outputs each -- measured 17.6 ms against this one's 17.05 for 128 000 variables.)
// Input straight from the solver's fp64 x_iters buffer; output = flattened f32 activations (rows x tokens*128) for the MLP head.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpbox_policy.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int FM = 80;                 // tokens per workgroup
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

constexpr int FT8 = 512;

// acc[mt - M0] += ... for token tiles [M0, M1) only
template <int M0, int M1>
__device__ __forceinline__ void gemm_f32_1(const float *act, int lda, const WF<1> &w, f32x4 (&acc)[M1 - M0], int lane) {
    constexpr int N = M1 - M0;
    const float *arow = act + (size_t)(M0 * 16 + (lane & 15)) * lda + 4 * (lane >> 4);
    f32x4 a[2][N];
#pragma unroll
    for (int mt = 0; mt < N; mt++) a[0][mt] = *(const f32x4 *)(arow + (size_t)mt * 16 * lda);
#pragma unroll
    for (int kb = 0; kb < 8; kb++) {
        if (kb < 7) {
#pragma unroll
            for (int mt = 0; mt < N; mt++) a[(kb + 1) & 1][mt] = *(const f32x4 *)(arow + (size_t)mt * 16 * lda + (kb + 1) * 16);
        }
#pragma unroll
        for (int s = 0; s < 4; s++)
#pragma unroll
            for (int mt = 0; mt < N; mt++) acc[mt] = mfma4(w.v[kb][0][s], a[kb & 1][mt][s], acc[mt]);
    }
}

template <int TOK>
__global__ void __launch_bounds__(FT8) policy_body_f32_kernel8(PolicyArgs pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    LdsF &S = *reinterpret_cast<LdsF *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int l15 = lane & 15, g4 = (lane >> 4) * 4;
    const int f0 = w * 16 + g4;                      // this lane's four features of a 128-wide output
    constexpr int VARS = FM / TOK;
    const long var0 = (long)blockIdx.x * VARS;
    const int nvar = (int)min((long)VARS, pa.rows - var0);
    const f32x4 *wbase = reinterpret_cast<const f32x4 *>(pa.weights);
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};

    WF<1> wa, wb;                                    // the GEMMs alternate between the two sets: one in use, the next one's fragments in flight
    load_wf<1>(wa, wbase, w, 1, lane);               // layer 0, heads 0-3, pass 1: tile w of [Q0-3 | K0-3 | V0-3]

    float *xs = S.ao;
    for (int e = tid; e < FM * 5; e += FT8) {
        const int tok = e / 5, c = e - tok * 5;
        const int v = tok / TOK, t = tok - v * TOK;
        float val = 0.f;
        if (v < nvar) val = (float)pa.x[pa.row_off[var0 + v] + (long)t * pa.tok_stride + c];
        xs[e] = val;
    }
    __syncthreads();

    f32x4 H[MT];
    {
        const float *win = pa.consts + POLICY_OFF_WIN, *bin = pa.consts + POLICY_OFF_BIN;
        f32x4 wi[5];
#pragma unroll
        for (int c = 0; c < 5; c++) wi[c] = *(const f32x4 *)(win + c * E + f0);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int tok = mt * 16 + l15;
            f32x4 a = *(const f32x4 *)(bin + (tok % TOK) * E + f0);
#pragma unroll
            for (int c = 0; c < 5; c++) a += xs[tok * 5 + c] * wi[c];
            H[mt] = a;
        }
    }
    __syncthreads();                                  // xs (aliasing ao) fully read
#pragma unroll
    for (int mt = 0; mt < MT; mt++) *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + f0) = H[mt];
    __syncthreads();

    constexpr int FRAGS_PER_LAYER = 96 * 2 + 64 + 4 * 64 + 4 * 64;
#pragma unroll
    for (int layer = 0; layer < 2; layer++) {
        const f32x4 *wl = wbase + (size_t)layer * FRAGS_PER_LAYER * 64;
        const float *cl = pa.consts + POLICY_OFF_LAYER(TOK) + layer * POLICY_LAYER_CONSTS;

#pragma unroll
        for (int half = 0; half < 2; half++) {
            const f32x4 *wh = wl + (size_t)half * 96 * 64;
            {   // pass 1: Q (waves 0-3) and K (waves 4-7) of head w & 3; weights in wa.  Meanwhile: the V tile of pass 2 into wb
                f32x4 acc[MT] = {z4, z4, z4, z4, z4};
                load_wf<1>(wb, wh, 8 + (w & 3), 1, lane);
                gemm_f32_1<0, MT>(S.h, LDF, wa, acc, lane);
                float *dst = (w < 4 ? S.u.a.q : S.u.a.k) + (w & 3) * 16 + g4;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) *(f32x4 *)(dst + (mt * 16 + l15) * LDH) = acc[mt];
            }
            {   // pass 2: V of head w & 3, token tiles 0-2 (waves 0-3) / 3-4 (waves 4-7); weights in wb.  Meanwhile into wa: the next GEMM's tile
                if (half == 0) load_wf<1>(wa, wl + (size_t)96 * 64, w, 1, lane);            // heads 4-7, pass 1
                else           load_wf<1>(wa, wl + (size_t)192 * 64, w, 1, lane);           // output projection
                float *dst = S.u.a.v + (w & 3) * 16 + g4;
                if (w < 4) {
                    f32x4 acc[3] = {z4, z4, z4};
                    gemm_f32_1<0, 3>(S.h, LDF, wb, acc, lane);
#pragma unroll
                    for (int mt = 0; mt < 3; mt++) *(f32x4 *)(dst + (mt * 16 + l15) * LDH) = acc[mt];
                } else {
                    f32x4 acc[2] = {z4, z4};
                    gemm_f32_1<3, 5>(S.h, LDF, wb, acc, lane);
#pragma unroll
                    for (int mt = 0; mt < 2; mt++) *(f32x4 *)(dst + ((3 + mt) * 16 + l15) * LDH) = acc[mt];
                }
            }
            __syncthreads();
            // ---- per (variable, head, query): softmax(q . k) v -- 320 items, one round ----
            if (tid < 4 * FM) {
                const int p = tid;
                const int v = p / (4 * TOK), rem = p - v * (4 * TOK);
                const int hh = rem / TOK, qi = rem - hh * TOK;
                const int tok0 = v * TOK;
                f32x4 q[4];
#pragma unroll
                for (int e = 0; e < 4; e++) q[e] = *(const f32x4 *)(S.u.a.q + (tok0 + qi) * LDH + hh * 16 + 4 * e);
                // keys in chunks with a running maximum and sum (the softmax is invariant to the shift; the chunking keeps 16 + 16 + CH values
                // live instead of 20 scores and every K / V row at once: 124 spilled registers otherwise at two waves per SIMD)
                constexpr int CH = (TOK % 4 == 0) ? 4 : TOK;
                float mx = -3.0e38f, sum = 0.f;
                f32x4 o[4] = {z4, z4, z4, z4};
#pragma unroll 1
                for (int j0 = 0; j0 < TOK; j0 += CH) {
                    float sc[CH], cm = mx;
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        float d = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const f32x4 kv = *(const f32x4 *)(S.u.a.k + (tok0 + j0 + j) * LDH + hh * 16 + 4 * e);
                            d = __builtin_fmaf(q[e][0], kv[0], d); d = __builtin_fmaf(q[e][1], kv[1], d);
                            d = __builtin_fmaf(q[e][2], kv[2], d); d = __builtin_fmaf(q[e][3], kv[3], d);
                        }
                        sc[j] = d;
                        cm = fmaxf(cm, d);
                    }
                    const float scale = __expf(mx - cm);          // first chunk: exp(-3e38 - cm) = 0, and o, sum are 0 anyway
                    sum *= scale;
#pragma unroll
                    for (int e = 0; e < 4; e++) o[e] *= scale;
                    mx = cm;
#pragma unroll
                    for (int j = 0; j < CH; j++) {
                        const float pj = __expf(sc[j] - mx);
                        sum += pj;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const f32x4 vv = *(const f32x4 *)(S.u.a.v + (tok0 + j0 + j) * LDH + hh * 16 + 4 * e);
                            o[e] = f32x4{__builtin_fmaf(pj, vv[0], o[e][0]), __builtin_fmaf(pj, vv[1], o[e][1]), __builtin_fmaf(pj, vv[2], o[e][2]),
                                         __builtin_fmaf(pj, vv[3], o[e][3])};
                        }
                    }
                }
                const float inv = 1.f / sum;
#pragma unroll
                for (int e = 0; e < 4; e++) o[e] *= inv;
#pragma unroll
                for (int e = 0; e < 4; e++) *(f32x4 *)(S.ao + (tok0 + qi) * LDF + (half * 4 + hh) * 16 + 4 * e) = o[e];
            }
            __syncthreads();
        }

        // ================= output projection + residual + BatchNorm (eval): weights in wa =================
        {
            f32x4 acc[MT] = {z4, z4, z4, z4, z4};
            load_wf<1>(wb, wl + (size_t)256 * 64, w, 1, lane);                     // FF-up chunk 0
            gemm_f32_1<0, MT>(S.ao, LDF, wa, acc, lane);
            const f32x4 s1 = *(const f32x4 *)(cl + POLICY_LC_S1 + f0), t1 = *(const f32x4 *)(cl + POLICY_LC_T1 + f0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const f32x4 hv = (H[mt] + acc[mt]) * s1 + t1;
                H[mt] = hv;
                *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + f0) = hv;
            }
        }
        __syncthreads();

        // ================= feed-forward: up chunk c in wb, down chunk c in wa =================
        // The chunk buffer is double-buffered (the attention-output image is idle here): relu(FF-up) of chunk c + 1 is produced right behind
        // the FF-down product of chunk c -- one barrier per chunk, and the MFMA stream of a wave continues across the down -> up boundary.
        {
            f32x4 acc2[MT] = {z4, z4, z4, z4, z4};
            auto ff_up = [&](int c, float *dst) {
                f32x4 acc[MT] = {z4, z4, z4, z4, z4};
                gemm_f32_1<0, MT>(S.h, LDF, wb, acc, lane);
                const f32x4 b1 = *(const f32x4 *)(cl + POLICY_LC_B1 + c * E + f0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 hv = acc[mt] + b1;
                    *(f32x4 *)(dst + (mt * 16 + l15) * LDF + f0) = f32x4{fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f)};
                }
            };
            load_wf<1>(wa, wl + (size_t)512 * 64, w, 1, lane);                                  // FF-down chunk 0
            ff_up(0, S.u.ff);
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 4; c++) {
                float *cur = (c & 1) ? S.ao : S.u.ff, *nxt = (c & 1) ? S.u.ff : S.ao;
                if (c < 3) load_wf<1>(wb, wl + (size_t)(256 + (c + 1) * 64) * 64, w, 1, lane);  // next FF-up chunk
                gemm_f32_1<0, MT>(cur, LDF, wa, acc2, lane);                                    // FF-down chunk c
                if (c < 3) {
                    load_wf<1>(wa, wl + (size_t)(512 + (c + 1) * 64) * 64, w, 1, lane);         // FF-down chunk c + 1
                    ff_up(c + 1, nxt);
                    __syncthreads();
                } else if (layer == 0) load_wf<1>(wa, wbase + (size_t)FRAGS_PER_LAYER * 64, w, 1, lane);   // next layer, heads 0-3, pass 1
            }
            const f32x4 b2 = *(const f32x4 *)(cl + POLICY_LC_B2 + f0), s2 = *(const f32x4 *)(cl + POLICY_LC_S2 + f0),
                        t2 = *(const f32x4 *)(cl + POLICY_LC_T2 + f0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const f32x4 hv = (H[mt] + acc2[mt] + b2) * s2 + t2;
                H[mt] = hv;
                *(f32x4 *)(S.h + (mt * 16 + l15) * LDF + f0) = hv;
            }
        }
        __syncthreads();
    }

    {
        float *out = reinterpret_cast<float *>(pa.out) + var0 * (long)(TOK * E);
        const int ntok = nvar * TOK;
        for (int e = tid; e < FM * (E / 4); e += FT8) {
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
    auto go = [&](auto kernel, int threads) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(threads), lds, s, pa);
        return hipGetLastError();
    };
    return tokens == 20 ? go(policy_body_f32_kernel8<20>, FT8) : go(policy_body_f32_kernel8<5>, FT8);
}
