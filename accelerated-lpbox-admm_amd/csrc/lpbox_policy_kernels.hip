// lpbox_policy_kernels.hip -- the early-fixing policy's attention encoder as ONE fused gfx950 kernel.
//
// Reference network: GraphAttentionEncoder (LP/mha.py:202-249): tokens of 5 iterates + position code -> Linear 10->128 ->
// 2 x [8-head self-attention (mha.py:20-122) + residual + BatchNorm1d + FF 128-512-128 + residual + BatchNorm1d]
// (mha.py:157-183) -> flatten -> MLP head (mha.py:185-199).  This kernel is everything up to the flatten, evaluated in
// eval mode (BatchNorm = per-channel affine map, folded on the host); the head is three small GEMMs on the flattened output.
//
// Mapping to the machine.  One workgroup (8 wavefronts) owns 160 tokens = 8 variables x 20 tokens (LP) or 32 x 5 (SEG).
// Every activation of those tokens lives on the chip for the whole network:
//   * the residual stream H (160 x 128, fp32) stays in the accumulator layout of the 16x16 MFMA tiles that produce it:
//     wave (wm, wn) of a 2 x 4 wave grid owns rows [80 wm, 80 wm + 80) x columns [32 wn, 32 wn + 32) = 5 x 2 tiles (40 VGPRs);
//   * fp16 images of the GEMM inputs (H, attention output, Q/K/V of four heads, one 128-wide chunk of the FF hidden layer)
//     sit in LDS, rows padded by 16 B so that the 16-byte fragment reads of 16 consecutive rows hit distinct banks;
//   * weights stream from L2 in the exact per-lane fragment order of v_mfma_f32_16x16x32_f16 (packed once on the host), one
//     coalesced 1 KiB read per fragment, each reused over the wave's 5 row tiles.
// GEMMs: v_mfma_f32_16x16x32_f16, fp32 accumulate.  Attention: per (variable, head) S^T = K Q^T is ONE 32x32x16 MFMA (head
// dimension 16 = its K); with the keys in the accumulator registers the softmax is in-lane (plus one exchange between the two
// lane halves), and the normalised probabilities feed P V directly as the A operand of the next MFMA (accumulator-as-operand,
// no LDS round trip).  Inputs are read straight from the solver's x_iters buffer (fp64), the output is the flattened fp16
// activation (rows x tokens*128) for the head.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lpbox_policy.h"

namespace {

using f16 = _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PT = POLICY_THREADS;     // 512
constexpr int PM = POLICY_TOKENS_PER_WG;   // 160
constexpr int E = 128;
constexpr int LDA = 136;               // halves per row of a 128-wide LDS image
constexpr int LDQ = 72;                // halves per row of a 64-wide LDS image (Q, K, V of four heads)
constexpr int QROWS = 176;             // 160 + 16 zeroed pad rows: the 32-row MFMA operands of the last variable stay in bounds
constexpr int MT = 5;                  // 16-row tiles per wave (2 x 4 wave grid over 10 x N/16 tiles)

struct Lds {
    f16 h[PM * LDA];          // fp16 image of the residual stream (A operand of the QKV and FF-up GEMMs)
    f16 ao[PM * LDA];         // attention output (A operand of the output projection); start of kernel: staging of x
    union {
        struct { f16 q[QROWS * LDQ], k[QROWS * LDQ], v[QROWS * LDQ]; } a;
        f16 ff[PM * LDA];     // one 128-wide chunk of relu(FF-up)
    } u;
};
static_assert(sizeof(Lds) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// acc[mt][nt] += A[rows of this wave's 5 row tiles][0, 32 KS) * W[:, this wave's NT column tiles]
// A: LDS fp16 image with row stride lda (halves).  wp: packed fragments, fragment (n-tile, k-step) at (ntile*KS + ks)*64 + lane.
template <int NT, int KS>
__device__ __forceinline__ void gemm_tiles(const f16 *A, int lda, int row0, const f16x8 *wp, int ntile0, f32x4 (&acc)[MT][NT], int lane) {
    const f16 *arow = A + (size_t)(row0 + (lane & 15)) * lda + 8 * (lane >> 4);
    f16x8 b[NT], bn[NT];
#pragma unroll
    for (int nt = 0; nt < NT; nt++) b[nt] = wp[((size_t)(ntile0 + nt) * KS) * 64 + lane];
#pragma unroll
    for (int ks = 0; ks < KS; ks++) {
        if (ks + 1 < KS) {
#pragma unroll
            for (int nt = 0; nt < NT; nt++) bn[nt] = wp[((size_t)(ntile0 + nt) * KS + ks + 1) * 64 + lane];
        }
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const f16x8 a = *(const f16x8 *)(arow + (size_t)mt * 16 * lda + ks * 32);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) acc[mt][nt] = mfma16(a, b[nt], acc[mt][nt]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; nt++) b[nt] = bn[nt];
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// element (mt, nt, r) of a wave's accumulator tiles: row = row0 + 16 mt + 4 (lane>>4) + r, column = 16 (ntile0 + nt) + (lane & 15)

template <int TOK>
__global__ void __launch_bounds__(PT) policy_body_kernel(PolicyArgs pa) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lds &S = *reinterpret_cast<Lds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int row0 = wm * 80;                       // first token row of this wave's tiles
    constexpr int VARS = PM / TOK;                  // variables per workgroup
    const long var0 = (long)blockIdx.x * VARS;
    const int nvar = (int)min((long)VARS, pa.rows - var0);

    // ---- stage x (fp64 in the solver's buffer) as float [160][5] in the `ao` region; zero the pad rows of Q/K/V ----
    float *xs = reinterpret_cast<float *>(S.ao);
    for (int e = tid; e < PM * 5; e += PT) {
        const int tok = e / 5, c = e - tok * 5;
        const int v = tok / TOK, t = tok - v * TOK;
        float val = 0.f;
        if (v < nvar) val = (float)pa.x[pa.row_off[var0 + v] + (long)t * pa.tok_stride + c];
        xs[e] = val;
    }
    for (int e = tid; e < (QROWS - PM) * LDQ; e += PT) {
        S.u.a.q[PM * LDQ + e] = (f16)0.f; S.u.a.k[PM * LDQ + e] = (f16)0.f; S.u.a.v[PM * LDQ + e] = (f16)0.f;
    }
    __syncthreads();

    // ---- embedding: H = x W_in + (position code W_pos + bias), straight into the accumulator layout ----
    f32x4 H[MT][2];
    {
        const float *win = pa.consts + POLICY_OFF_WIN, *bin = pa.consts + POLICY_OFF_BIN;
#pragma unroll
        for (int nt = 0; nt < 2; nt++) {
            const int col = (wn * 2 + nt) * 16 + (lane & 15);
            float w[5];
#pragma unroll
            for (int c = 0; c < 5; c++) w[c] = win[c * E + col];
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int tok = row0 + mt * 16 + 4 * (lane >> 4) + r;
                    float a = bin[(tok % TOK) * E + col];
#pragma unroll
                    for (int c = 0; c < 5; c++) a += xs[tok * 5 + c] * w[c];
                    H[mt][nt][r] = a;
                }
        }
    }
    __syncthreads();                                  // xs (aliasing ao) fully read
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < 2; nt++)
#pragma unroll
            for (int r = 0; r < 4; r++)
                S.h[(row0 + mt * 16 + 4 * (lane >> 4) + r) * LDA + (wn * 2 + nt) * 16 + (lane & 15)] = (f16)H[mt][nt][r];
    __syncthreads();

    const f16x8 *wbase = reinterpret_cast<const f16x8 *>(pa.weights);
#pragma unroll 1
    for (int layer = 0; layer < 2; layer++) {
        const f16x8 *wl = wbase + (size_t)layer * POLICY_FRAGS_PER_LAYER * 64;
        const float *cl = pa.consts + POLICY_OFF_LAYER(TOK) + layer * POLICY_LAYER_CONSTS;

        // ================= self-attention, four heads at a time =================
#pragma unroll 1
        for (int half = 0; half < 2; half++) {
            {   // Q | K | V of heads 4 half .. 4 half + 3:  (160 x 128) x (128 x 192)
                f32x4 acc[MT][3];
                zero_acc<3>(acc);
                gemm_tiles<3, 4>(S.h, LDA, row0, wl + (size_t)half * 48 * 64, wn * 3, acc, lane);
#pragma unroll
                for (int nt = 0; nt < 3; nt++) {
                    const int gt = wn * 3 + nt;                      // 0..11: 4 tiles each of Q, K, V
                    f16 *dst = gt < 4 ? S.u.a.q : (gt < 8 ? S.u.a.k : S.u.a.v);
                    const int col = (gt & 3) * 16 + (lane & 15);
#pragma unroll
                    for (int mt = 0; mt < MT; mt++)
#pragma unroll
                        for (int r = 0; r < 4; r++)
                            dst[(row0 + mt * 16 + 4 * (lane >> 4) + r) * LDQ + col] = (f16)acc[mt][nt][r];
                }
            }
            __syncthreads();
            // ---- per (variable, head): S^T = K Q^T (keys in registers, queries on lanes), softmax over registers, O = P V ----
            {
                const int r31 = lane & 31, hf = lane >> 5;
#pragma unroll 1
                for (int pair = wave; pair < VARS * 4; pair += PT / 64) {
                    const int v = pair >> 2, hh = pair & 3;
                    const int tok0 = v * TOK;
                    const int off = (tok0 + r31) * LDQ + hh * 16 + 8 * hf;
                    const f16x8 ka = *(const f16x8 *)(S.u.a.k + off);        // A: row = key r31, k = 8 hf + e
                    const f16x8 qb = *(const f16x8 *)(S.u.a.q + off);        // B: col = query r31
                    f32x16 st;
#pragma unroll
                    for (int i = 0; i < 16; i++) st[i] = 0.f;
                    st = mfma32(ka, qb, st);                                   // st[reg]: key (reg&3) + 8 (reg>>2) + 4 hf, query r31
                    float mx = -3.0e38f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int key = (i & 3) + 8 * (i >> 2) + 4 * hf;
                        st[i] = key < TOK ? st[i] * 0.25f : -3.0e38f;          // norm factor 1/sqrt(16) (mha.py:42)
                        mx = fmaxf(mx, st[i]);
                    }
                    mx = fmaxf(mx, __shfl_xor(mx, 32));
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int key = (i & 3) + 8 * (i >> 2) + 4 * hf;
                        st[i] = key < TOK ? __expf(st[i] - mx) : 0.f;
                        sum += st[i];
                    }
                    sum += __shfl_xor(sum, 32);
                    const float inv = 1.f / sum;
                    f32x16 o;
#pragma unroll
                    for (int i = 0; i < 16; i++) o[i] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        if (16 * s < TOK) {
                            f16x8 pfrag, vfrag;
#pragma unroll
                            for (int e = 0; e < 8; e++) {
                                pfrag[e] = (f16)(st[8 * s + e] * inv);         // P^T rows 16 s + 8 (e>>2) + 4 hf + (e&3)
                                const int key = 16 * s + 8 * (e >> 2) + 4 * hf + (e & 3);
                                vfrag[e] = S.u.a.v[(tok0 + key) * LDQ + hh * 16 + (r31 & 15)];
                            }
                            o = mfma32(pfrag, vfrag, o);                       // o[reg]: query (reg&3) + 8 (reg>>2) + 4 hf, dim r31
                        }
                    }
                    if (r31 < 16) {
#pragma unroll
                        for (int i = 0; i < 16; i++) {
                            const int qi = (i & 3) + 8 * (i >> 2) + 4 * hf;
                            if (qi < TOK) S.ao[(tok0 + qi) * LDA + (half * 4 + hh) * 16 + r31] = (f16)o[i];
                        }
                    }
                }
            }
            __syncthreads();
        }

        // ================= output projection + residual + BatchNorm (eval) =================
        {
            f32x4 acc[MT][2];
            zero_acc<2>(acc);
            gemm_tiles<2, 4>(S.ao, LDA, row0, wl + (size_t)96 * 64, wn * 2, acc, lane);
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int col = (wn * 2 + nt) * 16 + (lane & 15);
                const float s1 = cl[POLICY_LC_S1 + col], t1 = cl[POLICY_LC_T1 + col];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float hv = (H[mt][nt][r] + acc[mt][nt][r]) * s1 + t1;
                        H[mt][nt][r] = hv;
                        S.h[(row0 + mt * 16 + 4 * (lane >> 4) + r) * LDA + col] = (f16)hv;
                    }
            }
        }
        __syncthreads();

        // ================= feed-forward 128 -> 512 -> 128, hidden layer in four 128-wide chunks =================
        {
            f32x4 acc2[MT][2];
            zero_acc<2>(acc2);
#pragma unroll 1
            for (int c = 0; c < 4; c++) {
                {
                    f32x4 acc[MT][2];
                    zero_acc<2>(acc);
                    gemm_tiles<2, 4>(S.h, LDA, row0, wl + (size_t)(128 + c * 32) * 64, wn * 2, acc, lane);
#pragma unroll
                    for (int nt = 0; nt < 2; nt++) {
                        const int col = (wn * 2 + nt) * 16 + (lane & 15);
                        const float b1 = cl[POLICY_LC_B1 + c * E + col];
#pragma unroll
                        for (int mt = 0; mt < MT; mt++)
#pragma unroll
                            for (int r = 0; r < 4; r++)
                                S.u.ff[(row0 + mt * 16 + 4 * (lane >> 4) + r) * LDA + col] = (f16)fmaxf(acc[mt][nt][r] + b1, 0.f);
                    }
                }
                __syncthreads();
                gemm_tiles<2, 4>(S.u.ff, LDA, row0, wl + (size_t)(256 + c * 32) * 64, wn * 2, acc2, lane);
                __syncthreads();
            }
#pragma unroll
            for (int nt = 0; nt < 2; nt++) {
                const int col = (wn * 2 + nt) * 16 + (lane & 15);
                const float b2 = cl[POLICY_LC_B2 + col], s2 = cl[POLICY_LC_S2 + col], t2 = cl[POLICY_LC_T2 + col];
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const float hv = (H[mt][nt][r] + acc2[mt][nt][r] + b2) * s2 + t2;
                        H[mt][nt][r] = hv;
                        S.h[(row0 + mt * 16 + 4 * (lane >> 4) + r) * LDA + col] = (f16)hv;
                    }
            }
        }
        __syncthreads();
        if (layer == 0) {      // the FF chunk buffer aliased the Q/K/V images: restore their zero pad rows for the next layer
            for (int e = tid; e < (QROWS - PM) * LDQ; e += PT) {
                S.u.a.q[PM * LDQ + e] = (f16)0.f; S.u.a.k[PM * LDQ + e] = (f16)0.f; S.u.a.v[PM * LDQ + e] = (f16)0.f;
            }
            // no barrier needed here: the next writers of these images (QKV epilogue) touch rows < 160 only, and the next
            // readers (attention) sit behind that epilogue's barrier
        }
    }

    // ---- flattened activations (variables x TOK*128, fp16), coalesced 16-byte stores ----
    {
        f16 *out = reinterpret_cast<f16 *>(pa.out) + var0 * (long)(TOK * E);
        const int ntok = nvar * TOK;
        for (int e = tid; e < PM * (E / 8); e += PT) {
            const int tok = e >> 4, c8 = e & 15;
            if (tok < ntok) *(f16x8 *)(out + (long)tok * E + c8 * 8) = *(const f16x8 *)(S.h + tok * LDA + c8 * 8);
        }
    }
}

}  // namespace

size_t policy_lds_bytes() { return sizeof(Lds); }

hipError_t policy_launch_body(const PolicyArgs &pa, int tokens, hipStream_t s) {
    if (pa.rows <= 0) return hipSuccess;
    const size_t lds = sizeof(Lds);
    if (tokens == 20) {
        const long groups = (pa.rows + (PM / 20) - 1) / (PM / 20);
        hipError_t e = hipFuncSetAttribute((const void *)policy_body_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(policy_body_kernel<20>, dim3((unsigned)groups), dim3(PT), lds, s, pa);
    } else if (tokens == 5) {
        const long groups = (pa.rows + (PM / 5) - 1) / (PM / 5);
        hipError_t e = hipFuncSetAttribute((const void *)policy_body_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(policy_body_kernel<5>, dim3((unsigned)groups), dim3(PT), lds, s, pa);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
