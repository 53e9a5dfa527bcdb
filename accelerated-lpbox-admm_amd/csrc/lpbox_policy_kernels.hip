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
// GEMMs: v_mfma_f32_16x16x32_f16, fp32 accumulate, computed transposed (weights = A operand, activations = B operand) so that a
// lane's four accumulator registers are four consecutive features of one token: every epilogue store is 8 bytes wide.  The
// next GEMM's weight fragments are requested before the current GEMM starts (two register sets), hiding the L2 latency.  Attention: per (variable, head) S^T = K Q^T is ONE 32x32x16 MFMA (head
// dimension 16 = its K); with the keys in the accumulator registers the softmax is in-lane (plus one exchange between the two
// lane halves), and the normalised probabilities feed P V directly as the A operand of the next MFMA (accumulator-as-operand,
// no LDS round trip).  Inputs are read straight from the solver's x_iters buffer (fp64), the output is the flattened fp16
// activation (rows x tokens*128) for the head.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>

#include "lpbox_policy.h"

// timing experiments only (tools/): -DPOL_NO_ATTN / -DPOL_NO_FF cut a phase out; results are then wrong by construction
#ifdef POL_NO_ATTN
#define POL_ATTN_PAIRS(n) 0
#else
#define POL_ATTN_PAIRS(n) (n)
#endif
#ifdef POL_NO_FF
#define POL_FF_CHUNKS 0
#else
#define POL_FF_CHUNKS 4
#endif
// finer knock-outs (timing only; tools/policy_body.py): POL_NO_QKV (no Q|K|V GEMM, no epilogue), POL_NO_QKV_EPI / POL_NO_FF_EPI (GEMM kept, LDS
// stores behind a condition that is false at run time), POL_NO_PROJ, POL_NO_OUT (no global store of the result), POL_NO_IN (no global read
// of x), POL_W_ONCE (no weight traffic: zero fragments)
#if defined(POL_NO_QKV_EPI) || defined(POL_NO_FF_EPI)
#define POL_EPI_GUARD(pa) ((pa).rows < 0)
#endif

namespace {

using f16 = _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x4 = __attribute__((ext_vector_type(4))) _Float16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int PM = POLICY_TOKENS_PER_WG;   // 160
constexpr int E = 128;
#ifndef POL_SWIZZLE
constexpr int LDA = 136;               // halves per row of a 128-wide LDS image: rows padded by 16 bytes
#else
constexpr int LDA = 128;               // A/B build: no padding, 16-byte slots XOR-swizzled by the row (swz)
#endif
constexpr int LDQ = 72;                // halves per row of a 64-wide LDS image (Q, K, V of four heads)
constexpr int QROWS = 176;             // 160 + 16 pad rows: the 32-row MFMA operands of the last variable stay in bounds
constexpr int LDV = 184;               // halves per row of the transposed V image [feature][token]: 160 tokens + 24 zeroed pad columns
constexpr int MT = 5;                  // 16-row tiles per wave (2 x 4 wave grid over 10 x N/16 tiles)

// Address (in halves) of column col of row `row` in a 128-wide image.  -DPOL_SWIZZLE (A/B build): unpadded rows, the 16-byte slot s of row r at
// slot s ^ (r & 15).  ds_read_b128 serves a wave in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...; banks =
// 16-byte slots of a 256-byte row); a fragment read has lane (r = lane & 15, g = lane >> 4) on slot g + 4 ks of row r, and the padded
// rows put two lanes of every group on one slot (8 LDS cycles per read instead of 4).  The swizzle removes that -- measured:
// SQ_LDS_BANK_CONFLICT 333 M -> 105 M, conflict share 0.46 -> 0.21, LDS-active cycles -32 % -- and the kernel is 2.5 % SLOWER (2.86 vs
// 2.79 ms): the LDS is active 30-45 % of the time either way, the waves wait on the latency of their own fragment reads, and the
// swizzled addresses cost the paired stores (ds_write2_b64) and extra waits.  Default: padded.  
#ifndef POL_SWIZZLE
__device__ __forceinline__ int swz(int row, int col) { return row * LDA + col; }
#else
__device__ __forceinline__ int swz(int row, int col) { return row * LDA + (col ^ ((row & 15) << 3)); }
#endif

struct Lds {
    f16 h[PM * LDA];          // fp16 image of the residual stream (A operand of the QKV and FF-up GEMMs)
    f16 ao[PM * LDA];         // attention output (A operand of the output projection); start of kernel: staging of x
    union {
        struct { f16 q[QROWS * LDQ], k[QROWS * LDQ], vt[64 * LDV]; } a;   // V transposed: the P V product reads 4 keys at a time
        f16 ff[PM * LDA];     // one 128-wide chunk of relu(FF-up)
    } u;
};
static_assert(sizeof(Lds) <= 160 * 1024, "LDS budget");

__device__ __forceinline__ f32x4 mfma16(f16x8 a, f16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 mfma32(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }

// Weight fragments of one GEMM for this wave: KS k-steps x NT feature tiles, 16 bytes per lane each.
template <int NT>
struct WFrag { f16x8 v[4][NT]; };

// fragment (feature tile, k-step) at (tile*4 + ks)*64 + lane.  Issued well before use: the loads fly while the previous GEMM runs.
template <int NT>
__device__ __forceinline__ void load_w(WFrag<NT> &w, const f16x8 *wp, int tile0, int tstride, int lane) {
#ifdef POL_W_ONCE      // timing only: no weight traffic at all (what do the GEMMs wait for?)
    if (lane >= 0) {
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) w.v[ks][nt] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        return;
    }
#endif
#pragma unroll
    for (int nt = 0; nt < NT; nt++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) w.v[ks][nt] = wp[((size_t)(tile0 + nt * tstride) * 4 + ks) * 64 + lane];
}
// the wave's tiles of one Q|K|V block of four heads (12 feature tiles: Q of heads 0-3, K, V): PER of each third, starting at head PER * wn
template <int PER>
__device__ __forceinline__ void load_w_qkv(WFrag<3 * PER> &w, const f16x8 *wp, int wn, int lane) {
#ifdef POL_W_ONCE
    if (lane >= 0) {
#pragma unroll
        for (int nt = 0; nt < 3 * PER; nt++)
#pragma unroll
            for (int ks = 0; ks < 4; ks++) w.v[ks][nt] = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
        return;
    }
#endif
#pragma unroll
    for (int nt = 0; nt < 3 * PER; nt++)
#pragma unroll
        for (int ks = 0; ks < 4; ks++) w.v[ks][nt] = wp[((size_t)((nt / PER) * 4 + wn * PER + nt % PER) * 4 + ks) * 64 + lane];
}

// acc[mt][nt] += (W^T)[this wave's NT feature tiles][0,128) * (ACT^T)[0,128)[this wave's 5 token tiles]
// The weights are the MFMA's A operand and the activations its B operand, so a lane ends up with 4 CONSECUTIVE FEATURES of one
// token: element r of acc[mt][nt] is token row0 + 16 mt + (lane & 15), feature 16 (tile0 + nt) + 4 (lane >> 4) + r, and an
// epilogue stores it as one 8-byte LDS write.  ACT: LDS fp16 image with row stride lda (halves).
// PLAIN: the last PLAIN feature tiles are computed the other way round (activations = A operand): its lanes then hold 4 consecutive
// TOKENS of one feature, which is what the transposed V image wants.
template <int NT, int PLAIN = 0>
__device__ __forceinline__ void gemm_tiles(const f16 *act, int lda, int row0, const WFrag<NT> &w, f32x4 (&acc)[MT][NT], int lane) {
    // row0 is a multiple of 16, so row & 15 = lane & 15 for every tile: the (swizzled) column of k-step ks is cb ^ (32 ks) (= cb + 32 ks unswizzled)
    const f16 *arow = act + (size_t)(row0 + (lane & 15)) * lda;
#ifndef POL_SWIZZLE
    const int cb = 8 * (lane >> 4);
#else
    const int cb = (8 * (lane >> 4)) ^ ((lane & 15) << 3);
#endif
    if constexpr (NT >= 4) {
        // one wave per SIMD (the 2 x 2 wave grid): nobody else hides the LDS latency, so the activation fragments of k-step ks + 1 are
        // requested before the MFMAs of k-step ks are issued (two register sets of MT fragments)
        f16x8 a[2][MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) a[0][mt] = *(const f16x8 *)(arow + (size_t)mt * 16 * lda + cb);
#pragma unroll
        for (int ks = 0; ks < 4; ks++) {
            if (ks < 3) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) a[(ks + 1) & 1][mt] = *(const f16x8 *)(arow + (size_t)mt * 16 * lda + (cb ^ ((ks + 1) * 32)));
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt] = (nt >= NT - PLAIN) ? mfma16(a[ks & 1][mt], w.v[ks][nt], acc[mt][nt]) : mfma16(w.v[ks][nt], a[ks & 1][mt], acc[mt][nt]);
        }
        return;
    }
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const f16x8 a = *(const f16x8 *)(arow + (size_t)mt * 16 * lda + (cb ^ (ks * 32)));
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
                acc[mt][nt] = (nt >= NT - PLAIN) ? mfma16(a, w.v[ks][nt], acc[mt][nt]) : mfma16(w.v[ks][nt], a, acc[mt][nt]);
        }
    }
}

template <int NT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MT][NT]) {
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) acc[mt][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// value of the same lane in the other half of the wave (lanes l and l ^ 32), combined by the caller
__device__ __forceinline__ float swap_halves(float v) {
    const int x = __float_as_int(v);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // r[0]: lower half's values in both halves, r[1]: upper half's
    return __int_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

__device__ __forceinline__ void store4(f16 *dst, float a, float b, float c, float d) {
    f16x4 h = {(f16)a, (f16)b, (f16)c, (f16)d};
    *(f16x4 *)dst = h;
}

// NW = column groups of the wave grid (2 x NW waves).  NW = 4: eight waves, 80 x 32 outputs each (two waves per SIMD).  NW = 2: four waves
// of 80 x 64 (one per SIMD, up to 512 registers each): every activation fragment read from LDS feeds twice as many MFMAs, i.e. half
// the LDS fragment traffic per GEMM -- the resource that was level with the MFMA time in the eight-wave form.
template <int TOK, int NW>
__global__ void __launch_bounds__(128 * NW) policy_body_kernel(PolicyArgs pa) {
    constexpr int PT = 128 * NW;                    // threads
    constexpr int NTO = 8 / NW;                     // feature tiles per wave of a 128-wide GEMM output
    constexpr int PER = 4 / NW;                     // heads per wave of a Q|K|V block of four heads
    constexpr int NTQ = 3 * PER;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    Lds &S = *reinterpret_cast<Lds *>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / NW, wn = wave % NW;
    const int row0 = wm * 80;                       // first token of this wave's 5 token tiles
    const int l15 = lane & 15, g4 = (lane >> 4) * 4;
    constexpr int VARS = PM / TOK;                  // variables per workgroup
    const long var0 = (long)blockIdx.x * VARS;
    const int nvar = (int)min((long)VARS, pa.rows - var0);
    const f16x8 *wbase = reinterpret_cast<const f16x8 *>(pa.weights);

    WFrag<NTQ> w3a, w3b;
    WFrag<NTO> w2a, w2b;
    load_w_qkv<PER>(w3a, wbase, wn, lane);          // layer 0: this wave's feature tiles of Q, K, V (heads 0-3)

    // ---- stage x (fp64 in the solver's buffer) as float [160][5] in the `ao` region; zero the pad rows of Q/K/V ----
    float *xs = reinterpret_cast<float *>(S.ao);
    for (int e = tid; e < PM * 5; e += PT) {
        const int tok = e / 5, c = e - tok * 5;
        const int v = tok / TOK, t = tok - v * TOK;
        float val = 0.f;
#ifndef POL_NO_IN
        if (v < nvar) val = (float)pa.x[pa.row_off[var0 + v] + (long)t * pa.tok_stride + c];
#endif
        xs[e] = val;
    }
    // pad columns of V^T (keys beyond the last variable's tokens) meet probabilities that are exactly 0: they must be finite.
    // Nothing else writes them (the FF chunk buffer aliases Q and part of K only).  Pad rows of Q / K may hold anything:
    // padded queries are never stored, padded keys are masked.
    static_assert(PM * LDA * 2 <= 2 * QROWS * LDQ * 2, "the FF chunk buffer must not reach the V image");
    for (int e = tid; e < 64 * (LDV - PM); e += PT) S.u.a.vt[(e / (LDV - PM)) * LDV + PM + e % (LDV - PM)] = (f16)0.f;
    __syncthreads();

    // ---- embedding: H = x W_in + (position code W_pos + bias), straight into the accumulator layout ----
    f32x4 H[MT][NTO];
    {
        const float *win = pa.consts + POLICY_OFF_WIN, *bin = pa.consts + POLICY_OFF_BIN;
#pragma unroll
        for (int nt = 0; nt < NTO; nt++) {
            const int f0 = (wn * NTO + nt) * 16 + g4;
            f32x4 w[5];
#pragma unroll
            for (int c = 0; c < 5; c++) w[c] = *(const f32x4 *)(win + c * E + f0);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int tok = row0 + mt * 16 + l15;
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
        for (int nt = 0; nt < NTO; nt++)
            store4(S.h + swz(row0 + mt * 16 + l15, (wn * NTO + nt) * 16 + g4), H[mt][nt][0], H[mt][nt][1], H[mt][nt][2], H[mt][nt][3]);
    __syncthreads();

#pragma unroll
    for (int layer = 0; layer < 2; layer++) {
        const f16x8 *wl = wbase + (size_t)layer * POLICY_FRAGS_PER_LAYER * 64;
        const float *cl = pa.consts + POLICY_OFF_LAYER(TOK) + layer * POLICY_LAYER_CONSTS;

        // ================= self-attention, four heads at a time =================
#pragma unroll
        for (int half = 0; half < 2; half++) {
            {   // Q | K | V of heads 4 half .. 4 half + 3:  (160 x 128) x (128 x 192); meanwhile fetch the next GEMM's weights
                f32x4 acc[MT][NTQ];
                zero_acc<NTQ>(acc);
#ifdef POL_NO_QKV
                if (half == 0) load_w_qkv<PER>(w3b, wl + (size_t)48 * 64, wn, lane); else load_w<NTO>(w2a, wl + (size_t)96 * 64, wn * NTO, 1, lane);
                if (pa.rows < 0)
#else
                if (half == 0) { load_w_qkv<PER>(w3b, wl + (size_t)48 * 64, wn, lane); gemm_tiles<NTQ, PER>(S.h, LDA, row0, w3a, acc, lane); }
                else           { load_w<NTO>(w2a, wl + (size_t)96 * 64, wn * NTO, 1, lane); gemm_tiles<NTQ, PER>(S.h, LDA, row0, w3b, acc, lane); }
#endif
#ifdef POL_NO_QKV_EPI
                if (POL_EPI_GUARD(pa))
#endif
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int tok = row0 + mt * 16 + l15;
#pragma unroll
                    for (int j = 0; j < PER; j++) {
                        const int hd = wn * PER + j;           // head (of the four of this half) = feature tile inside Q, K, V
                        store4(S.u.a.q + tok * LDQ + hd * 16 + g4, acc[mt][j][0], acc[mt][j][1], acc[mt][j][2], acc[mt][j][3]);
                        store4(S.u.a.k + tok * LDQ + hd * 16 + g4, acc[mt][PER + j][0], acc[mt][PER + j][1], acc[mt][PER + j][2], acc[mt][PER + j][3]);
                        // V tile: lane = feature hd*16 + l15, registers = tokens row0 + 16 mt + g4 .. + 3
                        store4(S.u.a.vt + (hd * 16 + l15) * LDV + row0 + mt * 16 + g4, acc[mt][2 * PER + j][0], acc[mt][2 * PER + j][1],
                               acc[mt][2 * PER + j][2], acc[mt][2 * PER + j][3]);
                    }
                }
            }
            __syncthreads();
            // ---- per (variable, head): S^T = K Q^T (keys in registers, queries on lanes), softmax over registers, O^T = V^T P^T ----
            {
                const int r31 = lane & 31, hf = lane >> 5;
#pragma unroll 1
                for (int pair = wave; pair < POL_ATTN_PAIRS(VARS * 4); pair += PT / 64) {
                    const int v = pair >> 2, hh = pair & 3;
                    const int tok0 = v * TOK;
                    // rows past the variable's tokens only feed masked keys / unused queries; keep the read inside the image all the same
                    const int qrow = (TOK * (VARS - 1) + 31 < QROWS) ? tok0 + r31 : min(tok0 + r31, QROWS - 1);
                    const int off = qrow * LDQ + hh * 16 + 8 * hf;
                    const f16x8 ka = *(const f16x8 *)(S.u.a.k + off);        // A: row = key r31, k = 8 hf + e
                    const f16x8 qb = *(const f16x8 *)(S.u.a.q + off);        // B: col = query r31 (already scaled by 1/sqrt(16), mha.py:42)
                    // V^T fragments for the second product, requested now: element e of k-step s is key 16 s + 8 (e>>2) + 4 hf + (e&3)
                    f16x8 vfrag[2];
                    const f16 *vrow = S.u.a.vt + (hh * 16 + (r31 & 15)) * LDV + tok0 + 4 * hf;
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        if (16 * s < TOK) {
                            if (TOK % 4 == 0) {
                                const f16x4 lo = *(const f16x4 *)(vrow + 16 * s), hi = *(const f16x4 *)(vrow + 16 * s + 8);
                                vfrag[s] = f16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                            } else {
#pragma unroll
                                for (int e = 0; e < 8; e++) vfrag[s][e] = vrow[16 * s + 8 * (e >> 2) + (e & 3)];
                            }
                        }
                    }
                    f32x16 st;
#pragma unroll
                    for (int i = 0; i < 16; i++) st[i] = 0.f;
                    st = mfma32(ka, qb, st);                                   // st[reg]: key (reg&3) + 8 (reg>>2) + 4 hf, query r31
                    // register i holds key kmin (lower lane half) or kmin + 4 (upper half): valid always / for the lower half only / never
                    float mx = -3.0e38f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int kmin = (i & 3) + 8 * (i >> 2);
                        if (kmin >= TOK) continue;
                        if (kmin + 4 < TOK) mx = fmaxf(mx, st[i]);
                        else mx = fmaxf(mx, hf == 0 ? st[i] : -3.0e38f);
                    }
                    mx = fmaxf(mx, swap_halves(mx));
                    const float mxl = mx * 1.44269504088896f;
                    float sum = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const int kmin = (i & 3) + 8 * (i >> 2);
                        if (kmin >= TOK) { st[i] = 0.f; continue; }
                        float p = __builtin_amdgcn_exp2f(fmaf(st[i], 1.44269504088896f, -mxl));   // raw v_exp_f32 (exp2f() wraps it in five more instructions for denormal results): the argument is <= 0, a result below 2^-126 may flush to 0
                        if (kmin + 4 >= TOK) p = hf == 0 ? p : 0.f;
                        st[i] = p;
                        sum += p;
                    }
                    sum += swap_halves(sum);
                    const float inv = __builtin_amdgcn_rcpf(sum);                  // sum >= 1 (the maximum contributes exp2(0)); 1 ulp instead of the IEEE division's sequence
                    f32x16 o;
#pragma unroll
                    for (int i = 0; i < 16; i++) o[i] = 0.f;
#pragma unroll
                    for (int s = 0; s < 2; s++) {
                        if (16 * s < TOK) {
                            f16x8 pfrag;
#pragma unroll
                            for (int e = 0; e < 8; e++) pfrag[e] = (f16)(st[8 * s + e] * inv);   // P^T rows 16 s + 8 (e>>2) + 4 hf + (e&3)
                            o = mfma32(vfrag[s], pfrag, o);                    // o[reg]: dim (reg&3) + 8 (reg>>2) + 4 hf, query r31
                        }
                    }
                    if (r31 < TOK) {
                        const int oc = (half * 4 + hh) * 16 + 4 * hf;
                        store4(S.ao + swz(tok0 + r31, oc), o[0], o[1], o[2], o[3]);
                        store4(S.ao + swz(tok0 + r31, oc + 8), o[4], o[5], o[6], o[7]);
                    }
                }
            }
            __syncthreads();
        }

        // ================= output projection + residual + BatchNorm (eval) =================
        {
            f32x4 acc[MT][NTO];
            zero_acc<NTO>(acc);
            load_w<NTO>(w2b, wl + (size_t)128 * 64, wn * NTO, 1, lane);          // FF-up chunk 0
#ifndef POL_NO_PROJ
            gemm_tiles<NTO>(S.ao, LDA, row0, w2a, acc, lane);
#endif
#pragma unroll
            for (int nt = 0; nt < NTO; nt++) {
                const int f0 = (wn * NTO + nt) * 16 + g4;
                const f32x4 s1 = *(const f32x4 *)(cl + POLICY_LC_S1 + f0), t1 = *(const f32x4 *)(cl + POLICY_LC_T1 + f0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 hv = (H[mt][nt] + acc[mt][nt]) * s1 + t1;
                    H[mt][nt] = hv;
                    store4(S.h + swz(row0 + mt * 16 + l15, f0), hv[0], hv[1], hv[2], hv[3]);
                }
            }
        }
        __syncthreads();

        // ================= feed-forward 128 -> 512 -> 128, hidden layer in four 128-wide chunks =================
        {
            f32x4 acc2[MT][NTO];
            zero_acc<NTO>(acc2);
#pragma unroll
            for (int c = 0; c < POL_FF_CHUNKS; c++) {
                {
                    f32x4 acc[MT][NTO];
                    zero_acc<NTO>(acc);
                    load_w<NTO>(w2a, wl + (size_t)(256 + c * 32) * 64, wn * NTO, 1, lane);  // FF-down chunk c
                    gemm_tiles<NTO>(S.h, LDA, row0, w2b, acc, lane);
#ifdef POL_NO_FF_EPI
                    if (POL_EPI_GUARD(pa))
#endif
#pragma unroll
                    for (int nt = 0; nt < NTO; nt++) {
                        const int f0 = (wn * NTO + nt) * 16 + g4;
                        const f32x4 b1 = *(const f32x4 *)(cl + POLICY_LC_B1 + c * E + f0);
#pragma unroll
                        for (int mt = 0; mt < MT; mt++) {
                            const f32x4 hv = acc[mt][nt] + b1;
                            store4(S.u.ff + swz(row0 + mt * 16 + l15, f0), fmaxf(hv[0], 0.f), fmaxf(hv[1], 0.f), fmaxf(hv[2], 0.f), fmaxf(hv[3], 0.f));
                        }
                    }
                }
                __syncthreads();
                if (c < 3) load_w<NTO>(w2b, wl + (size_t)(128 + (c + 1) * 32) * 64, wn * NTO, 1, lane);             // next FF-up chunk
                else if (layer == 0) load_w_qkv<PER>(w3a, wbase + (size_t)POLICY_FRAGS_PER_LAYER * 64, wn, lane);  // next layer's Q|K|V
                gemm_tiles<NTO>(S.u.ff, LDA, row0, w2a, acc2, lane);
                __syncthreads();
            }
#pragma unroll
            for (int nt = 0; nt < NTO; nt++) {
                const int f0 = (wn * NTO + nt) * 16 + g4;
                const f32x4 b2 = *(const f32x4 *)(cl + POLICY_LC_B2 + f0), s2 = *(const f32x4 *)(cl + POLICY_LC_S2 + f0),
                            t2 = *(const f32x4 *)(cl + POLICY_LC_T2 + f0);
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const f32x4 hv = (H[mt][nt] + acc2[mt][nt] + b2) * s2 + t2;
                    H[mt][nt] = hv;
                    store4(S.h + swz(row0 + mt * 16 + l15, f0), hv[0], hv[1], hv[2], hv[3]);
                }
            }
        }
        __syncthreads();
    }

    // ---- flattened activations (variables x TOK*128, fp16), coalesced 16-byte stores ----
    {
        f16 *out = reinterpret_cast<f16 *>(pa.out) + var0 * (long)(TOK * E);
        const int ntok = nvar * TOK;
        for (int e = tid; e < PM * (E / 8); e += PT) {
            const int tok = e >> 4, c8 = e & 15;
#ifdef POL_NO_OUT
            if (tok < ntok && pa.rows < 0)
#else
            if (tok < ntok)
#endif
                *(f16x8 *)(out + (long)tok * E + c8 * 8) = *(const f16x8 *)(S.h + swz(tok, c8 * 8));
        }
    }
}

}  // namespace

size_t policy_lds_bytes() { return sizeof(Lds); }

hipError_t policy_launch_body(const PolicyArgs &pa, int tokens, hipStream_t s) {
    if (pa.rows <= 0) return hipSuccess;
    const size_t lds = sizeof(Lds);
    // eight waves of 80 x 32 outputs (default) or LPBOX_POLICY_WAVES=4: four waves of 80 x 64.  Measured (128 000 variables, forward incl.
    // the head): 3.68 vs 4.38 ms -- the wider tile does take 19 % off the FF GEMMs (1.20 -> 0.97 ms: half the LDS fragment traffic), but
    // with one wave per SIMD the attention phase (0.72 -> 1.10 ms) and the QKV / projection / IO phases (1.76 -> 2.31 ms) lose more.
    const bool eight = !(getenv("LPBOX_POLICY_WAVES") && atoi(getenv("LPBOX_POLICY_WAVES")) == 4);
    if (tokens != 20 && tokens != 5) return hipErrorInvalidValue;
    const long groups = (pa.rows + (PM / tokens) - 1) / (PM / tokens);
    auto go = [&](auto kernel, int threads) {
        hipError_t e = hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kernel, dim3((unsigned)groups), dim3(threads), lds, s, pa);
        return hipGetLastError();
    };
    if (tokens == 20) return eight ? go(policy_body_kernel<20, 4>, 512) : go(policy_body_kernel<20, 2>, 256);
    return eight ? go(policy_body_kernel<5, 4>, 512) : go(policy_body_kernel<5, 2>, 256);
}

// =====================================================================================================================
// The WHOLE network in fp32 (the reference's arithmetic: LP/mha.py:202-249 evaluates in float32), one workgroup per variable.
// Not a fast path: plain FMA loops, weights streamed from L2, activations in LDS.  It exists so that a fixing decision near a
// threshold is taken by fp32 arithmetic on the device without leaving this library (lpbox_hip.policy.FusedEarlyFixPolicy re-scores
// the rows whose fp16 score lies within its decision band of 0.9 / 0.1 with it), and as the fp32 reference of the fused kernel.
// Weight layout (floats, lpbox_hip/policy.py packs it): W_in (5x128) | position bias (TOK x 128) | per layer: W_qkv (128x384,
// columns Q|K|V, head h = columns 16h..16h+15 of each third) W_o (128x128) s1 t1 (128 each) W_1 (128x512) b1 (512) W_2 (512x128)
// b2 s2 t2 (128 each) | head: fc1 (TOK*128 x 256) b (256) fc2 (256x128) b (128) fc3 (128x16) b (16) fc4 (16) b (1).
// =====================================================================================================================
namespace {

constexpr int F32_T = 256;
constexpr long f32_layer_floats() { return 128L * 384 + 128 * 128 + 128 + 128 + 128 * 512 + 512 + 512 * 128 + 128 + 128 + 128; }

// band > 0: out_sig holds scores already (the fused fp16 path's); only the variables whose score lies within `band` of thr_hi or thr_lo
// are evaluated and overwritten, and *rescored counts them -- no host round trip to find them.
template <int TOK>
__global__ void __launch_bounds__(F32_T) policy_f32_kernel(const double *x, const long long *row_off, long rows, int tok_stride,
                                                           const float *W, float *out_sig, float *out_logit, float band, float thr_hi,
                                                           float thr_lo, unsigned long long *rescored) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *h = reinterpret_cast<float *>(smem);          // [TOK][128]
    float *qkv = h + TOK * 128;                          // [TOK][384]; later the head's intermediate vectors
    float *ao = qkv + TOK * 384;                         // [TOK][128]
    float *ff = ao + TOK * 128;                          // [TOK][512]
    float *xs = ff + TOK * 512;                          // [TOK][5]
    const int t = threadIdx.x;
    const float *w_in = W, *b_in = W + 5 * 128, *layer0 = b_in + TOK * 128;
    const float *head = layer0 + 2 * f32_layer_floats();
    for (long var = blockIdx.x; var < rows; var += gridDim.x) {
    if (band > 0.f) {                                    // uniform over the workgroup
        const float sc = out_sig[var];
        if (!(fabsf(sc - thr_hi) < band || fabsf(sc - thr_lo) < band)) continue;
        if (t == 0 && rescored) atomicAdd(rescored, 1ull);
    }
    __syncthreads();                                     // the previous variable's head has finished with the LDS buffers
    if (t < TOK * 5) xs[t] = (float)x[row_off[var] + (long)(t / 5) * tok_stride + t % 5];
    __syncthreads();
    for (int e = t; e < TOK * 128; e += F32_T) {
        const int tok = e >> 7, f = e & 127;
        float a = b_in[tok * 128 + f];
#pragma unroll
        for (int c = 0; c < 5; c++) a += xs[tok * 5 + c] * w_in[c * 128 + f];
        h[e] = a;
    }
    __syncthreads();
    const int col = t & 127, tok0 = t >> 7;              // 128-column stages: two threads per column, tokens tok0, tok0 + 2, ...
    constexpr int TH = (TOK + 1) / 2;
    for (int layer = 0; layer < 2; layer++) {
        const float *wqkv = layer0 + layer * f32_layer_floats(), *wo = wqkv + 128 * 384, *s1 = wo + 128 * 128, *t1 = s1 + 128;
        const float *w1 = t1 + 128, *b1 = w1 + 128 * 512, *w2 = b1 + 512, *b2 = w2 + 512 * 128, *s2 = b2 + 128, *t2 = s2 + 128;
        for (int j = t; j < 384; j += F32_T) {           // Q | K | V
            float acc[TOK];
#pragma unroll
            for (int k = 0; k < TOK; k++) acc[k] = 0.f;
            for (int k = 0; k < 128; k += 4) {          // four weights per 16-byte LDS read of the activations (the LDS issue rate binds)
                const float w0 = wqkv[k * 384 + j], w1_ = wqkv[(k + 1) * 384 + j], w2_ = wqkv[(k + 2) * 384 + j], w3 = wqkv[(k + 3) * 384 + j];
#pragma unroll
                for (int tk = 0; tk < TOK; tk++) {
                    const f32x4 hv = *(const f32x4 *)(h + tk * 128 + k);
                    acc[tk] += hv[0] * w0 + hv[1] * w1_ + hv[2] * w2_ + hv[3] * w3;
                }
            }
#pragma unroll
            for (int tk = 0; tk < TOK; tk++) qkv[tk * 384 + j] = acc[tk];
        }
        __syncthreads();
        for (int pr = t; pr < 8 * TOK; pr += F32_T) {    // one (head, query) pair per thread: softmax(q.k / sqrt(16)) v  (mha.py:42, :86-104)
            const int hh = pr / TOK, qi = pr - hh * TOK;
            float s[TOK], mx = -3.0e38f;
#pragma unroll
            for (int j = 0; j < TOK; j++) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 16; e++) d += qkv[qi * 384 + hh * 16 + e] * qkv[j * 384 + 128 + hh * 16 + e];
                s[j] = d * 0.25f;
                mx = fmaxf(mx, s[j]);
            }
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < TOK; j++) { s[j] = expf(s[j] - mx); sum += s[j]; }
            const float inv = 1.f / sum;
#pragma unroll
            for (int e = 0; e < 16; e++) {
                float o = 0.f;
#pragma unroll
                for (int j = 0; j < TOK; j++) o += (s[j] * inv) * qkv[j * 384 + 256 + hh * 16 + e];
                ao[qi * 128 + hh * 16 + e] = o;
            }
        }
        __syncthreads();
        {                                                // output projection + residual + BatchNorm (eval)
            float acc[TH];
#pragma unroll
            for (int k = 0; k < TH; k++) acc[k] = 0.f;
            for (int k = 0; k < 128; k += 4) {
                const float w0 = wo[k * 128 + col], w1_ = wo[(k + 1) * 128 + col], w2_ = wo[(k + 2) * 128 + col], w3 = wo[(k + 3) * 128 + col];
#pragma unroll
                for (int i = 0; i < TH; i++) {
                    const int tk = tok0 + 2 * i;
                    if (tk < TOK) { const f32x4 v = *(const f32x4 *)(ao + tk * 128 + k); acc[i] += v[0] * w0 + v[1] * w1_ + v[2] * w2_ + v[3] * w3; }
                }
            }
#pragma unroll
            for (int i = 0; i < TH; i++) { const int tk = tok0 + 2 * i; if (tk < TOK) h[tk * 128 + col] = (h[tk * 128 + col] + acc[i]) * s1[col] + t1[col]; }
        }
        __syncthreads();
        for (int j = t; j < 512; j += F32_T) {           // FF up + ReLU
            float acc[TOK];
#pragma unroll
            for (int k = 0; k < TOK; k++) acc[k] = 0.f;
            for (int k = 0; k < 128; k += 4) {
                const float w0 = w1[k * 512 + j], w1_ = w1[(k + 1) * 512 + j], w2_ = w1[(k + 2) * 512 + j], w3 = w1[(k + 3) * 512 + j];
#pragma unroll
                for (int tk = 0; tk < TOK; tk++) {
                    const f32x4 hv = *(const f32x4 *)(h + tk * 128 + k);
                    acc[tk] += hv[0] * w0 + hv[1] * w1_ + hv[2] * w2_ + hv[3] * w3;
                }
            }
#pragma unroll
            for (int tk = 0; tk < TOK; tk++) ff[tk * 512 + j] = fmaxf(acc[tk] + b1[j], 0.f);
        }
        __syncthreads();
        {                                                // FF down + residual + BatchNorm (eval)
            float acc[TH];
#pragma unroll
            for (int k = 0; k < TH; k++) acc[k] = 0.f;
            for (int k = 0; k < 512; k += 4) {
                const float w0 = w2[k * 128 + col], w1_ = w2[(k + 1) * 128 + col], w2_ = w2[(k + 2) * 128 + col], w3 = w2[(k + 3) * 128 + col];
#pragma unroll
                for (int i = 0; i < TH; i++) {
                    const int tk = tok0 + 2 * i;
                    if (tk < TOK) { const f32x4 v = *(const f32x4 *)(ff + tk * 512 + k); acc[i] += v[0] * w0 + v[1] * w1_ + v[2] * w2_ + v[3] * w3; }
                }
            }
#pragma unroll
            for (int i = 0; i < TH; i++) { const int tk = tok0 + 2 * i; if (tk < TOK) h[tk * 128 + col] = (h[tk * 128 + col] + acc[i] + b2[col]) * s2[col] + t2[col]; }
        }
        __syncthreads();
    }
    // head: flatten (TOK*128) -> 256 -> 128 -> 16 -> 1, ReLU between (mha.py:185-199, :240-247)
    const float *f1 = head, *c1 = f1 + (long)TOK * 128 * 256, *f2 = c1 + 256, *c2 = f2 + 256 * 128, *f3 = c2 + 128, *c3 = f3 + 128 * 16,
                *f4 = c3 + 16, *c4 = f4 + 16;
    float *z1 = qkv, *z2 = qkv + 256, *z3 = qkv + 384;
    {
        float a = c1[t];
        for (int i = 0; i < TOK * 128; i += 4) {
            const f32x4 hv = *(const f32x4 *)(h + i);
            a += hv[0] * f1[(long)i * 256 + t] + hv[1] * f1[(long)(i + 1) * 256 + t] + hv[2] * f1[(long)(i + 2) * 256 + t] + hv[3] * f1[(long)(i + 3) * 256 + t];
        }
        z1[t] = fmaxf(a, 0.f);
    }
    __syncthreads();
    if (t < 128) { float a = c2[t]; for (int i = 0; i < 256; i++) a += z1[i] * f2[i * 128 + t]; z2[t] = fmaxf(a, 0.f); }
    __syncthreads();
    if (t < 16) { float a = c3[t]; for (int i = 0; i < 128; i++) a += z2[i] * f3[i * 16 + t]; z3[t] = fmaxf(a, 0.f); }
    __syncthreads();
    if (t == 0) {
        float a = c4[0];
        for (int i = 0; i < 16; i++) a += z3[i] * f4[i];
        if (out_logit) out_logit[var] = a;
        out_sig[var] = 1.f / (1.f + expf(-a));
    }
    }
}

}  // namespace

long policy_f32_weight_floats(int tokens) {
    return 5L * 128 + (long)tokens * 128 + 2 * f32_layer_floats() + (long)tokens * 128 * 256 + 256 + 256 * 128 + 128 + 128 * 16 + 16 + 16 + 1;
}

hipError_t policy_launch_f32(const double *x, const long long *row_off, long rows, int tokens, int tok_stride, const float *W,
                             float *out_sig, float *out_logit, float band, float thr_hi, float thr_lo, unsigned long long *rescored,
                             hipStream_t s) {
    if (rows <= 0) return hipSuccess;
    const unsigned grid = (unsigned)std::min<long>(rows, band > 0.f ? 2048 : 65535L * 16);
    const size_t lds = sizeof(float) * (size_t)tokens * (128 + 384 + 128 + 512 + 5);
    if (tokens == 20) {
        hipError_t e = hipFuncSetAttribute((const void *)policy_f32_kernel<20>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(policy_f32_kernel<20>, dim3(grid), dim3(F32_T), lds, s, x, row_off, rows, tok_stride, W, out_sig, out_logit, band, thr_hi, thr_lo, rescored);
    } else if (tokens == 5) {
        hipError_t e = hipFuncSetAttribute((const void *)policy_f32_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(policy_f32_kernel<5>, dim3(grid), dim3(F32_T), lds, s, x, row_off, rows, tok_stride, W, out_sig, out_logit, band, thr_hi, thr_lo, rescored);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
