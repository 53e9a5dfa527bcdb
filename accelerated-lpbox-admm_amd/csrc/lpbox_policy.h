// lpbox_policy.h -- internal layout of the fused policy-encoder kernel (lpbox_policy_kernels.hip).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>

#define POLICY_THREADS 512
#define POLICY_TOKENS_PER_WG 160

// packed weights: fp16 fragments of v_mfma_f32_16x16x32_f16's B operand, 512 halves each; fragment (n-tile, k-step) of a
// (32 KS) x (16 NT) matrix at index n-tile*KS + k-step; element j of lane l = W[32 ks + 8 (l>>4) + j][16 nt + (l&15)].
// per layer: [0,48) Q|K|V of heads 0-3 (128x192), [48,96) heads 4-7, [96,128) output projection (128x128),
//            [128,256) FF-up in four column chunks (128x128 each), [256,384) FF-down in four row chunks (128x128 each)
#define POLICY_FRAGS_PER_LAYER 384
// float constants: W_in (5x128) | position bias (TOK x 128) | per layer: s1 t1 (128 each) b1 (512) b2 s2 t2 (128 each)
#define POLICY_OFF_WIN 0
#define POLICY_OFF_BIN 640
#define POLICY_OFF_LAYER(TOK) (640 + (TOK) * 128)
#define POLICY_LC_S1 0
#define POLICY_LC_T1 128
#define POLICY_LC_B1 256
#define POLICY_LC_B2 768
#define POLICY_LC_S2 896
#define POLICY_LC_T2 1024
#define POLICY_LAYER_CONSTS 1152

struct PolicyArgs {
    const double *x;            // the solver's packed x_iters (or any fp64 buffer)
    const long long *row_off;   // per variable: offset (in doubles) of its first iterate
    long rows;                  // variables
    int tok_stride;             // token t = iterates [t*tok_stride, t*tok_stride + 5)
    const void *weights;        // packed fp16 fragments, 2 layers
    const float *consts;
    void *out;                  // fp16 [rows][TOK*128]
};

size_t policy_lds_bytes();
hipError_t policy_launch_body(const PolicyArgs &pa, int tokens, hipStream_t s);

// the encoder in f32 on v_mfma_f32_16x16x4_f32 (lpbox_policy_f32_kernels.hip); pa.weights = f32 fragments, pa.out = float [rows][TOK*128]
long policy_f32frag_floats();
hipError_t policy_launch_body_f32(const PolicyArgs &pa, int tokens, hipStream_t s);

// the whole network in fp32, one workgroup per variable (reference arithmetic; used for decisions near a threshold)
long policy_f32_weight_floats(int tokens);
hipError_t policy_launch_f32(const double *x, const long long *row_off, long rows, int tokens, int tok_stride, const float *W,
                             float *out_sig, float *out_logit, float band, float thr_hi, float thr_lo, unsigned long long *rescored,
                             hipStream_t s);
