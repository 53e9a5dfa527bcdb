"""Inference-only early-fixing policy on the device.

The reference scores every live variable with `GraphAttentionEncoder` (LP/mha.py:202-249; SEG/mha.py is the same network
with 5 tokens): tokens of 5 consecutive iterates + a 5-dim sinusoidal position code (LP/common/utils.py:20-32) ->
Linear 10->128 -> 2 x [8-head self-attention d=128 (LP/mha.py:20-122) + BatchNorm1d + FF 128-512-128 + BatchNorm1d]
(:157-183) -> flatten -> MLP T*128-256-128-16-1 (:185-199) -> sigmoid.  Training stays with the reference's module on
PyTorch-ROCm; this class takes that module's `state_dict()` and evaluates it in the form that suits a solver loop on one
MI355X: weights re-laid once (Q/K/V of all heads in one 128x384 GEMM, the output projection as one 128x128 GEMM, the
position code folded into a constant (T,128) bias, eval-mode BatchNorm folded to scale/shift), rows in large chunks,
fused scaled-dot-product attention, nothing leaves the device.

Numerics: fp32 like the reference; agreement with the reference module is to rounding (different GEMM association),
tested at 2e-5 absolute on the sigmoid (tests/test_policy.py, golden vectors from the reference module itself).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

N_HEADS, EMBED, FF_HIDDEN, N_LAYERS, CODE_DIM = 8, 128, 512, 2, 5
BN_EPS = 1e-5            # torch.nn.BatchNorm1d default, which LP/mha.py:137 leaves untouched


def position_code(n_pos, d=CODE_DIM):
    """LP/common/utils.py:20-32: pos / 10000^(2*(j//2)/d), position 0 all zeros BEFORE sin/cos (so row 0 = 0,1,0,1,0)."""
    j = np.arange(d)
    pe = np.arange(n_pos, dtype=np.float64)[:, None] / np.power(10000.0, 2 * (j // 2) / d)[None, :]
    pe[0, :] = 0.0
    pe[:, 0::2] = np.sin(pe[:, 0::2])
    pe[:, 1::2] = np.cos(pe[:, 1::2])
    return torch.from_numpy(pe).to(torch.float32)


def reference_state_shapes(tokens=20):
    """name -> shape of every tensor in GraphAttentionEncoder().state_dict() (LP: tokens=20, SEG: tokens=5)."""
    hd = EMBED // N_HEADS
    shapes = {"init_embed.weight": (EMBED, 2 * CODE_DIM), "init_embed.bias": (EMBED,)}
    for i in range(N_LAYERS):
        p = "layers.%d." % i
        shapes.update({p + "0.module.W_query": (N_HEADS, EMBED, hd), p + "0.module.W_key": (N_HEADS, EMBED, hd),
                       p + "0.module.W_val": (N_HEADS, EMBED, hd), p + "0.module.W_out": (N_HEADS, hd, EMBED),
                       p + "2.module.0.weight": (FF_HIDDEN, EMBED), p + "2.module.0.bias": (FF_HIDDEN,),
                       p + "2.module.2.weight": (EMBED, FF_HIDDEN), p + "2.module.2.bias": (EMBED,)})
        for k in ("1", "3"):
            shapes.update({p + k + ".normalizer.weight": (EMBED,), p + k + ".normalizer.bias": (EMBED,),
                           p + k + ".normalizer.running_mean": (EMBED,), p + k + ".normalizer.running_var": (EMBED,),
                           p + k + ".normalizer.num_batches_tracked": ()})
    for name, (o, i) in (("fc1", (256, tokens * EMBED)), ("fc2", (128, 256)), ("fc3", (16, 128)), ("fc4", (1, 16))):
        shapes["classify.%s.weight" % name] = (o, i)
        shapes["classify.%s.bias" % name] = (o,)
    return shapes


def random_state(tokens=20, seed=0):
    """A state dict with the reference's initial distributions (LP/mha.py:52-56 uniform(+-1/sqrt(last dim)) for the attention
    weights, torch defaults elsewhere); stands in for the trained checkpoint, which the reference does not ship."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for name, shape in reference_state_shapes(tokens).items():
        if name.endswith("num_batches_tracked"):
            sd[name] = torch.zeros((), dtype=torch.long)
        elif name.endswith("running_mean") or name.endswith("normalizer.bias"):
            sd[name] = torch.zeros(shape)
        elif name.endswith("running_var") or name.endswith("normalizer.weight"):
            sd[name] = torch.ones(shape)
        else:
            fan = shape[-1] if ".W_" in name else (shape[1] if len(shape) == 2 else None)
            if fan is None:                                   # a Linear bias: bound from the matching weight's fan-in
                fan = reference_state_shapes(tokens)[name[:-4] + "weight"][1]
            sd[name] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan)
    return sd


class EarlyFixPolicy:
    """`policy(x)` with x a float32 tensor (rows, tokens, 5) -> sigmoid scores (rows,), all on `device`."""

    def __init__(self, state_dict, tokens=20, device="cuda", chunk_rows=65536, dtype=torch.float32):
        want = reference_state_shapes(tokens)
        missing = [k for k in want if k not in state_dict]
        if missing:
            raise KeyError("state dict lacks %s (is it a GraphAttentionEncoder checkpoint?)" % missing[:3])
        for k, shp in want.items():
            if tuple(state_dict[k].shape) != tuple(shp):
                raise ValueError("%s has shape %s, expected %s" % (k, tuple(state_dict[k].shape), shp))
        self.tokens, self.device, self.chunk_rows = tokens, torch.device(device), int(chunk_rows)
        self.dtype = dtype            # float32 = the reference's arithmetic; bfloat16 runs the GEMMs on the bf16 MFMA path (opt-in, scores move by ~1e-2)
        f = lambda k: state_dict[k].detach().to(torch.float32).cpu()
        we, be = f("init_embed.weight"), f("init_embed.bias")
        self.w_in = we[:, :CODE_DIM].t().contiguous().to(self.device)                     # (5,128): the iterates' half
        self.b_in = (position_code(tokens) @ we[:, CODE_DIM:].t() + be).to(self.device)     # (T,128): position half + bias
        self.layers = []
        for i in range(N_LAYERS):
            p = "layers.%d." % i
            heads = lambda k: f(p + "0.module." + k).permute(1, 0, 2).reshape(EMBED, EMBED)   # column block h = head h
            L = {"w_qkv": torch.cat([heads("W_query"), heads("W_key"), heads("W_val")], dim=1),
                 "w_o": f(p + "0.module.W_out").reshape(EMBED, EMBED),                      # row h*16+v (LP/mha.py:107-110)
                 "w1": f(p + "2.module.0.weight").t().contiguous(), "b1": f(p + "2.module.0.bias"),
                 "w2": f(p + "2.module.2.weight").t().contiguous(), "b2": f(p + "2.module.2.bias")}
            for k, tag in (("1", "n1"), ("3", "n2")):                                      # eval-mode BatchNorm = affine map
                s = f(p + k + ".normalizer.weight") / torch.sqrt(f(p + k + ".normalizer.running_var") + BN_EPS)
                L[tag + "_s"] = s
                L[tag + "_t"] = f(p + k + ".normalizer.bias") - f(p + k + ".normalizer.running_mean") * s
            self.layers.append({k: v.contiguous().to(self.device) for k, v in L.items()})
        self.head = [(f("classify.fc%d.weight" % k).t().contiguous().to(self.device), f("classify.fc%d.bias" % k).to(self.device))
                     for k in (1, 2, 3, 4)]
        if dtype != torch.float32:
            self.w_in, self.b_in = self.w_in.to(dtype), self.b_in.to(dtype)
            self.layers = [{k: v.to(dtype) for k, v in L.items()} for L in self.layers]
            self.head = [(w.to(dtype), b.to(dtype)) for w, b in self.head]

    @classmethod
    def random(cls, tokens=20, seed=0, **kw):
        return cls(random_state(tokens, seed), tokens=tokens, **kw)

    @torch.no_grad()
    def logits(self, x):
        x = x.to(self.device, self.dtype)
        if x.dim() != 3 or x.shape[1] != self.tokens or x.shape[2] != CODE_DIM:
            raise ValueError("expected (rows, %d, %d), got %s" % (self.tokens, CODE_DIM, tuple(x.shape)))
        out = torch.empty(x.shape[0], device=self.device, dtype=torch.float32)
        for r0 in range(0, x.shape[0], self.chunk_rows):
            out[r0:r0 + self.chunk_rows] = self._chunk(x[r0:r0 + self.chunk_rows])
        return out

    def _chunk(self, x):
        R, T = x.shape[0], self.tokens
        hd = EMBED // N_HEADS
        h = (x.reshape(R * T, CODE_DIM) @ self.w_in).view(R, T, EMBED) + self.b_in
        for L in self.layers:
            qkv = (h.view(R * T, EMBED) @ L["w_qkv"]).view(R, T, 3, N_HEADS, hd).permute(2, 0, 3, 1, 4)   # (3,R,heads,T,hd)
            a = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])                                      # scale 1/sqrt(hd) (LP/mha.py:42)
            h = h + (a.permute(0, 2, 1, 3).reshape(R * T, EMBED) @ L["w_o"]).view(R, T, EMBED)
            h = h * L["n1_s"] + L["n1_t"]
            h = h + (torch.relu(h.view(R * T, EMBED) @ L["w1"] + L["b1"]) @ L["w2"] + L["b2"]).view(R, T, EMBED)
            h = h * L["n2_s"] + L["n2_t"]
        z = h.reshape(R, T * EMBED)
        for k, (w, b) in enumerate(self.head):
            z = z @ w + b
            if k < 3:
                z = torch.relu(z)
        return z.view(R).to(torch.float32)

    def __call__(self, x):
        return torch.sigmoid(self.logits(x))


# ------------------------------------------------------------------------------------------------
# fused device path: the encoder (everything before the flatten) as ONE HIP kernel (csrc/lpbox_policy_kernels.hip)
# ------------------------------------------------------------------------------------------------
def _pack_fragments(w):
    """(K, N) fp32 matrix -> fp16 B-operand fragments of v_mfma_f32_16x16x32_f16 in the order the kernel streams them:
    fragment (n-tile, k-step) at n-tile*KS + k-step, element j of lane l = W[32 ks + 8 (l>>4) + j][16 nt + (l&15)]."""
    K, N = w.shape
    assert K % 32 == 0 and N % 16 == 0
    KS, NT = K // 32, N // 16
    # W[ks, g, j, nt, c] with k = 32 ks + 8 g + j, n = 16 nt + c ; lane = 16 g + c
    v = w.reshape(KS, 4, 8, NT, 16).permute(3, 0, 1, 4, 2)        # (nt, ks, g, c, j)
    return v.reshape(NT * KS * 64 * 8).to(torch.float16)


def _pack_fragments_f32(w):
    """(K, N) fp32 matrix -> A-operand fragments of v_mfma_f32_16x16x4_f32 in the order csrc/lpbox_policy_f32_kernels.hip streams them:
    fragment (n-tile, k-block of 16) at n-tile*KB + kb, float s of lane l = W[16 kb + 4 (l>>4) + s][16 nt + (l&15)] (the lane's four floats are
    the four k-steps of the block: the activations are read from LDS as one float4 in the same k order)."""
    K, N = w.shape
    assert K % 16 == 0 and N % 16 == 0
    KB, NT = K // 16, N // 16
    v = w.reshape(KB, 4, 4, NT, 16).permute(3, 0, 1, 4, 2)        # (nt, kb, kq, i, s); lane = 16 kq + i
    return v.reshape(-1).to(torch.float32)


class MfmaFp32Policy:
    """The reference's float32 arithmetic at usable speed: the encoder as ONE fused kernel on v_mfma_f32_16x16x4_f32 (f32 in, f32
    accumulate: a k-ordered fmaf chain, no reduced-precision step; C-ABI lpbox_policy_encode_f32) + the MLP head as three fp32 GEMMs.
    Same interface as FusedEarlyFixPolicy; agreement with the reference module: rounding (tested at 1e-4 on the sigmoid against the
    reference-generated golden vectors, 2e-5 against EarlyFixPolicy)."""

    def __init__(self, state_dict, tokens=20, device="cuda", chunk_rows=131072):
        import ctypes as C
        from . import _lib
        self._L, self._check = _lib.load(), _lib.check
        self.chunk_rows = int(chunk_rows)      # bounds the flattened activation buffer (rows x tokens*128 fp32 = 1.3 GB at 20 tokens)
        ref = EarlyFixPolicy(state_dict, tokens=tokens, device="cpu")
        self.tokens, self.device = tokens, torch.device(device)
        wf, cf = C.c_long(), C.c_long()
        self._check(self._L.lpbox_policy_f32frag_layout(tokens, C.byref(wf), C.byref(cf)), "lpbox_policy_f32frag_layout")
        frags, consts = [], [ref.w_in.reshape(-1), ref.b_in.reshape(-1)]
        for L in ref.layers:
            wq, wk, wv = L["w_qkv"][:, :EMBED], L["w_qkv"][:, EMBED:2 * EMBED], L["w_qkv"][:, 2 * EMBED:]
            for half in range(2):
                c = slice(64 * half, 64 * half + 64)                            # heads 4 half .. 4 half + 3
                frags.append(_pack_fragments_f32(torch.cat([wq[:, c] * 0.25, wk[:, c], wv[:, c]], dim=1)))   # 1/sqrt(16) folded into Q (exact)
            frags.append(_pack_fragments_f32(L["w_o"]))
            for c in range(4):
                frags.append(_pack_fragments_f32(L["w1"][:, 128 * c:128 * c + 128]))
            for c in range(4):
                frags.append(_pack_fragments_f32(L["w2"][128 * c:128 * c + 128, :]))
            consts += [L["n1_s"], L["n1_t"], L["b1"], L["b2"], L["n2_s"], L["n2_t"]]
        w = torch.cat(frags)
        c = torch.cat([t.reshape(-1).to(torch.float32) for t in consts])
        assert w.numel() == wf.value and c.numel() == cf.value, (w.numel(), wf.value, c.numel(), cf.value)
        self.w, self.c = w.contiguous().to(self.device), c.contiguous().to(self.device)
        self.head = [(wt.to(self.device, torch.float32), b.to(self.device, torch.float32)) for wt, b in ref.head]

    @classmethod
    def random(cls, tokens=20, seed=0, **kw):
        return cls(random_state(tokens, seed), tokens=tokens, **kw)

    @torch.no_grad()
    def encode(self, flat, row_off, tok_stride):
        """flat: fp64 CUDA tensor; row_off: int64 CUDA tensor (rows,) of offsets into flat -> fp32 (rows, tokens*128)."""
        rows = int(row_off.numel())
        out = torch.empty((rows, self.tokens * EMBED), device=self.device, dtype=torch.float32)
        if rows:
            if int(row_off.max().item()) + (self.tokens - 1) * tok_stride + CODE_DIM > flat.numel():
                raise ValueError("row offsets reach past the end of the iterate buffer")
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._L.lpbox_policy_encode_f32(flat.data_ptr(), row_off.contiguous().data_ptr(), rows, self.tokens, int(tok_stride),
                                                        self.w.data_ptr(), self.c.data_ptr(), out.data_ptr(), stream),
                        "lpbox_policy_encode_f32")
        return out

    @torch.no_grad()
    def logits_from_xiters(self, flat, row_off, tok_stride=None):
        out = torch.empty(row_off.numel(), device=self.device, dtype=torch.float32)
        for r0 in range(0, row_off.numel(), self.chunk_rows):
            z = self.encode(flat, row_off[r0:r0 + self.chunk_rows], CODE_DIM if tok_stride is None else tok_stride)
            for k, (w, b) in enumerate(self.head):
                z = torch.addmm(b, z, w)
                if k < 3:
                    z = torch.relu(z)
            out[r0:r0 + self.chunk_rows] = z.reshape(-1)
        return out

    @torch.no_grad()
    def scores_from_xiters(self, flat, row_off, tok_stride=None, logits=False):
        lg = self.logits_from_xiters(flat, row_off, tok_stride)
        return (torch.sigmoid(lg), lg) if logits else torch.sigmoid(lg)

    def __call__(self, x):
        x = x.to(self.device, torch.float64).contiguous()
        if x.dim() != 3 or x.shape[1] != self.tokens or x.shape[2] != CODE_DIM:
            raise ValueError("expected (rows, %d, %d), got %s" % (self.tokens, CODE_DIM, tuple(x.shape)))
        off = torch.arange(x.shape[0], device=self.device, dtype=torch.int64) * (self.tokens * CODE_DIM)
        return self.scores_from_xiters(x.reshape(-1), off, CODE_DIM)


class HipFp32Policy:
    """The whole network (encoder + MLP head) in fp32 by ONE plain HIP kernel, one workgroup per variable (C-ABI
    lpbox_policy_score_f32): the reference's float32 arithmetic on the device without torch in the path.  Slow by design (FMA
    loops, ~20x the fused fp16 encoder); FusedEarlyFixPolicy uses it for the rows whose score lies near a fixing threshold, and the
    tests use it as the fp32 check of the fused kernel.  Agreement with the reference module: rounding (tested at 1e-4 on the
    sigmoid against the reference-generated golden vectors, 2e-5 against EarlyFixPolicy)."""

    def __init__(self, state_dict, tokens=20, device="cuda"):
        import ctypes as C
        from . import _lib
        self._L, self._check, self._C = _lib.load(), _lib.check, C
        ref = EarlyFixPolicy(state_dict, tokens=tokens, device="cpu")          # validates names/shapes, folds BN and the position code
        self.tokens, self.device = tokens, torch.device(device)
        parts = [ref.w_in, ref.b_in]
        for L in ref.layers:
            parts += [L["w_qkv"], L["w_o"], L["n1_s"], L["n1_t"], L["w1"], L["b1"], L["w2"], L["b2"], L["n2_s"], L["n2_t"]]
        for w, b in ref.head:
            parts += [w, b]
        flat = torch.cat([t.reshape(-1).to(torch.float32) for t in parts])
        n = C.c_long()
        self._check(self._L.lpbox_policy_f32_layout(tokens, C.byref(n)), "lpbox_policy_f32_layout")
        assert flat.numel() == n.value, (flat.numel(), n.value)
        self.w = flat.contiguous().to(self.device)

    @classmethod
    def random(cls, tokens=20, seed=0, **kw):
        return cls(random_state(tokens, seed), tokens=tokens, **kw)

    @torch.no_grad()
    def scores_from_xiters(self, flat, row_off, tok_stride=None, logits=False):
        rows = int(row_off.numel())
        sig = torch.empty(rows, device=self.device, dtype=torch.float32)
        lg = torch.empty(rows, device=self.device, dtype=torch.float32) if logits else None
        if rows:
            ts = CODE_DIM if tok_stride is None else int(tok_stride)
            if int(row_off.max().item()) + (self.tokens - 1) * ts + CODE_DIM > flat.numel():
                raise ValueError("row offsets reach past the end of the iterate buffer")
            stream = torch.cuda.current_stream(self.device).cuda_stream
            row_off = row_off.contiguous()
            self._check(self._L.lpbox_policy_score_f32(flat.data_ptr(), row_off.data_ptr(), rows, self.tokens, ts, self.w.data_ptr(),
                                                       sig.data_ptr(), lg.data_ptr() if logits else None, stream), "lpbox_policy_score_f32")
        return (sig, lg) if logits else sig

    @torch.no_grad()
    def rescore_band(self, flat, row_off, sig, band, thresholds, tok_stride=None, counter=None):
        """In place: the entries of `sig` (float32 CUDA) within `band` of a threshold are replaced by the fp32 evaluation; the
        selection runs on the device (no host synchronisation).  `counter`: optional int64 CUDA tensor (1,) incremented per re-scored row."""
        rows = int(row_off.numel())
        if rows:
            ts = CODE_DIM if tok_stride is None else int(tok_stride)
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._L.lpbox_policy_rescore_f32(flat.data_ptr(), row_off.contiguous().data_ptr(), rows, self.tokens, ts, self.w.data_ptr(),
                                                         sig.data_ptr(), float(band), float(thresholds[0]), float(thresholds[1]),
                                                         counter.data_ptr() if counter is not None else None, stream), "lpbox_policy_rescore_f32")
        return sig

    def __call__(self, x):
        x = x.to(self.device, torch.float64).contiguous()
        if x.dim() != 3 or x.shape[1] != self.tokens or x.shape[2] != CODE_DIM:
            raise ValueError("expected (rows, %d, %d), got %s" % (self.tokens, CODE_DIM, tuple(x.shape)))
        off = torch.arange(x.shape[0], device=self.device, dtype=torch.int64) * (self.tokens * CODE_DIM)
        return self.scores_from_xiters(x.reshape(-1), off, CODE_DIM)


class FusedEarlyFixPolicy:
    """Same scores as EarlyFixPolicy, computed by the fused fp16-MFMA encoder kernel + a half-precision head.

    `policy.scores_from_xiters(flat, row_off, rows)` reads the solver's fp64 x_iters buffer in place; `policy(x)` accepts the
    reference's (rows, tokens, 5) float input for convenience.  Numerics: fp16 operands, fp32 accumulation -- within 2e-4 of the
    fp32 network on the sigmoid for weights drawn like the reference initialises them, within 2e-2 on the formula-weight stress
    fixture (both tested, tests/test_policy.py).  The callers threshold the score at 0.9 / 0.1 (deter_fix_2, LP/trainer.py:101-135),
    so a score within that error of a threshold could fix a different variable set than the reference's fp32 arithmetic: every
    row whose fp16 score lies within `decision_band` of 0.9 or 0.1 is therefore RE-SCORED in fp32 by the plain HIP kernel of the same
    library (HipFp32Policy, a few rows per window), which makes the fix decisions those of the fp32 network (tested).
    decision_band = 0 turns the re-scoring off (pure fp16 fast mode)."""

    def __init__(self, state_dict, tokens=20, device="cuda", chunk_rows=262144, decision_band=3e-2, thresholds=(0.9, 0.1)):
        from . import _lib
        self.chunk_rows = int(chunk_rows)      # bounds the flattened activation buffer (rows x tokens*128 fp16 = 1.3 GB at 20 tokens)
        self._L = _lib.load()
        self._check = _lib.check
        ref = EarlyFixPolicy(state_dict, tokens=tokens, device="cpu")          # validates names/shapes, folds BN and the position code
        self.tokens, self.device = tokens, torch.device(device)
        import ctypes as C
        wh, cf = C.c_long(), C.c_long()
        self._check(self._L.lpbox_policy_layout(tokens, C.byref(wh), C.byref(cf)), "lpbox_policy_layout")
        frags, consts = [], [ref.w_in.reshape(-1), ref.b_in.reshape(-1)]
        for L in ref.layers:
            wq, wk, wv = L["w_qkv"][:, :EMBED], L["w_qkv"][:, EMBED:2 * EMBED], L["w_qkv"][:, 2 * EMBED:]
            for half in range(2):
                c = slice(64 * half, 64 * half + 64)                            # heads 4 half .. 4 half + 3
                frags.append(_pack_fragments(torch.cat([wq[:, c] * 0.25, wk[:, c], wv[:, c]], dim=1)))   # 1/sqrt(16) folded into Q (exact)
            frags.append(_pack_fragments(L["w_o"]))
            for c in range(4):
                frags.append(_pack_fragments(L["w1"][:, 128 * c:128 * c + 128]))
            for c in range(4):
                frags.append(_pack_fragments(L["w2"][128 * c:128 * c + 128, :]))
            consts += [L["n1_s"], L["n1_t"], L["b1"], L["b2"], L["n2_s"], L["n2_t"]]
        w = torch.cat(frags)
        c = torch.cat([t.reshape(-1).to(torch.float32) for t in consts])
        assert w.numel() == wh.value and c.numel() == cf.value, (w.numel(), wh.value, c.numel(), cf.value)
        self.w = w.contiguous().to(self.device)
        self.c = c.contiguous().to(self.device)
        self.head = [(wt.to(self.device, torch.float16), b.to(self.device, torch.float16)) for wt, b in ref.head]
        self.decision_band, self.thresholds = float(decision_band), tuple(thresholds)
        self.ref32 = HipFp32Policy(state_dict, tokens=tokens, device=device) if self.decision_band > 0 else None   # fp32 HIP kernel, same library
        self._rescored = torch.zeros(1, dtype=torch.int64, device=self.device)

    @classmethod
    def random(cls, tokens=20, seed=0, **kw):
        return cls(random_state(tokens, seed), tokens=tokens, **kw)

    @torch.no_grad()
    def encode(self, flat, row_off, tok_stride):
        """flat: fp64 CUDA tensor; row_off: int64 CUDA tensor (rows,) of offsets into flat -> fp16 (rows, tokens*128)."""
        rows = int(row_off.numel())
        out = torch.empty((rows, self.tokens * EMBED), device=self.device, dtype=torch.float16)
        if rows:
            need = int(row_off.max().item()) + (self.tokens - 1) * tok_stride + CODE_DIM
            if need > flat.numel():
                raise ValueError("row offsets reach past the end of the iterate buffer")
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self._check(self._L.lpbox_policy_encode_f16(flat.data_ptr(), row_off.data_ptr(), rows, self.tokens, int(tok_stride),
                                                        self.w.data_ptr(), self.c.data_ptr(), out.data_ptr(), stream),
                        "lpbox_policy_encode_f16")
        return out

    @torch.no_grad()
    def logits_from_xiters(self, flat, row_off, tok_stride=None):
        out = torch.empty(row_off.numel(), device=self.device, dtype=torch.float32)
        for r0 in range(0, row_off.numel(), self.chunk_rows):
            z = self.encode(flat, row_off[r0:r0 + self.chunk_rows], CODE_DIM if tok_stride is None else tok_stride)
            for k, (w, b) in enumerate(self.head):
                z = torch.addmm(b, z, w)
                if k < 3:
                    z = torch.relu(z)
            out[r0:r0 + self.chunk_rows] = z.reshape(-1)
        return out

    @torch.no_grad()
    def scores_from_xiters(self, flat, row_off, tok_stride=None):
        sig = torch.sigmoid(self.logits_from_xiters(flat, row_off, tok_stride))
        if self.ref32 is not None and sig.numel():      # decisions near a threshold: the reference's fp32 arithmetic decides (selected on the device)
            self.ref32.rescore_band(flat, row_off, sig, self.decision_band, self.thresholds, tok_stride, self._rescored)
        return sig

    @property
    def rescored(self):
        """Rows re-scored in fp32 so far (diagnostic; reading it synchronises)."""
        return int(self._rescored.item()) if self.ref32 is not None else 0

    def logits(self, x):
        x = x.to(self.device, torch.float64).contiguous()
        if x.dim() != 3 or x.shape[1] != self.tokens or x.shape[2] != CODE_DIM:
            raise ValueError("expected (rows, %d, %d), got %s" % (self.tokens, CODE_DIM, tuple(x.shape)))
        off = torch.arange(x.shape[0], device=self.device, dtype=torch.int64) * (self.tokens * CODE_DIM)
        return self.logits_from_xiters(x.reshape(-1), off, CODE_DIM)

    def __call__(self, x):
        x = x.to(self.device, torch.float64).contiguous()
        if x.dim() != 3 or x.shape[1] != self.tokens or x.shape[2] != CODE_DIM:
            raise ValueError("expected (rows, %d, %d), got %s" % (self.tokens, CODE_DIM, tuple(x.shape)))
        off = torch.arange(x.shape[0], device=self.device, dtype=torch.int64) * (self.tokens * CODE_DIM)
        return self.scores_from_xiters(x.reshape(-1), off, CODE_DIM)
