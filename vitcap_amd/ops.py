"""Tensor-level wrappers over the C ABI (one function per entry point of include/vitcap_hip.h).

PyTorch is plumbing only: it owns device memory and the HIP stream; every computation is done by the
hand-written kernels in libvitcap_hip.so.  All functions enqueue on ``torch.cuda.current_stream()``.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import lib, check


# Optional live timing of the large GEMM launches of the TRAINING path (bench.py --mode train roofline): when TIMING is a list,
# every gemm_ex / gemm_tn launch with M > 256 is bracketed by events recorded on the launch stream (the current torch stream is
# the stream the kernel is launched on) and appended as (kind, flops, start, stop).
TIMING = None


class _Timed(object):
    def __init__(self, kind, flops, big):
        self.on = TIMING is not None and big
        self.kind, self.flops = kind, flops

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e1 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            TIMING.append((self.kind, self.flops, self.e0, self.e1))


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _dev_bf16(t):
    assert t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous(), 'expect contiguous cuda bf16 tensor'


def _dev_f32(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), 'expect contiguous cuda fp32 tensor'


def gemm_bias_act(a, w, bias=None, residual=None, act=L.ACT_NONE, out_dtype=torch.bfloat16, out=None,
                  row_group=0, out_group_rows=0, out_row_off=0, res_periodic=0, out_rows=None, tile_hint=0, split_k=0):
    """C = act(A @ W.T + bias) (+ residual).  a [M,K] bf16, w [N,K] bf16 (nn.Linear layout)."""
    _dev_bf16(a)
    _dev_bf16(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        shape = (split_k, M, N) if split_k > 1 else (out_rows or M, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if split_k > 1 else out_dtype)
    d = L.GemmDesc(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(-2),
                   ldr=residual.stride(0) if residual is not None else 0, act=act,
                   out_dtype=L.OUT_F32 if out.dtype == torch.float32 else L.OUT_BF16,
                   row_group=row_group, out_group_rows=out_group_rows, out_row_off=out_row_off,
                   res_periodic=res_periodic, tile_hint=tile_hint, split_k=split_k)
    check(lib.vitcap_gemm_bias_act(_p(a), _p(w), _p(bias), _p(residual), _p(out), C.byref(d), _stream()), 'gemm')
    return out


def gemm_layernorm(a, w, bias, residual, gamma, beta, eps, out=None, ln_out=None, want_f32=False, counters=None, tile_hint=0):
    """x = A @ W.T + bias (+ residual) in fp32 [M, 768], and LayerNorm(x) -> bf16 [M, 768] (and fp32 with want_f32): the residual
    GEMMs of a ViT / BERT block with the LayerNorm behind them (vitcap_gemm_desc.ln_*).  `counters`: int32 [>= M // 128 + 8], zero --
    with them (and a launch form that supports it: tile_hint 5 at M >= 2048) the kernel normalises its own rows; without, a
    LayerNorm launch follows.  `out` may be `residual` (in place); `ln_out` may alias `a`.  Returns (x, ln_bf16, ln_f32 or None)."""
    _dev_bf16(a)
    _dev_bf16(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    if ln_out is None:
        ln_out = torch.empty((M, N), device=a.device, dtype=torch.bfloat16)
    ln_f = torch.empty((M, N), device=a.device, dtype=torch.float32) if want_f32 else None
    d = L.GemmDesc(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(0),
                   ldr=residual.stride(0) if residual is not None else 0, act=L.ACT_NONE, out_dtype=L.OUT_F32, tile_hint=tile_hint,
                   ln_gamma=_p(gamma).value, ln_beta=_p(beta).value, ln_eps=float(eps), ln_out_bf16=ln_out.data_ptr(),
                   ln_out_f32=ln_f.data_ptr() if want_f32 else None, ln_counters=counters.data_ptr() if counters is not None else None)
    check(lib.vitcap_gemm_bias_act(_p(a), _p(w), _p(bias), _p(residual), _p(out), C.byref(d), _stream()), 'gemm_layernorm')
    return out, ln_out, ln_f


def gemm_tile_plan(M, N, K):
    """(256-row m-tiles, height class of the tiles behind them: 0 none / 3 = 192 rows / 2 = 128 rows, their count) of the
    256-column GEMM kernel for this problem on the current device (host-side query)."""
    plan = (C.c_int * 3)()
    check(lib.vitcap_gemm_tile_plan(int(M), int(N), int(K), plan), 'gemm_tile_plan')
    return int(plan[0]), int(plan[1]), int(plan[2])


def layernorm(x, gamma, beta, eps, want_bf16=True, want_f32=False):
    _dev_f32(x)
    M, D = x.shape
    yb = torch.empty((M, D), device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    yf = torch.empty((M, D), device=x.device, dtype=torch.float32) if want_f32 else None
    check(lib.vitcap_layernorm_fwd(_p(x), x.stride(0), _p(gamma), _p(beta), eps, _p(yb), _p(yf), M, D, _stream()),
          'layernorm')
    return yb, yf


def sum_layernorm(partials, bias, residual, gamma, beta, eps, act_before_ln=False, want_bf16=True, want_f32=True):
    """LayerNorm(sum_s partials[s] + bias (+ residual)) -- consumer of a split-K GEMM."""
    _dev_f32(partials)
    S, M, D = partials.shape
    yb = torch.empty((M, D), device=partials.device, dtype=torch.bfloat16) if want_bf16 else None
    yf = torch.empty((M, D), device=partials.device, dtype=torch.float32) if want_f32 else None
    check(lib.vitcap_sum_layernorm(_p(partials), S, partials.stride(0), _p(bias), _p(residual),
                                   residual.stride(0) if residual is not None else 0, int(act_before_ln), _p(gamma),
                                   _p(beta), eps, _p(yb), _p(yf), M, D, _stream()), 'sum_layernorm')
    return yb, yf


def patch_gather(image):
    assert image.is_cuda and image.is_contiguous() and image.shape[1:] == (3, 384, 384)
    B = image.shape[0]
    out = torch.empty((B * 576, 768), device=image.device, dtype=torch.bfloat16)
    check(lib.vitcap_patch_gather(_p(image), int(image.dtype == torch.bfloat16), _p(out), B, _stream()), 'patch_gather')
    return out


def attn_dense(qkv, B, S, scale=0.125):
    _dev_bf16(qkv)
    assert qkv.shape == (B * S, 2304)
    out = torch.empty((B * S, 768), device=qkv.device, dtype=torch.bfloat16)
    check(lib.vitcap_attn_dense_fwd(_p(qkv), _p(out), B, S, scale, _stream()), 'attn_dense')
    return out


def attn_dense_train(qkv, B, S, scale=0.125, ld_rows=None, out=None, p_drop=0.0, drop_seed=0, causal_from=0, mask_from=0,
                     q_range=None):
    """forward that also returns the log2-domain logsumexp (B,12,S) needed by attn_dense_bwd; ld_rows = rows per image
    in the buffers (>= S; only the first S rows of each image are attended / written).  q_range = (lo, hi): only the query
    rows of the 128-row blocks covering [lo, hi) are computed (lo % 128 == 0); the other rows of out / lse are zeros."""
    _dev_bf16(qkv)
    ld_rows = ld_rows or S
    lo, hi = q_range or (0, S)
    alloc = torch.zeros if q_range else torch.empty
    if out is None:
        out = alloc((B * ld_rows, 768), device=qkv.device, dtype=torch.bfloat16)
    lse = alloc((B, 12, S), device=qkv.device, dtype=torch.float32)
    check(lib.vitcap_attn_dense_fwd_train_rows(_p(qkv), _p(out), _p(lse), B, S, ld_rows, scale, p_drop, drop_seed, causal_from,
                                               mask_from, lo, hi, _stream()),
          'attn_dense_train')
    return out, lse


def attn_dense_bwd(qkv, out, dout, lse, B, S, scale=0.125, extra_dkv=None, ld_rows=None, dqkv=None, p_drop=0.0,
                   drop_seed=0, causal_from=0, mask_from=0, q_range=None):
    """q_range = (lo, hi): the forward ran for those query rows only; `dout` must be zero outside them (inside the 64-row
    tiles that intersect the range).  dQ outside the range's 128-row blocks is zero-filled here."""
    _dev_bf16(qkv); _dev_bf16(out); _dev_bf16(dout); _dev_f32(lse)
    ld_rows = ld_rows or S
    lo, hi = q_range or (0, S)
    if dqkv is None:
        dqkv = torch.zeros_like(qkv) if q_range else torch.empty_like(qkv)
    dsum = torch.zeros_like(lse) if q_range else torch.empty_like(lse)
    check(lib.vitcap_attn_dense_bwd_rows(_p(qkv), _p(out), _p(dout), _p(lse), _p(dsum), _p(extra_dkv), _p(dqkv), B, S, ld_rows,
                                         scale, p_drop, drop_seed, causal_from, mask_from, lo, hi, _stream()), 'attn_dense_bwd')
    return dqkv


def attn_decode_step(qkv_step, vis_qkv, text_kv, B, S_vis, t, max_len=20, seq_per_image=1, scale=0.125):
    _dev_bf16(qkv_step)
    _dev_bf16(vis_qkv)
    _dev_bf16(text_kv)
    out = torch.empty((B * 2, 768), device=qkv_step.device, dtype=torch.bfloat16)
    check(lib.vitcap_attn_decode_step(_p(qkv_step), _p(vis_qkv), _p(text_kv), _p(out), B, S_vis, t, max_len,
                                      seq_per_image, scale, _stream()), 'attn_decode')
    return out


def attn_decode_beams(qkv_step, vis_qkv, text_kv, n_images, K, S_vis, t, max_len=20, scale=0.125):
    """Decode-step attention for K sequences per image on the matrix pipe (vitcap_attn_decode_beams); builds V^T first."""
    _dev_bf16(qkv_step)
    _dev_bf16(vis_qkv)
    _dev_bf16(text_kv)
    vt = torch.empty((n_images, 12, 64, 608), device=qkv_step.device, dtype=torch.bfloat16)
    check(lib.vitcap_attn_beam_vt(_p(vis_qkv), _p(vt), n_images, S_vis, _stream()), 'attn_beam_vt')
    out = torch.empty((n_images * K * 2, 768), device=qkv_step.device, dtype=torch.bfloat16)
    check(lib.vitcap_attn_decode_beams(_p(qkv_step), _p(vis_qkv), _p(vt), _p(text_kv), _p(out), n_images, K, S_vis, t, max_len,
                                       scale, _stream()), 'attn_decode_beams')
    return out


def embed_step(ids, t, word, pos, typ, gamma, beta, eps=1e-12, mask_token=103):
    B, max_len = ids.shape
    xf = torch.empty((2 * B, 768), device=ids.device, dtype=torch.float32)
    xb = torch.empty((2 * B, 768), device=ids.device, dtype=torch.bfloat16)
    check(lib.vitcap_embed_step(_p(ids), max_len, t, mask_token, _p(word), _p(pos), _p(typ), _p(gamma), _p(beta), eps,
                                _p(xf), _p(xb), B, _stream()), 'embed_step')
    return xf, xb


def greedy_init(B, max_len=20, device='cuda', bos=101, pad=0):
    st = dict(ids=torch.empty((B, max_len), device=device, dtype=torch.int64),
              unf=torch.empty((B,), device=device, dtype=torch.int32),
              sum_lp=torch.empty((B,), device=device, dtype=torch.float32),
              cnt=torch.empty((B,), device=device, dtype=torch.float32),
              logprob=torch.zeros((B,), device=device, dtype=torch.float32),
              margin=torch.zeros((B, max_len), device=device, dtype=torch.float32))
    check(lib.vitcap_greedy_init(_p(st['ids']), _p(st['unf']), _p(st['sum_lp']), _p(st['cnt']), B, max_len, bos, pad,
                                 _stream()), 'greedy_init')
    return st


def greedy_step(logits, st, t, V=L.VOCAB, eos=102, pad=0):
    _dev_f32(logits)
    B, max_len = st['ids'].shape
    check(lib.vitcap_greedy_step(_p(logits), logits.stride(0), V, _p(st['ids']), _p(st['unf']), _p(st['sum_lp']),
                                 _p(st['cnt']), _p(st['logprob']), _p(st['margin']), _p(st.get('raw_last')), B, t, max_len, eos,
                                 pad, _stream()), 'greedy_step')


def gemm_rowstat(a, w, bias):
    """Vocabulary GEMM with row statistics: (logits fp32 [M][N], rowstat fp32 [M][2*ceil(N/64)][4])."""
    _dev_bf16(a)
    _dev_bf16(w)
    M, K = a.shape
    N = w.shape[0]
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    pieces = 2 * ((N + 63) // 64)
    rs = torch.zeros((M, pieces, 4), device=a.device, dtype=torch.float32)
    d = L.GemmDesc(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(0), act=L.ACT_NONE, out_dtype=L.OUT_F32,
                   rowstat=rs.data_ptr())
    check(lib.vitcap_gemm_bias_act(_p(a), _p(w), _p(bias), None, _p(out), C.byref(d), _stream()), 'gemm(rowstat)')
    return out, rs


def row_topk(logits, V, k, rowstat=None):
    """The k largest logits of every row (sorted, lowest column on ties) + logsumexp over the V columns: from the whole rows
    (vitcap_row_topk_lse) or, given the vocabulary GEMM's row statistics, from the k best 32-column pieces."""
    rows = logits.shape[0]
    val = torch.empty((rows, k), device=logits.device, dtype=torch.float32)
    idx = torch.empty((rows, k), device=logits.device, dtype=torch.int32)
    lse = torch.empty((rows,), device=logits.device, dtype=torch.float32)
    if rowstat is None:
        check(lib.vitcap_row_topk_lse(_p(logits), logits.stride(0), V, k, _p(val), _p(idx), _p(lse), rows, _stream()), 'row_topk_lse')
    else:
        check(lib.vitcap_row_topk_pieces(_p(logits), logits.stride(0), V, _p(rowstat), rowstat.shape[1], k, _p(val), _p(idx), _p(lse),
                                         rows, _stream()), 'row_topk_pieces')
    return val, idx, lse


def greedy_select_embed(rowstat, st, t, word, pos, typ, gamma, beta, eps=1e-12, mask_token=103, eos=102, pad=0):
    """greedy_step from row statistics + embed_step for t+1 in one launch; returns (x_f32, x_bf16) of step t+1 (None at the end)."""
    B, max_len = st['ids'].shape
    lastp = t == max_len - 1
    xf = None if lastp else torch.empty((2 * B, 768), device=rowstat.device, dtype=torch.float32)
    xb = None if lastp else torch.empty((2 * B, 768), device=rowstat.device, dtype=torch.bfloat16)
    check(lib.vitcap_greedy_select_embed(_p(rowstat), rowstat.shape[1], _p(st['ids']), _p(st['unf']), _p(st['sum_lp']), _p(st['cnt']),
                                         _p(st['logprob']), _p(st.get('raw_last')), B, t, max_len, eos, pad, mask_token, _p(word),
                                         _p(pos), _p(typ), _p(gamma), _p(beta), eps, _p(xf), _p(xb), _stream()), 'greedy_select_embed')
    return xf, xb


def sample_step(logits, st, t, temperature=1.0, top_k=0, top_p=1.0, seed=0, V=L.VOCAB, eos=102, pad=0):
    """do_sample variant of greedy_step (modeling_utils.py:839-851): same state dict, one draw per sequence."""
    from ._lib import SampleParams
    _dev_f32(logits)
    B, max_len = st['ids'].shape
    sp = SampleParams(1, float(temperature), int(top_k), float(top_p), int(seed) & 0xffffffff)
    check(lib.vitcap_sample_step(_p(logits), logits.stride(0), V, _p(st['ids']), _p(st['unf']), _p(st['sum_lp']),
                                 _p(st['cnt']), _p(st['logprob']), _p(st['margin']), _p(st.get('raw_last')), B, t, max_len, eos,
                                 pad, C.byref(sp), _stream()), 'sample_step')


def sigmoid_topk(logits, k=50, thresh=0.2, V=None):
    _dev_f32(logits)
    B = logits.shape[0]
    V = V or logits.shape[1]
    ids = torch.empty((B, k), device=logits.device, dtype=torch.int64)
    prob = torch.empty((B, k), device=logits.device, dtype=torch.float32)
    ln = torch.empty((B,), device=logits.device, dtype=torch.int64)
    check(lib.vitcap_sigmoid_topk(_p(logits), logits.stride(0), V, k, thresh, _p(ids), _p(prob), _p(ln), B, _stream()),
          'sigmoid_topk')
    return ids, prob, ln


# ------------------------------------------------------------------------------------------------ training ops
def gemm_ex(a, w, bias=None, residual=None, act=L.ACT_NONE, out=None, out_dtype=torch.bfloat16, aux=None, zout=None,
            split_k=0, tile_hint=0, colsum=None):
    """GEMM with the training extras of vitcap_gemm_ex (gelu' multiply, pre-activation output, ragged split-K slabs;
    colsum (fp32 [N]) += column sums of the bf16 output = the bias gradient of the layer this is the output gradient of)."""
    _dev_bf16(a)
    _dev_bf16(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        shape = (split_k, M, N) if split_k > 1 else (M, N)
        out = torch.empty(shape, device=a.device, dtype=torch.float32 if split_k > 1 else out_dtype)
    d = L.GemmDesc(M=M, N=N, K=K, lda=a.stride(0), ldw=w.stride(0), ldc=out.stride(-2),
                   ldr=residual.stride(0) if residual is not None else 0, act=act,
                   out_dtype=L.OUT_F32 if out.dtype == torch.float32 else L.OUT_BF16, tile_hint=tile_hint, split_k=split_k,
                   colsum=colsum.data_ptr() if colsum is not None else None)
    with _Timed('gemm_nt (forward / input gradients)', 2.0 * M * N * K, M > 256):
        check(lib.vitcap_gemm_ex(_p(a), _p(w), _p(bias), _p(residual), _p(out), C.byref(d), _p(aux),
                                 aux.stride(0) if aux is not None else 0, _p(zout), zout.stride(0) if zout is not None else 0,
                                 _stream()), 'gemm_ex')
    return out


def transpose_colsum(x, colsum=None, out=None):
    """x bf16 [R][C] -> x^T bf16 [C][Rp] (Rp = R rounded up to 64, zero padded); colsum[c] += sum_r x[r][c]."""
    _dev_bf16(x)
    R, Cc = x.shape
    Rp = (R + 63) // 64 * 64
    if out is None:
        out = torch.empty((Cc, Rp), device=x.device, dtype=torch.bfloat16)
    check(lib.vitcap_transpose_colsum(_p(x), x.stride(0), _p(out), out.stride(0), _p(colsum), R, Cc, _stream()), 'transpose')
    return out


def layernorm_bwd(x, dy, gamma, eps, dgamma, dbeta, dres=None, want_f32=True, want_bf16=True, dxb_colsum=None):
    """dxb_colsum (fp32 [768]) += column sums of the bf16 result (bias gradient of the layer dx is the output gradient of)."""
    _dev_f32(x)
    M, D = x.shape
    dxf = torch.empty((M, D), device=x.device, dtype=torch.float32) if want_f32 else None
    dxb = torch.empty((M, D), device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    check(lib.vitcap_layernorm_bwd(_p(x), x.stride(0), _p(dy), int(dy.dtype == torch.float32), _p(gamma), eps, _p(dres),
                                   _p(dxf), _p(dxb), _p(dgamma), _p(dbeta), _p(dxb_colsum), M, D, _stream()), 'layernorm_bwd')
    return dxf, dxb


def reduce_slabs(slabs, out, accumulate=False):
    S = slabs.shape[0]
    n = slabs[0].numel()
    check(lib.vitcap_reduce_slabs(_p(slabs), slabs.stride(0), S, _p(out), n, int(accumulate), _stream()), 'reduce_slabs')
    return out


def cast_bf16(x):
    _dev_f32(x)
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib.vitcap_cast_bf16(_p(x), _p(y), x.numel(), _stream()), 'cast_bf16')
    return y


def hidden_dropout(x, residual, rows_per_seq, row0, seed, p, out=None):
    """out = x * keep / (1 - p) (+ residual): nn.Dropout(hidden_dropout_prob) of the BERT parts, counter-based mask (csrc/train.hip)."""
    _dev_f32(x)
    M, D = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(lib.vitcap_hidden_dropout(_p(x), _p(residual), _p(out), M, D, int(rows_per_seq), int(row0), int(seed) & 0xffffffff, float(p),
                                    _stream()), 'hidden_dropout')
    return out


def cast_bf16_colsum(x, colsum):
    """bf16 copy of fp32 x [M][768] and colsum[c] += sum_m bf16(x)[m][c] in one pass (cast_bf16 + colsum_bf16)."""
    _dev_f32(x); _dev_f32(colsum)
    assert x.dim() == 2 and x.is_contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib.vitcap_cast_bf16_colsum(_p(x), _p(y), _p(colsum), x.shape[0], x.shape[1], _stream()), 'cast_bf16_colsum')
    return y



def gemm_tn(y, x, splits, slabs=None):
    """dW slabs [splits][N][K] fp32 = (pieces of) y^T x for bf16 y [M][N], x [M][K] (row-major, as the backward pass holds them)"""
    _dev_bf16(y); _dev_bf16(x)
    M, N = y.shape
    K = x.shape[1]
    assert x.shape[0] == M
    if slabs is None:
        slabs = torch.empty((splits, N, K), device=y.device, dtype=torch.float32)
    with _Timed('gemm_tn (weight gradients)', 2.0 * M * N * K, True):
        check(lib.vitcap_gemm_tn(_p(y), y.stride(0), _p(x), x.stride(0), _p(slabs), M, N, K, splits, _stream()), 'gemm_tn')
    return slabs


def gemm_tn_sum(y, x, splits, out, accumulate=False, slabs=None):
    """out[N][K] (+)= y^T x with the sum over the `splits` pieces of M done inside the launch (vitcap_gemm_tn_sum): the result of
    gemm_tn + reduce_slabs bit for bit, one launch; tiles x splits must fit the CUs."""
    _dev_bf16(y); _dev_bf16(x); _dev_f32(out)
    M, N = y.shape
    K = x.shape[1]
    assert x.shape[0] == M and tuple(out.shape) == (N, K) and out.is_contiguous()
    if slabs is None:
        slabs = torch.empty((splits, N, K), device=y.device, dtype=torch.float32)
    with _Timed('gemm_tn (weight gradients)', 2.0 * M * N * K, True):
        check(lib.vitcap_gemm_tn_sum(_p(y), y.stride(0), _p(x), x.stride(0), _p(slabs), _p(out), int(accumulate), M, N, K, splits,
                                     _stream()), 'gemm_tn_sum')
    return out


def colsum_bf16(y, out):
    """out[n] += sum_m y[m][n]"""
    _dev_bf16(y); _dev_f32(out)
    check(lib.vitcap_colsum_bf16(_p(y), y.stride(0), y.shape[0], y.shape[1], _p(out), _stream()), 'colsum')
    return out
