"""Parity tests of every HIP kernel against the CPU oracle / plain fp32 torch, through the C ABI.
Run on the GPU box:  python -m pytest tests -m gpu -x -q

Tolerances (stated per test): kernels take bf16 operands and accumulate in fp32; the reference value is
computed in fp32 on the SAME bf16-rounded operands, so the only differences are fp32 summation order
(~1e-6 relative) plus one bf16 rounding of the output where the kernel stores bf16 (2^-9 relative).
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available(), 'GPU tests need a GPU'
    from vitcap_amd import ops as o
    return o


def _bf(t):
    return t.to(torch.bfloat16)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _close(got, want, rtol, atol, what=''):
    got = got.detach().float().cpu()
    want = want.detach().float().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    bad = err > tol
    assert not bad.any(), '%s: %d/%d elements off, max err %.3e (want max %.3e), first bad idx %s' % (
        what, int(bad.sum()), bad.numel(), float(err.max()), float(want.abs().max()),
        bad.nonzero()[:4].tolist())


@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (300, 2304, 768), (1154, 768, 3072), (64, 768, 768),
                                   (130, 3072, 768), (5, 128, 128)])
def test_gemm_plain_f32_out(ops, M, N, K):
    from vitcap_amd import _lib as L
    a = _bf(_rand((M, K), 1)).cuda()
    w = _bf(_rand((N, K), 2, 0.05)).cuda()
    bias = _rand((N,), 3).cuda()
    got = ops.gemm_bias_act(a, w, bias, act=L.ACT_NONE, out_dtype=torch.float32)
    want = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    # fp32 accumulation of exact bf16 products: order-only differences
    _close(got, want, 1e-4, 1e-4, 'gemm f32 %dx%dx%d' % (M, N, K))


def test_gemm_transpose_detecting(ops):
    """A = shifted identity, asymmetric W: catches a transposed/permuted C write (guide rule 16)."""
    M = N = K = 128
    a = torch.zeros(M, K)
    a[torch.arange(M), (torch.arange(M) * 7 + 3) % K] = 1.0
    w = torch.arange(N * K, dtype=torch.float32).reshape(N, K) % 251 - 125.0
    got = ops.gemm_bias_act(_bf(a).cuda(), _bf(w).cuda(), None, out_dtype=torch.float32)
    want = _bf(a).float() @ _bf(w).float().t()
    _close(got, want, 0, 1e-3, 'gemm permutation')


@pytest.mark.parametrize('act', ['gelu', 'tanh'])
def test_gemm_act_bf16_out(ops, act):
    from vitcap_amd import _lib as L
    M, N, K = 577, 3072, 768
    a = _bf(_rand((M, K), 4)).cuda()
    w = _bf(_rand((N, K), 5, 0.05)).cuda()
    bias = _rand((N,), 6, 0.1).cuda()
    got = ops.gemm_bias_act(a, w, bias, act=L.ACT_GELU_ERF if act == 'gelu' else L.ACT_TANH)
    z = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    want = torch.nn.functional.gelu(z) if act == 'gelu' else torch.tanh(z)
    assert got.dtype == torch.bfloat16
    # one bf16 rounding of the output: 2^-8 relative, plus tiny absolute slack
    _close(got, want, 2 ** -7, 2e-3, 'gemm ' + act)


def test_gemm_residual_inplace(ops):
    M, N, K = 1154, 768, 768
    a = _bf(_rand((M, K), 7)).cuda()
    w = _bf(_rand((N, K), 8, 0.05)).cuda()
    bias = _rand((N,), 9, 0.1).cuda()
    x = _rand((M, N), 10).cuda()
    want = a.float().cpu() @ w.float().cpu().t() + bias.cpu() + x.cpu()
    ops.gemm_bias_act(a, w, bias, residual=x, out=x)          # x <- x + a@w.T + b, in place as the engine does
    _close(x, want, 1e-4, 1e-4, 'gemm residual in place')


@pytest.mark.parametrize('variant', ['bias_bf16', 'gelu_bf16', 'res_f32', 'rowmap_res_f32'])
def test_gemm_256_tile_heights_bit_identical(ops, variant):
    """The 256-column kernel with 256-row tiles only (hint 32), with every tile 192 / 128 rows (30 / 31) and with the planned
    mix of 256-row tiles + short tiles in the last round (hint 33 / auto): the same k order per output element and the same
    epilogue arithmetic, so every bit must agree -- and hint 32 is checked against fp32 torch."""
    from vitcap_amd import _lib as L
    M, N, K = (9232, 2304, 768) if variant != 'res_f32' else (10386, 768, 768)       # ragged last tiles; the plan mixes heights here
    if variant == 'rowmap_res_f32':
        M, N, K = 16 * 576, 768, 768
    a = _bf(_rand((M, K), 71)).cuda()
    w = _bf(_rand((N, K), 72, 0.05)).cuda()
    bias = _rand((N,), 73, 0.1).cuda()
    plan = ops.gemm_tile_plan(M, N, K)
    assert plan[1] in (2, 3) and plan[2] > 0 and plan[0] * 256 + plan[2] * 64 * plan[1] >= M, plan
    outs = {}
    for hint in (32, 30, 31, 33, 5, 0):
        if variant == 'bias_bf16':
            outs[hint] = ops.gemm_bias_act(a, w, bias, tile_hint=hint)
        elif variant == 'gelu_bf16':
            outs[hint] = ops.gemm_bias_act(a, w, bias, act=L.ACT_GELU_ERF, tile_hint=hint)
        elif variant == 'res_f32':
            x = _rand((M, N), 74).cuda()
            outs[hint] = ops.gemm_bias_act(a, w, bias, residual=x, out=x.clone(), tile_hint=hint)
        else:   # patch-embed row map: out row = (r / 576) * 577 + 1 + r % 576, residual = pos_embed[1 + r % 576] (periodic)
            pos = _rand((577, N), 75).cuda()
            out = torch.zeros(16 * 577, N, device='cuda')
            outs[hint] = ops.gemm_bias_act(a, w, bias, residual=pos[1:], out=out, row_group=576, out_group_rows=577,
                                           out_row_off=1, res_periodic=1, tile_hint=hint)
    torch.cuda.synchronize()
    z = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
    if variant == 'bias_bf16':
        _close(outs[32], z, 2 ** -7, 2e-3, 'hint 32 vs torch')
    elif variant == 'gelu_bf16':
        _close(outs[32], torch.nn.functional.gelu(z), 2 ** -7, 2e-3, 'hint 32 vs torch')
    elif variant == 'res_f32':
        _close(outs[32], z + _rand((M, N), 74), 1e-4, 1e-4, 'hint 32 vs torch')
    else:
        want = torch.zeros(16 * 577, N)
        want.view(16, 577, N)[:, 1:] = (z.view(16, 576, N) + pos.cpu()[1:])
        _close(outs[32], want, 1e-4, 1e-4, 'hint 32 vs torch')
    for hint in (30, 31, 33, 5, 0):
        assert torch.equal(outs[hint], outs[32]), 'tile_hint %d differs from 256-row tiles (%s)' % (hint, variant)


@pytest.mark.parametrize('shape', [(36928, 2304, 768, 0), (1000, 256, 768, 0), (20000, 3072, 3072, 1), (73856, 768, 2304, 0), (5000, 3072, 768, 1)],
                         ids=lambda t: '%dx%dx%d_act%d' % t)
def test_gemm_4wave_deferred_stores_bit_identical(ops, shape, monkeypatch):
    """DEFER form of the persistent 4-wave kernel (round 6): the upper half of every wave tile's bf16 stores is parked in registers and
    issued behind the NEXT tile's first MFMAs (a workgroup's first tile stores nothing parked: zero-record descriptor; its last tile
    flushes its own half).  Same values to the same addresses: the output must equal the 8-wave kernel's bit for bit -- six rounds of
    tiles per workgroup, one tile per workgroup (flush path only), K = 3072 with GELU, a ragged last m-tile, and a wider output buffer
    whose other columns (and the rows behind M) must keep their sentinel."""
    from vitcap_amd import _lib as L
    M, N, K, act = shape
    monkeypatch.setenv('VITCAP_GEMM_4W_DEFER', '2')          # every bf16 shape (the default policy defers plain outputs below 64k rows only)
    a = _bf(_rand((M, K), 91)).cuda()
    w = _bf(_rand((N, K), 92, 0.05)).cuda()
    bias = _rand((N,), 93, 0.1).cuda()
    outs = {}
    for hint in (32, 42):
        big = torch.full((M + 300, N + 64), -7.0, device='cuda', dtype=torch.bfloat16)        # sentinel around the output window
        ops.gemm_bias_act(a, w, bias, act=L.ACT_GELU_ERF if act else L.ACT_NONE, out=big[:M, :N], tile_hint=hint)
        outs[hint] = big
    torch.cuda.synchronize()
    assert torch.equal(outs[42], outs[32]), 'max |d| %g' % float((outs[42].float() - outs[32].float()).abs().max())
    assert bool((outs[42][M:] == -7.0).all()) and bool((outs[42][:, N:] == -7.0).all())
    assert not bool((outs[42][:M, :N] == -7.0).all())


@pytest.mark.parametrize('variant', ['bias_bf16', 'gelu_bf16', 'res_f32', 'rowmap_res_f32', 'bias_f32', 'res_bf16', 'k128', 'k3072_res_f32',
                                     'ragged_n', 'strided'])
def test_gemm_4wave_bit_identical(ops, variant):
    """The 4-wave / 512-register 256x256 kernel (tile_hint 40 / 41 / 42, csrc/gemm4w.hip) against the 8-wave one (tile_hint 32): same LDS image,
    same MFMA, same k order per output element and the same epilogue arithmetic, so every bit must agree; hint 32 itself is checked
    against fp32 torch by test_gemm_256_tile_heights_bit_identical."""
    from vitcap_amd import _lib as L
    M, N, K = {'bias_bf16': (9232, 2304, 768), 'gelu_bf16': (4099, 3072, 768), 'res_f32': (10386, 768, 768), 'rowmap_res_f32': (16 * 576, 768, 768),
               'bias_f32': (2500, 1024, 256), 'res_bf16': (3000, 768, 768), 'k128': (2048, 512, 128), 'k3072_res_f32': (5000, 768, 3072),
               'ragged_n': (2300, 2304 - 64, 768), 'strided': (4096, 768, 768)}[variant]
    a = _bf(_rand((M, K), 81)).cuda()
    w = _bf(_rand((N, K), 82, 0.05)).cuda()
    bias = _rand((N,), 83, 0.1).cuda()
    outs = {}
    for hint in (32, 40, 41, 42):
        if variant in ('bias_bf16', 'k128', 'ragged_n'):
            outs[hint] = ops.gemm_bias_act(a, w, bias, tile_hint=hint)
        elif variant == 'gelu_bf16':
            outs[hint] = ops.gemm_bias_act(a, w, bias, act=L.ACT_GELU_ERF, tile_hint=hint)
        elif variant == 'bias_f32':
            outs[hint] = ops.gemm_bias_act(a, w, bias, out_dtype=torch.float32, tile_hint=hint)
        elif variant in ('res_f32', 'k3072_res_f32'):
            x = _rand((M, N), 84).cuda()
            outs[hint] = ops.gemm_bias_act(a, w, bias, residual=x, out=x.clone(), tile_hint=hint)
        elif variant == 'res_bf16':
            x = _rand((M, N), 84).cuda()
            outs[hint] = ops.gemm_bias_act(a, w, bias, residual=x, out_dtype=torch.bfloat16, tile_hint=hint)
        elif variant == 'strided':       # the output is a column slice of a wider buffer (ldc > N)
            big = torch.zeros(M, 2 * N, device='cuda', dtype=torch.bfloat16)
            outs[hint] = ops.gemm_bias_act(a, w, bias, out=big[:, N:], tile_hint=hint).clone()
        else:
            pos = _rand((577, N), 75).cuda()
            out = torch.zeros(16 * 577, N, device='cuda')
            outs[hint] = ops.gemm_bias_act(a, w, bias, residual=pos[1:], out=out, row_group=576, out_group_rows=577,
                                           out_row_off=1, res_periodic=1, tile_hint=hint)
    torch.cuda.synchronize()
    if variant in ('bias_bf16', 'bias_f32', 'k128', 'ragged_n'):
        z = a.float().cpu() @ w.float().cpu().t() + bias.cpu()
        tol = (2 ** -7, 2e-3) if outs[40].dtype == torch.bfloat16 else (1e-4, 1e-4)
        _close(outs[40], z, tol[0], tol[1], 'hint 40 vs torch (%s)' % variant)
    for hint in (40, 41, 42):     # LDS epilogue / register epilogue / persistent pipeline
        assert torch.equal(outs[hint], outs[32]), 'tile_hint %d differs from the 8-wave 256x256 kernel (%s): max |d| %g' % (
            hint, variant, float((outs[hint].float() - outs[32].float()).abs().max()))


@pytest.mark.parametrize('M,N,K,split', [(128, 768, 768, 6), (128, 768, 3072, 12), (64, 768, 768, 6), (128, 2304, 768, 1),
                                          (64, 30592, 768, 1), (100, 3072, 768, 1), (256, 768, 3072, 4)])
def test_gemm_skinny_splitk(ops, M, N, K, split):
    """Register-fed skinny kernel (decode-step GEMMs); with split-K the partial slabs must sum to the product."""
    a = _bf(_rand((M, K), 40)).cuda()
    w = _bf(_rand((N, K), 41, 0.05)).cuda()
    bias = _rand((N,), 42).cuda()
    want = a.float().cpu() @ w.float().cpu().t()
    if split > 1:
        parts = ops.gemm_bias_act(a, w, None, split_k=split)
        assert parts.shape == (split, M, N)
        _close(parts.sum(0), want, 1e-4, 1e-4, 'split-K partial sum')
    else:
        got = ops.gemm_bias_act(a, w, bias, out_dtype=torch.float32, tile_hint=4)
        _close(got, want + bias.cpu(), 1e-4, 1e-4, 'skinny')


@pytest.mark.parametrize('M,N,K,hint,act', [(128, 2304, 768, 20, 'none'), (128, 3072, 768, 20, 'gelu'), (128, 768, 768, 21, 'none'),
                                            (128, 768, 3072, 20, 'slabs'), (64, 768, 768, 21, 'none'), (2, 2304, 768, 20, 'none'),
                                            (130, 768, 3072, 22, 'slabs'), (256, 2304, 768, 22, 'none')])
def test_gemm_resident_whole_k(ops, M, N, K, hint, act):
    """Resident form of the decode-step GEMMs (tile_hint 20/21/22): the whole 768-long k range of a tile is requested at once;
    K = 3072 is cut into 4 partial slabs."""
    from vitcap_amd import _lib as L
    a = _bf(_rand((M, K), 50)).cuda()
    w = _bf(_rand((N, K), 51, 0.05)).cuda()
    bias = _rand((N,), 52, 0.1).cuda()
    z = a.float().cpu() @ w.float().cpu().t()
    if act == 'slabs':
        out = torch.empty((K // 768, M, N), device='cuda', dtype=torch.float32)
        ops.gemm_bias_act(a, w, None, out=out, tile_hint=hint)
        for i in range(K // 768):
            zi = a.float().cpu()[:, i * 768:(i + 1) * 768] @ w.float().cpu()[:, i * 768:(i + 1) * 768].t()
            _close(out[i], zi, 1e-4, 1e-4, 'resident slab %d' % i)
        _close(out.sum(0), z, 1e-4, 1e-4, 'resident slab sum')
    elif act == 'gelu':
        got = ops.gemm_bias_act(a, w, bias, act=L.ACT_GELU_ERF, tile_hint=hint)
        _close(got, torch.nn.functional.gelu(z + bias.cpu()), 2 ** -7, 2e-3, 'resident gelu')
    else:
        got = ops.gemm_bias_act(a, w, bias, out_dtype=torch.float32, tile_hint=hint)
        _close(got, z + bias.cpu(), 1e-4, 1e-4, 'resident f32')
        x = _rand((M, N), 53).cuda()
        got = ops.gemm_bias_act(a, w, bias, residual=x, out_dtype=torch.float32, tile_hint=hint)
        _close(got, z + bias.cpu() + x.cpu(), 1e-4, 1e-4, 'resident f32 + residual')


@pytest.mark.parametrize('branch_a,cls,pos0', [(1, 1, 20), (1, 0, 20), (0, 1, 20), (0, 0, 20), (1, 0, 40), (0, 1, 40), (0, 0, 462)])
def test_tag_embed_four_forms(ops, branch_a, cls, pos0):
    """vitcap_tag_embed: the four tag-row embeddings of modeling_bert.py:1435-1489 against their torch definition."""
    import ctypes as C
    from vitcap_amd._lib import lib, check
    B, n, VP = 3, 50, 30592
    g = torch.Generator().manual_seed(8)
    mk = lambda *sh: _bf(torch.randn(*sh, generator=g) * 0.05).cuda()
    cls_w, word, pos, typ = mk(VP, 768), mk(VP, 768), mk(512, 768), mk(2, 768)
    xword, xpos, xtyp = mk(VP, 768), mk(512, 768), mk(2, 768)
    gam, bet = (1 + torch.randn(768, generator=g) * 0.1).cuda(), (torch.randn(768, generator=g) * 0.1).cuda()
    xgam, xbet = (1 + torch.randn(768, generator=g) * 0.1).cuda(), (torch.randn(768, generator=g) * 0.1).cuda()
    tags = torch.randint(0, 30522, (B, 50), generator=g).cuda()
    xf = torch.empty(B * n, 768, device='cuda')
    xb = torch.empty(B * n, 768, device='cuda', dtype=torch.bfloat16)
    p = lambda t: C.c_void_p(t.data_ptr())
    check(lib.vitcap_tag_embed(p(tags), n, pos0, branch_a, cls, p(cls_w), p(word), p(pos), p(typ), p(gam), p(bet), p(xword), p(xpos), p(xtyp),
                               p(xgam), p(xbet), 1e-12, p(xf), p(xb), B, None), 'tag_embed')
    t = tags.clone()
    t[:, -1] = 102
    j = torch.arange(50, device='cuda') + 20            # encode_tag_to_embedding's literal caption_len; pos0 reaches the last form only
    jx = torch.arange(50, device='cuda') + pos0
    if branch_a and cls:
        want = cls_w[t].float()
    elif branch_a:
        want = torch.nn.functional.layer_norm(word[t].float() + pos[j].float() + typ[0].float(), (768,), gam, bet, 1e-12)
    elif cls:
        want = torch.nn.functional.layer_norm(cls_w[t].float() + pos[j].float() + typ[0].float(), (768,), gam, bet, 1e-12)
    else:
        want = torch.nn.functional.layer_norm(xword[t].float() + xpos[jx].float() + xtyp[0].float(), (768,), xgam, xbet, 1e-12)
    _close(xf.view(B, 50, 768), want.cpu(), 2e-5, 2e-5, 'tag_embed')
    assert torch.equal(xb.cpu(), xf.cpu().to(torch.bfloat16))
    assert lib.vitcap_tag_embed(p(tags), n, 463, branch_a, cls, p(cls_w), p(word), p(pos), p(typ), p(gam), p(bet), p(xword), p(xpos), p(xtyp),
                                p(xgam), p(xbet), 1e-12, p(xf), p(xb), B, None) == -1          # past the position table


def test_greedy_select_embed_equals_step_plus_embed(ops):
    """The fused greedy step (vocabulary GEMM row statistics -> token, log-prob, bookkeeping, next step's embedding) against
    the three separate kernels it replaces, over a whole 19-step loop with early finishers: ids, unfinished flags and
    embeddings bit-identical, log-probs to fp32 summation order; ties resolve to the lowest column."""
    from vitcap_amd import _lib as L
    B, V, VP = 37, 30522, 30592
    g = torch.Generator().manual_seed(5)
    wl = torch.zeros(VP, 768)
    wl[:V] = torch.randn(V, 768, generator=g) * 0.05
    wl = _bf(wl).cuda()
    bias = torch.full((VP,), -1e30)
    bias[:V] = torch.randn(V, generator=g)
    bias = bias.cuda()
    word = _bf(torch.randn(VP, 768, generator=g) * 0.05).cuda()
    pos = _bf(torch.randn(512, 768, generator=g) * 0.05).cuda()
    typ = _bf(torch.randn(2, 768, generator=g) * 0.05).cuda()
    gam = (1 + torch.randn(768, generator=g) * 0.1).cuda()
    bet = (torch.randn(768, generator=g) * 0.1).cuda()
    st_a, st_b = ops.greedy_init(B), ops.greedy_init(B)
    for st in (st_a, st_b):
        st['raw_last'] = torch.zeros(B, dtype=torch.int64, device='cuda')
    eos = 102
    for t in range(1, 20):
        h = _bf(torch.randn(B, 768, generator=g)).cuda()
        bias_t = bias.clone()
        if t in (3, 9):
            bias_t[eos] = 50.0 if t == 9 else bias_t[eos]          # step 9: every row picks [SEP] -> all finished afterwards
        if t == 3:
            h[5] = 0                                                 # row 5: logits = bias only ...
            bias_t[700] = 40.0
            bias_t[12345] = 40.0                                     # ... with an exact tie: the lower column must win
        logits, rs = ops.gemm_rowstat(h, wl, bias_t)
        ref = h.float().cpu() @ wl.float().cpu().t() + bias_t.cpu()
        _close(logits[:, :V], ref[:, :V], 1e-4, 1e-4, 'rowstat logits')
        ops.greedy_step(logits, st_a, t, eos=eos)
        xf, xb = ops.greedy_select_embed(rs, st_b, t, word, pos, typ, gam, bet, eos=eos)
        assert torch.equal(st_a['ids'], st_b['ids']), t
        assert torch.equal(st_a['unf'], st_b['unf']) and torch.equal(st_a['cnt'], st_b['cnt'])
        np.testing.assert_allclose(st_a['sum_lp'].cpu().numpy(), st_b['sum_lp'].cpu().numpy(), rtol=0, atol=2e-5)
        if t == 3:
            assert int(st_b['ids'][5, 3]) == 700
        if t < 19:
            want_f, want_b = ops.embed_step(st_a['ids'], t + 1, word, pos, typ, gam, bet)
            assert torch.equal(xf, want_f) and torch.equal(xb, want_b)
    assert int(st_b['unf'].sum()) == 0 and torch.equal(st_a['raw_last'], st_b['raw_last'])
    np.testing.assert_allclose(st_a['logprob'].cpu().numpy(), st_b['logprob'].cpu().numpy(), rtol=0, atol=2e-6)
    assert (st_b['ids'][:, 10:] == 0).all()


@pytest.mark.parametrize('act', [False, True])
def test_sum_layernorm(ops, act):
    S, M = 6, 130
    parts = _rand((S, M, 768), 43).cuda()
    bias = _rand((768,), 44, 0.1).cuda()
    res = None if act else _rand((M, 768), 45).cuda()
    g = (1 + _rand((768,), 46, 0.2)).cuda()
    b = _rand((768,), 47, 0.1).cuda()
    yb, yf = ops.sum_layernorm(parts, bias, res, g, b, 1e-12, act_before_ln=act)
    v = parts.cpu().sum(0) + bias.cpu()
    v = torch.nn.functional.gelu(v) if act else v + res.cpu()
    want = torch.nn.functional.layer_norm(v, (768,), g.cpu(), b.cpu(), 1e-12)
    _close(yf, want, 2e-5, 2e-5, 'sum_layernorm')
    assert torch.equal(yb.cpu(), yf.cpu().to(torch.bfloat16))


def test_gemm_rejects_bad_shapes(ops):
    from vitcap_amd._lib import VitcapError
    a = _bf(torch.zeros(8, 96)).cuda()
    w = _bf(torch.zeros(16, 96)).cuda()
    with pytest.raises(VitcapError):
        ops.gemm_bias_act(a, w)                               # K % 64 != 0


@pytest.mark.parametrize('eps', [1e-6, 1e-12])
def test_layernorm(ops, eps):
    M = 1157
    x = (_rand((M, 768), 11, 3.0) + 0.5).cuda()
    g = (1 + _rand((768,), 12, 0.2)).cuda()
    b = _rand((768,), 13, 0.1).cuda()
    yb, yf = ops.layernorm(x, g, b, eps, want_bf16=True, want_f32=True)
    want = torch.nn.functional.layer_norm(x.cpu(), (768,), g.cpu(), b.cpu(), eps)
    _close(yf, want, 1e-5, 1e-5, 'layernorm f32')
    _close(yb, want, 2 ** -8, 1e-6, 'layernorm bf16')
    assert torch.equal(yb.cpu(), yf.cpu().to(torch.bfloat16)), 'bf16 output must be RNE of the fp32 output'


def test_patch_embed_matches_conv(ops, sd_t):
    """patch gather + GEMM(+bias,+pos) + cls rows == PatchEmbed conv + cls + pos (a1)."""
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    B = 2
    img = torch.from_numpy(W.gen_image_batch(B, 99))
    p = 'image_encoder.module.'
    r = O._R(True)
    sdw = {k: (r(v) if k == p + 'patch_embed.proj.weight' else v) for k, v in sd_t.items() if k.startswith(p)}
    want = O.patch_embed(sdw, r(img))
    patches = ops.patch_gather(img.cuda())
    x = torch.zeros((B * 577, 768), device='cuda')
    from vitcap_amd._lib import lib, check
    import ctypes as C
    wq = _bf(sd_t[p + 'patch_embed.proj.weight'].reshape(768, 768)).cuda()
    pos = sd_t[p + 'pos_embed'].reshape(577, 768).cuda().contiguous()
    ops.gemm_bias_act(patches, wq, sd_t[p + 'patch_embed.proj.bias'].cuda(), residual=pos[1:], out=x,
                      row_group=576, out_group_rows=577, out_row_off=1, res_periodic=1)
    cls = sd_t[p + 'cls_token'].reshape(768).cuda().contiguous()
    check(lib.vitcap_cls_rows(C.c_void_p(cls.data_ptr()), C.c_void_p(pos.data_ptr()), C.c_void_p(x.data_ptr()), B, 577,
                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'cls_rows')
    _close(x.view(B, 577, 768), want, 1e-4, 1e-4, 'patch embed')


# S = 5: left-over keys only (no tile at all); 40: one masked tail tile; 128 / 200 / 276 / 777: 2 / 3 / 4+tail / 12 tiles through the
# 3-slot LDS-DMA ring (wrap-around), 8 left-over keys on the vector ALU (200), tail tile behind full tiles (276)
@pytest.mark.parametrize('B,S', [(1, 64), (2, 577), (1, 578), (3, 130), (1, 5), (2, 40), (2, 128), (1, 200), (1, 276), (1, 777)])
def test_attn_dense(ops, B, S):
    from oracle import vitcap_oracle as O
    qkv = _bf(_rand((B, S, 2304), 20 + S, 2.0))
    got = ops.attn_dense(qkv.reshape(B * S, 2304).cuda().contiguous(), B, S)
    want_emu = O.attn_rounded(qkv.float(), S, O._R(True))               # device rounding points
    q, k, v = qkv.float().view(B, S, 3, 12, 64).permute(2, 0, 3, 1, 4)
    want_ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B, S, 768)
    # vs emulation: only summation order + 1 bf16 output rounding flip -> 1 bf16 ulp
    _close(got.view(B, S, 768), want_emu, 2 ** -7, 2e-3, 'attn vs emulation')
    # vs exact softmax in fp32: P rounded to bf16 inside (2^-9 relative per term, averaged)
    _close(got.view(B, S, 768), want_ref, 2 ** -6, 4e-3, 'attn vs fp32 softmax')


def test_attn_dense_spiked_scores(ops):
    """Forces large running-max jumps between key tiles (guide rule 26: exercise the rescale path)."""
    from oracle import vitcap_oracle as O
    B, S = 1, 577
    qkv = _bf(_rand((B, S, 2304), 77, 1.0))
    qkv = qkv.float()
    qkv[0, 5, :64] *= 6.0                    # query 5 head 0
    qkv[0, 300, 768:768 + 64] = qkv[0, 5, :64] * 1.5     # key 300 aligned with it: score spike in tile 4
    qkv[0, 570, 768:768 + 64] = qkv[0, 5, :64] * 3.0     # bigger spike in the last full tile
    qkv = _bf(qkv)
    got = ops.attn_dense(qkv.reshape(B * S, 2304).cuda().contiguous(), B, S)
    want = O.attn_rounded(qkv.float(), S, O._R(True))
    _close(got.view(B, S, 768), want, 2 ** -7, 2e-3, 'attn spiked')


@pytest.mark.parametrize('t', [1, 2, 10, 19])
def test_attn_decode_step(ops, t):
    from oracle import vitcap_oracle as O
    B, S = 3, 578
    r = O._R(True)
    vis = _bf(_rand((B, S, 2304), 30, 2.0))
    step = _bf(_rand((B, 2, 2304), 31 + t, 2.0))
    cache = _bf(_rand((B, 20, 2, 768), 32, 2.0))
    cache_dev = cache.cuda().contiguous()
    got = ops.attn_decode_step(step.reshape(B * 2, 2304).cuda().contiguous(), vis.reshape(B * S, 2304).cuda().contiguous(),
                               cache_dev, B, S, t)
    # reference: keys = visual | cached text 0..t-2 | this step's row 0 (pos t-1) | MASK row
    K = torch.cat([vis[..., 768:1536].float(), cache[:, :t - 1, 0].float(), step[:, :, 768:1536].float()], 1)
    V = torch.cat([vis[..., 1536:].float(), cache[:, :t - 1, 1].float(), step[:, :, 1536:].float()], 1)
    q = step[..., :768].float().view(B, 2, 12, 64).transpose(1, 2)
    Kh = K.view(B, -1, 12, 64).transpose(1, 2)
    Vh = V.view(B, -1, 12, 64).transpose(1, 2)
    s = q @ Kh.transpose(-1, -2)
    s[:, :, 0, -1] = float('-inf')
    want = O.softmax_pv_rounded(s, Vh, r).transpose(1, 2).reshape(B, 2, 768)
    _close(got.view(B, 2, 768), want, 2 ** -7, 2e-3, 'attn decode t=%d' % t)
    # cache row t-1 now holds this step's real-token K/V
    c = cache_dev.cpu()
    assert torch.equal(c[:, t - 1, 0], step[:, 0, 768:1536])
    assert torch.equal(c[:, t - 1, 1], step[:, 0, 1536:])
    if t >= 2:
        assert torch.equal(c[:, :t - 1], cache[:, :t - 1])


def test_embed_step(ops, sd_t):
    from oracle import vitcap_oracle as O
    e = 'module.bert.embeddings'
    word = _bf(sd_t[e + '.word_embeddings.weight']).cuda()
    pos = _bf(sd_t[e + '.position_embeddings.weight']).cuda()
    typ = _bf(sd_t[e + '.token_type_embeddings.weight']).cuda()
    g = sd_t[e + '.LayerNorm.weight'].cuda()
    b = sd_t[e + '.LayerNorm.bias'].cuda()
    B, t = 5, 7
    ids = torch.randint(0, 30522, (B, 20), generator=torch.Generator().manual_seed(3))
    xf, xb = ops.embed_step(ids.cuda(), t, word, pos, typ, g, b)
    tok = torch.stack([ids[:, t - 1], torch.full((B,), 103)], 1)
    x = (word.float().cpu()[tok] + pos.float().cpu()[torch.tensor([t - 1, t])] + typ.float().cpu()[0])
    want = torch.nn.functional.layer_norm(x, (768,), g.cpu(), b.cpu(), 1e-12).reshape(B * 2, 768)
    _close(xf, want, 1e-5, 1e-5, 'embed_step')
    assert torch.equal(xb.cpu(), xf.cpu().to(torch.bfloat16))


def test_greedy_step_bookkeeping(ops):
    """argmax with lowest-index tie-break, log-softmax gather, EOS/pad bookkeeping (a12), bit-exact ids."""
    B, V, ld = 6, 30522, 30592
    g = torch.Generator().manual_seed(5)
    st = ops.greedy_init(B)
    ids_ref = torch.zeros(B, 20, dtype=torch.long)
    ids_ref[:, 0] = 101
    unf = torch.ones(B, dtype=torch.long)
    lps, unfs = [], []
    for t in range(1, 20):
        logits = torch.randn(B, ld, generator=g)
        logits[:, V:] = 1e9                        # padding columns must be ignored
        logits[0, 777] = logits[0, 12345] = 50.0   # exact tie -> lowest index
        if t == 4:
            logits[1, 102] = 60.0                  # EOS for row 1
        if t == 9:
            logits[2, 102] = 60.0
        ops.greedy_step(logits.cuda(), st, t)
        row = logits[:, :V]
        nxt = row.argmax(-1)
        lp = torch.log_softmax(row, -1).gather(1, nxt[:, None])[:, 0]
        lps.append(lp)
        unfs.append(unf.clone())
        add = nxt * unf
        ids_ref[:, t] = add
        unf = unf * (add != 102).long()
    ids_ref[:, -1].masked_fill_(unf.bool(), 102)
    lp_ref = (torch.stack(lps, 1) * torch.stack(unfs, 1).float()).sum(1) / torch.stack(unfs, 1).float().sum(1)
    assert torch.equal(st['ids'].cpu(), ids_ref)
    assert ids_ref[0, 1] == 777
    _close(st['logprob'], lp_ref, 1e-5, 1e-5, 'greedy logprob')


def test_sigmoid_topk(ops):
    B, V, ld = 4, 30522, 30592
    g = torch.Generator().manual_seed(8)
    logits = torch.randn(B, ld, generator=g) * 2
    logits[:, V:] = 100.0
    ids, prob, ln = ops.sigmoid_topk(logits.cuda(), V=V)
    p, i = torch.sigmoid(logits[:, :V]).topk(50, dim=1)
    assert torch.equal(ids.cpu(), i)
    _close(prob, p, 1e-6, 1e-6, 'topk prob')
    assert torch.equal(ln.cpu(), (p >= 0.2).sum(1))


@pytest.mark.parametrize('B,S', [(1, 64), (2, 577), (1, 578), (2, 130)])
def test_attn_dense_backward(ops, B, S):
    """dq/dk/dv of softmax(qk^T/8)v vs torch autograd (fp32) on the same bf16 inputs.  P and dS are rounded to bf16
    inside the kernels: tolerance 3e-2 of the tensor's max magnitude."""
    qkv = _bf(_rand((B, S, 2304), 50 + S, 1.5))
    dout = _bf(_rand((B, S, 768), 51 + S, 1.0))
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.view(B, S, 3, 12, 64).permute(2, 0, 3, 1, 4)
    o_ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(B, S, 768)
    o_ref.backward(dout.float())
    want = x.grad
    qd = qkv.reshape(B * S, 2304).cuda().contiguous()
    out, lse = ops.attn_dense_train(qd, B, S)
    _close(out.view(B, S, 768), o_ref.detach(), 2 ** -6, 4e-3, 'attn fwd(train)')
    s = (q @ k.transpose(-1, -2) * 0.125).detach()
    lse_ref = torch.logsumexp(s, -1) * 1.4426950408889634
    _close(lse, lse_ref, 1e-4, 1e-3, 'lse2')
    dqkv = ops.attn_dense_bwd(qd, out, dout.reshape(B * S, 768).cuda().contiguous(), lse, B, S).view(B, S, 2304).float().cpu()
    for name, lo in (('dq', 0), ('dk', 768), ('dv', 1536)):
        g, w_ = dqkv[..., lo:lo + 768], want[..., lo:lo + 768]
        tol = 3e-2 * float(w_.abs().max())
        err = float((g - w_).abs().max())
        rel = float((g - w_).norm() / w_.norm())
        print(name, 'max err %.3e (max %.3e) rel L2 %.3e' % (err, float(w_.abs().max()), rel))
        assert err <= tol and rel < 2e-2, name


@pytest.mark.parametrize('M,k', [(37, 4), (300, 10), (1280, 16)])
def test_row_topk_from_pieces_equals_row_scan(ops, M, k):
    """Beam candidates from the vocabulary GEMM's row statistics (vitcap_row_topk_pieces) against the scan of the whole rows
    (vitcap_row_topk_lse) and against torch.topk: values and columns identical (ties: lowest column first), logsumexp to fp32
    summation order.  M = 37 runs the 64x64-tile form of the GEMM, 300 and 1280 (256 images x 5 beams) the 128x128 one; rows
    with exact ties inside one piece, across pieces, and with all k winners inside a single 32-column piece."""
    V, VP = 30522, 30592
    g = torch.Generator().manual_seed(100 + M)
    wl = torch.zeros(VP, 768)
    wl[:V] = torch.randn(V, 768, generator=g) * 0.05
    wl = _bf(wl).cuda()
    bias = torch.full((VP,), -1e30)
    bias[:V] = torch.randn(V, generator=g)
    h = _bf(torch.randn(M, 768, generator=g))
    h[3] = 0                                               # row 3: logits = bias only ...
    bias[700] = bias[12345] = bias[701] = 40.0             # ... with exact ties inside a piece (700, 701) and across pieces
    bias[2000:2000 + 20] = 30.0 + torch.arange(20) * 0.01  # 20 large values inside one piece (columns 1984..2015 hold 16 of them)
    logits, rs = ops.gemm_rowstat(h.cuda(), wl, bias.cuda())
    ref = (h.float() @ wl.float().cpu().t() + bias)[:, :V]
    assert float((logits[:, :V].cpu() - ref).abs().max()) < 2e-3
    v0, i0, l0 = ops.row_topk(logits, V, k)
    v1, i1, l1 = ops.row_topk(logits, V, k, rowstat=rs)
    torch.cuda.synchronize()
    assert torch.equal(v0, v1) and torch.equal(i0, i1)
    np.testing.assert_allclose(l1.cpu().numpy(), l0.cpu().numpy(), rtol=0, atol=2e-5)
    tv, ti = torch.topk(logits[:, :V].cpu(), k, dim=1)
    assert torch.equal(v1.cpu(), tv)
    untied = (tv[:, :-1] != tv[:, 1:]).all(1)
    assert torch.equal(i1.cpu().long()[untied], ti[untied])
    assert i1[3, :3].tolist() == [700, 701, 12345]
    np.testing.assert_allclose(l1.cpu().numpy(), torch.logsumexp(logits[:, :V].cpu().double(), 1).numpy(), atol=1e-4)


@pytest.mark.parametrize('K,t,L', [(5, 1, 20), (5, 7, 20), (2, 19, 20), (3, 12, 20), (8, 4, 20), (8, 19, 20), (5, 38, 40), (1, 5, 20)])
def test_attn_decode_beams_vs_single_sequences(ops, K, t, L):
    """Beam-size decode attention on the matrix pipe (one workgroup per (image, head), all 2*K query rows of the image in one
    MFMA, visual K rows loaded once) against the per-sequence vector-ALU kernel on the same inputs: same softmax, same bf16
    rounding of P, different fp32 summation order -> outputs within one bf16 ulp (2^-7 relative, 2e-3 absolute), cache rows
    published identically; and against the fp32 reference like test_attn_decode_step.  (8, 19) and (5, 38, max_length 40) run the
    large instantiation of the kernel, (1, 5) a single sequence per image."""
    from oracle import vitcap_oracle as O
    n_img, S = 3, 578
    B = n_img * K
    vis = _bf(_rand((n_img, S, 2304), 40 + K, 2.0))
    step = _bf(_rand((B, 2, 2304), 41 + t, 2.0))
    cache = _bf(_rand((B, L, 2, 768), 42, 2.0))
    c1, c2 = cache.cuda().contiguous(), cache.cuda().contiguous()
    step_d = step.reshape(B * 2, 2304).cuda().contiguous()
    vis_d = vis.reshape(n_img * S, 2304).cuda().contiguous()
    got = ops.attn_decode_beams(step_d, vis_d, c1, n_img, K, S, t, max_len=L)
    want = ops.attn_decode_step(step_d, vis_d, c2, B, S, t, max_len=L, seq_per_image=K)
    torch.cuda.synchronize()
    assert torch.equal(c1, c2)
    _close(got.float().cpu(), want.float().cpu(), 2 ** -7, 2e-3, 'beam attention vs per-sequence kernel K=%d t=%d' % (K, t))
    r = O._R(True)
    visr = vis.repeat_interleave(K, 0)
    Kk = torch.cat([visr[..., 768:1536].float(), cache[:, :t - 1, 0].float(), step[:, :, 768:1536].float()], 1)
    V = torch.cat([visr[..., 1536:].float(), cache[:, :t - 1, 1].float(), step[:, :, 1536:].float()], 1)
    q = step[..., :768].float().view(B, 2, 12, 64).transpose(1, 2)
    s = q @ Kk.view(B, -1, 12, 64).transpose(1, 2).transpose(-1, -2)
    s[:, :, 0, -1] = float('-inf')
    ref = O.softmax_pv_rounded(s, V.view(B, -1, 12, 64).transpose(1, 2), r).transpose(1, 2).reshape(B, 2, 768)
    _close(got.view(B, 2, 768), ref, 2 ** -7, 2e-3, 'beam attention vs reference K=%d t=%d' % (K, t))


@pytest.mark.parametrize('variant', ['proj', 'fc2', 'no_res', 'ragged', 'post_ln_f32'])
def test_gemm_with_fused_layernorm_bit_identical(ops, variant):
    """vitcap_gemm_desc.ln_*: the residual GEMM that normalises its own finished rows (the last of a row block's three column tiles
    to arrive does it: write-through stores, one ticket counter per row block, sc1 loads) against the same GEMM followed by
    vitcap_layernorm_fwd -- fp32 rows and LayerNorm outputs bit for bit, counters left at zero, repeated launches included (every
    launch deals the row blocks' last arrivers differently).  In place on the residual and with the LayerNorm output aliasing A,
    as the engine's proj launch does."""
    M, K = {'proj': (36928, 768), 'fc2': (9232, 3072), 'no_res': (4099, 768), 'ragged': (2049, 768), 'post_ln_f32': (5780, 768)}[variant]
    N = 768
    a = _bf(_rand((M, K), 91)).cuda()
    w = _bf(_rand((N, K), 92, 0.05)).cuda()
    bias = _rand((N,), 93, 0.1).cuda()
    g = (1.0 + _rand((N,), 94, 0.2)).cuda()
    b = _rand((N,), 95, 0.2).cuda()
    x0 = _rand((M, N), 96).cuda() if variant != 'no_res' else None
    eps = 1e-12 if variant == 'post_ln_f32' else 1e-6
    want_f32 = variant == 'post_ln_f32'
    # reference: plain GEMM (same kernel family, tile_hint 5) + the LayerNorm kernel
    x_ref = ops.gemm_bias_act(a, w, bias, residual=x0, out=x0.clone() if x0 is not None else None, out_dtype=torch.float32, tile_hint=5)
    hb_ref, hf_ref = ops.layernorm(x_ref, g, b, eps, want_bf16=True, want_f32=want_f32)
    cnt = torch.zeros(M // 128 + 8, dtype=torch.int32, device='cuda')
    for rep in range(3):
        xin = x0.clone() if x0 is not None else None
        if variant == 'proj':                       # LayerNorm output written over A's buffer (a row block's A rows are its own tiles')
            a_buf = a.clone()
            x, hb, hf = ops.gemm_layernorm(a_buf, w, bias, xin, g, b, eps, out=xin, ln_out=a_buf, counters=cnt, tile_hint=5)
        else:
            x, hb, hf = ops.gemm_layernorm(a, w, bias, xin, g, b, eps, out=xin, want_f32=want_f32, counters=cnt, tile_hint=5)
        torch.cuda.synchronize()
        assert torch.equal(x, x_ref), 'fp32 rows differ (%s, launch %d): max |d| %g' % (variant, rep, float((x - x_ref).abs().max()))
        assert torch.equal(hb, hb_ref), 'LayerNorm rows differ (%s, launch %d): %d rows' % (
            variant, rep, int((hb != hb_ref).any(1).sum()))
        if want_f32:
            assert torch.equal(hf, hf_ref)
        assert int(cnt.abs().sum()) == 0, 'ticket counters not back at zero'
    # without counters (or under a launch form without the pass) the LayerNorm launch follows: same bits
    x2, hb2, _ = ops.gemm_layernorm(a, w, bias, x0.clone() if x0 is not None else None, g, b, eps, tile_hint=5)
    x3, hb3, _ = ops.gemm_layernorm(a, w, bias, x0.clone() if x0 is not None else None, g, b, eps, counters=cnt, tile_hint=0)
    torch.cuda.synchronize()
    assert torch.equal(hb2, hb_ref) and torch.equal(x2, x_ref) and torch.equal(hb3, hb_ref)
