"""Parity of the training-step kernels (backward GEMM forms, LayerNorm backward, joint causal decoder attention, losses,
AdamW) against fp32 torch (autograd) on the same bf16-rounded inputs.  GPU only."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    assert torch.cuda.is_available()
    from vitcap_amd import ops as o
    return o


def _bf(t):
    return t.to(torch.bfloat16)


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _rel(got, want):
    got, want = got.detach().float().cpu(), want.detach().float().cpu()
    return float((got - want).norm() / (want.norm() + 1e-30))


def test_transpose_colsum(ops):
    R, Cc = 1154, 2304
    x = _bf(_rand((R, Cc), 1))
    cs = torch.zeros(Cc, device='cuda')
    xt = ops.transpose_colsum(x.cuda(), cs)
    assert xt.shape == (Cc, 1216)
    assert torch.equal(xt[:, :R].cpu(), x.t().contiguous()) and float(xt[:, R:].abs().sum()) == 0.0
    np.testing.assert_allclose(cs.cpu().numpy(), x.float().sum(0).numpy(), rtol=1e-4, atol=1e-3)


def test_weight_grad_splitk_and_dgrad(ops):
    """dW = dY^T X via transposes + ragged split-K slabs; dX = dY W via the transposed weight; gelu' epilogue."""
    M, N, K = 1154, 768, 3072            # y = x W^T, W [N][K]
    x = _bf(_rand((M, K), 2)); w = _bf(_rand((N, K), 3, 0.05)); dy = _bf(_rand((M, N), 4))
    xd, wd, dyd = x.cuda(), w.cuda(), dy.cuda()
    dyT = ops.transpose_colsum(dyd)                    # [N][Mp]
    xT = ops.transpose_colsum(xd)                      # [K][Mp]
    slabs = ops.gemm_ex(dyT, xT, split_k=5)            # [5][N][K] fp32
    dw = torch.zeros(N, K, device='cuda')
    ops.reduce_slabs(slabs, dw)
    want_dw = dy.float().t() @ x.float()
    assert _rel(dw, want_dw) < 1e-5
    ops.reduce_slabs(slabs, dw, accumulate=True)
    assert _rel(dw, 2 * want_dw) < 1e-5
    # dgrad with the gelu' epilogue: dz = (dy @ W) * gelu'(z)
    from vitcap_amd._lib import lib, check
    wt = torch.empty(K, N, device='cuda', dtype=torch.bfloat16)
    wb = torch.empty(N, K, device='cuda', dtype=torch.bfloat16)
    wf = w.float().cuda().contiguous()
    check(lib.vitcap_cast_transpose(C.c_void_p(wf.data_ptr()), C.c_void_p(wb.data_ptr()), C.c_void_p(wt.data_ptr()), N, K, N,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'cast_transpose')
    assert torch.equal(wb.cpu(), w) and torch.equal(wt.cpu(), w.t().contiguous())
    # forward with the activation's derivative stored for the backward: zout = gelu'(pre-activation), g = gelu(pre-activation)
    zout = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    bias = _rand((N,), 6).cuda()
    g = ops.gemm_ex(xd, wd, bias=bias, act=1, zout=zout)
    pre = (x.float() @ w.float().t() + bias.cpu()).requires_grad_(True)
    torch.nn.functional.gelu(pre).backward(torch.ones_like(pre))
    assert _rel(zout, pre.grad) < 4e-3 and _rel(g, torch.nn.functional.gelu(pre.detach())) < 4e-3
    assert torch.equal(g, ops.gemm_bias_act(xd, wd, bias, act=1)), 'the activation itself is the same with and without zout'
    # dgrad with that factor in the epilogue: dz = (dy @ W) * f
    f = _bf(_rand((M, K), 5, 0.6) + 0.5)
    dz = ops.gemm_ex(dyd, wt, aux=f.cuda())            # [M][K] bf16
    assert _rel(dz, (dy.float() @ w.float()) * f.float()) < 4e-3
    with pytest.raises(Exception, match='zout'):       # zout without the GELU has no meaning any more
        ops.gemm_ex(xd, wd, bias=bias, zout=zout)


@pytest.mark.parametrize('dy_f32', [False, True])
def test_layernorm_backward(ops, dy_f32):
    M = 1157
    x = (_rand((M, 768), 7, 2.0) + 0.3)
    gam = 1 + _rand((768,), 8, 0.2)
    dy = _rand((M, 768), 9)
    dy = dy if dy_f32 else _bf(dy)
    dres = _rand((M, 768), 10)
    xx = x.clone().requires_grad_(True)
    gg = gam.clone().requires_grad_(True)
    bb = torch.zeros(768, requires_grad=True)
    torch.nn.functional.layer_norm(xx, (768,), gg, bb, 1e-6).backward(dy.float())
    dgam = torch.zeros(768, device='cuda'); dbet = torch.zeros(768, device='cuda')
    dxf, dxb = ops.layernorm_bwd(x.cuda(), dy.cuda(), gam.cuda(), 1e-6, dgam, dbet, dres=dres.cuda())
    assert _rel(dxf, xx.grad + dres) < 1e-5
    assert torch.equal(dxb.cpu(), dxf.cpu().to(torch.bfloat16))
    assert _rel(dgam, gg.grad) < 1e-4 and _rel(dbet, bb.grad) < 1e-4
    # the same with the column sums of the bf16 result added on the way out (bias gradient of the layer dx is the output
    # gradient of): identical dx, sums of the ROUNDED values, accumulated into what the buffer held
    cs = torch.full((768,), 0.5, device='cuda')
    dgam2 = torch.zeros(768, device='cuda'); dbet2 = torch.zeros(768, device='cuda')
    dxf2, dxb2 = ops.layernorm_bwd(x.cuda(), dy.cuda(), gam.cuda(), 1e-6, dgam2, dbet2, dres=dres.cuda(), dxb_colsum=cs)
    assert torch.equal(dxf2, dxf) and torch.equal(dxb2, dxb)
    assert _rel(cs, 0.5 + dxb.float().sum(0)) < 1e-5 and _rel(dgam2, dgam) < 1e-5


def test_cast_transpose_table_equals_single_calls(ops):
    """vitcap_cast_transpose_multi (one launch over a device-resident table: what TrainEngine.refresh_weights runs after every
    optimizer step) writes exactly what one vitcap_cast_transpose per matrix writes -- ragged N (30522 -> ld 30592), a table entry
    without a transposed copy, a 2-row matrix."""
    from vitcap_amd import _lib as L
    from vitcap_amd._lib import lib, check
    shapes = [(768, 768, 768, True), (2304, 768, 2304, True), (30522, 768, 30592, True), (2, 768, 8, True), (768, 3072, 768, False)]
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
    src, single, multi, items, tile0 = [], [], [], [], 0
    for i, (N, K, ld, want_t) in enumerate(shapes):
        w = _rand((N, K), 300 + i).cuda()
        a = (torch.zeros(N, K, device='cuda', dtype=torch.bfloat16), torch.zeros(K, ld, device='cuda', dtype=torch.bfloat16) if want_t else None)
        b = (torch.zeros(N, K, device='cuda', dtype=torch.bfloat16), torch.zeros(K, ld, device='cuda', dtype=torch.bfloat16) if want_t else None)
        check(lib.vitcap_cast_transpose(p(w), p(a[0]), p(a[1]), N, K, ld, s), 'cast_transpose')
        items.append(L.CtItem(w.data_ptr(), b[0].data_ptr(), b[1].data_ptr() if want_t else None, N, K, ld, tile0))
        tile0 += ((N + 63) // 64) * (K // 64)
        src.append(w); single.append(a); multi.append(b)
    tab = torch.frombuffer(bytearray(bytes((L.CtItem * len(items))(*items))), dtype=torch.uint8).cuda()
    check(lib.vitcap_cast_transpose_multi(p(tab), len(items), tile0, s), 'cast_transpose_multi')
    for w, a, b, (N, K, ld, want_t) in zip(src, single, multi, shapes):
        assert torch.equal(a[0], b[0]) and torch.equal(b[0], w.to(torch.bfloat16))
        if want_t:
            assert torch.equal(a[1], b[1]) and torch.equal(b[1][:, :N], w.to(torch.bfloat16).t())
            assert float(b[1][:, N:].float().abs().sum()) == 0.0


def test_fused_bias_gradients(ops):
    """Bias gradients = column sums of a backward operand, added by the kernel that WRITES the operand: the fp32 -> bf16 cast of a
    residual-stream gradient (vitcap_cast_bf16_colsum), the 256x256 GEMM's epilogue (vitcap_gemm_desc.colsum, with the gelu'
    factor), against a separate pass over the stored bf16 values (vitcap_colsum_bf16) and torch."""
    M, N, K = 2308, 1024, 768
    x = _rand((M, 768), 31, 1.5).cuda()
    cs = torch.full((768,), -2.0, device='cuda')
    y = ops.cast_bf16_colsum(x, cs)
    assert torch.equal(y, ops.cast_bf16(x)) and _rel(cs, -2.0 + y.float().sum(0)) < 1e-5
    dy = _bf(_rand((M, K), 32)).cuda()
    wt = _bf(_rand((N, K), 33, 0.05)).cuda()
    z = _bf(_rand((M, N), 34, 2.0)).cuda()
    for aux in (z, None):
        plain = ops.gemm_ex(dy, wt, aux=aux, tile_hint=5)
        cs = torch.full((N,), 1.0, device='cuda')
        fused = ops.gemm_ex(dy, wt, aux=aux, colsum=cs)
        assert torch.equal(fused, plain)
        sep = torch.full((N,), 1.0, device='cuda')
        ops.colsum_bf16(plain, sep)
        assert _rel(cs, 1.0 + plain.float().sum(0)) < 1e-5 and _rel(cs, sep) < 1e-5
    with pytest.raises(Exception, match='colsum'):            # below the 256x256 kernel's range: refused, not dropped
        ops.gemm_ex(dy[:512], wt, colsum=torch.zeros(N, device='cuda'))


@pytest.mark.parametrize('M', [2308, 36928])
def test_gemm_4wave_training_extras_bit_identical(ops, M):
    """The training extras of vitcap_gemm_ex (zout = gelu'(pre-activation) next to the activation; aux = the stored factor multiplied in;
    colsum = bias gradient added by the epilogue) in the persistent 4-wave kernel's register epilogue (round 5; tile_hint 42)
    against the 8-wave kernel that carries them by default (tile_hint 5): `out` and `zout` bit for bit, column sums to the fp32 atomics'
    order; rows past M (a ragged last tile) add nothing to the sums."""
    from vitcap_amd._lib import lib
    N, K = 1024, 768
    x = _bf(_rand((M, K), 51)).cuda()
    w = _bf(_rand((N, K), 52, 0.05)).cuda()
    bias = _rand((N,), 53).cuda()
    z4 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    z8 = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    g4 = ops.gemm_ex(x, w, bias=bias, act=1, zout=z4, tile_hint=42)         # 4-wave persistent + extras
    g8 = ops.gemm_ex(x, w, bias=bias, act=1, zout=z8, tile_hint=5)          # 8-wave kernel
    assert torch.equal(g4, g8) and torch.equal(z4, z8)
    f = _bf(_rand((M, N), 54, 0.6) + 0.5).cuda()
    for aux in (f, None):
        c4 = torch.full((N,), 1.0, device='cuda')
        c8 = torch.full((N,), 1.0, device='cuda')
        d4 = ops.gemm_ex(x, w, aux=aux, colsum=c4, tile_hint=42)
        d8 = ops.gemm_ex(x, w, aux=aux, colsum=c8, tile_hint=5)
        assert torch.equal(d4, d8)
        assert _rel(c4, c8) < 1e-5 and _rel(c4, 1.0 + d4.float().sum(0)) < 1e-5
        assert torch.equal(ops.gemm_ex(x, w, aux=aux, tile_hint=42), d4)     # aux without the column sums


@pytest.mark.parametrize('p_drop', [0.0, 0.1, 0.5])
def test_attn_joint_causal_forward_backward(ops, p_drop):
    """The decoder's joint attention under teacher forcing in ONE pass of the dense MFMA kernels (causal_from = 578):
    visual rows attend visual rows, caption row q attends all visual rows and caption rows <= q.  Forward and the
    two backward kernels against autograd through the explicitly masked attention, same dropout decisions."""
    from oracle import vitcap_oracle as O
    B, SV, T = 2, 578, 20
    seed = 0x51f2ab17
    Lr = SV + T
    qkv = _bf(_rand((B, Lr, 2304), 21, 1.5))
    dout = _bf(_rand((B, Lr, 768), 22))
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.view(B, Lr, 3, 12, 64).permute(2, 0, 3, 1, 4)
    mask = torch.zeros(Lr, Lr)
    mask[:, :SV] = 1
    mask[SV:, SV:] = torch.tril(torch.ones(T, T))
    s = q @ k.transpose(-1, -2) * 0.125 + (1 - mask) * -10000.0
    pr = torch.softmax(s, -1)
    if p_drop > 0:
        pr = pr * torch.from_numpy(O.dropout_keep(seed, B, Lr, p_drop)) / (1 - p_drop)
    o_ref = (pr @ v).transpose(1, 2).reshape(B, Lr, 768)
    o_ref.backward(dout.float())
    want = x.grad
    qd = qkv.reshape(B * Lr, 2304).cuda().contiguous()
    dod = dout.reshape(B * Lr, 768).cuda().contiguous()
    dk = dict(p_drop=p_drop, drop_seed=seed, causal_from=SV)
    out, lse = ops.attn_dense_train(qd, B, Lr, **dk)
    assert _rel(out.view(B, Lr, 768), o_ref) < 6e-3
    assert _rel(out.view(B, Lr, 768)[:, SV:], o_ref[:, SV:]) < 6e-3          # caption rows on their own
    got = ops.attn_dense_bwd(qd, out, dod, lse, B, Lr, **dk).view(B, Lr, 2304).float().cpu()
    for name, lo in (('dq', 0), ('dk', 768), ('dv', 1536)):
        r = _rel(got[..., lo:lo + 768], want[..., lo:lo + 768])
        rc = _rel(got[:, SV:, lo:lo + 768], want[:, SV:, lo:lo + 768])
        print(name, 'rel L2 %.3e (caption rows %.3e)' % (r, rc))
        assert r < 1e-2 and rc < 1e-2, name


@pytest.mark.parametrize('M,N,K,splits', [(1154, 768, 768, 5), (4099, 2304, 768, 9), (2 * 598, 768, 3072, 3), (640, 256, 256, 1),
                                          (36928, 768, 768, 28)])
def test_gemm_tn_weight_gradient(ops, M, N, K, splits):
    """dW = dY^T X from the row-major operands via LDS transpose reads; slabs summed == fp32 reference (bf16 inputs are
    exact in fp32, so only the summation order differs: 1e-4 relative)."""
    y = _bf(_rand((M, N), 31, 1.0))
    x = _bf(_rand((M, K), 32, 1.0))
    # transpose-detecting inputs: a single hot element must land at [n][k], not [k][n]
    y[7, 3] = 64.0
    x[7, 200] = 32.0
    slabs = ops.gemm_tn(y.cuda(), x.cuda(), splits)
    got = slabs.sum(0).cpu()
    want = y.float().t() @ x.float()
    assert got.shape == (N, K)
    err = float((got - want).abs().max() / want.abs().max())
    print('gemm_tn M=%d N=%d K=%d splits=%d rel err %.2e' % (M, N, K, splits, err))
    assert err < 1e-4
    bias = torch.zeros(N, device='cuda')
    ops.colsum_bf16(y.cuda(), bias)
    assert _rel(bias, y.float().sum(0)) < 1e-5


@pytest.mark.parametrize('M,N,K,splits', [(36928, 2304, 768, 9), (36928, 768, 768, 28), (36928, 768, 3072, 7), (4099, 3072, 768, 7),
                                          (1154, 768, 768, 5), (640, 256, 256, 2)])
def test_gemm_tn_sum_equals_two_launches(ops, M, N, K, splits):
    """vitcap_gemm_tn_sum (the splits of a tile add their slabs up inside the launch: tickets, write-through slabs, slab order) against
    vitcap_gemm_tn + vitcap_reduce_slabs: bit for bit, plain and accumulating, on a scratch buffer that still holds an older launch's
    slabs and through repeated launches (the tickets must be back at zero every time); more workgroups than CUs is an error."""
    y = _bf(_rand((M, N), 61, 1.0)).cuda()
    x = _bf(_rand((M, K), 62, 1.0)).cuda()
    want = torch.empty(N, K, device='cuda')
    ops.reduce_slabs(ops.gemm_tn(y, x, splits), want)
    slabs = torch.full((splits, N, K), 7.0, device='cuda')
    for rep in range(4):
        got = torch.full((N, K), float('nan'), device='cuda')
        ops.gemm_tn_sum(y, x, splits, got, slabs=slabs)
        assert torch.equal(got, want), rep
    base = _rand((N, K), 63).cuda()
    acc = base.clone()
    ops.gemm_tn_sum(y, x, splits, acc, accumulate=True)
    want_acc = base.clone()
    ops.reduce_slabs(ops.gemm_tn(y, x, splits), want_acc, accumulate=True)
    assert torch.equal(acc, want_acc)
    if N * K // 65536 * 40 > 256:
        with pytest.raises(RuntimeError):
            ops.gemm_tn_sum(y, x, 40, torch.empty(N, K, device='cuda'))


def test_attn_probe_rows_forward_backward(ops):
    """[578 visual | 20 token rows | 19 [MASK] probe rows]: probe j sees visual + tokens 0..j + itself and nobody sees the
    probes (mask_from): forward and both backward kernels against autograd through the explicit mask."""
    B, SV, T, TP = 2, 578, 20, 19
    Lr = SV + T + TP
    qkv = _bf(_rand((B, Lr, 2304), 41, 1.5))
    dout = _bf(_rand((B, Lr, 768), 42))
    dout[:, :SV] = 0                                     # as in the SCST step, only text rows carry gradient at the top layer
    x = qkv.float().clone().requires_grad_(True)
    q, k, v = x.view(B, Lr, 3, 12, 64).permute(2, 0, 3, 1, 4)
    mask = torch.zeros(Lr, Lr)
    mask[:, :SV] = 1
    mask[SV:SV + T, SV:SV + T] = torch.tril(torch.ones(T, T))
    for j in range(TP):
        mask[SV + T + j, SV:SV + j + 1] = 1              # tokens 0..j
        mask[SV + T + j, SV + T + j] = 1                 # itself
    s = q @ k.transpose(-1, -2) * 0.125 + (1 - mask) * -10000.0
    o_ref = (torch.softmax(s, -1) @ v).transpose(1, 2).reshape(B, Lr, 768)
    o_ref.backward(dout.float())
    want = x.grad
    qd = qkv.reshape(B * Lr, 2304).cuda().contiguous()
    dod = dout.reshape(B * Lr, 768).cuda().contiguous()
    kw = dict(causal_from=SV, mask_from=SV + T)
    out, lse = ops.attn_dense_train(qd, B, Lr, **kw)
    for name, sl in (('visual', slice(0, SV)), ('token', slice(SV, SV + T)), ('probe', slice(SV + T, Lr))):
        r = _rel(out.view(B, Lr, 768)[:, sl], o_ref[:, sl])
        print('fwd', name, '%.3e' % r)
        assert r < 6e-3, name
    got = ops.attn_dense_bwd(qd, out, dod, lse, B, Lr, **kw).view(B, Lr, 2304).float().cpu()
    for name, lo in (('dq', 0), ('dk', 768), ('dv', 1536)):
        for rn, sl in (('visual', slice(0, SV)), ('token', slice(SV, SV + T)), ('probe', slice(SV + T, Lr))):
            r = _rel(got[:, sl, lo:lo + 768], want[:, sl, lo:lo + 768])
            print(name, rn, 'rel L2 %.3e  (|want| %.3e)' % (r, float(want[:, sl, lo:lo + 768].norm())))
            assert r < 1.5e-2 or float(want[:, sl, lo:lo + 768].norm()) < 1e-6, (name, rn)


def test_losses(ops, sd_t):
    from oracle import vitcap_oracle as O
    from vitcap_amd._lib import lib, check
    n, V, ld = 5, 30522, 30592
    logits = _rand((n, ld), 13, 3.0)
    tgt = torch.tensor([5, 30521, 1234, 0, 777])
    lg = logits[:, :V].clone().requires_grad_(True)
    want = O.label_smoothed_kl(lg, tgt)
    want.backward()
    loss = torch.zeros(1, device='cuda')
    dl = torch.empty(n, ld, device='cuda', dtype=torch.bfloat16)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ld_, tg_ = logits.cuda(), tgt.cuda()
    check(lib.vitcap_ls_kl_loss(C.c_void_p(ld_.data_ptr()), ld, V, C.c_void_p(tg_.data_ptr()), 0.1, n, None,
                                C.c_void_p(loss.data_ptr()), C.c_void_p(dl.data_ptr()), ld, s), 'ls_kl')
    assert abs(float(loss) - float(want)) < 1e-4 * abs(float(want)) + 1e-5
    assert _rel(dl[:, :V], lg.grad) < 4e-3 and float(dl[:, V:].float().abs().sum()) == 0.0
    B = 3
    tl = _rand((B, ld), 14, 4.0)
    label = torch.zeros(B, V)
    label[torch.arange(B).repeat_interleave(10), torch.randint(0, V, (B * 10,), generator=torch.Generator().manual_seed(1))] = 1
    out = torch.zeros(1, device='cuda')
    tl_, lb_ = tl.cuda(), label.cuda()
    check(lib.vitcap_focal_loss_sum(C.c_void_p(tl_.data_ptr()), ld, V, C.c_void_p(lb_.data_ptr()), 0.5,
                                    C.c_void_p(out.data_ptr()), B, s), 'focal')
    wantf = float(O.focal_neg_loss(tl[:, :V], label))
    assert abs(float(out) - wantf) < 2e-4 * abs(wantf)
    out.zero_()          # the tag loss of a configuration without `loss: focal`: torch.nn.BCEWithLogitsLoss() (modeling_bert.py:716-717)
    check(lib.vitcap_bce_logits_mean(C.c_void_p(tl_.data_ptr()), ld, V, C.c_void_p(lb_.data_ptr()), C.c_void_p(out.data_ptr()), B, s), 'bce')
    wantb = float(torch.nn.BCEWithLogitsLoss()(tl[:, :V], label))
    assert abs(float(out) - wantb) < 2e-4 * abs(wantb)


def test_adamw_clip_matches_oracle(ops):
    from oracle import vitcap_oracle as O
    from vitcap_amd._lib import lib, check
    n = 4096
    p = _rand((n,), 15); g = _rand((n,), 16, 3.0)
    lr = torch.tensor([1e-4, 1e-5, 0.0, 1e-4]); wd = torch.tensor([0.05, 0.0, 0.05, 0.0])
    P, G = p.cuda(), g.cuda()
    lr_d, wd_d = lr.cuda(), wd.cuda()
    Mm, Vv = torch.zeros(n, device='cuda'), torch.zeros(n, device='cuda')
    ss = torch.zeros(1, device='cuda')
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    pref, mref, vref = p.clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        ss.zero_()
        check(lib.vitcap_sumsq(C.c_void_p(G.data_ptr()), n, C.c_void_p(ss.data_ptr()), s), 'sumsq')
        check(lib.vitcap_adamw_multi(C.c_void_p(P.data_ptr()), C.c_void_p(G.data_ptr()), C.c_void_p(Mm.data_ptr()),
                                     C.c_void_p(Vv.data_ptr()), C.c_void_p(lr_d.data_ptr()), C.c_void_p(wd_d.data_ptr()),
                                     C.c_void_p(ss.data_ptr()), 1.0, 1.0, step, 0.9, 0.999, 1e-8, n // 1024, s), 'adamw')
        coef = min(1.0, 1.0 / (float(g.norm()) + 1e-6))
        for c in range(4):
            sl = slice(c * 1024, (c + 1) * 1024)
            if float(lr[c]) == 0.0:
                continue
            O.adamw_step(pref[sl], g[sl] * coef, mref[sl], vref[sl], step, float(lr[c]), float(wd[c]))
    assert abs(float(ss.sqrt()) - float(g.norm())) < 1e-3
    np.testing.assert_allclose(P.cpu().numpy(), pref.numpy(), rtol=1e-5, atol=1e-7)
    assert torch.equal(P.cpu()[2048:3072], p[2048:3072])          # lr 0 chunk untouched


@pytest.mark.parametrize('case', ['cls_row', 'caption_rows', 'caption_rows_dropout'])
def test_attention_query_range_equals_full(ops, case):
    """Row-restricted attention (last tag block: only the CLS query; last decoder layer: only the caption queries) against
    the full kernels: forward rows of the covered blocks bit-identical, backward dqkv bit-identical when dout is zero
    outside the range (the skipped query tiles would only have added exact zeros)."""
    B = 3
    if case == 'cls_row':
        S, cf, q_range, keep, pd = 577, 0, (0, 1), slice(0, 1), 0.0
    else:
        S, cf, q_range, keep, pd = 598, 578, (512, 598), slice(578, 598), (0.1 if case.endswith('dropout') else 0.0)
    qkv = _bf(_rand((B * S, 2304), 70, 1.5)).cuda()
    out_f, lse_f = ops.attn_dense_train(qkv, B, S, p_drop=pd, drop_seed=9, causal_from=cf)
    out_r, lse_r = ops.attn_dense_train(qkv, B, S, p_drop=pd, drop_seed=9, causal_from=cf, q_range=q_range)
    blk = slice(q_range[0], min(S, (q_range[1] + 127) // 128 * 128))
    assert torch.equal(out_r.view(B, S, 768)[:, blk], out_f.view(B, S, 768)[:, blk])
    assert torch.equal(lse_r[:, :, blk], lse_f[:, :, blk])
    assert float(out_r.view(B, S, 768)[:, :q_range[0]].abs().max() if q_range[0] else 0.0) == 0.0
    dout = torch.zeros(B, S, 768, device='cuda', dtype=torch.bfloat16)
    dout[:, keep] = _bf(_rand((B, keep.stop - keep.start, 768), 71, 1.0)).cuda()
    dout = dout.view(B * S, 768)
    d_full = ops.attn_dense_bwd(qkv, out_f, dout, lse_f, B, S, p_drop=pd, drop_seed=9, causal_from=cf)
    d_rows = ops.attn_dense_bwd(qkv, out_r, dout, lse_r, B, S, p_drop=pd, drop_seed=9, causal_from=cf, q_range=q_range)
    assert torch.equal(d_rows, d_full), 'max diff %g' % float((d_rows.float() - d_full.float()).abs().max())
    assert float(d_full.float().abs().max()) > 0
