"""CPU-side checks: the C-ABI library loads, exports every symbol include/vitcap_hip.h declares, rejects bad
arguments without touching a GPU, and the host-side mirror keeps the reference's checkpoint layout."""
import ctypes as C
import os
import re

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def L():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(REPO, 'vitcap_amd', 'libvitcap_hip.so')):
        g.build()
    from vitcap_amd import _lib
    return _lib


def test_header_symbols_exported(L):
    hdr = open(os.path.join(REPO, 'include', 'vitcap_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    names = set(re.findall(r'\b(vitcap_[a-z0-9_]+)\s*\(', hdr))
    assert len(names) >= 20
    raw = C.CDLL(L.LIB_PATH)
    for n in sorted(names):
        assert hasattr(raw, n), 'declared in include/vitcap_hip.h but not exported: ' + n
    assert names == set(L.EXPORTS), names ^ set(L.EXPORTS)


def test_jpeg_header_symbols_exported():
    """include/vitcap_jpeg.h (host front half of the JPEG decoder): every declared function is exported by libvitcap_jpeg.so, the ABI number
    matches the binding, and the ctypes mirror of vitcap_jpeg_info has the C struct's size (the device back half in libvitcap_hip.so
    takes the same struct inside vitcap_jpeg_image)."""
    import __graft_entry__ as g
    from vitcap_amd import jpegdec as J
    path = os.path.join(REPO, 'vitcap_amd', 'libvitcap_jpeg.so')
    if not os.path.exists(path):
        g.build()
    hdr = re.sub(r'/\*.*?\*/', '', open(os.path.join(REPO, 'include', 'vitcap_jpeg.h')).read(), flags=re.S)
    names = set(re.findall(r'\b(vitcap_jpeg_[a-z0-9_]+)\s*\(', hdr))
    assert names == {'vitcap_jpeg_abi', 'vitcap_jpeg_parse', 'vitcap_jpeg_decode_coefs', 'vitcap_jpeg_last_error'}
    raw = C.CDLL(path)
    for n in names:
        assert hasattr(raw, n), n
    assert raw.vitcap_jpeg_abi() == J.JPEG_ABI == int(re.search(r'#define VITCAP_JPEG_ABI (\d+)', hdr).group(1))
    # int32: abi, width, height, ncomp, hs[3], vs[3], blocks_w[3], blocks_h[3], samp_w[3], samp_h[3], block0[3], nblocks; uint16 qt[3][64]
    assert C.sizeof(J.JpegInfo) == 4 * (4 + 7 * 3 + 1) + 2 * 3 * 64
    lib = J.jpeg_lib()
    info = J.JpegInfo()
    assert lib.vitcap_jpeg_parse(b'not a jpeg', 10, C.byref(info)) == J.JPEG_EINVAL and b'SOI' in lib.vitcap_jpeg_last_error()


def test_argument_validation_without_gpu(L):
    d = L.GemmDesc(M=8, N=16, K=96, lda=96, ldw=96, ldc=16)
    rc = L.lib.vitcap_gemm_bias_act(None, None, None, None, None, C.byref(d), None)
    assert rc == -1 and b'null' in L.lib.vitcap_last_error()
    buf = (C.c_char * 4096)()
    a = C.c_void_p((C.addressof(buf) + 255) & ~255)
    rc = L.lib.vitcap_gemm_bias_act(a, a, None, None, a, C.byref(d), None)
    assert rc == -1 and b'multiple of 64' in L.lib.vitcap_last_error()
    assert L.lib.vitcap_layernorm_fwd(a, 768, a, a, 1e-6, a, None, 4, 512, None) == -1      # D != 768
    assert L.lib.vitcap_attn_decode_step(a, a, a, a, 2, 578, 20, 20, 1, 0.125, None) == -1  # t out of range
    assert L.lib.vitcap_sigmoid_topk(a, 30592, 30522, 65, 0.2, a, a, a, 1, None) == -1      # k > 64
    # device JPEG back half: null arguments, then a descriptor whose vitcap_jpeg_info did not come from vitcap_jpeg_parse
    assert L.lib.vitcap_jpeg_backhalf(None, 0, None, 0, None) == -1
    img = (L.JpegImage * 1)()
    img[0].coefs, img[0].rgb, img[0].pitch = a.value, a.value, 24
    assert L.lib.vitcap_jpeg_backhalf(img, 1, a, 1 << 20, None) == -1 and b'vitcap_jpeg_info' in L.lib.vitcap_last_error()


def test_abi_version_is_checked(L):
    """a struct or signature change bumps VITCAP_ABI_VERSION; the binding refuses a library of another version at load
    time, and the two option structs carry the version in their first field -- a caller built against an older header is rejected
    instead of having its (shorter / shifted) struct misread."""
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'include', 'vitcap_hip.h')).read()
    want = int(re.search(r'#define VITCAP_ABI_VERSION (\d+)', hdr).group(1))
    assert L.lib.vitcap_version() == want == L.ABI_VERSION
    buf = (C.c_char * 4096)()
    a = C.c_void_p((C.addressof(buf) + 255) & ~255)
    d = L.GemmDesc(M=8, N=16, K=64, lda=64, ldw=64, ldc=16)
    assert d.abi == want
    d.abi = want - 1
    assert L.lib.vitcap_gemm_bias_act(a, a, None, None, a, C.byref(d), None) == -1 and b'ABI' in L.lib.vitcap_last_error()
    o = L.gen_opts()
    assert o.abi == want and L.lib.vitcap_gen_opts_check(C.byref(o)) == 0
    stale = L.GenOpts()                 # zero-initialised, as a caller that never heard of the field would leave it
    assert L.lib.vitcap_gen_opts_check(C.byref(stale)) == -1 and b'ABI' in L.lib.vitcap_last_error()
    assert L.lib.vitcap_engine_workspace_bytes(1, C.byref(stale)) == 0


def test_gemm_tile_plan_host_logic(L):
    """vitcap_gemm_tile_plan (host only): the 256-row tiles plus the short tiles behind them cover every row exactly once,
    the 256-row region ends on a tile boundary, and the hot B = 64 shapes whose plain grids waste a partial round get a mix."""
    from vitcap_amd import ops
    for M in (577, 2048, 4616, 9232, 36928, 36992, 73856, 295424):
        for N, K in ((768, 768), (768, 3072), (2304, 768), (3072, 768)):
            tb, mts, ts = ops.gemm_tile_plan(M, N, K)
            assert mts in (0, 2, 3) and tb >= 0 and ts >= 0
            if mts == 0:
                assert ts == 0 and tb == (M + 255) // 256
            else:
                h = 64 * mts
                assert tb * 256 < M and (ts - 1) * h < M - tb * 256 <= ts * h, (M, N, K, tb, mts, ts)
    for N, K in ((768, 768), (768, 3072), (2304, 768)):       # 435 tiles = 1.7 rounds, 1305 = 5.1 rounds on 256 CUs
        assert ops.gemm_tile_plan(36928, N, K)[1] != 0
    assert L.lib.vitcap_gemm_tile_plan(0, 768, 768, (C.c_int * 3)()) == -1


def test_engine_lifecycle_and_workspace(L):
    h = C.c_void_p()
    assert L.lib.vitcap_engine_create(C.byref(h)) == 0 and h.value
    w1, w64 = L.lib.vitcap_engine_workspace_bytes(1, None), L.lib.vitcap_engine_workspace_bytes(64, None)
    assert 0 < w1 < w64 < 64 * w1 * 1.01 and w64 % 256 == 0
    assert L.lib.vitcap_engine_workspace_bytes(0, None) == 0
    # options are a per-call struct with the reference's test-time defaults; beams / max_length size the workspace
    o = L.gen_opts()
    assert (o.num_beams, o.max_length, o.eos_token_id, o.bos_token_id, o.mask_token_id, o.pad_token_id) == (1, 20, 102, 101, 103, 0)
    assert o.early_exit == 1 and o.use_graph == 0 and o.gemm_mode == L.GEMM_AUTO and abs(o.repetition_penalty - 1.0) < 1e-9
    assert L.lib.vitcap_engine_workspace_bytes(64, C.byref(o)) == w64
    assert L.lib.vitcap_engine_workspace_bytes(64, C.byref(L.gen_opts(num_beams=5))) > w64
    assert L.lib.vitcap_engine_workspace_bytes(64, C.byref(L.gen_opts(max_length=40))) > w64
    # constrained beam search: cbs_states * num_beams sequences per image size the workspace (the pointers are not followed here)
    ocbs = L.gen_opts(use_cbs=1, cbs_states=8, num_beams=2, fsm=4096, num_constraints=4096)
    assert L.lib.vitcap_gen_opts_check(C.byref(ocbs)) == 0
    assert L.lib.vitcap_engine_workspace_bytes(4, C.byref(ocbs)) > L.lib.vitcap_engine_workspace_bytes(4, C.byref(L.gen_opts(num_beams=8)))
    for bad in (dict(num_beams=9), dict(max_length=41), dict(max_length=1), dict(num_beams=1, num_keep_best=2),
                dict(num_beams=2, seqs_per_image=2), dict(repetition_penalty=0.0), dict(eos_token_id=30522), dict(gemm_mode=7), dict(tag_pos0=19), dict(tag_pos0=463),
                dict(use_cbs=1), dict(use_cbs=1, cbs_states=8, fsm=4096, num_constraints=4096, num_keep_best=2, num_beams=2),
                dict(use_cbs=1, cbs_states=33, fsm=4096, num_constraints=4096), dict(use_cbs=1, cbs_states=8, fsm=4096)):
        ob = L.gen_opts(**bad)
        assert L.lib.vitcap_gen_opts_check(C.byref(ob)) == -1 and L.lib.vitcap_last_error(), bad
        assert L.lib.vitcap_engine_workspace_bytes(4, C.byref(ob)) == 0
    # using the engine before binding weights is an error, not a crash
    buf = (C.c_char * 1024)()
    a = C.c_void_p((C.addressof(buf) + 255) & ~255)
    assert L.lib.vitcap_engine_prefill(h, 1, None, a, 512, None) == -4
    assert L.lib.vitcap_engine_graph_count(h) == 0
    w = L.Weights()
    assert L.lib.vitcap_engine_bind_weights(h, C.byref(w)) == -1 and b'NULL' in L.lib.vitcap_last_error()
    L.lib.vitcap_engine_destroy(h)


def test_struct_sizes_match_header(L):
    # every field is one pointer: 12/12/6 per block struct, total as laid out in vitcap_hip.h
    assert C.sizeof(L.VitBlockW) == 12 * 8 and C.sizeof(L.BertLayerW) == 12 * 8 and C.sizeof(L.LmHeadW) == 6 * 8
    assert C.sizeof(L.Weights) == (4 + 16 * 12 + 2 + 6 + 5 + 4 * 12 + 6 + 5) * 8
    # abi + 15 ints, three pointers (live, rowstat, colsum), the fused-LayerNorm block: 2 pointers, float + int, 3 pointers
    assert C.sizeof(L.GemmDesc) == 16 * 4 + 24 + 2 * 8 + 8 + 3 * 8
    # ... + the constrained-beam-search block: 4 ints, (4 bytes of alignment), 2 pointers, 16 bad-ending ids
    assert C.sizeof(L.GenOpts) == 4 + 10 * 4 + 5 * 4 + 7 * 4 + 3 * 4 + 4 + 4 * 4 + 4 + 2 * 8 + 16 * 4
    assert L.GenOpts.fsm.offset % 8 == 0 and C.sizeof(L.CbsState) == 8 * 8


def test_model_surface(L, sd_np):
    from vitcap_amd.model import ImageCaptioning
    m = ImageCaptioning(tie_weights=True).load_recipe(0)
    sd = m.state_dict()
    assert list(sd.keys()) == list(sd_np.keys()) and len(sd) == 288
    assert sd['module.bert.decoder.layer.3.attention.self.query.weight'].shape == (768, 768)
    assert sd['image_encoder.module.patch_embed.proj.weight'].shape == (768, 3, 16, 16)
    assert torch.equal(sd['module.bert.encoder.blocks.0.attn.qkv.weight'],
                       torch.from_numpy(sd_np['module.bert.encoder.blocks.0.attn.qkv.weight']))
    m2 = ImageCaptioning(tie_weights=False)
    m2.load_state_dict(sd)                     # a tied checkpoint loads into an untied model and vice versa
    assert m2.state_dict()['module.cls.predictions.decoder.weight'].data_ptr() != \
        m2.state_dict()['module.bert.embeddings.word_embeddings.weight'].data_ptr()
    m.train()
    with pytest.raises(RuntimeError, match='TrainEngine'):         # training forward needs the HIP engine attached
        m({'image': torch.zeros(1, 3, 384, 384), 'key': [0]})
    m.eval()
    for bad in ({'eos_token_ids': [102, 1012, 5, 6, 7]}, {'eos_token_ids': [102, 1012], 'num_beams': 2}, {'max_length': 41}, {'add_od_labels': False},
                {'num_return_sequences': 2}):
        keep = dict(m.test_extra_input)
        m.test_extra_input.update(bad)
        with pytest.raises(NotImplementedError):                    # unsupported generate() options are refused, not ignored
            m({'image': torch.zeros(1, 3, 384, 384), 'key': [0]})
        m.test_extra_input = keep
    # use_cbs without the machines: the reference dies on `fsm.shape` (modeling_bert.py:952); here the missing tensors are named
    keep = dict(m.test_extra_input)
    m.test_extra_input['use_cbs'] = True
    with pytest.raises(ValueError, match='fsm'):
        m({'image': torch.zeros(1, 3, 384, 384), 'key': [0]})
    m.test_extra_input = keep
    # the tag slots' start position follows generate(): max(od_labels_start_posid, max_length) (modeling_bert.py:958-959)
    assert m.gen_options().tag_pos0 == 20 and m.gen_options(od_labels_start_posid=40).tag_pos0 == 40
    assert m.gen_options(od_labels_start_posid=8, max_length=33).tag_pos0 == 33
    m.test_extra_input['num_keep_best'] = 3                        # greedy + n-best: the reference asserts (modeling_utils.py:790)
    with pytest.raises(AssertionError, match='greedy'):
        m({'image': torch.zeros(1, 3, 384, 384), 'key': [0]})


def test_recipe_is_reproducible():
    from vitcap_amd import weights as W
    a = W.gen_tensor('module.cls.predictions.bias', (30522,), 'vbias', 0)
    b = W.gen_tensor('module.cls.predictions.bias', (30522,), 'vbias', 0)
    assert (a == b).all() and W.tensor_digest(a) == W.tensor_digest(b)
    assert abs(float(a.std()) - 1.0) < 2e-2 and abs(float(a.mean())) < 2e-2 and a[102] == a.max()     # unigram-like prior, [SEP] on top
    w = W.gen_tensor('module.bert.encoder.blocks.0.attn.qkv.weight', (2304, 768), 'w', 0)
    assert abs(float(w.std()) - 0.02) < 1e-3
    t = torch.from_numpy(w)
    assert torch.equal(t.to(torch.bfloat16).float(), t), 'recipe matrices are exactly representable in bf16'
    v = W.gen_tensor('module.bert.encoder.blocks.0.attn.qkv.bias', (2304,), 'bias', 0)
    assert abs(float(v.std()) - 0.02) < 1e-3
    img4, img2 = W.gen_image_batch(4, 7), W.gen_image_batch(2, 7)
    assert (img4[:2] == img2).all() and img4.min() >= -1 and img4.max() < 1
    # the structured images of the image-dependent golden family (tests/golden/reference_imgdep.npz was produced on exactly these)
    st = W.gen_structured_images(48, 777)
    assert W.tensor_digest(st[:4]) == '1107fefeeaeba281' and st.min() >= -1 and st.max() <= 1
    assert (st.mean(axis=(1, 2, 3)).std() > 0.15), 'the images must differ in their colour offsets'
    vb = W.gen_tensor('module.cls.predictions.bias', (30522,), 'vbias', 0, 0.25)     # the untied golden's vocabulary-bias sigma
    assert abs(float(vb.std()) - 0.25) < 1e-2 and (vb * 4 == a).all()


def test_gemm_reserve_cus_host_state(L):
    """vitcap_gemm_reserve_cus (host-side atomic, no launch): returns the previous value, rounds up to whole XCD rows of 8, clamps
    negatives to 0 -- what BucketedAllReduce toggles around the buckets in flight."""
    assert L.lib.vitcap_gemm_reserve_cus(0) >= 0
    assert L.lib.vitcap_gemm_reserve_cus(13) == 0
    assert L.lib.vitcap_gemm_reserve_cus(-5) == 16
    assert L.lib.vitcap_gemm_reserve_cus(0) == 0
    assert L.lib.vitcap_set_dropout_salt(None) == 0
