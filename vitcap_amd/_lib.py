"""ctypes binding of libvitcap_hip.so (the C ABI declared in include/vitcap_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C vitcap_amd/csrc``.  There is NO
CPU fallback: if the shared object is missing or a symbol is absent, importing this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libvitcap_hip.so')

VOCAB = 30522
VOCAB_PAD = 30592
HID = 768
NVIS = 577
MAXLEN = 20
MAXLEN_CAP = 40
GEMM_AUTO, GEMM_TILES = 0, 1

ACT_NONE, ACT_GELU_ERF, ACT_TANH = 0, 1, 2
OUT_BF16, OUT_F32 = 0, 1

vp = C.c_void_p


ABI_VERSION = 6          # include/vitcap_hip.h VITCAP_ABI_VERSION: checked against vitcap_version() when the library is loaded


class GemmDesc(C.Structure):
    _fields_ = [(n, C.c_int) for n in ('abi', 'M', 'N', 'K', 'lda', 'ldw', 'ldc', 'ldr', 'act', 'out_dtype',
                                       'row_group', 'out_group_rows', 'out_row_off', 'res_periodic', 'tile_hint', 'split_k')] \
        + [('live', C.c_void_p), ('rowstat', C.c_void_p), ('colsum', C.c_void_p),
           # fused LayerNorm of the finished rows (vitcap_gemm_desc.ln_*)
           ('ln_gamma', C.c_void_p), ('ln_beta', C.c_void_p), ('ln_eps', C.c_float), ('ln_reserved', C.c_int),
           ('ln_out_bf16', C.c_void_p), ('ln_out_f32', C.c_void_p), ('ln_counters', C.c_void_p)]

    def __init__(self, *a, **kw):
        kw.setdefault('abi', ABI_VERSION)
        super().__init__(*a, **kw)


class CtItem(C.Structure):
    _fields_ = [('w', C.c_void_p), ('w_bf16', C.c_void_p), ('wt_bf16', C.c_void_p), ('N', C.c_int32), ('K', C.c_int32),
                ('ldt', C.c_int32), ('tile0', C.c_int32)]


class BeamState(C.Structure):
    _fields_ = [(n, vp) for n in ('ids_in', 'ids_out', 'beam_scores', 'parent', 'done', 'has_hyp', 'hyp_score',
                                  'hyp_len', 'hyp_tok')] + [('n_keep', C.c_int32)]


class SampleParams(C.Structure):
    _fields_ = [('do_sample', C.c_int), ('temperature', C.c_float), ('top_k', C.c_int), ('top_p', C.c_float),
                ('seed', C.c_uint32)]


class GenOpts(C.Structure):
    """vitcap_gen_opts: the kwargs of ViTCAP.generate (modeling_bert.py:928-933) + launch form, passed per call."""
    _fields_ = [('abi', C.c_int32), ('num_beams', C.c_int32), ('seqs_per_image', C.c_int32), ('num_keep_best', C.c_int32),
                ('max_length', C.c_int32), ('bos_token_id', C.c_int32), ('eos_token_id', C.c_int32),
                ('pad_token_id', C.c_int32), ('mask_token_id', C.c_int32), ('length_penalty', C.c_float),
                ('repetition_penalty', C.c_float), ('sampling', SampleParams), ('gemm_mode', C.c_int32),
                ('early_exit', C.c_int32), ('use_graph', C.c_int32), ('tag_visible', C.c_int32), ('tagemb_cls', C.c_int32),
                ('decode_streams', C.c_int32), ('encode_parts', C.c_int32), ('eos_extra', C.c_int32 * 3), ('tag_pos0', C.c_int32),
                # constrained beam search (ViTCAP.generate use_cbs / fsm / num_constraints / min_constraints_to_satisfy)
                ('use_cbs', C.c_int32), ('cbs_states', C.c_int32), ('min_constraints_to_satisfy', C.c_int32),
                ('cbs_no_repeat', C.c_int32), ('fsm', C.c_void_p), ('num_constraints', C.c_void_p), ('cbs_bad_ending', C.c_int32 * 16)]


class CbsState(C.Structure):
    """vitcap_cbs_state: device arrays of the constrained beam search bookkeeping (csrc/cbs.hip)."""
    _fields_ = [(n, C.c_void_p) for n in ('ids_in', 'ids_out', 'scores_in', 'scores_out', 'parent', 'unfinished', 'n_pred', 'live')]


class Image(C.Structure):
    _fields_ = [('rgb', C.c_void_p), ('height', C.c_int), ('width', C.c_int), ('pitch', C.c_int)]


class JpegImage(C.Structure):
    """vitcap_jpeg_image: vitcap_jpeg_info (vitcap_amd/jpegdec.py JpegInfo, include/vitcap_jpeg.h) + device pointers of the coefficient
    blocks and of the RGB image the back half writes."""
    from .jpegdec import JpegInfo as _Info
    _fields_ = [('info', _Info), ('coefs', C.c_void_p), ('rgb', C.c_void_p), ('pitch', C.c_int32)]


class TrainAug(C.Structure):
    """vitcap_train_aug: RandomResizedCrop box, ColorJitter operations in order (0 brightness, 1 contrast, 2 saturation,
    -1 none) with their factors, horizontal flip."""
    _fields_ = [('top', C.c_int32), ('left', C.c_int32), ('height', C.c_int32), ('width', C.c_int32),
                ('op', C.c_int32 * 3), ('factor', C.c_float * 3), ('flip', C.c_int32)]


class VitBlockW(C.Structure):
    _fields_ = [(n, vp) for n in ('qkv_w', 'qkv_b', 'proj_w', 'proj_b', 'fc1_w', 'fc1_b', 'fc2_w', 'fc2_b',
                                  'n1_g', 'n1_b', 'n2_g', 'n2_b')]


class BertLayerW(C.Structure):
    _fields_ = [(n, vp) for n in ('qkv_w', 'qkv_b', 'ao_w', 'ao_b', 'ao_g', 'ao_beta', 'i_w', 'i_b',
                                  'o_w', 'o_b', 'o_g', 'o_beta')]


class LmHeadW(C.Structure):
    _fields_ = [(n, vp) for n in ('dense_w', 'dense_b', 'ln_g', 'ln_b', 'dec_w', 'dec_b')]


class Weights(C.Structure):
    _fields_ = [('patch_w', vp), ('patch_b', vp), ('cls_token', vp), ('pos_embed', vp),
                ('blocks', VitBlockW * 12), ('tag_blocks', VitBlockW * 4),
                ('pooler_w', vp), ('pooler_b', vp), ('tag_logit', LmHeadW),
                ('word_emb', vp), ('pos_emb', vp), ('type_emb', vp), ('emb_ln_g', vp), ('emb_ln_b', vp),
                ('dec', BertLayerW * 4), ('cls', LmHeadW),
                ('xword_emb', vp), ('xpos_emb', vp), ('xtype_emb', vp), ('xemb_ln_g', vp), ('xemb_ln_b', vp)]


_SIGS = {
    'vitcap_last_error': (C.c_char_p, []),
    'vitcap_version': (C.c_int, []),
    'vitcap_gemm_bias_act': (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(GemmDesc), vp]),
    'vitcap_gemm_tile_plan': (C.c_int, [C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_gemm_large_form': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int]),
    'vitcap_gemm_reserve_cus': (C.c_int, [C.c_int]),
    'vitcap_set_dropout_salt': (C.c_int, [vp]),
    'vitcap_layernorm_fwd': (C.c_int, [vp, C.c_int, vp, vp, C.c_float, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_sum_layernorm': (C.c_int, [vp, C.c_int, C.c_size_t, vp, vp, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp,
                                       C.c_int, C.c_int, vp]),
    'vitcap_gemm_ex': (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(GemmDesc), vp, C.c_int, vp, C.c_int, vp]),
    'vitcap_transpose_colsum': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    'vitcap_layernorm_bwd': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_float, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_reduce_slabs': (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.c_size_t, C.c_int, vp]),
    'vitcap_cast_bf16': (C.c_int, [vp, vp, C.c_size_t, vp]),
    'vitcap_hidden_dropout': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_float, vp]),
    'vitcap_cast_bf16_colsum': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_embed_bwd': (C.c_int, [vp, vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_ls_kl_loss': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_float, C.c_int, vp, vp, vp, C.c_int, vp]),
    'vitcap_focal_loss_sum': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_float, vp, C.c_int, vp]),
    'vitcap_bce_logits_mean': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, C.c_int, vp]),
    'vitcap_sumsq': (C.c_int, [vp, C.c_size_t, vp, vp]),
    'vitcap_adamw_multi': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_int, C.c_float, C.c_float,
                                     C.c_float, C.c_size_t, vp]),
    'vitcap_gemm_tn': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_gemm_tn_sum': (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_colsum_bf16': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'vitcap_image_preproc_workspace_bytes': (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
    'vitcap_image_preproc': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    'vitcap_image_train_preproc_workspace_bytes': (C.c_size_t, [vp, vp, C.c_int, C.c_int]),
    'vitcap_image_train_preproc': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    'vitcap_jpeg_backhalf_workspace_bytes': (C.c_size_t, [vp, C.c_int]),
    'vitcap_jpeg_backhalf': (C.c_int, [vp, C.c_int, vp, C.c_size_t, vp]),
    'vitcap_resample_coeffs': (C.c_int, [C.c_int, C.c_int, vp, vp, vp, C.c_int]),
    'vitcap_resized_geometry': (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    'vitcap_cast_transpose': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_cast_transpose_multi': (C.c_int, [vp, C.c_int, C.c_int, vp]),
    'vitcap_gelu_bwd': (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
    'vitcap_sum_over_batch': (C.c_int, [vp, C.c_size_t, C.c_int, vp, C.c_size_t, vp]),
    'vitcap_embed_rows': (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_patch_gather': (C.c_int, [vp, C.c_int, vp, C.c_int, vp]),
    'vitcap_cls_rows': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_attn_dense_fwd': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_float, vp]),
    'vitcap_attn_dense_fwd_rows': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    'vitcap_attn_dense_fwd_train': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32,
                                              C.c_int, C.c_int, vp]),
    'vitcap_attn_dense_bwd': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                        C.c_uint32, C.c_int, C.c_int, vp]),
    'vitcap_attn_dense_fwd_train_rows': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_uint32,
                                                   C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_attn_dense_bwd_rows': (C.c_int, [vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                             C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_attn_decode_step': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_float, vp]),
    'vitcap_attn_decode_step_tags': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp,
                                               C.c_int, vp, vp]),
    'vitcap_tag_embed': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, vp, vp,
                                   C.c_int, vp]),
    'vitcap_copy_row_blocks': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_int, C.c_int, vp]),
    'vitcap_embed_step': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_float, vp, vp,
                                    C.c_int, vp]),
    'vitcap_greedy_init': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_greedy_step': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, vp]),
    'vitcap_greedy_select_embed': (C.c_int, [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, vp, vp, vp, vp, vp, C.c_float, vp, vp, vp]),
    'vitcap_sample_step': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, vp, vp]),
    'vitcap_sample_step_offset': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, vp, C.c_int, vp]),
    'vitcap_sigmoid_topk': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp, C.c_int, vp]),
    'vitcap_row_topk_lse': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    'vitcap_attn_beam_vt': (C.c_int, [vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_attn_decode_beams': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    'vitcap_attn_decode_beam_groups': (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp]),
    'vitcap_row_topk_pieces': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]),
    'vitcap_beam_init': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_beam_step': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_float, vp]),
    'vitcap_beam_sample_candidates': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp, vp]),
    'vitcap_beam_step_sampled': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_float, vp]),
    'vitcap_beam_reorder_cache': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_beam_finalize': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_cbs_init': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    'vitcap_cbs_start': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'vitcap_cbs_pair_flags': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    'vitcap_cbs_candidates': (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp,
                                        C.c_int, vp, vp, vp, vp, vp]),
    'vitcap_cbs_select': (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    'vitcap_cbs_finalize': (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp, vp]),
    'vitcap_assemble_visual': (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp]),
    'vitcap_gather_rows_bf16': (C.c_int, [vp, C.c_int, vp, C.c_int, C.c_int, vp]),
    'vitcap_engine_create': (C.c_int, [C.POINTER(vp)]),
    'vitcap_engine_destroy': (None, [vp]),
    'vitcap_engine_bind_weights': (C.c_int, [vp, C.POINTER(Weights)]),
    'vitcap_gen_opts_init': (None, [C.POINTER(GenOpts)]),
    'vitcap_gen_opts_check': (C.c_int, [C.POINTER(GenOpts)]),
    'vitcap_engine_workspace_bytes': (C.c_size_t, [C.c_int, C.POINTER(GenOpts)]),
    'vitcap_engine_generate': (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(GenOpts), vp, C.c_size_t, vp, vp, vp, vp, vp]),
    'vitcap_engine_encode': (C.c_int, [vp, vp, C.c_int, C.c_int, C.POINTER(GenOpts), vp, C.c_size_t, vp]),
    'vitcap_engine_prefill': (C.c_int, [vp, C.c_int, C.POINTER(GenOpts), vp, C.c_size_t, vp]),
    'vitcap_engine_decode': (C.c_int, [vp, C.c_int, C.POINTER(GenOpts), vp, C.c_size_t, vp, vp, vp, vp]),
    'vitcap_engine_tags': (C.c_int, [vp, C.c_int, C.POINTER(GenOpts), vp, vp, vp, vp]),
    'vitcap_engine_graph_count': (C.c_int, [vp]),
    'vitcap_engine_tap': (vp, [vp, C.c_char_p, vp, C.c_int, C.POINTER(GenOpts)]),
    'vitcap_repetition_penalty': (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_float, C.c_int, vp]),
    'vitcap_engine_timing_begin': (C.c_int, [vp, C.c_int]),
    'vitcap_engine_timing_sample': (C.c_int, [vp, C.c_int]),
    'vitcap_engine_timing_end': (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'vitcap_engine_timing_end_ex': (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                              C.POINTER(C.c_double)]),
    'vitcap_engine_timing_end_kernel': (C.c_int, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int),
                                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
}

EXPORTS = tuple(_SIGS)


class VitcapError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libvitcap_hip.so is not built (%s). Run `python -c "import __graft_entry__ as g; g.build()"` or '
            '`make -C vitcap_amd/csrc`. There is no CPU fallback for the product path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    got = lib.vitcap_version()
    if got != ABI_VERSION:         # a stale .so (or binding): struct layouts / signatures differ, every call would be misread
        raise ImportError('libvitcap_hip.so reports ABI version %d, this binding is written for %d: rebuild (make -C vitcap_amd/csrc)'
                          % (got, ABI_VERSION))
    return lib


lib = _load()


def gen_opts(**kw):
    """vitcap_gen_opts with the reference's test-time defaults (..._bertemb.py:588-608), fields overridden by keyword."""
    o = GenOpts()
    lib.vitcap_gen_opts_init(C.byref(o))
    for k, v in kw.items():
        if k == 'sampling':
            o.sampling = v
        elif k == 'eos_extra':
            ids = list(v) + [-1] * (3 - len(v))
            o.eos_extra = (C.c_int32 * 3)(*[int(x) for x in ids[:3]])
        elif k == 'cbs_bad_ending':
            ids = list(v) + [-1] * 16
            o.cbs_bad_ending = (C.c_int32 * 16)(*[int(x) for x in ids[:16]])
        else:
            if not hasattr(o, k):
                raise AttributeError('vitcap_gen_opts has no field %r' % k)
            setattr(o, k, v)
    return o


def check(rc, what=''):
    if rc != 0:
        raise VitcapError('%s failed (%d): %s' % (what or 'vitcap call', rc, lib.vitcap_last_error().decode()))
