"""Pipeline surface of the reference (run.py -> param.pipeline_type -> CaptionUniPipeline) for the captioning hot path.

Mirrors ``src/pipelines/uni_pipeline.py`` (UniPipeline.ensure_train / ensure_predict / ensure_evaluate / monitor_train
/ is_train_finished / full_expid, lines 342-378, 673-697, 884-911, 1021-1038) and
``src/pipelines/tagger_caption_uni_pipeline_expanding_bertemb.py`` (CaptionUniPipeline defaults 195-228,
get_raw_model 566-618, predict_output_to_tsv_row 620-630) for what the hot path needs:

* same YAML keys and defaults, ``Config`` semantics (unknown key -> None);
* ``ensure_predict``: builds the HIP ``ImageCaptioning`` model, loads ``basemodel`` / the latest snapshot with the
  reference's suffix-matching loader, captions the test split, writes ``<model>.…predict.tsv`` rows
  ``key \\t json([{"caption": str, "conf": exp(logprob)}])`` per rank and merges them on rank 0;
* the input side: ``test_data`` names a TSV of (key, base64 JPEG) rows read through ``.lineidx``
  (tsv.py), decoded by worker processes on the host (jpegdec.py) and resized / cropped / normalised on the GPU bit-identical to
  Pillow (csrc/preproc.hip); ``data: synthetic`` (seeded uniform(-1,1) images) and an iterable passed as ``test_batches`` remain;
* ``ensure_train`` runs the HIP training engine (vitcap_amd/train.py) on synthetic or caller-provided batches and
  writes reference-format snapshots; ``ensure_evaluate`` needs the external
  coco-caption tools the reference does not vendor either and is a logged no-op without them.
"""
import copy
import json
import logging
import os
import os.path as op
import time

import torch

from . import dist_util as D



def _release_slab(shm):
    """close() raises BufferError while a numpy view of the segment is still exported; the name must be unlinked all the same, or the
    segment outlives the process until the resource tracker reaps it."""
    try:
        shm.close()
    except Exception:
        pass
    try:
        shm.unlink()
    except Exception:
        pass


def _pin_slabs(slabs):
    """hipHostRegister on every shared-memory slab -> list of registered base addresses (empty when the runtime refuses: the copies
    then take the pageable path, slower but correct).  VITCAP_LOADER_PIN=0 disables."""
    import ctypes
    if os.environ.get('VITCAP_LOADER_PIN', '1') == '0':
        return []
    done = []
    try:
        rt = torch.cuda.cudart()
        for shm in slabs:
            addr = ctypes.addressof(ctypes.c_char.from_buffer(shm.buf))
            err = rt.cudaHostRegister(addr, shm.size, 0)
            if int(getattr(err, 'value', err)) != 0:
                logging.warning('hipHostRegister refused a loader slab (error %s): host -> device copies stay pageable', err)
                break
            done.append(addr)
    except Exception as e:          # noqa
        logging.warning('loader slabs not page-locked (%s)', e)
    return done


def _unpin_slabs(addrs):
    try:
        rt = torch.cuda.cudart()
        for a in addrs:
            rt.cudaHostUnregister(a)
    except Exception:
        pass


def _prefetched(gen, device, depth=2):
    """Runs the batch generator in a background thread, `depth` batches ahead: reading the TSV rows, handing them to the decode workers,
    the host -> device copies and the launch of the transform kernel no longer sit between two caption launches of the consumer
    (measured: the consumer thread was the limit at 26 ms per batch of 64, the GPU needs 17).  A batch is handed over after the producer's
    stream has been synchronised, so the consumer may use it on any stream."""
    import queue
    import threading
    q = queue.Queue(maxsize=max(1, depth))
    END = object()
    stop = threading.Event()            # set when the consumer goes away (exception, generator closed): the producer must not block in put

    def put(item):
        while not stop.is_set():
            try:
                q.put(item, timeout=0.2)
                return True
            except queue.Full:
                continue
        return False

    def work():
        try:
            if device.type == 'cuda':
                torch.cuda.set_device(device)
            ev = torch.cuda.Event(blocking=True) if device.type == 'cuda' else None     # sleep, do not spin (see generate_async)
            # the producer's copies and transform kernels go to a stream of their own: on the (per-process) default stream they shared a
            # queue with the consumer's blocking device -> host copies of the finished captions, and each side waited behind the other's
            # work -- with 24 batches of prefetch a `.cpu()` of 5 KB could sit behind a whole batch of host -> device copies
            # (docs/LAB_r05.md 5: the 22 ms-per-batch mode of the round script's input-side runs)
            from .model import role_stream
            side = role_stream(device, 'loader', lambda: torch.cuda.Stream(device)) if device.type == 'cuda' else None
            if side is not None:
                torch.cuda.set_stream(side)          # thread-local: this thread only
            for b in gen:
                t0 = time.perf_counter()
                if ev is not None:
                    ev.record(torch.cuda.current_stream(device))
                    ev.synchronize()
                t1 = time.perf_counter()
                ok = put(b)
                LOADER_TIMES['copy_sync'] = LOADER_TIMES.get('copy_sync', 0.0) + t1 - t0
                LOADER_TIMES['queue_full'] = LOADER_TIMES.get('queue_full', 0.0) + time.perf_counter() - t1
                if not ok:
                    break
            else:
                put(END)
        except BaseException as e:          # surfaces in the consumer
            put(e)
        finally:
            close = getattr(gen, 'close', None)     # runs the generator's own `finally` (worker pool, shared-memory slabs) in this thread
            if close is not None:
                try:
                    close()
                except Exception:
                    pass
    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            t0 = time.perf_counter()
            b = q.get()
            LOADER_TIMES['queue_empty'] = LOADER_TIMES.get('queue_empty', 0.0) + time.perf_counter() - t0
            if b is END:
                break
            if isinstance(b, BaseException):
                raise b
            yield b
    finally:
        stop.set()
        t.join(timeout=30)


LOADER_TIMES = {}           # seconds, summed over the batches of the last predict: where the producer thread and the consumer waited
LAST_PREDICT_STATS = {}     # filled by CaptionUniPipeline.predict: rows, steady-state images/sec of this rank (tools/input_side_bench.py)
from .checkpoint import Checkpointer
from .config import Config

_UNI_DEFAULT = {   # the subset of UniPipeline._default (uni_pipeline.py:93-148) the hot path reads
    'dist_backend': 'nccl', 'effective_batch_size': 256, 'test_batch_size': 48, 'max_iter': 10, 'log_step': 100,
    'base_lr': 1e-4, 'weight_decay': 0.05, 'snapshot_steps': 5000, 'test_crop_size': 384, 'train_crop_size': 384,
    'force_train': False, 'force_predict': False, 'ignore_predict': False, 'use_amp': False, 'random_seed': 88,
    'expid_prefix': 'Jacob', 'num_workers': 8,
}
_CAPTION_DEFAULT = {   # CaptionUniPipeline._default.update (..._bertemb.py:195-228)
    'mask_type': 'seq2seq', 'max_seq_a_length': 40, 'max_seq_length': 70, 'add_od_labels': True, 'drop_out': 0.1,
    'tie_weights': True, 'label_smoothing': 0.1, 'img_layer_norm_eps': 1e-5, 'max_img_seq_length': 50,
    'max_gen_length': 20, 'max_masked_tokens': 3, 'num_beams': 1, 'mask_prob': 0.15, 'replace_by_mask_prob': 0.8,
    'replace_by_rand_prob': 0.1, 'temperature': 1, 'top_k': 0, 'top_p': 1, 'gradient_clip': 1.,
    'optimizer_type': 'MAdamW', 'bias_no_weight_decay': True, 'ln_no_weight_decay': True, 'scheduler_type': 'linear',
    'pad_to_max': True, 'pert_img_prob': None,
}


# Configuration keys that select another model or training rule in the reference and that this build implements for ONE value (the
# shipped YAML's): a config that sets another value is refused, not run as if it had not been said.  (key, accepted explicit
# values, what the reference does with it.)  A key the config does not mention is accepted: the built value is then the default.
_BUILT_FOR = (
    ('image_encoder_type', ('VitEmb_vit_base_patch16_384',), 'config.net: another timm ViT (..._bertemb.py:545)'),
    ('split_blocks', (4, '4'), 'depth of the tag branch (config.split_blocks, modeling_bert.py:440-478)'),
    ('topk', (50,), 'tag tokens kept per image (modeling_bert.py:1424-1432); None selects the threshold mode'),
    ('use_img_layernorm', (False, 0), 'LayerNorm on the image features (config.use_img_layernorm)'),
    ('tie_tag_weights', (False, 0), 'tag head tied to the word embeddings (modeling_bert.py:724-726)'),
    ('mask_type', ('seq2seq',), 'attention mask family (dataset.py:377-417)'),
    ('category', ('bert',), "tag vocabulary: 'vinvl' maps tags through tokenizer_file into another label space (config.category)"),
    ('train_transform', ('vit',), 'training image transform family (get_transform, ..._bertemb.py:373-518)'),
    ('scheduler_type', ('linear',), 'LR schedule (..._bertemb.py:358-371)'),
    ('optimizer_type', ('MAdamW',), 'optimizer (..._bertemb.py:346-356)'),
    ('bias_no_weight_decay', (True, 1), 'parameter groups (..._bertemb.py:280-322)'),
    ('ln_no_weight_decay', (True, 1), 'parameter groups (..._bertemb.py:280-322)'),
    ('later_captioning', (None, False, 0), 'config.later_captioning'),
    ('attn_token_sample', (None, False, 0), 'config.attn_token_sample'),
    ('topktagger', (None, False, 0), 'config.topktagger'),
    ('tagemb_gradient', (None, False, 0), 'config.tagemb_gradient'),
    ('pert_img_prob', (None, 0, 0.0), 'image perturbation during training'),
    ('use_amp', (False, 0), 'apex / autocast mixed precision; this build computes in bf16 with fp32 accumulation throughout'),
)


def check_model_config(cfg, training):
    """Refuses configurations that name a model variant or training rule this build does not implement (see _BUILT_FOR).
    `drop_out` (-> BertConfig.hidden_dropout_prob, ..._bertemb.py:535; 0 in the shipped YAML, 0.1 by the pipeline's own default) is
    built since round 4 (TrainEngine(hidden_dropout=...), csrc/train.hip vitcap_hidden_dropout); it must be a probability."""
    check_text_encoder_config(cfg.text_encoder_type)
    given = cfg.overwrite
    for key, ok, what in _BUILT_FOR:
        if key in given and given[key] not in ok:
            raise NotImplementedError('%s: %r is not built (accepted: %s) -- %s' % (key, given[key], ', '.join(repr(v) for v in ok), what))
    if training and not (0.0 <= float(cfg.drop_out or 0) < 1.0):
        raise ValueError('drop_out: %r is not a dropout probability' % (cfg.drop_out,))


_BERT_CONFIG_BUILT = {   # BertConfig fields of <text_encoder_type>/config.json the kernels are sized for (the shipped VILT-L12-H784 values)
    'hidden_size': 768, 'num_attention_heads': 12, 'intermediate_size': 3072, 'vocab_size': 30522, 'max_position_embeddings': 512,
    'type_vocab_size': 2, 'layer_norm_eps': 1e-12, 'hidden_act': 'gelu',
}


def check_text_encoder_config(text_encoder_type):
    """<text_encoder_type>/config.json, when present, must describe the BERT geometry this build implements (num_hidden_layers is
    overridden to 4 by the reference itself, modeling_bert.py:1342-1346); another geometry is refused, not mis-run."""
    import json
    cj = op.join(text_encoder_type or '.', 'config.json')
    if not op.isfile(cj):
        return
    with open(cj) as fp:
        cfg = json.load(fp)
    for k, want in _BERT_CONFIG_BUILT.items():
        if k in cfg and cfg[k] != want:
            raise NotImplementedError('%s: %s = %r, this build implements %r' % (cj, k, cfg[k], want))


class _EngineState(object):
    """state_dict() / load_state_dict() views of the training engine for Checkpointer (same keys as the reference
    checkpoint: 'model' / 'optimizer' / 'scheduler', src/tools/opt/checkpoint.py:60-102)."""

    def __init__(self, eng, what='model'):
        self.eng, self.what = eng, what

    def state_dict(self):
        if self.what == 'optimizer':
            return self.eng.optimizer_state_dict()
        if self.what == 'scheduler':
            return self.eng.scheduler_state_dict()
        return {k: v.cpu() for k, v in self.eng.state_dict().items()}

    def load_state_dict(self, sd, strict=False):
        if self.what == 'optimizer':
            return self.eng.load_optimizer_state_dict(sd)
        if self.what == 'scheduler':
            return self.eng.load_scheduler_state_dict(sd)
        self.eng.load_model_state_dict(sd)
        return torch.nn.modules.module._IncompatibleKeys([], [])


class CaptionUniPipeline(object):
    def __init__(self, **kwargs):
        self._default = copy.deepcopy(_UNI_DEFAULT)
        self._default.update(_CAPTION_DEFAULT)
        self.cfg = Config(self._default, dict(kwargs))
        self.rank, self.world, self.local_rank = D.env_rank_world()
        self._tokenizer = None
        self._initialized = False
        self._n_train = None

    # ------------------------------------------------------------------ naming (uni_pipeline.py:150-170)
    @property
    def full_expid(self):
        return self.cfg.full_expid or '_'.join(str(x) for x in (self.cfg.expid_prefix, self.cfg.data, self.cfg.net,
                                                                  self.cfg.expid) if x)

    @property
    def output_folder(self):
        return op.join('output', self.full_expid)

    def get_snapshot_dir(self):
        return op.join(self.output_folder, 'snapshot')

    def parse_iter(self, i):
        """uni_pipeline.py:253-261: an integer, or '<x>e' = x epochs = int(x * number of training samples / effective_batch_size);
        the samples are the (image, caption) pairs of the training split's caption index (CaptionIdxTSVDataset).  The shipped
        YAML trains for `max_iter: 30e`."""
        if isinstance(i, str) and i.endswith('e'):
            if not self.cfg.data or self.cfg.data == 'synthetic':
                raise ValueError("max_iter given in epochs ('%s') needs a training set (`data: <name>`); use an integer with synthetic data" % i)
            if self._n_train is None:
                from .dataset import CaptionIdx
                self._n_train = len(CaptionIdx(self.cfg.data_root or 'data', self.cfg.data, 'train', self.cfg.train_version))
            return int(float(i[:-1]) * (1. * self._n_train / int(self.cfg.effective_batch_size)))
        return int(i)

    def get_checkpoint_file(self, iteration=None):
        """uni_pipeline.py:614-622: `model_file` if given (and no iteration asked for), else snapshot/model_iter_<max_iter>.pt.
        `basemodel` is the training INIT only (..._bertemb.py:259-267) and never the file that is evaluated."""
        if iteration is None and self.cfg.model_file is not None:
            return self.cfg.model_file
        it = self.parse_iter(self.cfg.max_iter if iteration is None else iteration)
        return op.join(self.get_snapshot_dir(), 'model_iter_{:07d}.pt'.format(it))

    def is_train_finished(self):
        return op.isfile(self.get_checkpoint_file())

    def get_predict_file(self, model_file):
        cc = [model_file, self.cfg.test_data or 'synthetic', self.cfg.test_split or 'test']
        if self.cfg.num_beams not in (None, 1):
            cc.append('beam{}'.format(self.cfg.num_beams))
        return '.'.join(cc) + '.predict.tsv'

    # ------------------------------------------------------------------ process group (torch_common.py:125-142)
    def _ensure_initialized(self):
        if self._initialized:
            return
        if self.world > 1:
            backend = self.cfg.dist_backend if torch.cuda.is_available() else 'gloo'
            D.init(backend, torch.device('cuda', self.local_rank) if backend == 'nccl' else None)
        if torch.cuda.is_available():
            torch.cuda.set_device(self.local_rank)
        # host threads sized by what the job may use, not by what the machine shows (D.host_cpu_budget); the loader's decode
        # processes keep their cores
        D.cap_host_threads(reserved=int(self.cfg.num_workers or 0))
        self._initialized = True

    # ------------------------------------------------------------------ model / tokenizer
    @property
    def tokenizer(self):
        if self._tokenizer is None:
            from .tokenizer import CaptionDetokenizer
            vf = op.join(self.cfg.text_encoder_type or '.', 'vocab.txt')
            if not op.isfile(vf):
                raise FileNotFoundError('BERT vocab not found at %s (text_encoder_type: %s)' % (vf, self.cfg.text_encoder_type))
            self._tokenizer = CaptionDetokenizer(vf)
        return self._tokenizer

    def get_raw_model(self, is_train):
        from .model import ImageCaptioning
        if is_train:       # ensure_train attaches the TrainEngine; the module itself is the same tree in train mode
            return ImageCaptioning(tie_weights=bool(self.cfg.tie_weights), tagemb=self.cfg.tagemb or 'bert', cfg=self.cfg).train()
        extra = {'max_length': self.cfg.max_gen_length, 'num_beams': self.cfg.num_beams,
                 'temperature': self.cfg.temperature, 'top_k': self.cfg.top_k, 'top_p': self.cfg.top_p,
                 'add_od_labels': self.cfg.add_od_labels, 'od_labels_start_posid': self.cfg.max_seq_a_length}
        for k in ('repetition_penalty', 'length_penalty', 'num_keep_best', 'do_sample', 'num_return_sequences', 'use_cbs',
                  'use_graph'):          # optional YAML keys forwarded to generate() when present
            if getattr(self.cfg, k) is not None:
                extra[k] = getattr(self.cfg, k)
        return ImageCaptioning(tie_weights=bool(self.cfg.tie_weights), tagemb=self.cfg.tagemb or 'bert',
                               test_extra_input=extra, cfg=self.cfg).eval()

    def load_test_model(self, model, model_file):
        if model_file and op.isfile(model_file):
            Checkpointer(model=model, save_dir=self.get_snapshot_dir()).load(model_file, model_only=True,
                                                                              load_if_has=False)
        elif self.cfg.init_recipe_seed is not None:
            model.load_recipe(int(self.cfg.init_recipe_seed))     # synthetic runs: seeded random init
        else:
            raise FileNotFoundError('no model file: {}'.format(model_file))
        return model

    # ------------------------------------------------------------------ entry points used by run.py
    def iter_train_batches(self, per_gpu, start_iter=0):
        """`data: synthetic` -> endless seeded batches with the training collate's layout (per-rank seeds).  `start_iter`
        batches are skipped so that a resumed job sees the batches the uninterrupted one would have seen."""
        if self.cfg.train_batches is not None:
            n = 0
            while True:
                for b in self.cfg.train_batches:
                    n += 1
                    if n > start_iter:
                        yield b
        if self.cfg.data and self.cfg.data != 'synthetic':
            for b in self.real_train_batches(per_gpu, start_iter):
                yield b
            return
        from . import weights as W
        from .synthetic import synthetic_train_inputs
        it = start_iter
        while True:
            seed = D.shard_seed(int(self.cfg.synthetic_seed or 1234), self.rank) + 1000 * it
            batch = synthetic_train_inputs(per_gpu, seed=seed)
            batch['image'] = torch.from_numpy(W.gen_image_batch(per_gpu, seed))
            if self.cfg.scst:      # ground-truth captions for the CIDEr-D reward: the (synthetic) caption tokens as text
                ids = batch['input_ids'][:, :20]
                batch['captions'] = [[self.tokenizer.decode(r.tolist(), skip_special_tokens=True)] for r in ids]
            yield batch
            it += 1

    def real_train_batches(self, per_gpu, start_iter=0):
        """`data: <name>` -> data/<name>/train.tsv (+ .caption.tsv, optional .label / .num_caption): the reference's training
        transform chain (get_transform(is_train=True), ..._bertemb.py:373-518) via vitcap_amd/dataset.py; images are cropped,
        resized, jittered, flipped and normalised on this rank's GPU."""
        from .dataset import CaptionTrainSet, TagLabelTensorizer, TrainBatchLoader
        from .augment import TrainAugmentation
        from .imageio import TrainImagePreprocessor
        from .tensorizer import CaptionTensorizer
        from .tokenizer import BertWordPieceTokenizer
        c = self.cfg
        if int(c.max_seq_a_length) > 20:
            # not a size limit: ViTSplitCLSEmbModel.forward hard-codes 20 caption slots -- `topk_len[0] + 20 <= L` and
            # `embedding_output[:, -50:] = tag_embedding` (modeling_bert.py:1435, 1467) overwrite rows 20..69 of the 70 text rows,
            # i.e. the caption slots 20..max_seq_a_length-1 themselves; the shipped YAML sets 20 for that reason
            raise NotImplementedError('max_seq_a_length = %s: the reference model itself only works with 20 caption slots (its forward '
                                      'writes the 50 tag embeddings over text rows 20..69, modeling_bert.py:1435-1467); set '
                                      'max_seq_a_length: 20 as the shipped YAML does' % c.max_seq_a_length)
        vf = op.join(c.text_encoder_type or '.', 'vocab.txt')
        tok = BertWordPieceTokenizer(vf)
        tz = CaptionTensorizer(tok, max_img_seq_length=int(c.max_img_seq_length), max_seq_length=int(c.max_seq_length),
                               max_seq_a_length=int(c.max_seq_a_length), mask_prob=float(c.mask_prob),
                               max_masked_tokens=int(c.max_masked_tokens), mask_type=c.mask_type, is_train=True,
                               mask_b=bool(c.mask_b), replace_by_mask_prob=float(c.replace_by_mask_prob),
                               replace_by_rand_prob=float(c.replace_by_rand_prob), ignore_sep=bool(c.ignore_sep))
        tagger = TagLabelTensorizer(tok, threshold=float(c.od_label_conf if c.od_label_conf is not None else 0.2),
                                    encode=c.encode if c.encode is not None else 'nltk', caption_only=bool(c.caption_only),
                                    pos_tagger=c.pos_tagger)
        seed = int(c.random_seed or 0)
        ds = CaptionTrainSet(c.data_root or 'data', c.data, tz, tagger, split='train', caption_version=c.train_version,
                             label_version=c.train_label_version,
                             augmentation=TrainAugmentation(seed=seed, small_scale=c.input_small_scale), device_jpeg=c.device_jpeg)
        tf = TrainImagePreprocessor(torch.device('cuda', self.local_rank), train_crop_size=int(c.train_crop_size))
        return TrainBatchLoader(ds, per_gpu, tf, rank=self.rank, world=self.world, seed=seed, workers=int(c.num_workers),
                                want_captions=bool(c.scst), start_iter=start_iter)

    def ensure_train(self):
        """do_train_dict (trainer.py:33-213) on the HIP training engine: per-GPU batch = effective_batch_size // world,
        AdamW on the reference's parameter groups, linear LR decay, snapshot every snapshot_steps and at max_iter."""
        check_model_config(self.cfg, training=True)
        self._ensure_initialized()
        last = self.get_checkpoint_file()
        if op.isfile(last) and not self.cfg.force_train:
            logging.info('skip to train')
            return last
        from .model import ImageCaptioning
        from .train import TrainEngine
        dev = torch.device('cuda', self.local_rank)
        model = ImageCaptioning(tie_weights=bool(self.cfg.tie_weights), tagemb=self.cfg.tagemb or 'bert', cfg=self.cfg)
        if self.cfg.basemodel and op.isfile(self.cfg.basemodel):
            Checkpointer(model=model).load(self.cfg.basemodel, model_only=True, load_if_has=False)
            sd = model.state_dict()                      # tag_blocks <- copy of blocks[-4:] (..._bertemb.py:265-267)
            for i in range(4):
                for k in [k for k in sd if k.startswith('module.bert.encoder.blocks.%d.' % (8 + i))]:
                    sd[k.replace('blocks.%d.' % (8 + i), 'tag_blocks.%d.' % i)].copy_(sd[k])
        elif self.cfg.init_recipe_seed is not None:
            model.load_recipe(int(self.cfg.init_recipe_seed))
        else:
            raise FileNotFoundError('basemodel not found: {}'.format(self.cfg.basemodel))
        max_iter = self.parse_iter(self.cfg.max_iter)
        dist = None
        if self.world > 1:
            import torch.distributed as dist
        # BertConfig.from_pretrained(text_encoder_type) carries attention_probs_dropout_prob (0.1 in the shipped config);
        # only hidden_dropout_prob is overridden by cfg.drop_out (..._bertemb.py:535)
        attn_drop = 0.1
        cj = op.join(self.cfg.text_encoder_type or '.', 'config.json')
        if op.isfile(cj):
            import json
            with open(cj) as fp:
                attn_drop = float(json.load(fp).get('attention_probs_dropout_prob', attn_drop))
        if self.cfg.attention_probs_dropout_prob is not None:
            attn_drop = float(self.cfg.attention_probs_dropout_prob)
        eng = TrainEngine(model, dev, base_lr=float(self.cfg.base_lr), weight_decay=float(self.cfg.weight_decay),
                          lr_multiplier=float(self.cfg.lr_multiplier or 1.0), clip=float(self.cfg.gradient_clip),
                          max_iter=max_iter, label_smoothing=float(self.cfg.label_smoothing), dist=dist,
                          attn_dropout=attn_drop, hidden_dropout=float(self.cfg.drop_out or 0), dropout_seed=int(self.cfg.random_seed or 0),
                          tag_loss='focal' if self.cfg.loss == 'focal' else 'bce')     # modeling_bert.py:713-717
        # graph mode (train.py train_step_graph): the cross-entropy step replays hipGraph segments captured once per batch shape -- the
        # host issues ~11 calls per step instead of ~750 (41.7 -> 0.9-8 ms of host time per step, profiles/r05_train_graph.txt), which is
        # what keeps 8 ranks x (loader processes + launcher) on one host from serialising on Python.  `train_graph: false` keeps eager.
        eng.use_graphs = self.cfg.train_graph is None or bool(self.cfg.train_graph)
        per_gpu = max(1, int(self.cfg.effective_batch_size) // self.world)
        ckpt = Checkpointer(model=_EngineState(eng), optimizer=_EngineState(eng, 'optimizer'),
                            scheduler=_EngineState(eng, 'scheduler'), save_dir=self.get_snapshot_dir(),
                            save_to_disk=self.rank == 0)
        # resume (checkpoint.py recover_or_load + trainer.py:95 start_iter): an interrupted job continues from the latest
        # snapshot named by snapshot/last_checkpoint -- parameters, AdamW moments, step count and LR schedule
        start_iter = 0
        if ckpt.has_checkpoint() and not self.cfg.force_train:
            extra = ckpt.load()
            start_iter = int(extra.get('iteration', 0))
            eng.sync_from_rank0()       # every rank continues from rank 0's parameters and moments, whatever file it found
            logging.info('resuming from %s at iteration %d', ckpt.get_checkpoint_file(), start_iter)
        # trainer.py:134-137: a NaN loss saves `NaN_context_<rank>` through the checkpointer and raises (TrainEngine.flush_nan_check)
        def nan_dump(name):
            # every rank writes its own context (the file name carries the rank); `last_checkpoint` keeps naming the last good
            # snapshot -- the reference's save() re-tags it to the NaN context on rank 0, which a resume would then load
            c = Checkpointer(model=_EngineState(eng), optimizer=_EngineState(eng, 'optimizer'), scheduler=_EngineState(eng, 'scheduler'),
                             save_dir=self.get_snapshot_dir(), save_to_disk=True)
            prev = c.get_checkpoint_file() if c.has_checkpoint() else None
            f = c.save(name, iteration=eng.step_no)
            if prev:
                c.tag_last_checkpoint(prev)
            elif op.isfile(op.join(c.save_dir, 'last_checkpoint')):
                os.remove(op.join(c.save_dir, 'last_checkpoint'))
            logging.info('NaN context saved to %s', f)
        eng.nan_dump = nan_dump
        t0, log_step = time.time(), int(self.cfg.log_step)
        batches = self.iter_train_batches(per_gpu, start_iter=start_iter)
        scst = None
        if self.cfg.scst:          # BASELINE config 5 (..._expanding.py:404-478): self-critical step instead of cross-entropy
            from .scst import ScstTrainer
            scst = ScstTrainer(model, eng, self.tokenizer, num_return=int(self.cfg.scst_num_return or 5),
                               seed=int(self.cfg.random_seed or 0) + self.rank)
        for it in range(start_iter + 1, max_iter + 1):
            b = next(batches)
            b = {k: (v.to(dev, non_blocking=True) if torch.is_tensor(v) else v) for k, v in b.items()}
            if scst is not None:
                o = scst.step(b['image'].to(torch.bfloat16).contiguous(), b['captions'])
                out = {'masked_loss': o['scst_loss']}
            else:
                out = eng.train_step(b)
            if it % log_step == 0 or it == max_iter:
                torch.cuda.synchronize()
                dt = time.time() - t0
                n = log_step if it % log_step == 0 else (it % log_step)
                logging.info('iter %d  masked_loss %.4f  speed: %.1f images/sec', it, float(out['masked_loss']),
                             self.world * n * per_gpu / dt)
                t0 = time.time()
            if it % int(self.cfg.snapshot_steps) == 0 or it == max_iter:
                eng.flush_text_check()          # no snapshot from steps trained on a mask the kernels do not implement
                ckpt.save('model_iter_{:07d}'.format(it), iteration=it)
            if self.cfg.stop_after_iter is not None and it >= int(self.cfg.stop_after_iter):
                logging.info('stop_after_iter=%s: leaving the training loop early (resume test / pre-emption drill)', self.cfg.stop_after_iter)
                break
        eng.flush_text_check()                  # the steps since the last periodic read
        if dist is not None:
            dist.barrier()      # uni_pipeline.py:376 synchronize(): no rank may reach ensure_predict before rank 0's final snapshot exists
        return self.get_checkpoint_file(iteration=max_iter)

    def iter_test_batches(self):
        """Per-rank shard of the test set.  `data: synthetic` -> seeded images, keys '<rank>_<i>'."""
        if self.cfg.test_batches is not None:
            for i, b in enumerate(self.cfg.test_batches):
                if i % self.world == self.rank:
                    yield b
            return
        tsv = self.test_image_tsv()
        if tsv is not None:
            # (key, base64 JPEG) rows; rank r takes rows r, r+world, ... like DistributedSampler(shuffle=False)
            # (uni_pipeline.py:782-850); decode on the host, transform on the GPU (csrc/preproc.hip)
            from .imageio import ImagePreprocessor
            from .tsv import TSVFile
            tsv = op.abspath(tsv)
            rows = TSVFile(tsv)
            pre = ImagePreprocessor(torch.device('cuda', self.local_rank), int(self.cfg.test_crop_size),
                                    float(self.cfg.crop_pct or 1.0))
            bs = int(self.cfg.test_batch_size)
            mine = list(range(self.rank, len(rows), self.world))
            # JPEG decoding is host work: `num_workers` worker PROCESSES (the reference's DataLoader(num_workers),
            # uni_pipeline.py:333-339) decode the images of the next batches while the GPU captions the current one.  Measured on the
            # GPU box (tools/input_side_bench.py, profiles/r04_input_side.json): one core decodes 600 images/s, threads of one process
            # top out near 2 000 (GIL), a GPU captions 3 700.  The workers are SPAWNED (never forked: this process owns a GPU
            # context) and import vitcap_amd.jpegdec only.  `loader_threads: true` keeps the decode in threads of this process.
            # Rows are read by the WORKERS (round 5: a task names the file and its row numbers), batches come back in order.
            from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor
            from .jpegdec import coef_item, decode_rows, decode_rows_into, jpeg_lib
            from .imageio import CoefImage
            # device JPEG back half (round 6): the workers only entropy-decode baseline JPEGs (libvitcap_jpeg.so); dequantisation, inverse
            # DCT, upsampling and the colour conversion run on the GPU in front of the resize, bit-identical to Pillow.  `device_jpeg:
            # false` (or a missing library) keeps the whole decode in the workers.
            device_jpeg = (self.cfg.device_jpeg is None or bool(self.cfg.device_jpeg)) and jpeg_lib() is not None and not self.cfg.loader_threads
            import numpy as np
            workers = max(1, int(self.cfg.num_workers or 1))
            chunk = int(os.environ.get('VITCAP_LOADER_CHUNK', 8))      # images per worker task
            per_batch = (bs + chunk - 1) // chunk
            # batches being decoded ahead of the one handed out: ten tasks per worker.  Decode times jitter (image sizes, a host shared
            # with other tenants) and the consumer can never run faster than the GPU to make up for a gap, so every gap is lost for good:
            # with 3 batches ahead + 4 queued 0.93-0.94 of the resident rate, with 10 + 24 0.96-0.975 (profiles/r05_input_side.json)
            ahead = min(12, max(3, (10 * workers + per_batch - 1) // per_batch))     # <= 15 batches of slabs (2.9 GB nominal) however many workers
            ahead = int(os.environ.get('VITCAP_LOADER_AHEAD', ahead))
            starts = list(range(0, len(mine), bs))
            slabs, free = [], []
            if self.cfg.loader_threads:
                pool = ThreadPoolExecutor(max_workers=workers)
            else:
                import multiprocessing as mp
                from multiprocessing import shared_memory
                ctx = mp.get_context('spawn')
                # VITCAP_LOADER_CPUS=<first>[:<stride>] (experiment): worker i is pinned to CPU first + i * stride (default stride 1) -- under a
                # CFS quota on a many-core host the scheduler otherwise migrates the decoders across the whole machine
                pin = os.environ.get('VITCAP_LOADER_CPUS')
                if pin:
                    from .jpegdec import pin_worker
                    first, _, stride = pin.partition(':')
                    pool = ProcessPoolExecutor(max_workers=workers, mp_context=ctx, initializer=pin_worker,
                                               initargs=(ctx.Value('i', 0), int(first), int(stride or 1)))
                else:
                    pool = ProcessPoolExecutor(max_workers=workers, mp_context=ctx)
                # decoded pixels come back through shared memory (vitcap_amd/jpegdec.py); a slab holds one task's images and is reused
                # two batches after the batch that read it went to the GPU
                slab_bytes = int(self.cfg.loader_slab_mb or 24) << 20
                # the slabs are page-locked shared memory (x ranks on one host): the decode-ahead depth is capped in BYTES, not batches
                # (at test_batch_size 512 the batch-count rule asked for 9-23 GB per rank).  `loader_shm_gb` (default 4); the
                # floor is one batch ahead + the three in retirement.
                budget = int(float(self.cfg.loader_shm_gb or os.environ.get('VITCAP_LOADER_SHM_GB', 4)) * 2**30)
                ahead = max(1, min(ahead, budget // (per_batch * slab_bytes) - 3))
                n_slabs = (ahead + 3) * per_batch
                logging.info('loader: %d slabs x %d MB = %.1f GB of page-locked shared memory (%d batches ahead)', n_slabs, slab_bytes >> 20,
                             n_slabs * slab_bytes / 2**30, ahead)
                # a /dev/shm too small for the slabs (container default: 64 MB) would kill the workers with SIGBUS on first touch:
                # check the free space first and fall back to returning the pixels through the pool's pipe (decode_many)
                try:
                    st = os.statvfs('/dev/shm')
                    room = st.f_bavail * st.f_frsize
                except OSError:
                    room = 0
                if room >= (n_slabs * slab_bytes * 5) // 4:
                    try:
                        for _ in range(n_slabs):
                            slabs.append(shared_memory.SharedMemory(create=True, size=slab_bytes))
                    except OSError as e:
                        logging.warning('shared-memory slabs unavailable (%s): decoded images return through the worker pipe', e)
                        for shm in slabs:
                            _release_slab(shm)
                        slabs = []
                else:
                    logging.warning('/dev/shm has %.0f MB free, the loader wants %.0f MB of slabs: decoded images return through the worker '
                                    'pipe (lower num_workers / loader_slab_mb, or enlarge /dev/shm)', room / 2**20, n_slabs * slab_bytes / 2**20)
                free = list(range(len(slabs)))
                # page-lock the slabs: a pageable source makes every host -> device copy go through the runtime's staging buffer (a
                # single-threaded memcpy of ~1 MB per image in THIS process, then the DMA); registered, the DMA reads the slab itself
                pinned = _pin_slabs(slabs)
            retired = []                                               # slab ids of the last batches handed to the GPU
            try:
                with pool:
                    inflight = []

                    def submit(k):
                        # a task names the file and the row numbers: the WORKER reads and base64-decodes its rows (jpegdec.py)
                        ids = mine[starts[k]:starts[k] + bs]
                        tasks = []
                        for c in range(0, len(ids), chunk):
                            if slabs:
                                sid = free.pop()
                                tasks.append((sid, pool.submit(decode_rows_into, slabs[sid].name, tsv, ids[c:c + chunk], device_jpeg)))
                            else:
                                tasks.append((None, pool.submit(decode_rows, tsv, ids[c:c + chunk])))
                        inflight.append(tasks)
                    nxt = 0
                    while nxt < len(starts) and len(inflight) < ahead:
                        submit(nxt)
                        nxt += 1
                    while inflight:
                        tasks = inflight.pop(0)
                        keys, imgs, used = [], [], []
                        t_wait = time.perf_counter()
                        for sid, f in tasks:
                            ks, items = f.result()
                            keys.extend(ks)
                            for item in items:
                                if sid is not None and isinstance(item, tuple) and item[0] == 'coef':
                                    imgs.append(CoefImage(*coef_item(item, slabs[sid].buf)))     # the GPU finishes the decode (csrc/jpeg.hip)
                                elif sid is not None and isinstance(item, tuple):
                                    off, h, w = item
                                    imgs.append(np.ndarray((h, w, 3), dtype=np.uint8, buffer=slabs[sid].buf, offset=off))
                                else:
                                    imgs.append(item)
                            if sid is not None:
                                used.append(sid)
                        t_pre = time.perf_counter()
                        batch = {'image': pre(imgs), 'key': keys}       # host -> device copies of pageable memory return once staged
                        del imgs
                        LOADER_TIMES['decode_wait'] = LOADER_TIMES.get('decode_wait', 0.0) + t_pre - t_wait
                        LOADER_TIMES['transform_enqueue'] = LOADER_TIMES.get('transform_enqueue', 0.0) + time.perf_counter() - t_pre
                        retired.append(used)
                        if len(retired) > 2:
                            free.extend(retired.pop(0))
                        if nxt < len(starts):
                            submit(nxt)
                            nxt += 1
                        yield batch
            finally:
                if slabs:
                    try:
                        torch.cuda.current_stream(torch.device('cuda', self.local_rank)).synchronize()     # no copy still reads a slab
                    except Exception:
                        pass
                    _unpin_slabs(pinned)
                for shm in slabs:
                    _release_slab(shm)
            return
        from . import weights as W
        n = int(self.cfg.synthetic_num_images or 8)
        bs = int(self.cfg.test_batch_size)
        seed = D.shard_seed(int(self.cfg.synthetic_seed or 1234), self.rank)
        per_rank = (n + self.world - 1) // self.world
        done = 0
        while done < per_rank:
            b = min(bs, per_rank - done)
            img = torch.from_numpy(W.gen_image_batch(done + b, seed)[done:])
            yield {'image': img, 'key': ['%d_%d' % (self.rank, done + i) for i in range(b)]}
            done += b

    def test_image_tsv(self):
        """Image rows of the test set: `test_image_tsv: <file>` or the reference layout data/<test_data>/<split>.tsv
        (TSVDataset, tsv_io.py:1175-1230); None for `test_data: synthetic`."""
        if self.cfg.test_image_tsv:
            return self.cfg.test_image_tsv
        data = self.cfg.test_data
        if not data or data == 'synthetic':
            return None
        f = op.join(self.cfg.data_root or 'data', data, '{}.tsv'.format(self.cfg.test_split or 'test'))
        if not op.isfile(f):
            raise FileNotFoundError('test images not found: {} (rows of key <tab> base64 JPEG)'.format(f))
        return f

    def predict_output_to_tsv_row(self, data, output):
        all_caps, all_confs = output[0], torch.exp(output[1])
        for key, caps, confs in zip(data['key'], all_caps, all_confs):
            res = [{'caption': self.tokenizer.decode(c.tolist(), skip_special_tokens=True), 'conf': float(p)}
                   for c, p in zip(caps, confs)]
            yield key, json.dumps(res)

    def predict(self, model_file, predict_result_file):
        model = self.load_test_model(self.get_raw_model(is_train=False), model_file)
        dev = torch.device('cuda', self.local_rank)
        model.pack(dev)
        sub = predict_result_file if self.world == 1 else '{}_{}_{}.tsv'.format(predict_result_file, self.rank, self.world)
        from .tsv import TSVFile, tsv_writer

        te = model.test_extra_input
        overlap = not te.get('do_sample', False)
        # the same option validation ImageCaptioning.forward applies (max_length, repetition_penalty, num_keep_best, use_cbs,
        # token ids ...): an option this build does not implement raises here instead of being decoded with defaults
        base = model.gen_options(gemm_mode=1) if overlap else None
        popts = {}                                # vitcap_gen_opts per number of visible tag slots: the caller's mask decides

        def opts_for(n_tag):
            if n_tag not in popts:
                if base.tag_visible and n_tag != base.tag_visible:
                    raise ValueError('test_extra_input tag_visible=%d but the attention_mask shows %d tag slots' % (base.tag_visible, n_tag))
                popts[n_tag] = base if n_tag == base.tag_visible else model.gen_options(gemm_mode=1, tag_visible=n_tag)
            return popts[n_tag]

        seen = {'n': None}                        # visible tag slots of the last batch whose text tensors live on the device

        def collect(entry):
            b, out, flag, expect_n = entry
            t_c0 = time.perf_counter()
            out = out.result() if overlap else out          # synchronises with the batch's decode stream
            LOADER_TIMES['consumer_wait_gpu'] = LOADER_TIMES.get('consumer_wait_gpu', 0.0) + time.perf_counter() - t_c0
            if flag is not None and not bool(flag):
                # the device-side comparison against the count of the earlier batches failed: either this batch's (valid) mask shows
                # another number of visible tag slots -- then it is decoded again with the options of ITS count, as a host-resident
                # batch would have been from the start -- or the mask is not one the engine implements
                n_tag = model.check_text_inputs(b, base.max_length)         # raises NotImplementedError for an unsupported mask
                # compared with the count THIS batch was submitted against (one batch is always in flight: by now `seen` may already
                # hold the count an earlier re-decode found -- ADVICE r4)
                if n_tag == expect_n:
                    raise NotImplementedError('a batch\'s attention_mask / token_type_ids on the device do not describe the mask structure '
                                              'the HIP engine implements (ImageCaptioning.check_text_inputs)')
                logging.info('visible tag slots changed from %s to %d: batch decoded again with its own options', expect_n, n_tag)
                seen['n'] = n_tag
                out = model.generate_async(b['image'], opts=opts_for(n_tag)).result()
            t_c1 = time.perf_counter()
            rows = list(self.predict_output_to_tsv_row(b, (out[0].cpu(), out[1].cpu())))
            LOADER_TIMES['consumer_rows'] = LOADER_TIMES.get('consumer_rows', 0.0) + time.perf_counter() - t_c1
            return rows

        def gen_rows():
            pending = []                          # greedy / beam: batch i decodes while batch i+1 is encoded (generate_async)
            with torch.no_grad():
                # batches queued on the device: capped in IMAGES (24 batches of 64 = 1 536 images = 1.4 GB of bf16 crops; the same count of
                # 512-image batches would be 11 GB)
                depth = int(self.cfg.loader_prefetch or os.environ.get('VITCAP_LOADER_PREFETCH', 0)) or max(2, min(24, 1536 // max(1, int(self.cfg.test_batch_size))))
                for batch in _prefetched(self.iter_test_batches(), dev, depth=depth):
                    batch = dict(batch)
                    batch['image'] = batch['image'].to(dev, non_blocking=True).contiguous()
                    flag = None
                    expect_n = None
                    if overlap:
                        # host tensors (what the loader yields) are checked on the host; tensors already on the device cost ONE host
                        # synchronisation (the first batch, to read the number of visible tag slots), afterwards they are compared on
                        # their stream and the verdict is read when the batch's captions are collected -- nothing stalls the
                        # 2-slot pipeline
                        am = batch.get('attention_mask')
                        if am is not None and am.is_cuda and seen['n'] is not None:
                            expect_n = seen['n']
                            n_tag, flag = model.check_text_inputs(batch, base.max_length, expect_n_tag=expect_n)
                        else:
                            n_tag = model.check_text_inputs(batch, base.max_length)
                            if am is not None and am.is_cuda:
                                seen['n'] = n_tag
                        t_s0 = time.perf_counter()
                        out = model.generate_async(batch['image'], opts=opts_for(n_tag))
                        LOADER_TIMES['consumer_submit'] = LOADER_TIMES.get('consumer_submit', 0.0) + time.perf_counter() - t_s0
                    else:
                        out = model(batch)
                    pending.append((batch, out, flag, expect_n))
                    # TWO batches stay in flight behind the one just submitted: batch k-2 is collected (its decode finished before
                    # batch k-1's started) while k-1 decodes and k encodes, so the detokeniser / JSON / TSV work of this thread (~5 ms
                    # per 64 captions) never sits between two submissions -- with one batch in flight the GPU ran batch k's decode
                    # alone while this thread was still writing batch k-1's rows (round 5: 21 vs 16 ms per batch from disk).  The
                    # outputs of a batch are its own tensors; its workspace slot is re-used by batch k+1 behind a stream-side wait.
                    while len(pending) > (2 if overlap else 0):
                        for key, js in collect(pending.pop(0)):
                            yield key, js
                for entry in pending:
                    for key, js in collect(entry):
                        yield key, js
        LOADER_TIMES.clear()
        stats = {'rows': 0, 't_first': None, 'rows_first': 0}

        def timed_rows():
            # images/s of the steady state: from the moment the first two batches' captions have come back (model warm, loader
            # workers running) to the last row -- the figure the reference's trainer logs as images/sec (trainer.py:155-170)
            for row in gen_rows():
                stats['rows'] += 1
                if stats['t_first'] is None and stats['rows'] >= 2 * int(self.cfg.test_batch_size):
                    stats['t_first'], stats['rows_first'] = time.perf_counter(), stats['rows']
                yield row
        tsv_writer(timed_rows(), sub)               # .tsv + .lineidx + .lineidx.8b (tsv_io.py:959-998)
        if stats['t_first'] is not None and stats['rows'] > stats['rows_first']:
            dt = time.perf_counter() - stats['t_first']
            LAST_PREDICT_STATS.update(rows=stats['rows'], steady_rows=stats['rows'] - stats['rows_first'], steady_seconds=dt, t_end=time.perf_counter(),
                                      loader_seconds={k: round(v, 3) for k, v in LOADER_TIMES.items()},
                                      images_per_sec=(stats['rows'] - stats['rows_first']) / dt)
            logging.info('predict: %d rows, steady state %.1f images/sec on rank %d', stats['rows'], LAST_PREDICT_STATS['images_per_sec'], self.rank)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            if self.rank == 0:      # concatenate and de-duplicate by key (uni_pipeline.py:816-831)
                def merged():
                    seen = set()
                    for r in range(self.world):
                        for row in TSVFile('{}_{}_{}.tsv'.format(predict_result_file, r, self.world)):
                            if row[0] not in seen:
                                seen.add(row[0])
                                yield row
                tsv_writer(merged(), predict_result_file)
            dist.barrier()
        return predict_result_file

    def ensure_predict(self, model_file=None):
        check_model_config(self.cfg, training=False)
        if self.cfg.ignore_predict:
            logging.info('ignore to predict as instructed')
            return None
        self._ensure_initialized()
        if model_file is None:
            model_file = self.get_checkpoint_file()        # uni_pipeline.py:680-683; never `basemodel`
        predict_result_file = self.get_predict_file(model_file)
        if not op.isfile(model_file) and self.cfg.init_recipe_seed is None:
            logging.info('ignore to run predict since %s does not exist', model_file)
            return predict_result_file
        if op.isfile(predict_result_file) and not self.cfg.force_predict:
            logging.info('ignore to do prediction %s', predict_result_file)
            return predict_result_file
        return self.predict(model_file, predict_result_file)

    def get_evaluate_file(self, predict_file):
        """uni_pipeline.py:852-882 for a captioning run (evaluate_method 'map', no test_version): <predict_file minus .tsv>.report."""
        assert predict_file.endswith('.tsv')
        return op.splitext(predict_file)[0] + '.report'

    def ensure_evaluate(self, predict_file=None):
        """uni_pipeline.py:884-911: rank 0 only; skipped when prediction / evaluation is switched off; re-done when the report is
        older than the prediction or `force_evaluate`."""
        if self.rank != 0:
            logging.info('skip because the rank %d != 0', self.rank)
            return None
        if self.cfg.ignore_evaluate or self.cfg.ignore_predict:
            logging.info('ignore evaluate as instructed')
            return None
        if not predict_file:
            predict_file = self.get_predict_file(self.get_checkpoint_file())
        if not op.isfile(predict_file):
            logging.info('ignore evaluate: no prediction file %s', predict_file)
            return None
        evaluate_file = self.get_evaluate_file(predict_file)
        fresh = op.isfile(evaluate_file) and op.getmtime(evaluate_file) >= op.getmtime(predict_file)
        if fresh and not self.cfg.force_evaluate:
            logging.info('ignore %s', evaluate_file)
            return evaluate_file
        return evaluate_file if self.evaluate(predict_file, evaluate_file) is not None else None

    def evaluate(self, predict_file, evaluate_file):
        """..._bertemb.py:632-647 scores the predict TSV against data/<test_data>/<split>.caption.tsv with the external coco_caption
        package (pycocoevalcap: PTB tokenizer jar, BLEU / METEOR jar / ROUGE / CIDEr / SPICE jar), which neither the reference
        vendors (README:24) nor this image has.  Native restatements here of the two metrics that need no Java: corpus BLEU-1..4
        (Papineni et al., closest reference length) and CIDEr-D (vitcap_amd/scst.py, document frequencies from the references) on
        lower-cased whitespace tokens.  PARITY-UNPINNED: no coco_caption output exists to check them against; the report says so."""
        from .scst import CiderD, corpus_bleu
        from .tsv import TSVFile
        if not self.cfg.test_data or self.cfg.test_data == 'synthetic':
            logging.info('evaluate: synthetic test data has no reference captions; skipped')
            return None
        cap_file = op.join(self.cfg.data_root or 'data', self.cfg.test_data, '{}.caption.tsv'.format(self.cfg.test_split or 'test'))
        if not op.isfile(cap_file):
            logging.info('evaluate: no reference captions (%s); skipped', cap_file)
            return None
        gts = {}
        for row in TSVFile(cap_file):
            gts[row[0]] = [str(c['caption']).lower().strip() for c in json.loads(row[1])]
        keys, res = [], []
        for row in TSVFile(predict_file):
            if row[0] in gts and gts[row[0]]:
                keys.append(row[0])
                res.append(str(json.loads(row[1])[0]['caption']).lower().strip())
        if not keys:
            logging.info('evaluate: no predicted key has reference captions; skipped')
            return None
        refs = [gts[k] for k in keys]
        result = {'Bleu_%d' % (n + 1): b for n, b in enumerate(corpus_bleu(refs, res))}
        result['CIDEr'] = CiderD().compute_score(refs, res)[0]      # CiderD's own x10: the scale coco_caption reports under this key
        result['images'] = len(keys)
        result['note'] = ('native BLEU / CIDEr-D restatements on lower-cased whitespace tokens; parity-unpinned (the reference\'s scorer '
                          'is the external coco_caption package: PTB tokenizer, METEOR and SPICE are Java and not reproduced)')
        with open(evaluate_file, 'w') as fp:
            json.dump(result, fp)
        logging.info('evaluation result: %s', {k: v for k, v in result.items() if k != 'note'})
        logging.info('evaluation result saved to %s', evaluate_file)
        return result

    def monitor_train(self):
        """uni_pipeline.py:1021-1038: predict + evaluate every intermediate snapshot of the run and collect the scores per iteration
        (the reference plots them to tensorboard, which this image lacks: they go to <output>/monitor_train.json).  Called after
        training here (run.py has no watcher process), so no snapshot is still being written."""
        import glob
        import re
        self._ensure_initialized()
        steps = sorted(int(re.search(r'model_iter_(\d+)\.pt$', f).group(1))
                       for f in glob.glob(op.join(self.get_snapshot_dir(), 'model_iter_*.pt')))
        iter_to_eval = {}
        for it in steps:
            pf = self.ensure_predict(self.get_checkpoint_file(iteration=it))
            ef = self.ensure_evaluate(pf) if pf else None
            if ef and op.isfile(ef):
                with open(ef) as fp:
                    iter_to_eval[it] = {k: v for k, v in json.load(fp).items() if k != 'note'}
        if self.rank == 0 and iter_to_eval:
            with open(op.join(self.output_folder, 'monitor_train.json'), 'w') as fp:
                json.dump(iter_to_eval, fp, indent=1)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
        return iter_to_eval
