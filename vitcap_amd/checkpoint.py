"""Checkpoint load/save with the reference's contract (src/tools/opt/checkpoint.py:18-273,
src/tools/qd_pytorch.py:17-49): files are ``torch.save({'model': sd, 'optimizer':…, 'scheduler':…, 'iteration': n})``;
on load, ``module.`` prefixes are stripped, each model key takes the loaded key that is its LONGEST SUFFIX, shape
mismatches are skipped (non-strict), and ``last_checkpoint`` in the save dir names the latest snapshot."""
import logging
import os

import torch


def remove_prefix(sd, prefix):
    if not all(k.startswith(prefix) for k in sd):
        return sd
    return type(sd)((k[len(prefix):], v) for k, v in sd.items())


def align_state_dicts(model_keys, loaded_sd):
    """{model key: loaded key} by longest-suffix matching."""
    loaded_keys = sorted(loaded_sd.keys())
    out = {}
    for mk in sorted(model_keys):
        best = None
        for lk in loaded_keys:
            if mk.endswith(lk) and (best is None or len(lk) > len(best)):
                best = lk
        if best is not None:
            out[mk] = best
    return out


def load_state_dict_tolerant(model, loaded_sd):
    """Returns (loaded, shape_mismatch, unmatched_model_keys)."""
    loaded_sd = remove_prefix(loaded_sd, 'module.')
    msd = model.state_dict()
    match = align_state_dicts(msd.keys(), loaded_sd)
    good, bad = {}, []
    for mk, lk in match.items():
        if tuple(msd[mk].shape) == tuple(loaded_sd[lk].shape):
            good[mk] = loaded_sd[lk]
        else:
            logging.info('%s shape is not consistent, expected: %s; got %s', mk, tuple(msd[mk].shape),
                         tuple(loaded_sd[lk].shape))
            bad.append(mk)
    res = model.load_state_dict(good, strict=False)
    return sorted(good), bad, list(res.missing_keys)


def torch_load(path):
    # whole pickled dicts (optimizer/scheduler payloads): explicit policy, torch >= 2.6 defaults to weights_only=True
    return torch.load(path, map_location='cpu', weights_only=False)


class Checkpointer(object):
    def __init__(self, model, optimizer=None, scheduler=None, save_dir='', save_to_disk=None, suffix='pt'):
        self.model = model
        self.optimizer = optimizer
        self.scheduler = scheduler
        self.save_dir = save_dir
        self.save_to_disk = save_to_disk
        self.suffix = suffix

    def save(self, name, **kwargs):
        if not self.save_dir or not self.save_to_disk:
            return None
        data = {'model': self.model.state_dict()}
        if self.optimizer is not None:
            data['optimizer'] = self.optimizer.state_dict()
        if self.scheduler is not None:
            data['scheduler'] = self.scheduler.state_dict()
        data.update(kwargs)
        os.makedirs(self.save_dir, exist_ok=True)
        f = os.path.join(self.save_dir, '{}.{}'.format(name, self.suffix))
        torch.save(data, f)
        self.tag_last_checkpoint(f)
        return f

    def has_checkpoint(self):
        return os.path.exists(os.path.join(self.save_dir, 'last_checkpoint'))

    def get_checkpoint_file(self):
        try:
            with open(os.path.join(self.save_dir, 'last_checkpoint')) as fp:
                return fp.read().strip()
        except IOError:
            return ''

    def tag_last_checkpoint(self, last_filename):
        with open(os.path.join(self.save_dir, 'last_checkpoint'), 'w') as fp:
            fp.write(last_filename)

    def load(self, f=None, model_only=False, load_if_has=True):
        if self.has_checkpoint() and load_if_has:
            f = self.get_checkpoint_file()
            model_only = False
        if not f:
            logging.info('No checkpoint found. Initializing model from scratch')
            return {}
        ck = torch_load(f)
        if 'model' not in ck:
            ck = {'model': ck}
        load_state_dict_tolerant(self.model, ck.pop('model'))
        if 'optimizer' in ck:
            opt = ck.pop('optimizer')
            if self.optimizer and not model_only:
                self.optimizer.load_state_dict(opt)
        if 'scheduler' in ck:
            sch = ck.pop('scheduler')
            if self.scheduler and not model_only:
                self.scheduler.load_state_dict(sch)
        return {} if model_only else ck

    recover_or_load = load
