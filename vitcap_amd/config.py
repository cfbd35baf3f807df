"""Config / plugin semantics of the reference's run.py surface, restated.

Reference behaviour reproduced (file:line in the reference tree):
* ``-c yaml`` with ``_base_`` inheritance (src/tools/common.py:227-240), ``-p`` inline YAML and ``-bp`` base64 YAML
  overrides applied on top with the same precedence (common.py:282-320);
* ``a$b$c`` path keys expanded into nested dicts (common.py dict_ensure_path_key_converted / dict_update_path_value);
* ``execute_func({'from': module, 'import': name, 'param': {...}})`` plugin loading (common.py:133-139);
* ``Config(default, overwrite)``: attribute access falls back overwrite -> default -> None, never raises
  (src/pipelines/uni_pipeline.py:63-84).
"""
import argparse
import base64
import copy
import os.path as op
from importlib import import_module

import yaml


def load_from_yaml_str(s):
    return yaml.safe_load(s)


def dict_update_path_value(d, path, v, sep='$'):
    ps = path.split(sep)
    cur = d
    for p in ps[:-1]:
        if p not in cur or not isinstance(cur[p], dict):
            cur[p] = {}
        cur = cur[p]
    cur[ps[-1]] = v


def dict_ensure_path_key_converted(d, sep='$'):
    """{'a$b': 1} -> {'a': {'b': 1}} in place, recursively."""
    for k in list(d.keys()):
        v = d[k]
        if isinstance(v, dict):
            dict_ensure_path_key_converted(v, sep)
        if isinstance(k, str) and sep in k:
            del d[k]
            dict_update_path_value(d, k, v, sep)


def dict_update_nested_dict(a, b):
    """a <- b, recursing into dicts (leaf values of b win)."""
    for k, v in b.items():
        if isinstance(v, dict) and isinstance(a.get(k), dict):
            dict_update_nested_dict(a[k], v)
        else:
            a[k] = copy.deepcopy(v)


def load_from_yaml_file(file_name):
    with open(file_name, 'r') as fp:
        data = load_from_yaml_str(fp.read())
    while isinstance(data, dict) and '_base_' in data:
        base = load_from_yaml_file(op.join(op.dirname(file_name), data['_base_']))
        assert isinstance(base, dict)
        del data['_base_']
        dict_update_nested_dict(base, data)
        data = base
    return data


def parse_general_args(argv=None):
    parser = argparse.ArgumentParser(description='General Parser')
    parser.add_argument('-c', '--config_file', type=str, help='config file')
    parser.add_argument('-p', '--param', type=str, help='parameter string, yaml format')
    parser.add_argument('-bp', '--base64_param', type=str, help='base64 encoded yaml format')
    args = parser.parse_args(argv)
    kwargs = {}
    if args.config_file:
        kwargs.update(load_from_yaml_file(args.config_file))
    if args.base64_param:
        kwargs.update(load_from_yaml_str(base64.b64decode(args.base64_param)))
    if args.param:
        cfg = load_from_yaml_str(args.param)
        dict_ensure_path_key_converted(cfg)
        for k, v in cfg.items():
            if isinstance(v, dict) and isinstance(kwargs.get(k), dict):
                dict_update_nested_dict(kwargs[k], v)
            else:
                kwargs[k] = v
    return kwargs


def execute_func(info):
    module = import_module(info['from'])
    fn = getattr(module, info['import'])
    return fn(**info['param']) if 'param' in info else fn()


class Config(object):
    def __init__(self, default, overwrite):
        self.default = default
        self.overwrite = overwrite

    def get(self, k):
        if k in self.overwrite:
            return self.overwrite[k]
        return self.default.get(k)

    def __getattr__(self, k):
        if k in ('default', 'overwrite'):
            raise AttributeError(k)
        return self.get(k)

    def get_dict(self):
        d = copy.deepcopy(self.default)
        d.update(copy.deepcopy(self.overwrite))
        return d

    def update(self, cfg):
        self.overwrite.update(cfg)
