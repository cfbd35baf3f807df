"""python run.py -c <yaml> [-p 'yaml overrides'] [-bp base64]   -- same surface as the reference's run.py:
the YAML's top-level `type` names one of the entry points below, called with the remaining keys; `param.pipeline_type`
(`{from: module, import: class}`) selects the pipeline class.  With the reference YAML unchanged except
`pipeline_type.from: vitcap_amd.pipeline`, `pipeline_eval_multi` captions on the MI355X path.

The four names the YAMLs and the reference's callers bind to (`create_pipeline`, `load_pipeline`, `pipeline_eval_multi`,
`pipeline_train_eval_multi`; reference run.py:16-71) are kept with their signatures; they are thin views of one builder."""
import copy
import glob
import logging
import os.path as op
from pprint import pformat

from vitcap_amd.config import (dict_ensure_path_key_converted, dict_update_nested_dict, execute_func,
                               load_from_yaml_file, parse_general_args)


def _saved_parameters(full_expid):
    """The newest output/<expid>/parameters_<timestamp>.yaml a training run left behind (src/tools/qd_pytorch.py:52-73);
    an empty dict when the experiment was never trained here."""
    found = sorted(glob.glob(op.join('output', full_expid or '', 'parameters_*.yaml')))
    return load_from_yaml_file(found[-1]) if found else {}


def _build(param, test_info=None, from_saved=False):
    """One pipeline object from a parameter dict.

    test_info   one entry of `all_test_data` (test_data / test_split / ...), layered over `param`;
    from_saved  start from the parameters the experiment was trained with and layer the above over those, so that evaluation
                sees the training-time settings it does not override."""
    merged = copy.deepcopy(param)
    if test_info is not None:
        dict_ensure_path_key_converted(test_info)
        dict_update_nested_dict(merged, test_info)
    if from_saved:
        saved = _saved_parameters(merged.get('full_expid'))
        dict_update_nested_dict(saved, merged)
        merged = saved
    spec = copy.deepcopy(merged['pipeline_type'])
    if 'param' in spec:
        raise ValueError('pipeline_type must not carry its own `param`')
    spec['param'] = merged
    return execute_func(spec), merged


def create_pipeline(kwargs):
    return _build(kwargs)[0]


def load_pipeline(**kwargs):
    return _build(kwargs, from_saved=True)[0]


def pipeline_eval_multi(param, all_test_data, **kwargs):
    for info in all_test_data:
        pipeline, used = _build(param, test_info=info, from_saved=True)
        trained = pipeline.is_train_finished() or used.get('init_recipe_seed') is not None
        if not trained:
            logging.info('the model specified by the following is not ready\n%s', pformat(param))
            return
        pipeline.ensure_predict()
        pipeline.ensure_evaluate()


def pipeline_train_eval_multi(all_test_data, param, **kwargs):
    trainer, _ = _build(param, test_info=copy.deepcopy(all_test_data[0]) if all_test_data else None)
    trainer.ensure_train()
    param['full_expid'] = trainer.full_expid
    pipeline_eval_multi(param, all_test_data)
    return trainer.full_expid


ENTRY_POINTS = {f.__name__: f for f in (create_pipeline, load_pipeline, pipeline_eval_multi, pipeline_train_eval_multi)}

if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    kwargs = parse_general_args()
    logging.info('param:\n%s', pformat(kwargs))
    ENTRY_POINTS[kwargs.pop('type')](**kwargs)
