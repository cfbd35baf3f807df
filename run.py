"""python run.py -c <yaml> [-p 'yaml overrides'] [-bp base64]   -- same surface as the reference's run.py:
the YAML's top-level `type` names one of the functions below, called with the remaining keys; `param.pipeline_type`
(`{from: module, import: class}`) selects the pipeline class.  With the reference YAML unchanged except
`pipeline_type.from: vitcap_amd.pipeline`, `pipeline_eval_multi` captions on the MI355X path."""
import copy
import logging
import os.path as op
from pprint import pformat

from vitcap_amd.config import (dict_ensure_path_key_converted, dict_update_nested_dict, execute_func,
                               load_from_yaml_file, parse_general_args)


def create_pipeline(kwargs):
    info = copy.deepcopy(kwargs.get('pipeline_type'))
    assert 'param' not in info
    info['param'] = kwargs
    return execute_func(info)


def load_latest_parameters(folder):
    """output/<expid>/parameters_<timestamp>.yaml written at train time (src/tools/qd_pytorch.py:52-73)."""
    import glob
    files = sorted(glob.glob(op.join(folder, 'parameters_*.yaml')))
    return load_from_yaml_file(files[-1]) if files else {}


def load_pipeline(**kwargs):
    kwargs = copy.deepcopy(kwargs)
    kwargs_f = load_latest_parameters(op.join('output', kwargs.get('full_expid', '')))
    dict_update_nested_dict(kwargs_f, kwargs)
    return create_pipeline(kwargs_f)


def pipeline_eval_multi(param, all_test_data, **kwargs):
    for test_data_info in all_test_data:
        curr_param = copy.deepcopy(param)
        dict_ensure_path_key_converted(test_data_info)
        dict_update_nested_dict(curr_param, test_data_info)
        pip = load_pipeline(**curr_param)
        if not pip.is_train_finished() and curr_param.get('init_recipe_seed') is None:
            logging.info('the model specified by the following is not ready\n%s', pformat(param))
            return
        pip.ensure_predict()
        pip.ensure_evaluate()


def pipeline_train_eval_multi(all_test_data, param, **kwargs):
    curr_param = copy.deepcopy(param)
    if len(all_test_data) > 0:
        dict_update_nested_dict(curr_param, all_test_data[0])
    pip = create_pipeline(curr_param)
    pip.ensure_train()
    full_expid = pip.full_expid
    param['full_expid'] = full_expid
    pipeline_eval_multi(param, all_test_data)
    return full_expid


if __name__ == '__main__':
    logging.basicConfig(level=logging.INFO)
    kwargs = parse_general_args()
    logging.info('param:\n%s', pformat(kwargs))
    function_name = kwargs.pop('type')
    locals()[function_name](**kwargs)
