import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.hookimpl(trylast=True)
def pytest_collection_modifyitems(config, items):
    """The config-2 grow test waits for ~150 s of CPU oracle work that depends on nothing else: run it last, and start its oracle
    trajectory in a background process when the session begins (tests/_config2_oracle.py), beside the other GPU tests."""
    late = [it for it in items if it.name.startswith('test_config2_grow_run_against_oracle')]
    if late:
        items[:] = [it for it in items if it not in late] + late


def pytest_collection_finish(session):
    if getattr(session.config.option, 'collectonly', False) or not torch.cuda.is_available():
        return
    if len(session.items) > 1 and any(it.name.startswith('test_config2_grow_run_against_oracle') for it in session.items):
        import tempfile
        import _config2_oracle
        session.config._config2_tmp = tempfile.TemporaryDirectory()
        _config2_oracle.start_background(session.config._config2_tmp.name)


def pytest_unconfigure(config):
    mod = sys.modules.get('_config2_oracle')
    if mod is not None and mod._BACKGROUND:
        proc = mod._BACKGROUND['proc']
        if proc.poll() is None:              # (session ended early: -x, Ctrl-C)
            proc.kill()
            proc.wait()
    tmp = getattr(config, '_config2_tmp', None)
    if tmp is not None:
        tmp.cleanup()


def load_fixture(name):
    """Returns (meta dict, npz mapping) of a golden fixture written by tests/golden/make_golden.py."""
    with open(os.path.join(GOLDEN, name + '.json')) as f:
        meta = json.load(f)
    data = np.load(os.path.join(GOLDEN, name + '.npz'))
    return meta, data


def fixture_params(data, prefix):
    """Oracle-style parameter dict from a fixture: tensors by state_dict name + '<layer>.c' floats."""
    out = {}
    pre = prefix + '/'
    for k in data.files:
        if k.startswith(pre):
            name = k[len(pre):]
            v = data[k]
            out[name] = float(v) if name.endswith('.c') else torch.from_numpy(v.copy())
    return out


def _cpu64(t):
    if torch.is_tensor(t):
        return t.detach().cpu().to(torch.float64).reshape(-1)
    return torch.as_tensor(np.asarray(t), dtype=torch.float64).reshape(-1)


def rel_err(a, b):
    """max|a-b| / max|b|  (tensors on any device, numpy arrays or scalars)."""
    a, b = _cpu64(a), _cpu64(b)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope='session')
def oracle():
    from oracle import pggan_cpu
    return pggan_cpu
