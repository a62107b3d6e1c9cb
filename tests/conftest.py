import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    meta = json.loads(str(d.pop('meta'))) if 'meta' in d else {}
    return d, meta


def sub_sd(d, prefix):
    import torch
    # 'enc.<T>' / 'x_dec.<T>' style keys are data vectors, not parameters
    return {k[len(prefix):]: torch.from_numpy(v) for k, v in d.items()
            if k.startswith(prefix) and not k[len(prefix):].isdigit()}


@pytest.fixture(scope='session')
def golden():
    return load_golden
