import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'oracle')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the C-ABI library is a build product (git-ignored): build it once if a fresh checkout has none and hipcc is here
    so = os.path.join(ROOT, 'diff-vit_amd', 'csrc', 'libp2vit_hip.so')
    if not os.path.exists(so) and os.path.exists('/opt/rocm/bin/hipcc'):
        import subprocess
        subprocess.run(['make', '-C', os.path.dirname(so)], check=True, stdout=subprocess.DEVNULL)


@pytest.fixture(scope='session')
def oracle():
    import p2vit_oracle
    return p2vit_oracle


@pytest.fixture(scope='session')
def synth():
    import diff_vit_amd
    return diff_vit_amd.synth


def load_golden(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def golden_calib(g, oracle_mod):
    flat = {k[len('calib/'):]: torch.from_numpy(g[k]) for k in g.files if k.startswith('calib/')}
    return oracle_mod.unflatten_calib(flat)


def golden_weights(g):
    return {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith('w/')}


@pytest.fixture(scope='session')
def micro(oracle, synth):
    g = load_golden('micro_vit')
    return dict(g=g, arch=synth.ARCHS['micro'], sd=golden_weights(g), calib=golden_calib(g, oracle),
                x_cal=torch.from_numpy(g['x_cal']), x_ev=torch.from_numpy(g['x_ev']))


def planted_state_dict(g, synth_mod, arch):
    """weights of a fixture generated with oracle/gen_golden.py::plant_head_margin: the seeded synthetic model plus the stored
    head rows (one planted class per evaluation image)."""
    sd = synth_mod.vit_state_dict(arch, int(g['seed']))
    hw = sd['head.weight'].clone()
    hw[torch.from_numpy(g['head_classes'])] += torch.from_numpy(g['head_delta'])
    sd['head.weight'] = hw
    return sd


def gpu_ok():
    return torch.cuda.is_available()
