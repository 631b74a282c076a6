"""Shared helpers for the parity tests."""
import os
from collections import OrderedDict

import numpy as np
import torch

from lime_cikm25_amd import synth
import golden_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def load_golden(name):
    with np.load(os.path.join(GOLDEN_DIR, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def synth_state_dict(keys_and_shapes, seed=golden_cases.WEIGHT_SEED):
    sd = OrderedDict()
    for k, shape in keys_and_shapes:
        sd[k] = synth.synth_tensor(k, shape, seed)
    return sd


def rel_err(a, b, floor=None):
    """max |a-b| / max(|b|, floor).  ``floor`` defaults to the mean magnitude of the non-zero reference
    entries: an element at or above the tensor's typical scale is judged by its TRUE relative error, an
    element below it (and the exact zeros the saturated lifetime weight produces, SURVEY.md section 7
    'Exact zeros and ties') by its absolute error against that floor."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    if floor is None:
        nz = np.abs(b[b != 0])
        floor = float(nz.mean()) if nz.size else 1.0
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), floor)))
