"""Shared helpers for the test-suite (golden loading, tolerances)."""
import os

import numpy as np
import torch

import lxmert_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_cfg(g):
    kw = {k[4:]: g[k].item() for k in g if k.startswith("cfg_")}
    return O.OracleConfig(**kw)


def golden_inputs(g):
    return {k[3:]: torch.from_numpy(g[k]) for k in g if k.startswith("in_")}


def maxdiff(a, b):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    return (a - b).abs().max().item()


def slice_idx(n, count=192):
    """indices of the gradient samples stored in full_955.npz (`gslice:*`; same rule as oracle/gen_golden.py::slice_idx)."""
    if n <= count:
        return np.arange(n)
    return np.unique(np.concatenate([np.arange(64), np.linspace(64, n - 1, count - 64).astype(np.int64)]))
