"""Small host-side helpers with the reference's names (reference: libs/utils.py): seeding, parameter
counting, timing, pickling.  Nothing here is on the device hot path."""
from __future__ import annotations

import os
import pickle
import random
import time
from contextlib import contextmanager

import numpy as np
import torch


def get_seed(s, printout=True, cudnn=True):
    """Seed python, numpy and torch (all devices); make MIOpen deterministic where it can be."""
    os.environ['PYTHONHASHSEED'] = str(s)
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(s)
        torch.cuda.manual_seed_all(s)
        if cudnn:
            torch.backends.cudnn.deterministic = True
            torch.backends.cudnn.benchmark = False
        try:
            from . import _hip
            _hip.set_seed(int(s))
        except Exception:
            pass
    if printout:
        print(f"The following code snippets have been run.\n{'=' * 50}\nseed = {s}\n{'=' * 50}")


def get_num_params(model):
    """Number of trainable scalars; complex parameters count twice (real + imaginary)."""
    total = 0
    for p in model.parameters():
        if p.requires_grad:
            total += p.numel() * (2 if p.is_complex() else 1)
    return total


class Colors:
    red, green, yellow, blue, magenta, cyan, white, end = ("\033[91m", "\033[92m", "\033[93m", "\033[94m",
                                                           "\033[95m", "\033[96m", "\033[37m", "\033[0m")


def color(string: str, color=Colors.yellow) -> str:
    return f"{color}{string}{Colors.end}"


@contextmanager
def timer(label: str = "", compact=False):
    """Wall-clock timer context: ``with timer("forward"): ...``"""
    t0 = time.perf_counter()
    try:
        yield
    finally:
        dt = time.perf_counter() - t0
        msg = f"{label}: {dt:.4f} s" if compact else f"{label} done in {dt:.4f} seconds."
        print(color(msg, Colors.blue))


def save_pickle(obj, path):
    with open(path, 'wb') as f:
        pickle.dump(obj, f, protocol=pickle.HIGHEST_PROTOCOL)


def load_pickle(path):
    with open(path, 'rb') as f:
        return pickle.load(f)


def is_interactive() -> bool:
    import __main__ as main
    return not hasattr(main, '__file__')


def get_date():
    return time.strftime("%Y-%m-%d", time.localtime())
