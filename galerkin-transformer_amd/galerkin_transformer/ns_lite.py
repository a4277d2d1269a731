"""(2+1)-D Navier-Stokes example (ex4): dataset, model alias and the 10-step autoregressive train / validate
loops, with the names ``examples/ex4_navier_stokes_2+1d.py`` imports (reference: libs/ns_lite.py).

``FourierTransformer2DLite`` is the class in ``model.py`` (the reference keeps an identical copy in ns_lite.py:109-202
and model.py:1186-1283).  The dataset reads the reference's ``ns_V1000_N5000_T50.mat`` (MATLAB v7.3 = HDF5) when
``h5py`` and the file are present and otherwise falls back to a deterministic synthetic vorticity field of the same
shapes, so the example, the benchmark workload and the tests run without the download.
"""
from __future__ import annotations

import gc
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .ft import *            # noqa: F401,F403
from .layers import *        # noqa: F401,F403
from .model import *         # noqa: F401,F403
from .model import FourierTransformer2DLite  # noqa: F401
from .utils import *         # noqa: F401,F403
from .utils import get_seed, timer
from .utils_ft import *      # noqa: F401,F403
from .utils_ft import _finish_step

__all__ = [n for n in dir() if not n.startswith("_")] + ["NavierStokesDatasetLite", "train_batch_ns",
                                                          "validate_epoch_ns"]


class NavierStokesDatasetLite(Dataset):
    """Vorticity trajectories on a 64 x 64 periodic grid (Li et al. 2020), ns_lite.py:14-107.

    item: node (n, n, T_in) first T_in time slices, pos (n*n, 2), grid (n, n, 2), target (n, n, T_out) the next
    T_out slices, target_grad (n, n, 2, T_out) their zero-padded central differences."""

    def __init__(self, data_path=None, train_data=True, train_len=1024, valid_len=200, time_steps_input=10,
                 time_steps_output=10, return_boundary=True, random_state=1127802, synthetic_len=None):
        self.data_path = data_path
        self.n_grid = 64
        self.h = 1 / self.n_grid
        self.train_data = train_data
        self.time_steps_input, self.time_steps_output = time_steps_input, time_steps_output
        self.train_len, self.valid_len = train_len, valid_len
        self.return_boundary = return_boundary
        self.random_state = random_state
        self.synthetic_len = synthetic_len
        self.eps = 1e-8
        if self.data_path is not None or synthetic_len is not None:
            self._initialize()

    def __len__(self):
        return self.n_samples

    # ---- data -----------------------------------------------------------------------------------------
    def _load(self):
        path = self.data_path
        if path is not None and self.synthetic_len is None:
            # a data file was asked for: fail loudly like the reference (ns_lite.py:60-75) instead of training on
            # synthetic fields under the file's name
            if not os.path.exists(path):
                raise FileNotFoundError(f"NavierStokesDatasetLite: {path} does not exist (pass synthetic_len=... for "
                                        "the synthetic advection-diffusion set)")
            import h5py                                                   # ImportError if the reader is missing
            with timer(f"Loading {os.path.basename(path)}"):
                with h5py.File(path, mode='r') as data:
                    return np.transpose(data['u'])                        # (N, n, n, T)
        n = self.synthetic_len if self.synthetic_len is not None else (self.train_len + self.valid_len)
        return self.synthetic_trajectories(n, self.n_grid, self.time_steps_input + self.time_steps_output,
                                           self.random_state)

    @staticmethod
    def synthetic_trajectories(n_samples, n, steps, seed):
        """Smooth periodic fields advected by a per-sample constant velocity with mild diffusion: the exact solution
        of a linear advection-diffusion problem in Fourier space (cheap, deterministic, learnable)."""
        rng = np.random.RandomState(seed)
        k = np.fft.fftfreq(n, 1.0 / n)
        kx, ky = np.meshgrid(k, k, indexing="ij")
        k2 = kx ** 2 + ky ** 2
        amp = 1.0 / (1.0 + k2) ** 1.25
        amp[0, 0] = 0.0
        out = np.empty((n_samples, n, n, steps), dtype=np.float32)
        for i in range(n_samples):
            w0 = (rng.randn(n, n) + 1j * rng.randn(n, n)) * amp
            c = rng.uniform(-1.0, 1.0, size=2)
            for t in range(steps):
                tau = 0.05 * t
                phase = np.exp(-2j * np.pi * (kx * c[0] + ky * c[1]) * tau - 1e-3 * 4 * np.pi ** 2 * k2 * tau)
                out[i, ..., t] = np.real(np.fft.ifft2(w0 * phase))
        return out / out.std()

    def _initialize(self):
        get_seed(self.random_state, printout=False)
        x = self._load()
        a = x[..., :self.time_steps_input]
        u = x[..., self.time_steps_input:self.time_steps_input + self.time_steps_output]
        del x
        gc.collect()
        if self.train_data:
            a, u = a[:self.train_len], u[:self.train_len]
        else:
            a, u = a[-self.valid_len:], u[-self.valid_len:]
        self.n_samples = len(a)
        self.nodes, self.target, self.target_grad = self.get_data(a, u)
        ax = np.linspace(0, 1, self.n_grid)
        gx, gy = np.meshgrid(ax, ax)
        self.grid = np.stack([gx, gy], axis=-1)
        self.pos = np.c_[gx.ravel(), gy.ravel()]

    def get_data(self, nodes, targets):
        gx, gy = self.central_diff(targets, self.h)
        return nodes, targets, np.stack([gx, gy], axis=-2)

    @staticmethod
    def central_diff(x, h, padding=True):
        """x (N, n, n, t) -> d/dx, d/dy by dilation-2 central differences on the zero-padded field."""
        if padding:
            x = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)), 'constant', constant_values=0)
        d, s = 2, 1
        gx = (x[:, d:, s:-s] - x[:, :-d, s:-s]) / d
        gy = (x[:, s:-s, d:] - x[:, s:-s, :-d]) / d
        return gx / h, gy / h

    def __getitem__(self, idx):
        f = lambda v: torch.from_numpy(np.ascontiguousarray(v)).float()
        return dict(node=f(self.nodes[idx]), pos=f(self.pos), grid=f(self.grid), target=f(self.target[idx]),
                    target_grad=f(self.target_grad[idx]))


# ---- autoregressive rollout (ns_lite.py:205-264) ---------------------------------------------------------
def rollout_loss(model, loss_func, x, pos, grid, u, gradu):
    """Sum over the T_out steps of loss + regulariser, feeding every prediction back as the newest input slice.
    Returns (loss_total tensor, reg_total float, list of per-step predictions)."""
    steps = x.size(-1)
    loss_total, reg_total, preds = 0, 0.0, []
    for t in range(steps):
        u_pred = model(x, None, pos=pos, grid=grid)['preds']                 # (b, n, n, 1)
        loss, reg, _, _ = loss_func(u_pred[..., 0], u[..., t], targets_prime=gradu[..., t])
        loss_total = loss_total + (loss + reg)
        reg_total += reg.item()
        x = torch.cat((x[..., 1:], u_pred), dim=-1)
        preds.append(u_pred)
    return loss_total, reg_total, preds


def train_batch_ns(model, loss_func, data, optimizer, lr_scheduler, device, grad_clip=0.99):
    optimizer.zero_grad()
    x = data["node"].to(device)
    pos, grid = data['pos'].to(device), data['grid'].to(device)
    u, gradu = data["target"].to(device), data["target_grad"].to(device)
    steps = x.size(-1)
    loss_total, reg_total, preds = rollout_loss(model, loss_func, x, pos, grid, u, gradu)
    _finish_step(model, loss_total, optimizer, lr_scheduler, grad_clip)      # ONE backward through all steps
    u_preds = torch.cat(preds, dim=-1).detach()
    return (loss_total.item() / steps, reg_total / steps), u_preds, None


def validate_epoch_ns(model, metric_func, valid_loader, device):
    model.eval()
    metric_val = []
    for data in valid_loader:
        with torch.no_grad():
            x, u = data["node"].to(device), data["target"].to(device)
            pos, grid = data['pos'].to(device), data['grid'].to(device)
            steps = x.size(-1)
            acc = 0
            for t in range(steps):
                u_pred = model(x, None, pos=pos, grid=grid)['preds']
                _, _, metric, _ = metric_func(u_pred[..., 0], u[..., t])
                x = torch.cat((x[..., 1:], u_pred), dim=-1)
                acc += metric
        metric_val.append(acc / steps)
    return dict(metric=np.mean(metric_val, axis=0))
