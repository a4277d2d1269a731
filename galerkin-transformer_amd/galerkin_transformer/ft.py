"""Data containers, normaliser and losses with the reference's names (reference: libs/ft.py).

These are host-side / few-kernels-per-step pieces outside the HIP hot path:
they are plain PyTorch.  The ``.mat`` datasets of the reference are not redistributable; the
dataset classes load them when ``DATA_PATH`` provides the files and otherwise synthesise tensors
of the same shapes (``synthetic=True``), which is what the benchmarks use.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import numpy as np
import torch
from torch.nn.modules.loss import _WeightedLoss
from torch.utils.data import Dataset


class UnitGaussianNormalizer:
    """Pointwise (x - mean) / (std + eps) fitted over the sample axis (ft.py:373-415)."""

    def __init__(self, eps=1e-5):
        self.eps = eps
        self.mean = None
        self.std = None

    def fit_transform(self, x):
        self.mean = x.mean(0)
        self.std = x.std(0)
        return self.transform(x)

    def transform(self, x):
        return (x - self.mean) / (self.std + self.eps)

    def inverse_transform(self, x):
        return x * (self.std + self.eps) + self.mean

    def _as_tensor(self, v):
        return v if torch.is_tensor(v) else torch.from_numpy(np.asarray(v))

    def to(self, device, *args, **kwargs):
        self.mean = self._as_tensor(self.mean).float().to(device)
        self.std = self._as_tensor(self.std).float().to(device)
        return self

    def cuda(self, device=None):
        self.mean = self._as_tensor(self.mean).float().cuda(device)
        self.std = self._as_tensor(self.std).float().cuda(device)
        return self

    def cpu(self):
        self.mean = self._as_tensor(self.mean).float().cpu()
        self.std = self._as_tensor(self.std).float().cpu()
        return self


def _uniform_grid_2d(n_grid: int):
    t = np.linspace(0, 1, n_grid)
    gx, gy = np.meshgrid(t, t)
    return gx, gy


class DarcyDataset(Dataset):
    """Darcy-flow samples: coefficient ``a`` on the fine grid -> solution ``u`` (ex2) or the inverse
    map (ex3).  Items are dicts with the reference's keys: node, coeff, pos, grid, edge, mass,
    target, target_grad (ft.py:788-845)."""

    def __init__(self, data_path=None, normalizer_x=None, normalization=True, renormalization=False,
                 subsample_attn: int = 15, subsample_nodes: int = 1, subsample_inverse: int = 1,
                 subsample_method='nearest', subsample_method_inverse='average', n_krylov: int = 3,
                 uniform: bool = True, train_data=True, train_len=0.9, valid_len=0.0, online_features=False,
                 sparse_edge=False, return_edge=False, return_lap_only=True, return_boundary=True,
                 noise=0, random_state=1127802, inverse_problem=False, synthetic=None, n_fine_full=421,
                 n_samples_synthetic=64):
        self.data_path = data_path
        self.n_grid_fine = n_fine_full
        self.subsample_attn = subsample_attn
        self.subsample_nodes = subsample_nodes
        self.n_grid = int(((self.n_grid_fine - 1) / self.subsample_attn) + 1)
        self.n_f = int(((self.n_grid_fine - 1) / self.subsample_nodes) + 1)
        self.h = 1 / self.n_grid_fine
        self.train_data = train_data
        self.train_len = train_len
        self.valid_len = valid_len
        self.normalization = normalization
        self.normalizer_x = normalizer_x
        self.noise = noise
        self.return_boundary = return_boundary
        self.inverse_problem = inverse_problem
        self.random_state = random_state
        self.return_edge = return_edge
        self.eps = 1e-8
        if synthetic is None:
            synthetic = not (data_path is not None and os.path.exists(data_path))
        self.synthetic = synthetic
        self.n_samples_synthetic = n_samples_synthetic
        self._initialize()

    def __len__(self):
        return self.n_samples

    # ---- static helpers used by the example scripts -----------------------------------------------
    @staticmethod
    def get_grid(n_grid, subsample=1, return_elem=True, return_boundary=True):
        """Uniform [0,1]^2 grid.  return_elem=True: (nodes (n^2,2), triangles); else (n/s, n/s, 2)
        coordinate array (ft.py:643-672)."""
        gx, gy = _uniform_grid_2d(n_grid)
        if return_elem:
            nodes = np.c_[gx.ravel(), gy.ravel()]
            i, j = np.meshgrid(np.arange(n_grid - 1), np.arange(n_grid - 1))
            a = (i + j * n_grid).ravel()
            b, d = a + 1, a + n_grid
            c = d + 1
            elem = np.empty((2 * a.size, 3), dtype=np.int32)
            elem[0::2] = np.stack([a, c, d], 1)
            elem[1::2] = np.stack([b, c, a], 1)
            return nodes, elem
        s = subsample
        gx, gy = gx[::s, ::s], gy[::s, ::s]
        if not return_boundary:
            gx, gy = gx[1:-1, 1:-1], gy[1:-1, 1:-1]
        return np.stack([gx, gy], axis=-1)

    @staticmethod
    def get_scaler_sizes(n_f, n_c, scale_factor=True):
        """Interpolation factors/sizes of the CNN scalers for fine size n_f and coarse size n_c: the
        factor sqrt(n_c/n_f) rounded up onto the 0.005 lattice (ft.py:699-714)."""
        factor = np.round(np.sqrt(n_c / n_f), 4)
        last_digit = float(str(factor)[-1])
        factor = np.round(factor, 3)
        if last_digit < 5:
            factor += 5e-3
        factor = int(factor / 5e-3 + 5e-1) * 5e-3
        n_m = round(n_f * factor) - 1
        up_size = ((n_m, n_m), (n_f, n_f))
        if scale_factor:
            return (factor, factor), up_size
        return ((n_m, n_m), (n_c, n_c)), up_size

    @staticmethod
    def central_diff(x, h, padding=True):
        """(N, n, n) -> two (N, n, n) central differences, edge-padded (ft.py:630-641)."""
        if padding:
            x = np.pad(x, ((0, 0), (1, 1), (1, 1)), 'constant', constant_values=0)
        d, s = 2, 1
        grad_x = (x[:, d:, s:-s] - x[:, :-d, s:-s]) / d
        grad_y = (x[:, s:-s, d:] - x[:, s:-s, :-d]) / d
        return grad_x / h, grad_y / h

    # ---- data ------------------------------------------------------------------------------------------
    def _load_mat(self):
        from scipy.io import loadmat
        data = loadmat(self.data_path)
        return data['coeff'], data['sol']

    def _synthesize(self):
        rng = np.random.RandomState(self.random_state)
        N, n = self.n_samples_synthetic, self.n_grid_fine
        # piecewise-constant two-phase coefficient from a smooth random field (same shapes / value
        # ranges as piececonst_r421_N1024_smooth*.mat).  The "solution" is a deterministic smooth
        # functional of the coefficient -- the sine-series solve of  -lap(u) = 1/a  with zero Dirichlet
        # data, truncated to 12 modes per axis -- so a model can actually learn the map a -> u.  It is
        # not the variable-coefficient Darcy solve of the real dataset.
        k = 6
        coef = rng.randn(N, k, k)
        t = np.linspace(0, np.pi, n)
        basis = np.stack([np.cos(i * t) for i in range(k)], 0)
        field = np.einsum('nij,ix,jy->nxy', coef, basis, basis)
        a = np.where(field > 0, 12.0, 3.0)
        ks = 12
        sb = np.stack([np.sin((i + 1) * t) for i in range(ks)], 0)              # [ks, n]
        w = np.full(n, 1.0 / (n - 1)); w[0] = w[-1] = 0.5 / (n - 1)             # trapezoid weights on [0,1]
        sw = sb * w[None]
        fhat = 4.0 * np.einsum('nxj,ix->nij', (1.0 / a) @ sw.T, sw)
        idx = (1 + np.arange(ks)) ** 2
        fhat = fhat / (np.pi ** 2 * (idx[:, None] + idx[None, :]))[None]
        sol = np.einsum('ix,nij->nxj', sb, fhat) @ sb
        return a, sol

    def _initialize(self):
        a, u = self._synthesize() if self.synthetic else self._load_mat()
        n_total = a.shape[0]
        if self.train_data:
            n_take = int(self.train_len * n_total) if self.train_len <= 1 else int(self.train_len)
            a, u = a[:n_take], u[:n_take]
        else:
            if self.valid_len and self.valid_len > 0:
                n_take = int(self.valid_len * n_total) if self.valid_len <= 1 else int(self.valid_len)
            else:
                n_take = n_total
            a, u = a[-n_take:], u[-n_take:]
        self.n_samples = a.shape[0]
        s, sa = self.subsample_nodes, self.subsample_attn
        h = self.h * s
        gx, gy = self.central_diff(u[:, ::s, ::s], h)
        self.target_grad = np.stack([gx, gy], -1).astype(np.float32)
        nodes = a[:, ::s, ::s]
        targets = u[:, ::s, ::s]
        if self.inverse_problem:
            # ex3: recover the coarse coefficient from the (noisy) fine solution
            nodes, targets = u[:, ::s, ::s], a[:, ::sa, ::sa]
            if self.noise > 0:
                rng = np.random.RandomState(self.random_state)
                nodes = nodes + self.noise * np.abs(nodes).max() * rng.randn(*nodes.shape)
        self.coeff = a[:, ::s, ::s][..., None].astype(np.float32)
        nodes, targets = nodes[..., None].astype(np.float32), targets[..., None].astype(np.float32)
        if self.train_data and self.normalization:
            self.normalizer_x = UnitGaussianNormalizer()
            self.normalizer_y = UnitGaussianNormalizer()
            nodes = self.normalizer_x.fit_transform(nodes)
            if self.return_boundary:
                _ = self.normalizer_y.fit_transform(x=targets)
            else:
                _ = self.normalizer_y.fit_transform(x=targets[:, 1:-1, 1:-1, :])
        elif self.normalization and self.normalizer_x is not None:
            nodes = self.normalizer_x.transform(nodes)
        self.node_features, self.target = nodes, targets
        self.pos = self.get_grid(self.n_grid_fine, subsample=sa, return_elem=False).reshape(-1, 2)
        n_out = targets.shape[1]
        sub_out = (self.n_grid_fine - 1) // (n_out - 1)
        self.pos_fine = self.get_grid(self.n_grid_fine, subsample=sub_out, return_elem=False,
                                      return_boundary=self.return_boundary)
        self.nodes_elem = None

    @property
    def elem(self):
        return self.get_grid(self.n_grid)[1]

    def __getitem__(self, index):
        f = lambda v: torch.from_numpy(np.ascontiguousarray(v)).float()
        zero = torch.tensor([1.0])
        return dict(node=f(self.node_features[index]), coeff=f(self.coeff[index]), pos=f(self.pos),
                    grid=f(self.pos_fine), edge=zero, mass=zero, target=f(self.target[index]),
                    target_grad=f(self.target_grad[index]))


class BurgersDataset(Dataset):
    """1-D Burgers initial condition -> solution at t=1 (ft.py:24-370).  Loads burgers_data_R10.mat when
    present, otherwise synthesises smooth periodic pairs of the same shape."""

    def __init__(self, subsample: int, n_grid_fine=2 ** 13, viscosity: float = 0.1, n_krylov: int = 2,
                 smoother=None, uniform: bool = True, train_data=True, train_portion=0.5, valid_portion=0.1,
                 super_resolution: int = 1, data_path=None, online_features=False, return_edge=False,
                 renormalization=False, return_distance_features=True, return_mass_features=False,
                 return_downsample_grid: bool = True, random_sampling=False, random_state=1127802,
                 debug=False, synthetic=None, n_samples_synthetic=64):
        self.subsample = subsample
        self.n_grid_fine = n_grid_fine
        self.n_grid = n_grid_fine // subsample
        self.h = 1 / n_grid_fine
        self.train_data = train_data
        self.train_portion = train_portion
        self.valid_portion = valid_portion
        self.data_path = data_path
        self.random_state = random_state
        if synthetic is None:
            synthetic = not (data_path is not None and os.path.exists(data_path))
        self.synthetic = synthetic
        self.n_samples_synthetic = n_samples_synthetic
        self._initialize()

    def __len__(self):
        return self.n_samples

    @staticmethod
    def central_diff(x, h):
        pad = np.c_[x[:, -2:-1], x, x[:, 1:2]]          # periodic
        return (pad[:, 2:] - pad[:, :-2]) / (2 * h)

    def _initialize(self):
        if self.synthetic:
            rng = np.random.RandomState(self.random_state)
            t = np.linspace(0, 1, self.n_grid_fine, endpoint=False)
            k = np.arange(1, 9)
            amp = rng.randn(self.n_samples_synthetic, k.size, 2) / k[None, :, None]
            x = np.einsum('nk,kx->nx', amp[..., 0], np.sin(2 * np.pi * np.outer(k, t))) + \
                np.einsum('nk,kx->nx', amp[..., 1], np.cos(2 * np.pi * np.outer(k, t)))
            y = 0.6 * np.roll(x, self.n_grid_fine // 16, axis=1)
        else:
            from scipy.io import loadmat
            data = loadmat(self.data_path)
            x, y = data['a'], data['u']
        n = x.shape[0]
        n_train = int(self.train_portion * n) if self.train_portion <= 1 else int(self.train_portion)
        n_valid = int(self.valid_portion * n) if self.valid_portion <= 1 else int(self.valid_portion)
        sl = slice(0, n_train) if self.train_data else slice(n - n_valid, n)
        s = self.subsample
        x, y = x[sl, ::s], y[sl, ::s]
        self.n_samples = x.shape[0]
        self.node_features = x[..., None].astype(np.float32)
        self.target = y[..., None].astype(np.float32)
        self.target_diff = self.central_diff(y, self.h * s)[..., None].astype(np.float32)
        self.grid = np.linspace(0, 1, self.n_grid)[:, None].astype(np.float32)

    def __getitem__(self, index):
        f = lambda v: torch.from_numpy(np.ascontiguousarray(v)).float()
        one = torch.tensor([1.0])
        return dict(node=f(self.node_features[index]), pos=f(self.grid), grid=f(self.grid), edge=one,
                    mass=one, target=f(self.target[index]), target_grad=f(self.target_diff[index]))


# --------------------------------------------------------------------------------------- losses
class WeightedL2Loss(_WeightedLoss):
    """1-D relative L2 loss with optional H1-seminorm regulariser and latent orthogonaliser
    (ft.py:848-980).  As in the reference the weights gamma, alpha and delta are stored pre-multiplied by the
    mesh size h (ft.py:872-874), ``periodic`` is accepted but the difference stencil is always the interior
    one (ft.py:894-899), and ``orthogonal_reg`` gates the term on the latent features.
    Returns (loss, regularizer, orthogonalizer, metric)."""

    def __init__(self, dilation=2, regularizer=False, h=1 / 512, beta=1.0, gamma=1e-1, alpha=0.0,
                 metric_reduction='L1', periodic=False, return_norm=True, orthogonal_reg=False,
                 orthogonal_mode='global', delta=1e-4, noise=0.0, debug=False):
        super().__init__()
        assert dilation % 2 == 0
        self.dilation, self.regularizer, self.h = dilation, regularizer, h
        self.beta = beta
        self.gamma, self.alpha, self.delta = gamma * h, alpha * h, delta * h
        self.metric_reduction, self.periodic, self.return_norm = metric_reduction, periodic, return_norm
        self.orthogonal_reg, self.orthogonal_mode = orthogonal_reg, orthogonal_mode
        self.noise, self.eps, self.debug = noise, 1e-8, debug

    def central_diff(self, x, h=None):
        h = self.h if h is None else h
        d = self.dilation
        return (x[:, d:] - x[:, :-d]) / d / h

    def _reduce(self, v):
        return v.sqrt().mean() if self.return_norm else v.mean()

    def _orthogonalizer(self, preds_latent):
        # per latent tensor (N, L, E): mean squared off-diagonal of the Gram matrix (E x E for the global /
        # galerkin modes, L x L for local / fourier), scaled by delta*h
        terms = []
        for y in preds_latent:
            local = self.orthogonal_mode in ('local', 'fourier')
            if not local and self.orthogonal_mode not in ('global', 'galerkin', 'linear'):
                raise ValueError(self.orthogonal_mode)
            gram = y @ y.transpose(-2, -1) if local else y.transpose(-2, -1) @ y
            with torch.no_grad():
                diag = torch.diag_embed(y.pow(2).sum(dim=-1 if local else -2))
            terms.append(self.delta * (gram - diag).pow(2).mean(dim=(-1, -2)))
        return self._reduce(torch.stack(terms, dim=-1))

    def forward(self, preds, targets, preds_prime=None, targets_prime=None, preds_latent: list = [],
                K=None):
        h = self.h
        if self.noise > 0:
            assert 0 <= self.noise <= 0.2
            with torch.no_grad():
                targets = targets * (1.0 + self.noise * torch.rand_like(targets))
        target_norm = h * targets.pow(2).sum(dim=1)
        targets_prime_norm = h * targets_prime.pow(2).sum(dim=1) if targets_prime is not None else 1
        loss = self.beta * (h * (preds - targets).pow(2)).sum(dim=1) / target_norm
        if preds_prime is not None and self.alpha > 0:
            loss = loss + self.alpha * (h * (preds_prime - K * targets_prime).pow(2)).sum(dim=1) \
                / targets_prime_norm
        if self.metric_reduction == 'L2':
            metric = loss.mean().sqrt().item()
        elif self.metric_reduction == 'L1':
            metric = loss.sqrt().mean().item()
        elif self.metric_reduction == 'Linf':
            metric = loss.sqrt().max().item()
        loss = self._reduce(loss)
        zero = lambda: torch.tensor([0.0], requires_grad=True, device=preds.device)
        if self.regularizer and self.gamma > 0 and targets_prime is not None:
            s = self.dilation // 2
            reg = self.gamma * h * (targets_prime[:, s:-s] - self.central_diff(preds)).pow(2).sum(dim=1) \
                / targets_prime_norm
            reg = self._reduce(reg)
        else:
            reg = zero()
        ortho = self._orthogonalizer(preds_latent) if (self.orthogonal_reg > 0 and preds_latent) else zero()
        return loss, reg, ortho, metric


class WeightedL2Loss2d(_WeightedLoss):
    """2-D relative L2 loss + gamma*h*H1-seminorm regulariser by central differences (ft.py:983-1105).
    Returns (loss, regularizer, metric, norms)."""

    def __init__(self, dim=2, dilation=2, regularizer=False, h=1 / 421, beta=1.0, gamma=1e-1, alpha=0.0,
                 delta=0.0, metric_reduction='L1', return_norm=True, noise=0.0, eps=1e-10, debug=False):
        super().__init__()
        assert dilation % 2 == 0
        self.noise, self.regularizer, self.dilation, self.dim, self.h = noise, regularizer, dilation, dim, h
        self.beta, self.gamma, self.alpha = beta, gamma, alpha
        self.delta = delta * h ** dim
        self.eps, self.metric_reduction, self.return_norm = eps, metric_reduction, return_norm

    def central_diff(self, u, h=None):
        h = self.h if h is None else h
        d = self.dilation
        s = d // 2
        gx = (u[:, d:, s:-s] - u[:, :-d, s:-s]) / d
        gy = (u[:, s:-s, d:] - u[:, s:-s, :-d]) / d
        return torch.stack([gx, gy], dim=-1) / h

    def terms(self, preds, targets, preds_prime=None, targets_prime=None, weights=None, K=None):
        """Per-sample relative errors without any host read-back (what a captured training step uses):
        (loss [N], regulariser [N] or None, norms).  forward() reduces them and adds the metric."""
        h = self.h if weights is None else weights
        d = self.dim
        K = torch.tensor(1) if K is None else K
        if self.noise > 0:
            assert 0 <= self.noise <= 0.2
            with torch.no_grad():
                targets = targets * (1.0 + self.noise * torch.rand_like(targets))
        target_norm = targets.pow(2).mean(dim=(1, 2)) + self.eps
        if targets_prime is not None:
            targets_prime_norm = d * (K * targets_prime.pow(2)).mean(dim=(1, 2, 3)) + self.eps
        else:
            targets_prime_norm = 1
        loss = self.beta * (preds - targets).pow(2).mean(dim=(1, 2)) / target_norm
        if preds_prime is not None and self.alpha > 0:
            gd = (K * (preds_prime - targets_prime)).pow(2)
            loss = loss + self.alpha * gd.mean(dim=(1, 2, 3)) / targets_prime_norm
        reg = None
        if self.regularizer and targets_prime is not None:
            s = self.dilation // 2
            pd = self.central_diff(preds)
            tp = targets_prime[:, s:-s, s:-s, :].contiguous()
            if K.ndim > 1:
                K = K[:, s:-s, s:-s].contiguous()
            reg = self.gamma * h * (K * (tp - pd)).pow(2).mean(dim=(1, 2, 3)) / targets_prime_norm
        return loss, reg, dict(L2=target_norm, H1=targets_prime_norm)

    def reduce(self, v):
        return v.sqrt().mean() if self.return_norm else v.mean()

    def forward(self, preds, targets, preds_prime=None, targets_prime=None, weights=None, K=None):
        loss, reg, norms = self.terms(preds, targets, preds_prime, targets_prime, weights, K)
        if self.metric_reduction == 'L2':
            metric = loss.mean().sqrt().item()
        elif self.metric_reduction == 'L1':
            metric = loss.sqrt().mean().item()
        elif self.metric_reduction == 'Linf':
            metric = loss.sqrt().max().item()
        loss = self.reduce(loss)
        if reg is not None:
            reg = self.reduce(reg)
        else:
            reg = torch.tensor([0.0], requires_grad=True, device=preds.device)
        return loss, reg, metric, norms
