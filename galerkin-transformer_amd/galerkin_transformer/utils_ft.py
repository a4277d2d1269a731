"""Training driver, CLI arguments and paths with the reference's names (reference: libs/utils_ft.py).

Host-side orchestration only: it calls ``model(node, edge, pos, grid)`` whose hot path runs in
libgt_hip.so.  Plotting helpers degrade to no-ops when matplotlib / plotly are not importable.
"""
from __future__ import annotations

import argparse
import copy
import math
import os
from collections import OrderedDict, defaultdict  # noqa: F401  (star-imported by the examples)
from datetime import date

import numpy as np
import torch
import torch.nn.functional as F  # noqa: F401
import yaml  # noqa: F401
from torch import nn
from torch.optim.lr_scheduler import OneCycleLR  # noqa: F401
from torch.utils.data import DataLoader  # noqa: F401
from tqdm.auto import tqdm

from .utils import Colors, color, get_num_params, get_seed, is_interactive, load_pickle, save_pickle  # noqa: F401

try:
    import matplotlib.pyplot as plt  # noqa: F401
except Exception:                                  # pragma: no cover
    plt = None

current_path = os.path.dirname(os.path.abspath(__file__))
SRC_ROOT = os.path.dirname(current_path)
MODEL_PATH = os.environ.get('MODEL_PATH') or os.path.join(SRC_ROOT, 'models')
DATA_PATH = os.environ.get('DATA_PATH') or os.path.join(SRC_ROOT, 'data')
FIG_PATH = os.environ.get('FIG_PATH') or os.path.join(os.path.dirname(SRC_ROOT), 'figures')
EPOCH_SCHEDULERS = ['ReduceLROnPlateau', 'StepLR', 'MultiplicativeLR', 'MultiStepLR', 'ExponentialLR',
                    'LambdaLR']
PI = math.pi
SEED = int(os.environ.get('SEED') or 1127802)


def clones(module, N):
    return nn.ModuleList([copy.deepcopy(module) for _ in range(N)])


def get_model_name(model='burgers', num_encoder_layers=4, n_hidden=96, attention_type='fourier',
                   layer_norm=True, grid_size=512, inverse_problem=False, additional_str: str = ''):
    """Checkpoint / result file names, e.g. darcy_141_6gt_128d_qkv_32f_2026-09-25.pt."""
    tag = {'fourier': 'ft', 'galerkin': 'gt', 'linear': 'lt', 'softmax': 'st'}.get(attention_type, 'att')
    parts = ['burgers' if model == 'burgers' else 'darcy']
    if inverse_problem:
        parts.append('inv')
    parts += [str(grid_size), f'{num_encoder_layers}{tag}', f'{n_hidden}d', 'ln' if layer_norm else 'qkv']
    if additional_str:
        parts.append(additional_str.rstrip('_'))
    stem = '_'.join(parts) + '_' + str(date.today())
    return stem + '.pt', stem + '.pkl'


def _common_args(parser, attention_default, ffn_dropout, encoder_dropout, decoder_dropout, gamma):
    a = parser.add_argument
    a('--attention-type', type=str, default=attention_default, metavar='attn_type',
      help='encoder attention: fourier (integral, local) or galerkin (global) run on the HIP path')
    a('--xavier-init', type=float, default=0.01, metavar='xavier_init')
    a('--diagonal-weight', type=float, default=0.01, metavar='diagonal weight')
    a('--ffn-dropout', type=float, default=ffn_dropout, metavar='ffn_dropout')
    a('--encoder-dropout', type=float, default=encoder_dropout, metavar='encoder_dropout')
    a('--decoder-dropout', type=float, default=decoder_dropout, metavar='decoder_dropout')
    a('--layer-norm', action='store_true', default=False,
      help='conventional LayerNorm after the residuals instead of the per-head attention norm')
    a('--epochs', type=int, default=100, metavar='N')
    a('--lr', type=float, default=1e-3, metavar='LR')
    a('--gamma', type=float, default=gamma, metavar='regularizer')
    a('--no-cuda', action='store_true', default=False)
    a('--show-batch', action='store_true', default=False)
    a('--seed', type=int, default=SEED, metavar='Seed')


def get_args_1d():
    parser = argparse.ArgumentParser(description="Example 1: Burgers equation")
    parser.add_argument('--subsample', type=int, default=4, metavar='subsample',
                        help='input sampling from 8192 (default: 4 i.e., 2048 grid)')
    parser.add_argument('--batch-size', type=int, default=8, metavar='bsz')
    parser.add_argument('--val-batch-size', type=int, default=4, metavar='bsz')
    _common_args(parser, 'fourier', 0.0, 0.0, 0.0, 0.1)
    return parser.parse_args()


def get_args_2d(subsample_nodes=3, subsample_attn=10, gamma=0.5, noise=0.0, ffn_dropout=0.1,
                encoder_dropout=0.05, decoder_dropout=0.0, dropout=0.0, inverse=False, **kwargs):
    title = ('Example 3: inverse coefficient identification problem for Darcy interface flow' if inverse
             else 'Example 2: Darcy interface flow')
    parser = argparse.ArgumentParser(description=title)
    parser.add_argument('--subsample-nodes', type=int, default=subsample_nodes, metavar='subsample')
    parser.add_argument('--subsample-attn', type=int, default=6, metavar='subsample_attn')
    parser.add_argument('--batch-size', type=int, default=4, metavar='bsz')
    parser.add_argument('--val-batch-size', type=int, default=4, metavar='bsz')
    parser.add_argument('--noise', type=float, default=noise, metavar='noise')
    parser.add_argument('--dropout', type=float, default=dropout, metavar='dropout')
    parser.add_argument('--no-scale-factor', action='store_true', default=False)
    _common_args(parser, 'galerkin', ffn_dropout, encoder_dropout, decoder_dropout, gamma)
    return parser.parse_args()


# --------------------------------------------------------------------------------------- batches
def _forward(model, data, device):
    x, edge = data["node"].to(device), data["edge"].to(device)
    pos, grid = data['pos'].to(device), data['grid'].to(device)
    out_ = model(x, edge, pos=pos, grid=grid)
    return out_['preds'] if isinstance(out_, dict) else out_[0], out_


def _finish_step(model, loss, optimizer, lr_scheduler, grad_clip):
    loss.backward()
    if hasattr(optimizer, "flat_grad") and hasattr(optimizer, "max_norm"):
        # optim.FlatClipAdam: the clip is part of its fused step, applied to the all-reduced (averaged) gradient -- also
        # when the optimizer was built with max_norm=None (a local clip here would see the pre-average norm)
        optimizer.max_norm = float(grad_clip) if grad_clip else 0.0
    else:
        nn.utils.clip_grad_norm_(model.parameters(), grad_clip)
    optimizer.step()
    if lr_scheduler:
        lr_scheduler.step()
    try:                                    # one seed bump per step: fresh stateless-RNG dropout masks
        from . import _hip
        p = next(model.parameters())
        if p.is_cuda:
            _hip.advance_seed(p.device)
    except Exception:
        pass


def train_batch_burgers(model, loss_func, data, optimizer, lr_scheduler, device, grad_clip=0.999):
    optimizer.zero_grad()
    out, out_ = _forward(model, data, device)
    y_latent = out_['preds_latent'] if isinstance(out_, dict) else None
    target = data["target"].to(device)
    tgrad = data["target_grad"].to(device) if "target_grad" in data else None
    if target.size(-1) >= 2:
        u, up = target[..., 0], target[..., 1]
    else:
        u, up = target[..., 0], (tgrad[..., 0] if tgrad is not None else None)
    if out.size(2) == 2:
        u_pred, up_pred = out[..., 0], out[..., 1]
        loss, reg, ortho, _ = loss_func(u_pred, u, up_pred, up, preds_latent=y_latent)
    else:
        u_pred = up_pred = out[..., 0]
        loss, reg, ortho, _ = loss_func(u_pred, u, targets_prime=up, preds_latent=y_latent)
    total = loss + reg + ortho
    _finish_step(model, total, optimizer, lr_scheduler, grad_clip)
    return (total.item(), reg.item(), ortho.item()), u_pred, up_pred


def train_batch_darcy(model, loss_func, data, optimizer, lr_scheduler, device, grad_clip=0.99):
    optimizer.zero_grad()
    a = data["coeff"].to(device)
    u, gradu = data["target"].to(device), data["target_grad"].to(device)
    out, _ = _forward(model, data, device)
    if out.ndim == 4:
        u_pred, pred_grad, target = out[..., 0], out[..., 1:], u[..., 0]
        loss, reg, _, _ = loss_func(u_pred, target, pred_grad, gradu, K=a)
    else:
        u_pred, pred_grad = out[..., 0], out[..., 1:]
        loss, reg, _, _ = loss_func(u_pred, u[..., 0], targets_prime=gradu, K=a)
    total = loss + reg
    _finish_step(model, total, optimizer, lr_scheduler, grad_clip)
    up_pred = pred_grad if pred_grad.numel() else u_pred
    return (total.item(), reg.item()), u_pred, up_pred


def _validate(model, metric_func, valid_loader, device, which):
    model.eval()
    vals = []
    for data in valid_loader:
        with torch.no_grad():
            out, _ = _forward(model, data, device)
            target = data["target"].to(device)
            if which == 'burgers':
                tgrad = data.get("target_grad")
                up = target[..., 1] if target.size(-1) >= 2 else (tgrad.to(device)[..., 0]
                                                                  if tgrad is not None else None)
                _, _, _, metric = metric_func(out[..., 0], target[..., 0], targets_prime=up)
            else:
                _, _, metric, _ = metric_func(out[..., 0], target[..., 0])
            vals.append(metric.item() if torch.is_tensor(metric) else metric)
    return dict(metric=np.mean(vals, axis=0))


def validate_epoch_burgers(model, metric_func, valid_loader, device):
    return _validate(model, metric_func, valid_loader, device, 'burgers')


def validate_epoch_darcy(model, metric_func, valid_loader, device):
    return _validate(model, metric_func, valid_loader, device, 'darcy')


# --------------------------------------------------------------------------------------- training loop
def run_train(model, loss_func, metric_func, train_loader, valid_loader, optimizer, lr_scheduler,
              train_batch=None, validate_epoch=None, epochs=10, device="cuda", mode='min',
              tqdm_mode='batch', patience=10, grad_clip=0.999, start_epoch: int = 0,
              model_save_path=MODEL_PATH, save_mode='state_dict', model_name='model.pt',
              result_name='result.pt'):
    """Epoch loop with best-validation checkpointing (state_dict or whole module) and a pickled history
    (loss_train, loss_val, lr_history, optimizer_state), as the reference's run_train."""
    os.makedirs(model_save_path, exist_ok=True)
    patience = epochs if not patience else patience
    better = (lambda a, b: a > b) if mode == 'max' else (lambda a, b: a < b)
    best_metric, best_epoch, stale = (-np.inf if mode == 'max' else np.inf), None, 0
    epoch_sched = lr_scheduler is not None and any(s in str(lr_scheduler.__class__) for s in EPOCH_SCHEDULERS)
    hist_train, hist_val, hist_lr = [], [], []
    result = {}
    per_epoch_bar = tqdm_mode != 'batch'
    with tqdm(total=epochs, disable=not per_epoch_bar) as bar_ep:
        for epoch in range(start_epoch, start_epoch + epochs):
            model.train()
            losses = []
            with tqdm(total=len(train_loader), disable=per_epoch_bar) as bar:
                for batch in train_loader:
                    loss, _, _ = train_batch(model, loss_func, batch, optimizer,
                                             None if epoch_sched else lr_scheduler, device,
                                             grad_clip=grad_clip)
                    losses.append(np.array(loss))
                    lr = optimizer.param_groups[0]['lr']
                    hist_lr.append(lr)
                    mean = np.mean(losses, axis=0)
                    txt = f"epoch: [{epoch + 1}/{start_epoch + epochs}]"
                    txt += (f" loss: {mean:.3e}" if mean.ndim == 0 else
                            "".join(f" | loss {j}: {v:.3e}" for j, v in enumerate(mean) if v > 0))
                    bar.set_description(txt + f" | current lr: {lr:.3e}")
                    bar.update()
            hist_train.append(mean)
            val = validate_epoch(model, metric_func, valid_loader, device)
            hist_val.append(val["metric"])
            val_metric = float(np.sum(val["metric"]))
            if better(val_metric, best_metric):
                best_metric, best_epoch, stale = val_metric, epoch, 0
                target = os.path.join(model_save_path, model_name)
                torch.save(model.state_dict() if save_mode in (None, 'state_dict') else model, target)
            else:
                stale += 1
            if epoch_sched:
                if 'ReduceLROnPlateau' in str(lr_scheduler.__class__):
                    lr_scheduler.step(val_metric)
                else:
                    lr_scheduler.step()
            msg = (color(f"| val metric: {val_metric:.3e} ", Colors.blue) +
                   color(f"| best val: {best_metric:.3e} at epoch {best_epoch + 1}", Colors.yellow) +
                   color(f" | early stop: {stale} ", Colors.red) + color(f" | current lr: {lr:.3e}", Colors.magenta))
            if per_epoch_bar:
                bar_ep.set_description(msg)
                bar_ep.update()
            else:
                tqdm.write("\n" + msg + "\n")
            result = dict(best_val_epoch=best_epoch, best_val_metric=best_metric,
                          loss_train=np.asarray(hist_train), loss_val=np.asarray(hist_val),
                          lr_history=np.asarray(hist_lr), optimizer_state=optimizer.state_dict())
            save_pickle(result, os.path.join(model_save_path, result_name))
            if stale > patience:
                print(f"Early stop at epoch {epoch}")
                break
    return result


# --------------------------------------------------------------------------------------- plotting (optional)
def showsolution(node, elem, u, **kwargs):
    """3-D surface of a nodal function on a triangulation (plotly); returns None without plotly."""
    try:
        import plotly.figure_factory as ff
    except Exception:
        return None
    fig = ff.create_trisurf(x=node[:, 0], y=node[:, 1], z=u, simplices=elem, colormap="Viridis",
                            showbackground=True, aspectratio=dict(x=1, y=1, z=1))
    fig.update_layout(**{k: v for k, v in kwargs.items() if k in ('width', 'height', 'template')})
    fig.show()
    return fig


def showcontour(z, **kwargs):
    """Contour plot of a grid function (plotly); returns None without plotly."""
    try:
        import plotly.graph_objects as go
    except Exception:
        return None
    fig = go.Figure(data=go.Contour(z=z, colorscale='RdYlBu', line_smoothing=0.85,
                                    contours=dict(coloring='heatmap')))
    fig.update_layout(**{k: v for k, v in kwargs.items() if k in ('width', 'height', 'template')})
    fig.show()
    return fig


# --------------------------------------------------------------------------------------- data path
class DeviceResidentLoader:
    """Batches served from HBM: the whole map-style dataset is uploaded ONCE (ex2's 1024 x 141^2 train set is 81 MB
    per field -- nothing next to 288 GB), fields that are identical for every sample (``pos``, ``grid``, ``edge``)
    are stored once and expanded per batch, and an epoch is a device-side permutation + gather.

    Replaces the reference's per-sample numpy -> torch re-wrap, default-collate and per-batch H2D copy
    (libs/ft.py:788-845, libs/utils_ft.py:658-661) for the train loops in this module, which then find every
    ``data[key].to(device)`` a no-op.  Same iteration protocol as ``DataLoader`` (``len``, ``iter``, dict batches,
    ``drop_last``, ``shuffle``); the shuffle order comes from a seeded device generator, so it differs from the CPU
    ``DataLoader``'s order for the same seed."""

    def __init__(self, dataset, batch_size: int, device, shuffle: bool = False, drop_last: bool = False,
                 seed: int = 1127802):
        self.n, self.batch_size, self.shuffle, self.drop_last = len(dataset), int(batch_size), shuffle, drop_last
        self.device = torch.device(device)
        first = dataset[0]
        second = dataset[1] if self.n > 1 else first
        self.shared, self.fields = {}, {}
        per_sample = [k for k, v in first.items() if not (torch.is_tensor(v) and torch.equal(v, second[k]) and
                                                          k in ("pos", "grid", "edge", "mass", "pos_fine"))]
        stacks = {k: [] for k in per_sample}
        for i in range(self.n):
            item = dataset[i] if i > 1 else (first if i == 0 else second)
            for k in per_sample:
                stacks[k].append(item[k])
        for k, v in first.items():
            if k in stacks:
                host = torch.stack(stacks[k])
                if torch.cuda.is_available() and self.device.type == "cuda":
                    host = host.pin_memory()
                self.fields[k] = host.to(self.device, non_blocking=True)
            else:
                self.shared[k] = v.to(self.device)
        self._gen = torch.Generator(device=self.device if self.device.type == "cuda" else "cpu")
        self._gen.manual_seed(seed)

    def __len__(self):
        return self.n // self.batch_size if self.drop_last else -(-self.n // self.batch_size)

    def __iter__(self):
        order = (torch.randperm(self.n, device=self.device, generator=self._gen) if self.shuffle
                 else torch.arange(self.n, device=self.device))
        for b in range(len(self)):
            idx = order[b * self.batch_size:(b + 1) * self.batch_size]
            batch = {k: v.index_select(0, idx) for k, v in self.fields.items()}
            for k, v in self.shared.items():
                batch[k] = v.unsqueeze(0).expand(idx.numel(), *v.shape)
            yield batch
