"""N>1 path on CPU: two gloo ranks.  The flat-bucket gradient all-reduce must reproduce the
single-process gradient of the concatenated batch, parameters must be synchronised by the
broadcast, and clipping after the reduction must see the same (global) norm on every rank."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(12, 16), torch.nn.SiLU(), torch.nn.Linear(16, 3))


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "galerkin-transformer_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from galerkin_transformer.distributed import (FlatGradAllReducer, broadcast_parameters,
                                                  init_distributed, shard_range)
    r, _, w = init_distributed("gloo")
    assert (r, w) == (rank, world)
    model = _model(100 + rank)                    # deliberately different init per rank
    broadcast_parameters(model, src=0)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    lo, hi = shard_range(8, rank, world)
    loss = ((model(X[lo:hi]) - Y[lo:hi]) ** 2).mean()
    loss.backward()
    red = FlatGradAllReducer(model.parameters())
    red.reduce()
    norm = torch.nn.utils.clip_grad_norm_(model.parameters(), 0.05)
    # numpy arrays travel through the queue by value (torch tensors would be shared through a file descriptor
    # served by this process, which may exit before the parent reads them)
    q.put((rank, [p.detach().numpy().copy() for p in model.parameters()],
           [p.grad.numpy().copy() for p in model.parameters()], float(norm), red.nbytes))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_flat_allreduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=100) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    # single-process reference on the full batch with rank 0's initial parameters
    ref = _model(100)
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    ((ref(X) - Y) ** 2).mean().backward()
    ref_norm = torch.nn.utils.clip_grad_norm_(ref.parameters(), 0.05)
    for rank, params, grads, norm, nbytes in res:
        params = [torch.from_numpy(a) for a in params]
        grads = [torch.from_numpy(a) for a in grads]
        for p, rp in zip(params, ref.parameters()):
            assert torch.equal(p, rp.detach())                    # broadcast made the ranks identical
        for gr, rp in zip(grads, ref.parameters()):
            assert torch.allclose(gr, rp.grad, rtol=1e-5, atol=1e-7)
        assert abs(norm - float(ref_norm)) < 1e-6 * max(1.0, float(ref_norm))
        assert nbytes == 4 * sum(p.numel() for p in ref.parameters())


def test_shard_range_and_rank_seed():
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "galerkin-transformer_amd"))
    from galerkin_transformer.distributed import rank_seed, shard_range
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert shard_range(10, 3, 4, drop_last=False) == (9, 10)
    assert len({rank_seed(1127802, r) for r in range(8)}) == 8


def _worker8(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "galerkin-transformer_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from galerkin_transformer.distributed import broadcast_parameters, init_distributed, rank_seed, shard_range
    from galerkin_transformer.optim import FlatClipAdam
    init_distributed("gloo")
    model = _model(300 + rank)
    broadcast_parameters(model, src=0)
    opt = FlatClipAdam(model.parameters(), lr=1e-3, max_norm=0.99)      # buckets + collective; the HIP update is not run here
    assert opt.world == world
    g = torch.Generator().manual_seed(9)
    X, Y = torch.randn(64, 12, generator=g), torch.randn(64, 3, generator=g)
    lo, hi = shard_range(64, rank, world)
    ((model(X[lo:hi]) - Y[lo:hi]) ** 2).mean().backward()
    opt.gather_grads()
    opt.all_reduce()
    try:
        opt.apply()
        refused = False
    except RuntimeError as e:
        refused = "no CPU fallback" in str(e)
    q.put((rank, (lo, hi), rank_seed(1127802, rank), opt.flat_grad.numpy().copy() / world, list(opt.offsets), refused))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_flat_clip_adam_bucket_allreduce_eight_ranks_gloo():
    """the path the driver's 8-GPU run takes -- FlatClipAdam's flat gradient bucket (16-byte
    aligned parameter offsets), ONE in-place sum-all-reduce, the 1 / world average folded in afterwards -- with EIGHT ranks
    over gloo: the averaged bucket equals the single-process gradient of the whole batch on every rank, shards partition the
    batch, dropout seeds differ per rank, and the optimizer update itself refuses CPU tensors (no CPU fallback)."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    ref = _model(300)
    g = torch.Generator().manual_seed(9)
    X, Y = torch.randn(64, 12, generator=g), torch.randn(64, 3, generator=g)
    ((ref(X) - Y) ** 2).mean().backward()
    shards = [r[1] for r in res]
    assert shards == [(8 * i, 8 * i + 8) for i in range(8)]
    assert len({r[2] for r in res}) == 8
    for rank, _, _, flat, offsets, refused in res:
        assert refused, "FlatClipAdam.apply() must not run on CPU tensors"
        flat = torch.from_numpy(flat)
        for p, o in zip(ref.parameters(), offsets):
            assert o % 4 == 0
            assert torch.allclose(flat[o:o + p.numel()].view_as(p), p.grad, rtol=1e-5, atol=1e-7), rank
