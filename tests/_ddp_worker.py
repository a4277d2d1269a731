"""Worker of tests/test_train_gpu.py::test_ddp_step_equals_single_process (one rank of a two-rank gloo group sharing
cuda:0, or the single-process run on the concatenated batch): one eager training step with FlatClipAdam, every dropout
off, and the flat averaged gradient / clip norm / parameters after the step written to a file."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
import torch.distributed as dist


def main():
    out_path, per_rank = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    import galerkin_transformer as gt
    cfg = bench.darcy_config()
    for k in ("dropout", "downscaler_dropout", "upscaler_dropout", "ffn_dropout", "encoder_dropout", "decoder_dropout"):
        cfg[k] = 0.0
    torch.manual_seed(1127802)
    model = gt.FourierTransformer2D(**cfg).to(dev).train()
    gt.set_attention_dropout("off")
    full = bench.synthetic_batch(per_rank * max(world, 2), dev, seed=4242)       # the same 2B samples in every process
    lo, hi = (rank * per_rank, (rank + 1) * per_rank) if world > 1 else (0, per_rank * 2)
    batch = {k: v[lo:hi].contiguous() for k, v in full.items()}
    tr = bench.Trainer(model, batch, world, lr=1e-3, clip=0.99, use_graph=False)
    tr.eager_step()
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"grad": (tr.opt.flat_grad / world).cpu(), "norm": tr.opt.grad_norm(), "param": tr.opt.flat_param.cpu(),
                    "loss": float(tr.loss.item()), "world": world}, out_path)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
