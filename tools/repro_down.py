import faulthandler, os, sys
faulthandler.enable()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch, bench
import galerkin_transformer as gt
dev = torch.device("cuda:0")
B = int(sys.argv[1]); mode = sys.argv[2]
torch.manual_seed(0)
model = gt.FourierTransformer2D(**bench.darcy_config()).to(dev)
model.train(mode != "eval")
node, pos, grid, target = bench.synthetic_batch(B, dev, 1)
mods = {"down": (model.downscaler, lambda m: m(node)),
        "up": (model.upscaler, lambda m: m(torch.randn(B, 43, 43, 128, device=dev)))}
for name, (m, f) in mods.items():
    for it in range(3):
        for p in m.parameters(): p.grad = None
        y = f(m); y.square().mean().backward(); torch.cuda.synchronize()
    print(name, mode, "ok", float(y.square().mean()), flush=True)
