import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
from galerkin_transformer import ops, _hip as H
dev = torch.device("cuda")
T = 1272384
x = torch.randn(T, 32, device=dev, requires_grad=True)
w1 = torch.randn(128, 32, device=dev, requires_grad=True); b1 = torch.zeros(128, device=dev, requires_grad=True)
w2 = torch.randn(1, 128, device=dev, requires_grad=True); b2 = torch.zeros(1, device=dev, requires_grad=True)
g = torch.randn(T, 1, device=dev)
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / n * 1e3
for name, f in (("fused", lambda: ops.mlp_head(x, w1, b1, w2, b2, act="silu")),
                ("unfused", lambda: ops.linear(ops.linear(x, w1, b1, act="silu"), w2, b2))):
    fw = t(f)
    y = f()
    def fb():
        for p_ in (x, w1, b1, w2, b2): p_.grad = None
        f().backward(g)
    print(name, "fwd %.0f us" % fw, "fwd+bwd %.0f us" % t(fb))
