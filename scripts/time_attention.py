"""SelfAttention (256 -> 64 head, N = 2048, B = 32): fused HIP kernel (K5) vs the torch formulation, fwd and fwd+bwd."""
import sys, time, torch
sys.path.insert(0, __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))), "gfs-3dseg_gws_amd"))
from gfs3d import ops
dev = torch.device("cuda:0")
B, N = 32, 2048
g = torch.Generator().manual_seed(0)
q, k, v = (torch.randn(B, 64, N, generator=g).to(dev).requires_grad_(True) for _ in range(3))
cot = torch.randn(B, 64, N, device=dev)
def fused(train):
    y = ops.attention(q, k, v, 0.125, 0.1 if train else 0.0, 7)
    if train: y.backward(cot)
def ref(train):
    attn = torch.softmax(torch.bmm(q.transpose(1, 2) * 0.125, k), dim=-1)
    if train: attn = torch.nn.functional.dropout(attn, 0.1, True)
    y = torch.bmm(attn, v.transpose(1, 2)).transpose(1, 2)
    if train: y.backward(cot)
for name, fn in (("fused", fused), ("torch", ref)):
    for train in (False, True):
        for _ in range(2): fn(train)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): fn(train)
        torch.cuda.synchronize()
        print(name, "fwd+bwd (dropout 0.1)" if train else "fwd (eval)", "%.3f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
