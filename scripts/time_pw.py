"""Time the point-wise stack (K4) at BASELINE config 2 size; run under rocprofv3 --kernel-trace --stats for per-kernel times.
    python scripts/time_pw.py [f32|f16|bf16] [B = 32] [store16 = 1]      (B = 64: the configs[4] batch; store16 0: fp32 storage in a reduced mode)"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gfs-3dseg_gws_amd"))
from gfs3d import ops
from model.dgcnn import conv1d
dev = torch.device("cuda:0")
torch.manual_seed(0)
conv = conv1d(192, [512, 256]).to(dev).train()
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
ops.set_store16(bool(int(sys.argv[3])) if len(sys.argv) > 3 else True)
xs = [torch.randn(B, 2048, 64, device=dev).requires_grad_(True) for _ in range(3)]
cot = torch.randn(B, 256, 2048, device=dev)
parts = conv.fused_parts()
mode = sys.argv[1] if len(sys.argv) > 1 else "f32"          # f32 | f16 | bf16
def step():
    for p in conv.parameters(): p.grad = None
    for t in xs: t.grad = None
    with ops.compute(mode):
        y = ops.pointwise_stack(xs, parts, True)
        y.backward(cot)
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 50
for _ in range(n): step()
torch.cuda.synchronize()
print("pointwise fwd+bwd ms/iter", (time.perf_counter() - t0) / n * 1e3)
