"""Where does the FIRST c3 flow of a process spend its extra Lloyd time?  Wraps the pieces of get_basis._lloyd with synchronised timers and
runs bench.c3_end_to_end twice.  python scripts/c3_first_call.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "gfs-3dseg_gws_amd")):
    sys.path.insert(0, _p)
import torch, bench, get_basis
from gfs3d import ops
acc = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
        return r
    return w
get_basis._lloyd_run_ahead = timed("run_ahead", get_basis._lloyd_run_ahead)
get_basis._lloyd = timed("lloyd_total", get_basis._lloyd)
ops.KMeansDelta.__init__ = timed("delta_init", ops.KMeansDelta.__init__)
ops.KMeansDelta.load = timed("delta_load", ops.KMeansDelta.load)
_pin = torch.Tensor.pin_memory
torch.Tensor.pin_memory = timed("pin_memory", _pin)
dev = torch.device("cuda:0")
for i in range(2):
    acc.clear()
    d = bench.c3_end_to_end(dev)
    print(i, d["total_ms"], d["stage_ms"]["lloyd"], {k: round(v * 1e3, 1) for k, v in acc.items()}, flush=True)
