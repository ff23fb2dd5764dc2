"""Child process of tests/test_host_sanitizer.py: loads the AddressSanitizer + UndefinedBehaviorSanitizer build of the library's
HOST code (GFS3D_SAN_LIB; the sanitizer runtime is preloaded by the parent) and replays tests/host_cases.py against it.
No torch, no GPU: nothing is launched - argument checks, workspace plans and the launch sequences' first HIP error."""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "gfs-3dseg_gws_amd"))

import host_cases                      # noqa: E402
from gfs3d import _lib                 # noqa: E402  (struct mirrors and the symbol table only; lib() is not called)

L = ctypes.CDLL(os.environ["GFS3D_SAN_LIB"])
for name, (res, args) in _lib.SYMBOLS.items():
    fn = getattr(L, name)
    fn.restype, fn.argtypes = res, args
assert L.gfs3d_version() > 0
host_cases.round2_cases(L, _lib)
host_cases.round3_cases(L, _lib)
host_cases.round5_cases(L, _lib)
host_cases.round6_cases(L, _lib)
host_cases.workspace_sweep(L)
print("SAN_DRIVER_OK")
