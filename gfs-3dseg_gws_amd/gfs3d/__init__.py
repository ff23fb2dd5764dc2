"""gfs3d - host-side binding of the MI355X hot path (libgfs3d_hip.so).

Layout of the package directory ``gfs-3dseg_gws_amd/`` (put it on sys.path and
the reference's own import paths work unchanged):

    model/dgcnn.py model/capl.py model/attention.py   drop-in module surface
    runs/eval.py  get_basis.py  util/checkpoint_util.py
    gfs3d/        ctypes binding + autograd operators + data-parallel helpers
    csrc/         HIP kernels and the C ABI (include/gfs3d.h at the repo root)
"""
from ._lib import build_library, library_path  # noqa: F401
