"""MI355X-native meshless multigrid Poisson V-cycle (hot path of
michaelxu3/MeshlessMultigridPoisson): gfx950 HIP kernels behind a C-ABI
(include/mmgp.h, libmmgp.so) plus a host C++ mirror of the reference's
Grid/Multigrid classes (libmmgp_host.so).  The Python layer is ctypes plumbing
for tests and bench.py; it holds no compute path."""
from . import _capi  # noqa: F401

__all__ = ["_capi"]
