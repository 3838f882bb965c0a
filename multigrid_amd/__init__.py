"""MI355X-native 2-D Poisson multigrid V-cycle.

Layout of the package:
  multigrid_module    the eight kernel functions of the reference's `multigrid_module`, on the GPU
  parallel_multigrid  the reference's `Multigrid` / `split` / `gather_blocks` / `residual` surface
  engine              the row-slab V-cycle engine the above are built on (1..N GPUs)
  levels              level shapes and slab partition arithmetic (host only)
  csrc/               hand-written HIP kernels + the C ABI (include/mg_hip.h), built to libmg_hip.so
"""
from ._lib import MgError  # noqa: F401

__all__ = ["MgError"]
