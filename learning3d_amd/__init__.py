"""learning3d_amd -- learning3d's point-cloud hot path on AMD MI355X (gfx950).

Drop-in for `learning3d.utils`, `learning3d.losses` and `learning3d.models` on the path
pairwise distance -> kNN / ball query / grouping -> shared MLP -> Chamfer / EMD -> 3x3 SVD head.
Everything below the Python API is hand-written HIP behind the C ABI in include/l3d_hip.h
(libl3d_hip.so); see DESIGN.md.
"""
from . import _lib  # noqa: F401

__version__ = "0.1.0"
