"""Mirror of learning3d/ops for the pieces the hot path's callers use."""
from . import transform_functions
