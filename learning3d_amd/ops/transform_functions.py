"""Device-side mirror of the parts of learning3d/ops/transform_functions.py that sit on either side of the hot path.

reference: ops/transform_functions.py:24-35 (transform_point_cloud, convert2transformation) and :271-345 (DCPTransform,
DeepGMRTransform).  The reference transforms ONE template on the CPU per call (numpy draw + scipy Rotation per sample,
invoked from Dataset.__getitem__); here a transform object handles a whole device batch in one launch
(l3d_euler_transform) and draws its angles / translations on the device, so a registration step never touches the host.
"""
import math

import torch

from .._lib import check, f32c, lib, ptr, require_gpu, stream_ptr
from ..models.dcp import convert2transformation, transform_point_cloud  # noqa: F401  (same functions, one definition)


def euler_transform(template, euler_zyx, translation):
    """template [B,N,3] (device), euler_zyx [B,3] = (anglez, angley, anglex) in radians, translation [B,3]
    -> (source [B,N,3], igt [B,4,4]) with the reference's conventions (igt's 3x3 block is the transposed rotation)."""
    require_gpu(template, euler_zyx, translation)
    t, e, tr = f32c(template), f32c(euler_zyx), f32c(translation)
    B, N, _ = t.shape
    source = torch.empty_like(t)
    igt = torch.empty((B, 4, 4), dtype=torch.float32, device=t.device)
    check(lib().l3d_euler_transform(ptr(t), ptr(e), ptr(tr), B, N, ptr(source), ptr(igt), stream_ptr()), "l3d_euler_transform")
    return source, igt


class DCPTransform:
    """reference: ops/transform_functions.py:271-315.  Same constructor; `__call__(template)` takes a device batch
    [B,N,3] (or one cloud [N,3]) and returns the transformed source; `.igt` holds the batch's [B,4,4] ground truth
    afterwards (the reference's attribute, there 4x4 for its single cloud).  Angles ~ U(0, angle_range) per axis and
    translations ~ U(-translation_range, translation_range)^3 as in generate_transform (:277-283), drawn with a device
    torch.Generator when one is given (reproducible), else the device's default generator."""

    def __init__(self, angle_range=45, translation_range=1, generator=None):
        self.angle_range = angle_range * (math.pi / 180)
        self.translation_range = translation_range
        self.index = 0
        self.generator = generator

    def generate_transform(self, batch, device):
        r = torch.rand((batch, 6), device=device, generator=self.generator)
        self.anglex, self.angley, self.anglez = (r[:, 0] * self.angle_range, r[:, 1] * self.angle_range, r[:, 2] * self.angle_range)
        self.translation = (r[:, 3:6] * 2 - 1) * self.translation_range

    def apply_transformation(self, template):
        euler = torch.stack([self.anglez, self.angley, self.anglex], dim=1)
        source, self.igt = euler_transform(template, euler, self.translation)
        return source

    def __call__(self, template):
        single = template.dim() == 2
        t = template.unsqueeze(0) if single else template
        self.generate_transform(t.shape[0], t.device)
        source = self.apply_transformation(t)
        if single:
            self.igt = self.igt[0]
            return source[0]
        return source


class DeepGMRTransform(DCPTransform):
    """reference: ops/transform_functions.py:317-345 -- the same draw and the same application as DCPTransform."""
