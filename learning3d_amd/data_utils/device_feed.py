"""Synthetic registration batches generated entirely on the device.

reference: data_utils/dataloaders.py:250-330 (RegistrationData: template from the dataset, source = transform(template),
igt from the transform) with ops/transform_functions.py:271-315 as the transform; one sample at a time on host cores.
RegistrationFeed yields whole batches {template, source, igt} without a host tensor or a copy: clouds from
l3d_uniform_clouds (seeded by (seed, batch index, element)), rigid transforms from DCPTransform (l3d_euler_transform).
"""
import torch

from .._lib import check, lib, ptr, stream_ptr
from ..ops.transform_functions import DCPTransform


def uniform_clouds(batch, num_points, lo=0.0, hi=1.0, seed=0, device="cuda"):
    """[batch, num_points, 3] ~ U(lo, hi) generated on `device` (reproducible in (seed, shape))."""
    out = torch.empty((batch, num_points, 3), dtype=torch.float32, device=device)
    with torch.cuda.device(out.device):
        check(lib().l3d_uniform_clouds(int(seed), batch, num_points, float(lo), float(hi), ptr(out), stream_ptr()), "l3d_uniform_clouds")
    return out


class RegistrationFeed:
    """Iterator of device batches for DCP-style registration: (template [B,N,3], source [B,N,3], igt [B,4,4])."""

    def __init__(self, batch_size, num_points=1024, angle_range=45, translation_range=1, lo=-0.5, hi=0.5, seed=0,
                 device="cuda", length=None):
        self.batch_size, self.num_points = batch_size, num_points
        self.lo, self.hi, self.seed, self.device, self.length = lo, hi, seed, torch.device(device), length
        gen = torch.Generator(device=self.device)
        gen.manual_seed(seed)
        self.transform = DCPTransform(angle_range, translation_range, generator=gen)
        self._i = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.length is not None and self._i >= self.length:
            raise StopIteration
        template = uniform_clouds(self.batch_size, self.num_points, self.lo, self.hi,
                                  seed=(self.seed << 20) + self._i, device=self.device)
        source = self.transform(template)
        self._i += 1
        return template, source, self.transform.igt
