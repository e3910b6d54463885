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


class ResidentRegistrationFeed:
    """RegistrationData over a dataset that lives in HBM (reference: data_utils/dataloaders.py:184-227 ModelNet40Data --
    `data[idx][:num_points]`, optional per-sample point shuffle -- and :250-330 RegistrationData -- source =
    transform(template)).  ModelNet40's 9840 x 2048 x 3 training clouds are 242 MB: they fit once, and every batch is an
    index gather + one transform launch on the device -- no DataLoader workers, no host tensor, no copy per step.

    data [M,P,3] device tensor (labels [M] optional); `transform` is any of the device transforms of
    ops/transform_functions.py (DCPTransform by default).  Yields (template [B,N,3], source [B,N,3], igt, labels or None);
    one epoch = every cloud once, in a permutation seeded by (seed, epoch)."""

    def __init__(self, data, labels=None, batch_size=32, num_points=1024, transform=None, randomize_points=False, seed=0,
                 drop_last=True):
        if not (data.is_cuda and data.dim() == 3 and data.shape[2] == 3):
            raise ValueError("data must be a device tensor [M, P, 3]")
        if num_points > data.shape[1]:
            raise ValueError("num_points exceeds the points per cloud of the dataset")
        self.data, self.labels = data.float().contiguous(), labels
        self.batch_size, self.num_points, self.randomize_points, self.drop_last = batch_size, num_points, randomize_points, drop_last
        self.gen = torch.Generator(device=data.device)
        self.seed, self.epoch = seed, 0
        self.transform = transform if transform is not None else DCPTransform(45, 1, generator=self.gen)

    def __len__(self):
        M = self.data.shape[0]
        return M // self.batch_size if self.drop_last else -(-M // self.batch_size)

    def __iter__(self):
        self.gen.manual_seed((self.seed << 20) + self.epoch)
        self.epoch += 1
        perm = torch.randperm(self.data.shape[0], device=self.data.device, generator=self.gen)
        for i in range(len(self)):
            idx = perm[i * self.batch_size:(i + 1) * self.batch_size]
            if self.randomize_points:                          # ModelNet40Data.randomize (:214-216): shuffle, then keep the first N
                order = torch.rand((idx.numel(), self.data.shape[1]), device=self.data.device, generator=self.gen).argsort(dim=1)
                template = torch.gather(self.data[idx], 1, order[:, :self.num_points, None].expand(-1, -1, 3)).contiguous()
            else:
                template = self.data[idx, :self.num_points].contiguous()
            source = self.transform(template)
            yield template, source, self.transform.igt, (self.labels[idx] if self.labels is not None else None)
