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


def _on_gpu(t):
    """The reference's Dataset.__getitem__ hands these transforms CPU tensors [N,3] (data_utils/dataloaders.py:290-296): they are
    moved to the current GPU, transformed there, and the results handed back on the CPU; device tensors stay where they are.
    From a fork-started DataLoader worker the device cannot be opened: that raises with the ways out (below)."""
    if t.is_cuda:
        return t, False
    try:
        return t.cuda(), True
    except RuntimeError as exc:
        if "forked subprocess" in str(exc):
            # a DataLoader worker started by fork (num_workers > 0 with the default start method) cannot open the device, and
            # this package has no CPU implementation to fall back to (by design: nothing on the path computes on the host)
            raise RuntimeError(
                "learning3d_amd transforms run on the GPU: call them from the main process (DataLoader num_workers=0), start the "
                "workers with multiprocessing_context='spawn', or -- what they are built for -- hand them whole device batches "
                "(data_utils.device_feed.RegistrationFeed / ResidentRegistrationFeed)") from exc
        raise


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
        template, back = _on_gpu(template)
        single = template.dim() == 2
        t = template.unsqueeze(0) if single else template
        self.generate_transform(t.shape[0], t.device)
        source = self.apply_transformation(t)
        if single:
            self.igt = self.igt[0]
            source = source[0]
        if back:
            self.igt = self.igt.cpu()
            return source.cpu()
        return source


class DeepGMRTransform(DCPTransform):
    """reference: ops/transform_functions.py:317-345 -- the same draw and the same application as DCPTransform."""


def twist_transform(template, twist):
    """template [B,N,3] (device), twist [B,6] = (w, v) -> (source [B,N,3], igt [B,4,4] = se3.exp(twist), gt [B,4,4] =
    se3.exp(-twist)): ops/transform_functions.py:133-141 for a whole batch in one launch (l3d_twist_transform)."""
    require_gpu(template, twist)
    t, x = f32c(template), f32c(twist)
    B, N, _ = t.shape
    source = torch.empty_like(t)
    igt = torch.empty((B, 4, 4), dtype=torch.float32, device=t.device)
    gt = torch.empty((B, 4, 4), dtype=torch.float32, device=t.device)
    check(lib().l3d_twist_transform(ptr(t), ptr(x), B, N, ptr(source), ptr(igt), ptr(gt), stream_ptr()), "l3d_twist_transform")
    return source, igt, gt


def quat_transform(template, pose7):
    """template [B,N,3], pose7 [B,7] = (quaternion w x y z -- normalised here as create_pose_7d does --, translation)
    -> source = qrot(q, template) + t: PCRNetTransform.__call__ (ops/transform_functions.py:265-269) for a batch."""
    require_gpu(template, pose7)
    t, p = f32c(template), f32c(pose7)
    B, N, _ = t.shape
    source = torch.empty_like(t)
    check(lib().l3d_quat_transform(ptr(t), ptr(p), B, N, ptr(source), stream_ptr()), "l3d_quat_transform")
    return source


class PNLKTransform:
    """reference: ops/transform_functions.py:109-145.  Same constructor; `__call__(tensor)` takes a device batch [B,N,3]
    (or one cloud [N,3]); `.gt` / `.igt` hold [B,4,4] (or [4,4]) afterwards.  The twist is drawn on the device: a unit
    6-vector times `mag` (times U(0,1) with mag_randomly), generate_transform :119-127."""

    def __init__(self, mag=1, mag_randomly=False, generator=None):
        self.mag = mag
        self.randomly = mag_randomly
        self.gt = None
        self.igt = None
        self.index = 0
        self.generator = generator

    def generate_transform(self, batch=1, device="cuda"):
        amp = self.mag
        if self.randomly:
            amp = torch.rand((batch, 1), device=device, generator=self.generator) * self.mag
        x = torch.randn((batch, 6), device=device, generator=self.generator)
        return x / x.norm(p=2, dim=1, keepdim=True) * amp

    def apply_transform(self, p0, x):
        single = p0.dim() == 2
        p = p0.unsqueeze(0) if single else p0
        p1, igt, gt = twist_transform(p[..., :3].contiguous(), x.reshape(-1, 6))
        if p.shape[-1] == 6:                       # RPMNetTransform: the normals turn with the rotation only (:170-174)
            xr = x.reshape(-1, 6).clone()
            xr[:, 3:] = 0
            n1, _, _ = twist_transform(p[..., 3:6].contiguous(), xr)
            p1 = torch.cat([p1, n1], dim=-1)
        self.gt, self.igt = (gt[0], igt[0]) if single else (gt, igt)
        return p1[0] if single else p1

    def transform(self, tensor):
        batch = 1 if tensor.dim() == 2 else tensor.shape[0]
        return self.apply_transform(tensor, self.generate_transform(batch, tensor.device))

    def __call__(self, tensor):
        tensor, back = _on_gpu(tensor)
        out = self.transform(tensor)
        if back:
            self.gt, self.igt = self.gt.cpu(), self.igt.cpu()
            return out.cpu()
        return out


class RPMNetTransform(PNLKTransform):
    """reference: ops/transform_functions.py:148-192 -- PNLKTransform that also rotates the normals of [.., 6] clouds."""


def _qmul(q, r):
    """Hamilton product of [B,4] quaternions (ops/transform_functions.py:37-55)."""
    w0, x0, y0, z0 = q.unbind(1)
    w1, x1, y1, z1 = r.unbind(1)
    return torch.stack([w0 * w1 - x0 * x1 - y0 * y1 - z0 * z1, w0 * x1 + x0 * w1 + y0 * z1 - z0 * y1,
                        w0 * y1 - x0 * z1 + y0 * w1 + z0 * x1, w0 * z1 + x0 * y1 - y0 * x1 + z0 * w1], dim=1)


def euler_to_quaternion_xyz(e):
    """[B,3] Euler angles -> [B,4] quaternions, order "xyz", with the reference's sign convention (:62-107)."""
    z0 = torch.zeros_like(e[:, 0])
    rx = torch.stack([torch.cos(e[:, 0] / 2), torch.sin(e[:, 0] / 2), z0, z0], dim=1)
    ry = torch.stack([torch.cos(e[:, 1] / 2), z0, torch.sin(e[:, 1] / 2), z0], dim=1)
    rz = torch.stack([torch.cos(e[:, 2] / 2), z0, z0, torch.sin(e[:, 2] / 2)], dim=1)
    return -_qmul(_qmul(rx, ry), rz)


class PCRNetTransform:
    """reference: ops/transform_functions.py:194-269.  The reference pre-draws `data_size` poses on the host and its dataset
    picks one by index; here `__call__(template [B,N,3] or [N,3])` draws one pose per cloud on the device (Euler angles
    ~ U(-angle_range, angle_range)^3 degrees -> quaternion "xyz", translation ~ U(-translation_range, translation_range)^3,
    create_random_transform :205-214) and applies it (l3d_quat_transform); `.igt` holds the [B,7] poses."""

    def __init__(self, data_size=None, angle_range=45, translation_range=1, generator=None):
        self.angle_range = angle_range
        self.translation_range = translation_range
        self.dtype = torch.float32
        self.index = 0
        self.generator = generator
        self.data_size = data_size
        self.transformations = None          # [data_size, 7], drawn on first use (the reference pre-draws them on the host, :200-203)

    @staticmethod
    def deg_to_rad(deg):
        return math.pi / 180 * deg

    def create_random_transform(self, batch, device):
        mr = self.deg_to_rad(self.angle_range)
        r = torch.rand((batch, 6), device=device, generator=self.generator) * 2 - 1
        return torch.cat([euler_to_quaternion_xyz(r[:, :3] * mr), r[:, 3:] * self.translation_range], dim=1)

    def __call__(self, template):
        template, back = _on_gpu(template)
        single = template.dim() == 2
        t = template.unsqueeze(0) if single else template
        if single and self.data_size:
            # the reference's dataset semantics (:216-218, dataloaders.py:291): a fixed pose per sample index
            if self.transformations is None:
                self.transformations = self.create_random_transform(int(self.data_size), t.device)
            self.igt = self.transformations[self.index % int(self.data_size)].unsqueeze(0)
        else:
            self.igt = self.create_random_transform(t.shape[0], t.device)
        source = quat_transform(t, self.igt)
        if single:
            self.igt = self.igt[0:1]
            source = source[0]
        if back:
            self.igt = self.igt.cpu()
            return source.cpu()
        return source
