"""A CPU stand-in for the reference's `pointnet2_cuda` extension module, used ONLY by make_golden.py.

The reference's FlowNet3D (models/flownet3d.py) needs `utils/lib/pointnet2_utils.py`, which imports the compiled
`pointnet2_cuda` module (utils/lib/src/pointnet2_api.cpp:10-25) -- THC-era code that does not build against torch 2.x
(SURVEY.md 8(c)).  Its kernels are plain CUDA C, though, and oracle/oracle.c restates each of them (K7-K16); those
restatements are pinned bit-for-bit against the reference's own kernels compiled for gfx950
(tests/test_gpu_ref_kernels.py, oracle/_ref/libref_pointnet2.so).  This module exposes the restatements under the ten
wrapper names and calling conventions of pointnet2_api.cpp (explicit dims, caller-allocated outputs written in place), so
that the reference's OWN pointnet2_utils.py and models/flownet3d.py run unmodified on CPU tensors and produce the FlowNet3D
golden.  make_golden.py also points torch.cuda.IntTensor / FloatTensor (which pointnet2_utils.py allocates with,
utils/lib/pointnet2_utils.py:25-28 etc.) at their CPU twins for the duration."""
import ctypes as C
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import oracle  # noqa: E402


def _p(t):
    assert t.is_contiguous() and not t.is_cuda
    return C.c_void_p(t.data_ptr())


def make_module():
    L = oracle.lib()
    m = types.ModuleType("pointnet2_cuda")

    def ball_query_wrapper(b, n, m_, radius, nsample, new_xyz, xyz, idx):
        L.orc_ball_query(b, n, m_, C.c_float(radius), nsample, _p(new_xyz), _p(xyz), _p(idx))
        return 1

    def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
        L.orc_group_points(b, c, n, npoints, nsample, _p(points), _p(idx), _p(out))
        return 1

    def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
        L.orc_group_points_grad(b, c, n, npoints, nsample, _p(grad_out), _p(idx), _p(grad_points))
        return 1

    def gather_points_wrapper(b, c, n, npoints, points, idx, out):
        L.orc_gather_points(b, c, n, npoints, _p(points), _p(idx), _p(out))
        return 1

    def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
        L.orc_gather_points_grad(b, c, n, npoints, _p(grad_out), _p(idx), _p(grad_points))
        return 1

    def furthest_point_sampling_wrapper(b, n, m_, points, temp, idx):
        L.orc_furthest_point_sampling(b, n, m_, _p(points), _p(temp), _p(idx))
        return 1

    def knn_wrapper(b, n, m_, k, unknown, known, dist2, idx):
        L.orc_knn_pair(b, n, m_, k, _p(unknown), _p(known), _p(dist2), _p(idx))

    def three_nn_wrapper(b, n, m_, unknown, known, dist2, idx):
        L.orc_three_nn(b, n, m_, _p(unknown), _p(known), _p(dist2), _p(idx))

    def three_interpolate_wrapper(b, c, m_, n, points, idx, weight, out):
        L.orc_three_interpolate(b, c, m_, n, _p(points), _p(idx), _p(weight), _p(out))

    def three_interpolate_grad_wrapper(b, c, n, m_, grad_out, idx, weight, grad_points):
        L.orc_three_interpolate_grad(b, c, n, m_, _p(grad_out), _p(idx), _p(weight), _p(grad_points))

    for f in (ball_query_wrapper, group_points_wrapper, group_points_grad_wrapper, gather_points_wrapper,
              gather_points_grad_wrapper, furthest_point_sampling_wrapper, knn_wrapper, three_nn_wrapper,
              three_interpolate_wrapper, three_interpolate_grad_wrapper):
        setattr(m, f.__name__, f)
    return m
