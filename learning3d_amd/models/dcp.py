"""Drop-in for learning3d/models/dcp.py (reference: models/dcp.py:10-55), DCP-v2:
DGCNN embedding (HIP kNN + fused EdgeConv + MFMA conv5) x2 -> Transformer pointer (torch) ->
SVDHead (HIP Kabsch kernel, no per-item host syncs) -> rigid transform."""
import torch
import torch.nn as nn

from ..utils.svd import SVDHead
from ..utils.transformer import Identity, Transformer
from .dgcnn import DGCNN


def _mm(a, b):
    """a @ b for the head's small products: on the library's own batched GEMM (l3d_bmm_f32, differentiable) for fp32 device tensors
    -- no rocBLAS launch in the forward trace --, torch.matmul otherwise (CPU tensors, other dtypes)."""
    if (a.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.shape[:-2] == b.shape[:-2] and a.dim() == 3
            and a.shape[0] <= 65535):                  # l3d_bmm_f32's grid limit: larger batches fall through to torch.matmul
        from . import _rows
        return _rows.matmul(a, b)
    return torch.matmul(a, b)


def transform_point_cloud(point_cloud, rotation, translation):
    """ops/transform_functions.py:24-29 for rotation matrices."""
    return (_mm(rotation, point_cloud.permute(0, 2, 1)) + translation.unsqueeze(2)).permute(0, 2, 1)


def convert2transformation(rotation_matrix, translation_vector):
    """ops/transform_functions.py:31-35."""
    B = rotation_matrix.shape[0]
    bottom = torch.tensor([[[0.0, 0.0, 0.0, 1.0]]]).repeat(B, 1, 1).to(rotation_matrix)
    top = torch.cat([rotation_matrix, translation_vector.unsqueeze(-1)], dim=2)
    return torch.cat([top, bottom], dim=1)


class DCP(nn.Module):
    def __init__(self, feature_model=None, cycle=False, pointer_='transformer', head='svd'):
        super(DCP, self).__init__()
        self.cycle = cycle
        self.emb_nn = feature_model if feature_model is not None else DGCNN()
        if pointer_ == 'identity':
            self.pointer = Identity()
        elif pointer_ == 'transformer':
            self.pointer = Transformer(self.emb_nn.emb_dims, n_blocks=1, dropout=0.0, ff_dims=1024, n_heads=4)
        else:
            raise Exception("Not implemented")
        if head == 'svd':
            self.head = SVDHead(self.emb_nn.emb_dims)
        else:
            # the reference's MLPHead is dead code (undefined quat2mat, models/dcp.py:82)
            raise Exception('Not implemented')

    def forward(self, template, source):
        source_features = self.emb_nn(source)
        template_features = self.emb_nn(template)
        source_features_p, template_features_p = self.pointer(source_features, template_features)
        source_features = source_features + source_features_p
        template_features = template_features + template_features_p

        rotation_ab, translation_ab = self.head(source_features, template_features, source, template)
        if self.cycle:
            rotation_ba, translation_ba = self.head(template_features, source_features, template, source)
        else:
            rotation_ba = rotation_ab.transpose(2, 1).contiguous()
            translation_ba = -_mm(rotation_ba, translation_ab.unsqueeze(2)).squeeze(2)
        transformed_source = transform_point_cloud(source, rotation_ab, translation_ab)
        return {'est_R': rotation_ab, 'est_t': translation_ab, 'est_R_': rotation_ba, 'est_t_': translation_ba,
                'est_T': convert2transformation(rotation_ab, translation_ab),
                'r': template_features - source_features, 'transformed_source': transformed_source}
