"""Mirror of learning3d/utils/__init__.py:1-23 for the hot-path symbols."""
from .svd import SVDHead, kabsch, svd3x3_rotation
from .transformer import Transformer, Identity
from .model_common_utils import (
    knn,
    get_graph_feature,
    square_distance,
    index_points,
    farthest_point_sample,
    knn_point,
    query_ball_point,
)
from .ppfnet_util import angle_difference, pc_normalize, sample_and_group, sample_and_group_multi
from .pointconv_util import PointConvDensitySetAbstraction
from . import pointconv_util, pointnet2_utils
