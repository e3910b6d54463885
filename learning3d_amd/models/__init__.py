"""Mirror of learning3d/models/__init__.py:1-24 for the models on the accelerated hot path."""
from .pointnet import PointNet
from .dgcnn import DGCNN
from .pooling import Pooling
from .classifier import Classifier
from .pcn import PCN
from .dcp import DCP
from .flownet3d import FlowNet3D, PointNetSetAbstraction, FlowEmbedding, PointNetSetUpConv, PointNetFeaturePropogation
