"""Mirror of learning3d/losses/__init__.py:1-12."""
from .chamfer_distance import ChamferDistanceLoss, ChamferDistance, chamfer_distance
from .emd import EMDLoss
from .simple import RMSEFeaturesLoss, FrobeniusNormLoss, ClassificationLoss, CorrespondenceLoss
