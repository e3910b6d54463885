"""reference: models/classifier.py:6-29 -- 3 FC layers on a [B, emb] vector (config 1 plumbing; tiny,
stays torch/rocBLAS).  The max-pool in front of them (:23) is taken inside the feature model's last conv kernel when the model
offers forward_pooled (PointNet, DGCNN at inference)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .pooling import Pooling


class Classifier(nn.Module):
    def __init__(self, feature_model, num_classes=40):
        super(Classifier, self).__init__()
        self.feature_model = feature_model
        self.num_classes = num_classes
        self.linear1 = torch.nn.Linear(self.feature_model.emb_dims, 512)
        self.bn1 = torch.nn.BatchNorm1d(512)
        self.dropout1 = torch.nn.Dropout(p=0.7)
        self.linear2 = torch.nn.Linear(512, 256)
        self.bn2 = torch.nn.BatchNorm1d(256)
        self.dropout2 = torch.nn.Dropout(p=0.7)
        self.linear3 = torch.nn.Linear(256, self.num_classes)
        self.pooling = Pooling('max')

    def forward(self, input_data):
        output = None
        pooled = getattr(self.feature_model, "forward_pooled", None)
        if pooled is not None and self.pooling.pool_type == 'max':
            output = pooled(input_data)              # max-pool in the last conv's epilogue: no [B,emb,N] feature map
        if output is None:
            output = self.pooling(self.feature_model(input_data))
        output = F.relu(self.bn1(self.linear1(output)))
        output = self.dropout1(output)
        output = F.relu(self.bn2(self.linear2(output)))
        output = self.dropout2(output)
        output = self.linear3(output)
        return output
