"""Drop-in for learning3d/models/pointnet.py on MI355X (reference: models/pointnet.py:7-73).
The 5 x Conv1d(k=1)(+BN)+ReLU stack runs as 5 fp32-MFMA GEMM launches with BN/bias/ReLU folded into
the epilogue (inference); parameter names match the reference so its checkpoints load unchanged."""
import torch

from . import _fused
from .pooling import Pooling


class PointNet(torch.nn.Module):
    def __init__(self, emb_dims=1024, input_shape="bnc", use_bn=False, global_feat=True):
        super(PointNet, self).__init__()
        if input_shape not in ["bcn", "bnc"]:
            raise ValueError("Allowed shapes are 'bcn' (batch * channels * num_in_points), 'bnc' ")
        self.input_shape = input_shape
        self.emb_dims = emb_dims
        self.use_bn = use_bn
        self.global_feat = global_feat
        if not self.global_feat:
            self.pooling = Pooling('max')
        self.layers = self.create_structure()

    def create_structure(self):
        self.conv1 = torch.nn.Conv1d(3, 64, 1)
        self.conv2 = torch.nn.Conv1d(64, 64, 1)
        self.conv3 = torch.nn.Conv1d(64, 64, 1)
        self.conv4 = torch.nn.Conv1d(64, 128, 1)
        self.conv5 = torch.nn.Conv1d(128, self.emb_dims, 1)
        self.relu = torch.nn.ReLU()
        convs = [self.conv1, self.conv2, self.conv3, self.conv4, self.conv5]
        if self.use_bn:
            self.bn1 = torch.nn.BatchNorm1d(64)
            self.bn2 = torch.nn.BatchNorm1d(64)
            self.bn3 = torch.nn.BatchNorm1d(64)
            self.bn4 = torch.nn.BatchNorm1d(128)
            self.bn5 = torch.nn.BatchNorm1d(self.emb_dims)
            bns = [self.bn1, self.bn2, self.bn3, self.bn4, self.bn5]
            layers = []
            for c, b in zip(convs, bns):
                layers += [c, b, self.relu]
        else:
            layers = []
            for c in convs:
                layers += [c, self.relu]
        return layers

    def _stack(self):
        convs = [self.conv1, self.conv2, self.conv3, self.conv4, self.conv5]
        bns = [self.bn1, self.bn2, self.bn3, self.bn4, self.bn5] if self.use_bn else [None] * 5
        return list(zip(convs, bns))

    def forward_pooled(self, input_data):
        """max over the points of forward()'s [B,emb,N] output -> [B,emb] (what models/classifier.py:23 computes next): conv5's
        kernel takes partial maxima in its epilogue, the feature map is never written.  None when this route does not
        apply (batch statistics, global_feat=False); the caller then pools forward()'s output."""
        if not self.global_feat or not input_data.is_cuda or _fused._stochastic_or_batch_dependent(self):
            return None
        return _fused.checkpointed(self, self._forward_pooled, input_data)

    def forward(self, input_data):
        return _fused.checkpointed(self, self._forward, input_data)

    def _forward_pooled(self, input_data):
        if not _fused.can_fuse(self, input_data):
            return self._forward(input_data).max(dim=2)[0]
        channel_last = self.input_shape == "bnc"
        if input_data.shape[2 if channel_last else 1] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")
        x = input_data
        stack = self._stack()
        for i, (conv, bn) in enumerate(stack):
            w, sc, sh = _fused.fold_conv_bn(conv, bn)
            if i == len(stack) - 1:
                return _fused.conv_global_max(x, w, sc, sh, True)
            x = _fused.pointwise_conv(x, w, sc, sh, relu=True, channel_last=(channel_last and i == 0))

    def _forward(self, input_data):
        if self.input_shape == "bnc":
            num_points = input_data.shape[1]
            input_data = input_data.permute(0, 2, 1)
        else:
            num_points = input_data.shape[2]
        if input_data.shape[1] != 3:
            raise RuntimeError("shape of x must be of [Batch x 3 x NumInPoints]")

        output = input_data
        if _fused.can_fuse(self, input_data) and input_data.is_cuda:
            # "bnc" input is consumed channel-last directly (no transpose copy)
            channel_last = self.input_shape == "bnc"
            x = input_data.permute(0, 2, 1) if channel_last else input_data
            for i, (conv, bn) in enumerate(self._stack()):
                w, sc, sh = _fused.fold_conv_bn(conv, bn)
                if i == 0 and not self.global_feat and self.use_bn:
                    # the reference taps layers[1]: bn1's output BEFORE the ReLU (pointnet.py:66)
                    point_feature = _fused.pointwise_conv(x, w, sc, sh, relu=False, channel_last=channel_last)
                    x = torch.relu(point_feature)
                    continue
                x = _fused.pointwise_conv(x, w, sc, sh, relu=True, channel_last=(channel_last and i == 0))
                if i == 0 and not self.global_feat:
                    point_feature = x
            output = x
        else:
            from ._train import conv_bn_act, hip_layers_ok
            if hip_layers_ok(input_data):
                # autograd is live: every layer on the HIP conv / dgrad / wgrad kernels (_train.py), BatchNorm with batch
                # statistics in train mode and running statistics in eval mode
                for i, (conv, bn) in enumerate(self._stack()):
                    if i == 0 and not self.global_feat and self.use_bn:
                        point_feature = conv_bn_act(output, conv, bn, relu=False)       # layers[1] = bn1 (pointnet.py:66)
                        output = torch.relu(point_feature)
                        continue
                    output = conv_bn_act(output, conv, bn, relu=True)
                    if i == 0 and not self.global_feat:
                        point_feature = output                                          # layers[1] = the ReLU
            else:
                for idx, layer in enumerate(self.layers):
                    output = layer(output)
                    if idx == 1 and not self.global_feat:
                        point_feature = output

        if self.global_feat:
            return output
        else:
            output = self.pooling(output)
            output = output.view(-1, self.emb_dims, 1).repeat(1, 1, num_points)
            return torch.cat([output, point_feature], 1)
