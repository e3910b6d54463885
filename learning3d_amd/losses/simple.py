"""The four small torch-only losses of learning3d/losses (rmse_features.py, frobenius_norm.py,
classification.py, correspondence_loss.py), restated so `learning3d_amd.losses` is a complete drop-in
for `learning3d.losses`.  Nothing here is on the accelerated path."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class RMSEFeaturesLoss(nn.Module):
    """sum of squared feature differences (reference: mse_loss against zeros, size_average=False)"""

    def forward(self, feature_difference):
        return (feature_difference * feature_difference).sum()


class FrobeniusNormLoss(nn.Module):
    """16 * mean((predicted @ igt - I)^2) over [B,4,4] transforms (reference: losses/frobenius_norm.py:5-14)"""

    def forward(self, predicted, igt):
        if predicted.shape != igt.shape or predicted.shape[-2:] != (4, 4):
            raise AssertionError("predicted and igt must both be [B,4,4]")
        err = predicted.matmul(igt) - torch.eye(4, dtype=predicted.dtype, device=predicted.device)
        return (err * err).mean() * 16


class ClassificationLoss(nn.Module):
    """negative log-likelihood of log-probabilities (reference: losses/classification.py:5-6)"""

    def forward(self, prediction, target):
        return F.nll_loss(prediction, target)


class CorrespondenceLoss(nn.Module):
    """cross entropy of the predicted correspondence rows [B, Ns, Nt] against the arg-max column of the
    ground-truth matrix [B, Nt, Ns] (reference: losses/correspondence_loss.py:4-10)"""

    def forward(self, template, source, corr_mat_pred, corr_mat):
        B, _, n_template = template.shape
        n_source = source.shape[2]
        labels = corr_mat.transpose(1, 2).reshape(-1, n_template).argmax(dim=1)
        return F.cross_entropy(corr_mat_pred.reshape(B * n_source, n_template), labels)
