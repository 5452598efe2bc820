"""EdgeLoss with the reference interface (reference src/models/loss.py:5-17)."""
import torch.nn as nn
import torch.nn.functional as F


class EdgeLoss(nn.Module):
    def __init__(self, loss_type="mse"):
        super().__init__()
        if loss_type == "mse":
            self.loss_func = F.mse_loss
        elif loss_type == "l1":
            self.loss_func = F.l1_loss
        else:
            raise ValueError(f"unknown loss_type {loss_type!r}")

    def forward(self, pred_edge, gt_edge):
        return self.loss_func(pred_edge, gt_edge, reduction="mean")
