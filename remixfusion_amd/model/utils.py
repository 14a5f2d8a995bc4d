"""Loss helpers of the mapping path (reference model/utils.py:149-256), device-agnostic torch."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def batchify(fn, chunk=1024 * 64):
    """Apply ``fn`` in chunks along dim 0 (reference model/utils.py:149-167)."""
    if chunk is None:
        return fn

    def ret(inputs):
        return torch.cat([fn(inputs[i:i + chunk]) for i in range(0, inputs.shape[0], chunk)], 0)

    return ret


def compute_loss(prediction, target, loss_type="l2"):
    if loss_type == "l2":
        return F.mse_loss(prediction, target)
    if loss_type == "l1":
        return F.l1_loss(prediction, target)
    raise Exception("Unsupported loss type")


def get_masks(z_vals, target_d, truncation):
    """front / sdf masks and their balancing weights (reference :170-198)."""
    one, zero = torch.ones_like(z_vals), torch.zeros_like(z_vals)
    front_mask = torch.where(z_vals < (target_d - truncation), one, zero)
    back_mask = torch.where(z_vals > (target_d + truncation), one, zero)
    depth_mask = torch.where(target_d > 0.0, torch.ones_like(target_d), torch.zeros_like(target_d))
    sdf_mask = (1.0 - front_mask) * (1.0 - back_mask) * depth_mask
    num_fs = torch.count_nonzero(front_mask)
    num_sdf = torch.count_nonzero(sdf_mask)
    total = num_sdf + num_fs
    return front_mask, sdf_mask, 1.0 - num_fs / total, 1.0 - num_sdf / total


def get_sdf_loss(z_vals, target_d, predicted_sdf, truncation, loss_type=None, grad=None, middle_mask=None):
    """free-space + sdf losses (reference :219-256); the weights are computed before ``middle_mask``."""
    front_mask, sdf_mask, fs_weight, sdf_weight = get_masks(z_vals, target_d, truncation)
    if middle_mask is not None:
        front_mask = front_mask * middle_mask[..., None]
        sdf_mask = sdf_mask * middle_mask[..., None]
    fs_loss = compute_loss(predicted_sdf * front_mask, torch.ones_like(predicted_sdf) * front_mask, loss_type) * fs_weight
    sdf_loss = compute_loss((z_vals + predicted_sdf * truncation) * sdf_mask, target_d * sdf_mask, loss_type) * sdf_weight
    if grad is not None:
        eikonal_loss = (((grad.norm(2, dim=-1) - 1) ** 2) * sdf_mask / sdf_mask.sum()).sum()
        return fs_loss, sdf_loss, eikonal_loss
    return fs_loss, sdf_loss


def mse2psnr(x):
    return -10.0 * torch.log(x) / torch.log(torch.tensor(10.0, device=x.device))
