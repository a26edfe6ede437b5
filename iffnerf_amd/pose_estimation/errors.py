"""Pose error metrics with the reference's names (pose_estimation/errors.py:3-9); tiny host-side torch ops."""
import torch


def compute_translation_error(translation1, translation2):
    return (translation1 - translation2).norm()


def compute_angular_error(rotation_gt, rotation_est):
    rel = rotation_gt @ torch.linalg.inv(rotation_est)
    cos = ((rel.diagonal().sum() - 1) / 2).clamp(-1, 1)
    return torch.rad2deg(torch.acos(cos))
