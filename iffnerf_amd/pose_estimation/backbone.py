"""Image backbone factory with the reference's name (pose_estimation/backbone.py:3-14).

DINOv2 ViT-S/14 is a third-party model outside the accelerated path; it stays the stock
PyTorch-ROCm module.  ``torch.hub`` needs network access or a warm hub cache.
"""
import torch


def create_backbone(type="dino", pretrained=False, filter_size=4, pool_only=True, _force_nonfinetuned=False, **kwargs):
    if type != "dino":
        raise RuntimeError("only the 'dino' backbone exists in the reference (backbone.py:11-14)")
    model = torch.hub.load("facebookresearch/dinov2", "dinov2_vits14")
    return model, (16, 16), 384
