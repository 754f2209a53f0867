"""Host-side mirror of the reference's ``models`` package for the reg_transformer hot path."""
from . import hand_net, hrnet, resnet, vision_performer, vision_transformer, vision_transformer_attn, vit  # noqa: F401
