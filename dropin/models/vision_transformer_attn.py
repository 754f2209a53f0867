"""Drop-in alias: `from models.vision_transformer_attn import ...` resolves to the MI355X implementation."""
from scat_amd.models.vision_transformer_attn import *  # noqa: F401,F403
from scat_amd.models import vision_transformer_attn as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
