"""Drop-in alias: `from models.vision_transformer import ...` resolves to the MI355X implementation."""
from scat_amd.models.vision_transformer import *  # noqa: F401,F403
from scat_amd.models import vision_transformer as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
