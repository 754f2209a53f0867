"""Drop-in alias: `from models.resnet import ...` resolves to the MI355X implementation."""
from scat_amd.models.resnet import *  # noqa: F401,F403
from scat_amd.models import resnet as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
