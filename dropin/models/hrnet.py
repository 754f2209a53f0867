"""Drop-in alias: `from models.hrnet import ...` resolves to the MI355X implementation."""
from scat_amd.models.hrnet import *  # noqa: F401,F403
from scat_amd.models import hrnet as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
