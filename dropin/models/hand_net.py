"""Drop-in alias: `from models.hand_net import ...` resolves to the MI355X implementation."""
from scat_amd.models.hand_net import *  # noqa: F401,F403
from scat_amd.models import hand_net as _m

globals().update({k: v for k, v in vars(_m).items() if not k.startswith("__")})
