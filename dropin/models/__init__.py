"""Put this directory on PYTHONPATH in front of the reference checkout: train.py / train_coarse.py / eval.py then
import the MI355X `models` package unchanged.

Only the modules of the hot path are mirrored here (hand_net, resnet, hrnet, vit, vision_transformer,
vision_transformer_attn, vision_performer).  Everything else the reference keeps under ``models/`` — ``mano``,
``inception``, ``loss``, ``motion_discriminator``, ``helper`` (out of scope) — must keep resolving to the
reference checkout, e.g. ``eval.py:25 from models.mano import ManoHand``: the package path is extended with every
``models`` directory found later on ``sys.path``, this one first."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
