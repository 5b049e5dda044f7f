"""ref: vilmedic/blocks/losses/__init__.py:1-6 -- also re-exports every torch.nn loss so YAML can name them."""
from torch.nn.modules.loss import *  # noqa: F401,F403

from .mvqa import LabelSmoothingCrossEntropy  # noqa: F401
from .selfsup import ConVIRTLoss, GLoRIALoss, InfoNCELoss, VICREGLoss, cosine_similarity, gloria_attention_fn  # noqa: F401
