"""Drop-in alias: ``import vilmedic`` resolves to the MI355X implementation (``vilmedic_amd``) under the reference's
module paths (``vilmedic.blocks.vision``, ``vilmedic.models``, ``vilmedic.executors`` ...), so code written against
jbdel/vilmedic's plugin surface (class lookup by name) imports unchanged."""
import importlib
import sys

import vilmedic_amd

__version__ = "1.3.6"
_ALIASES = [
    "blocks", "blocks.vision", "blocks.vision.visual_encoder", "blocks.huggingface", "blocks.huggingface.decoder",
    "blocks.huggingface.decoder.decoder_model", "blocks.huggingface.decoder.evaluation", "blocks.huggingface.encoder",
    "blocks.huggingface.encoder.encoder_model", "blocks.losses", "blocks.classifier", "blocks.classifier.evaluation",
    "blocks.rl", "blocks.rl.SCST", "blocks.scorers", "models", "models.utils", "models.rrg.RRG", "models.rrg.RRG_SCST",
    "models.selfsup.conVIRT", "models.mvqa.MVQA", "executors", "executors.utils", "datasets",
    "models.rrg.RRG_HF", "models.rrs.RRS", "models.selfsup.GLoRIA", "zoo", "zoo.modeling_auto", "blocks.schedulers",
]
for _name in _ALIASES:
    sys.modules["vilmedic." + _name] = importlib.import_module("vilmedic_amd." + _name)
blocks, models, executors, datasets = (sys.modules["vilmedic." + n] for n in ("blocks", "models", "executors", "datasets"))
