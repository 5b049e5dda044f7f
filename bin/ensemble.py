#!/usr/bin/env python3
"""n-best checkpoint ensembling with the reference's command line (ref: bin/ensemble.py:13-84):

    python bin/ensemble.py config/RRG/rrg-vit-synthetic.yml ensemblor.mode=best-3 [ensemblor.ckpt=path.pth] ...

``ensemblor.mode``: ``best-N`` keeps the N best ``*.pth`` of ``ckpt_dir`` (names sort by score, bin/ensemble.py:22-33);
every kept checkpoint becomes one model and the Validator's decode driver sums the models' next-token logits before
the log-softmax (vilmedic_amd.generation.EnsembleState; ref: blocks/huggingface/decoder/beam_search.py:243-262)."""
import glob
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vilmedic_amd.config import executor_view, get_config  # noqa: E402
from vilmedic_amd.executors import Validator  # noqa: E402
from vilmedic_amd.executors.utils import create_model, get_logger  # noqa: E402


def get_n_best(mode):
    return int(mode.split("-")[-1]) if "-" in mode else 1


def get_ckpts(path, mode):
    ckpts = sorted(glob.glob(path), reverse=True)
    return ckpts[:get_n_best(mode)] if "best" in mode else ckpts


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    config = get_config(sys.argv[1], sys.argv[2:])
    ecfg = executor_view(config, "ensemblor")
    logger = get_logger()
    ckpt_dir = ecfg.get("ckpt_dir") or config.get("ckpt_dir") or "ckpt"
    mode = str(ecfg.get("mode") or "best-1")
    evaluator = Validator(config=ecfg, models=None, train_dl=None, seed="{}_{}".format(mode, int(config.get("seed") or 0)),
                          from_training=False, logger=logger)
    ckpts = get_ckpts(os.path.join(ckpt_dir, "*.pth"), mode)
    if ecfg.get("ckpt") is not None:
        ck = ecfg.ckpt if os.path.isfile(ecfg.ckpt) else os.path.join(ckpt_dir, ecfg.ckpt)
        assert os.path.isfile(ck), "Specified checkpoint does not exist"
        ckpts = [ck]
    if not ckpts:
        logger.settings("No checkpoints found")
        sys.exit()
    logger.settings("Checkpoints are {}".format("\n".join(ckpts)))
    evaluator.models = [create_model(config=ecfg, dl=evaluator.splits[0][1], logger=logger, from_training=False,
                                     state_dict=torch.load(c, map_location="cpu")).cuda().eval() for c in ckpts]
    evaluator.start()


if __name__ == "__main__":
    main()
