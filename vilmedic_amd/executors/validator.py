"""Validator -- ref: vilmedic/executors/validator.py:52-114 (eval_func dispatch under no_grad; metrics reduced to the
loss and the decode outputs -- the reference's text scorers are CPU metrics over absent packages)."""
import numpy as np
import torch

from .utils import create_data_loader, get_eval_func


class Validator(object):
    def __init__(self, config, models, train_dl, seed, from_training, logger, rank=0, world=1):
        self.config, self.models, self.seed, self.from_training, self.logger = config, models, seed, from_training, logger
        self.epoch = 0
        self.splits = [(s, create_data_loader(config, s, logger, called_by_validator=True, rank=rank, world=world))
                       for s in (config.get("splits") or ["validate"])]
        self.scores = []
        self.world, self.rank = world, rank

    def _dist(self):
        from ..parallel import active
        return active()

    def start(self):
        models = [m.eval() for m in self.models]
        eval_func = get_eval_func(models)
        self.scores = []
        with torch.no_grad():
            for split, dl in self.splits:
                self.logger.info("Running split: {} by ensembling {} models. Using {}.".format(split, len(models), eval_func.__name__))
                results = eval_func(models, self.config, dl, from_training=self.from_training)
                dist = self._dist()
                if dist is not None:       # every rank evaluated its shard: merge, so that all ranks score the SAME full split
                    from ..parallel import gather_interleaved, mean_over_ranks
                    results = dict(results)
                    if "loss" in results:
                        results["loss"] = mean_over_ranks(float(results["loss"]), dist, weight=max(1, len(dl.dataset)))
                    if isinstance(results.get("hyps"), list) and isinstance(results.get("refs"), list):
                        results["refs"], results["hyps"] = gather_interleaved(results["refs"], dist), gather_interleaved(results["hyps"], dist)
                    elif isinstance(results.get("hyps"), np.ndarray) and isinstance(results.get("refs"), np.ndarray):   # classifier outputs
                        results["refs"] = np.stack(gather_interleaved(list(results["refs"]), dist))
                        results["hyps"] = np.stack(gather_interleaved(list(results["hyps"]), dist))
                scores = {}
                if "loss" in results:
                    scores["validation_loss"] = float(results["loss"])
                metrics = self.config.get("metrics")
                if metrics and "refs" in results and "hyps" in results:     # the config's metric list (scorers/scores.py:34-151)
                    from ..blocks.scorers import compute_scores
                    scores.update(compute_scores(list(metrics), results["refs"], results["hyps"], split, self.seed,
                                                 self.config.get("ckpt_dir"), self.epoch, self.logger, dump=self.rank == 0))
                elif "refs" in results and "hyps" in results and isinstance(results["hyps"], list):
                    from ..blocks.scorers import RougeL
                    scores["ROUGEL"] = RougeL()(results["refs"], results["hyps"])[0]
                    scores["n_hyps"] = len(results["hyps"])
                self.scores.append(scores)
                self.logger.info(str({k: round(v, 4) if isinstance(v, float) else v for k, v in scores.items()}))
        return self.scores
