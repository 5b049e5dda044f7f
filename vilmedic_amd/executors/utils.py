"""Factories and training-control objects.  ref: vilmedic/executors/utils.py:26-491 (semantics kept: class lookup by
``eval(proto)``, any ``torch.optim`` / ``lr_scheduler`` by name, early stopping, one-best checkpoint naming)."""
import copy
import inspect
import json
import logging
import operator
import os
import re
import sys

import numpy as np
import torch
import torch.nn as nn
from torch.optim import *  # noqa: F401,F403
from torch.optim.lr_scheduler import *  # noqa: F401,F403
from ..blocks.schedulers import DecreasingCosineAnnealingWarmRestarts, LinearWarmupCosineAnnealingLR  # noqa: F401  (eval(lr_decay) namespace)
from torch.utils.data import DataLoader
from torch.utils.data.sampler import BatchSampler, RandomSampler, SequentialSampler

from ..datasets import *  # noqa: F401,F403
from ..models import *  # noqa: F401,F403
from ..optim import FusedAdam  # noqa: F401

__version__ = "1.3.6"


def vilmedic_state_dict_versioning(params, version):
    params = {k.replace("module.", ""): v for k, v in params.items()}
    if version is None or version < "1.3.2":
        params = {k.replace("enc.0.cnn.", "enc.model."): v for k, v in params.items()}
        params = {k.replace("enc.1.weight", "enc.visual_projection.weight"): v for k, v in params.items()}
        params = {k.replace("enc.1.bias", "enc.visual_projection.bias"): v for k, v in params.items()}
    return params


def get_eval_func(models):
    dummy = models[0]
    assert hasattr(dummy, "eval_func")
    return dummy.eval_func


def get_logger(name="vilmedic_amd", path=None):
    logger = logging.getLogger(name)
    if not logger.handlers:
        logger.setLevel(logging.INFO)
        h = logging.StreamHandler(sys.stdout)
        h.setFormatter(logging.Formatter("%(asctime)s %(message)s", "%H:%M:%S"))
        logger.addHandler(h)
        if path:
            logger.addHandler(logging.FileHandler(path))
    if not hasattr(logger, "settings"):
        logger.settings = logger.info
    return logger


def create_model(config, dl, logger, from_training=True, state_dict=None):
    cfg = copy.deepcopy(config.model)
    if cfg.get("proto") is None:
        raise ValueError("config.model.proto is required")
    proto = cfg.pop("proto")
    model = eval(proto)(**cfg, dl=dl, logger=logger, from_training=from_training)
    logger.settings("Model {} created".format(type(model).__name__))
    if state_dict is not None:
        if "model" not in state_dict:
            logger.critical('This checkpoint is not valid. Key "model" is missing from dict.')
            sys.exit()
        model.load_state_dict(vilmedic_state_dict_versioning(state_dict["model"], state_dict.get("__version__", None)), strict=True)
        logger.info("Model state loaded")
    # one process per GPU: place on this rank's device; no nn.DataParallel (replaced by vilmedic_amd.parallel.ArenaDDP)
    return model.cuda()


def create_optimizer(config, logger, model, state_dict=None):
    if config.get("optim_params") is None or config.optim_params.get("lr") is None:
        raise ValueError("config.optim_params.lr is required")
    od = dict(config.optim_params)
    if isinstance(od.get("betas"), list):
        od["betas"] = tuple(od["betas"])
    name = config.get("optimizer")
    if name is None:
        raise ValueError("config.optimizer is required")
    if name == "FusedAdam":
        optimizer = FusedAdam(model, **od)
    elif hasattr(torch.optim, name):
        optimizer = getattr(torch.optim, name)(model.parameters(), **od)
    else:
        raise NotImplementedError(name)
    logger.settings("Optimizer {} created".format(type(optimizer).__name__))
    if state_dict is not None and "optimizer" in state_dict:
        optimizer.load_state_dict(state_dict["optimizer"])
        logger.info("Optimizer state loaded")
    return optimizer


def create_data_loader(config, split, logger, called_by_validator=False, rank=0, world=1):
    dcfg = copy.deepcopy(config.dataset)
    proto = dcfg.pop("proto")
    if proto in ("ImSeq", "ImLabel", "ImSeqLabel", "ImSeqAny", "Seq2Seq", "ImSeq2Seq"):
        dcfg.setdefault("ckpt_dir", config.get("ckpt_dir") or "ckpt")
        if called_by_validator and isinstance(dcfg.get("image"), dict) and split == "train":
            dcfg["image"]["called_by_ensemblor"] = True       # evaluation transform on the train split (ImageDataset.py:83-84)
    dataset = eval(proto)(split=split, **dcfg)
    if hasattr(dataset, "get_collate_fn"):
        collate = dataset.get_collate_fn()
    else:
        collate = torch.utils.data.dataloader.default_collate
    if world > 1:   # shard the samples by rank (what accelerator.prepare(dl) does, trainor_accelerate.py:91-93)
        # the TRAINING shards are cut to equal length (the remainder of N / world is dropped, like DistributedSampler(drop_last)):
        # one rank with an extra batch would wait alone in that batch's gradient all-reduce
        n = len(dataset) // world * world if (split == "train" and not called_by_validator) else len(dataset)
        dataset = torch.utils.data.Subset(dataset, list(range(rank, n, world)))
        for attr in ("tokenizer", "tokenizer_max_len", "seq", "src", "tgt", "tgt_tokenizer", "tgt_tokenizer_max_len", "labels_map"):
            if hasattr(dataset.dataset, attr):
                setattr(dataset, attr, getattr(dataset.dataset, attr))
    training = split == "train" and not called_by_validator
    workers = int(config.get("num_workers") or 0)
    # the training loader runs under a PrefetchLoader (datasets/prefetch.py): batches are collated into reusable pinned buffers
    # and copied to the GPU by a background thread, ahead of the step.  ``prefetch: 0`` in the config turns it off.
    prefetch = int(config.get("prefetch", 2) if config.get("prefetch", 2) is not None else 2) if training else 0
    gen = None
    if prefetch:
        # the shuffling order comes from a generator of its own, seeded here from the global RNG: the sampler then runs on the
        # prefetch thread without drawing from the global generator concurrently with the training step (reproducible runs)
        gen = torch.Generator()
        gen.manual_seed(int(torch.empty((), dtype=torch.int64).random_().item()))
    if training:
        sampler = BatchSampler(RandomSampler(dataset, generator=gen), batch_size=config.batch_size, drop_last=True)
    else:
        sampler = BatchSampler(SequentialSampler(dataset), batch_size=config.batch_size, drop_last=False)
    logger.settings("DataLoader {} ({}, {} samples) created".format(proto, split, len(dataset)))
    loader = DataLoader(dataset, num_workers=workers, collate_fn=collate, batch_sampler=sampler, generator=gen,
                        pin_memory=not hasattr(dataset, "device_transform") and not prefetch)
    if prefetch:
        from ..datasets import PrefetchLoader
        loader = PrefetchLoader(loader, depth=prefetch)
    base = dataset.dataset if isinstance(dataset, torch.utils.data.Subset) else dataset
    if hasattr(base, "device_transform"):          # decoded uint8 images -> the device-side Resize / crop / flip / normalise kernel
        from ..datasets import DeviceBatchLoader
        return DeviceBatchLoader(loader)
    return loader


class CheckpointSaver(object):
    """keeps exactly one best checkpoint named {tag}_{epoch}_{seed}.pth (utils.py:237-267)."""

    def __init__(self, ckpt_dir, logger, seed, ckpt=None):
        self.ckpt_dir, self.seed, self.logger = ckpt_dir, seed, logger
        self.current_tag = self.current_epoch = None
        if ckpt is not None:
            g = re.match(".*/(.*?)_(.*?)_(.*?).pth", ckpt)
            self.current_tag, self.current_epoch = float(g.group(1)), int(g.group(2))

    def save(self, state_dict, tag, current_epoch):
        if self.current_tag is not None:
            old = os.path.join(self.ckpt_dir, "{}_{}_{}.pth".format(self.current_tag, self.current_epoch, self.seed))
            if os.path.exists(old):
                os.remove(old)
        tag = np.round(tag, 6)
        path = os.path.join(self.ckpt_dir, "{}_{}_{}.pth".format(tag, current_epoch, self.seed))
        torch.save(state_dict, path)
        self.logger.info("{} saved.".format(path))
        self.current_tag, self.current_epoch = tag, current_epoch


class TrainingScheduler(object):
    """LR scheduling (per-iteration / per-epoch / on-validation, optional linear warm-up) + early stopping (utils.py:324-491)."""
    ITER_STEP = {"CyclicLR", "OneCycleLR", "CosineAnnealingWarmRestarts"}
    VAL_STEP = {"ReduceLROnPlateau"}

    def __init__(self, lr_decay_func, optimizer, early_stop_metric, early_stop_limit, lr_decay_params):
        self.epoch = self.iteration_count = self.early_stop = 0
        self.scheduler_name = lr_decay_func
        self.early_stop_limit, self.early_stop_metric = early_stop_limit, early_stop_metric
        if early_stop_metric in ("validation_loss", "training_loss"):
            self.metric_comp_func, self.mode, self.current_best_metric = operator.lt, "min", float("inf")
        else:
            self.metric_comp_func, self.mode, self.current_best_metric = operator.gt, "max", -float("inf")
        p = dict(lr_decay_params or {})
        self.decay_on_training_loss = p.pop("decay_on_training_loss", False)
        self.warmup_steps = p.pop("warmup_steps", 0)
        p.pop("warmup_ratio", None)
        self.base_lr = optimizer.param_groups[0]["lr"]
        self.optimizer = optimizer
        if lr_decay_func == "ReduceLROnPlateau" and "mode" not in p:
            p["mode"] = self.mode
        if lr_decay_func is not None:
            cls = eval(lr_decay_func)
            sig = inspect.signature(cls).parameters
            self.lr_decay_params = {k: v for k, v in p.items() if k in sig}
            self.scheduler = cls(optimizer, **self.lr_decay_params)
        else:
            self.lr_decay_params = {}
            self.scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda _: 1.0)

    def iteration_step(self, epoch_value=None):
        """``epoch_value`` = epoch + iteration / len(dl): iteration-stepped schedulers are stepped with the fractional epoch, so e.g.
        CosineAnnealingWarmRestarts counts T_0 in epochs (ref:vilmedic/executors/utils.py:404-424, trainor.py:125-126)"""
        self.iteration_count += 1
        if self.warmup_steps and self.iteration_count <= self.warmup_steps:
            for g in self.optimizer.param_groups:
                g["lr"] = self.base_lr * self.iteration_count / float(self.warmup_steps)
        elif self.scheduler_name in self.ITER_STEP:
            if epoch_value is not None and self.scheduler_name == "CosineAnnealingWarmRestarts":
                self.scheduler.step(epoch_value)          # (CyclicLR / OneCycleLR count calls; their step() ignores-or-rejects an epoch)
            else:
                self.scheduler.step()

    def epoch_step(self):
        self.epoch += 1
        if self.scheduler_name is not None and self.scheduler_name not in self.ITER_STEP | self.VAL_STEP:
            self.scheduler.step()

    def eval_step(self, decay_metric=None, early_stop_score=None):
        ret = {"done_training": False, "save_state": False}
        if decay_metric is not None and self.scheduler_name in self.VAL_STEP and self.iteration_count > self.warmup_steps:
            self.scheduler.step(decay_metric)
        if early_stop_score is not None:
            if self.metric_comp_func(early_stop_score, self.current_best_metric):
                self.current_best_metric, self.early_stop = early_stop_score, 0
                ret["save_state"] = True
            else:
                self.early_stop += 1
                if self.early_stop == self.early_stop_limit:
                    ret["done_training"] = True
        return ret

    def state_dict(self):
        d = {k: v for k, v in self.__dict__.items() if k not in ("scheduler", "optimizer", "metric_comp_func")}
        d["scheduler"] = self.scheduler.state_dict()
        return d

    def load_state_dict(self, sd):
        sd = dict(sd)
        self.scheduler.load_state_dict(sd.pop("scheduler"))
        self.__dict__.update(sd)

    def __repr__(self):
        return "TrainingScheduler(\n{}\n{}\nearly_stop_limit: {}, mode: {}\n)".format(
            self.scheduler_name, json.dumps(self.lr_decay_params, indent=4, sort_keys=True, default=str), self.early_stop_limit, self.mode)


def create_training_scheduler(config, optimizer, logger, state_dict=None):
    ts = TrainingScheduler(lr_decay_func=config.get("lr_decay"), optimizer=optimizer, early_stop_metric=config.get("early_stop_metric"),
                           early_stop_limit=config.get("early_stop"), lr_decay_params=config.get("lr_decay_params") or {})
    logger.settings("Training scheduler created")
    if state_dict is not None and "training_scheduler" in state_dict:
        ts.load_state_dict(state_dict["training_scheduler"])
    return ts
