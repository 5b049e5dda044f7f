#!/usr/bin/env python3
"""Entry point with the reference's command line (ref: bin/train.py:13-58):

    python bin/train.py config/RRG/rrg-vit-synthetic.yml trainor.batch_size=32 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bin/train.py <config> ...   (data parallel)

As in the reference: checkpoints, the ``{seed}.log`` file and a ``config_{seed}.json`` dump go to ``<ckpt_dir>/<name>/``; ``ckpt=`` names
a checkpoint to resume from (relative names are looked up in that directory) and the seed is then taken from its file name.
"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vilmedic_amd.config import executor_view, get_config, to_container  # noqa: E402
from vilmedic_amd.executors import Trainor  # noqa: E402
from vilmedic_amd.executors.utils import get_logger  # noqa: E402


def prepare(config):
    """the reference's run-directory conventions (bin/train.py:19-33, bin/utils.py:17-20) -> (config, seed)"""
    seed = int(config.get("seed") or 0)
    config["ckpt_dir"] = os.path.join(config.get("ckpt_dir") or "ckpt", str(config.get("name") or "run"))
    os.makedirs(config["ckpt_dir"], exist_ok=True)
    if config.get("ckpt"):
        if not os.path.exists(config["ckpt"]):
            config["ckpt"] = os.path.join(config["ckpt_dir"], config["ckpt"])
        assert os.path.exists(config["ckpt"]), "Path '{}' does not exist".format(config["ckpt"])
        seed = int(re.match(r".*_(.*?)\.pth", config["ckpt"]).group(1))          # 1.68_10_560435.pth -> 560435
    return config, seed


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    config, seed = prepare(get_config(sys.argv[1], sys.argv[2:]))
    # ref:bin/utils.py:158 (get_seed, called by the reference's bin/train.py): cudnn.benchmark on -- on ROCm MIOpen's search over its convolution solvers
    # per shape.  The first steps of a CNN-tower run pay for it (minutes on a box without MIOpen's kernel cache); the steady step gains 8-27 %
    # (profiles/r06_h_cudnn_benchmark.txt: MVQA 122.2 -> 113.0 ms fp32, 73.5 -> 64.7 ms bf16 towers).  VM_CUDNN_BENCHMARK=0 keeps the immediate-mode choice.
    import torch
    torch.backends.cudnn.benchmark = os.environ.get("VM_CUDNN_BENCHMARK", "1") != "0"
    rank0 = int(os.environ.get("RANK", "0")) == 0
    logger = get_logger(path=os.path.join(config["ckpt_dir"], "{}.log".format(seed)) if rank0 else None)
    if rank0:
        with open(os.path.join(config["ckpt_dir"], "config_{}.json".format(seed)), "w") as f:
            json.dump(to_container(config), f, indent=4, default=str)
    tcfg = executor_view(config, "trainor")
    tcfg["validator_view"] = executor_view(config, "validator") if config.get("validator") else None
    Trainor(tcfg, seed, logger=logger).start()


if __name__ == "__main__":
    main()
