#!/usr/bin/env python3
"""Entry point with the reference's command line (ref: bin/train.py:13-58):

    python bin/train.py config/RRG/rrg-vit-synthetic.yml trainor.batch_size=32 ...
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 bin/train.py <config> ...   (data parallel)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from vilmedic_amd.config import executor_view, get_config  # noqa: E402
from vilmedic_amd.executors import Trainor  # noqa: E402


def main():
    if len(sys.argv) < 2:
        raise SystemExit(__doc__)
    config = get_config(sys.argv[1], sys.argv[2:])
    seed = int(config.get("seed") or 0)
    tcfg = executor_view(config, "trainor")
    tcfg["validator_view"] = executor_view(config, "validator") if config.get("validator") else None
    Trainor(tcfg, seed).start()


if __name__ == "__main__":
    main()
