#!/usr/bin/env python3
"""Where does the time of one Trainor iteration go?  Runs the reference-shaped training loop (executors/trainor.py, built from the
same YAML as bin/train.py) on the C2 config under four input conditions and prints ms / iteration for each:

    loader      the training loader alone (no model)
    fixed       Trainor.start() over ONE resident device batch, repeated (the loop without an input pipeline: what bench.py times)
    prefetch    Trainor.start() over the PrefetchLoader (the shipped default)
    graph       the same with trainor.graph_step (iterations replayed from a captured HIP graph)
    plain       Trainor.start() over the bare DataLoader (prefetch: 0 -- what round 1 shipped)

    python tools/trainor_loop_bench.py [iters=100] [num_samples=3200]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from vilmedic_amd.config import executor_view, get_config  # noqa: E402
from vilmedic_amd.executors import Trainor  # noqa: E402
from vilmedic_amd.executors.utils import create_data_loader, get_logger  # noqa: E402


class Timed:
    """iterable that times its consumer: wall clock from the ``skip``-th batch to exhaustion, device-synchronised at both ends"""

    def __init__(self, inner, n, skip=10):
        self.inner, self.n, self.skip, self.ms = inner, n, skip, None
        self.dataset = getattr(inner, "dataset", None)

    def __len__(self):
        return self.n

    def __iter__(self):
        t0, k = None, 0
        while k < self.n:
            for b in self.inner:
                if k == self.skip:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                if k >= self.n:
                    break
                k += 1
                yield b
        torch.cuda.synchronize()
        self.ms = (time.perf_counter() - t0) * 1e3 / (self.n - self.skip)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    ns = int(sys.argv[2]) if len(sys.argv) > 2 else 3200
    cfgfile = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "config/RRG/rrg-vit-synthetic.yml")
    config = get_config(cfgfile, ["dataset.num_samples=%d" % ns, "trainor.epochs=0", "ckpt_dir=/tmp/vm_loop_bench"])
    logger = get_logger()
    logger.setLevel("WARNING")
    tcfg = executor_view(config, "trainor")
    tcfg["validator_view"] = None
    tr = Trainor(tcfg, 0, logger=logger)
    tr.saver.save = lambda *a, **k: None
    bs = int(tcfg.batch_size)
    res = {}

    t = Timed(tr.dl, iters)
    for b in t:
        b["images"].add_(0)
    res["loader"] = t.ms

    first = next(iter(tr.dl))
    fixed = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in first.items()}

    def run(name, dl, n):
        tr.dl = Timed(dl, n)
        tr.training_scheduler.epoch = 0
        tr.start()
        res[name] = tr.dl.ms

    run("fixed", [fixed], iters)
    run("prefetch", create_data_loader(tcfg, "train", logger), iters)
    tr.graph_any = True          # trainor.graph_step: every iteration replayed from the HIP graph captured for the batch's shapes
    run("graph", create_data_loader(tcfg, "train", logger), iters)
    tr.graph_any = False
    tcfg["prefetch"] = 0
    run("plain", create_data_loader(tcfg, "train", logger), max(30, iters // 3))
    for k, v in res.items():
        print("%-9s %7.2f ms/iter  %8.0f pairs/s" % (k, v, bs / v * 1e3))


if __name__ == "__main__":
    main()
