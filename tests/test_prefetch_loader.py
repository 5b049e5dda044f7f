"""PrefetchLoader (datasets/prefetch.py): same batches in the same order as the wrapped DataLoader, exceptions surface in the
consumer, an abandoned iteration does not leave the producer thread behind."""
import threading

import pytest
import torch
from torch.utils.data import BatchSampler, DataLoader, RandomSampler

from vilmedic_amd.datasets import PrefetchLoader, SyntheticImSeq


def _loader(ds, seed, bs=8):
    g = torch.Generator().manual_seed(seed)
    return DataLoader(ds, collate_fn=ds.get_collate_fn(), batch_sampler=BatchSampler(RandomSampler(ds, generator=g), bs, True))


def _clone(b):
    return {k: (v.clone() if torch.is_tensor(v) else v) for k, v in b.items()}


def test_same_batches_and_buffer_lifetime():
    ds = SyntheticImSeq(num_samples=100, image_size=16, tokenizer_max_len=12, vocab_size=50)
    ref = [_clone(b) for b in _loader(ds, 3)]
    pl = PrefetchLoader(_loader(ds, 3), depth=2, device=None)
    assert len(pl) == len(ref) == 12
    prev = None
    for a, b in zip(ref, pl):
        assert b["images_mask"] is None
        for k in ("images", "input_ids", "attention_mask"):
            assert torch.equal(a[k], b[k])
        if prev is not None:          # the previous batch is still intact while the next one is consumed (the loop's usage)
            assert torch.equal(prev[0]["images"], prev[1]["images"])
        prev = (a, b)
    # staging buffers are reused: at most depth + 3 distinct image buffers over 12 batches
    ptrs = {b["images"].data_ptr() for b in pl}
    assert len(ptrs) <= 5


def test_error_and_early_exit():
    class Bad(SyntheticImSeq):
        def __getitem__(self, i):
            if i == 7:
                raise KeyError("sample 7")
            return super().__getitem__(i)

    ds = Bad(num_samples=32, image_size=8, tokenizer_max_len=8, vocab_size=20)
    with pytest.raises(KeyError):
        for _ in PrefetchLoader(_loader(ds, 0), device=None):
            pass
    good = SyntheticImSeq(num_samples=64, image_size=8, tokenizer_max_len=8, vocab_size=20)
    for _ in PrefetchLoader(_loader(good, 0), device=None):
        break
    assert not [t for t in threading.enumerate() if t.name == "vm-prefetch" and t.is_alive()]


@pytest.mark.gpu
def test_device_batches_match_host():
    ds = SyntheticImSeq(num_samples=96, image_size=32, tokenizer_max_len=12, vocab_size=50)
    ref = [_clone(b) for b in _loader(ds, 5)]
    pl = PrefetchLoader(_loader(ds, 5), depth=2)
    seen = []
    for a, b in zip(ref, pl):
        assert b["images"].is_cuda and b["input_ids"].is_cuda
        y = b["images"] * 2.0                      # consume on the current stream
        seen.append((a, y))
    torch.cuda.synchronize()
    for a, y in seen:
        assert torch.equal(a["images"] * 2.0, y.cpu())


def test_create_data_loader_wraps_the_training_loader_reproducibly():
    """executors/utils.create_data_loader: the training split runs under a PrefetchLoader whose shuffle order is a function of the
    global seed alone (its own generator, so the producer thread never draws from the global RNG); ``prefetch: 0`` and the
    evaluation splits keep the bare DataLoader"""
    import logging

    from vilmedic_amd.config import wrap
    from vilmedic_amd.executors.utils import create_data_loader
    log = logging.getLogger("t")
    log.settings = log.info
    cfg = wrap({"dataset": {"proto": "SyntheticImSeq", "num_samples": 40, "image_size": 8, "vocab_size": 30, "tokenizer_max_len": 8}, "batch_size": 4})
    orders = []
    for _ in range(2):
        torch.manual_seed(123)
        dl = create_data_loader(cfg, "train", log)
        assert isinstance(dl, PrefetchLoader) and len(dl) == 10
        orders.append(torch.cat([b["input_ids"] for b in dl]))
    assert torch.equal(orders[0], orders[1])
    ev = create_data_loader(cfg, "validate", log)
    assert not isinstance(ev, PrefetchLoader)
    off = wrap({"dataset": dict(cfg.dataset), "batch_size": 4, "prefetch": 0})
    assert not isinstance(create_data_loader(off, "train", log), PrefetchLoader)
