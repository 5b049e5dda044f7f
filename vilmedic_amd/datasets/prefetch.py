"""Host side of the training input path: batches are collated by a background thread into a ring of reusable pinned buffers
and copied to the GPU on a copy stream, ``depth`` batches ahead of the step that consumes them.

Why (measured, profiles/r02_q_trainor_loop.txt): a collated C2 batch is 64 x 3 x 224 x 224 fp32 = 38.5 MB.  ``torch.stack`` into a
FRESH allocation of that size page-faults every 4 KiB page (the previous batch is still alive, so the allocator cannot hand
the same block back): 90-140 ms per batch on the host, against a 25 ms training step -- the Trainor loop ran at 550 pairs/s
with the GPU idle 3/4 of the time.  Stacking into a buffer that already exists costs 2-10 ms, and doing it on another
thread takes it off the step's critical path altogether.

The reference leaves this to ``DataLoader(num_workers=4, pin_memory=True)`` (executors/utils.py:121-129); worker processes
pay the same page faults plus a shared-memory hop per batch, which is why SURVEY §8(f) ranks the input pipeline first among
what bounds the reference's own training loop.
"""
import queue
import threading

import numpy as np
import torch

_tls = threading.local()


class StagingRing:
    """``slots`` sets of host buffers, one set per in-flight batch; a buffer is created on first use (pinned when a GPU is
    present) and reused every ``slots`` batches.  ``fence`` attaches the event that marks the end of the slot's H2D copies;
    the slot is not written again before that event has completed."""

    def __init__(self, slots, pin):
        self.slots, self.pin = slots, pin
        self.bufs = [dict() for _ in range(slots)]
        self.events = [None] * slots
        self.cur = 0

    def buffer(self, key, shape, dtype):
        d = self.bufs[self.cur]
        k = (key, tuple(shape), dtype)
        b = d.get(k)
        if b is None:
            b = torch.empty(shape, dtype=dtype, pin_memory=self.pin)
            d[k] = b
        return b

    def flat(self, key, nbytes):
        """a byte buffer of at least ``nbytes`` (grown by 25 % steps; for payloads whose size changes from batch to batch)"""
        d = self.bufs[self.cur]
        b = d.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(nbytes + nbytes // 4, dtype=torch.uint8, pin_memory=self.pin)
            d[key] = b
        return b

    def fence(self, event):
        self.events[self.cur] = event
        self.cur = (self.cur + 1) % self.slots
        ev = self.events[self.cur]
        if ev is not None:
            ev.synchronize()
            self.events[self.cur] = None


def _copy_rows(out, tensors):
    """out[j] <- tensors[j] as plain memcpys (numpy, GIL released), one after the other.  ATen's own copy / cat kernels open an
    OpenMP parallel region, and a parallel region entered from a second Python thread builds a second thread team that
    spin-waits against the first: measured 124 ms (8 cores) to 345 ms (256 threads) per 38 MB batch, against 7 ms for this loop."""
    try:
        dst = out.numpy()
        for j, t in enumerate(tensors):
            np.copyto(dst[j], t.numpy())
    except TypeError:                       # dtypes numpy does not have (bf16): byte views
        dst = out.view(torch.uint8).numpy()
        for j, t in enumerate(tensors):
            np.copyto(dst[j], t.contiguous().view(torch.uint8).numpy())


def stack(tensors, key="images"):
    """``torch.stack`` for collate functions: into the active staging ring when the caller runs under a PrefetchLoader's
    thread, a plain stack otherwise (DataLoader worker processes, validation loaders, user code)."""
    ring = getattr(_tls, "ring", None)
    if ring is None:
        return torch.stack(tensors)
    out = ring.buffer(key, (len(tensors),) + tuple(tensors[0].shape), tensors[0].dtype)
    _copy_rows(out, tensors)
    return out


class PrefetchLoader:
    """iterates ``loader`` on a background thread and yields its batches with every tensor already on ``device``.

    The yielded tensors are valid until ``depth + 2`` further batches have been produced -- a training loop that consumes each
    batch before asking for the next (executors/trainor.py) never sees one change; code that keeps batches must clone them.
    On a machine without a GPU (``device=None``) the thread and the ring still run, tensors stay on the host."""

    def __init__(self, loader, depth=2, device="auto"):
        self.loader = loader
        self.dataset = loader.dataset
        self.depth = depth
        if device == "auto":
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        self.device = device
        self.ring = StagingRing(depth + 3, pin=device is not None)
        self.copy_stream = torch.cuda.Stream(device) if device is not None else None
        base = getattr(self.dataset, "dataset", self.dataset)          # through a torch Subset (rank shards)
        self.stage_hook = getattr(base, "stage_batch", None)            # dataset-specific host staging (imseq.py: image packing)

    def __len__(self):
        return len(self.loader)

    def __getattr__(self, name):          # batch_sampler, sampler, collate_fn, ... of the wrapped loader
        if name in ("loader", "__setstate__", "__getstate__"):      # (copy / pickle probe these before __init__ has run)
            raise AttributeError(name)
        return getattr(self.loader, name)

    # -- producer side --------------------------------------------------------------------------------------------------
    def _stage(self, key, t):
        if t.device.type != "cpu":
            return t
        if self.device is None:
            return t
        if not t.is_pinned():
            buf = self.ring.buffer(("stage", key), t.shape, t.dtype)
            _copy_rows(buf.reshape(1, *t.shape), [t.contiguous()])
            t = buf
        return t.to(self.device, non_blocking=True)

    def _to_device(self, batch):
        if isinstance(batch, dict):
            return {k: (self._stage(k, v) if torch.is_tensor(v) else v) for k, v in batch.items()}
        if torch.is_tensor(batch):
            return self._stage("batch", batch)
        return batch

    def _produce(self, q, stop):
        _tls.ring = self.ring
        try:
            if self.device is not None:
                torch.cuda.set_device(self.device)
            for batch in self.loader:
                ev = None
                if self.stage_hook is not None:
                    batch = self.stage_hook(batch, self.ring)
                if self.device is not None:
                    with torch.cuda.stream(self.copy_stream):
                        batch = self._to_device(batch)
                        ev = torch.cuda.Event()
                        ev.record(self.copy_stream)
                self.ring.fence(ev)
                if not _put(q, (batch, ev, None), stop):
                    return
            _put(q, (None, None, StopIteration), stop)
        except BaseException as e:        # surfaces in the consuming thread
            _put(q, (None, None, e), stop)
        finally:
            _tls.ring = None

    # -- consumer side --------------------------------------------------------------------------------------------------
    def __iter__(self):
        q, stop = queue.Queue(maxsize=self.depth), threading.Event()
        th = threading.Thread(target=self._produce, args=(q, stop), daemon=True, name="vm-prefetch")
        th.start()
        try:
            while True:
                batch, ev, err = q.get()
                if err is StopIteration:
                    return
                if err is not None:
                    raise err
                if ev is not None:
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(ev)
                    for v in (batch.values() if isinstance(batch, dict) else [batch]):
                        if torch.is_tensor(v) and v.is_cuda:
                            v.record_stream(cur)
                yield batch
        finally:
            stop.set()
            while th.is_alive():          # unblock a producer waiting on a full queue, then let it finish
                try:
                    q.get_nowait()
                except queue.Empty:
                    pass
                th.join(timeout=0.05)


def _put(q, item, stop):
    while not stop.is_set():
        try:
            q.put(item, timeout=0.1)
            return True
        except queue.Full:
            continue
    return False
