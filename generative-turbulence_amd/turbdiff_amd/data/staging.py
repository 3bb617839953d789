"""Host -> device staging of batches (the role of ``pin_memory=True`` + Lightning's batch transfer in the
reference, ofles.py:629-643): ``DeviceStager`` keeps one batch in flight on a copy stream while the
current one trains, from rotating pinned host buffers, so the H2D copy of step i + 1 overlaps the kernels of
step i and the training stream never waits on pageable memory."""

from __future__ import annotations

from collections import OrderedDict

import torch

from .ofles import OpenFOAMBatch, OpenFOAMData, OpenFOAMStats


class DeviceStager:
    def __init__(self, batches, device, depth: int = 2, cache_size: int = 64):
        self.it, self.device, self.depth = iter(batches), torch.device(device), depth
        self.stream = torch.cuda.Stream(device=self.device)
        self._pinned: dict = {}   # (slot, key) -> pinned host buffer, reused while the shape fits
        # geometry / statistics move once per OBJECT: entries hold the CPU object they were made from (so its id
        # cannot be recycled for another geometry while the entry lives) and the maps are small LRUs
        self._meta: OrderedDict = OrderedDict()   # id(cpu metadata) -> (cpu metadata, device metadata)
        self._stats: OrderedDict = OrderedDict()
        self.cache_size = cache_size
        self._queue: list = []
        self._slot = 0
        self._slot_done: dict = {}  # slot -> event of the last H2D copy that read the slot's pinned buffers

    def _pin(self, slot, key, t):
        buf = self._pinned.get((slot, key))
        if buf is None or buf.shape != t.shape or buf.dtype != t.dtype:
            buf = self._pinned[(slot, key)] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
        buf.copy_(t)
        return buf

    def _cached(self, cache: OrderedDict, obj, make):
        hit = cache.get(id(obj))
        if hit is not None and hit[0] is obj:
            cache.move_to_end(id(obj))
            return hit[1]
        dev_obj = make(obj)
        cache[id(obj)] = (obj, dev_obj)
        while len(cache) > self.cache_size:
            cache.popitem(last=False)
        return dev_obj

    def _stage(self, batch: OpenFOAMBatch):
        slot, self._slot = self._slot, (self._slot + 1) % (self.depth + 1)
        if slot in self._slot_done:
            self._slot_done[slot].synchronize()  # the copy that last read these pinned buffers has finished
        meta = self._cached(self._meta, batch.data.metadata, lambda m: m.to(self.device))
        stats = self._cached(self._stats, batch.stats, lambda st: st.to(self.device) if hasattr(st, "to") else st)
        with torch.cuda.stream(self.stream):
            samples = {v: self._pin(slot, v, s).to(self.device, non_blocking=True) for v, s in batch.data.samples.items()}
            t = batch.data.t.to(self.device, non_blocking=True) if torch.is_tensor(batch.data.t) else batch.data.t
            done = torch.cuda.Event()
            done.record(self.stream)
        self._slot_done[slot] = done
        return OpenFOAMBatch(OpenFOAMData(meta, t, samples), stats), done

    def _fill(self):
        while len(self._queue) < self.depth:
            try:
                self._queue.append(self._stage(next(self.it)))
            except StopIteration:
                break

    def __iter__(self):
        return self

    def __next__(self):
        self._fill()
        if not self._queue:
            raise StopIteration
        batch, done = self._queue.pop(0)
        torch.cuda.current_stream(self.device).wait_event(done)  # device-side wait only
        for s in batch.data.samples.values():
            s.record_stream(torch.cuda.current_stream(self.device))
        self._fill()
        return batch
