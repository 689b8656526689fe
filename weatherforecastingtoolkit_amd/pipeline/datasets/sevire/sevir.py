"""SEVIR frame loader contract of the reference (pipeline/datasets/sevire/sevir.py)
for the AE train step, with the uint8 -> fp32 / 255 + layout change done on the
GPU by a HIP kernel.

Contract reproduced (reference lines):
  * events are uint8 VIL arrays (N_ev, H, W, raw_T)                 :453-482, 681-716
  * num_seq_per_event = 1 + (raw_T - seq_len) // stride             :403-404
  * batch `index` = batch_size consecutive (event, seq) pairs from
    event (index*B)//nspe, seq (index*B)%nspe                       :979-1003
  * len = total_num_seq // batch_size                               :657-661
  * x = (1/255) * (u8.float() + 0), layout 'NHWT' -> 'NTHW'         :749-794, 98-139, 163, 168
  * the outer DataLoader yields the dict of ONE pre-formed batch    :1132-1151
  * train events are shuffled ONCE with random_state=1              :363-367 (here: numpy RandomState(1) permutation)
Events come either from memory (synthetic `synth.blob_events`, or any uint8 array) or from an event store
(`catalog.CatalogEventStore`: filtered SEVIR catalog + .npy / HDF5 files).
`prefetch()` overlaps the host gather and the uint8 H2D copy of batch i+1 (pinned staging buffers, a copy
stream) with the training step of batch i.  AWS download of the reference is out of scope.
"""
from __future__ import annotations

import queue
import threading

import numpy as np
import torch

from .... import ops

PREPROCESS_SCALE_01 = {"vil": 1 / 255}
PREPROCESS_OFFSET_01 = {"vil": 0}


# out_layout values of the reference's change_layout_torch (sevire/sevir.py:98-139).  The device kernel writes the frames
# as contiguous 'NTHW' (the ae_v2 setting, train.py:290-304; T plays the role of the channel dimension); the other
# layouts are the reference's own permute / unsqueeze VIEWS of that tensor (the reference returns views too unless
# ret_contiguous is set), so values, shapes and dtypes agree with change_layout_torch(x_nhwt, 'NHWT', layout).
LAYOUTS = {
    "NTHW": lambda t: t,
    "NHWT": lambda t: t.permute(0, 2, 3, 1),
    "NTCHW": lambda t: t.unsqueeze(2),
    "NTHWC": lambda t: t.unsqueeze(-1),
    "TNHW": lambda t: t.permute(1, 0, 2, 3),
    "TNCHW": lambda t: t.permute(1, 0, 2, 3).unsqueeze(2),
}


def change_layout_torch(data, in_layout="NHWT", out_layout="NHWT", ret_contiguous=False):
    """the reference's layout switch (sevire/sevir.py:98-139) for tensors that already live on the device"""
    to_nhwt = {"NHWT": lambda d: d, "NTHW": lambda d: d.permute(0, 2, 3, 1),
               "NTCHW": lambda d: d[:, :, 0].permute(0, 2, 3, 1), "NTHWC": lambda d: d[..., 0].permute(0, 2, 3, 1),
               "TNHW": lambda d: d.permute(1, 2, 3, 0), "TNCHW": lambda d: d[:, :, 0].permute(1, 2, 3, 0)}
    if in_layout not in to_nhwt or out_layout not in LAYOUTS:
        raise NotImplementedError
    data = LAYOUTS[out_layout](to_nhwt[in_layout](data).permute(0, 3, 1, 2))
    return data.contiguous() if ret_contiguous else data


class SEVIRFrameLoader:
    def __init__(self, events_u8, batch_size, seq_len=1, stride=1, layout="NTHW", shuffle=False,
                 shuffle_seed=1, device=None, num_shard=1, rank=0):
        if layout not in LAYOUTS:
            raise NotImplementedError(f"layout {layout!r}: the reference's change_layout_torch knows {sorted(LAYOUTS)}")
        self.layout = layout
        if hasattr(events_u8, "read") and hasattr(events_u8, "event_shape"):
            # an event store (catalog + files): events are read on demand; shuffling is the catalog's job
            self.store, self.events = events_u8, None
            self.n_events = len(events_u8)
            self.raw_seq_len = events_u8.event_shape[2]
        else:
            ev = np.ascontiguousarray(events_u8)
            assert ev.dtype == np.uint8 and ev.ndim == 4, "events must be uint8 (N_ev, H, W, T)"
            if shuffle:
                ev = ev[np.random.RandomState(shuffle_seed).permutation(ev.shape[0])]
            self.store, self.events = None, ev
            self.n_events = ev.shape[0]
            self.raw_seq_len = ev.shape[3]
        self._cache = {}
        self.batch_size, self.seq_len, self.stride = int(batch_size), int(seq_len), int(stride)
        self.device = torch.device(device) if device is not None else None
        self.num_shard, self.rank = int(num_shard), int(rank)

    @property
    def num_seq_per_event(self):
        return 1 + (self.raw_seq_len - self.seq_len) // self.stride

    @property
    def total_num_seq(self):
        return int(self.num_seq_per_event * self.n_events)

    def __len__(self):
        return (self.total_num_seq // self.batch_size) // self.num_shard

    def sample_indices(self, index):
        """[(event_idx, seq_idx)] of batch `index` — reference _idx_sample :992-1003."""
        event_idx = (index * self.batch_size) // self.num_seq_per_event
        seq_idx = (index * self.batch_size) % self.num_seq_per_event
        out = []
        while len(out) < self.batch_size:
            out.append((event_idx, seq_idx))
            seq_idx += 1
            if seq_idx >= self.num_seq_per_event:
                event_idx += 1
                seq_idx = 0
        return out

    def batch_u8(self, index):
        """uint8 (B, H, W, seq_len) host batch, before preprocessing."""
        idx = self.sample_indices(index * self.num_shard + self.rank)
        return np.stack([self._event(e)[:, :, s * self.stride:s * self.stride + self.seq_len] for e, s in idx], 0)

    def _event(self, e):
        if self.events is not None:
            return self.events[e]
        ev = self._cache.get(e)
        if ev is None:
            if len(self._cache) >= 4:                 # consecutive batches walk the events in order
                self._cache.pop(next(iter(self._cache)))
            ev = self._cache[e] = self.store.read(e)
            if ev.dtype != np.uint8:
                raise TypeError(f"event {e}: expected uint8, got {ev.dtype}")
        return ev

    def prefetch(self, depth=2, start=0):
        """iterate (from batch `start`) with the host gather + uint8 H2D copy of the next `depth` batches running
        ahead on a copy stream"""
        return _Prefetcher(self, depth, start)

    def __getitem__(self, index):
        if index >= len(self):
            raise IndexError(index)
        u8 = torch.from_numpy(self.batch_u8(index))
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("SEVIRFrameLoader preprocesses on the GPU: pass device='cuda:N'")
        u8 = u8.to(self.device, non_blocking=True)
        return {"vil": LAYOUTS[self.layout](ops.vil_u8_to_f32(u8, PREPROCESS_SCALE_01["vil"]))}

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]


class _Prefetcher:
    """Background thread: gather batch i+1 into a pinned uint8 buffer and start its H2D copy on a side stream while
    the consumer trains on batch i; the u8 -> fp32/255 + layout kernel runs on the consumer's stream after an event
    wait.  `depth` pinned buffers / device buffers are recycled."""

    def __init__(self, loader, depth=2, start=0):
        if loader.device is None or loader.device.type != "cuda":
            raise RuntimeError("prefetch() needs a CUDA(HIP) device")
        self.loader, self.depth, self.start = loader, max(1, int(depth)), max(0, int(start))

    def __iter__(self):
        ld = self.loader
        n = len(ld)
        if n == 0 or self.start >= n:
            return
        dev = ld.device
        shape = ld.batch_u8(0).shape
        pinned = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(self.depth + 1)]
        dbuf = [torch.empty(shape, dtype=torch.uint8, device=dev) for _ in range(self.depth + 1)]
        free_slots = queue.Queue()
        for i in range(self.depth + 1):
            free_slots.put(i)
        ready = queue.Queue(maxsize=self.depth)
        copy_stream = torch.cuda.Stream(device=dev)
        stop = threading.Event()

        def producer():
            try:
                torch.cuda.set_device(dev)
                for i in range(self.start, n):
                    if stop.is_set():
                        return
                    slot = free_slots.get()
                    pinned[slot].numpy()[...] = ld.batch_u8(i)
                    with torch.cuda.stream(copy_stream):
                        dbuf[slot].copy_(pinned[slot], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record(copy_stream)
                    ready.put((slot, ev, None))
                ready.put((None, None, None))
            except BaseException as e:      # surface loader errors in the consumer
                ready.put((None, None, e))

        th = threading.Thread(target=producer, daemon=True)
        th.start()
        try:
            while True:
                slot, ev, err = ready.get()
                if err is not None:
                    raise err
                if slot is None:
                    break
                torch.cuda.current_stream(dev).wait_event(ev)
                out = {"vil": LAYOUTS[self.loader.layout](ops.vil_u8_to_f32(dbuf[slot], PREPROCESS_SCALE_01["vil"]))}
                done = torch.cuda.Event()
                done.record(torch.cuda.current_stream(dev))
                yield out
                done.synchronize()           # the conversion kernel has consumed the slot: recycle it
                free_slots.put(slot)
        finally:
            stop.set()                       # an early `break` in the consumer: unblock and retire the producer
            try:
                while True:
                    ready.get_nowait()
            except queue.Empty:
                pass
            for i in range(self.depth + 1):
                free_slots.put(i)
            th.join(timeout=5)
