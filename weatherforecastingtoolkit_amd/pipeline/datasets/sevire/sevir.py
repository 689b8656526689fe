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
HDF5 / catalog / AWS download of the reference are I/O and out of scope
(SURVEY.md §2 row 6); events come from memory (synthetic `synth.blob_events`
or any uint8 array the caller read elsewhere).
"""
from __future__ import annotations

import numpy as np
import torch

from .... import ops

PREPROCESS_SCALE_01 = {"vil": 1 / 255}
PREPROCESS_OFFSET_01 = {"vil": 0}


class SEVIRFrameLoader:
    def __init__(self, events_u8, batch_size, seq_len=1, stride=1, layout="NTHW", shuffle=False,
                 shuffle_seed=1, device=None, num_shard=1, rank=0):
        ev = np.ascontiguousarray(events_u8)
        assert ev.dtype == np.uint8 and ev.ndim == 4, "events must be uint8 (N_ev, H, W, T)"
        if layout != "NTHW":
            raise ValueError("only layout='NTHW' (the ae_v2 setting, train.py:290-304) is built")
        if shuffle:
            ev = ev[np.random.RandomState(shuffle_seed).permutation(ev.shape[0])]
        self.events = ev
        self.batch_size, self.seq_len, self.stride = int(batch_size), int(seq_len), int(stride)
        self.raw_seq_len = ev.shape[3]
        self.device = torch.device(device) if device is not None else None
        self.num_shard, self.rank = int(num_shard), int(rank)

    @property
    def num_seq_per_event(self):
        return 1 + (self.raw_seq_len - self.seq_len) // self.stride

    @property
    def total_num_seq(self):
        return int(self.num_seq_per_event * self.events.shape[0])

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
        return np.stack([self.events[e, :, :, s * self.stride:s * self.stride + self.seq_len] for e, s in idx], 0)

    def __getitem__(self, index):
        if index >= len(self):
            raise IndexError(index)
        u8 = torch.from_numpy(self.batch_u8(index))
        if self.device is None or self.device.type != "cuda":
            raise RuntimeError("SEVIRFrameLoader preprocesses on the GPU: pass device='cuda:N'")
        u8 = u8.to(self.device, non_blocking=True)
        return {"vil": ops.vil_u8_to_f32(u8, PREPROCESS_SCALE_01["vil"])}

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]
