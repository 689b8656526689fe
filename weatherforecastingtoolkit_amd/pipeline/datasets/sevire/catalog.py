"""Real-data side of the SEVIR loader: catalog filtering and event stores.

Reference: `SEVIRDataLoader.__init__` :316-343 (date / datetime / catalog filters), `_compute_samples` :347-361
(events that have every requested image type, duplicated ids dropped), `_df_to_series` :366-375, `_open_files`
:377-389, `_read_data` :455-482 (HDF5 dataset `vil`, uint8 (N, H, W, T), one event per `file_index`).

The HDF5 files themselves need `h5py`, which this image does not have: `H5EventSource` imports it lazily and fails
loudly; `NpyEventSource` reads the same logical arrays from `.npy` files (memory-mapped), which is what the tests
and the synthetic pipeline use.  No CPU preprocessing happens here: batches leave as uint8 and are converted on
the GPU (`ops.vil_u8_to_f32`).
"""
from __future__ import annotations

import os

import numpy as np


class SEVIRCatalog:
    """The filtered, per-event sample table of the reference loader for `data_types` (default ['vil'])."""

    def __init__(self, catalog, data_types=("vil",), start_date=None, end_date=None, datetime_filter=None,
                 catalog_filter="default", shuffle=False, shuffle_seed=1):
        import pandas as pd
        if isinstance(catalog, str):
            catalog = pd.read_csv(catalog, parse_dates=["time_utc"], low_memory=False)
        cat = catalog
        if start_date is not None:
            cat = cat[cat.time_utc > start_date]                       # reference :333-334 (strict)
        if end_date is not None:
            cat = cat[cat.time_utc <= end_date]                        # :335-336 (inclusive)
        if datetime_filter:
            cat = cat[datetime_filter(cat.time_utc)]
        if catalog_filter is not None:
            if catalog_filter == "default":
                catalog_filter = lambda c: c.pct_missing == 0          # noqa: E731  (:341)
            cat = cat[catalog_filter(cat)]
        self.data_types = list(data_types)
        imgts = set(self.data_types)
        filt = cat[np.logical_or.reduce([cat.img_type == i for i in self.data_types])]
        filt = filt.groupby("id").filter(lambda x: imgts.issubset(set(x["img_type"])))
        filt = filt.groupby("id").filter(lambda x: x.shape[0] == len(self.data_types))   # repeated ids are dropped
        rows = []
        for ev_id, df in filt.groupby("id"):                           # groupby sorts by id, like the reference
            df = df.set_index("img_type")
            row = {"id": ev_id}
            for t in self.data_types:
                s = df.loc[t]
                row[f"{t}_filename"] = s.file_name
                row[f"{t}_index"] = s.file_index if t != "lght" else s.id
            rows.append(row)
        self.samples = pd.DataFrame(rows)
        if shuffle and len(self.samples):
            self.samples = self.samples.sample(frac=1, random_state=int(shuffle_seed))   # :363-364
        self.samples = self.samples.reset_index(drop=True)

    def __len__(self):
        return len(self.samples)


class NpyEventSource:
    """`<data_dir>/<file_name with its extension replaced by .npy>` holds the file's `vil` dataset, uint8 (N,H,W,T)"""

    def __init__(self, data_dir):
        self.data_dir = data_dir
        self._files = {}

    def _open(self, file_name):
        arr = self._files.get(file_name)
        if arr is None:
            path = os.path.join(self.data_dir, os.path.splitext(file_name)[0] + ".npy")
            arr = np.load(path, mmap_mode="r")
            if arr.dtype != np.uint8 or arr.ndim != 4:
                raise ValueError(f"{path}: expected uint8 (N, H, W, T), got {arr.dtype} {arr.shape}")
            self._files[file_name] = arr
        return arr

    def read(self, file_name, index, img_type="vil"):
        return self._open(file_name)[int(index)]

    def close(self):
        self._files = {}


class H5EventSource:
    """the reference's HDF5 layout (`h5py.File(dir/file)[img_type][index]`, :377-389, :476)"""

    def __init__(self, data_dir):
        try:
            import h5py  # noqa: F401
        except ImportError as e:
            raise RuntimeError("H5EventSource needs h5py, which is not installed in this environment; convert the "
                               "SEVIR files to .npy and use NpyEventSource") from e
        self.data_dir = data_dir
        self._files = {}

    def read(self, file_name, index, img_type="vil"):
        import h5py
        f = self._files.get(file_name)
        if f is None:
            f = self._files[file_name] = h5py.File(os.path.join(self.data_dir, file_name), "r")
        return f[img_type][int(index)]

    def close(self):
        for f in self._files.values():
            f.close()
        self._files = {}


class CatalogEventStore:
    """events addressed by their row in the filtered catalog; what SEVIRFrameLoader indexes instead of an array"""

    def __init__(self, catalog: SEVIRCatalog, source, img_type="vil"):
        self.catalog, self.source, self.img_type = catalog, source, img_type
        first = self.read(0)
        self.event_shape = tuple(first.shape)            # (H, W, T)

    def __len__(self):
        return len(self.catalog)

    def read(self, event_idx):
        row = self.catalog.samples.iloc[int(event_idx)]
        ev = self.source.read(row[f"{self.img_type}_filename"], row[f"{self.img_type}_index"], self.img_type)
        return np.asarray(ev)
