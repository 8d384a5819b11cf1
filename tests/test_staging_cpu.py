"""Host-side pieces of the input staging (crimac_classifiers_unet_amd/staging.py) that need no GPU: when a DataLoader's
batches may have their pages released early, and what that release does.  (The staged training loop itself -- pinned ring,
copy stream, equality with the in-line copy of the reference's pipeline.py:161-164 -- is tests/test_gpu_unet.py.)"""
import numpy as np
import pytest
import torch
from torch.utils.data import DataLoader, Dataset

from crimac_classifiers_unet_amd import staging


class Crops(Dataset):
    """The reference's batch dict with small crops."""

    def __init__(self, n=8, hw=128):
        self.data = np.random.default_rng(0).standard_normal((n, 4, hw, hw)).astype(np.float32)
        self.labels = np.zeros((n, hw, hw), np.int16)

    def __len__(self):
        return len(self.data)

    def __getitem__(self, i):
        return {"data": self.data[i], "labels": self.labels[i], "center_coordinates": np.array([i, i], dtype=np.int64)}


def test_only_worker_collated_default_batches_are_released_early():
    ds = Crops()
    assert staging.collated_in_worker(DataLoader(ds, batch_size=4, num_workers=2))
    assert not staging.collated_in_worker(DataLoader(ds, batch_size=4, num_workers=0))          # parent-owned memory
    assert not staging.collated_in_worker(DataLoader(ds, batch_size=None, num_workers=2))       # the Dataset's own tensors
    assert not staging.collated_in_worker(DataLoader(ds, batch_size=4, num_workers=2, collate_fn=lambda b: b))
    assert not staging.collated_in_worker([{"data": torch.zeros(1)}])                           # any other iterable


def test_release_shared_pages_frees_a_worker_collated_batch_and_leaves_other_tensors_alone():
    ds = Crops(n=8, hw=128)                                  # 4 x 4 x 128 x 128 x 4 B = 1 MiB per batch of `data`
    dl = DataLoader(ds, batch_size=4, num_workers=1)
    batch = next(iter(dl))
    assert batch["data"].is_shared() and float(batch["data"].abs().sum()) > 0
    ref = torch.from_numpy(ds.data[:4])
    assert torch.equal(batch["data"], ref)
    freed = staging.release_shared_pages(batch)
    assert freed >= batch["data"].numel() * 4 - 2 * 4096     # whole pages inside the tensor
    assert float(batch["data"].abs().sum()) == 0.0           # a hole in the segment reads as zeros; the mapping stays valid
    assert int(batch["center_coordinates"][1, 0]) == 1       # small tensors (below min_bytes) are not touched
    # private memory is never touched
    own = {"data": torch.ones(1 << 19)}
    assert staging.release_shared_pages(own) == 0 and float(own["data"].sum()) == float(1 << 19)
    # ... nor a non-contiguous view of a shared tensor
    sh = torch.ones(2, 1 << 19).share_memory_()
    assert staging.release_shared_pages({"data": sh[:, ::2]}) == 0 and float(sh.sum()) == float(2 << 19)
    del batch, dl
