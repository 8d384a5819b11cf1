"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the on-GPU augmentation + data transform.

Arithmetic follows the reference: add_noise (batch/data_augmentation/add_noise.py:21-41), flip_x_axis
(flip_x_axis.py:21-25), remove_nan_inf (remove_nan_inf.py:23-34), db_with_limits
(db_with_limits.py:20-24, :36-38).  The reference draws from numpy's global MT19937 in DataLoader
workers; the GPU path draws from Philox4x32-10 keyed on (seed, sample) with the element index as
counter, restated here bit for bit so the arithmetic can be compared exactly; the DISTRIBUTIONS
(p=.5 per sample, 5 % of values, half U(1,10) / half U(0,1), p=.5 flip) are what the reference
specifies and are checked statistically in tests/test_augment.py against the reference functions.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10 (Salmon et al. 2011); all arguments uint64 arrays holding 32-bit values."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) for v in (c0, c1, c2, c3))
    k0 = np.uint64(k0) & MASK
    k1 = np.uint64(k1) & MASK
    for _ in range(10):
        p0, p1 = M0 * c0, M1 * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & MASK, p1 & MASK, \
                         ((p0 >> np.uint64(32)) ^ c3 ^ k1) & MASK, p0 & MASK
        k0 = (k0 + np.uint64(W0)) & MASK
        k1 = (k1 + np.uint64(W1)) & MASK
    return c0, c1, c2, c3


def u01(x):
    return (np.asarray(x, dtype=np.uint64) >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def augment_db(data, labels, seed, do_noise=True, do_flip=True, return_linear=False, scaled=False):
    """data [B,C,H,W] float32 linear sv, labels [B,H,W] -> (dB data [B,C,H,W] float32, labels int16,
    noisy[B], flipped[B]).  return_linear: instead of the final labels return the RAW labels (flipped with
    the data, no NaN rule) and the augmented LINEAR data -- the inputs of the label transform, which the
    reference runs between augmentation and data transform (batch/dataset.py:89-103).  ``scaled``:
    db_with_limits_scaled (db_with_limits.py:27-33: 1 + dB / 75), the transform of the metadata configurations."""
    B, C, H, W = data.shape
    out = np.empty_like(data, dtype=np.float32)
    lab_out = np.empty(labels.shape, dtype=np.int16)
    lin_out = np.empty_like(data, dtype=np.float32)
    noisy_f, flip_f = np.zeros(B, bool), np.zeros(B, bool)
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    idx = np.arange(C * H * W, dtype=np.uint64)
    for b in range(B):
        s = philox4x32_10([0], [0], [0], [0xA5A5A5A5], lo ^ b, hi)
        noisy = bool(do_noise and u01(s[0])[0] < 0.5)
        flip = bool(do_flip and u01(s[1])[0] < 0.5)
        d = data[b].astype(np.float32).copy()
        if noisy:
            r = philox4x32_10(idx & MASK, idx >> np.uint64(32), np.ones_like(idx), np.zeros_like(idx), lo ^ b, hi)
            change = u01(r[0]) < np.float32(0.05)
            inc = u01(r[1]) < np.float32(0.5)
            f = np.where(inc, np.float32(1.0) + np.float32(9.0) * u01(r[2]), u01(r[3])).astype(np.float32)
            d = (d.reshape(-1) * np.where(change, f, np.float32(1.0))).reshape(C, H, W).astype(np.float32)
        nonfinite0 = ~np.isfinite(d[0])
        lin = d.copy()
        d = np.where(np.isfinite(d), d, np.float32(0))
        with np.errstate(divide="ignore"):
            d = np.clip(np.float32(10) * np.log10(d + np.float32(1e-10)), -75, 0).astype(np.float32)
        if scaled:
            d = (np.float32(1) + d / np.float32(75)).astype(np.float32)
        lab = labels[b].astype(np.int64).copy()
        if not return_linear:
            lab[nonfinite0] = -100
        if flip:
            d = d[:, :, ::-1]
            lab = lab[:, ::-1]
            lin = lin[:, :, ::-1]
        out[b], lab_out[b], lin_out[b] = d, lab.astype(np.int16), lin
        noisy_f[b], flip_f[b] = noisy, flip
    if return_linear:
        return out, lab_out, noisy_f, flip_f, lin_out
    return out, lab_out, noisy_f, flip_f


# ---- the worker-side chain with numpy's own generator (bench.py's host-chain baseline leg) ---------------------------
def worker_train_chain(data, labels, rng, thr_channel_idx=-1, threshold_val=(1e-7, 1e-4)):
    """What the reference's ``Dataset.__getitem__`` does to ONE raw crop inside a DataLoader worker (batch/dataset.py:
    89-103 with the factories of batch/transforms.py:39-78):
        add_noise (add_noise.py:21-41)  ->  flip_x_axis (flip_x_axis.py:21-25)
        ->  refine_label_boundary + convert_label_indexing  ->  remove_nan_inf  ->  db_with_limits.
    ``rng``: a ``numpy.random.RandomState`` (the reference draws from numpy's global MT19937 stream, seeded per worker);
    the draws are made in the reference's order and with its distributions -- randint(2); binomial(1, .05), binomial(1,
    .5), uniform(1, 10), uniform(0, 1), each of the crop's shape; randint(2) -- so the CPU time spent is the reference's.
    data [C, H, W] linear sv (modified in place where the reference does), labels [H, W] raw annotation ids.
    Returns (dB data in data's dtype, int16 labels in {0, 1, 2, -100})."""
    from . import labels_oracle as lo
    if rng.randint(2):
        change = rng.binomial(1, 0.05, data.shape)
        up = rng.binomial(1, 0.5, data.shape)
        factor = up * rng.uniform(1, 10, data.shape) + (1 - up) * rng.uniform(0, 1, data.shape)
        data *= (1 - change) + change * factor
    if rng.randint(2):
        data, labels = data[:, :, ::-1].copy(), labels[:, ::-1].copy()
    lab = lo.convert_label_indexing(lo.refine_label_boundary(data[thr_channel_idx], labels, threshold_val, closing=_closing()))
    bad = ~np.isfinite(data)
    lab[bad[0]] = lo.LABEL_IGNORE_VAL
    data[bad] = 0
    with np.errstate(divide="ignore"):
        db = 10 * np.log10(data + 1e-10)
    np.clip(db, -75, 0, out=db)
    return db.astype(data.dtype, copy=False), lab.astype(np.int16)


def _closing():
    """scipy.ndimage.binary_closing with the 7x7 disk -- the function the reference itself calls
    (refine_label_boundary.py:95) -- when scipy is there; the explicit-shift restatement otherwise."""
    try:
        from scipy.ndimage import binary_closing
        from .labels_oracle import CLOSING
        return lambda m: binary_closing(m, structure=CLOSING)
    except Exception:      # pragma: no cover
        return None
