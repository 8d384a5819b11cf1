"""ORACLE -- TEST INFRASTRUCTURE ONLY.  numpy restatement of the training label transform.

Reference: define_label_transform_train (batch/transforms.py:71-78) =
  refine_label_boundary (batch/label_transforms/refine_label_boundary.py:35-104), then
  convert_label_indexing (batch/label_transforms/convert_label_indexing.py:24-35),
applied after the augmentation and before remove_nan_inf / db_with_limits (batch/dataset.py:89-103), whose
label rule (labels = -100 where channel 0 is non-finite, remove_nan_inf.py:23-34) therefore comes last.

The reference calls scipy.ndimage.binary_closing (scipy is a pip dependency of the reference, not vendored:
requirements.txt `scipy`); its published algorithm -- dilation then erosion with the same structuring
element, both with border_value = 0, on the array it is GIVEN (here: the bounding-box crop of the pixels
that are not LABEL_BOUNDARY_VAL) -- is restated with explicit shifts below and pinned against the reference
itself in tests/golden/labels.npz (tools/make_golden_labels.py).
"""
import numpy as np

LABEL_IGNORE_VAL = -100            # constants.py:25
LABEL_BOUNDARY_VAL = -100          # constants.py:26
LABEL_REFINE_BOUNDARY_VAL = -30    # constants.py:29
BACKGROUND, SANDEEL, OTHER = 0, 1, 2   # constants.py:20-22

# refine_label_boundary.py:50-58
CLOSING = np.array([[0, 0, 1, 1, 1, 0, 0],
                    [0, 1, 1, 1, 1, 1, 0],
                    [1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1],
                    [1, 1, 1, 1, 1, 1, 1],
                    [0, 1, 1, 1, 1, 1, 0],
                    [0, 0, 1, 1, 1, 0, 0]], dtype=bool)
OFFSETS = [(dy - 3, dx - 3) for dy in range(7) for dx in range(7) if CLOSING[dy, dx]]


def _shifted(a, dy, dx):
    """b[y, x] = a[y + dy, x + dx], False outside the array (border_value = 0)."""
    H, W = a.shape
    b = np.zeros_like(a)
    ys, ye = max(0, -dy), min(H, H - dy)
    xs, xe = max(0, -dx), min(W, W - dx)
    if ys < ye and xs < xe:
        b[ys:ye, xs:xe] = a[ys + dy:ye + dy, xs + dx:xe + dx]
    return b


def binary_closing_disk7(mask):
    """scipy.ndimage.binary_closing(mask, structure=CLOSING): dilation, then erosion, border_value 0."""
    dil = np.zeros_like(mask, dtype=bool)
    for dy, dx in OFFSETS:
        dil |= _shifted(mask, -dy, -dx)
    ero = np.ones_like(mask, dtype=bool)
    for dy, dx in OFFSETS:
        ero &= _shifted(dil, dy, dx)
    return ero


def refine_label_boundary(thr_channel, labels, threshold_val=(1e-7, 1e-4), ignore_zero_inside_bbox=True):
    """refine_label_boundary.py:35-104 for one patch.  thr_channel: data[freq_idx] [H,W] (linear sv, as the
    reference compares it: in the array's own dtype), labels [H,W] raw annotation ids."""
    below = LABEL_REFINE_BOUNDARY_VAL if ignore_zero_inside_bbox else 0
    new_labels = labels.copy()
    idxs = np.argwhere(new_labels != LABEL_BOUNDARY_VAL)
    if len(idxs) == 0:                      # :81-83: patch entirely outside the data, labels unchanged
        return new_labels
    y0, y1 = idxs[:, 0].min(), idxs[:, 0].max() + 1
    x0, x1 = idxs[:, 1].min(), idxs[:, 1].max() + 1
    lo = np.asarray(threshold_val[0]).astype(thr_channel.dtype)
    hi = np.asarray(threshold_val[1]).astype(thr_channel.dtype)
    with np.errstate(invalid="ignore"):
        mask_threshold = (labels > 0) & (thr_channel > lo) & (thr_channel < hi)     # :92-93
    closed = binary_closing_disk7(mask_threshold[y0:y1, x0:x1])                  # :95
    mask = np.zeros(labels.shape, dtype=bool)
    mask[y0:y1, x0:x1] = (~closed) & (new_labels[y0:y1, x0:x1] > 0)             # :97-98
    new_labels[mask] = below                                                     # :100
    new_labels[labels == LABEL_IGNORE_VAL] = LABEL_IGNORE_VAL                    # :101
    return new_labels


def convert_label_indexing(labels, ignore_val=-100):
    """convert_label_indexing.py:24-35: 0 -> BACKGROUND, 27 -> SANDEEL, 1 -> OTHER, everything else ignored."""
    out = np.full(labels.shape, ignore_val, dtype=np.int64)
    out[labels == 0] = BACKGROUND
    out[labels == 27] = SANDEEL
    out[labels == 1] = OTHER
    return out


def train_label_transform(data, labels, thr_channel_idx, threshold_val=(1e-7, 1e-4), nan_rule=True):
    """Batch form of the training chain: data [B,C,H,W] linear sv (after augmentation), labels [B,H,W] raw ids
    -> int16 labels in {0, 1, 2, -100} as they reach the loss (dataset.py:97-105)."""
    out = np.empty(labels.shape, dtype=np.int16)
    for b in range(labels.shape[0]):
        lab = convert_label_indexing(refine_label_boundary(data[b, thr_channel_idx], labels[b], threshold_val))
        if nan_rule:
            lab[~np.isfinite(data[b, 0])] = LABEL_IGNORE_VAL       # remove_nan_inf.py:30-32
        out[b] = lab.astype(np.int16)
    return out
