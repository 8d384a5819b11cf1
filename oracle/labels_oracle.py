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


def refine_label_boundary(thr_channel, labels, threshold_val=(1e-7, 1e-4), ignore_zero_inside_bbox=True, closing=None):
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
    closed = (closing or binary_closing_disk7)(mask_threshold[y0:y1, x0:x1])     # :95 (`closing`: scipy's, for timing legs)
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


# ---- test-time chain (validation / evaluate flows) -------------------------------------------------------------------
LABEL_OVERLAP_VAL = -70            # constants.py:27
LABEL_SEABED_MASK_VAL = -50        # constants.py:28
LABEL_UNUSED_SPECIES = -10         # constants.py:30
SEABED_PAD = 10                    # mask_label_seabed.py:50-52


def convert_label_indexing_unused_species(labels):
    """convert_label_indexing.py:37-47: convert_label_indexing, then foreign species (> 0, not 1 / 27) -> -10."""
    out = convert_label_indexing(labels)
    out[(labels > 0) & (labels != 1) & (labels != 27)] = LABEL_UNUSED_SPECIES
    return out


def extend_boxes(boxes, mask_type, extend_size, shape0):
    """extend_label_masks.py:69-80: the school bounding boxes (y0, y1, x0, x1) as the mask uses them -- 'region': grown by
    ``extend_size`` on all four sides; 'trace': the whole first axis of the reader (``echogram.shape[0]``: the range axis
    of a memmap Echogram, the PING axis of the zarr reader -- restated as the reference has it), pings grown."""
    bb = np.array(boxes, dtype=np.int64).reshape(-1, 4).copy()
    if mask_type == "region":
        bb[:, 0] -= extend_size
        bb[:, 1] += extend_size
    else:
        bb[:, 0] = 0
        bb[:, 1] = shape0
    bb[:, 2] -= extend_size
    bb[:, 3] += extend_size
    return bb


def extend_label_mask(labels, centre, boxes_extended, ignore_val=-1):
    """get_extended_label_mask_for_crop.__call__ (extend_label_masks.py:57-98) for one patch: everything outside the
    (extended) boxes -> ``ignore_val`` (the reference's constructor default, -1: define_label_transform_test does not
    pass one, batch/transforms.py:88-89).  The patch is placed at centre - shape // 2 (:64)."""
    H, W = labels.shape
    yul, xul = int(centre[0]) - H // 2, int(centre[1]) - W // 2
    out = np.full(labels.shape, ignore_val, dtype=labels.dtype)
    for b0, b1, b2, b3 in np.asarray(boxes_extended).reshape(-1, 4):
        if min(b1, yul + H) - max(b0, yul) >= 0 and min(b3, xul + W) - max(b2, xul) >= 0:      # overlap(), :21-29
            ya, yb = max(b0 - yul, 0), min(b1 - yul, H)
            xa, xb = max(b2 - xul, 0), min(b3 - xul, W)
            out[ya:yb, xa:xb] = labels[ya:yb, xa:xb]
    return out


def test_label_transform(data, labels, centre, thr_channel_idx, seabed, n_range, patch_overlap=0, seabed_rule="zarr",
                         seabed_mask=None, threshold_val=(1e-7, 1e-4), nan_rule=True, boxes_extended=None):
    """define_label_transform_test (batch/transforms.py:81-99, label_masks='all') followed by remove_nan_inf's label rule
    (remove_nan_inf.py:30-32), for ONE patch: convert_label_indexing_unused_species -> refine_label_boundary (on the
    converted labels and the RAW linear-sv crop: the label transform runs before the data transform, batch/dataset.py:
    89-103) -> mask_label_seabed (mask_label_seabed.py:24-68) -> mask_label_overlap (mask_label_overlap.py:23-48).

    data [C,H,W] linear sv crop, labels [H,W] raw annotation ids (-100 outside the data), centre (range idx, GLOBAL ping
    idx) of the patch, ``seabed`` per-ping seabed index vector indexed by global ping (``seabed_mask`` [n_pings, n_range],
    zarr rule only: the reader's own 2-D mask instead).  seabed_rule: 'zarr' -- the reader shifts the mask down by the pad
    INSIDE the slice it is asked for (data_reader.py:837-841) -- or 'memm' -- ``Echogram.get_seabed_mask``: absolute rows
    >= seabed + pad (data_reader.py:407-431).  ``boxes_extended`` (``extend_boxes``): eval_mode 'region' / 'trace', the
    chain's optional last link.  H and W even (the reference's own coordinate helpers disagree by one pixel
    for odd sizes).  Returns int16 labels in {-100, -70, -50, -30, -10, 0, 1, 2}."""
    H, W = labels.shape
    assert H % 2 == 0 and W % 2 == 0
    lab = convert_label_indexing_unused_species(np.asarray(labels).astype(np.int64))
    lab = refine_label_boundary(np.asarray(data)[thr_channel_idx], lab, threshold_val)
    cy, cx = int(centre[0]), int(centre[1])
    y_data = cy - H // 2 + 1 + np.arange(H)                 # patch pixel p <-> data coordinate centre - n/2 + 1 + p
    x_data = cx - W // 2 + 1 + np.arange(W)
    y_top = max(cy - H // 2 + 1, 0)
    below = np.zeros((H, W), dtype=bool)
    ok_rows = (y_data >= 0) & (y_data < n_range)
    n_pings = len(seabed) if seabed_mask is None else np.asarray(seabed_mask).shape[0]
    for j, x in enumerate(x_data):
        if x < 0 or x >= n_pings:
            continue
        if seabed_rule == "zarr" and seabed_mask is not None:
            src = np.clip(y_data - SEABED_PAD, 0, n_range - 1)
            below[:, j] = ok_rows & (y_data - y_top >= SEABED_PAD) & (np.asarray(seabed_mask)[x, src] != 0)
        elif seabed_rule == "zarr":
            below[:, j] = ok_rows & (y_data - y_top >= SEABED_PAD) & ((y_data - SEABED_PAD) >= seabed[x])
        else:
            below[:, j] = ok_rows & ((y_data - SEABED_PAD) >= seabed[x])
    lab[below & (lab == BACKGROUND)] = LABEL_SEABED_MASK_VAL          # boundary / fish labels take precedence
    if patch_overlap > 0:
        o = patch_overlap
        out = np.full(lab.shape, LABEL_OVERLAP_VAL, dtype=np.int64)
        out[o:-o, o:-o] = lab[o:-o, o:-o]
        out[lab == LABEL_BOUNDARY_VAL] = LABEL_BOUNDARY_VAL
        lab = out
    if boxes_extended is not None:          # eval_mode 'region' / 'trace' (batch/transforms.py:87-90)
        lab = extend_label_mask(lab, centre, boxes_extended)
    if nan_rule:
        lab[~np.isfinite(np.asarray(data)[0])] = LABEL_IGNORE_VAL
    return lab.astype(np.int16)


test_label_transform.__test__ = False      # (not a pytest test: the name mirrors define_label_transform_test)
