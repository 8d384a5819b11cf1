"""ORACLE -- TEST INFRASTRUCTURE ONLY.  CPU (numpy) restatement of the reference's tiled
whole-survey inference plumbing around the U-Net.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this.

Restated pieces (each cites the reference lines it follows; pinned against the reference's own
functions run on a fake in-memory reader, fixture ``tests/golden/tiling.npz`` made by
``tools/make_golden_tiling.py``):

* chunk plan       -- utils/preload_data_split.py:22-30 (``get_data_split``)
* patch-centre grid -- batch/samplers/gridded.py:22-54, :121-163 (``get_data_grid`` mode 'all')
* crop geometry    -- utils/np.py:38-46 (``getGrid``), :347-375 (``new_get_crop_2d/3d``),
                      batch/dataset.py:192-205 (``get_preload_data_labels``)
* label masks that decide which output pixels are written -- label_transforms/
  convert_label_indexing.py:24-47, mask_label_seabed.py:24-68, mask_label_overlap.py:23-48,
  data_transforms/remove_nan_inf.py:23-34
* data transform   -- data_transforms/remove_nan_inf.py:23-34, db_with_limits.py:20-24, :36-38
* scatter          -- pipeline_train_predict/save_predict.py:41-65 (``fill_out_array``)
"""
from __future__ import annotations

import numpy as np

LABEL_IGNORE_VAL = -100
LABEL_BOUNDARY_VAL = -100
LABEL_OVERLAP_VAL = -70
LABEL_SEABED_MASK_VAL = -50
LABEL_UNUSED_SPECIES = -10
SANDEEL, OTHER = 1, 2
SEABED_PAD = 10          # mask_label_seabed.py:50-52
SEABED_MARGIN = 50       # gridded.py:150-156


def get_data_split(valid_pings_ranges, max_n_pings=1000):
    """Equal-sized chunks of at most ``max_n_pings`` pings (preload_data_split.py:22-30)."""
    splits = []
    for start, end in valid_pings_ranges:
        n_splits = int(np.ceil((end - start) / max_n_pings))
        edges = np.linspace(start, end, n_splits + 1).astype(int)
        splits.extend([[edges[i], edges[i + 1]] for i in range(n_splits)])
    return np.array(splits)


def get_data_grid(n_range, max_seabed, start_ping, end_ping, patch_size=(256, 256), patch_overlap=20):
    """Patch centres (range idx, ping idx), ping index fastest (gridded.py:35-54, :150-159).

    ``max_seabed`` is the deepest seabed index in [start_ping, end_ping); the range extent is capped
    at ``max_seabed + 50``.
    """
    end_range = n_range
    cap = max_seabed + SEABED_MARGIN
    if cap < end_range:
        end_range = cap
    pw, ph = patch_size
    ys = np.arange(0 - (patch_overlap + 1), end_range - (patch_overlap + 1), ph - 2 * patch_overlap) + ph // 2
    xs = np.arange(start_ping - (patch_overlap + 1), end_ping - (patch_overlap + 1),
                   pw - 2 * patch_overlap) + pw // 2
    return np.array(np.meshgrid(ys, xs)).T.reshape(-1, 2)


def patch_offsets(n):
    """Offsets of the n patch pixels relative to the centre: -((n+1)//2)+1 .. n//2 (np.py:40-46)."""
    return np.arange(-((n + 1) // 2) + 1, n // 2 + 1)


def crop(arr, centre, size, boundary_val):
    """``new_get_crop_2d/3d``: arr[..., H, W] sampled on the patch grid, ``boundary_val`` outside."""
    ys = centre[0] + patch_offsets(size[0])[:, None] + np.zeros((1, size[1]), dtype=int)
    xs = centre[1] + patch_offsets(size[1])[None, :] + np.zeros((size[0], 1), dtype=int)
    n0, n1 = arr.shape[-2:]
    oob = (ys < 0) | (xs < 0) | (ys >= n0) | (xs >= n1)
    ys = np.where(oob, 0, ys)
    xs = np.where(oob, 0, xs)
    out = arr[..., ys, xs].copy()
    out[..., oob] = boundary_val
    return out


def data_transform(data):
    """remove_nan_inf + db_with_limits on a linear-sv crop (returns data, non-finite mask of ch 0)."""
    nonfinite0 = ~np.isfinite(data[0])
    data = np.where(np.isfinite(data), data, data.dtype.type(0))
    db = 10 * np.log10(data + 1e-10)
    db = np.clip(db, -75, 0).astype(data.dtype)
    return db, nonfinite0


def patch_labels(raw_labels, centre, size, seabed, n_range, patch_overlap, nonfinite0, seabed_rule="zarr",
                 seabed_mask=None):
    """Labels of one patch after the test-time transforms, as far as they decide validity.

    raw_labels: [n_range, chunk_pings] crop source (chunk-local ping axis, already offset);
    seabed: per-ping seabed index vector indexable by the GLOBAL ping of each patch column.
    ``seabed_mask`` [n_pings_total, n_range] (zarr rule only): the reader's own 2-D mask (``get_seabed_mask``, 1 below the
    seabed) -- what mask_label_seabed.py:47-49 really reads; ``seabed`` is then only its argmax (data_reader.py:864-865)
    and the two differ for pings without a detected bottom (all-zero column) and for masks with holes.
    Returns the label patch with values in {-100, -70, -50, -10, 0, 1, 2} (refine_label_boundary's
    -30 never changes validity and is not modelled).
    """
    lab = crop(raw_labels, centre["local"], size, LABEL_BOUNDARY_VAL).astype(np.int64)
    raw = lab.copy()
    # convert_label_indexing_unused_species
    new = np.full(raw.shape, LABEL_IGNORE_VAL, dtype=np.int64)
    new[raw == 0] = 0
    new[raw == 27] = SANDEEL
    new[raw == 1] = OTHER
    new[(raw > 0) & (raw != 1) & (raw != 27)] = LABEL_UNUSED_SPECIES
    lab = new
    # mask_label_seabed (reader pad semantics: the mask slice is shifted down by 10 INSIDE the slice)
    cy, cx = centre["global"]
    offs_y, offs_x = patch_offsets(size[0]), patch_offsets(size[1])
    y_data = cy + offs_y
    x_data = cx + offs_x
    y_top = max(cy - size[0] // 2 + 1, 0)
    below = np.zeros(lab.shape, dtype=bool)
    for j, x in enumerate(x_data):
        if x < 0 or x >= len(seabed):
            continue
        ok_rows = (y_data >= 0) & (y_data < n_range)
        r = y_data - y_top
        if seabed_rule == "zarr" and seabed_mask is not None:
            src = np.clip(y_data - SEABED_PAD, 0, n_range - 1)
            below[:, j] = ok_rows & (r >= SEABED_PAD) & (np.asarray(seabed_mask)[x, src] != 0)
        elif seabed_rule == "zarr":    # zarr reader: the 10-pixel pad shifts the mask down INSIDE the requested slice
            below[:, j] = ok_rows & (r >= SEABED_PAD) & ((y_data - SEABED_PAD) >= seabed[x])
        else:                          # Echogram.get_seabed_mask (data_reader.py:407-431): absolute rows >= seabed + pad
            below[:, j] = ok_rows & ((y_data - SEABED_PAD) >= seabed[x])
    lab[below & (lab == 0)] = LABEL_SEABED_MASK_VAL
    # mask_label_overlap
    if patch_overlap > 0:
        out = np.full(lab.shape, LABEL_OVERLAP_VAL, dtype=np.int64)
        o = patch_overlap
        out[o:-o, o:-o] = lab[o:-o, o:-o]
        out[lab == LABEL_BOUNDARY_VAL] = LABEL_BOUNDARY_VAL
        lab = out
    # remove_nan_inf
    if nonfinite0 is not None:
        lab[nonfinite0] = LABEL_IGNORE_VAL
    return lab


def fill_out_array(out_array, preds, labels, centre, ping_start):
    """save_predict.py:41-65: write softmax channels [SANDEEL, OTHER] at the valid pixels."""
    valid = (labels != LABEL_OVERLAP_VAL) & (labels != LABEL_SEABED_MASK_VAL) & (labels != LABEL_BOUNDARY_VAL)
    yl, xl = np.nonzero(valid)
    if len(yl) == 0:
        return out_array
    y = yl + centre[0] - labels.shape[0] // 2 + 1
    x = xl + centre[1] - labels.shape[1] // 2 + 1 - ping_start
    out_array[:, y, x] = preds[[SANDEEL, OTHER]][:, yl, xl]
    return out_array


def predict_chunk(sv, raw_labels, seabed, start_ping, end_ping, predict_fn, patch_size=(256, 256),
                  patch_overlap=20, seabed_mask=None):
    """One chunk of ``save_survey_predictions_zarr`` (save_predict.py:171-209) on in-memory arrays.

    sv [C, n_pings_total, n_range] linear (zarr orientation), raw_labels [n_pings_total, n_range],
    seabed [n_pings_total]; ``predict_fn(data[C,H,W] dB) -> softmax [3,H,W]``.
    Returns out_array [2, n_range, end_ping - start_ping] float64 (zeros where nothing is written).
    """
    n_total, n_range = sv.shape[1], sv.shape[2]
    grid = get_data_grid(n_range, int(seabed[start_ping:end_ping].max()), start_ping, end_ping, patch_size,
                         patch_overlap)
    lo = max(0, grid[0, 1] - patch_size[1] // 2)                   # dataset.py:175-177
    hi = min(n_total, grid[-1, 1] + patch_size[1] // 2)
    data = sv[:, lo:hi].swapaxes(1, 2)                             # [C, H, pings]
    labels = raw_labels[start_ping:end_ping].T                     # [H, chunk pings]
    out = np.zeros([2, n_range, end_ping - start_ping])
    for c in grid:
        d = crop(data, (c[0], c[1] - lo), patch_size, 0)
        d, nonfinite0 = data_transform(d)          # stays in the reader's dtype (float32 on the preload path)
        lab = patch_labels(labels, {"local": (c[0], c[1] - start_ping), "global": (c[0], c[1])},
                           patch_size, seabed, n_range, patch_overlap, nonfinite0, seabed_mask=seabed_mask)
        preds = predict_fn(d.astype(np.float32))
        fill_out_array(out, preds, lab, c, start_ping)
    return out, grid


def predict_echogram_memm(sv_hw, raw_labels_hw, seabed, predict_fn, patch_size=(256, 256), patch_overlap=20):
    """``save_reader_predictions_memm`` (save_predict.py:222-265) on in-memory arrays in the memm orientation:
    sv_hw [C, n_range, n_pings] linear, raw_labels_hw [n_range, n_pings], seabed [n_pings].

    Differences from the zarr-preload flavour (``predict_chunk``), each from the reference:
      * the whole echogram is one grid, ping_start = 0 (save_predict.py:244-247, :258);
      * ``get_crop_memmap`` (dataset.py:251-287): a water column not deeper than the patch puts every centre row at
        n_range // 2; non-finite samples are zeroed in the crop itself, so ``remove_nan_inf`` never touches labels;
      * ``define_data_transform_test`` (transforms.py:57-64) ends with ``set_data_border_value``
        (set_data_border_value.py:20-23): data := 0.0 (after the dB transform) wherever the TRANSFORMED label is -100
        -- outside the echogram, and at annotations that ``convert_label_indexing`` maps to ignore;
      * the seabed mask comes from ``Echogram.get_seabed_mask`` (absolute rows >= seabed + 10);
      * predictions are rounded to float16 before they are written (save_predict.py:252).
    Returns out_array [2, n_range, n_pings] float64."""
    n_range, n_pings = sv_hw.shape[1], sv_hw.shape[2]
    grid = get_data_grid(n_range, int(np.max(seabed)), 0, n_pings, patch_size, patch_overlap)
    out = np.zeros([2, n_range, n_pings])
    for c in grid:
        c = np.array(c)
        if n_range <= patch_size[0]:
            c[0] = n_range // 2
        d = crop(sv_hw, c, patch_size, 0)
        d = np.where(np.isfinite(d), d, d.dtype.type(0))
        lab = patch_labels(raw_labels_hw, {"local": (c[0], c[1]), "global": (c[0], c[1])}, patch_size, seabed,
                           n_range, patch_overlap, None, seabed_rule="memm")
        db, _ = data_transform(d)
        db[:, lab == LABEL_BOUNDARY_VAL] = 0.0
        preds = predict_fn(db.astype(np.float32)).astype(np.float16)
        fill_out_array(out, preds, lab, c, 0)
    return out


META_KEYS = ("portion_year", "portion_day", "time_diff", "depth_rel", "depth_abs_surface", "depth_abs_seabed")


def meta_planes(centre, window_size, meta_channels, portion_year, portion_day_vector, time_vector_diff, seabed):
    """The metadata planes of one crop: the ``meta`` half of ``get_crop_memmap`` (batch/dataset.py:288-351), float64
    [Cm, H, W].  ``centre`` = (range idx, ping idx) AFTER the water-column adjustment of dataset.py:256-258;
    ``meta_channels``: dict of the six booleans.  Note the index ranges: arange(c - w // 2, c + w // 2) -- one pixel up
    and left of the data crop's grid (patch_offsets above) -- and the clamping of the vector indices (< 0 -> 0,
    >= size -> last element)."""
    H, W = window_size
    meta = []
    if meta_channels["portion_year"]:
        meta.append(np.full((H, W), float(portion_year)))
    if meta_channels["portion_day"]:
        i = int(centre[1])
        i = 0 if i < 0 else (-1 if i >= portion_day_vector.size else i)
        t = portion_day_vector[i]
        meta.append(np.full((H, W), np.sin(2 * np.pi * t)))
        meta.append(np.full((H, W), np.cos(2 * np.pi * t)))
    cols = np.arange(centre[1] - W // 2, centre[1] + W // 2)
    rows = np.arange(centre[0] - H // 2, centre[0] + H // 2)
    if meta_channels["time_diff"]:
        idx = cols.copy()
        idx[idx < 0] = 0
        idx[idx >= time_vector_diff.size] = -1
        meta.append(time_vector_diff[idx].reshape(1, -1) * np.ones((H, 1)))
    if any(meta_channels[k] for k in ("depth_rel", "depth_abs_surface", "depth_abs_seabed")):
        idx = cols.copy()
        idx[idx < 0] = 0
        idx[idx >= seabed.size] = -1
        sb = seabed[idx].reshape(1, -1)
        if meta_channels["depth_rel"]:
            with np.errstate(divide="ignore", invalid="ignore"):
                meta.append(rows.reshape(-1, 1) / sb)
        if meta_channels["depth_abs_surface"]:
            meta.append(rows.reshape(-1, 1) * np.ones((1, W)) / H)
        if meta_channels["depth_abs_seabed"]:
            meta.append((sb - rows.reshape(-1, 1)) / H)
    return np.stack(meta, 0)
