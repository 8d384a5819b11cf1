"""Deterministic synthetic echograms, labels and name-keyed weights.

There is no network on the build or GPU boxes, so benchmarks and parity tests run on
synthetic data: linear sv = 10^U(-7.5, 0) per pixel, passed through the
reference's dB transform (crimac_unet/batch/data_transforms/db_with_limits.py:20-24:
10*log10(x + 1e-10) clamped to [-75, 0]); labels drawn from {0, 1, 2, -100}.

Weights come from a generator keyed on the ``state_dict`` key *name* (crc32 of the name +
seed -> numpy PCG64), so the same tensors are regenerated bit-identically in the build
container (where golden fixtures are captured from the imported reference) and on the GPU box
(where the reference cannot travel), independent of the torch version.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np
import torch

LABEL_IGNORE_VAL = -100  # crimac_unet/constants.py:25


def unet_state_shapes(n_classes=3, in_channels=4, depth=5, start_filts=64, meta_in_channels=0):
    """Key -> shape of ``UNet_Baseline.state_dict()`` (crimac_unet/models/unet.py:200-289).

    Key order follows module registration order in the reference: down_convs, up_convs, conv_final.
    """
    shapes = OrderedDict()

    def bn(prefix, c):
        shapes[prefix + ".weight"] = (c,)
        shapes[prefix + ".bias"] = (c,)
        shapes[prefix + ".running_mean"] = (c,)
        shapes[prefix + ".running_var"] = (c,)
        shapes[prefix + ".num_batches_tracked"] = ()

    outs = in_channels
    for i in range(depth):
        ins = in_channels if i == 0 else outs
        outs = start_filts * (2 ** i)
        p = f"down_convs.{i}.main."
        shapes[p + "0.weight"] = (outs, ins, 3, 3)
        shapes[p + "0.bias"] = (outs,)
        bn(p + "1", outs)
        shapes[p + "3.weight"] = (outs, outs, 3, 3)
        shapes[p + "3.bias"] = (outs,)
        bn(p + "4", outs)
    for i in range(depth - 1):
        ins = outs
        outs = ins // 2
        p = f"up_convs.{i}."
        shapes[p + "upconv.weight"] = (ins, outs, 2, 2)
        shapes[p + "upconv.bias"] = (outs,)
        shapes[p + "conv1.weight"] = (outs, 2 * outs, 3, 3)
        shapes[p + "conv1.bias"] = (outs,)
        shapes[p + "conv2.weight"] = (outs, outs, 3, 3)
        shapes[p + "conv2.bias"] = (outs,)
        bn(p + "bn1", outs)
        bn(p + "bn2", outs)
    if meta_in_channels == 0:
        shapes["conv_final.weight"] = (n_classes, outs, 1, 1)
        shapes["conv_final.bias"] = (n_classes,)
    else:
        # UNet_LateMetInject (unet.py:346-391): conv_final = conv1x1(65, 3), then the metadata perceptron
        shapes["conv_final.weight"] = (3, 65, 1, 1)
        shapes["conv_final.bias"] = (3,)
        p = "post_processing_weights.main."
        for idx, (o, i) in (("0", (32, meta_in_channels)), ("2", (32, 32)), ("4", (1, 32))):
            shapes[p + idx + ".weight"] = (o, i)
            shapes[p + idx + ".bias"] = (o,)
    return shapes


def _is_bn_key(key: str) -> bool:
    parts = key.split(".")
    if parts[0] == "down_convs":
        return parts[3] in ("1", "4")
    if parts[0] == "up_convs":
        return parts[2].startswith("bn")
    return False


def synth_tensor(key: str, shape, seed: int = 0) -> np.ndarray:
    """Deterministic values for one ``state_dict`` entry.

    Convolution weights/biases: U(-1/sqrt(fan_in), +1/sqrt(fan_in)) -- the scale torch's default
    init gives (the reference never calls reset_params).  BatchNorm affine
    and running statistics are made non-trivial so eval-mode folding is actually exercised.
    """
    rng = np.random.Generator(np.random.PCG64((zlib.crc32(key.encode()) + 7919 * seed) & 0xFFFFFFFF))
    leaf = key.split(".")[-1]
    if leaf == "num_batches_tracked":
        return np.zeros((), dtype=np.int64)
    if _is_bn_key(key):
        if leaf == "weight":
            return rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        if leaf == "bias":
            return rng.uniform(-0.2, 0.2, size=shape).astype(np.float32)
        if leaf == "running_mean":
            return rng.normal(0.0, 0.1, size=shape).astype(np.float32)
        if leaf == "running_var":
            return rng.uniform(0.5, 1.5, size=shape).astype(np.float32)
        raise KeyError(key)
    # convolution / transposed convolution
    if leaf == "weight":
        if len(shape) == 2:  # nn.Linear [out, in] (metadata perceptron)
            fan_in = shape[1]
        elif "upconv" in key:  # ConvTranspose2d weight [Cin, Cout, 2, 2]: torch fan_in = Cout*k*k
            fan_in = shape[1] * shape[2] * shape[3]
        else:
            fan_in = shape[1] * shape[2] * shape[3]
        bound = 1.0 / np.sqrt(fan_in)
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)
    if leaf == "bias":
        # bound of the owning conv is not known from the bias shape alone; use 1/sqrt(9*C)
        bound = 1.0 / np.sqrt(9.0 * max(int(shape[0]), 1))
        return rng.uniform(-bound, bound, size=shape).astype(np.float32)
    raise KeyError(key)


def synth_state_dict(n_classes=3, in_channels=4, depth=5, start_filts=64, seed=0, meta_in_channels=0):
    """Full deterministic ``state_dict`` (torch CPU tensors) for ``UNet_Baseline`` (``meta_in_channels`` > 0:
    for ``UNet_LateMetInject``)."""
    sd = OrderedDict()
    for k, shp in unet_state_shapes(n_classes, in_channels, depth, start_filts, meta_in_channels).items():
        sd[k] = torch.from_numpy(np.ascontiguousarray(synth_tensor(k, shp, seed)))
    return sd


def db_with_limits(x: np.ndarray) -> np.ndarray:
    """dB transform of the reference (db_with_limits.py:20-24): 10*log10(x+1e-10) in [-75, 0]."""
    out = 10.0 * np.log10(x + 1e-10)
    out[out > 0] = 0
    out[out < -75] = -75
    return out


def synth_echogram_batch(batch, channels=4, height=256, width=256, seed=1):
    """[B,C,H,W] float32 model input in dB (linear sv = 10^U(-7.5,0) through db_with_limits)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.uniform(-7.5, 0.0, size=(batch, channels, height, width))
    return db_with_limits(np.power(10.0, u)).astype(np.float32)


def synth_labels(batch, height=256, width=256, seed=2, p=(0.90, 0.04, 0.04, 0.02)):
    """[B,H,W] int16 labels ~ Categorical{0,1,2,-100} with probabilities ``p`` (iid per pixel)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    vals = np.array([0, 1, 2, LABEL_IGNORE_VAL], dtype=np.int16)
    idx = rng.choice(4, size=(batch, height, width), p=np.asarray(p))
    return vals[idx]


class SyntheticSurveyReader:
    """In-memory survey with the five reader members the tiled-inference path uses (the reference's zarr reader,
    crimac_unet/data/data_reader.py:510-1120: ``shape``, ``get_data_slice`` -> [freq, ping, range] linear sv,
    ``get_label_slice`` -> [ping, range], ``get_seabed``, ``get_seabed_mask`` = 1 below the seabed).

    BASELINE configs[3]: sv [4, n_pings, 1024] fp32 linear, 10^U(-7.5, 0), flat seabed index 900.
    One ``block``-ping random block is tiled along the ping axis (generating 268 M random floats would dominate
    the benchmark's start-up); every chunk still moves and transforms its own bytes."""
    data_format = "zarr"

    class _Val:
        def __init__(self, v):
            self.values = v

        def max(self):
            return SyntheticSurveyReader._Val(np.max(self.values))

    def __init__(self, n_pings=65536, n_range=1024, channels=4, seabed_index=900, block=4096, seed=1, schools=0,
                 bad_frac=0.0):
        """``schools`` > 0: that many annotated schools per ``block`` pings -- elliptic regions of species id 27 (sandeel),
        1 (other), 12 and 5000 (species the model does not use) whose pixels in the LAST frequency channel lie mostly
        between the refine thresholds [1e-7, 1e-4] (refine_label_boundary.py:27), with weak holes and rims, so the label
        refinement has its real work to do; ``bad_frac``: that share of NaN (channel 0) and Inf (last channel) samples."""
        rng = np.random.Generator(np.random.PCG64(seed))
        nb = min(block, n_pings)
        blk = np.power(10.0, rng.uniform(-7.5, 0.0, size=(channels, nb, n_range))).astype(np.float32)
        lab_blk = np.zeros((nb, n_range), dtype=np.int16)
        if schools > 0:
            blk[-1] = np.power(10.0, rng.uniform(-9.0, -2.0, size=(nb, n_range))).astype(np.float32)
            for k in range(schools):
                cx, cy = int(rng.integers(0, nb)), int(rng.integers(0, max(min(seabed_index, n_range), 1)))
                rx, ry = int(rng.integers(6, 70)), int(rng.integers(4, 40))
                x0, x1, y0, y1 = max(cx - rx, 0), min(cx + rx + 1, nb), max(cy - ry, 0), min(cy + ry + 1, n_range)
                xx, yy = np.mgrid[x0:x1, y0:y1]
                blob = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2 <= 1.0
                lab_blk[x0:x1, y0:y1][blob] = (27, 1, 27, 12, 1, 27, 5000)[k % 7]
                strong = blob & (rng.random(blob.shape) < 0.8)
                blk[-1, x0:x1, y0:y1][strong] = np.power(10.0, rng.uniform(-6.9, -4.1, size=int(strong.sum()))).astype(np.float32)
        if bad_frac > 0:
            blk[0][rng.random((nb, n_range)) < bad_frac] = np.nan
            blk[-1][rng.random((nb, n_range)) < bad_frac] = np.inf
        reps = -(-n_pings // blk.shape[1])
        self.sv = np.ascontiguousarray(np.tile(blk, (1, reps, 1))[:, :n_pings])
        self.labels = np.ascontiguousarray(np.tile(lab_blk, (reps, 1))[:n_pings])
        self.seabed = np.full(n_pings, seabed_index, dtype=np.int64)
        self._mask = None
        self.shape = (n_pings, n_range)
        self.time_vector = np.arange(n_pings)
        self.range_vector = np.arange(n_range) * 0.19
        self.name = "synthetic_survey"

    def get_data_slice(self, idx_ping, n_pings, idx_range=None, n_range=None, frequencies=None, drop_na=False,
                       return_numpy=True):
        return self.sv[:, idx_ping:idx_ping + n_pings]

    def get_label_slice(self, idx_ping, n_pings, idx_range=None, n_range=None, drop_na=False, categories=None,
                        return_numpy=True, correct_transducer_offset=False, mask=True):
        return self.labels[idx_ping:idx_ping + n_pings]

    def get_seabed(self, idx_ping, n_pings=1, idx_range=None, n_range=None, return_numpy=True):
        v = self.seabed[idx_ping:idx_ping + n_pings]
        return v.copy() if return_numpy else SyntheticSurveyReader._Val(v)

    def get_seabed_mask(self, idx_ping, n_pings, idx_range=None, n_range=None, return_numpy=False, seabed_pad=0):
        idx_range = 0 if idx_range is None else idx_range
        hi = self.shape[1] if n_range is None else idx_range + n_range
        if self._mask is None:
            # stored like the reader's `bottom_range` array (1 below the seabed), not rebuilt per request
            self._mask = (np.arange(self.shape[1])[None, :] >= self.seabed[:, None]).astype(np.uint8)
        m = self._mask[idx_ping:idx_ping + n_pings, idx_range:min(hi, self.shape[1])]
        if seabed_pad != 0:
            out = np.zeros_like(m)
            out[:, seabed_pad:] = m[:, :-seabed_pad]
            return out
        return m


def synth_metadata(batch, channels=7, height=256, width=256, seed=3):
    """[B,Cm,H,W] float32 metadata planes in the ranges the reference Dataset produces (dataset.py:288-351):
    portion of year (constant plane in [0,1]), sin / cos of the time of day, time difference along pings, relative
    depth, depth below surface and above seabed along range (order of [0, 1])."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((batch, channels, height, width), dtype=np.float32)
    r = (np.arange(height, dtype=np.float32) / height)[None, :, None]
    for c in range(channels):
        kind = c % 7
        if kind == 0:
            out[:, c] = rng.uniform(0, 1, size=(batch, 1, 1))
        elif kind in (1, 2):
            t = rng.uniform(0, 1, size=(batch, 1, 1))
            out[:, c] = np.sin(2 * np.pi * t) if kind == 1 else np.cos(2 * np.pi * t)
        elif kind == 3:
            out[:, c] = rng.uniform(0.5, 1.5, size=(batch, 1, width))
        elif kind == 4:
            out[:, c] = r / rng.uniform(0.6, 1.0, size=(batch, 1, width))
        elif kind == 5:
            out[:, c] = r * np.ones((batch, 1, width), dtype=np.float32)
        else:
            out[:, c] = rng.uniform(0.6, 1.0, size=(batch, 1, width)) - r
    return out


def raw_crop(reader, centre, window_size, dtype=np.float32):
    """One RAW crop around ``centre`` = (range idx, ping idx) as the reference's ``get_crop`` hands it to the transform
    chain (batch/dataset.py:358-407 for zarr readers, :254-287 for memmap ones): ``data`` [C, H, W] LINEAR sv (H = range,
    W = pings; patch pixel p <-> data coordinate centre - n // 2 + 1 + p, utils/np.py:378-380), 0 outside the survey and
    where the reader holds NaN; ``labels`` [H, W] raw annotation ids, -100 (LABEL_BOUNDARY_VAL) outside the survey.
    ``dtype``: float32 is what the memmap flavour returns (the dtype of the .dat files), float64 what the zarr flavour
    returns (its output array is ``np.ones(...) * 0``, dataset.py:361)."""
    H, W = int(window_size[0]), int(window_size[1])
    n_pings, n_range = reader.shape
    y0, x0 = int(centre[0]) - H // 2 + 1, int(centre[1]) - W // 2 + 1
    ys, ye = max(y0, 0), min(y0 + H, n_range)
    xs, xe = max(x0, 0), min(x0 + W, n_pings)
    n_ch = reader.sv.shape[0] if hasattr(reader, "sv") else 4
    data = np.zeros((n_ch, H, W), dtype=dtype)
    labels = np.full((H, W), LABEL_IGNORE_VAL, dtype=np.int16)
    if ye > ys and xe > xs:
        sv = reader.get_data_slice(idx_ping=xs, n_pings=xe - xs, idx_range=ys, n_range=ye - ys, return_numpy=True)
        sv = np.asarray(sv)[:, :, ys:ye] if np.asarray(sv).shape[2] != ye - ys else np.asarray(sv)
        lab = np.asarray(reader.get_label_slice(idx_ping=xs, n_pings=xe - xs, idx_range=ys, n_range=ye - ys,
                                                return_numpy=True))
        lab = lab[:, ys:ye] if lab.shape[1] != ye - ys else lab
        blockd = data[:, ys - y0:ye - y0, xs - x0:xe - x0]
        blockd[...] = sv.swapaxes(1, 2)
        nan = np.isnan(blockd)                                   # (dataset.py:402: NaN -> 0)
        if nan.any():
            blockd[nan] = 0
        labels[ys - y0:ye - y0, xs - x0:xe - x0] = lab.T
    return data, labels


class RawCropDataset:
    """Map-style dataset of RAW training crops over an in-memory survey: what the reference's ``Dataset`` yields when it
    is built for the on-GPU transform chain -- ``augmentation_function=None, label_transform_function=None,
    data_transform_function=None`` (batch/dataset.py:76-110 then applies nothing): a random patch centre, the crop
    gather, and the batch dict ``{'data': linear sv [C, H, W], 'labels': raw annotation ids int16 [H, W],
    'center_coordinates': int64 [2]}``.  ``transform(data, labels, index) -> (data, labels)`` runs in the worker on the
    raw crop (bench.py's host-chain baseline plugs the reference's transform chain in there).

    The centre of sample ``i`` is a function of (seed, i) alone -- the same crops whatever the number of workers -- drawn
    uniformly over the survey (crops at its rim carry -100 labels, like the reference's)."""

    def __init__(self, reader, window_size, n_samples, seed=0, dtype=np.float32, transform=None):
        self.reader, self.window_size, self.n_samples = reader, tuple(window_size), int(n_samples)
        self.seed, self.dtype, self.transform = int(seed), dtype, transform

    def __len__(self):
        return self.n_samples

    def centre(self, index):
        rng = np.random.Generator(np.random.PCG64([self.seed, int(index)]))
        n_pings, n_range = self.reader.shape
        return np.array([int(rng.integers(0, n_range)), int(rng.integers(0, n_pings))], dtype=np.int64)

    def __getitem__(self, index):
        c = self.centre(index)
        data, labels = raw_crop(self.reader, c, self.window_size, self.dtype)
        if self.transform is not None:
            data, labels = self.transform(data, labels, int(index))
        return {"data": data, "labels": np.asarray(labels).astype(np.int16), "center_coordinates": c}
