"""In-memory stand-in for the reference's zarr reader (crimac_unet/data/data_reader.py:510-1120,
not importable here: needs xarray/zarr).  Test/bench infrastructure: implements the five members the
hot path's callers use: shape/data_format/time_vector/range_vector/name,
get_data_slice, get_label_slice, get_seabed, get_seabed_mask -- with the real reader's semantics
(data [freq, ping, range]; labels [ping, range]; seabed mask = 1 below the seabed, ``seabed_pad``
shifting the mask down INSIDE the requested slice, data_reader.py:837-841)."""
import numpy as np


class _Val:
    def __init__(self, v):
        self.values = v

    def max(self):
        return _Val(np.max(self.values))


class FakeZarrReader:
    data_format = "zarr"

    def __init__(self, sv, labels, seabed, name="fake_survey", mask=None, boxes=None):
        # boxes [n, 4] = (y0, y1, x0, x1) school bounding boxes: the real zarr reader has no get_object_bounding_boxes
        # (only the memmap Echogram does, data_reader.py:404); a reader that offers it can serve eval_mode 'region' / 'trace'
        if boxes is not None:
            self._boxes = np.asarray(boxes).astype(int)
            self.get_object_bounding_boxes = lambda: self._boxes.copy()
        self.sv = sv                    # [C, pings, range] linear sv, float32
        self.labels = labels            # [pings, range] raw species labels
        self.seabed = seabed            # [pings] seabed range index
        # mask [pings, range]: the stored `bottom_range` array (1 below the seabed) is then the authority and the
        # seabed vector is its argmax, as in the real reader (data_reader.py:837, :864-865)
        self.mask = None if mask is None else np.asarray(mask)
        if self.mask is not None:
            self.seabed = self.mask.argmax(axis=1).astype(np.int64)
        self.shape = (sv.shape[1], sv.shape[2])
        self.time_vector = np.arange(sv.shape[1])
        self.range_vector = np.arange(sv.shape[2]) * 0.19
        self.name = name
        self.objects = []

    def get_data_slice(self, idx_ping, n_pings, idx_range=None, n_range=None, frequencies=None,
                       drop_na=False, return_numpy=True):
        return self.sv[:, idx_ping:idx_ping + n_pings].copy()

    def get_label_slice(self, idx_ping, n_pings, idx_range=None, n_range=None, drop_na=False,
                        categories=None, return_numpy=True, correct_transducer_offset=False, mask=True):
        return self.labels[idx_ping:idx_ping + n_pings].copy()

    def get_seabed(self, idx_ping, n_pings=1, idx_range=None, n_range=None, return_numpy=True):
        v = self.seabed[idx_ping:idx_ping + n_pings]
        return v.copy() if return_numpy else _Val(v)

    def get_seabed_mask(self, idx_ping, n_pings, idx_range=None, n_range=None, return_numpy=False,
                        seabed_pad=0):
        idx_range = 0 if idx_range is None else idx_range
        hi = self.shape[1] if n_range is None else idx_range + n_range
        r = np.arange(idx_range, min(hi, self.shape[1]))
        if self.mask is not None:
            m = self.mask[idx_ping:idx_ping + n_pings, r[0]:r[-1] + 1].astype(np.float64)
        else:
            m = (r[None, :] >= self.seabed[idx_ping:idx_ping + n_pings, None]).astype(np.float64)
        if seabed_pad != 0:
            out = np.zeros_like(m)
            out[:, seabed_pad:] = m[:, :-seabed_pad]
            return out
        return m


def holey_seabed_mask(seabed, n_range):
    """A stored seabed mask the seabed VECTOR cannot express: pings 100-130 have no detected bottom (all-zero column,
    ``fillna(0)`` in the real reader) and pings 300-340 have a hole of zeros below the first seabed rows."""
    m = (np.arange(n_range)[None, :] >= np.asarray(seabed)[:, None]).astype(np.uint8)
    m[100:131] = 0
    for x in range(300, 341):
        m[x, seabed[x] + 15:seabed[x] + 40] = 0
    return m


def synth_survey(n_pings=1200, n_range=600, channels=4, seed=7):
    """Synthetic survey: linear sv 10^U(-7.5,0) with a few non-finite samples, undulating seabed,
    species blobs (27 sandeel, 1 other, 12 unused species, -1 ignore)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sv = np.power(10.0, rng.uniform(-7.5, 0.0, size=(channels, n_pings, n_range))).astype(np.float32)
    bad = rng.random((n_pings, n_range)) < 2e-4
    sv[0][bad] = np.nan
    sv[2][rng.random((n_pings, n_range)) < 1e-4] = np.inf
    x = np.arange(n_pings)
    seabed = (0.75 * n_range + 0.12 * n_range * np.sin(x / 97.0) + 0.03 * n_range * np.sin(x / 13.0)).astype(np.int64)
    seabed = np.clip(seabed, 40, n_range - 5)
    labels = np.zeros((n_pings, n_range), dtype=np.int64)
    for val in (27, 1, 12, -1, 27, 1):
        for _ in range(6):
            px, py = rng.integers(0, n_pings - 40), rng.integers(0, n_range - 30)
            labels[px:px + rng.integers(8, 40), py:py + rng.integers(6, 30)] = val
    return sv, labels, seabed


def linear_predictor(data):
    """Deterministic stand-in for the network in tiling tests: softmax over 3 fixed linear maps of the
    (dB) input channels.  data [C,H,W] -> [3,H,W] float32."""
    a = np.array([[0.02, -0.01, 0.015, 0.005], [-0.015, 0.02, 0.0, 0.01], [0.005, 0.005, -0.02, 0.0]],
                 dtype=np.float32)[:, : data.shape[0]]
    z = np.tensordot(a, data.astype(np.float32), axes=(1, 0))
    z = z - z.max(0, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(0, keepdims=True)).astype(np.float32)


class FakeEchogram:
    """In-memory stand-in for the reference's memmap reader ``Echogram`` (crimac_unet/data/data_reader.py:44-508;
    it needs a directory of .dat memory maps, pickled metadata and py<3.10 idioms, so it is replaced here).  Implements
    the members the memm flavour of the hot path's callers use: ``data_format = 'memmap'``, ``shape = (range, pings)``,
    ``data_memmaps(freqs)`` -> list of [range, pings] arrays, ``label_memmap()``, ``get_seabed(idx, n)`` and
    ``get_seabed_mask`` with the Echogram semantics (data_reader.py:407-431: 1 where range index >= seabed + pad,
    in ABSOLUTE range coordinates -- unlike the zarr reader, whose pad shifts the mask inside the requested slice)."""
    data_format = "memmap"

    def __init__(self, sv_hw, labels_hw, seabed, frequencies=(18, 38, 120, 200), name="fake_echogram", boxes=None):
        self.object_bounding_boxes = np.zeros((0, 4), int) if boxes is None else np.asarray(boxes).astype(int)
        self.sv = sv_hw                 # [C, range, pings] linear sv
        self.labels = labels_hw         # [range, pings] raw species labels
        self._seabed = np.asarray(seabed).astype(np.int64)
        self.shape = (sv_hw.shape[1], sv_hw.shape[2])
        self.frequencies = list(frequencies)
        self.name = name
        self.objects = []

    def data_memmaps(self, frequencies=None):
        if frequencies is None:
            frequencies = self.frequencies
        if not isinstance(frequencies, (list, tuple, np.ndarray)):
            frequencies = [frequencies]
        return [self.sv[self.frequencies.index(f)] for f in frequencies]

    def label_memmap(self, heave=True):
        return self.labels

    def get_object_bounding_boxes(self):            # data_reader.py:404-405
        return self.object_bounding_boxes.copy()

    def get_seabed(self, idx_ping=None, n_pings=1, save_to_file=True, ignore_saved=False):
        return self._seabed[idx_ping:idx_ping + n_pings].copy()

    def get_seabed_mask(self, idx_ping=0, n_pings=None, idx_range=None, n_range=None, seabed_pad=0):
        if n_pings is None:
            n_pings = self.shape[1]
        sb = self.get_seabed(idx_ping, n_pings) + seabed_pad
        idx_range = 0 if idx_range is None else idx_range
        n_range = self.shape[0] if n_range is None else n_range
        sb = np.maximum(sb - idx_range, 0)
        return (np.arange(n_range)[:, None] >= sb[None, :]).astype(np.float64)
