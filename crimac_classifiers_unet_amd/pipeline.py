"""``SegPipe`` / ``SegPipeUNet`` for MI355X: the reference's training / prediction pipeline surface.

Mirrors crimac_unet/pipeline_train_predict/pipeline.py (class SegPipe :39-376, SegPipeUNet :379-410,
get_in_channels :413-425): same constructor keywords (so ``SegPipeUNet(**yaml_config)`` works with
the reference's ``pipeline_config.yaml`` keys unchanged), same public attributes and methods, same
batch dictionaries (``{'data','labels','center_coordinates'}`` from the reference ``Dataset``) and
the same ``state_dict`` checkpoints (``best.pt`` / ``last.pt``).

The compute behind ``train_model`` / ``predict_batch`` is the HIP engine; with ``torch.distributed``
initialised (one process per GPU) ``train_model`` all-reduces gradients over RCCL.
New optional keywords (all defaulted, so the baseline yaml still works): ``precision`` -- the TRAINING precision
('bf16' | 'fp16' | 'h3p' | 'f32h3' | 'f32x3' | 'f32x6'), ``infer_precision`` -- the precision of every eval-mode forward
(``predict_batch``, validation, ``save_predict`` / ``evaluate`` flows; default 'h3p': logits within 1e-5 of the reference's,
identical argmax masks), ``loss_flush`` (how often queued train losses are handed to the logger).
"""
from __future__ import annotations

from pathlib import Path

import numpy as np
import torch

from . import parallel
from .train_ops import ExponentialLR, SGDMomentum, WeightedCrossEntropy
from .unet import UNet_Baseline, UNet_LateMetInject

# crimac_unet/constants.py:20-33
BACKGROUND, SANDEEL, OTHER = 0, 1, 2
LABEL_IGNORE_VAL = -100
LABEL_BOUNDARY_VAL = -100
LABEL_OVERLAP_VAL = -70
LABEL_SEABED_MASK_VAL = -50
LABEL_REFINE_BOUNDARY_VAL = -30
LABEL_UNUSED_SPECIES = -10

CE_CLASS_WEIGHTS = (10.0, 300.0, 250.0)     # pipeline.py:135


def _tqdm(it, **kw):
    try:
        from tqdm import tqdm
        return tqdm(it, **kw)
    except Exception:  # pragma: no cover
        return it


class SegPipe:
    """Segmentation training-prediction pipeline (reference pipeline.py:39-376)."""

    def __init__(self, checkpoint_dir, data_mode, frequencies, patch_size, loss_type, lr, lr_reduction,
                 lr_step, momentum, batch_size, num_workers, iterations, test_iter, log_step,
                 save_model_params, meta_channels, late_meta_inject, eval_mode, experiment_name,
                 precision="bf16", infer_precision="h3p", loss_flush=50, gpu_augment=False, random_seed=0,
                 gpu_metrics=False, gpu_label_transform=False, sync_bn=False, pin_batches=True,
                 release_batch_pages=True, collate_float32=True, **kwargs):
        assert not (save_model_params and (checkpoint_dir is None))
        self.model = None
        self.model_is_loaded = False

        self.data_mode = data_mode
        self.frequencies = frequencies
        if self.frequencies == "all":
            self.frequencies = [18, 38, 120, 200]
        if self.data_mode == "zarr":
            self.frequencies = sorted([freq for freq in self.frequencies])
        self.window_size = patch_size

        self.device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
        self.loss_type = loss_type
        self.lr, self.lr_reduction, self.momentum, self.lr_step = lr, lr_reduction, momentum, lr_step
        self.iterations, self.test_iter, self.log_step = iterations, test_iter, log_step
        self.batch_size, self.num_workers = batch_size, num_workers
        self.save_model_params = save_model_params
        self.checkpoint_dir = checkpoint_dir

        self.meta_channels = meta_channels
        self.late_meta_inject = late_meta_inject
        self.use_metadata = len(self.meta_channels) > 0

        self.model_name = experiment_name
        self.eval_mode = eval_mode
        self.best_F1_val = -np.inf

        # training runs in `precision` (default bf16: throughput); everything that PREDICTS -- predict_batch, the
        # validation loops, the save_predict / evaluate flows -- runs in `infer_precision` (default h3p: the precision
        # that meets the parity bar of the reference's fp32 path, pipeline.py:205-219).  Asking for a 16-bit inference
        # precision is allowed (2.2x the throughput) and announced once, see _warn_infer_precision
        self.precision = precision
        self.infer_precision = infer_precision or precision
        self._warned_infer = False
        # pin_batches: train_model stages the DataLoader's batches through a pinned ring on a copy stream (H2D of step
        # i + 1 under step i); False = the reference's in-line `.to(device)` (pipeline.py:163-164)
        self.pin_batches = bool(pin_batches)
        # release_batch_pages (with pin_batches): once a worker-collated batch has been copied to pinned memory its shared-
        # memory pages are handed back with madvise(MADV_REMOVE) off the training thread (staging.release_shared_pages) --
        # the batch's tensors read as zeros afterwards.  train_model never hands those batches out; set False if a
        # DataLoader wrapper of yours keeps references to the batches it yields
        self.release_batch_pages = bool(release_batch_pages)
        # collate_float32: train_model swaps the training DataLoader's `default_collate` for one that casts the float64 crops
        # of the zarr Dataset (batch/dataset.py:361) to float32 before stacking them -- the values of the reference's
        # `.float()` (pipeline.py:163) at half the bytes through collate, hand-over and upload; other collate functions stay
        self.collate_float32 = bool(collate_float32)
        self.loss_flush = max(int(loss_flush), 1)
        # gpu_augment: the train Dataset hands RAW linear-sv crops (augmentation_function=None,
        # data_transform_function=None) and add_noise / flip_x_axis / remove_nan_inf / db_with_limits
        # run fused on the GPU (BASELINE configs[4]); default keeps the reference's numpy workers
        self.gpu_augment = bool(gpu_augment)
        # gpu_label_transform (with gpu_augment): the train Dataset also gets label_transform_function=None
        # and hands RAW annotation ids; refine_label_boundary (threshold channel = last frequency, thresholds
        # [1e-7, 1e-4], batch/transforms.py:74) + convert_label_indexing run on the GPU
        # sync_bn (N > 1): BatchNorm statistics over the global batch (engine.sync_bn); default keeps them per
        # rank like torch DDP
        self.sync_bn = bool(sync_bn)
        self.gpu_label_transform = bool(gpu_label_transform)
        if self.gpu_label_transform and not self.gpu_augment:
            raise ValueError("gpu_label_transform needs gpu_augment (the transform runs on the augmented crop)")
        if self.gpu_augment and self.use_metadata and not self.late_meta_inject:
            raise NotImplementedError("gpu_augment with metadata planes as extra INPUT channels (late_meta_inject=False): the "
                                      "on-GPU data transform would take them for sv; use late_meta_inject=True or the host "
                                      "transforms")
        self.random_seed = int(random_seed)
        # gpu_metrics: in-training validation builds the PR curve / F1 from GPU histograms instead of
        # shipping every pixel's probability to sklearn (same numbers; the logger gets no pr_curve)
        self.gpu_metrics = bool(gpu_metrics)

    # ------------------------------------------------------------------------------------------
    def _warn_infer_precision(self):
        """One line, once, when predictions are about to be made in a precision that does not reproduce the
        reference's argmax masks."""
        why = UNet_Baseline.PARITY_FAILING.get(self.infer_precision)
        if why and not self._warned_infer:
            import warnings
            warnings.warn(f"infer_precision={self.infer_precision!r}: predicted class masks differ from the fp32 reference "
                          f"pipeline at {why} of the golden crop; use infer_precision='h3p' (the default) for "
                          "reference-identical masks", stacklevel=3)
        self._warned_infer = True

    def load_model_params(self, checkpoint_path=None):
        """Load a ``state_dict`` checkpoint (reference pipeline.py:109-130)."""
        self._warn_infer_precision()
        if self.model_is_loaded:
            return
        assert self.model is not None
        if checkpoint_path is None:
            checkpoint_path = Path(self.checkpoint_dir, "best.pt")
        with torch.no_grad():
            self.model.to(self.device)
            self.model.load_state_dict(torch.load(checkpoint_path, map_location=self.device))
            self.model.eval()
        print("loaded model", checkpoint_path)
        self.model_is_loaded = True

    def get_criterion(self):
        """Weighted CE, weight [10, 300, 250] (reference pipeline.py:132-141)."""
        if self.loss_type == "CE":
            return WeightedCrossEntropy(CE_CLASS_WEIGHTS).to(self.device)
        raise ValueError("`loss_type` not recognized")

    # ------------------------------------------------------------------------------------------
    def train_model(self, dataloader_train, dataloader_test, logger=None):
        """Training loop of the reference (pipeline.py:144-203) on the fused HIP step.

        Each iteration = forward + weighted CE + backward (+ RCCL gradient all-reduce) + SGD.
        The reference logs ``loss.item()`` every step (a host sync, :181); here losses stay on
        the device and are flushed to ``logger`` every ``loss_flush`` steps with their original
        ``global_step``.
        """
        assert not Path(self.checkpoint_dir).is_dir() if self.checkpoint_dir is not None else True, f"""
            Attempting to train a model that already exists: {str(self.checkpoint_dir)}
            Use a different model name or delete the saved model params file
        """
        self.model.to(self.device)
        optimizer = SGDMomentum(self.model, lr=self.lr, momentum=self.momentum)
        scheduler = ExponentialLR(optimizer, gamma=self.lr_reduction)
        criterion = self.get_criterion()
        engine = self.model.engine
        engine.sync_bn = self.sync_bn
        engine.loss_scale_check_every = 0       # (the dynamic loss scale is re-evaluated in flush() below)
        grad_sync = parallel.GradSync()
        is_rank0 = parallel.env_world()[1] == 0
        pending = []

        def flush():
            engine.update_loss_scale()          # (fp16 precision only; this is where the loop synchronises anyway)
            if pending:
                # the logged loss is the loss of the GLOBAL batch: sum of (sum w*nll, sum w) over ranks -- one small
                # all-reduce per flush, not per step
                sums = parallel.all_reduce_scalars(torch.stack([s for _, s in pending]))
                if logger is not None:
                    vals = (sums[:, 0] / sums[:, 1]).float().cpu().tolist()
                    for (step, _), v in zip(pending, vals):
                        logger.add_scalar(tag="train/loss", scalar_value=v, global_step=step)
            pending.clear()

        def batches():
            """(i, data float32 on the device, labels on the device): the reference's in-step ``.float().to(device)``
            (pipeline.py:163-164), staged one batch ahead through pinned memory and a copy stream (staging.py) so that
            the upload of step i + 1 runs under step i; ``pin_batches: False`` keeps the in-line copy."""
            if self.collate_float32:
                from .staging import use_collate_float32
                use_collate_float32(dataloader_train)
            if self.pin_batches and self.device.type == "cuda":
                from .staging import BatchStager
                for i, x, lab, _ in BatchStager(dataloader_train, self.device, yield_batch=False,
                                                stats=getattr(self, "stager_stats", None),
                                                release_pages=self.release_batch_pages):
                    yield i, x, lab
                return
            for i, batch in enumerate(dataloader_train):
                x = batch["data"]
                if x.dtype != torch.float32:
                    x = x.float()
                # BLOCKING copies, like the reference's: the sources are pageable (and `x.float()` is a temporary) -- an
                # asynchronous copy would still be reading them when the next batch reuses the memory (measured: with
                # DataLoader workers the first steps trained on the wrong crops)
                yield i, x.to(self.device), batch["labels"].to(self.device)

        for i, inputs_train, labels_train in _tqdm(batches(), desc="Training model",
                                                   total=len(dataloader_train), disable=not is_rank0):
            self.model.train()
            meta_train = None
            if self.late_meta_inject:            # pipeline.py:170-174: data planes | metadata planes
                nf = len(self.frequencies)
                inputs_train, meta_train = inputs_train[:, :nf], inputs_train[:, nf:]
            if self.gpu_augment:
                if i == 0 and bool((inputs_train < 0).any()):
                    # (one device -> host sync, first batch only) linear sv is non-negative; dB data is not
                    raise ValueError("gpu_augment=True, but the training Dataset yields negative values: it still applies "
                                     "its own data transform (db_with_limits).  Build it with augmentation_function=None, "
                                     "data_transform_function=None (and label_transform_function=None with "
                                     "gpu_label_transform) so that it hands RAW linear-sv crops, or switch gpu_augment off")
                rank = parallel.env_world()[1]
                # (late metadata injection: add_noise_metadata / flip_x_axis_metadata, batch/transforms.py:41-42 -- the
                # metadata planes take the flip, and the data transform is db_with_limits_scaled, :50-51)
                loss = engine.train_step_augmented(
                    inputs_train, labels_train, criterion.weight, optimizer.param_groups[0]["lr"],
                    self.momentum, seed=(self.random_seed << 40) ^ (rank << 32) ^ i, grad_sync=grad_sync,
                    refine_labels=(len(self.frequencies) - 1, 1e-7, 1e-4) if self.gpu_label_transform else None,
                    meta=meta_train)
            else:
                loss = engine.train_step(inputs_train, labels_train, criterion.weight,
                                         optimizer.param_groups[0]["lr"], self.momentum,
                                         grad_sync=grad_sync, meta=meta_train)
            pending.append((i + 1, engine.last_loss_sums))
            if len(pending) >= self.loss_flush:
                flush()

            if (i + 1) % self.log_step == 0:
                flush()
                self.validate_model_training(dataloader_test, criterion, logger, i)

            if (i + 1) % self.lr_step == 0:
                scheduler.step()
                if logger is not None:
                    for group_idx, group in enumerate(optimizer.param_groups):
                        logger.add_scalar(tag=f"learning_rate_{group_idx}", scalar_value=group["lr"],
                                          global_step=i + 1)
        flush()
        print("Training complete")
        self.model_is_loaded = True

        if self.save_model_params and is_rank0:
            Path(self.checkpoint_dir).mkdir(parents=True, exist_ok=True)
            checkpoint_path = Path(self.checkpoint_dir) / "last.pt"
            torch.save(self.model.state_dict(), checkpoint_path)
            print("Trained model parameters saved to file:", str(checkpoint_path))

    # ------------------------------------------------------------------------------------------
    def predict_batch(self, batch, return_softmax=False):
        """Eval-mode forward of one batch dict (reference pipeline.py:205-219); result stays on GPU."""
        self.model.eval()
        with torch.no_grad():
            inputs = batch["data"].float().to(self.device)
            if self.late_meta_inject:            # pipeline.py:210-216
                nf = len(self.frequencies)
                data, meta = inputs[:, :nf].contiguous(), inputs[:, nf:].contiguous()
                return self.model.predict_softmax(data, meta) if return_softmax else self.model(data, meta)
            if return_softmax:
                return self.model.predict_softmax(inputs)
            return self.model(inputs)

    # -- test-time transforms on the GPU (the validation / evaluate flows) ---------------------
    _test_source = None
    TEST_SEABED_BLOCK = 4096         # pings per block of the reader's seabed mask that is read and checked at a time

    def use_gpu_test_transform(self, reader, patch_overlap=0, seabed_pad=10, extend_size=20):
        """From here on the validation / test DataLoaders hand RAW crops: the reference ``Dataset`` built with
        ``label_transform_function=None`` and ``data_transform_function=None`` (batch/dataset.py:89-103 then applies
        nothing), i.e. ``data`` = linear sv, ``labels`` = raw annotation ids (-100 outside the data),
        ``center_coordinates`` = (range idx, ping idx).  ``define_label_transform_test`` (batch/transforms.py:81-99) and
        ``define_data_transform`` (:49-55) run on the GPU instead: ``crimac_labels_test_transform`` (+
        ``crimac_labels_extend_mask`` when ``eval_mode`` is 'region' / 'trace', :87-90, ``extend_size`` as
        define_label_transform_test's default) + ``crimac_augment_db_nhwc`` (remove_nan_inf + db_with_limits -- the scaled
        form with metadata channels, :50-51 -- straight into the first convolution's layout).
        ``reader``: the survey's zarr reader (``shape``, ``get_seabed``, ``get_seabed_mask``) the crops come from.  Its
        per-ping seabed vector is read once (4 bytes per ping); the 2-D seabed mask -- which the vector must reproduce
        exactly, ``tiled_inference.seabed_vector_or_mask`` -- is read and checked lazily, ``TEST_SEABED_BLOCK`` pings at
        a time, for the blocks the batches actually touch; a batch touching a block the vector cannot express takes the
        reader's mask per patch, as the reference does (mask_label_seabed.py:40-52).  ``None`` switches back to
        host-transformed batches."""
        if reader is None:
            self._test_source = None
            return
        if getattr(reader, "data_format", "zarr") != "zarr":
            raise NotImplementedError("use_gpu_test_transform: zarr readers only (the memmap flow's set_data_border_value "
                                      "lives in the tiled path, tiled_inference.predict_echogram_memm)")
        if self.use_metadata and not self.late_meta_inject:
            raise NotImplementedError("use_gpu_test_transform with metadata planes as extra input channels "
                                      "(late_meta_inject=False)")
        n_pings, n_range = (int(v) for v in reader.shape)
        boxes = None
        if self.eval_mode in ("region", "trace"):
            # get_extended_label_mask_for_crop (extend_label_masks.py:57-80); the reference asks the reader for
            # get_object_bounding_boxes(), which only its memmap Echogram defines (data_reader.py:404) -- a zarr reader
            # without it fails there with an AttributeError
            if not hasattr(reader, "get_object_bounding_boxes"):
                raise NotImplementedError(
                    f"eval_mode={self.eval_mode!r}: the reader has no get_object_bounding_boxes() (the reference's "
                    "get_extended_label_mask_for_crop needs it, extend_label_masks.py:67); use eval_mode='all'")
            bb = np.array(reader.get_object_bounding_boxes(), dtype=np.int64).reshape(-1, 4)
            if self.eval_mode == "region":
                bb[:, 0] -= int(extend_size)
                bb[:, 1] += int(extend_size)
            else:
                bb[:, 0] = 0
                bb[:, 1] = int(reader.shape[0])          # (the reference's `echogram.shape[0]`, :78)
            bb[:, 2] -= int(extend_size)
            bb[:, 3] += int(extend_size)
            boxes = torch.from_numpy(np.ascontiguousarray(bb.astype(np.int32))).to(self.device)
        elif self.eval_mode != "all":
            raise ValueError(f"eval_mode={self.eval_mode!r}: 'all', 'region' or 'trace' (batch/transforms.py:87)")
        sb = np.ascontiguousarray(np.asarray(reader.get_seabed(0, n_pings, return_numpy=True)).astype(np.int32))
        self._test_source = {"reader": reader, "seabed_host": sb, "seabed_dev": None, "blocks": {},
                             "boxes": boxes, "n_pings": n_pings, "n_range": n_range,
                             "overlap": int(patch_overlap), "pad": int(seabed_pad)}

    def _test_seabed(self, centres, B, W):
        """(seabed vector on the device, None) when the seabed vector reproduces the reader's mask on every block of pings
        the batch touches, else (None, per-patch mask [B * W, n_range] uint8 on the device)."""
        from .tiled_inference import seabed_vector_or_mask
        src = self._test_source
        n_pings, n_range, blk = src["n_pings"], src["n_range"], self.TEST_SEABED_BLOCK
        spans = []
        for cx in centres[:, 1].tolist():
            x0 = int(cx) - W // 2 + 1
            spans.append((max(x0, 0), min(x0 + W, n_pings), x0))
        ok = True
        for xs, xe, _ in spans:
            for k in range(xs // blk, (xe - 1) // blk + 1) if xe > xs else ():
                if k not in src["blocks"]:
                    s_, e_ = k * blk, min((k + 1) * blk, n_pings)
                    sb2, mask = seabed_vector_or_mask(src["reader"], s_, e_, n_range, src["seabed_host"], 0)
                    if sb2 is not src["seabed_host"]:            # pings without a detected bottom were patched in
                        src["seabed_host"], src["seabed_dev"] = np.ascontiguousarray(sb2), None
                    src["blocks"][k] = mask is None
                ok = ok and src["blocks"][k]
        if ok:
            if src["seabed_dev"] is None:
                src["seabed_dev"] = torch.from_numpy(src["seabed_host"]).to(self.device)
            return src["seabed_dev"], None
        m = np.zeros((B, W, n_range), dtype=np.uint8)
        for b, (xs, xe, x0) in enumerate(spans):
            if xe > xs:
                got = src["reader"].get_seabed_mask(xs, xe - xs, 0, n_range, return_numpy=True)
                m[b, xs - x0:xe - x0] = np.asarray(getattr(got, "values", got)) != 0
        return None, torch.from_numpy(m.reshape(B * W, n_range)).to(self.device)

    def _predict_raw_batch(self, batch):
        """(logits, transformed int16 labels on the device) of one batch of RAW crops (``use_gpu_test_transform``)."""
        from .hip import MASK_PER_PATCH, call, ptr
        src, dev = self._test_source, self.device
        self.model.eval()
        data = batch["data"].to(dev)
        if data.dtype != torch.float32:
            data = data.float()
        meta = None
        if self.late_meta_inject:                # pipeline.py:210-216: data planes | metadata planes
            nf = len(self.frequencies)
            data, meta = data[:, :nf], data[:, nf:].contiguous()
        data = data.contiguous()
        labels = batch["labels"].to(dev)
        if labels.dtype not in (torch.int16, torch.int32, torch.int64):
            labels = labels.long()
        labels = labels.contiguous()
        cen_host = torch.as_tensor(batch["center_coordinates"]).long().reshape(-1, 2)
        cen = cen_host.to(dev).contiguous()
        B, C, H, W = data.shape
        seabed, mask = self._test_seabed(cen_host.numpy(), B, W)
        out = torch.empty((B, H, W), dtype=torch.int16, device=dev)
        eng = self.model.infer_engine
        with torch.no_grad(), torch.cuda.device(dev):
            call("crimac_labels_test_transform", ptr(labels), labels.element_size(), ptr(data), len(self.frequencies) - 1,
                 1e-7, 1e-4, ptr(cen), ptr(seabed), 0, src["n_pings"] if seabed is not None else 0,
                 ptr(mask), MASK_PER_PATCH if mask is not None else 0, B * W if mask is not None else 0, src["n_range"],
                 src["pad"], 0, src["overlap"], ptr(out), B, C, H, W)
            if src["boxes"] is not None:         # eval_mode 'region' / 'trace'
                call("crimac_labels_extend_mask", ptr(out), ptr(data), C, ptr(cen), ptr(src["boxes"]),
                     int(src["boxes"].shape[0]), -1, B, H, W)
            # remove_nan_inf + db_with_limits (db_with_limits_scaled with metadata channels)
            x, _ = eng.augment_batch(data, None, 0, do_noise=False, do_flip=False, db_scaled=self.use_metadata)
            logits = eng.forward_nhwc(x, B, H, W, training=False, meta=meta)
        return logits, out

    def set_label_ignore_val(self, labels):
        """Reference pipeline.py:222-239 (in place, like the reference)."""
        labels[labels == LABEL_OVERLAP_VAL] = LABEL_IGNORE_VAL
        labels[labels == LABEL_REFINE_BOUNDARY_VAL] = LABEL_IGNORE_VAL
        labels[labels == LABEL_BOUNDARY_VAL] = LABEL_IGNORE_VAL
        labels[labels == LABEL_UNUSED_SPECIES] = LABEL_IGNORE_VAL
        labels[labels == LABEL_SEABED_MASK_VAL] = 0
        return labels

    def _softmax(self, logits):
        """F.softmax(logits, dim=1) (pipeline.py:269) as a HIP kernel; the product path has no torch arithmetic."""
        from .hip import call, ptr
        logits = logits.contiguous().float()
        B, C, H, W = logits.shape
        out = torch.empty_like(logits)
        with torch.cuda.device(logits.device):
            call("crimac_softmax_nchw", ptr(logits), ptr(out), B, C, H, W)
        return out

    def get_predictions_dataloader(self, dataloader, criterion=None, disable_tqdm=False):
        """Sandeel-probability vector + labels over a dataloader (reference pipeline.py:242-282)."""
        preds, labels = [], []
        sum_loss = None
        self.model.eval()
        with torch.no_grad():
            for ii, batch_test in _tqdm(enumerate(dataloader), desc="Evaluating model",
                                        total=len(dataloader), disable=disable_tqdm):
                outputs_test = self.predict_batch(batch_test, return_softmax=False)
                if criterion is not None:
                    labels_input = batch_test["labels"].long().to(self.device)
                    labels_input = self.set_label_ignore_val(labels_input)
                    loss_test = criterion(outputs_test, labels_input)
                    sum_loss = loss_test if sum_loss is None else sum_loss + loss_test
                preds_softmax = self._softmax(outputs_test)             # (the logits above feed the loss)
                preds += [preds_softmax[:, SANDEEL].to(torch.float16)]
                labels += [batch_test["labels"].numpy().ravel()]
        preds = torch.cat([p.reshape(-1) for p in preds]).cpu().numpy().astype(np.float16)
        labels = np.hstack(labels).astype(np.int8)
        mean_loss = (float(sum_loss) if sum_loss is not None else 0.0) / len(dataloader)
        return labels, preds, mean_loss

    # -- histogram form of the same metrics: no per-pixel vectors leave the GPU --------------------
    PR_BINS = 16384            # float16 bit patterns of [0, 1] are 0 .. 0x3C00

    def get_pr_histograms_dataloader(self, dataloader, criterion=None, disable_tqdm=True):
        """GPU form of get_predictions_dataloader + the masking of validate_model_training
        (pipeline.py:242-282, :317-320): returns (hist_pos, hist_neg, mean_loss) where hist_*[k] counts
        the valid pixels whose float16 sandeel probability has bit pattern k."""
        from .hip import call, ptr
        dev = self.device
        hist = torch.zeros(2, self.PR_BINS, dtype=torch.int32, device=dev)
        sum_loss = None
        world, rank, _ = parallel.env_world()
        if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
            world, rank = 1, 0
        self.model.eval()
        with torch.no_grad():
            for bi, batch in enumerate(_tqdm(dataloader, desc="Evaluating model", total=len(dataloader),
                                             disable=disable_tqdm)):
                if bi % world != rank:            # batches are dealt round-robin to the ranks
                    continue
                if self._test_source is not None:      # RAW crops: label + data transforms on the GPU
                    logits, labels = self._predict_raw_batch(batch)
                else:
                    logits = self.predict_batch(batch, return_softmax=False)
                    labels = batch["labels"].to(dev)
                    if labels.dtype not in (torch.int16, torch.int32, torch.int64):
                        labels = labels.long()
                    labels = labels.contiguous()
                if criterion is not None:
                    loss = criterion(logits, self.set_label_ignore_val(labels.clone().long()))
                    sum_loss = loss if sum_loss is None else sum_loss + loss
                B, nc, H, W = logits.shape
                call("crimac_pr_histogram", ptr(logits), nc, ptr(labels), labels.element_size(), B, H, W,
                     ptr(hist[0]), ptr(hist[1]))
        if world > 1:                          # the histograms ARE the metric's sufficient statistic: 128 KB all-reduce
            torch.distributed.all_reduce(hist)
            sl = torch.zeros(1, dtype=torch.float64, device=dev) if sum_loss is None else sum_loss.double().reshape(1)
            torch.distributed.all_reduce(sl)
            sum_loss = sl[0]
        h = hist.cpu().numpy().astype(np.int64)
        if h[:, self.PR_BINS - 1].any():      # CRIMAC_PR_NAN_BIN: sklearn raises on NaN scores as well
            raise ValueError("Input contains NaN (sandeel probabilities of the validation set)")
        mean_loss = (float(sum_loss) if sum_loss is not None else 0.0) / len(dataloader)
        return h[0], h[1], mean_loss

    @staticmethod
    def compute_evaluation_metrics_from_histograms(hist_pos, hist_neg):
        """precision_recall_curve (sklearn) + F1 (pipeline.py:284-295) from the two histograms: the same
        arrays sklearn returns for the float16 prediction vector the reference builds."""
        bins = np.nonzero((hist_pos + hist_neg) > 0)[0]
        thr = bins.astype(np.uint16).view(np.float16).astype(np.float64)       # ascending with the bits
        order = np.argsort(-thr, kind="stable")                                 # sklearn: descending scores
        tps = np.cumsum(hist_pos[bins][order]).astype(np.float64)
        fps = np.cumsum(hist_neg[bins][order]).astype(np.float64)
        ps = tps + fps
        precision = np.divide(tps, ps, out=np.zeros_like(tps), where=ps != 0)
        recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
        precision = np.hstack((precision[::-1], 1.0))
        recall = np.hstack((recall[::-1], 0.0))
        den = recall + precision
        f1 = np.divide(2 * recall * precision, den, out=np.zeros_like(den), where=(den != 0))
        return {"precision": precision, "recall": recall, "thresholds": thr[order][::-1], "F1": f1}

    def compute_evaluation_metrics(self, labels, preds):
        """PR curve and F1 (reference pipeline.py:284-295)."""
        from sklearn.metrics import precision_recall_curve
        precision, recall, thresholds = precision_recall_curve(labels, preds, pos_label=SANDEEL)
        numerator = 2 * recall * precision
        denom = recall + precision
        f1_scores = np.divide(numerator, denom, out=np.zeros_like(denom), where=(denom != 0))
        return {"precision": precision, "recall": recall, "thresholds": thresholds, "F1": f1_scores}

    def select_valid_predictions(self, labels, preds):
        labels = self.set_label_ignore_val(labels)
        idx = np.where(labels != LABEL_IGNORE_VAL)
        return labels[idx], preds[idx]

    def validate_model_training(self, dataloader_test, criterion, logger, iteration_no):
        """Reference pipeline.py:305-341."""
        multi = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
        if self.gpu_metrics or multi:
            # N > 1: validation is sharded over the ranks; only the histogram form has a small sufficient statistic
            # to exchange (the per-pixel vectors of the sklearn form would have to be gathered on one rank)
            hp, hn, loss_test = self.get_pr_histograms_dataloader(dataloader_test, criterion=criterion)
            metrics = self.compute_evaluation_metrics_from_histograms(hp, hn)
            labels = preds = None
        else:
            labels, preds, loss_test = self.get_predictions_dataloader(dataloader_test, criterion=criterion)
            preds[labels == LABEL_SEABED_MASK_VAL] = 0
            labels, preds = self.select_valid_predictions(labels=labels, preds=preds)
            metrics = self.compute_evaluation_metrics(labels=labels, preds=preds)
        F1 = metrics["F1"]
        argmax_F1 = np.argmax(F1)
        iter_step = iteration_no + 1
        if logger is not None:
            logger.add_scalar(tag="test/F1_score", scalar_value=F1[argmax_F1], global_step=iter_step)
            logger.add_scalar(tag="test/precision", scalar_value=metrics["precision"][argmax_F1],
                              global_step=iter_step)
            logger.add_scalar(tag="test/recall", scalar_value=metrics["recall"][argmax_F1],
                              global_step=iter_step)
            logger.add_scalar(tag="test/loss", scalar_value=loss_test, global_step=iter_step)
            if hasattr(logger, "add_pr_curve") and labels is not None:
                logger.add_pr_curve(tag="test/pr_curve", labels=labels, predictions=preds,
                                    global_step=iter_step)
        if F1[argmax_F1] > self.best_F1_val:
            self.best_F1_val = F1[argmax_F1]
            if self.checkpoint_dir is not None and parallel.env_world()[1] == 0:
                Path(self.checkpoint_dir).mkdir(parents=True, exist_ok=True)
                torch.save(self.model.state_dict(), Path(self.checkpoint_dir) / "best.pt")
        return metrics

    def validate_model_testing(self, dataloader, save_path_metrics, save_path_plot):
        """Test-set evaluation (reference pipeline.py:343-376): PR curve / F1 of the SANDEEL probability over the
        valid pixels of ``dataloader``; the curve is written as csv (columns as the reference's DataFrame:
        precision, recall, thresholds, F1; the last threshold is NaN) and, if asked for, drawn."""
        if not self.model_is_loaded:
            self.load_model_params()
        if self.gpu_metrics:
            hp, hn, _ = self.get_pr_histograms_dataloader(dataloader, disable_tqdm=False)
            metrics = self.compute_evaluation_metrics_from_histograms(hp, hn)
        else:
            labels, preds, _ = self.get_predictions_dataloader(dataloader, disable_tqdm=False)
            preds[labels == LABEL_SEABED_MASK_VAL] = 0
            labels, preds = self.select_valid_predictions(labels=labels, preds=preds)
            metrics = self.compute_evaluation_metrics(labels=labels, preds=preds)
        metrics["thresholds"] = np.append(np.asarray(metrics["thresholds"], dtype=np.float64), np.nan)
        if parallel.env_world()[1] == 0:
            if save_path_metrics is not None:
                write_pr_csv(metrics, save_path_metrics)
            if save_path_plot is not None:
                plot_pr_curve(metrics, save_path_plot)
        F1 = metrics["F1"]
        print(f"F1 score: {F1[np.argmax(F1)]}")
        return metrics


def write_pr_csv(metrics, path):
    """precision / recall / thresholds / F1 rows, index column first (what DataFrame(metrics).to_csv writes)."""
    import csv
    cols = ("precision", "recall", "thresholds", "F1")
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(("",) + cols)
        for i in range(len(metrics["F1"])):
            w.writerow([i] + ["" if np.isnan(metrics[c][i]) else repr(float(metrics[c][i])) for c in cols])


def plot_pr_curve(metrics, path):
    """Precision over recall, best-F1 point marked (needs matplotlib; skipped with a note when it is absent)."""
    try:
        import matplotlib
        matplotlib.use("Agg")
        from matplotlib import pyplot
    except Exception as e:  # pragma: no cover
        print(f"PR plot skipped ({type(e).__name__}: matplotlib not usable)")
        return
    best = int(np.argmax(metrics["F1"]))
    fig = pyplot.figure(figsize=(6, 6))
    axes = fig.add_subplot(111, xlabel="recall", ylabel="precision", xlim=(-0.05, 1.05), ylim=(-0.05, 1.05))
    axes.step(metrics["recall"], metrics["precision"], where="post", linewidth=1)
    axes.plot([metrics["recall"][best]], [metrics["precision"][best]], marker="o",
              label=f"max F1 = {metrics['F1'][best]:.3f}")
    axes.legend(loc="lower left")
    fig.savefig(path, dpi=120)
    pyplot.close(fig)


class SegPipeUNet(SegPipe):
    """``SegPipe`` with the U-Net of the reference (pipeline.py:379-410)."""

    def __init__(self, checkpoint_dir=None, start_filts=64, depth=5, **kwargs):
        super().__init__(checkpoint_dir, **kwargs)
        # reference: depth 5, 64 filters (pipeline.py:390-410); start_filts=128 is BASELINE configs[4]
        if not self.late_meta_inject:
            self.model = UNet_Baseline(n_classes=3, in_channels=4 + get_in_channels(self.meta_channels),
                                       late_meta_inject=False, depth=depth, start_filts=start_filts,
                                       up_mode="transpose", merge_mode="concat", precision=self.precision,
                                       infer_precision=self.infer_precision)
        else:
            self.model = UNet_LateMetInject(n_classes=3, in_channels=4,
                                            meta_in_channels=get_in_channels(self.meta_channels),
                                            late_meta_inject=True, depth=depth, start_filts=start_filts,
                                            up_mode="transpose", merge_mode="concat", precision=self.precision,
                                            infer_precision=self.infer_precision)


def get_in_channels(meta_channels):
    """Extra input channels contributed by metadata (reference pipeline.py:413-425)."""
    if len(meta_channels) != 0:
        weights = {"portion_year": 1, "portion_day": 2, "depth_rel": 1, "depth_abs_surface": 1,
                   "depth_abs_seabed": 1, "time_diff": 1}
        return int(np.sum([meta_channels[kw] * weights[kw] for kw in weights.keys()]))
    return 0
