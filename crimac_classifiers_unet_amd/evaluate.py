"""Test-set evaluation flow of the reference (crimac_unet/pipeline_train_predict/evaluate.py:39-117) on the
MI355X pipeline.

The reference's ``validate_model_survey_zarr`` / ``validate_model_survey_memm`` build a gridded Dataset over one
survey (zarr) or over every echogram of a survey (memm), wrap it in a DataLoader and hand it to
``SegPipe.validate_model_testing``.  Datasets, readers and label transforms are the reference's host-side numpy
stack (unchanged, fed through the batch-dict API); what runs on the GPU here is
``predict_batch`` (HIP U-Net) and, with ``gpu_metrics``, the PR histograms.  The two functions below keep the
reference's names, keyword arguments and file naming; the Dataset class and the transform factories are taken from
the reference package when it is importable (``batch.dataset``, ``batch.transforms``) or passed in explicitly
(``dataset_cls`` / ``data_transform`` / ``label_transform``), so the module imports without it.
"""
from __future__ import annotations

import os

from torch.utils.data import ConcatDataset, DataLoader


def _reference_factories(dataset_cls, data_transform_factory, label_transform_factory, memm):
    if dataset_cls is None or data_transform_factory is None or label_transform_factory is None:
        try:
            from batch.dataset import DatasetGriddedReader                          # the reference package
            from batch.transforms import (define_data_transform, define_data_transform_test,
                                          define_label_transform_test)
        except Exception as e:  # noqa: BLE001
            raise ImportError("the reference's batch.dataset / batch.transforms are not importable: pass dataset_cls, "
                              "data_transform_factory and label_transform_factory explicitly") from e
        dataset_cls = dataset_cls or DatasetGriddedReader
        data_transform_factory = data_transform_factory or (define_data_transform_test if memm else define_data_transform)
        label_transform_factory = label_transform_factory or define_label_transform_test
    return dataset_cls, data_transform_factory, label_transform_factory


def _is_use_metadata(meta_channels):
    return len(meta_channels) > 0


def validate_model_survey_zarr(readers, segpipe, meta_channels, patch_size, patch_overlap, eval_mode, batch_size,
                               num_workers, save_path_metrics, save_path_plot, preload_n_pings=0, survey="survey",
                               dataset_cls=None, data_transform_factory=None, label_transform_factory=None,
                               worker_init_fn=None, **kwargs):
    """evaluate.py:39-81: one zarr file = one survey; gridded patches of the whole survey -> PR curve / F1."""
    dataset_cls, dtf, ltf = _reference_factories(dataset_cls, data_transform_factory, label_transform_factory, False)
    assert len(readers) == 1, "Current evaluation code assumes one zarr file contains an entire survey"
    assert preload_n_pings == 0, "Current evaluation code for zarr only works when 'preloading_n_pings' = 0"
    frequencies = segpipe.frequencies
    data_transform = dtf(_is_use_metadata(meta_channels))
    label_transform = ltf(frequencies=frequencies, label_masks=eval_mode, patch_overlap=patch_overlap)
    metrics = None
    for reader in readers:
        dataset = dataset_cls(reader, patch_size, frequencies, meta_channels=meta_channels, grid_start=None,
                              grid_end=None, patch_overlap=patch_overlap, data_preload=False,
                              augmentation_function=None, label_transform_function=label_transform,
                              data_transform_function=data_transform, grid_mode="all")
        dataloader = DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=num_workers,
                                worker_init_fn=worker_init_fn)
        metrics = segpipe.validate_model_testing(
            dataloader, save_path_metrics=os.path.join(save_path_metrics, f"{survey}_test.csv"),
            save_path_plot=os.path.join(save_path_plot, f"{survey}_pr.png"))
    return metrics


def validate_model_survey_memm(readers, segpipe, meta_channels, patch_size, patch_overlap, eval_mode, batch_size,
                               num_workers, save_path_metrics, save_path_plot, survey="survey", dataset_cls=None,
                               data_transform_factory=None, label_transform_factory=None, worker_init_fn=None,
                               **kwargs):
    """evaluate.py:84-117: one gridded Dataset per echogram, concatenated -> PR curve / F1 of the survey."""
    dataset_cls, dtf, ltf = _reference_factories(dataset_cls, data_transform_factory, label_transform_factory, True)
    frequencies = segpipe.frequencies
    data_transform = dtf(_is_use_metadata(meta_channels))
    label_transform = ltf(frequencies=frequencies, label_masks=eval_mode, patch_overlap=patch_overlap)
    datasets = [dataset_cls(reader, patch_size, frequencies, meta_channels=meta_channels, grid_start=None,
                            grid_end=None, patch_overlap=patch_overlap, augmentation_function=None,
                            label_transform_function=label_transform, data_transform_function=data_transform,
                            grid_mode="all") for reader in readers]
    dataloader = DataLoader(ConcatDataset(datasets), batch_size=batch_size, shuffle=False, num_workers=num_workers,
                            worker_init_fn=worker_init_fn)
    return segpipe.validate_model_testing(
        dataloader, save_path_metrics=os.path.join(save_path_metrics, f"{survey}_test.csv"),
        save_path_plot=os.path.join(save_path_plot, f"{survey}_pr.png"))
