"""Test-set evaluation flow of the reference (crimac_unet/pipeline_train_predict/evaluate.py:39-117) on the
MI355X pipeline.

The reference's ``validate_model_survey_zarr`` / ``validate_model_survey_memm`` build a gridded Dataset over one
survey (zarr) or over every echogram of a survey (memm), wrap it in a DataLoader and hand it to
``SegPipe.validate_model_testing``.  Datasets, readers and label transforms are the reference's host-side numpy
stack (unchanged, fed through the batch-dict API); what runs on the GPU here is
``predict_batch`` (HIP U-Net) and, with ``gpu_metrics``, the PR histograms.  The two functions below keep the
reference's names, keyword arguments and file naming.  This package never reaches into the reference on its own: the
Dataset class and the transform factories are INJECTED (``dataset_cls`` / ``data_transform_factory`` /
``label_transform_factory``), and the command line (``python -m crimac_classifiers_unet_amd.evaluate``, the reference's
``evaluate.py`` ``__main__``, :120-167) takes the location of the host data stack as an explicit ``--data-stack`` path.
"""
from __future__ import annotations

import os

from torch.utils.data import ConcatDataset, DataLoader


def _reference_factories(dataset_cls, data_transform_factory, label_transform_factory, memm):
    """The host-side Dataset and transform factories must be handed in (no implicit import of another package)."""
    missing = [n for n, v in (("dataset_cls", dataset_cls), ("data_transform_factory", data_transform_factory),
                              ("label_transform_factory", label_transform_factory)) if v is None]
    if missing:
        raise ImportError(
            f"validate_model_survey_{'memm' if memm else 'zarr'}: pass {', '.join(missing)} explicitly -- the gridded "
            "Dataset class (the reference's batch.dataset.DatasetGriddedReader), the data-transform factory "
            f"(batch.transforms.{'define_data_transform_test' if memm else 'define_data_transform'}) and the "
            "label-transform factory (batch.transforms.define_label_transform_test); `data_stack_factories(path)` "
            "builds the three from an explicitly named checkout of the host data stack")
    return dataset_cls, data_transform_factory, label_transform_factory


def data_stack_factories(path, memm):
    """The three host-side factories from the data stack at ``path`` (a checkout of the reference's ``crimac_unet``
    directory, NAMED BY THE CALLER): ``(DatasetGriddedReader, data-transform factory, label-transform factory)``."""
    import importlib
    import sys
    path = os.path.abspath(str(path))
    if not os.path.isdir(os.path.join(path, "batch")):
        raise ImportError(f"{path} does not look like the host data stack (no batch/ package)")
    if path not in sys.path:
        sys.path.insert(0, path)
    ds = importlib.import_module("batch.dataset")
    tr = importlib.import_module("batch.transforms")
    return (ds.DatasetGriddedReader, tr.define_data_transform_test if memm else tr.define_data_transform,
            tr.define_label_transform_test)


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


def main(argv=None, data_partition_factory=None, factories=None):
    """``evaluate.py`` ``__main__`` (evaluate.py:120-167): yaml + command line -> SegPipeUNet with loaded parameters ->
    for every evaluation survey of the data partition: PR curve / F1 csv + plot.

    The host data stack (readers, partition, Dataset, transforms: SURVEY.md §2 rows 5-9, out of scope here) is named
    explicitly: ``--data-stack PATH`` (a checkout of the reference's ``crimac_unet`` directory), or injected by the
    caller -- ``data_partition_factory(**config) -> object with get_evaluation_surveys() / get_survey_readers(survey)``
    and ``factories = (dataset_cls, data_transform_factory, label_transform_factory)``."""
    import argparse
    import time
    from pathlib import Path

    import yaml

    from .pipeline import SegPipeUNet
    ap = argparse.ArgumentParser(description="Test-set evaluation (PR curve, F1) of a trained U-Net on MI355X")
    ap.add_argument("--yaml_path", type=lambda p: Path(p).resolve(strict=True), required=True)
    ap.add_argument("--checkpoint_path", type=lambda p: Path(p).resolve(strict=True), required=True)
    ap.add_argument("--save_path_metrics", type=lambda p: Path(p).resolve(strict=True), required=True)
    ap.add_argument("--save_path_plot", type=lambda p: Path(p).resolve(strict=True), required=True)
    ap.add_argument("--save_model_params", action="store_true", default=False)
    ap.add_argument("--batch_size", type=int)
    ap.add_argument("--data_mode", choices=["memm", "zarr"])
    ap.add_argument("--data-stack", dest="data_stack", default=None,
                    help="checkout of the host data stack (the reference's crimac_unet directory): readers, partition, "
                         "Dataset, transforms")
    args = ap.parse_args(argv)
    config = yaml.safe_load(open(args.yaml_path))
    for k, v in vars(args).items():                     # command line takes precedence (utils/general.py:128-136)
        if v is not None or k not in config:
            config[k] = v
    data_stack = config.pop("data_stack", None)
    experiment_name = Path(config["yaml_path"]).stem
    memm = config["data_mode"] == "memm"
    if config["data_mode"] not in ("zarr", "memm"):
        raise ValueError('data_mode not in ["zarr", "memm"]')
    if factories is None:
        if data_stack is None:
            raise SystemExit("evaluate: name the host data stack with --data-stack PATH (or call main() with injected "
                             "factories): this package does not import it on its own")
        factories = data_stack_factories(data_stack, memm)
    if data_partition_factory is None:
        if data_stack is None:
            raise SystemExit("evaluate: --data-stack PATH is needed for the data partition (data.partition.DataZarr / DataMemm)")
        import importlib
        data_stack_factories(data_stack, memm)          # (puts the named path on sys.path)
        part = importlib.import_module("data.partition")
        data_partition_factory = part.DataMemm if memm else part.DataZarr
    seed = config.get("random_seed", 0)
    import random

    import numpy as np
    import torch
    np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)
    segpipe = SegPipeUNet(**config, experiment_name=experiment_name)
    segpipe.load_model_params(checkpoint_path=config["checkpoint_path"])
    print(f'\nLoading {config["data_mode"]} data partition object...')
    t0 = time.time()
    partition = data_partition_factory(**config)
    print(f"Executed time for loading data partition object (min): {round((time.time() - t0) / 60, 2)}")
    run = os.path.normpath(str(config["checkpoint_path"])).split(os.path.sep)[-2]
    config["save_path_metrics"] = os.path.join(str(config["save_path_metrics"]), experiment_name, run)
    config["save_path_plot"] = os.path.join(str(config["save_path_plot"]), experiment_name, run)
    os.makedirs(config["save_path_metrics"], exist_ok=True)
    os.makedirs(config["save_path_plot"], exist_ok=True)
    print("\nMetrics directory:", config["save_path_metrics"])
    print("Plot directory:", config["save_path_plot"], "\n")
    ds, dtf, ltf = factories
    results = {}
    for survey in partition.get_evaluation_surveys():
        readers = partition.get_survey_readers(survey)
        print("Running evaluation for", survey)
        fn = validate_model_survey_memm if memm else validate_model_survey_zarr
        kw = dict(config)
        kw.update(survey=survey, dataset_cls=ds, data_transform_factory=dtf, label_transform_factory=ltf)
        results[survey] = fn(readers, segpipe, **kw)
    return results


if __name__ == "__main__":
    main()
