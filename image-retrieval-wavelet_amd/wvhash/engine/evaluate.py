"""Evaluation entry points with the signatures and return shape of
/root/reference/main/engine/evaluate.py (evaluate :143-169, evaluate_multi_k :172-245, get_tester
:67-91), re-implemented without pytorch_metric_learning and kept on the GPU end to end:

* the embedding sweep (compute_all_embeddings :26-64) writes every batch's codes into a preallocated
  DEVICE tensor (the reference copies each batch to the CPU and ranks there);
* a dataset built with a deferred ``SWTTransform(defer=True)`` yields raw uint8 ``[3, H, W]`` tensors:
  16x fewer bytes cross PCIe than the reference's pre-expanded ``[3, 4, H, W]`` fp32 tensors, and the
  model wrapper runs the batched HIP SWT after the H2D copy;
* metrics come from ``CustomCalculator`` (engine/accuracy_calculator.py) and are reported under the
  reference's key names ``<metric>_level0``.

Label levels.  PML's tester slices ``labels[:, level]`` before calling the calculator; whether the
reference's multi-hot label matrices reach ``calculate_maphashing`` whole or as column 0 cannot be
checked without PML.  ``label_hierarchy_level=None`` (default) passes the whole label
matrix -- the semantics of studies/measure_random_baseline.py:107 and of the published numbers'
"shares >= 1 tag" relevance; an int reproduces the column slice.
"""
import logging
from collections import defaultdict

import torch
from torch.utils.data import DataLoader

from .. import _lib
from .accuracy_calculator import get_accuracy_calculator

LOGGER = logging.getLogger("RETRIEVAL")


def get_data(batch):
    return batch["image"].cuda(non_blocking=True), batch["label"]


class GlobalEmbeddingSpaceTester:
    def __init__(self, normalize_embeddings=False, batch_size=64, dataloader_num_workers=16,
                 accuracy_calculator=None, label_hierarchy_level=None, pca=None, data_and_label_getter=get_data):
        if pca is not None:
            raise NotImplementedError("pca is not supported")
        self.normalize_embeddings = normalize_embeddings
        self.batch_size = batch_size
        self.dataloader_num_workers = dataloader_num_workers
        self.accuracy_calculator = accuracy_calculator
        self.label_hierarchy_level = label_hierarchy_level
        self.data_and_label_getter = data_and_label_getter
        self.reference_split_names = {}

    # -- embedding sweep -----------------------------------------------------------------------
    def compute_all_embeddings(self, dataloader, trunk_model, embedder_model=None):
        if len(dataloader.dataset) == 0:
            raise ValueError("compute_all_embeddings got an empty dataset")
        _lib.require_gpu()
        s = 0
        all_q = labels = None
        expand = self._deferred_expander(dataloader.dataset, trunk_model)
        with torch.no_grad():
            LOGGER.info("Computing embeddings")
            for i, data in enumerate(dataloader):
                img, label = self.data_and_label_getter(data)
                if expand is not None and img.dim() == 4:
                    img = expand(img)
                q = trunk_model(img)
                if embedder_model is not None:
                    q = embedder_model(q)
                if self.normalize_embeddings:
                    q = torch.nn.functional.normalize(q, p=2, dim=1)
                label = torch.as_tensor(label).to(q.device, non_blocking=True)
                if label.dim() == 1:
                    label = label.unsqueeze(1)
                if i == 0:
                    n = len(dataloader.dataset)
                    labels = torch.zeros(n, label.size(1), device=q.device, dtype=label.dtype)
                    all_q = torch.zeros(n, q.size(1), device=q.device, dtype=q.dtype)
                e = s + q.size(0)
                all_q[s:e] = q
                labels[s:e] = label
                s = e
        return all_q, labels

    @staticmethod
    def _deferred_expander(dataset, trunk_model):
        """A dataset whose wavelet plugin was built with defer=True yields raw uint8 [3,H,W] images.  The plugin
        object itself says how to expand them (class, wavelet, level, copies): bind it to the model when the model
        can expand on its own (band-major, no second copy), else expand here after the H2D copy."""
        from ..transforms.custom_transforms import find_wavelet_transform
        tf = find_wavelet_transform(dataset)
        if tf is None or not tf.defer:
            return None
        inner = getattr(trunk_model, "module", trunk_model)          # nn.DataParallel (evaluate.py:70-71)
        if hasattr(inner, "bind_transform"):
            inner.bind_transform(tf)
            LOGGER.info(f"deferred transform bound to the model: {tf}")
            return None
        LOGGER.info(f"deferred transform applied by the engine after the H2D copy: {tf}")
        return tf.apply_batch

    def get_all_embeddings(self, dataset, trunk_model, embedder_model=None):
        dl = DataLoader(dataset, batch_size=self.batch_size, num_workers=self.dataloader_num_workers,
                        shuffle=False, drop_last=False, pin_memory=True)
        return self.compute_all_embeddings(dl, trunk_model, embedder_model)

    def get_splits_to_compute_embeddings(self, dataset_dict, splits_to_eval):
        splits_to_eval = splits_to_eval or [(k, [k]) for k in dataset_dict]
        needed = []
        for q, refs in splits_to_eval:
            for name in [q] + list(refs):
                if name not in needed:
                    needed.append(name)
        return splits_to_eval, needed

    def get_all_embeddings_for_all_splits(self, dataset_dict, trunk_model, embedder_model, splits, collate_fn=None):
        trunk_model.eval()
        return {name: self.get_all_embeddings(dataset_dict[name], trunk_model, embedder_model) for name in splits}

    # -- ranking ---------------------------------------------------------------------------------
    def _levels(self, labels):
        lvl = self.label_hierarchy_level
        if lvl is None:
            return [None]
        if lvl == "all":
            return list(range(labels.shape[1]))
        return [lvl] if isinstance(lvl, int) else list(lvl)

    def do_knn_and_accuracies(self, accuracies, embeddings_and_labels, query_split_name, reference_split_names):
        q_emb, q_lab = embeddings_and_labels[query_split_name]
        r_emb = torch.cat([embeddings_and_labels[n][0] for n in reference_split_names], dim=0)
        r_lab = torch.cat([embeddings_and_labels[n][1] for n in reference_split_names], dim=0)
        same_source = query_split_name in reference_split_names
        LOGGER.info("label mode: " + ("whole label matrix (relevance = shares >= 1 tag)" if self.label_hierarchy_level
                                      is None else f"label column(s) {self.label_hierarchy_level} (PML level slicing)"))
        for level in self._levels(q_lab):
            ql = q_lab if level is None else q_lab[:, level]
            rl = r_lab if level is None else r_lab[:, level]
            acc = self.accuracy_calculator.get_accuracy(q_emb, ql, r_emb, rl, same_source)
            for metric, v in acc.items():
                accuracies[f"{metric}_level{0 if level is None else level}"] = v

    def test(self, dataset_dict, epoch, trunk_model, embedder_model=None, splits_to_eval=None, collate_fn=None):
        splits_to_eval, needed = self.get_splits_to_compute_embeddings(dataset_dict, splits_to_eval)
        emb = self.get_all_embeddings_for_all_splits(dataset_dict, trunk_model, embedder_model, needed, collate_fn)
        all_accuracies = defaultdict(dict)
        for query_split_name, reference_split_names in splits_to_eval:
            all_accuracies[query_split_name]["epoch"] = epoch
            self.reference_split_names[query_split_name] = reference_split_names
            self.do_knn_and_accuracies(all_accuracies[query_split_name], emb, query_split_name, reference_split_names)
        return dict(all_accuracies)


def get_tester(normalize_embeddings=False, batch_size=64, num_workers=16, pca=None, exclude_ranks=None, k=5000,
               label_hierarchy_level=None, **kwargs):
    calculator = get_accuracy_calculator(exclude_ranks=exclude_ranks, k=k, **kwargs)
    return GlobalEmbeddingSpaceTester(normalize_embeddings=normalize_embeddings, batch_size=batch_size,
                                      dataloader_num_workers=num_workers, accuracy_calculator=calculator,
                                      label_hierarchy_level=label_hierarchy_level, pca=pca)


def _build_dataset_dict_and_splits(train_dataset, val_dataset, test_dataset, custom_eval):
    """Same branching as the reference (:95-140), minus the my_at_R bookkeeping it no longer uses."""
    dataset_dict, splits_to_eval = {}, []
    if train_dataset is not None:
        dataset_dict["train"] = train_dataset
        splits_to_eval.append(('train', ['train']))
    if val_dataset is not None:
        dataset_dict["val"] = val_dataset
        splits_to_eval.append(('val', ['val']))
    if test_dataset is not None:
        if isinstance(test_dataset, dict):
            if 'gallery' in test_dataset:
                dataset_dict.update(test_dataset)
                splits_to_eval.append(('test', ['gallery']))
            elif 'distractor' in test_dataset:
                dataset_dict.update(test_dataset)
                splits_to_eval.append(('test', ['test', 'distractor']))
        elif isinstance(test_dataset, list):
            for dts in test_dataset:
                dataset_dict.update(dts)
                names = list(dts.keys())
                splits_to_eval.append((names[0] if names[0].startswith("query") else names[1],
                                       [names[0] if names[0].startswith("gallery") else names[1]]))
        else:
            dataset_dict["test"] = test_dataset
            splits_to_eval.append(('test', ['test']))
    if custom_eval is not None:
        dataset_dict = custom_eval["dataset"]
        splits_to_eval = custom_eval["splits"]
    return dataset_dict, splits_to_eval


def _preserve_rng(fn):
    """@lib.get_set_random_state of the reference: evaluation must not disturb the training RNG streams."""
    def wrapped(*args, **kwargs):
        cpu_state = torch.get_rng_state()
        cuda_state = torch.cuda.get_rng_state_all() if torch.cuda.is_available() else None
        try:
            return fn(*args, **kwargs)
        finally:
            torch.set_rng_state(cpu_state)
            if cuda_state is not None:
                torch.cuda.set_rng_state_all(cuda_state)
    wrapped.__name__ = fn.__name__
    wrapped.__doc__ = fn.__doc__
    return wrapped


@_preserve_rng
def evaluate(net, train_dataset=None, val_dataset=None, test_dataset=None, epoch=None, tester=None,
             custom_eval=None, **kwargs):
    dataset_dict, splits_to_eval = _build_dataset_dict_and_splits(train_dataset, val_dataset, test_dataset, custom_eval)
    if tester is None:
        tester = get_tester(**kwargs)
    return tester.test(dataset_dict=dataset_dict, epoch=f"{epoch}", trunk_model=net, splits_to_eval=splits_to_eval)


@_preserve_rng
def evaluate_multi_k(net, train_dataset=None, val_dataset=None, test_dataset=None, epoch=None, custom_eval=None,
                     k_list=(5000,), **kwargs):
    """Embeds once and ranks once: the calculators of all k share a RankCache -- packed codes, packed labels and the
    ranked lists at the largest k; every smaller k is a prefix.  Returns {k: {split: {metric: value}}} like the
    reference, which re-runs the whole ranking per k (main/engine/evaluate.py:226-243)."""
    from .accuracy_calculator import RankCache
    dataset_dict, splits_to_eval = _build_dataset_dict_and_splits(train_dataset, val_dataset, test_dataset, custom_eval)
    k_list = list(k_list)
    tester = get_tester(k=k_list[0], **kwargs)
    net.eval()
    splits_to_eval, needed = tester.get_splits_to_compute_embeddings(dataset_dict, splits_to_eval)
    LOGGER.info(f"Computing embeddings once, reused for k in {k_list}")
    emb = tester.get_all_embeddings_for_all_splits(dataset_dict, net, None, needed)
    ints = [k for k in k_list if isinstance(k, int)]
    cache = RankCache(kmax_hint=max(ints) if ints else 0)
    results_by_k = {}
    for k in k_list:
        tester.accuracy_calculator = get_tester(k=k, rank_cache=cache, **kwargs).accuracy_calculator
        all_accuracies = defaultdict(dict)
        for query_split_name, reference_split_names in splits_to_eval:
            all_accuracies[query_split_name]["epoch"] = f"{epoch}"
            tester.do_knn_and_accuracies(all_accuracies[query_split_name], emb, query_split_name, reference_split_names)
        results_by_k[k] = dict(all_accuracies)
    return results_by_k
