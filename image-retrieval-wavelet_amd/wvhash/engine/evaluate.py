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


# ---------------------------------------------------------------------------------------------------------------
# Sharded evaluation: one process per GPU (the c3 / c4 shape: COCO's 117k-image database over 8 GPUs)
def _even_slice(n, world, rank):
    per = (n + world - 1) // world
    return min(n, rank * per), min(n, (rank + 1) * per), per


@_preserve_rng
def evaluate_sharded(net, test_dataset, k=5000, epoch=None, batch_size=64, num_workers=16, group=None, **kwargs):
    """Hashing retrieval metrics of ``{"test": queries, "gallery": database}`` with the work split over the ranks of
    ``group`` (torch.distributed, backend nccl = RCCL; every rank calls this with the same arguments):

    * rank r embeds database rows [lo_r, hi_r) and its slice of the queries -- the codes stay on its GPU, packed;
    * the search is wvhash.parallel.sharded_hamming_topk: all_gather of the query codes, per-shard histograms and list
      prefixes, all_to_all, GPU merge -- the role faiss.index_cpu_to_all_gpus(shards=True) plays in the reference
      (main/engine/get_knn.py:41-44), whose embedding sweep runs under nn.DataParallel (evaluate.py:70-71);
    * mAP comes from relevance strings (wvhash.parallel.sharded_hamming_map_at_k: a shard ranks against its own rows' labels,
      1 bit per list entry on the wire) whenever the shape allows -- up to 128 classes, 128 bits, any k (mAP@ALL included), shards of at
      most 32,768 rows: MIRFLICKR and COCO at k = 5000; otherwise lists are exchanged, the packed database labels
      all-gathered (8 bytes per row and label word) and AP computed from the merged lists: the same numbers;
    * the AP sums and the per-bit counts are all-reduced.
    Returns ``{"test": {"epoch", "maphashing_level0", "map_level0", "bit_balance_level0", "worst_bit_balance_level0"}}`` -- the
    four columns evaluate_all_checkpoints.py:173-174 reads into its CSVs --, the same
    numbers as evaluate() on one GPU (lists are identical for every world size; AP sums differ by fp64 rounding only).
    Every rank's query slice is padded to a common length by repeating its last query; the copies are not counted."""
    import torch.distributed as dist
    from torch.utils.data import Subset
    from ..parallel import sharded_hamming_topk, sharded_hamming_map_at_k, _all_gather, _all_reduce
    from . import hamming as H
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if not (isinstance(test_dataset, dict) and "gallery" in test_dataset and "test" in test_dataset):
        raise ValueError("evaluate_sharded expects {'test': query dataset, 'gallery': database dataset}")
    tester = get_tester(batch_size=batch_size, num_workers=num_workers, k=k, **kwargs)
    net.eval()
    n_db, n_q = len(test_dataset["gallery"]), len(test_dataset["test"])
    lo, hi, _ = _even_slice(n_db, world, rank)
    qlo, qhi, q_per = _even_slice(n_q, world, rank)

    def embed(ds, a, b):
        if b <= a:
            return None, None
        return tester.get_all_embeddings(Subset(ds, range(a, b)) if (a, b) != (0, len(ds)) else ds, net)

    r_codes, r_lab = embed(test_dataset["gallery"], lo, hi)
    q_codes, q_lab = embed(test_dataset["test"], qlo, qhi)
    dev = torch.device("cuda", torch.cuda.current_device())
    nbits = (r_codes if r_codes is not None else q_codes).shape[1]
    words = (nbits + 63) // 64
    # ---- local queries, padded to q_per rows
    n_local_q = max(0, qhi - qlo)
    if n_local_q:
        qp, qlp = H.pack_codes(q_codes), H.pack_labels(q_lab.float())
    else:
        qp = torch.zeros((0, words), dtype=torch.int64, device=dev)
        qlp = torch.zeros((0, 1), dtype=torch.int64, device=dev)
    lw = torch.tensor([qlp.shape[1] if n_local_q else 0], dtype=torch.int32, device=dev)
    _all_reduce(lw, dist.ReduceOp.MAX, group) if world > 1 else None
    lw = int(lw.item())
    if not n_local_q:
        qlp = torch.zeros((0, lw), dtype=torch.int64, device=dev)
    pad = q_per - n_local_q
    if pad:
        filler_q = qp[-1:].expand(pad, -1) if n_local_q else torch.zeros((pad, words), dtype=torch.int64, device=dev)
        filler_l = qlp[-1:].expand(pad, -1) if n_local_q else torch.zeros((pad, lw), dtype=torch.int64, device=dev)
        qp, qlp = torch.cat([qp, filler_q]).contiguous(), torch.cat([qlp, filler_l]).contiguous()
    # ---- database shard + labels of the whole database
    rp = H.pack_codes(r_codes) if r_codes is not None else torch.zeros((0, words), dtype=torch.int64, device=dev)
    per_db = _even_slice(n_db, world, 0)[2]
    rlp_local = torch.zeros((per_db, lw), dtype=torch.int64, device=dev)
    if r_codes is not None:
        rlp_local[:hi - lo] = H.pack_labels(r_lab.float())
    k_eff = min(int(k), n_db) if k is not None else n_db
    ap = None
    if world > 1 and lw in (1, 2) and nbits <= 128 and min(k_eff, per_db) <= H.RANK_K_MAX and per_db <= H.SHARD_ROWS_MAX:
        # mAP from relevance strings: no list and no database label leaves its GPU (decided from values every rank shares)
        shard_labels = H.PreparedLabels(rlp_local[:hi - lo].contiguous()) if hi > lo else None
        got = sharded_hamming_map_at_k(qp, qlp, H.PreparedDB(rp, nbits), shard_labels, nbits, k_eff, n_db, None, group=group)
        ap = got[0] if got is not None else None
    if ap is None:
        if world > 1:
            rlp_all = torch.empty((world * per_db, lw), dtype=torch.int64, device=dev)
            _all_gather(rlp_all, rlp_local, group)
        else:
            rlp_all = rlp_local
        idx, _ = sharded_hamming_topk(qp, rp, nbits, k_eff, n_db, group=group, want_dist=False)
        # global row g of shard s sits at row s * per_db + (g - lo_s) of the gathered label table = g (shards are contiguous
        # slices of equal length per_db, the last one shorter): the gathered table is indexed by the global row directly
        ap, _ = H.map_at_k(idx, qlp, rlp_all)
    # map_level0 (calculate_map, accuracy_calculator.py:156-167): the same average precisions, averaged over the queries
    # that are not "lone" -- at least one database row anywhere shares a label with them (a lone query's AP is 0, so the
    # numerator is the one of maphashing).  Every rank tests all queries' label words against its shard's; MAX-reduced.
    if world > 1:
        ql_all = torch.empty((world * q_per, lw), dtype=torch.int64, device=dev)
        _all_gather(ql_all, qlp.contiguous(), group)
    else:
        ql_all = qlp
    has = torch.zeros(ql_all.shape[0], dtype=torch.int32, device=dev)
    shard_lab = rlp_local[:hi - lo]
    if hi > lo:
        for s in range(0, ql_all.shape[0], 256):
            has[s:s + 256] = ((ql_all[s:s + 256, None, :] & shard_lab[None, :, :]) != 0).any(-1).any(1).int()
    if world > 1:
        _all_reduce(has, dist.ReduceOp.MAX, group)
    not_lone = has[rank * q_per:rank * q_per + n_local_q] if world > 1 else has[:n_local_q]
    sums = torch.zeros(3 + nbits, dtype=torch.float64, device=dev)
    sums[0] = ap[:n_local_q].double().sum()
    sums[1] = float(hi - lo)
    sums[2] = not_lone.double().sum()
    if hi > lo:
        sums[3:] = H.bit_counts(rp, nbits).double()
    if world > 1:
        _all_reduce(sums, dist.ReduceOp.SUM, group)
    frac = sums[3:] / sums[1]
    balance = 1.0 - 2.0 * (frac - 0.5).abs()
    kept = float(sums[2].item())
    return {"test": {"epoch": f"{epoch}", "maphashing_level0": float(sums[0].item()) / n_q,
                     "map_level0": float(sums[0].item()) / kept if kept else 0.0,
                     "bit_balance_level0": float(balance.mean().item()),
                     "worst_bit_balance_level0": float(balance.min().item())}}
