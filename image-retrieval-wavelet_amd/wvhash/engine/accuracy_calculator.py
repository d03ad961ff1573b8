"""GPU accuracy calculator with the method surface of
/root/reference/main/engine/accuracy_calculator.py (CustomCalculator :16-349,
get_accuracy_calculator :352-403).

The reference subclasses pytorch_metric_learning's AccuracyCalculator and runs on CPU tensors; this
class is self-contained (PML is not a dependency), keeps everything on the GPU -- or, with an explicit ``device='cpu'`` (the
reference's own configuration, main/engine/evaluate.py:76-81; BASELINE config c0), on the host through the `_cpu` twins
of the same entry points (csrc/host_rank.cpp; same integers, same AP bits) -- and routes the arithmetic through libwvhash:
  calc_hamming_dist       -> wv_hamming_dist        (:183-186)
  calculate_maphashing    -> wv_hamming_topk + wv_map_at_k   (:203-231, the reported metric)
  calculate_bit_balance / calculate_worst_bit_balance -> wv_bit_counts   (:188-200)
  calculate_map           -> get_knn + wv_map_at_k  (:156-167, torchmetrics RetrievalMAP)
  calculate_rpr / calculate_pr / calculate_pr_rc / calculate_pr_rc_hashing -> wv_hit_prefix (:131-181, :235-273)
Ranking ties are broken by ascending reference index (see engine/get_knn.py).
"""
import logging

import torch

from .. import _lib
from . import hamming as H
from . import hamming_host as HH
from .get_knn import get_knn, _to_gpu, _is_pm1

LOGGER = logging.getLogger("RETRIEVAL")

_RECALL_KS = (1, 2, 4, 8, 10, 16, 20, 30, 32, 100, 1000)


class RankCache(object):
    """Packed codes, packed labels and ranked lists of ONE (query, reference) embedding pair, shared by the calculators
    of an evaluate_multi_k run: the database is packed once and ranked once at the largest k asked for; every smaller k
    reads a prefix (the reference re-runs its whole per-query loop for every k, main/engine/evaluate.py:226-243).
    Entries are keyed by tensor identity (storage pointer, shape, version counter): new embeddings invalidate them."""

    def __init__(self, kmax_hint=0):
        self.kmax_hint = int(kmax_hint)
        self._codes, self._labels, self._lists = {}, {}, None

    @staticmethod
    def _key(t):
        return (t.data_ptr(), tuple(t.shape), t.dtype, t._version)

    def packed_codes(self, x):
        key = self._key(x)
        if key not in self._codes:
            self._codes = {k_: v for k_, v in self._codes.items() if len(self._codes) < 4}
            self._codes[key] = H.pack_codes(x)
        return self._codes[key]

    def packed_labels(self, query_labels, reference_labels):
        key = (self._key(query_labels), self._key(reference_labels))
        if key not in self._labels:
            self._labels = {key: CustomCalculator._packed_labels(query_labels, reference_labels)}
        return self._labels[key]

    def lists(self, query, reference, k):
        """int32 [Q, >= k] ranked lists (ascending distance, then index); only the first k columns are meaningful to the caller."""
        key = (self._key(query), self._key(reference))
        if self._lists is None or self._lists[0] != key or self._lists[1].shape[1] < k:
            kk = min(reference.shape[0], max(k, self.kmax_hint))
            idx, dist = H.hamming_topk(self.packed_codes(query), self.packed_codes(reference), reference.shape[1], kk)
            self._lists = (key, idx, dist)
        return self._lists[1]

    def knn(self, reference, query, num_k, same_source):
        """get_knn(..., distance_metric='hamming') from the cached ranking: (indices int64, inner products fp32)."""
        num_k += int(same_source)
        if num_k > reference.shape[0]:
            raise RuntimeError(f"selected index k out of range (k={num_k}, references={reference.shape[0]})")
        idx = self.lists(query, reference, num_k)[:, :num_k]
        ip = float(reference.shape[1]) - 2.0 * self._lists[2][:, :num_k].float()
        first = int(same_source)
        return idx[:, first:].long(), ip[:, first:]


class CustomCalculator(object):

    def __init__(self, include=(), exclude=(), avg_of_avgs=False, return_per_class=False, k=None,
                 label_comparison_fn=None, device=None, knn_func=None, kmeans_func=None,
                 with_faiss=True, distance_metric="l2", **kwargs):
        if label_comparison_fn is not None or knn_func is not None or kmeans_func is not None:
            raise NotImplementedError("custom label_comparison_fn / knn_func / kmeans_func are not supported")
        if avg_of_avgs or return_per_class:
            raise NotImplementedError("avg_of_avgs / return_per_class are not supported")
        if not (isinstance(k, int) and k > 0) and k not in (None, "max_bin_count"):
            raise ValueError("k must be a positive int, None or 'max_bin_count'")
        self.k = k
        self.num_top_k = k
        self.with_faiss = with_faiss
        self.pr_rc_path, self.last_pr_rc = kwargs.pop("pr_rc_path", "pr_rc.csv"), None
        self.distance_metric = distance_metric
        self.rank_cache = kwargs.pop("rank_cache", None)          # shared by the calculators of evaluate_multi_k
        # device=None / 'cuda': the ranking stage lives on the GPU.  device='cpu' (what the reference pins its calculator
        # to, main/engine/evaluate.py:76-81): the host twins of the same entry points -- explicit, never a fallback: without
        # a GPU and without device='cpu' every metric raises WvhashUnavailable.
        self.requested_device = device
        self.host = device is not None and torch.device(device).type == "cpu"
        self.H = HH if self.host else H
        if self.host and self.rank_cache is not None:
            raise ValueError("a shared RankCache holds GPU lists: not available with device='cpu'")
        self.original_function_dict = {name[len("calculate_"):]: getattr(self, name)
                                       for name in dir(self) if name.startswith("calculate_")}
        self.check_primary_metrics(include, exclude)
        self.original_function_dict = self.get_function_dict(include, exclude)
        self.curr_function_dict = self.get_function_dict()
        LOGGER.info(f"Initializing CustomCalculator with with_faiss={with_faiss} and "
                    f"distance_metric={distance_metric} device: {'cpu (host twins)' if self.host else 'cuda (HIP)'}")

    # ------------------------------------------------------------------ bookkeeping (PML surface)
    @property
    def device(self):
        if self.host:
            return torch.device("cpu")
        _lib.require_gpu()
        return torch.device("cuda", torch.cuda.current_device())

    def _dev(self, x):
        """Tensor on the calculator's device (accuracy_calculator.py:290-293 moves everything to self.device)."""
        if not self.host:
            return _to_gpu(x)
        return (x if torch.is_tensor(x) else torch.as_tensor(x)).detach().cpu()

    def check_primary_metrics(self, include=(), exclude=()):
        # unlike PML, names this implementation does not compute are tolerated in `exclude`
        # (the reference's exclude lists name PML metrics such as NMI / AMI / mean_reciprocal_rank)
        for m in include:
            if m not in self.original_function_dict:
                raise ValueError(f"{m} is not a metric computed by wvhash; valid: {sorted(self.original_function_dict)}")

    def get_function_dict(self, include=(), exclude=()):
        if len(include) == 0:
            include = list(self.original_function_dict.keys())
        included = [k for k in include if k not in exclude]
        return {k: v for k, v in self.original_function_dict.items() if k in included}

    def get_curr_metrics(self):
        return [k for k in self.curr_function_dict.keys()]

    def requires_knn(self):
        return ["precision_at_1", "recall_classic", "rpr", "pr", "pr_rc", "map"] + \
               [f"recall_at_{k}" for k in _RECALL_KS]

    def requires_clustering(self):
        return []

    def description(self):
        return "avg_of_avgs" if False else ""

    def determine_k(self, bin_counts, num_reference_embeddings, embeddings_come_from_same_source):
        self_count = int(embeddings_come_from_same_source)
        if self.k == "max_bin_count":
            return int(torch.max(bin_counts).item()) - self_count
        if self.k is None:
            return num_reference_embeddings - self_count
        return self.k

    # ------------------------------------------------------------------ relevance (:31-37)
    def label_comparison_fn(self, query_labels, reference_labels):
        if query_labels.ndim > 1 and reference_labels.ndim > 1:
            if query_labels.dim() == 2 and reference_labels.dim() == 2:
                return torch.matmul(query_labels.float(), reference_labels.t().float()) > 0
            return (query_labels.float() * reference_labels.float()).sum(dim=-1) > 0
        return query_labels.unsqueeze(1) == reference_labels

    def _match_counts(self, query_labels, reference_labels, chunk=256):
        """#references relevant to each query (what PML's get_label_match_counts feeds determine_k /
        the lone-query mask with), chunked so the [Q, N] relevance matrix is never whole in memory."""
        out = torch.empty(query_labels.shape[0], dtype=torch.long, device=query_labels.device)
        for s in range(0, query_labels.shape[0], chunk):
            out[s:s + chunk] = self.label_comparison_fn(query_labels[s:s + chunk], reference_labels).sum(dim=1)
        return out

    # ------------------------------------------------------------------ hashing primitives
    def calc_hamming_dist(self, qB, rB):
        """0.5 * (B - qB @ rB.T) for +-1 codes (:183-186) -> fp32 [Q, N] like the reference."""
        qB, rB = self._dev(qB), self._dev(rB)
        return self.H.hamming_dist(self.H.pack_codes(qB), self.H.pack_codes(rB), nbits=qB.shape[1]).float()

    def per_bit_balance(self, reference):
        reference = self._dev(reference)
        nbits = reference.shape[1]
        counts = self.H.bit_counts(self.H.pack_codes(reference, check=False), nbits)
        frac_positive = counts.float() / float(reference.shape[0])
        return 1.0 - 2.0 * (frac_positive - 0.5).abs()

    def calculate_bit_balance(self, reference, **kwargs):
        return self.per_bit_balance(reference).mean().item()

    def calculate_worst_bit_balance(self, reference, **kwargs):
        return self.per_bit_balance(reference).min().item()

    def _ranked_lists(self, query, reference, topk):
        """int32 [Q, >= topk]: the first topk columns are the ranked list (a shared RankCache may hold longer lists)."""
        if self.rank_cache is not None:
            return self.rank_cache.lists(query, reference, topk)
        nbits = reference.shape[1]
        rp = self.H.pack_codes(reference)
        if rp.shape[0] > self.H.SHARD_ROWS_MAX:          # large database: virtual shards through the windowed kernel
            rp = self.H.PreparedDB(rp, nbits)
        return self.H.hamming_topk(self.H.pack_codes(query), rp, nbits, topk, want_dist=False)[0]

    @staticmethod
    def _packed_labels(query_labels, reference_labels, Hm=H):
        if query_labels.ndim == 1:  # class-id labels: one-hot them onto bits
            classes = torch.unique(torch.cat([query_labels, reference_labels]))
            query_labels = (query_labels.unsqueeze(1) == classes).float()
            reference_labels = (reference_labels.unsqueeze(1) == classes).float()
        return Hm.pack_labels(query_labels), Hm.pack_labels(reference_labels)

    def _average_precisions(self, idx, query_labels, reference_labels, k=None):
        packed = (self.rank_cache.packed_labels(query_labels, reference_labels) if self.rank_cache is not None
                  else self._packed_labels(query_labels, reference_labels, self.H))
        return self.H.map_at_k(idx, *packed, k=k)

    def _hits(self, idx, query_labels, reference_labels):
        """Running hit counts along the ranked lists, int32 [Q, k] (wv_hit_prefix)."""
        return self.H.hit_prefix(idx.int(), *self._packed_labels(query_labels, reference_labels, self.H))

    def calculate_maphashing(self, query, query_labels, reference, reference_labels, topk,
                             ref_includes_query=False, return_per_query=False, **kwargs):
        while isinstance(topk, (tuple, list)):
            topk = topk[0] if len(topk) else None
        query, reference = self._dev(query), self._dev(reference)
        query_labels, reference_labels = self._dev(query_labels), self._dev(reference_labels)
        if topk == "max_bin_count":
            topk = int(self._match_counts(reference_labels, reference_labels).max().item()) - int(ref_includes_query)
        num_ref = reference.shape[0]
        topk = num_ref if topk is None else min(int(topk), num_ref)  # gnd[0:topk] clips at N
        num_query = query.shape[0]
        if num_query == 0:
            raise ZeroDivisionError("calculate_maphashing: no queries")
        ap = None
        nbits = reference.shape[1]
        if self.rank_cache is None and not self.host and nbits <= 128:
            # one k, nothing to share: ranking and AP in one kernel, the lists never leave the GPU's LDS
            # (same numbers as the two steps below; None = shape outside the fused kernel)
            qlp, rlp = self._packed_labels(query_labels, reference_labels)
            if rlp.shape[1] <= 2:
                prepared = H.PreparedDB(H.pack_codes(reference), nbits)
                fused = H.hamming_map_at_k(H.pack_codes(query), prepared, H.PreparedLabels(rlp), qlp, nbits, topk)
                if fused is not None:
                    ap = fused[0]
                else:                                    # outside the fused kernel: rank with the database already prepared
                    idx = H.hamming_topk(H.pack_codes(query), prepared, nbits, topk, want_dist=False)[0]
                    ap, _ = H.map_at_k(idx, qlp, rlp, k=topk)
        if ap is None:
            idx = self._ranked_lists(query, reference, topk)
            ap, _ = self._average_precisions(idx, query_labels, reference_labels, k=topk)
        result = ap.double().sum().item() / num_query
        if return_per_query:
            return result, ap
        return result

    # ------------------------------------------------------------------ knn metrics
    def calculate_map(self, query_labels, knn_indices, reference_labels, not_lone_query_mask, **kwargs):
        """RetrievalMAP over the k-NN lists (:156-167): AP per kept query, no-hit queries count 0."""
        ap, _ = self._average_precisions(knn_indices.int(), query_labels, reference_labels)
        kept = ap[not_lone_query_mask]
        return kept.double().mean().item() if kept.numel() else 0.0

    def calculate_rpr(self, query_labels, knn_indices, reference_labels, not_lone_query_mask, **kwargs):
        """RetrievalRPrecision over the k-NN lists (:131-142): relevant among the first R / R, R = relevant
        entries of the list; lists in k-NN order (ties: ascending reference index)."""
        hits = self._hits(knn_indices, query_labels, reference_labels)[not_lone_query_mask].long()
        if not hits.numel():
            return 0.0
        R = hits[:, -1]
        top = torch.gather(hits, 1, (R - 1).clamp(min=0).unsqueeze(1)).squeeze(1)
        return torch.where(R > 0, top.double() / R.clamp(min=1).double(), torch.zeros_like(R, dtype=torch.float64)) \
            .mean().item()

    def calculate_pr(self, query_labels, knn_indices, reference_labels, not_lone_query_mask, **kwargs):
        """RetrievalPrecision(top_k=1) (:144-154)."""
        first = self._hits(knn_indices[:, :1], query_labels, reference_labels)[not_lone_query_mask]
        return first.double().mean().item() if first.numel() else 0.0

    def _curves(self, hits):
        """precision@j, recall@j (j = 1..k) averaged over the rows of `hits`; rows without a hit count zero."""
        h = hits.double()
        tot = h[:, -1:]
        j = torch.arange(1, h.shape[1] + 1, dtype=torch.float64, device=h.device)
        has = tot > 0
        prec = torch.where(has, h / j, torch.zeros_like(h)).mean(0)
        rec = torch.where(has, h / tot.clamp(min=1), torch.zeros_like(h)).mean(0)
        return prec, rec

    def _write_pr_rc(self, prec, rec):
        self.last_pr_rc = (prec, rec)
        if self.pr_rc_path:                           # the reference writes ./pr_rc.csv as a side effect
            import pandas as pd
            pd.DataFrame({"pr": prec.cpu().numpy(), "rc": rec.cpu().numpy()}).to_csv(self.pr_rc_path, index=False)

    def calculate_pr_rc(self, query_labels, knn_indices, reference_labels, not_lone_query_mask, **kwargs):
        """RetrievalPrecisionRecallCurve over the k-NN lists (:169-181): writes the curve, returns 0."""
        hits = self._hits(knn_indices, query_labels, reference_labels)[not_lone_query_mask]
        if hits.numel():
            self._write_pr_rc(*self._curves(hits))
        return 0

    def calculate_pr_rc_hashing(self, query, query_labels, reference, reference_labels, not_lone_query_mask=None,
                                **kwargs):
        """Full-gallery precision / recall curves along the Hamming ranking (:235-273), averaged over the
        queries that are not lone and have a relevant item; writes the curve, returns 0."""
        query, reference = self._dev(query), self._dev(reference)
        query_labels, reference_labels = self._dev(query_labels), self._dev(reference_labels)
        hits = self._hits(self._ranked_lists(query, reference, reference.shape[0])[:, :reference.shape[0]], query_labels,
                          reference_labels)
        ok = hits[:, -1] > 0
        if not_lone_query_mask is not None:
            ok &= not_lone_query_mask
        if bool(ok.any()):
            self._write_pr_rc(*self._curves(hits[ok]))
        return 0

    def _knn_relevance(self, query_labels, knn_labels, k):
        return self.label_comparison_fn(query_labels[:, None], knn_labels[:, :k]) if query_labels.ndim > 1 \
            else (query_labels[:, None] == knn_labels[:, :k])

    def recall_at_k(self, knn_labels, query_labels, k):
        return self._knn_relevance(query_labels, knn_labels, k).any(1).float().mean().item()

    def calculate_precision_at_1(self, knn_labels, query_labels, not_lone_query_mask, **kwargs):
        rel = self._knn_relevance(query_labels, knn_labels, 1)[:, 0][not_lone_query_mask]
        return rel.float().mean().item() if rel.numel() else 0.0

    # ------------------------------------------------------------------ driver (:279-349)
    def get_accuracy(self, query, query_labels, reference, reference_labels,
                     embeddings_come_from_same_source, include=(), exclude=(), return_indices=False):
        query, reference = self._dev(query), self._dev(reference)
        query_labels, reference_labels = self._dev(query_labels), self._dev(reference_labels)

        if query_labels.ndim == 1 or (query_labels.ndim == 2 and query_labels.size(1) == 1):
            query_labels = query_labels.view(-1)
            reference_labels = reference_labels.view(-1)

        self.curr_function_dict = self.get_function_dict(include, exclude)

        kwargs = {
            "query": query,
            "reference": reference,
            "query_labels": query_labels,
            "reference_labels": reference_labels,
            "embeddings_come_from_same_source": embeddings_come_from_same_source,
            "label_comparison_fn": self.label_comparison_fn,
            "ref_includes_query": embeddings_come_from_same_source,
            "topk": self.num_top_k,
        }

        knn_indices = None
        if "pr_rc_hashing" in self.get_curr_metrics():
            kwargs["not_lone_query_mask"] = (self._match_counts(query_labels, reference_labels)
                                             - int(embeddings_come_from_same_source)) > 0
        if any(x in self.requires_knn() for x in self.get_curr_metrics()):
            match_counts = self._match_counts(query_labels, reference_labels)
            self_count = int(embeddings_come_from_same_source)
            not_lone_query_mask = (match_counts - self_count) > 0
            num_k = self.determine_k(match_counts, len(reference), embeddings_come_from_same_source)
            if (self.rank_cache is not None and self.distance_metric == "hamming" and reference.shape[1] <= 128
                    and _is_pm1(reference) and _is_pm1(query)):
                knn_indices, knn_distances = self.rank_cache.knn(reference, query, num_k, embeddings_come_from_same_source)
            elif self.host:
                knn_indices, knn_distances = self._host_knn(reference, query, num_k, embeddings_come_from_same_source)
            else:
                knn_indices, knn_distances = get_knn(
                    reference, query, num_k, embeddings_come_from_same_source,
                    with_faiss=self.with_faiss, distance_metric=self.distance_metric,
                )
            if not bool(not_lone_query_mask.any()):
                LOGGER.warning("None of the query labels are in the reference set.")
            kwargs["knn_indices"] = knn_indices
            kwargs["knn_distances"] = knn_distances
            kwargs["not_lone_query_mask"] = not_lone_query_mask
            if any(m.startswith("recall_at_") or m == "precision_at_1" for m in self.get_curr_metrics()):
                kwargs["knn_labels"] = reference_labels[knn_indices[:, :1000]]

        result = self._get_accuracy(self.curr_function_dict, **kwargs)
        if return_indices:
            return knn_indices, result
        return result

    def _host_knn(self, reference, query, num_k, same_source):
        """get_knn (get_knn.py:9-24) on the host: +-1 codes under the hamming metric through the packed twins, everything
        else through wv_knn_float_cpu -- the same lists and values as the GPU path in both cases."""
        from .get_knn import knn_float_host
        num_k += int(same_source)
        nbits = reference.shape[1]
        if num_k > reference.shape[0]:
            raise RuntimeError(f"selected index k out of range (k={num_k}, references={reference.shape[0]})")
        first = int(same_source)
        if self.distance_metric == "hamming" and nbits <= 128 and _is_pm1(reference) and _is_pm1(query):
            idx, dist = HH.hamming_topk(HH.pack_codes(query, check=False), HH.pack_codes(reference, check=False), nbits, num_k)
            return idx[:, first:].long(), (float(nbits) - 2.0 * dist.float())[:, first:]
        if self.distance_metric in ("hamming", "cosine"):
            metric = _lib.WV_METRIC_IP
        else:                                   # faiss IndexFlatL2 returns squared distances, torch.cdist the root
            metric = _lib.WV_METRIC_L2_SQUARED if self.with_faiss else _lib.WV_METRIC_L2
        val, idx = knn_float_host(reference, query, num_k, metric)
        return idx[:, first:].long(), val[:, first:]

    def _get_accuracy(self, function_dict, **kwargs):
        return {k: v(**kwargs) for k, v in function_dict.items()}


def _make_recall(k):
    def calculate(self, knn_labels, query_labels, **kwargs):
        return self.recall_at_k(knn_labels, query_labels, k)
    calculate.__name__ = f"calculate_recall_at_{k}"
    return calculate


for _k in _RECALL_KS:
    setattr(CustomCalculator, f"calculate_recall_at_{_k}", _make_recall(_k))


def get_accuracy_calculator(exclude_ranks=None, k=19581, with_AP=True, **kwargs):
    """Same exclude-list construction as the reference (:352-403)."""
    caller_exclude = kwargs.pop('exclude', [])
    exclude = list(caller_exclude)
    if with_AP:
        exclude.extend(['NMI', 'AMI'])
    else:
        exclude.extend(['NMI', 'AMI', 'mean_average_precision', 'mean_average_precision_at_r'])
    if exclude_ranks:
        for r in exclude_ranks:
            exclude.append(f'recall_at_{r}')
    base_exclude = [
        "mean_reciprocal_rank", "precision_at_1", "recall_at_1", "recall_at_1000", "recall_at_100",
        "recall_at_10", "recall_at_16", "recall_at_20", "recall_at_30", "recall_at_32",
        "recall_at_4", "recall_at_8", "recall_at_2", "recall_at_10", "pr_rc_hashing",
    ]
    exclude = sorted(set(exclude) | set(base_exclude))
    LOGGER.info(f"Excluding metrics: {exclude}")
    return CustomCalculator(exclude=exclude, k=k, **kwargs)
