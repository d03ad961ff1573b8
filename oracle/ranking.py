"""oracle/ranking.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (stock torch CPU ops, the very ops the reference calls) of the ranking half
of the hot path.  Each function cites the reference lines it follows
(/root/reference/main/engine/...):

* ``label_comparison_fn``      accuracy_calculator.py:31-37
* ``calc_hamming_dist``        accuracy_calculator.py:183-186
* ``per_bit_balance`` & co.    accuracy_calculator.py:188-200
* ``calculate_maphashing``     accuracy_calculator.py:203-231   (the reported metric)
* ``get_knn`` / ``get_knn_torch``  get_knn.py:9-24, 60-71
* ``retrieval_map``            accuracy_calculator.py:156-167 (torchmetrics RetrievalMAP;
  torchmetrics is absent here -> its published formula restated, parity unpinned)

The reference modules cannot be imported in this image (pytorch_metric_learning, faiss and
torchmetrics are missing: ordinary ModuleNotFoundError), but these bodies use only torch ops,
so the restatement runs the same kernels.  Golden vectors produced by this file live in
tests/golden/ (made by tests/golden/make_golden.py).

TIE ORDER.  The reference ranks with ``torch.argsort(hamm)`` (default ``stable=False``) and
``torch.topk``; with <= B+1 distinct distances nearly every element is tied and the order
inside a tie is implementation-defined.  ``stable=True`` below selects the canonical
tie-break (ascending database index) that the HIP path implements; ``stable=False`` is the
reference's literal call.  Distances, sorted distance sequences and the index *sets* of
every complete distance bucket are identical between the two.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch


# ----------------------------------------------------------------------------- relevance
def label_comparison_fn(query_labels, reference_labels):
    if query_labels.ndim > 1 and reference_labels.ndim > 1:
        if query_labels.dim() == 2 and reference_labels.dim() == 2:
            return torch.matmul(query_labels.float(), reference_labels.t().float()) > 0
        return (query_labels.float() * reference_labels.float()).sum(dim=-1) > 0
    return query_labels.unsqueeze(1) == reference_labels


# ----------------------------------------------------------------------------- distances
def calc_hamming_dist(qB, rB):
    q = qB.shape[1]
    return 0.5 * (q - torch.matmul(qB, rB.t()))


def per_bit_balance(reference):
    frac_positive = (reference > 0).float().mean(dim=0)
    return 1.0 - 2.0 * (frac_positive - 0.5).abs()


def calculate_bit_balance(reference):
    return per_bit_balance(reference).mean().item()


def calculate_worst_bit_balance(reference):
    return per_bit_balance(reference).min().item()


# ----------------------------------------------------------------------------- mAP@k
def resolve_topk(topk, reference_labels=None, ref_includes_query=False):
    """topk unwrapping of accuracy_calculator.py:204-212 ("max_bin_count" needs PML's
    get_label_match_counts: for 2-D multi-hot labels that is max_r #{r' : shares a tag})."""
    while isinstance(topk, (tuple, list)):
        topk = topk[0] if len(topk) else None
    if topk == "max_bin_count":
        rel = label_comparison_fn(reference_labels, reference_labels)
        topk = int(rel.sum(dim=1).max().item()) - int(ref_includes_query)
    if topk is not None:
        topk = int(topk)
    return topk


def calculate_maphashing(query, query_labels, reference, reference_labels, topk,
                         ref_includes_query=False, stable=False, return_per_query=False):
    topk = resolve_topk(topk, reference_labels, ref_includes_query)
    num_query = query.shape[0]
    topkmap = 0.0
    per_query = []
    for i in range(num_query):
        gnd = label_comparison_fn(query_labels[i:i + 1], reference_labels).float().squeeze()
        hamm = calc_hamming_dist(query[i:i + 1], reference).squeeze()
        indices = torch.argsort(hamm, stable=stable)
        gnd = gnd[indices]
        tgnd = gnd[0:topk]
        tsum = torch.sum(tgnd).int().item()
        ap = 0.0
        if tsum > 0:
            tindex = torch.where(tgnd == 1)[0].float() + 1.0
            count = torch.arange(1, tsum + 1).float()
            ap = torch.mean(count / tindex).item()
        topkmap += ap
        per_query.append(ap)
    result = topkmap / num_query
    if return_per_query:
        return result, per_query
    return result


# ----------------------------------------------------------------------------- k-NN
def get_knn_torch(references, queries, num_k, distance_metric="l2"):
    if distance_metric in ["hamming", "cosine"]:
        scores = queries @ references.t()
        distances, indices = torch.topk(scores, num_k, largest=True)
    else:
        dist_matrix = torch.cdist(queries, references, p=2)
        distances, indices = torch.topk(dist_matrix, num_k, largest=False)
    return distances, indices


def get_knn(references, queries, num_k, embeddings_come_from_same_source, with_faiss=False,
            distance_metric="l2"):
    num_k += embeddings_come_from_same_source
    distances, indices = get_knn_torch(references, queries, num_k, distance_metric)
    if embeddings_come_from_same_source:
        return indices[:, 1:], distances[:, 1:]
    return indices, distances


# ----------------------------------------------------------------------------- canonical forms
def hamming_matrix_u8(query, reference):
    """[Q,N] integer Hamming distances of +-1 codes (exact: fp32 matmul of small integers)."""
    d = calc_hamming_dist(query.float(), reference.float())
    return d.round().to(torch.int64)


def hamming_topk_stable(query, reference, k):
    """Canonical tie-break: ascending distance, then ascending database index."""
    d = hamming_matrix_u8(query, reference)
    order = torch.argsort(d, dim=1, stable=True)[:, :k]
    return order, torch.gather(d, 1, order)


def knn_stable(references, queries, num_k, distance_metric="l2"):
    """get_knn_torch with the canonical tie-break (value order, then ascending index)."""
    if distance_metric in ["hamming", "cosine"]:
        scores = queries @ references.t()
        order = torch.argsort(-scores, dim=1, stable=True)[:, :num_k]
        return torch.gather(scores, 1, order), order
    dist_matrix = torch.cdist(queries, references, p=2)
    order = torch.argsort(dist_matrix, dim=1, stable=True)[:, :num_k]
    return torch.gather(dist_matrix, 1, order), order


def bucket_sets(indices_row, dist_row):
    """{distance: frozenset(indices)} for every bucket except the last (possibly truncated)."""
    out = {}
    ds = dist_row.tolist()
    idx = indices_row.tolist()
    last = ds[-1] if ds else None
    for d, i in zip(ds, idx):
        if d == last:
            continue
        out.setdefault(d, set()).add(i)
    return {d: frozenset(s) for d, s in out.items()}


# ----------------------------------------------------------------------------- map_level0
def map_tie_bounds(dist, gnd, k):
    """[lowest, highest] mAP@k any ordering can give that sorts by distance and breaks ties arbitrarily -- what separates
    the reference's `torch.argsort(hamm)` (unstable: the order inside a distance bucket is implementation-defined,
    accuracy_calculator.py:220) from the canonical (distance, index) order.  dist: integer distances [Q, N]; gnd: bool
    relevance [Q, N].  Inside a complete bucket the hits' precision terms are largest with the relevant rows first and
    smallest with them last (the hit counts at the bucket's borders are fixed); for the bucket the cut at k falls into,
    every possible number of relevant rows among the kept ones is tried (leaving a late hit out can RAISE the mean)."""
    import numpy as np
    dist, gnd = np.asarray(dist).astype(np.int64), np.asarray(gnd).astype(bool)
    Q, N = dist.shape
    k = N if k is None else min(int(k), N)
    lo_sum = hi_sum = 0.0

    def terms(start_rank, hits_before, n_rel, n_slots, first):
        """sum of j / rank over n_rel hits placed at the first / last of n_slots consecutive ranks (1-based start_rank)"""
        ranks = np.arange(n_rel) + (start_rank if first else start_rank + n_slots - n_rel)
        return float(((hits_before + 1 + np.arange(n_rel)) / ranks.astype(np.float64)).sum())

    for i in range(Q):
        lo_opts, hi_opts = [(0.0, 0)], [(0.0, 0)]               # (sum of terms, hits) reachable so far: min / max tracks
        pos, hits = 0, 0
        s_lo = s_hi = 0.0
        for b in np.unique(dist[i]):
            m = dist[i] == b
            size, rel = int(m.sum()), int((m & gnd[i]).sum())
            if pos + size <= k:                                   # complete bucket
                s_hi += terms(pos + 1, hits, rel, size, True)
                s_lo += terms(pos + 1, hits, rel, size, False)
                pos, hits = pos + size, hits + rel
                if pos == k:
                    lo_opts, hi_opts = [(s_lo, hits)], [(s_hi, hits)]
                    break
            else:                                                 # the cut at k: `slots` of the bucket's rows are kept
                slots = k - pos
                cand_lo, cand_hi = [], []
                for h in range(max(0, slots - (size - rel)), min(rel, slots) + 1):
                    cand_hi.append((s_hi + terms(pos + 1, hits, h, slots, True), hits + h))
                    cand_lo.append((s_lo + terms(pos + 1, hits, h, slots, False), hits + h))
                lo_opts, hi_opts = cand_lo, cand_hi
                break
        else:
            lo_opts, hi_opts = [(s_lo, hits)], [(s_hi, hits)]
        ap = lambda sh: sh[0] / sh[1] if sh[1] else 0.0          # noqa: E731
        lo_sum += min(ap(x) for x in lo_opts)
        hi_sum += max(ap(x) for x in hi_opts)
    return lo_sum / Q, hi_sum / Q


def retrieval_map(knn_scores, relevances, not_lone_query_mask=None):
    """torchmetrics RetrievalMAP over (preds, target, indexes) as built at
    accuracy_calculator.py:156-167: per query AP over its k retrieved items ordered by
    descending pred, queries without a positive count 0 (empty_target_action='neg'), mean
    over the kept queries.  Canonical tie-break = input (k-NN) order."""
    Q = knn_scores.shape[0]
    keep = torch.ones(Q, dtype=torch.bool) if not_lone_query_mask is None else not_lone_query_mask
    aps = []
    for i in range(Q):
        if not bool(keep[i]):
            continue
        order = torch.argsort(-knn_scores[i], stable=True)
        t = relevances[i][order]
        if not t.sum():
            aps.append(0.0)
            continue
        positions = torch.arange(1, len(t) + 1, dtype=torch.float32)[t > 0]
        aps.append(((torch.arange(len(positions), dtype=torch.float32) + 1) / positions).mean().item())
    return float(sum(aps) / max(len(aps), 1))


def retrieval_rprecision(relevances, not_lone_query_mask=None):
    """torchmetrics RetrievalRPrecision as called at accuracy_calculator.py:131-142: per query R = number of
    relevant items in its list, value = relevant among the first R / R (0 when R = 0), mean over kept queries.
    Lists are taken in k-NN order (canonical tie-break; the reference perturbs tied scores by 1e-8 * position)."""
    Q = relevances.shape[0]
    keep = torch.ones(Q, dtype=torch.bool) if not_lone_query_mask is None else not_lone_query_mask
    vals = []
    for i in range(Q):
        if not bool(keep[i]):
            continue
        t = relevances[i].bool()
        R = int(t.sum())
        vals.append(float(t[:R].sum()) / R if R else 0.0)
    return float(sum(vals) / max(len(vals), 1))


def retrieval_precision_at_1(relevances, not_lone_query_mask=None):
    """RetrievalPrecision(top_k=1) (accuracy_calculator.py:144-154)."""
    keep = torch.ones(relevances.shape[0], dtype=torch.bool) if not_lone_query_mask is None else not_lone_query_mask
    r = relevances[keep][:, 0].float()
    return float(r.mean()) if r.numel() else 0.0


def retrieval_pr_curve(relevances, not_lone_query_mask=None):
    """RetrievalPrecisionRecallCurve (accuracy_calculator.py:169-181): precision@j and recall@j for
    j = 1..k averaged over the kept queries, queries without a positive contributing zeros."""
    keep = torch.ones(relevances.shape[0], dtype=torch.bool) if not_lone_query_mask is None else not_lone_query_mask
    t = relevances[keep].double()
    k = t.shape[1]
    csum = t.cumsum(1)
    j = torch.arange(1, k + 1, dtype=torch.float64)
    tot = csum[:, -1:]
    prec = csum / j
    rec = torch.where(tot > 0, csum / tot.clamp(min=1), torch.zeros_like(csum))
    prec = torch.where(tot > 0, prec, torch.zeros_like(prec))
    return prec.mean(0), rec.mean(0)


def pr_rc_hashing(query, query_labels, reference, reference_labels, not_lone_query_mask, stable=True):
    """calculate_pr_rc_hashing (accuracy_calculator.py:235-273): per query the full-gallery precision and
    recall curves along the Hamming ranking; mean over the queries that are not lone and reach recall 1.
    Returns (precision [N], recall [N]) or None when no query qualifies."""
    nq, ng = query.shape[0], reference.shape[0]
    all_prec = torch.zeros((nq, ng))
    all_rec = torch.zeros((nq, ng))
    for i in range(nq):
        gnd = label_comparison_fn(query_labels[i:i + 1], reference_labels).float().squeeze()
        hamm = calc_hamming_dist(query[i:i + 1], reference).squeeze()
        gnd = gnd[torch.argsort(hamm, stable=stable)]
        tot = gnd.sum()
        if tot > 0:
            c = torch.cumsum(gnd, 0)
            all_prec[i] = c / torch.arange(1, ng + 1).float()
            all_rec[i] = c / tot
    ok = (all_rec[:, -1] == 1.0) & not_lone_query_mask
    if not bool(ok.any()):
        return None
    return all_prec[ok].mean(0), all_rec[ok].mean(0)


# ----------------------------------------------------------------------------- synthetic inputs
def make_codes(n_query, n_db, nbits, seed=0):
    """Random +-1 codes exactly as studies/measure_random_baseline.py:84,105-106 builds them
    (queries first, then database, one generator)."""
    g = torch.Generator().manual_seed(seed)
    q = torch.randint(0, 2, (n_query, nbits), generator=g).float() * 2 - 1
    r = torch.randint(0, 2, (n_db, nbits), generator=g).float() * 2 - 1
    return q, r


def make_labels(n, n_classes, p, seed):
    """Multi-hot fp32 labels, Bernoulli(p) per tag, at least one active tag."""
    g = torch.Generator().manual_seed(seed)
    lab = (torch.rand(n, n_classes, generator=g) < p).float()
    empty = lab.sum(1) == 0
    fill = torch.randint(0, n_classes, (n,), generator=g)
    lab[empty, fill[empty]] = 1.0
    return lab


def make_structured_codes(labels, nbits, w_seed, noise_seed, noise=0.5):
    """Label-correlated codes sign(labels @ W + noise * N(0,1)) -> mAP well above the floor.
    Queries and database share `w_seed` and differ in `noise_seed`."""
    W = torch.randn(labels.shape[1], nbits, generator=torch.Generator().manual_seed(w_seed))
    g = torch.Generator().manual_seed(noise_seed)
    z = labels @ W + noise * torch.randn(labels.shape[0], nbits, generator=g)
    c = torch.sign(z)
    c[c == 0] = 1.0
    return c
