"""The sharded search with the REAL kernels on every rank: two processes share this box's one GPU, the collectives run
over gloo (host-staged, like bench.py's rehearsal) -- RCCL needs one GPU per rank and cannot run here.  Complements
tests/test_parallel_gloo.py, where the kernels are replaced by oracle stand-ins."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash import parallel, synth
    from wvhash.engine import hamming as H
    out = {}
    # last case: the c3 shard shape (COCO: 117,218 rows over 8 GPUs = 14,653 rows per shard, 128-bit codes, 80 classes = TWO
    # label words per row, k = 5000) with as many shards as there are ranks here
    for n_db, nbits, k, ql, prepared, lc, p in ((25000, 64, 5000, 96, True, 38, 0.1), (1001, 128, 600, 7, False, 38, 0.1),
                                                (40000, 64, 3000, 5, True, 38, 0.1), (14653 * world, 128, 5000, 40, True, 80, 0.036)):
        labels_q = synth.multi_hot_labels(world * ql, lc, p, 1)
        labels_r = synth.multi_hot_labels(n_db, lc, p, 2)
        q, r = synth.structured_codes(labels_q, nbits, 3, 4), synth.structured_codes(labels_r, nbits, 3, 5)
        lo, hi, _ = parallel.shard_bounds(n_db, world, rank)
        qp = H.pack_codes(q[rank * ql:(rank + 1) * ql].cuda())
        shard = H.pack_codes(r[lo:hi].cuda())
        shard = H.PreparedDB(shard, nbits) if prepared else shard
        idx, d, need = parallel.sharded_hamming_topk(qp, shard, nbits, k, n_db, return_need=True)
        hint = min(k, (int(need.item()) * 9 // 8 + 63) // 64 * 64)
        idx_h, d_h, need_h = parallel.sharded_hamming_topk(qp, shard, nbits, k, n_db, send_hint=hint, return_need=True)
        assert parallel.exchange_ok([need_h], hint, min(k, hi - lo + 1))
        full_idx, full_d = H.hamming_topk(H.pack_codes(q[rank * ql:(rank + 1) * ql].cuda()), H.pack_codes(r.cuda()), nbits, k)
        # mAP without lists on the wire: relevance strings + histograms, merged on the receiving rank
        qlp = H.pack_labels(labels_q[rank * ql:(rank + 1) * ql].cuda())
        rlp = H.pack_labels(labels_r.cuda())
        ap_ref, nrel_ref = H.map_at_k(full_idx, qlp, rlp)
        map_ok = None
        if prepared and hi - lo <= H.SHARD_ROWS_MAX:
            got = parallel.sharded_hamming_map_at_k(qp, qlp, shard, H.PreparedLabels(rlp[lo:hi].contiguous()), nbits, k, n_db, hint)
            ap, nrel, need_m = got
            map_ok = (torch.equal(ap, ap_ref) and torch.equal(nrel, nrel_ref) and parallel.exchange_ok([need_m], hint, min(k, hi - lo + 1)))
            # ... and sized exactly by the call itself (send_hint=None: what evaluate_sharded uses)
            ap_x, nrel_x, _ = parallel.sharded_hamming_map_at_k(qp, qlp, shard, H.PreparedLabels(rlp[lo:hi].contiguous()), nbits, k, n_db, None)
            map_ok = map_ok and torch.equal(ap_x, ap_ref) and torch.equal(nrel_x, nrel_ref)
        out[(n_db, nbits, k, lc)] = (torch.equal(idx, full_idx) and torch.equal(d, full_d),
                                 torch.equal(idx_h, full_idx) and torch.equal(d_h, full_d), int(need.item()), hint, map_ok)
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_real_kernels(tmp_path):
    port = 29700 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, f"r{rank}.pt"))
        assert len(got) == 4
        for key, (exact, hinted, need, hint, map_ok) in got.items():
            assert exact and hinted, (rank, key)
            assert need <= hint
            assert map_ok is not False, (rank, key)                  # None: shape not taken by the relevance-string path
        assert sum(v[4] is True for v in got.values()) >= 2
        assert got[(14653 * 2, 128, 5000, 80)][4] is True            # c3: two label words through parallel.py, AP bit-identical


def _eval_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from test_gpu_dropin import NODES, SynthHashing, make_net
    from wvhash.engine import evaluate, evaluate_sharded
    from wvhash.transforms import build_transform
    out = {}
    for defer, classes in ((True, 38), (False, 38), (True, 80)):      # 80 classes (COCO): two label words per row
        tf = build_transform(NODES[0], defer=defer)
        dts = {"test": SynthHashing(11, 1, tf, classes=classes),          # ragged: 11 queries, 45 rows / 2 ranks
               "gallery": SynthHashing(45, 2, tf, classes=classes)}
        net = make_net()
        m = evaluate_sharded(net, dts, k=20, epoch=2, batch_size=8, num_workers=2 if not defer else 0,
                             distance_metric="hamming")
        single = evaluate(net, test_dataset=dts, k=20, epoch=2, batch_size=8, num_workers=0, distance_metric="hamming",
                          exclude=["precision_at_1", "rpr", "pr", "pr_rc", "mean_reciprocal_rank", "r_precision"])
        out[(defer, classes)] = (m["test"], {k_: v for k_, v in single["test"].items() if k_ in m["test"]})

    torch.save(out, os.path.join(out_dir, f"e{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluate_sharded_equals_single_gpu_evaluate(tmp_path):
    """evaluate_sharded on two ranks (each embeds its slice of the database and of the queries, sharded search, gathered
    labels, all-reduced sums) reports the numbers of evaluate() -- with deferred raw batches and with the unedited
    transform node running in DataLoader workers."""
    port = 29800 + os.getpid() % 2000
    mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        got = torch.load(os.path.join(tmp_path, f"e{rank}.pt"))
        for defer, (sharded, single) in got.items():
            assert sharded["epoch"] == single["epoch"] == "2"
            for key in ("maphashing_level0", "map_level0", "bit_balance_level0", "worst_bit_balance_level0"):
                assert abs(sharded[key] - single[key]) < 1e-6, (rank, defer, key, sharded[key], single[key])


def _run_bench(extra, env_extra=None, timeout=420):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WV_DIST_BACKEND")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, capture_output=True, text=True,
                          timeout=timeout)


@pytest.mark.parametrize("gpus", [2, 4])
def test_bench_launches_its_own_ranks(tmp_path, gpus):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment (how the driver calls it) starts N ranks itself and
    prints ONE JSON line; on this 1-GPU box the ranks share the GPU and the collectives are staged over gloo, which the
    line says.  (4 ranks + this process = 5 of the 6 GPU processes a box allows; the 8-rank exchange is rehearsed on the
    CPU with kernel stand-ins, tests/test_parallel_gloo.py.)  A steady-state step issues exactly one all_gather and one
    all_to_all; the line carries every rank's kernel / collective milliseconds."""
    import json
    out = _run_bench(["--gpus", str(gpus), "--steps", "3", "--warmup", "1", "--queries", "256"])
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == gpus and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    ex = rec["config"]["exchange"]
    assert ex["verified_exact"] is True
    assert ex["collectives_per_timed_step"] == {"all_gather": 1.0, "all_to_all": 1.0, "all_reduce": 0.0}
    assert ex["one_all_gather_one_all_to_all_per_step"] is True and ex["prefix_entries"] >= 1
    assert ex["ap_equals_unsharded_kernel_on_every_rank"] is True
    assert [r["rank"] for r in ex["per_rank"]] == list(range(gpus))
    for r in ex["per_rank"]:
        assert r["step_ms"] > 0 and r["kernels_ms"] > 0 and set(r["collectives_ms"]) <= {"all_gather", "all_to_all"}
        assert r["bytes_sent_per_step"]["all_to_all"] > 0
    if torch.cuda.device_count() < gpus:
        assert "REHEARSAL" in rec["config"]["backend"]
    else:
        assert ex["host_syncs_in_timed_steps"] == 0


def test_bench_rank_failing_in_setup_ends_the_run_with_its_reason():
    """A rank whose setup throws says why on stderr, the others learn it in the setup handshake, and the launch exits
    non-zero within seconds instead of hanging in the first collective."""
    import time
    t0 = time.time()
    out = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--queries", "128"], {"WV_BENCH_FAIL_SETUP_RANK": "1"}, timeout=240)
    assert out.returncode != 0 and time.time() - t0 < 200
    assert "rank 1] setup failed: RuntimeError: injected setup failure" in out.stderr
    assert "rank 0] another rank failed during setup" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_single_gpu_bench_line_carries_the_untuned_number_and_no_host_sync():
    import json
    out = _run_bench(["--steps", "3", "--warmup", "1", "--queries", "256", "--no-cpu-baseline", "--no-grid", "--clock-steps", "2"])
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["value_first_allocation"] > 0 and rec["first_allocation"]["ms_per_step"] > 0
    assert rec["config"]["host_syncs_in_timed_steps"] == 0
    assert rec["roofline"]["frac"] > 0 and rec["config"]["swt_output_placement"]["candidates"] >= 1


# ----------------------------------------------------------------------------------------- real-valued k-NN, row-sharded
def _knn_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "image-retrieval-wavelet_amd"))
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from wvhash import parallel, _lib
    from wvhash.engine.get_knn import knn_float
    ok = {}
    for n_db, D, ql, k in ((25000, 64, 48, 5000), (1001, 20, 5, 700), (40000, 384, 9, 100)):
        g = torch.Generator().manual_seed(n_db)
        q_all, r = torch.randn(world * ql, D, generator=g), torch.randn(n_db, D, generator=g)
        r[n_db // 2:n_db // 2 + 40] = r[:40].clone()              # ties that straddle the shards
        lo, hi, _ = parallel.shard_bounds(n_db, world, rank)
        mine = q_all[rank * ql:(rank + 1) * ql].cuda()
        for metric in (_lib.WV_METRIC_IP, _lib.WV_METRIC_L2, _lib.WV_METRIC_L2_SQUARED):
            v, i = parallel.sharded_knn_float(mine, r[lo:hi].cuda(), k, metric, n_db)
            v0, i0 = knn_float(r.cuda(), mine, k, metric)
            ok[(n_db, metric)] = torch.equal(i, i0) and torch.equal(v.view(torch.int32), v0.view(torch.int32))
    torch.save(ok, os.path.join(out_dir, f"knn{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sharded_float_knn_real_kernels(tmp_path):
    """wvhash.parallel.sharded_knn_float with wv_knn_float and wv_rank_scores on both ranks (one GPU, gloo staging): the merged
    lists equal the unsharded search bit for bit, every metric, k above a shard's share (5000 of 12,500 rows) and below."""
    port = 34100 + os.getpid() % 2000
    mp.spawn(_knn_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for rank in range(2):
        ok = torch.load(os.path.join(tmp_path, f"knn{rank}.pt"))
        assert len(ok) == 9 and all(ok.values()), ok
