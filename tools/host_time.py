import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.argv = ["bench.py"]
import torch, bench
p = bench.Pipeline(2048, 0, 1, torch.device("cuda", 0))
for _ in range(3): p.step()
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(20): p.step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue per step: {(t1-t0)/20*1e3:.3f} ms; total per step {(t2-t0)/20*1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(20): p.step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
