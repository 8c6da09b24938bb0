import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.multiprocessing as mp
from tests import test_gpu_rccl_single_rank as t
if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    worst = {}
    for rep in range(6):
        q = ctx.Queue()
        p = ctx.Process(target=t._worker, args=(t._free_port(), q))
        p.start()
        status, report = q.get(timeout=600)
        p.join(timeout=120)
        assert status == "ok", report
        for name, r in report.items():
            if not isinstance(r, dict) or "forced_vs_whole" not in r:
                continue
            w = worst.setdefault(name, {"forced_vs_whole": 0.0, "ratio": 0.0, "run_to_run_rel": 0.0})
            w["forced_vs_whole"] = max(w["forced_vs_whole"], r["forced_vs_whole"])
            w["run_to_run_rel"] = max(w["run_to_run_rel"], r["run_to_run_rel"])
            if r["run_to_run_max"] > 0:
                w["ratio"] = max(w["ratio"], r["staged_vs_forced_rel"] / max(r["run_to_run_rel"], 1e-30))
    for k, v in worst.items():
        print(k, {a: float(f"{b:.3g}") for a, b in v.items()})
