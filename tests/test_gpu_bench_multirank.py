"""bench.py's N > 1 path (one process per GPU, barrier + max-over-ranks timing, the three scaling legs with their exchanges),
executed with TWO ranks on the ONE MI355X of the test box: `--one-device-gloo` puts both ranks on cuda:0 and runs the exchanges
over gloo (RCCL refuses two ranks on one device; the RCCL calls themselves are covered by tests/test_gpu_rccl_single_rank.py).
The numbers mean nothing — two processes share the GPU — the contract and the sharded code paths are what is checked."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_bench_line_on_one_device():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--one-device-gloo"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "particle-shard x2" and out["config"]["particles_per_gpu"] == 1_000_000
    assert out["value"] == pytest.approx(2 * 1e6 * 100 / (out["ms_per_step"] * 1e-3), rel=1e-6)   # whole-job aggregate
    legs = out["scaling_legs"]
    assert legs["n_gpus"] == 2
    assert legs["c3_batch_shard"]["settings_per_rank"] == 2048 and legs["c3_batch_shard"]["collectives"] == "none"
    assert legs["c2_strong"]["particles_total"] == 1_000_000 and legs["c4_particle_shard"]["particles_total"] == 1_000_000
    # the global sigma_x of the sharded headline beam equals the single-rank one (exact merge of the ranks' moments)
    assert out["config"]["sigma_x_out"] == pytest.approx(1.9558527e-4, rel=2e-3)


def test_eight_rank_bench_line_on_one_device():
    """The driver's scaling run is --gpus 8: the same contract with EIGHT ranks (all on cuda:0, exchanges over gloo): one line
    from rank 0, the whole-job aggregate, the 4096 settings of C3 split 8 x 512, the particles of C2 / C4 split 8 x 1.25e5, the
    global sigma_x from the merge of eight ranks' moments, and the `distributed` block naming every rank's device."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
           "--one-device-gloo"]
    res = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 2 and out["scaling"] == "weak"
    assert out["config"]["parallelism"] == "particle-shard x8" and out["config"]["particles_per_gpu"] == 1_000_000
    assert out["value"] == pytest.approx(8 * 1e6 * 100 / (out["ms_per_step"] * 1e-3), rel=1e-6)
    assert out["config"]["sigma_x_out"] == pytest.approx(1.9558527e-4, rel=2e-3) and out["config"]["sigma_x_checked"] is True
    legs = out["scaling_legs"]
    assert legs["n_gpus"] == 8
    assert legs["c3_batch_shard"]["settings_per_rank"] == 512 and legs["c3_batch_shard"]["collectives"] == "none"
    assert legs["c2_strong"]["particles_per_rank"] == 125_000 and legs["c2_strong"]["fused_in_register"]["ms_per_step"] > 0
    assert legs["c4_particle_shard"]["particles_total"] == 1_000_000
    d = out["distributed"]
    assert d["initialized"] is True and d["world_size"] == 8 and d["backend"] == "gloo"
    assert [r["rank"] for r in d["ranks"]] == list(range(8)) and d["distinct_devices"] == 1       # (one device: this box has one GPU)
