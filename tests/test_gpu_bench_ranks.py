"""bench.py with N > 1, rehearsed on the one-GPU box: two ranks in fresh child processes (started through
torch.distributed.run before anything in them touches the GPU), both on HIP device 0, torch.distributed over gloo and
the engine's collectives over the host-staged shm transport (`--one-gpu`).  No rate is asserted -- two ranks share one
device -- only that the N > 1 legs run the problems they say they run and describe themselves:
the consensus leg is the 8-slice problem (4 local slices per rank), one partition of the rows for every leg,
collective counts and payloads per leg, the scaling basis."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("transport", ["shm", "p2p"])
def test_bench_two_ranks_one_gpu_json_shape(gpu, transport):
    rows, cols = 4100, 512  # 4100 = 8*512 + 4: slicemaker(0, 8, .) gives four slices of 513 and four of 512
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", OPENBLAS_NUM_THREADS="4")
    if transport == "shm":
        env["ADMM_BENCH_P2P_LEG"] = "1"  # the extra leg an RCCL run adds: the consensus problem once more over P2P
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), "bench.py", "--gpus", "2", "--one-gpu", "--steps", "5",
           "--warmup", "2", "--no-cpu-baseline", "--rows", str(rows), "--cols", str(cols), "--transport", transport]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 2
    assert out["config"]["communicator"]["ranks"] == 2 and out["config"]["communicator"]["transport"] == transport
    assert out["config"]["rows"] == rows and out["config"]["cols"] == cols
    assert "extras_error" not in out, out.get("extras_error")
    cons = out["consensus_lasso"]
    assert cons["slices_total"] == 8 and cons["slices_per_gpu"] == 4
    assert cons["collectives_per_iter"] == 1 and cons["allreduce_doubles_per_iter"] == 2 * cols + 1
    basis = out["scaling_basis"]
    assert set(basis["strong_scaling_legs"]) <= set(out)
    assert basis["collectives_per_iter"]["headline"] == 0 and basis["collectives_per_iter"]["a_streaming"] == 1
    for leg in ("a_streaming", "objevals1_literal", "matrix_free"):
        assert out[leg]["iters_per_s"] > 0
    assert out["value"] > 0 and out["roofline"]["frac"] is not None
    lat = out["config"]["communicator"]["allreduce_us"]
    assert set(lat) == {"1_doubles", f"{cols}_doubles", f"{2 * cols + 1}_doubles", f"{3 * cols + 16}_doubles"}
    if transport == "shm":
        extra = out["consensus_lasso_p2p"]
        assert "error" not in extra, extra
        assert extra["slices_total"] == 8 and extra["allreduce_us"][f"{cols}_doubles"] < lat[f"{cols}_doubles"]
