"""bench.py's own launcher (round 5): `python3 bench.py --gpus N` without WORLD_SIZE / RANK in the environment starts the one-rank-per-GPU job itself, as a CHILD
process and before torch is imported, and hands back the child's exit code.  Without a GPU the ranks refuse to run ("bench.py needs an MI355X"): what this CPU
test sees is the launch itself and the relayed failure.  (With a GPU: tests/test_cd_gpu.py::test_bench_started_plainly_launches_its_own_ranks.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_start_launches_ranks_and_relays_their_exit_code():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the GPU suite runs the real thing")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--quads", "8"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert "launching" in out.stderr and "torch.distributed.run" in out.stderr and "--nproc-per-node 2" in out.stderr
    assert out.returncode != 0                                             # the ranks' refusal comes back as the launcher's code
    assert "needs an MI355X" in out.stderr
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]   # and no line is invented


def test_launcher_is_not_used_under_a_launcher():
    """With WORLD_SIZE set (the driver's own torch.distributed.run) the script is a rank, not a launcher: a world that does not match --gpus is an error of its own."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29599")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "launching" not in out.stderr and "WORLD_SIZE=1" in out.stderr
