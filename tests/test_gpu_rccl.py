"""The RCCL leg of the tile-parallel path on the one GPU a test box has: a process group of ONE rank
over the "nccl" backend (= RCCL on ROCm), driven exactly like bench.py --gpus N drives N ranks."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_gather_pipeline_with_one_rank():
    """RCCL initialisation, the asynchronous all-gather on RCCL's own stream ordered after the final pass's stream, and
    GatherPipeline.image(), for three pipelined frames: the gathered image equals the frame buffer and the frames of a
    sequential loop, bit for bit. Runs in a fresh child process (a process group belongs to a process)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py"), "3"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0 and "RCCL_ONE_RANK_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])
