"""GPU, 2 ranks: the row-sharded search with its one RCCL exchange on real hardware.  Self-skips when fewer
than two devices are visible (gpurun's box has one; the driver's 8-GPU node runs it)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_rank_sharded_search_native_and_torch_exchange():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29577", os.path.join(ROOT, "tests", "two_rank_driver.py")],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "TWO_RANK_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
