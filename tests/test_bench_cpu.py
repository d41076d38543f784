"""CPU: bench.py's launch contract and its N > 1 branch, end to end to the JSON line (that branch had never run
anywhere, and `python bench.py --gpus N` without torch.distributed.run printed nothing)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "bench_world2_driver.py")
ARGS = ["--gpus", "2", "--rows", "6001", "--batch", "16", "--steps", "2", "--warmup", "1"]


def _line(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{") and '"metric"' in l]
    assert len(lines) == 1, out[-3000:]                      # exactly ONE JSON line, from rank 0
    return json.loads(lines[0])


def _check(o):
    assert o["n_gpus"] == 2 and o["steps"] == 2 and o["warmup"] == 1 and o["scaling"] == "strong" and o["higher_is_better"] is True
    c = o["config"]
    assert c["n_ranks_seen"] == 2 and c["rows"] == 6001 and c["rows_per_rank"] == 3001 and c["parallelism"].startswith("row-shard x2")
    assert "all_gather_into_tensor" in c["exchange"]        # gloo: the torch.distributed exchange (the RCCL one needs GPUs)
    # the merged answer is the global one: every planted neighbour (all in rank 0's shard) is the top hit, lists are sorted,
    # and the default and exact paths agree on every rank (MIN over ranks)
    assert o["planted_top1"] == 1.0 and o["sorted"] is True and o["identical_to_exact_f32_scan"] is True
    assert o["value"] > 0 and abs(o["value"] - 16 / (o["ms_per_step"] * 1e-3)) / o["value"] < 1e-3
    assert o["secondary"] == [] and o["cpu_baseline"] is None      # N > 1: the headline only


def test_world2_under_torch_distributed_run():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), DRIVER, *ARGS], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    _check(_line(r.stdout))


def test_plain_python_invocation_self_launches_its_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, DRIVER, *ARGS], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    o = _line(r.stdout)
    _check(o)
    assert o["config"]["self_launched"] is True


def test_mismatched_world_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, DRIVER, *ARGS], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2 and "must agree" in r.stderr
