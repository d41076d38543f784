"""CPU: bench.py's launch contract and its N > 1 branch, end to end to the JSON line (that branch had never run
anywhere, and `python bench.py --gpus N` without torch.distributed.run printed nothing)."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DRIVER = os.path.join(HERE, "bench_world2_driver.py")
ARGS = ["--gpus", "2", "--rows", "6001", "--batch", "16", "--steps", "2", "--warmup", "1"]


def _strict(text: str) -> dict:
    def bad(c):
        raise ValueError(f"non-finite constant {c} in the bench line")
    return json.loads(text, parse_constant=bad)


def _line(out: str) -> dict:
    lines = out.splitlines()
    assert len(lines) == 1, out[-3000:]                      # stdout holds exactly ONE line (rank 0's JSON) and nothing else
    assert len(lines[0].encode()) <= 8192, len(lines[0])     # a 23-KB line was not parsed by the driver
    return _strict(lines[0])


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


def test_compact_line_bounds_a_real_full_record():
    """The stdout line built from round 5's real 23-KB record (14 secondary legs): <= 8 KB, strict JSON, every key the bench
    contract and the tier's measurement section name; the full record goes to bench_secondary.json, not to stdout."""
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    full = json.load(open(os.path.join(os.path.dirname(HERE), "profiles", "r05_bench_full_final.json")))
    ids = "exact b128 b32 b16 b1 emu8 c2 l2 c1 mmr chat embed index rerank".split()
    for leg, i in zip(full["secondary"], ids):
        leg["id"] = i
    assert len(json.dumps(full)) > 20000
    text = json.dumps(bench.compact_line(full), allow_nan=False)
    assert len(text.encode()) <= bench.LINE_BUDGET <= 8192 and "\n" not in text
    o = _strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"):
        assert k in o, k
    assert o["config"]["workload"].startswith("10000000x384")
    r = o["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and r["unit"] in ("GB/s", "TFLOP/s") and "traffic" in r
    assert r["north_star"]["batch"] in (16, 32) and r["emulated_shard_8"]["emulated"] is True
    c = o["cpu_baseline"]
    assert c["value"] > 0 and c["cores"] >= 1 and c["kind"] in ("port", "reference") and c["sample"]
    assert [s["id"] for s in o["secondary"]] == ids
    for s in o["secondary"]:
        assert len(s["id"]) <= 40 and s["value"] > 0 and s["roofline"]["frac"] > 0
    # a pathological record (huge strings, 60 legs) still fits: detail is dropped in steps
    fat = dict(full, secondary=[dict(full["secondary"][i % 14], id=f"leg{i}", note="x" * 5000) for i in range(60)])
    assert len(json.dumps(bench.compact_line(fat), allow_nan=False).encode()) <= bench.LINE_BUDGET
    # non-finite values never reach the line
    assert bench._denan({"a": float("nan"), "b": [float("inf"), 1.0]}) == {"a": None, "b": [None, 1.0]}


def test_stdout_guard_sends_stray_output_to_stderr():
    code = ("import os, sys, ctypes; sys.path.insert(0, %r); import bench; g = bench._StdoutGuard(); "
            "print('stray python print'); os.write(1, b'stray fd write\\n'); ctypes.CDLL(None).puts(b'stray C puts'); g.emit('{\"ok\": 1}')") % os.path.dirname(HERE)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout == '{"ok": 1}\n', r.stdout
    assert "stray python print" in r.stderr and "stray fd write" in r.stderr and "stray C puts" in r.stderr
