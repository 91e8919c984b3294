"""The launch shape the driver uses for its multi-GPU runs — `python -m torch.distributed.run --nproc-per-node N
bench.py ...`, one rank per GPU over RCCL — on however many GPUs this box has: the default bench line and BASELINE
config 4's sharded list (bench.py --config4), each rank decoding the share of one of the node's eight GPUs."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(args, port):
    import torch
    n = max(1, torch.cuda.device_count())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(n), *args]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]     # rank 0 prints ONE json line
    return n, json.loads(lines[0])


def test_config4_under_torchrun_on_every_visible_gpu():
    n, line = _torchrun(["--config4", "--recordings", "16", "--batch", "2", "--steps", "4"], 29541)
    cfg = line["config"]
    assert cfg["shares_measured"] == list(range(min(n, 8)))
    assert cfg["shares_not_measured"] == list(range(min(n, 8), 8))
    assert len(line["per_device"]) == min(n, 8) and line["n_gpus"] == min(n, 8)
    for d in line["per_device"]:
        assert d["recordings"] == 2 and d["host_fed"]["all_decoded"] and d["rows_of_first_recording"] > 1700
        assert d["device_resident"]["value"] > 0 and d["host_fed"]["value"] > 0
    assert line["value"] > 0 and line["value_host_fed"] > 0


def test_default_bench_line_under_torchrun():
    n, line = _torchrun(["--steps", "6", "--warmup", "2", "--seconds", "60", "--batch", "4", "--no-extras",
                         "--no-cpu-baseline"], 29543)
    assert line["n_gpus"] == n and line["steps"] == 6 and line["scaling"] == "weak"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
