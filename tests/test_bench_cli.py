"""bench.py pieces that run without a GPU: the cpu_baseline leg (C oracle on a bounded sample)
and the refusal to run the hot path on a box without an MI355X."""
import json
import subprocess
import sys

from tests.conftest import ROOT


def test_cpu_baseline_leg():
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(L=4, B=1, H=2, T=256, D=64, sample_layers=2)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "GB/s" and cb["value"] > 0
    assert "2/4 layers" in cb["sample"]
    json.dumps(cb)


def test_bench_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    proc = subprocess.run([sys.executable, "bench.py", "--steps", "1"], cwd=ROOT, capture_output=True, text=True)
    assert proc.returncode != 0 and "no CPU path" in (proc.stderr + proc.stdout)
