"""bench.py pieces that run without a GPU: the cpu_baseline leg (CPU restatements on a bounded
sample), the refusal to run the hot path on a box without an MI355X, and the ``--gpus N``
self-launcher (children that only join the process group and report rank / world)."""
import json
import os
import subprocess
import sys

import pytest
import torch

from tests.conftest import ROOT


def _bench(*argv, env=None, timeout=300):
    return subprocess.run([sys.executable, "bench.py", *argv], cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                          env=dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
                                   **(env or {})))


def test_cpu_baseline_leg():
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(L=4, B=1, H=2, T=256, D=64, sample_layers=2, reps=2)
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["unit"] == "GB/s" and cb["value"] > 0
    assert "2/4 layers" in cb["sample"]
    for leg in ("port", "vectorised", "literal"):  # the three CPU figures, each with cores + reps
        assert cb[leg]["value"] > 0 and cb[leg]["cores"] >= 1 and cb[leg]["reps"] >= 2, leg
    from oracle import hostcpu
    usable = hostcpu.usable_cores()
    assert 1 <= usable["usable"] <= usable["machine"] == (os.cpu_count() or 1) and usable["usable"] <= usable["affinity"]
    assert cb["vectorised"]["cores"] == usable["usable"] == cb["host_cores_available"] and cb["host_cpu"]["usable"] == usable["usable"]
    # the port on 1 core AND on every usable core; every entry `reps` repetitions; the torch entries say
    # which pool size they were timed at; no median may sit on a scheduler quantum unnoticed
    assert cb["port"]["cores"] == 1
    if usable["usable"] > 1:
        ac = cb["port"]["all_cores"]
        assert ac["cores"] == usable["usable"] and ac["value"] > 0 and ac["quantise_value"] > 0 and ac["int8"]["dequantise_value"] > 0
        assert f"over {usable['usable']} pthreads" in ac["sample"]
    assert isinstance(cb["timer_quantum_suspects"], list) and cb["eviction"]["pool_reps"] == cb["eviction"]["reps"] == 2
    assert set(cb["eviction"]["torch_threads"]) == {"window_torch", "pool_torch"} and cb["eviction"]["pool_port"]["cores_1"] > 0
    assert all(v["used"] in (1, usable["usable"]) for v in cb["vectorised"]["torch_threads"].values())
    assert cb["eviction"]["pool_value"] > 0 and cb["eviction"]["window_value"] > 0 and cb["vectorised"]["int8"]["quantise_value"] > 0
    json.dumps(cb)
    # the per-op parity section: handed what a correct GPU run would hand it (here: the oracle's own outputs), every op
    # reports bit-exact; one flipped output byte is seen
    import numpy as np
    from oracle import c_oracle as C
    from oracle import kvq_oracle as O
    rng = np.random.default_rng(3)
    B, H, T, D = 1, 2, 512, 64
    chk = {"kinds": ("int8", "int4")}
    for name, kind in (("k", "int8"), ("v", "int4")):
        x = rng.standard_normal((1, B, H, T, D), dtype=np.float32).astype(np.float16)
        q, sc = C.quantize_tokens(x, kind)
        chk.update({f"x_{name}": x, f"q_{name}": q, f"s_{name}": sc, f"out_{name}": C.dequantize_tokens(q, sc, kind, D, "f16")})
    xe = chk["x_k"][0]
    chk.update(evict_x=xe, evict_pool=O.chunk_summarize_kv(xe, 64, 256), evict_window=O.trim_kv_sliding_window(xe, 256))
    cb = bench.cpu_baseline(L=1, B=B, H=H, T=T, D=D, sample_layers=1, reps=2, gpu_check=chk)
    ops = [k for k in cb["parity"] if k != "sample"]
    assert sorted(ops) == ["chunk_summarize_kv", "dequantise_int4", "dequantise_int8", "quantise_int4", "quantise_int8", "trim_kv_sliding_window"]
    assert all(cb["parity"][k]["bit_exact"] and cb["parity"][k]["max_rel_err"] == 0.0 for k in ops) and cb["max_rel_err_vs_gpu"] == 0.0
    chk["out_v"] = chk["out_v"].copy()
    chk["out_v"].view(np.uint16)[0, 0, 0, 3, 5] ^= 1
    cb = bench.cpu_baseline(L=1, B=B, H=H, T=T, D=D, sample_layers=1, reps=2, gpu_check=chk)
    assert not cb["parity"]["dequantise_int4"]["bit_exact"] and cb["parity"]["dequantise_int4"]["max_rel_err"] > 0


def test_bench_refuses_without_gpu():
    if torch.cuda.is_available():
        return
    proc = _bench("--steps", "1")
    assert proc.returncode != 0 and "no CPU path" in (proc.stderr + proc.stdout)


def test_gpus_flag_launches_ranks():
    """--gpus 2 with no launcher around it: the script starts its own two ranks and relays rank 0's line"""
    proc = _bench("--gpus", "2", "--launcher-selftest")
    assert proc.returncode == 0, proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["max_rank"] == 1.0 and res["rank_sum"] == 1.0 and res["timing_reduction_backend"] == "gloo"


def test_gpus_flag_fails_loudly():
    if not torch.cuda.is_available():  # more ranks than GPUs without --share-gpu: refused before anything starts
        proc = _bench("--gpus", "2")
        assert proc.returncode != 0 and "GPU" in proc.stderr and not proc.stdout.strip()
    # a rank that dies takes the job down with a non-zero exit code and no result line
    proc = _bench("--gpus", "2", "--launcher-selftest", env={"KVQ_SELFTEST_FAIL_RANK": "1"})
    assert proc.returncode != 0 and "rank 1" in proc.stderr and "{" not in proc.stdout
    # a rank whose WORKLOAD raises after the rendezvous leaves at once (no 300 s wait in shutdown's barrier behind its
    # peers' collective): the launcher reports it within seconds
    import time
    t0 = time.monotonic()
    proc = _bench("--gpus", "2", "--launcher-selftest", env={"KVQ_SELFTEST_RAISE_RANK": "1"}, timeout=120)
    assert proc.returncode != 0 and "rank 1" in proc.stderr and "workload failure" in proc.stderr and "{" not in proc.stdout
    assert time.monotonic() - t0 < 90
    # a rank count that disagrees with the launcher's WORLD_SIZE is refused
    proc = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--launcher-selftest"], cwd=ROOT, capture_output=True,
                          text=True, env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert proc.returncode != 0 and "WORLD_SIZE" in proc.stderr


@pytest.mark.benchcli
def test_bench_line_carries_the_contract_fields(tmp_path):
    """One real run of the default workload on the GPU (few steps, CPU baseline on a tiny sample): ONE JSON line on
    stdout with every field the driver's contract names, the roofline and cpu_baseline objects, self-consistent."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                           "--cpu-sample-layers", "1", "--cpu-reps", "1", "--cpu-sample-tokens", "2048"], capture_output=True, text=True, timeout=900, cwd=root)
    assert proc.returncode == 0, proc.stderr[-2000:]
    lines = [ln for ln in proc.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    print("bench phases_s:", j.get("phases_s"), "run_s:", j.get("run_s"))
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "decode", "configs", "public_api"):
        assert key in j, key
    assert j["n_gpus"] == 1 and j["steps"] == 4 and j["warmup"] == 2 and j["higher_is_better"] is True and j["vs_baseline"] is None
    assert j["unit"] == "GB/s" and j["scaling"] == "weak" and j["data"] == "synthetic" and j["config"]["workload"] == "llama3_8b_mixed_seq16k"
    r = j["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # value = bytes per step / step time; the INT4 launch is part of the step
    assert abs(j["value"] - j["config"]["bytes_per_step"] / (j["ms_per_step"] * 1e-3) / 1e9) / j["value"] < 2e-3
    assert r["avg_launch_ms"] < j["ms_per_step"] and 0.5 < r["frac"] < 1.0
    # every `kernel` label is what the library launched (kvq_kernel_log), i.e. what a profiler prints
    assert r["kernel"] == "dequant_tokens_fast_k<0, 4, 8, 4, true, true, false, 64>", r["kernel"]
    assert j["roofline_k"]["kernel"] == "dequant_tokens_fast_k<0, 8, 8, 2, true, true, true, 64>"
    assert j["roofline_quantise"]["quant_int4"]["kernel"] == "quant_tile_k<0, 4, 8, 16, 4, 0>"
    assert j["roofline_quantise"]["quant_int8"]["kernel"] == "quant_tile_k<0, 8, 8, 16, 4, 0>"
    assert j["public_api"]["ms_per_call"] >= j["ms_per_step"] * 0.75 and j["public_api"]["call"].startswith("QuantizedKVCache.to_past_key_values")
    # the rest of BASELINE.json's metric rides in the same line
    d = j["decode"]
    assert "error" not in d, d
    assert d["workload"] == "decode:gpt2:quant_int8:512:512" and d["tokens_per_sec"] > 0 and d["full_cache_tokens_per_sec"] > 0
    assert d["total_new_tokens"] == 512 and 17.0 < d["est_kv_cache_mb"] < 19.0 and abs(d["full_cache_kv_mb"] - 36.0) < 1e-6
    assert d["graph_decode_tokens_per_sec"] > 0 and d["fused_attention_tokens_per_sec"] > 0
    c3 = j["configs"]["gpt2m_int4_seq4k"]
    assert "error" not in c3, c3
    assert c3["shape_LBHTD"] == [24, 1, 16, 4096, 64] and c3["rotating_sets"] >= 4
    assert c3["roofline"]["kernel"].startswith("dequant_tokens_fast_k<0, 4,") and c3["roofline_quantise"]["kernel"] == "quant_tile_k<0, 4, 16, 8, 8, 0>"
    assert 0.3 < c3["roofline"]["frac"] < 1.0 and 0.3 < c3["roofline_quantise"]["frac"] < 1.0
    g2 = j["configs"]["gpt2_shape_seq32k"]
    assert "error" not in g2 and g2["quant_int8"]["kernel"] == "quant_tile_k<0, 8, 12, 8, 8, 0>", g2
    ev = j["configs"]["llama3_8b_evict_seq32k"]
    assert "error" not in ev, ev
    assert ev["roofline"]["kernel"].startswith("chunk_pool_wave_k<0, 4, 16>") and ev["roofline_window"]["kernel"].startswith("copy_rows_k<")
    assert 0.5 < ev["roofline"]["frac"] < 1.0 and 0.3 < ev["roofline_window"]["frac"] < 1.0
    sp = j["configs"]["llama3_8b_sparse_seq32k"]  # scope row N3 on the same tensors (round 4)
    assert "error" not in sp, sp
    for op in ("trim_kv_strided", "trim_kv_block_old", "trim_kv_budget_old", "trim_kv_prefix_window"):
        assert sp[op]["kernel"].startswith("gather_rows_k") and sp[op]["kept_tokens"] > 0 and 0.0 < sp[op]["frac"] < 1.0, (op, sp[op])
    assert sp["paged_get_kv"]["kernel"].startswith("copy_rows_k<") and sp["trim_kv_strided"]["kept_tokens"] == 32 + (32768 - 256 - 32 + 3) // 4 + 256
    for kind in ("int8", "int4"):  # the quantise launch on padded rows and on the outlier-channel input
        rq = j["roofline_quantise"][f"quant_{kind}"]
        assert rq["padded_rows"]["kernel"] == rq["kernel"] == rq["outlier_channels"]["kernel"] and 0.3 < rq["padded_rows"]["frac"] < 1.0
    sq = j["sharded_quant"]  # one rank: the batch-64 slice in ONE pass, the two phases of an N > 1 rank beside it
    assert "error" not in sq, sq
    assert sq["kernels"] == "quant_wide_k<0, 8, 1024, 16> + quant_wide_k<0, 4, 1024, 16>", sq["kernels"]
    assert sq["two_phase"]["kernels"].startswith("quant_tile_k<0, 8, 8, 16, 4, 1> + quant_tile_k<0, 8, 8, 16, 4, 2>"), sq["two_phase"]
    assert sq["value"] > sq["two_phase"]["value"] > 0 and sq["config"]["collective_backend"] == "none (1 rank)"
    c = j["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert c["host_cpu"]["usable"] == c["host_cores_available"] <= c["host_cpu"]["machine"] and c["timer_quantum_suspects"] == [], c["timer_quantum_suspects"]
    if c["host_cores_available"] > 1:
        assert c["port"]["all_cores"]["cores"] == c["host_cores_available"] and c["port"]["all_cores"]["value"] > 0
    assert c["max_rel_err_vs_gpu"] == 0.0 and c["eviction"]["pool_value"] > 0
    assert all(v["bit_exact"] and v["max_rel_err"] == 0.0 for k, v in c["parity"].items() if k != "sample"), c["parity"]
    assert j["run_s"] < 200
    ds = j["device_state"]  # clocks / power / temperature read while the headline step was queued: never fatal, always present
    assert isinstance(ds["gpu_still_busy_when_read"], bool) and ("mclk_mhz" in ds or "rocm_smi_error" in ds), ds


def test_usable_cores_reads_the_cgroup_quota(monkeypatch):
    """oracle/hostcpu.py (the cpu_baseline leg's pool size): a CFS quota of 16 cores on a 256-core machine — cgroup v2's
    `cpu.max` = "1600000 100000", what the GPU box reports — caps the usable cores at 16; "max" and v1's -1 mean no quota; the
    throttle counters come from cpu.stat. Files are faked: the container this suite runs in has no quota."""
    from oracle import hostcpu
    files = {"/proc/self/cgroup": "0::/process_api/abc\n", "/sys/fs/cgroup/cpu.max": "1600000 100000",
             "/sys/fs/cgroup/cpu.stat": "usage_usec 5\nnr_periods 100\nnr_throttled 7\nthrottled_usec 1234\n"}
    monkeypatch.setattr(hostcpu, "_read", lambda p: files.get(p))
    monkeypatch.setattr(hostcpu.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(hostcpu.os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    monkeypatch.delenv("KVQ_CPU_BASELINE_THREADS", raising=False)
    u = hostcpu.usable_cores()
    assert u == {"machine": 256, "affinity": 256, "quota": 16.0, "usable": 16, "override": None}
    assert hostcpu.throttle_counters() == (100, 7, 1234)
    files["/sys/fs/cgroup/cpu.max"] = "max 100000"
    assert hostcpu.usable_cores()["usable"] == 256 and hostcpu.quota_cores() is None
    # cgroup v1: quota / period files of the cpu controller; -1 = unlimited; a nested, tighter group wins
    files.clear()
    files.update({"/proc/self/cgroup": "4:cpu,cpuacct:/docker/xyz\n", "/sys/fs/cgroup/cpu/docker/xyz/cpu.cfs_quota_us": "250000",
                  "/sys/fs/cgroup/cpu/docker/xyz/cpu.cfs_period_us": "100000", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1",
                  "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000"})
    assert hostcpu.quota_cores() == 2.5 and hostcpu.usable_cores()["usable"] == 2
    monkeypatch.setenv("KVQ_CPU_BASELINE_THREADS", "5")
    assert hostcpu.usable_cores()["usable"] == 5
