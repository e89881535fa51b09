"""bench.py: the SURVEY 8(d) byte formula on CPU, and the one-JSON-line contract on the GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_algorithmic_bytes_formula():
    sys.path.insert(0, ROOT)
    import bench
    # smallCapture S = 92 -> B = 225 + 4928 e; tiny S = 76 -> 193 + 4480 e; 20x20 S = 128 -> 297 + 12800 e
    assert bench.algorithmic_bytes(11, 14, 4) == 225 + 4928 * 4 == 19937
    assert bench.algorithmic_bytes(11, 14, 1) == 225 + 4928
    assert bench.algorithmic_bytes(7, 20, 2) == 193 + 4480 * 2
    assert bench.algorithmic_bytes(20, 20, 4) == 297 + 12800 * 4
    assert bench.WORKLOADS["small16384"] == ("smallCapture", 16384)          # BASELINE configs[2], the default workload
    r = bench.reference_ratio("smallCapture")
    assert r is None or r > 10                                                # port / reference speed ratio, if the file is present


@pytest.mark.gpu
def test_bench_prints_one_json_line_with_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "30", "--warmup", "5", "--workload", "tiny4096",
                          "--no-ppo", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["steps"] == 30 and d["warmup"] == 5 and d["n_gpus"] == 1 and d["unit"] == "env-steps/s" and d["vs_baseline"] is None
    assert abs(d["value"] - 4096 / (d["ms_per_step"] / 1e3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes_per_launch"] == 4096 * 4 * 8 * 7 * 20 * 4
    # the headline figure is the all-bytes-to-HBM one (sweep direction fixed); the cache-assisted figure of the default path in
    # this loop is reported beside it and can only be faster
    assert "expand_alt 0" in r["measured_with"] and r["cache_assisted"]["GBps"] > 0 and "workload" in d["config"]
    assert d["tick_all_bytes_to_hbm"]["us_per_tick"] > 0
