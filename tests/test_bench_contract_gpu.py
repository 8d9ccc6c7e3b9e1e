"""bench.py prints ONE JSON line that carries the benchmark's contract: the headline metric with its
configuration, the roofline object of the dominant kernel, the CPU baseline timed beside it and the
parity of the same frames.  Run here on a reduced batch (the arithmetic and the schedule are those of the
full run) as a child process, the way the driver runs it."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_line_contract():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--frames", "1024",
           "--cpu-frames", "128", "--no-hamming", "--no-extra"]
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=110)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, "exactly one line on stdout"
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "u8" and "synthetic" in d["data"]
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = frames of the whole job / the timed region
    assert abs(d["value"] - 1024 * 2 / (d["ms_per_step"] * 2 / 1000.0)) <= 0.01 * d["value"]
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3 and 0 < roof["frac"] < 1
    assert roof["traffic"] is None or roof["traffic"] > 0
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1 and cpu["value"] > 0
    # the same frames, CPU checker against the device: no differing keypoint or descriptor
    assert d["parity"]["frames"] == 128 and d["parity"]["mismatched_keypoints"] == 0 and d["parity"]["mismatched_descriptors"] == 0
