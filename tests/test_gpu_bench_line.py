"""The driver's contract for `python bench.py` (no flags but a short CPU budget): ONE JSON line on stdout with the headline
metric of BASELINE.json on its configuration, the `roofline` and `cpu_baseline` objects, and under `extra` the other
configurations.  Run as the driver runs it: a child process, N = 1."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_bench_line_has_what_the_driver_reads():
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--cpu-seconds", "2"],
                       capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "extra", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "reports/s"      # BASELINE's metric (which adds "at 1/2/4/8 GPUs")
    assert d["vs_baseline"] is None and d["data"].startswith("synthetic")
    cfg = d["config"]
    assert cfg["reports_per_gpu"] == 125000 and cfg["status_nonzero"] == 0 and cfg["sort_and_keep_mask_in_step"] is True
    assert "vec_size 1024" in cfg["workload"] or "1024" in cfg["workload"]
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 125000) < 1            # whole-job throughput of exactly the timed steps
    roof = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in roof, k
    assert roof["bound"] in ("hbm", "mfma", "valu") and 0.2 < roof["frac"] < 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    cpu = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cpu, k
    assert cpu["kind"] in ("port", "reference") and cpu["value"] > 0 and cpu["cores"] >= 1
    assert sum(v["ms_per_step"] for k, v in d["kernels"].items() if k not in ("expand", "sort")) <= d["ms_per_step"] * 1.02   # (side streams overlap)
    # the other configurations ride along
    assert set(d["extra"]) == {"sumvec_cfg1", "histogram", "average", "hh_t0.01", "hh_t0.001"}
    for name, e in d["extra"].items():
        assert e["value"] > 0 and e["ms_per_step"] > 0 and e["status_nonzero"] == 0, name
        assert 0.2 < e["roofline"]["frac"] < 1.0, name
