"""bench.py contract (driver reads this JSON line) on the tiny BASELINE config 1, and __graft_entry__.smoke()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_emits_one_contract_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c1", "--steps", "3", "--warmup", "1",
                          "--cpu-sample", "2", "--cpu-threads", "4"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["metric"] == "utterances/sec" and j["n_gpus"] == 1 and j["steps"] == 3 and j["scaling"] == "weak"
    assert j["dtype"] == "f32" and j["vs_baseline"] is None and j["value"] > 0
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])
    assert abs(j["roofline"]["frac"] - j["roofline"]["achieved"] / j["roofline"]["peak"]) < 1e-3
    assert j["roofline"]["launches"] > 0 and j["roofline"]["profiled_steps"].startswith("1 of the 3 timed steps")
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 4 and j["cpu_baseline"]["value"] > 0
    assert j["loss_rel_delta"] < 1e-4          # north_star tolerance on the whole path (mel, label) -> loss
    assert "workload" in j["config"] and "model" not in j["config"]
    assert set(j["grad_max_abs_dev"]) == {"fc.weight", "encoder.rnn.weight_hh_l0", "decoder.embedding.weight"}
    for v in j["grad_max_abs_dev"].values():   # vs the FLOAT64 oracle at the INITIAL weights: the fixture tests' bound, no widening
        assert v["max_abs_dev"] <= 2e-4 * max(v["ref_max_abs"], 1e-3)
    assert j["parity"]["ok"] is True and j["parity"]["all_rows_finite"] is True and "INITIAL weights" in j["parity"]["oracle"]
    assert j["parity_after_training"]["informational"] is True      # reported next to the fp32 oracle's own distance, never a gate
    assert "fp32_oracle_max_abs_dev" in j["parity_after_training"]["grad_max_abs_dev"]["fc.weight"]
    assert j["rccl_ranks"] == 1
    assert j["cpu_baseline"]["cpu_model"] and j["cpu_baseline"]["host_logical_cpus"] >= 4
    assert "1 warm-up" in j["cpu_baseline"]["sample"] and "3 timed" in j["cpu_baseline"]["sample"]
    assert j["memory"]["peak_allocated_bytes"] > 0


def test_smoke_entry_point():
    sys.path.insert(0, ROOT)
    import __graft_entry__
    __graft_entry__.smoke()
