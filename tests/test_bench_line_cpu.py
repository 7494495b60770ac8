"""The bench contract line: <= 4 KB, strict JSON, contract fields + roofline + cpu_baseline, whatever the legs hold.

Canned input: tests/golden/bench_full_record_r4.json = the full 25 KB record round 4's bench produced (the one the
driver could not parse); the line builder must cut it down to a line the driver can."""
import copy
import io
import json
import os
import sys
from contextlib import redirect_stderr, redirect_stdout

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture()
def full():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r4.json")))


def strict(text):
    def no_constants(name):
        raise ValueError(f"non-strict JSON constant {name}")

    return json.loads(text, parse_constant=no_constants)


def test_contract_line_is_small_strict_and_complete(bench, full):
    assert len(json.dumps(full)) > 20000  # the record that broke the driver's parse
    text = bench.contract_line(full, "gpurun_out/bench_legs.json")
    assert "\n" not in text and len(text.encode()) < 4096
    line = strict(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-6) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-6)
    assert line["dtype"] == "f32" and line["config"]["workload"].startswith("cfg3") and "model" not in line["config"]
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-4)
    assert r["traffic"] == pytest.approx(full["roofline"]["traffic"], rel=1e-5) and r["kernel"] == "conv3x3_wino4_kernel"
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 16 and c["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-4) and c["parity_on_this_sample"]["peak_indices_equal"] is True
    s = line["legs_summary"]
    assert s["infer_cfg1"]["value"] == pytest.approx(full["infer_cfg1"]["value"], rel=1e-4) and s["published_workload"]["end_to_end_frames_per_s"] > 0
    assert line["h2d_inclusive"]["value"] == pytest.approx(full["h2d_inclusive"]["value"], rel=1e-6)
    for leg in bench.LEG_KEYS:  # no leg body in the line
        assert leg not in line


def test_contract_line_survives_nan_and_oversized_legs(bench, full):
    big = copy.deepcopy(full)
    big["roofline"]["traffic"] = float("nan")
    big["cpu_baseline"]["sample"] = "x" * 5000
    big["train_cfg4"]["value"] = float("inf")
    big["roofline"]["per_op_ms"] = {f"op{i}": 0.1 for i in range(2000)}
    for i in range(40):
        big[f"infer_extra_{i}"] = {"value": 1.0, "blob": "y" * 1000}
    text = bench.contract_line(big, None)
    assert len(text.encode()) < 4096
    line = strict(text)
    assert line["roofline"]["traffic"] is None and line["legs_summary"]["train_cfg4"]["value"] is None and len(line["cpu_baseline"]["sample"]) <= 260


def test_a_line_that_outgrows_the_bound_sheds_prose_before_the_per_leg_summary(bench):
    """Round 6's own record (tests/golden/bench_full_record_r6.json: every leg present, the line 3 999 bytes) with a shard check of eight ranks on top -- the
    N = 8 shape of the line: it must stay under the bound by dropping explanatory strings, and keep `roofline.sampling`, `roofline.traffic` and the per-leg numbers."""
    full = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_full_record_r6.json")))
    base = strict(bench.contract_line(copy.deepcopy(full), "gpurun_out/bench_legs.json"))
    assert base["roofline"]["sampling"] == "ok" and base["roofline"]["traffic"] > 0 and base["legs_summary"]["peaks_kernel"]["traffic"] > 0
    assert base["legs_summary"]["infer_cfg5"]["mfma_frac"] > 0.3 and "accounting" in base["roofline"]
    fat = copy.deepcopy(full)
    fat["config"]["frames_per_step_by_rank"] = [4] * 8
    fat["config"]["shard_check"] = {"result": "8 ranks x 3 lanes: head outputs and keypoints equal the single-rank recompute" + "; rank detail" * 20, "host_stage_wait_ms_per_step_by_rank": [0.123456] * 8}
    text = bench.contract_line(fat, "gpurun_out/bench_legs.json")
    assert len(text.encode()) <= 4096
    line = strict(text)
    assert "accounting" not in line["roofline"] and line["roofline"]["sampling"] == "ok"
    assert line["legs_summary"]["infer_cfg5"]["mfma_frac"] == base["legs_summary"]["infer_cfg5"]["mfma_frac"] and line["config"]["shard_check"].startswith("8 ranks")


def test_contract_line_refuses_to_exceed_the_bound(bench, full):
    bad = copy.deepcopy(full)
    bad["config"]["workload"] = "w" * 5000
    with pytest.raises(RuntimeError):
        bench.contract_line(bad, None)


def test_emit_prints_the_contract_line_last_and_writes_the_legs_file(bench, full, tmp_path):
    out, err = io.StringIO(), io.StringIO()
    legs = tmp_path / "legs.json"
    with redirect_stdout(out), redirect_stderr(err):
        bench.emit(full, str(legs))
    lines = out.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096 and strict(lines[0])["value"] == pytest.approx(full["value"], rel=1e-6)
    rec = strict(legs.read_text())
    assert set(bench.LEG_KEYS) - {"weak_scaling"} <= set(rec) and rec["infer_cfg5"]["value"] == full["infer_cfg5"]["value"]
    assert err.getvalue().startswith("bench.py full record: ")


def test_training_headline_goes_through_the_same_line(bench, full):
    leg = dict(full["train_cfg4"], n_gpus=1, warmup=2, higher_is_better=True, scaling="strong", vs_baseline=None)
    line = strict(bench.contract_line(leg, None))
    assert line["metric"].startswith("frames/sec training step") and line["roofline"]["frac"] == pytest.approx(full["train_cfg4"]["roofline"]["frac"], rel=1e-4)


def test_every_conv_kernel_family_the_library_reports_has_a_name_in_the_bench():
    """A per-rank batch of a multi-GPU run can route layers to kernel families the 32-frame forward never takes (round 5: conv3x3_sm_kernel at 2 - 3 frames per rank):
    the bench's per-kernel accounting must know every PH_KV_* code a 3x3 conv can come back with."""
    import bench
    from sleap_nn_amd import _lib as L

    short, long_ = bench.conv_kernel_short_names(), bench.conv_kernel_long_names()
    conv_codes = {v for k, v in vars(L).items() if k.startswith("KV_") and isinstance(v, int)} - {L.KV_NONE, L.KV_FUSED, L.KV_STEM, L.KV_MLP}  # (KV_MLP: a pair of Linear ops, never a 3x3 conv)
    assert L.KV_MLP in L.KV_NAMES and L.KV_MFMA_SHARE[L.KV_MLP] == 1.0
    assert conv_codes <= set(short), sorted(conv_codes - set(short))
    assert set(short.values()) <= set(long_) and set(L.KV_NAMES) >= conv_codes and set(L.KV_MFMA_SHARE) >= conv_codes


def test_per_op_sampling_takes_medians_never_the_first_steps_and_flags_an_inflated_stack(bench):
    """the driver's 20-step run left two event-profiled forwards, one of them step 0 behind a barrier on an idle GPU, and the line's roofline was
    the mean of the two (conv stack 11.40 ms beside a 10.83-ms step).  Now: >= 5 samples at --steps 20 and 50, none of them step 0 / 1; the per-op figure is the
    MEDIAN over the samples, so ONE inflated sample changes nothing; and a stack that still exceeds 1.03 x the median step is marked, not published as a roofline."""
    for steps in (20, 50, 200):
        at = bench.profile_steps(steps)
        assert len(at) >= 5 and min(at) >= 2 and max(at) < steps and len(set(at)) == len(at), (steps, at)
    assert bench.profile_steps(3) == [2] and bench.profile_steps(2) == []
    good = [1.00, 0.40, 0.40, 0.10]            # per-op ms of a forward: stem, two convs, a head
    samples = [[1.9, 0.55, 0.52, 0.30]] + [[g * (1 + 0.004 * k) for g in good] for k in range(5)]   # the first sample is cold
    s = bench.summarize_op_samples(samples, [0, 1, 2], step_ms_median=1.95)
    assert s["n"] == 6 and s["sampling"] == "ok"
    assert all(abs(m - g * 1.01) < 0.011 * g for m, g in zip(s["op_ms"], good)), s["op_ms"]      # the medians sit on the warm samples
    assert abs(s["stack_ms"] - 1.8 * 1.01) < 0.02 and s["stack_ms_by_sample"][0] > 2.9
    # the mean would have been pulled up by the cold sample -- and a stack above 1.03 x the step is flagged
    assert sum(r[0] for r in samples) / 6 > 1.14
    assert bench.summarize_op_samples(samples, [0, 1, 2], step_ms_median=1.70)["sampling"] == "inflated"
    assert bench.summarize_op_samples([], [0], 1.0)["sampling"] == "none"
