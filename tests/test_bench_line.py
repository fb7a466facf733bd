"""bench.py's LAST stdout line must fit the driver's 2 000-character tail (round 3's 21.5 KB line was never parsed)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _full_record():
    # a real full record of a default run (every leg, cpu_baseline): the round-3 line as it was printed then
    with open(os.path.join(ROOT, "profiles", "r03_bench_full.json")) as f:
        return json.load(f)


def test_compact_line_fits_and_keeps_the_contract():
    import bench
    full = _full_record()
    assert len(json.dumps(full)) > 10000
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.COMPACT_LIMIT < 2000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("Sycamore n30 m14")
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert abs(line["value"] - full["value"]) / full["value"] < 1e-5
    assert set(line["workloads"]) == set(full["workloads"])
    assert json.loads(text) == line


def test_compact_line_degrades_instead_of_overflowing():
    import bench
    full = _full_record()
    full["workloads"] = {f"leg_number_{i}_with_a_long_name": v for i in range(6) for v in full["workloads"].values()}
    full["config"]["failed_workloads"] = list(full["workloads"])
    assert len(json.dumps(bench.compact_line(full))) <= 2000


def test_sig():
    import bench
    assert bench.sig(98.62362187104885, 6) == 98.6236
    assert bench.sig(5370515172608.006, 6) == 5.37052e12
    assert bench.sig(20.0) == 20 and isinstance(bench.sig(20.0), int)
    assert bench.sig(None) is None and bench.sig("ok") == "ok" and bench.sig(True) is True


def test_gpus_n_without_enough_devices_is_a_clean_error(capsys):
    """`python bench.py --gpus 8` as a plain command on a box without 8 GPUs: one JSON error line, exit code 2,
    no rank started (the parent must not touch the GPU either: torch.cuda.device_count() does not initialise it)."""
    import bench
    rc = bench.launch_ranks(4096)
    out = capsys.readouterr().out.strip().splitlines()
    assert rc == 2 and json.loads(out[-1])["n_gpus"] == 4096
