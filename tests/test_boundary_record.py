"""The drop-in boundary validated on LIVE reference objects: tests/golden/check_boundary.py imports the
reference next to this package in the build container (planner objects: ContractionTree,
TensorNetworkSimulation; entry points: the compilers, from_planned, the one-call API) and writes
tests/golden/boundary_check.json.  This test asserts that record on every box, and re-runs the check
itself when the reference is present and ARTN_CHECK_BOUNDARY=1 (about a minute of planning)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORD = os.path.join(ROOT, "tests", "golden", "boundary_check.json")

EXPECTED = [
    "contraction_scheme(live ContractionTree) == reference",
    "contraction_scheme_sparse(live tree, sc_target=30) == reference",
    "contraction_scheme_sparse(live tree, sc_target=8) == reference",
    "contraction_scheme_sparse(1 500 bitstrings, sc_target=12) == reference",
    "contraction_scheme_sparse(sliced live tree) == reference scheme held by the planned object",
    "from_planned(dense)", "from_planned(sparse)", "from_planned(sliced)",
    "signature of contraction_scheme", "signature of tensor_contraction", "signature of contraction_scheme_sparse",
    "signature of tensor_contraction_sparse", "signature of tensor_network_contraction",
    "signature of quantum_circuit_simulation",
    "TensorNetworkSimulation.prepare_contraction parameters", "TensorNetworkSimulation.contraction parameters",
    "from_circuit_file + prepare_contraction (dense) == reference plan",
    "from_circuit_file + prepare_contraction (sliced) == reference plan",
    "quantum_circuit_simulation(n12, 40 bitstrings) == reference",
    "quantum_circuit_simulation(n12, full amplitude) == reference",
]


def test_boundary_record_is_complete_and_green():
    rec = json.load(open(RECORD))
    assert rec["all_passed"] is True
    names = [c["name"] for c in rec["checks"]]
    assert all(c["passed"] for c in rec["checks"])
    for want in EXPECTED:
        assert any(want in n for n in names), want
    # the chunked sparse compile really produced chunked steps, the sliced plan really sliced
    by = {c["name"]: c for c in rec["checks"]}
    assert by["contraction_scheme_sparse(live tree, sc_target=8) == reference"]["chunked"] >= 1
    assert by["a sliced sparse plan exists"]["sliced_bonds"] >= 2
    assert by["contraction_scheme_sparse(1 500 bitstrings, sc_target=9) == reference"]["chunked"] >= 1
    assert by["contraction_scheme_sparse(1 500 bitstrings, sc_target=30) == reference"]["selects"] >= 1
    errs = [c["rel_err"] for c in rec["checks"] if "rel_err" in c]
    assert len(errs) >= 5 and max(errs) < 1e-5


def test_boundary_record_was_made_for_these_sources():
    """The record is evidence about contraction.py and simulation.py AS THEY ARE: it carries their hash, and a record
    made before the last edit of either file fails here (re-run tests/golden/check_boundary.py in the build container)."""
    from artensor_amd.simulation import boundary_source_sha16
    rec = json.load(open(RECORD))
    assert rec.get("source_sha16") == boundary_source_sha16(), "boundary_check.json is stale: re-run tests/golden/check_boundary.py"


@pytest.mark.skipif(not (os.path.isdir("/root/reference/artensor") and os.environ.get("ARTN_CHECK_BOUNDARY") == "1"),
                    reason="needs the reference (build container) and ARTN_CHECK_BOUNDARY=1")
def test_boundary_check_runs_against_the_live_reference(tmp_path):
    env = dict(os.environ, PYTHONHASHSEED="0", PYTHONDONTWRITEBYTECODE="1")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "golden", "check_boundary.py")], env=env)
