"""The parts of __graft_entry__.smoke() that can be checked without a GPU: its forced-MFMA step
really is planned onto the tiled kernel (a silent move to the strided kernel would hollow the
smoke test out)."""
import re

import artensor_amd as A


def test_smoke_step_is_planned_on_the_mfma_kernel():
    import os
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "__graft_entry__.py")).read()
    eq = re.search(r'eq = "([A-Za-z,>-]+)"', src).group(1)
    lhs = eq.split("->")[0].split(",")
    info = A.step_info(eq, (2,) * len(lhs[0]), (2,) * len(lhs[1]))
    assert info["kernel"] == 1 and info["n_tiles"] >= 32
