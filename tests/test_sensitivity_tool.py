"""tools/oracle_sensitivity.py (the fp64 oracle against itself with a float32-rounded start state): the tool runs, an
unperturbed copy stays bit-identical, and a 6e-8 perturbation is visible but small over the first env steps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_oracle_sensitivity_tool_short_run():
    import oracle_sensitivity as osens
    q, o, ends, split = osens.run("f32_once", nsteps=12)
    assert q.shape == (16, 12) and all(s is None for s in split)
    assert 0 < q.max() < 1e-5 and np.isfinite(o).all()          # rounding the start state to float32 moves the trajectory, a little
    q0, *_ = osens.run("none", nsteps=4)                          # no perturbation mode: two identical copies
    assert q0.max() == 0.0
