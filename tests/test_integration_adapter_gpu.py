"""GPU: INTEGRATION.md section B on a real device.  integration/_build/adapter_check is
integration/adapter_check.cpp compiled in the development container against the REFERENCE's own
src/multivariate/multivariate.h:132-146 (`make -C integration`, also part of
__graft_entry__.build()); it drives a HipOptimizer through a `MultivariateOptimizer*` exactly as
the reference drives any of its optimizers: optimize() on the README example (Rosenbrock, n = 10,
box [-10, 10]) and the stepwise init / iterate / solution.  The binary travels to the GPU box with
the snapshot (git-ignored, like oracle/_ref); the reference itself does not and is not needed."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "integration", "_build", "adapter_check")


def test_adapter_optimizes_through_the_reference_interface(hip):
    if not os.path.exists(EXE):
        pytest.skip("integration/_build/adapter_check was not built (no /root/reference where "
                    "this snapshot was made)")
    run = subprocess.run([EXE], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, run.stdout + run.stderr
    out = run.stdout
    # multivariate_solution::toString() of the reference (multivariate.h:97-114)
    assert "converged: yes" in out, out
    m = re.search(r"objective calls: (\d+)", out)
    assert m and 0 < int(m.group(1)) <= 20000, out
    xs = re.search(r"x\*: ([-+0-9.e ]+)", out)
    assert xs, out
    x = [float(v) for v in xs.group(1).split()]
    assert len(x) == 10 and max(abs(v - 1.) for v in x) < 1e-2, x
    # the stepwise interface: 5 generations of lambda = 20 after init
    assert "after 5 generations: 100 evaluations" in out, out
