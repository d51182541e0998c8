"""The parity gate must SEE a small loss of accuracy, not only a wrong formula.

libbrush_hip_inject.so is the same library with one deliberate defect in the compositing backward: every per-pixel
gradient term carries a one-signed error of 4 eps32 of its magnitude (BRUSH_INJECT_VVA_ULPS, brush_amd/csrc/Makefile).
That is 2.4e-7 relative per term, far inside the reference's own rtol of 1e-4 (render.rs:815-830) wherever an
element's terms add up, but on elements whose terms cancel it is the difference between "as accurate as the f32
restatement" and "several times worse".  The oracle-parity tests run against that build in a child interpreter and
must FAIL, through the tracked margins (tests/margins.py, profiles/parity_margins.json) or the allowance itself."""
import json
import os
import subprocess
import sys

import pytest

from tests import margins as M

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INJECT = os.path.join(ROOT, "brush_amd", "csrc", "build", "libbrush_hip_inject.so")


@pytest.mark.timeout(900)
def test_injected_4eps_error_turns_the_gate_red():
    assert os.path.exists(INJECT), "build the test twin with make -C brush_amd/csrc (target all)"
    with open(M.TRACKED) as f:
        tracked = json.load(f)
    assert tracked.get("modes", {}).get("default"), "profiles/parity_margins.json holds no tracked margins"
    env = dict(os.environ, BRUSH_HIP_LIB=INJECT, BRUSH_MARGINS_NO_FLUSH="1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("BRUSH_DETERMINISTIC", None)
    sel = "matches_oracle or headline_size_matches or s4_sizes"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_render.py", "-q", "-m", "gpu", "-k", sel,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    out = r.stdout[-6000:] + r.stderr[-2000:]
    assert r.returncode != 0, "the gate stayed green on a build with 4 eps32 of injected error per term:\n" + out
    assert "grew past" in r.stdout or "elements over" in r.stdout, out
    failed = r.stdout.count("FAILED")
    print(f"injected build: {failed} parity tests failed (expected: most of them)")
    assert failed >= 3, out
