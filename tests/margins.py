"""Parity margins: what the GPU parity tests measured, kept next to the gate so that a later loosening shows.

Every oracle-parity test records, per gradient tensor, its worst err/tol and the parts of the tolerance at that
element; per test the pixel figures.  A GPU run writes them to gpurun_out/parity_margins.json;
tools/update_parity_margins.py folds those into the TRACKED profiles/parity_margins.json, and from then on the
tests assert that no recorded ratio more than doubles (GROWTH) — the gate that turns a kernel change which makes every
gradient several times less accurate red although it still fits the mechanism-based allowance.
"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRACKED = os.path.join(ROOT, "profiles", "parity_margins.json")
GROWTH = 2.0      # a recorded worst ratio may at most double ...
ABS_SLACK = 0.02  # ... plus this much (float atomics: the sums differ run to run in the last bits)

_recorded = {}
_tracked = None


def mode():
    from brush_amd import render as R

    return "deterministic" if R.deterministic_default() else "default"


def test_id():
    return os.environ.get("PYTEST_CURRENT_TEST", "?").split(" (")[0]


def record(section, name, value, tid=None):
    _recorded.setdefault(mode(), {}).setdefault(tid or test_id(), {}).setdefault(section, {})[name] = value


def tracked(section, name, tid=None):
    """The tracked value for the current test (None when the test was never recorded)."""
    global _tracked
    if _tracked is None:
        try:
            with open(TRACKED) as f:
                _tracked = json.load(f)
        except (OSError, ValueError):
            _tracked = {}
    return _tracked.get("modes", {}).get(mode(), {}).get(tid or test_id(), {}).get(section, {}).get(name)


def check_growth(section, name, value, tid=None):
    """Assert `value` (a worst err/tol ratio) has not grown past GROWTH x the tracked one."""
    old = tracked(section, name, tid)
    if old is None or os.environ.get("BRUSH_MARGINS_REBASE"):
        # REBASE: the run that re-records the margins after the GATE itself changed (new allowance terms or constants);
        # the allowance is still asserted, only the comparison with the old gate's ratios is skipped.  Never set by the
        # driver's runs; tools/update_parity_margins.py --note says why a rebase was made.
        return
    old = float(old["worst"] if isinstance(old, dict) else old)
    limit = GROWTH * old + ABS_SLACK
    assert value <= limit, (f"{tid or test_id()} {section}.{name}: worst err/tol {value:.4f} grew past {GROWTH} x the tracked "
                            f"{old:.4f} (+{ABS_SLACK}) of profiles/parity_margins.json")


def flush():
    if not _recorded or os.environ.get("BRUSH_MARGINS_NO_FLUSH"):
        return None
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "parity_margins.json")
    old = {}
    if os.path.exists(path):  # several pytest invocations of one GPU call accumulate
        try:
            with open(path) as f:
                old = json.load(f)
        except ValueError:
            old = {}
    for m, tests in _recorded.items():
        old.setdefault(m, {}).update(tests)
    with open(path, "w") as f:
        json.dump(old, f, indent=1, sort_keys=True)
    return path
