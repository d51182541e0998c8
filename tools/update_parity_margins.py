#!/usr/bin/env python3
"""Folds the parity margins a GPU run measured (gpurun_out/parity_margins.json, written by tests/conftest.py) into
the TRACKED profiles/parity_margins.json that tests/margins.py reads back: per mode, test and gradient tensor the
worst err/tol with the parts of the allowance at that element, the worst ratio of every candidate constant set, and the
pixel figures.  Run it after a green `pytest -m gpu` whenever a kernel change moves the margins on purpose:

    python tools/update_parity_margins.py [--from gpurun_out/parity_margins.json] [--note "why"]
"""
import argparse
import json
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--from", dest="src", default=os.path.join(ROOT, "gpurun_out", "parity_margins.json"))
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    with open(a.src) as f:
        modes = json.load(f)
    try:
        head = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        head = "unknown"
    out = {
        "_note": "Measured by pytest -m gpu on one MI355X; read back by tests/margins.py (no recorded worst err/tol may "
                 "grow past 2x + 0.02). Regenerate with tools/update_parity_margins.py.",
        "recorded_after_commit": head,
        "recorded_at": time.strftime("%Y-%m-%d %H:%M:%S UTC", time.gmtime()),
        "note": a.note,
        "modes": modes,
    }
    dst = os.path.join(ROOT, "profiles", "parity_margins.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    n = sum(len(v) for v in modes.values())
    print(f"{dst}: {n} tests in {len(modes)} modes")


if __name__ == "__main__":
    main()
