"""The tile walks classify every candidate tile with a cheap head and send only the undecided ones through the expensive
tail of the reference's ellipse / box test (brush_amd/csrc/splat_math.hpp: tile_test_head, make_tile_reach).  The head's
"miss" exit is the one place where the HIP path does not evaluate the reference's expression tree, so it is checked here
against the oracle's exact test (helpers.wgsl:220-262 restated in oracle/brush_oracle.c) on tens of millions of random
ellipse / tile pairs, extreme aspect ratios included: not one exit may contradict the exact test.  (The GPU suite then
compares every tile list bit for bit.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_conservative_miss_exit_never_contradicts_the_exact_test(tmp_path):
    exe = str(tmp_path / "tile_reach_check")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-mavx2", "-fopenmp",
                           "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "aux", "tile_reach_check.c"),
                           "-lm", "-o", exe], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe, "40000000"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout
    fields = dict(zip(r.stdout.split()[0::2], r.stdout.split()[1::2]))
    assert int(fields["BAD"]) == 0
    # the test has teeth: a good share of the cases takes each exit
    assert int(fields["miss_exit"]) > 1_000_000 and int(fields["hit_exit"]) > 1_000_000 and int(fields["edge"]) > 1_000_000


def test_head_constants_match_the_device_code():
    """The C check restates the head; keep its constants in step with splat_math.hpp."""
    dev = open(os.path.join(ROOT, "brush_amd", "csrc", "splat_math.hpp")).read()
    chk = open(os.path.join(ROOT, "tests", "aux", "tile_reach_check.c")).read()
    for token in ("1.001f", "1024.0f", "3.0e37f"):
        assert token in dev and token in chk, token
    assert "half + 0.02f" in dev and "8.02f" in chk
