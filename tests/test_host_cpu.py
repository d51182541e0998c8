"""CPU-side checks: the C-ABI library loads and exports every symbol of include/brush_hip.h,
argument validation works without a GPU, host camera/uniform packing mirrors camera.rs."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as G

    if not os.path.exists(os.path.join(ROOT, "brush_amd", "lib", "libbrush_hip.so")):
        G.build()
    from brush_amd import _lib

    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "brush_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(brush_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 12
    from brush_amd import _lib

    assert sorted(_lib.SYMBOL_NAMES) == declared
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.brush_version()


def test_integration_doc_matches_header():
    """INTEGRATION.md's Rust FFI block is generated from include/brush_hip.h: regenerate and compare
    (symbol set, arity and types), and hold the ctypes binding to the same arity per symbol."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    hdr = open(os.path.join(ROOT, "include", "brush_hip.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    block = doc[doc.index(gen.BEGIN) + len(gen.BEGIN):doc.index(gen.END)].strip()
    assert block == gen.generate(hdr).strip(), "run `python tools/gen_rust_ffi.py --write`"
    # no second, hand-written extern block that could drift
    assert doc.count('extern "C" {') == 1
    structs, funcs = gen.parse_header(hdr)
    from brush_amd import _lib

    bound = {name: (restype, argtypes) for name, restype, argtypes in _lib._SYMBOLS}
    assert sorted(bound) == sorted(f[0] for f in funcs)
    for name, ret, params in funcs:
        restype, argtypes = bound[name]
        assert len(argtypes) == len(params), name
        assert (restype is None) == (ret == "void"), name
        for (pname, rtype), ct in zip(params, argtypes):
            is_ptr = rtype.startswith("*")
            ct_ptr = ct is C.c_void_p or ct is C.c_char_p or hasattr(ct, "contents")
            assert is_ptr == bool(ct_ptr), f"{name}.{pname}: {rtype} vs {ct}"
            if not is_ptr:
                want = {"u32": C.c_uint32, "i32": C.c_int, "f32": C.c_float, "usize": C.c_size_t}[rtype]
                assert ct is want, f"{name}.{pname}: {rtype} vs {ct}"
    for sname, fields in structs:
        cs = getattr(_lib, sname)
        assert [f[0] for f in cs._fields_] == [f[0] for f in fields], sname


def test_argument_validation_without_gpu(lib):
    from brush_amd import _lib

    n = C.c_size_t()
    assert lib.brush_fwd_workspace_size(1 << 20, 1920, 1080, 3, 8388480, C.byref(n)) == 0 and n.value > 0
    assert lib.brush_fwd_workspace_size(10, 32, 32, 5, 100, C.byref(n)) == -1  # sh_degree > 4
    assert lib.brush_bwd_workspace_size(1 << 20, 1920, 1080, 3, C.byref(n)) == 0 and n.value >= (1 << 20) * 36
    assert lib.brush_radix_argsort_workspace_size(1000, C.byref(n)) == 0 and n.value >= 8000
    assert lib.brush_inclusive_scan_workspace_size(1000, C.byref(n)) == 0 and n.value > 0
    # bits > 32 is rejected before any device work (brush-sort/src/lib.rs:38-39)
    assert lib.brush_radix_argsort_u32(None, None, None, None, None, 16, 33, None, 0, None) == -1
    assert lib.brush_radix_argsort_u32(None, None, None, None, None, 16, 32, None, 0, None) == -1  # null ptrs
    assert lib.brush_render_forward(None, None, None, None, None, None, 0, 0, None, None, None, 0, None) == -1
    assert lib.brush_status_string(-2) == b"workspace too small"
    # render.rs:204-206
    assert lib.brush_default_max_intersects(1 << 20, 1920, 1080) == 128 * 65535
    assert lib.brush_default_max_intersects(10, 32, 32) == 40
    assert _lib.BrushUniforms and C.sizeof(_lib.BrushUniforms) == 28 * 4


def test_camera_matches_reference_test_setup():
    """camera.rs:28-58 and the uniform packing of render.rs:102-116 vs the oracle's restatement."""
    import brush_amd
    from brush_amd.render import pack_uniforms
    from oracle import oracle as O

    w, h = 123, 82
    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    assert abs(focal - 61.5) < 1e-12
    cam = brush_amd.Camera([0.3, -0.2, -8.0], [0.1, 0.2, 0.3, math.sqrt(1 - 0.14)], brush_amd.focal_to_fov(focal, w),
                           brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
    u = pack_uniforms(cam, (w, h), 3, 10)
    o = O.make_uniforms(cam.position, cam.rotation, cam.fov_x, cam.fov_y, cam.center_uv, (w, h), 3)
    assert np.allclose(np.array(u.viewmat[:]), o["viewmat"], atol=1e-6)
    assert np.allclose(np.array(u.focal[:]), o["focal"]) and np.allclose(np.array(u.pixel_center[:]), o["pixel_center"])
    assert list(u.tile_bounds[:]) == [8, 6] and list(u.img_size[:]) == [w, h]
    # world_to_local really inverts local_to_world
    assert np.allclose(cam.world_to_local().astype(np.float64) @ cam.local_to_world(), np.eye(4), atol=1e-6)


def test_no_cpu_fallback():
    """The op refuses CPU tensors instead of silently computing elsewhere."""
    import torch

    import brush_amd

    cam = brush_amd.Camera([0, 0, -8], [0, 0, 0, 1], 1.0, 1.0)
    z = torch.zeros
    with pytest.raises(AssertionError, match="no CPU path"):
        brush_amd.render_splats(cam, (32, 32), z((4, 3)), None, z((4, 3)), z((4, 4)), z((4, 1, 3)), z((4,)))
    with pytest.raises(AssertionError):
        brush_amd.prefix_sum(torch.zeros(4, dtype=torch.int32))


def test_product_and_bench_do_not_import_the_oracle():
    """The oracle is test infrastructure: importing the package, or bench.py up to its timed path
    (synthetic inputs included), must not load it; only bench.py's cpu_baseline leg and the tests do."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, importlib.util; sys.argv=['bench.py']; import brush_amd; "
            "spec=importlib.util.spec_from_file_location('b','bench.py'); m=importlib.util.module_from_spec(spec); "
            "spec.loader.exec_module(m); m.synthetic_cloud(8, 1); "
            "print(sorted(k for k in sys.modules if k.split('.')[0] == 'oracle'))")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "[]"
    for dirpath, _, files in os.walk(os.path.join(root, "brush_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+tests\b", src, flags=re.M), f
    # the bench and the tools do not depend on the test package either
    for f in ["bench.py"] + [os.path.join("tools", t) for t in os.listdir(os.path.join(root, "tools")) if t.endswith(".py")]:
        assert not re.search(r"^\s*(from|import)\s+tests\b", open(os.path.join(root, f)).read(), flags=re.M), f
