"""PLY import/export: round trip, encodings, property rules of splat_import.rs."""
import numpy as np
import pytest

from brush_amd import ply as P


def _cloud(n=37, c=16, seed=0):
    rng = np.random.default_rng(seed)
    return dict(means=rng.normal(size=(n, 3)).astype(np.float32), log_scales=rng.normal(size=(n, 3)).astype(np.float32),
                rotation=rng.normal(size=(n, 4)).astype(np.float32), raw_opacity=rng.normal(size=n).astype(np.float32),
                sh_coeffs=rng.normal(size=(n, c, 3)).astype(np.float32))


@pytest.mark.parametrize("c", [1, 4, 16])
def test_round_trip_bit_exact(c):
    d = _cloud(c=c)
    blob = P.splat_to_ply(**d)
    assert blob.startswith(b"ply\nformat binary_little_endian 1.0\ncomment Exported from Brush\n")
    back = P.load_splat_from_ply(blob)
    for k in d:
        assert np.array_equal(back[k], d[k]), k


def test_inria_rest_layout_and_truncation():
    """f_rest is channel-major ([R.., G.., B..]); more than degree 3 is truncated (splat_import.rs:241-246)."""
    d = _cloud(n=5, c=25)
    blob = P.splat_to_ply(**d)
    hdr = blob[: blob.find(b"end_header")].decode()
    assert "property float f_rest_71" in hdr and "f_rest_72" not in hdr
    # first rest property of the first vertex is the RED channel of coefficient 1
    off = blob.find(b"end_header\n") + len("end_header\n")
    row = np.frombuffer(blob, "<f4", count=14 + 72, offset=off)
    assert row[14] == d["sh_coeffs"][0, 1, 0] and row[14 + 24] == d["sh_coeffs"][0, 1, 1]
    back = P.load_splat_from_ply(blob)
    assert back["sh_coeffs"].shape == (5, 16, 3)
    assert np.array_equal(back["sh_coeffs"], d["sh_coeffs"][:, :16])


def test_ascii_big_endian_and_extra_properties():
    d = _cloud(n=4, c=1)
    names = ["x", "y", "z", "nx", "scale_0", "scale_1", "scale_2", "opacity", "rot_0", "rot_1", "rot_2", "rot_3",
             "f_dc_0", "f_dc_1", "f_dc_2"]
    cols = np.concatenate([d["means"], np.zeros((4, 1), np.float32), d["log_scales"], d["raw_opacity"][:, None],
                           d["rotation"], d["sh_coeffs"][:, 0, :]], axis=1)
    hdr = "ply\nformat {} 1.0\nelement vertex 4\n" + "".join(f"property float {n}\n" for n in names) + "end_header\n"
    ascii_blob = hdr.format("ascii").encode() + "".join(" ".join(repr(float(v)) for v in r) + "\n" for r in cols).encode()
    be_blob = hdr.format("binary_big_endian").encode() + cols.astype(">f4").tobytes()
    for blob in (ascii_blob, be_blob):
        back = P.load_splat_from_ply(blob)
        assert np.allclose(back["means"], d["means"]) and np.allclose(back["rotation"], d["rotation"])
        assert back["sh_coeffs"].shape == (4, 1, 3)


def test_missing_property_is_an_error():
    hdr = b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nend_header\n"
    with pytest.raises(ValueError, match="Missing properties"):
        P.load_splat_from_ply(hdr + np.zeros(2, "<f4").tobytes())
    with pytest.raises(ValueError):
        P.load_splat_from_ply(b"not a ply")


def test_from_ply_normalises_rotations():
    """Every import ends with Splats::norm_rotations (splat_import.rs:140,150; gaussian_splats.rs:202):
    non-unit rot_* come back as unit quaternions with the same direction, and re-export keeps them unit."""
    import torch

    import brush_amd

    d = _cloud(n=7, c=4)
    d["rotation"] *= np.float32(3.7)
    blob = P.splat_to_ply(d["means"], d["log_scales"], d["rotation"], d["raw_opacity"], d["sh_coeffs"])
    raw = P.load_splat_from_ply(blob)
    assert np.allclose(raw["rotation"], d["rotation"])  # the low-level reader returns the file's values
    splats = brush_amd.Splats.from_ply(blob, torch.device("cpu"))
    rot = splats.rotation.detach().numpy()
    assert np.allclose(np.linalg.norm(rot, axis=1), 1.0, atol=1e-6)
    want = d["rotation"] / np.linalg.norm(d["rotation"], axis=1, keepdims=True)
    assert np.allclose(rot, want, atol=1e-6)
    again = P.load_splat_from_ply(splats.to_ply())
    assert np.allclose(np.linalg.norm(again["rotation"], axis=1), 1.0, atol=1e-6)
