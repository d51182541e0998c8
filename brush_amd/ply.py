"""Inria-layout splat PLY import / export — host-side numpy, not on the hot path.

Mirrors crates/brush-dataset/src/splat_import.rs:168-312 (ASCII / binary LE / binary BE, required
properties, `f_rest_*` channel-major de-interleave, truncation to SH degree 3) and
crates/brush-dataset/src/splat_export.rs:11-106 (property order, comments, binary little endian).
`scale_*` are log-scales, `opacity` is the raw (pre-sigmoid) value, `rot_*` is (w, x, y, z).
"""
from __future__ import annotations

import io

import numpy as np

_PLY_TYPES = {
    "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
    "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
    "double": "f8", "float64": "f8",
}
_MIN_PROPS = ["x", "y", "z", "scale_0", "scale_1", "scale_2", "opacity", "rot_0", "rot_1", "rot_2", "rot_3",
              "f_dc_0", "f_dc_1", "f_dc_2"]  # splat_import.rs:198-202
_MAX_SH_COEFFS = 16  # "Limit the number of imported SH channels for now" (splat_import.rs:241-246)


def _parse_header(buf: bytes):
    end = buf.find(b"end_header")
    if end < 0 or not buf.startswith(b"ply"):
        raise ValueError("Invalid ply file")
    nl = buf.find(b"\n", end)
    lines = buf[:end].decode("ascii", "replace").splitlines()
    fmt, elements, cur = None, [], None
    for ln in lines[1:]:
        t = ln.split()
        if not t or t[0] == "comment":
            continue
        if t[0] == "format":
            fmt = t[1]
        elif t[0] == "element":
            cur = {"name": t[1], "count": int(t[2]), "props": []}
            elements.append(cur)
        elif t[0] == "property":
            if t[1] == "list":
                raise ValueError("list properties are not supported in splat ply files")
            cur["props"].append((t[2], _PLY_TYPES[t[1]]))
    if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
        raise ValueError(f"Invalid ply format {fmt}")
    return fmt, elements, nl + 1


def load_splat_from_ply(data) -> dict:
    """Returns dict(means[N,3], log_scales[N,3], rotation[N,4], raw_opacity[N], sh_coeffs[N,C,3]) f32."""
    if not isinstance(data, (bytes, bytearray)):
        with open(data, "rb") as f:
            data = f.read()
    fmt, elements, off = _parse_header(bytes(data))
    verts = None
    for el in elements:
        names = [p[0] for p in el["props"]]
        if fmt == "ascii":
            txt = io.BytesIO(data[off:])
            arr = np.loadtxt(txt, dtype=np.float64, max_rows=el["count"], ndmin=2) if el["count"] else np.zeros((0, len(names)))
            consumed = sum(len(l) for l in data[off:].splitlines(keepends=True)[: el["count"]])
            table = {n: arr[:, i] for i, n in enumerate(names)}
            off += consumed
        else:
            e = "<" if fmt == "binary_little_endian" else ">"
            dt = np.dtype([(n, e + t) for n, t in el["props"]])
            arr = np.frombuffer(data, dtype=dt, count=el["count"], offset=off)
            table = {n: arr[n] for n in names}
            off += dt.itemsize * el["count"]
        if el["name"] == "vertex":
            verts = table
    if verts is None:
        raise ValueError("Invalid ply file")
    if not all(p in verts for p in _MIN_PROPS):
        raise ValueError("Invalid splat ply. Missing properties!")

    def col(*names):
        return np.stack([np.asarray(verts[n], dtype=np.float32) for n in names], axis=1)

    n = len(verts["x"])
    if n == 0:
        raise ValueError("No splats found")
    rest_ids = sorted(int(k[len("f_rest_"):]) for k in verts if k.startswith("f_rest_") and k[len("f_rest_"):].isdigit())
    n_rest = (max(rest_ids) + 1) if rest_ids else 0
    dc = col("f_dc_0", "f_dc_1", "f_dc_2")  # [N,3]
    if n_rest:
        rest = col(*[f"f_rest_{i}" for i in range(n_rest)])  # channel-major: [R.., G.., B..]
        cpc = n_rest // 3
        rest = rest[:, : cpc * 3].reshape(n, 3, cpc).transpose(0, 2, 1)  # interleave_coeffs -> [N, cpc, 3]
        sh = np.concatenate([dc[:, None, :], rest], axis=1)
    else:
        sh = dc[:, None, :]
    sh = sh[:, :_MAX_SH_COEFFS]
    return {
        "means": col("x", "y", "z"),
        "log_scales": col("scale_0", "scale_1", "scale_2"),
        "rotation": col("rot_0", "rot_1", "rot_2", "rot_3"),
        "raw_opacity": np.asarray(verts["opacity"], dtype=np.float32).copy(),
        "sh_coeffs": np.ascontiguousarray(sh, dtype=np.float32),
    }


def splat_to_ply(means, log_scales, rotation, raw_opacity, sh_coeffs) -> bytes:
    """Binary little-endian PLY in the reference's property order (splat_export.rs:67-105)."""
    means = np.asarray(means, np.float32)
    n, c = means.shape[0], np.asarray(sh_coeffs).shape[1]
    sh = np.asarray(sh_coeffs, np.float32).transpose(0, 2, 1)  # Inria layout [n, channel, coeffs]
    dc = sh[:, :, 0]
    rest = sh[:, :, 1:].reshape(n, 3 * (c - 1))
    names = list(_MIN_PROPS) + [f"f_rest_{i}" for i in range(3 * (c - 1))]
    cols = np.concatenate([means, np.asarray(log_scales, np.float32), np.asarray(raw_opacity, np.float32)[:, None],
                           np.asarray(rotation, np.float32), dc, rest], axis=1).astype("<f4")
    header = ["ply", "format binary_little_endian 1.0", "comment Exported from Brush", "comment Vertical axis: y",
              f"element vertex {n}"] + [f"property float {nm}" for nm in names] + ["end_header"]
    return ("\n".join(header) + "\n").encode("ascii") + np.ascontiguousarray(cols).tobytes()
