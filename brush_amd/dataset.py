"""Dataset readers: NeRF-synthetic and COLMAP camera conventions.

Caller-side host code (no device work): turns the files of the two dataset families the reference
loads into `Camera`s for the op and uint8 images for the trainer.

  nerf_camera / read_nerf_synthetic   <- crates/brush-dataset/src/formats/nerf_synthetic.rs:26-160
  colmap_camera / read_colmap         <- crates/brush-dataset/src/formats/colmap.rs:15-146
  read_colmap_cameras/images/points3d <- crates/colmap-reader/src/lib.rs (binary and text)
  colmap_initial_points               <- colmap.rs:148-195 (+ Splats.from_point_cloud)
  clamp_img_to_max_size               <- crates/brush-dataset/src/lib.rs:57-69
  Scene.bounds                        <- crates/brush-train/src/scene.rs:42-55

A dataset root is a directory or a .zip file (the reference only takes zips, zip.rs); files are
located by suffix like `DatasetZip::find_base_path` (zip.rs:81-92).  No dataset files exist in the
build image, so the tests drive these readers with small files written by the tests themselves.
"""
from __future__ import annotations

import io
import json
import math
import os
import struct
import zipfile
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from .camera import Camera, focal_to_fov, fov_to_focal


# ---------------------------------------------------------------------------- archive access
class DatasetFiles:
    """Uniform view of a directory tree or a zip archive (zip.rs:20-92)."""

    def __init__(self, root: str):
        self.root = root
        self._zip = zipfile.ZipFile(root) if os.path.isfile(root) else None
        if self._zip is not None:
            self._names = [n for n in self._zip.namelist() if not n.endswith("/")]
        else:
            self._names = []
            for d, _, files in os.walk(root):
                for f in files:
                    self._names.append(os.path.relpath(os.path.join(d, f), root).replace(os.sep, "/"))
        self._names.sort()

    def find_base_path(self, search_path: str) -> Optional[str]:
        """Directory prefix of the first file whose path ends with `search_path` (zip.rs:81-92)."""
        parts = search_path.split("/")
        for n in self._names:
            comps = n.split("/")
            if comps[-len(parts):] == parts:
                return "/".join(comps[:-len(parts)])
        return None

    def read_bytes(self, path: str) -> bytes:
        path = path.lstrip("/")
        if self._zip is not None:
            return self._zip.read(path)
        with open(os.path.join(self.root, path), "rb") as f:
            return f.read()

    @staticmethod
    def join(base: str, rel: str) -> str:
        return os.path.normpath(os.path.join(base, rel)).replace(os.sep, "/") if base else os.path.normpath(rel).replace(os.sep, "/")


# ---------------------------------------------------------------------------- scene types
@dataclass
class SceneView:
    """scene.rs:13-18"""
    name: str
    camera: Camera
    image: np.ndarray  # [h, w, 3|4] uint8

    def image_f32(self) -> np.ndarray:
        """The trainer's target tensor: u8 / 255, alpha kept when present (brush-train/src/image.rs)."""
        return self.image.astype(np.float32) / 255.0


@dataclass
class Scene:
    views: List[SceneView] = field(default_factory=list)

    def bounds(self, cam_near: float, cam_far: float) -> Tuple[np.ndarray, np.ndarray]:
        """(min, max) of the camera frusta's near/far points along +z (scene.rs:42-55)."""
        from .camera import _quat_xyzw_to_mat3

        lo, hi = np.full(3, np.inf), np.full(3, -np.inf)
        for v in self.views:
            fwd = _quat_xyzw_to_mat3(v.camera.rotation) @ np.array([0.0, 0.0, 1.0])
            for d in (cam_near, cam_far):
                p = v.camera.position.astype(np.float64) + fwd * d
                lo, hi = np.minimum(lo, p), np.maximum(hi, p)
        return lo.astype(np.float32), hi.astype(np.float32)


@dataclass
class Dataset:
    """brush-dataset/src/lib.rs:32-55"""
    train: Scene
    eval: Optional[Scene] = None

    @classmethod
    def from_views(cls, train_views, eval_views):
        return cls(Scene(list(train_views)), Scene(list(eval_views)) if eval_views else None)


def _decode_image(data: bytes) -> np.ndarray:
    from PIL import Image

    img = Image.open(io.BytesIO(data))
    img = img.convert("RGBA" if ("A" in img.getbands() or "transparency" in img.info) else "RGB")
    return np.asarray(img, dtype=np.uint8)


def clamp_img_to_max_size(image: np.ndarray, max_size: int) -> np.ndarray:
    """lib.rs:57-69 (Lanczos3; `DynamicImage::resize` keeps the aspect ratio inside the new box)."""
    h, w = image.shape[:2]
    if w <= max_size and h <= max_size:
        return image
    aspect = np.float32(w) / np.float32(h)
    if w > h:
        nw, nh = max_size, int(np.float32(max_size) / aspect)
    else:
        nw, nh = int(np.float32(max_size) * aspect), max_size
    # image::DynamicImage::resize fits the image into (nw, nh) preserving the aspect ratio
    ratio = min(nw / w, nh / h)
    fw, fh = max(int(round(w * ratio)), 1), max(int(round(h * ratio)), 1)
    from PIL import Image

    return np.asarray(Image.fromarray(image).resize((fw, fh), Image.LANCZOS), dtype=np.uint8)


# ---------------------------------------------------------------------------- small glam equivalents
def _quat_from_mat3(m: np.ndarray) -> np.ndarray:
    """Unit quaternion (x, y, z, w) of a rotation matrix (columns = rotated axes)."""
    m = np.asarray(m, dtype=np.float64)
    t = m[0, 0] + m[1, 1] + m[2, 2]
    if t > 0:
        s = math.sqrt(t + 1.0) * 2
        q = [(m[2, 1] - m[1, 2]) / s, (m[0, 2] - m[2, 0]) / s, (m[1, 0] - m[0, 1]) / s, 0.25 * s]
    elif m[0, 0] > m[1, 1] and m[0, 0] > m[2, 2]:
        s = math.sqrt(1.0 + m[0, 0] - m[1, 1] - m[2, 2]) * 2
        q = [0.25 * s, (m[0, 1] + m[1, 0]) / s, (m[0, 2] + m[2, 0]) / s, (m[2, 1] - m[1, 2]) / s]
    elif m[1, 1] > m[2, 2]:
        s = math.sqrt(1.0 + m[1, 1] - m[0, 0] - m[2, 2]) * 2
        q = [(m[0, 1] + m[1, 0]) / s, 0.25 * s, (m[1, 2] + m[2, 1]) / s, (m[0, 2] - m[2, 0]) / s]
    else:
        s = math.sqrt(1.0 + m[2, 2] - m[0, 0] - m[1, 1]) * 2
        q = [(m[0, 2] + m[2, 0]) / s, (m[1, 2] + m[2, 1]) / s, 0.25 * s, (m[1, 0] - m[0, 1]) / s]
    q = np.array(q)
    return (q / np.linalg.norm(q)).astype(np.float32)


def _scale_rotation_translation(m: np.ndarray):
    """glam Mat4::to_scale_rotation_translation for an affine 4x4 (row-major numpy)."""
    m = np.asarray(m, dtype=np.float64)
    det = np.linalg.det(m[:3, :3])
    scale = np.array([np.linalg.norm(m[:3, 0]) * (1.0 if det >= 0 else -1.0), np.linalg.norm(m[:3, 1]),
                      np.linalg.norm(m[:3, 2])])
    rot = m[:3, :3] / scale[None, :]
    return scale, _quat_from_mat3(rot), m[:3, 3].astype(np.float32)


# ---------------------------------------------------------------------------- NeRF synthetic
def nerf_camera(transform_matrix, fovx: float, img_w: int, img_h: int) -> Camera:
    """nerf_synthetic.rs:56-88.  `transform_matrix` is the camera-to-world 4x4 of transforms_*.json
    (rows as in the file).  The y and z axes are flipped (OpenGL camera -> y-down, z-forward) and the
    world is rotated by +90 degrees about x (z-up -> the kernel's y-down frame)."""
    t = np.asarray(transform_matrix, dtype=np.float64).reshape(4, 4).copy()
    t[:, 1] *= -1.0  # transform.y_axis *= -1  (columns of the row-major matrix)
    t[:, 2] *= -1.0
    c, s = math.cos(math.pi / 2.0), math.sin(math.pi / 2.0)
    rx = np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1]], dtype=np.float64)
    t = rx @ t
    _, rotation, translation = _scale_rotation_translation(t)
    fovy = focal_to_fov(fov_to_focal(fovx, img_w), img_h)
    return Camera(translation, rotation, fovx, fovy, (0.5, 0.5))


def _read_transforms(files: DatasetFiles, name: str, max_frames, max_resolution) -> Optional[List[SceneView]]:
    base = files.find_base_path(name)
    if base is None:
        return None
    scene = json.loads(files.read_bytes(DatasetFiles.join(base, name)).decode("utf-8"))
    fovx = float(scene["camera_angle_x"])
    views = []
    for frame in scene["frames"][: max_frames if max_frames is not None else None]:
        path = DatasetFiles.join(base, frame["file_path"] + ".png")
        img = _decode_image(files.read_bytes(path))
        if max_resolution is not None:
            img = clamp_img_to_max_size(img, max_resolution)
        views.append(SceneView(path, nerf_camera(frame["transform_matrix"], fovx, img.shape[1], img.shape[0]), img))
    return views


def read_nerf_synthetic(root: str, max_frames: Optional[int] = None, max_resolution: Optional[int] = None,
                        eval_split_every: Optional[int] = None) -> Dataset:
    """nerf_synthetic.rs:98-160: transforms_train.json (+ transforms_val.json as eval views;
    transforms_test.json is ignored, as in the reference)."""
    files = DatasetFiles(root)
    train_all = _read_transforms(files, "transforms_train.json", max_frames, max_resolution)
    if train_all is None:
        raise FileNotFoundError("No transforms file found")
    val = _read_transforms(files, "transforms_val.json", max_frames, max_resolution)
    train, evals = [], []
    for i, v in enumerate(train_all):
        # the reference moves every eval_period-th train view to eval only when a val file exists
        if eval_split_every is not None and i % eval_split_every == 0 and val is not None:
            evals.append(v)
        else:
            train.append(v)
    evals.extend(val or [])
    return Dataset.from_views(train, evals)


# ---------------------------------------------------------------------------- COLMAP
_COLMAP_MODELS = {  # id: (name, num_params, focal-y index, principal-x index, principal-y index)
    0: ("SIMPLE_PINHOLE", 3, 0, 1, 2), 1: ("PINHOLE", 4, 1, 2, 3), 2: ("SIMPLE_RADIAL", 4, 0, 1, 2),
    3: ("RADIAL", 5, 0, 1, 2), 4: ("OPENCV", 8, 1, 2, 3), 5: ("OPENCV_FISHEYE", 8, 1, 2, 3),
    6: ("FULL_OPENCV", 12, 1, 2, 3), 7: ("FOV", 5, 1, 2, 3), 8: ("SIMPLE_RADIAL_FISHEYE", 4, 0, 1, 2),
    9: ("RADIAL_FISHEYE", 5, 0, 1, 2), 10: ("THIN_PRISM_FISHEYE", 12, 1, 2, 3),
}
_COLMAP_MODEL_IDS = {v[0]: k for k, v in _COLMAP_MODELS.items()}


@dataclass
class ColmapCamera:
    """colmap-reader/src/lib.rs:57-133"""
    id: int
    model: int
    width: int
    height: int
    params: List[float]

    def focal(self) -> Tuple[float, float]:
        return self.params[0], self.params[_COLMAP_MODELS[self.model][2]]

    def principal_point(self) -> Tuple[float, float]:
        m = _COLMAP_MODELS[self.model]
        return float(np.float32(self.params[m[3]])), float(np.float32(self.params[m[4]]))


@dataclass
class ColmapImage:
    tvec: np.ndarray      # world-to-camera translation
    quat_wxyz: np.ndarray  # world-to-camera rotation, COLMAP order (w, x, y, z)
    camera_id: int
    name: str
    xys: np.ndarray
    point3d_ids: np.ndarray


@dataclass
class ColmapPoint3D:
    xyz: np.ndarray
    rgb: Tuple[int, int, int]
    error: float


def _model_id(token: str) -> int:
    """cameras.txt names the model (COLMAP's own writer); the reference parses a numeric id
    (colmap-reader lib.rs:163) — both are accepted."""
    if token in _COLMAP_MODEL_IDS:
        return _COLMAP_MODEL_IDS[token]
    mid = int(token)
    if mid not in _COLMAP_MODELS:
        raise ValueError("Invalid camera model")
    return mid


def read_colmap_cameras(data: bytes, is_binary: bool) -> Dict[int, ColmapCamera]:
    cams: Dict[int, ColmapCamera] = {}
    if is_binary:  # lib.rs:196-228
        (n,) = struct.unpack_from("<Q", data, 0)
        off = 8
        for _ in range(n):
            cid, mid, w, h = struct.unpack_from("<iiQQ", data, off)
            off += 24
            if mid not in _COLMAP_MODELS:
                raise ValueError("Invalid camera model")
            k = _COLMAP_MODELS[mid][1]
            params = list(struct.unpack_from(f"<{k}d", data, off))
            off += 8 * k
            cams[cid] = ColmapCamera(cid, mid, w, h, params)
        return cams
    for line in data.decode("utf-8").splitlines():  # lib.rs:140-194
        if line.startswith("#") or not line.strip():
            continue
        parts = line.split()
        if len(parts) < 4:
            raise ValueError("Invalid camera data")
        mid = _model_id(parts[1])
        params = [float(x) for x in parts[4:]]
        if len(params) != _COLMAP_MODELS[mid][1]:
            raise ValueError("Invalid number of camera parameters")
        cams[int(parts[0])] = ColmapCamera(int(parts[0]), mid, int(parts[2]), int(parts[3]), params)
    return cams


def read_colmap_images(data: bytes, is_binary: bool) -> Dict[int, ColmapImage]:
    imgs: Dict[int, ColmapImage] = {}
    if is_binary:  # lib.rs:290-346
        (n,) = struct.unpack_from("<Q", data, 0)
        off = 8
        for _ in range(n):
            (iid,) = struct.unpack_from("<i", data, off)
            q = np.array(struct.unpack_from("<4d", data, off + 4), dtype=np.float32)
            t = np.array(struct.unpack_from("<3d", data, off + 36), dtype=np.float32)
            (cid,) = struct.unpack_from("<i", data, off + 60)
            off += 64
            end = data.index(b"\0", off)
            name = data[off:end].decode("utf-8")
            off = end + 1
            (np2,) = struct.unpack_from("<Q", data, off)
            off += 8
            rec = np.frombuffer(data, dtype=np.dtype([("x", "<f8"), ("y", "<f8"), ("id", "<i8")]), count=np2, offset=off)
            off += 24 * np2
            imgs[iid] = ColmapImage(t, q, cid, name, np.stack([rec["x"], rec["y"]], 1).astype(np.float32),
                                    rec["id"].astype(np.int64))
        return imgs
    # Text: COLMAP writes two lines per image (pose line, then the 2-D points line).  The
    # reference reads points from the tail of the pose line (lib.rs:230-288); here the second
    # line is consumed as COLMAP defines it, and a tail on the pose line is accepted too.
    lines = [l for l in data.decode("utf-8").splitlines() if not l.startswith("#")]
    i = 0
    while i < len(lines):
        parts = lines[i].split()
        i += 1
        if not parts:
            continue
        if len(parts) < 10:
            raise ValueError("Invalid image data")
        q = np.array([float(x) for x in parts[1:5]], dtype=np.float32)
        t = np.array([float(x) for x in parts[5:8]], dtype=np.float32)
        tail = parts[10:]
        if not tail and i < len(lines):
            tail = lines[i].split()
            i += 1
        if len(tail) % 3:
            raise ValueError("Invalid image point data")
        pts = np.array(tail, dtype=np.float64).reshape(-1, 3) if tail else np.zeros((0, 3))
        imgs[int(parts[0])] = ColmapImage(t, q, int(parts[8]), parts[9], pts[:, :2].astype(np.float32),
                                          pts[:, 2].astype(np.int64))
    return imgs


def read_colmap_points3d(data: bytes, is_binary: bool) -> Dict[int, ColmapPoint3D]:
    pts: Dict[int, ColmapPoint3D] = {}
    if is_binary:  # lib.rs:401-440
        (n,) = struct.unpack_from("<Q", data, 0)
        off = 8
        for _ in range(n):
            pid, x, y, z, r, g, b, err, track = struct.unpack_from("<Q3d3BdQ", data, off)
            off += 8 + 24 + 3 + 8 + 8 + 8 * track
            pts[pid] = ColmapPoint3D(np.array([x, y, z], dtype=np.float32), (r, g, b), err)
        return pts
    for line in data.decode("utf-8").splitlines():  # lib.rs:348-399
        if line.startswith("#") or not line.strip():
            continue
        parts = line.split()
        if len(parts) < 8:
            raise ValueError("Invalid point3D data")
        pts[int(parts[0])] = ColmapPoint3D(np.array([float(x) for x in parts[1:4]], dtype=np.float32),
                                           (int(parts[4]), int(parts[5]), int(parts[6])), float(parts[7]))
    return pts


def _quat_wxyz_to_mat3(q):
    from .camera import _quat_xyzw_to_mat3

    w, x, y, z = [float(v) for v in q]
    n = math.sqrt(w * w + x * x + y * y + z * z)
    return _quat_xyzw_to_mat3([x / n, y / n, z / n, w / n])


def colmap_camera(quat_wxyz, tvec, cam: ColmapCamera) -> Camera:
    """colmap.rs:73-96: field of view from the focal lengths, principal point as a uv fraction,
    pose = inverse of COLMAP's world-to-camera (quat, tvec)."""
    focal = cam.focal()
    fovx, fovy = focal_to_fov(focal[0], cam.width), focal_to_fov(focal[1], cam.height)
    cx, cy = cam.principal_point()
    center_uv = (float(np.float32(cx) / np.float32(cam.width)), float(np.float32(cy) / np.float32(cam.height)))
    r_wc = _quat_wxyz_to_mat3(quat_wxyz)          # world -> camera
    r_cw = r_wc.T                                  # camera -> world
    position = -r_cw @ np.asarray(tvec, dtype=np.float64)
    return Camera(position.astype(np.float32), _quat_from_mat3(r_cw), fovx, fovy, center_uv)


def _colmap_paths(files: DatasetFiles):
    for is_binary, ext in ((True, "bin"), (False, "txt")):
        base = files.find_base_path(f"sparse/0/cameras.{ext}")
        if base is not None:
            return is_binary, base, ext
    raise FileNotFoundError("No COLMAP data found (either text or binary.")


def read_colmap(root: str, max_frames: Optional[int] = None, max_resolution: Optional[int] = None,
                eval_split_every: Optional[int] = None, load_images: bool = True) -> Dataset:
    """colmap.rs:15-146: views sorted by image id; every eval_split_every-th view goes to eval."""
    files = DatasetFiles(root)
    is_binary, base, ext = _colmap_paths(files)
    cams = read_colmap_cameras(files.read_bytes(DatasetFiles.join(base, f"sparse/0/cameras.{ext}")), is_binary)
    imgs = read_colmap_images(files.read_bytes(DatasetFiles.join(base, f"sparse/0/images.{ext}")), is_binary)
    train, evals = [], []
    for i, iid in enumerate(sorted(imgs)[: max_frames if max_frames is not None else None]):
        info = imgs[iid]
        path = DatasetFiles.join(base, f"images/{info.name}")
        if load_images:
            img = _decode_image(files.read_bytes(path))
            if max_resolution is not None:
                img = clamp_img_to_max_size(img, max_resolution)
        else:
            img = np.zeros((0, 0, 3), dtype=np.uint8)
        view = SceneView(path, colmap_camera(info.quat_wxyz, info.tvec, cams[info.camera_id]), img)
        (evals if eval_split_every is not None and i % eval_split_every == 0 else train).append(view)
    return Dataset.from_views(train, evals)


def colmap_initial_points(root: str) -> Tuple[np.ndarray, np.ndarray]:
    """colmap.rs:148-195: SfM points -> (positions [n,3] f32, colours [n,3] f32 in 0..1)."""
    files = DatasetFiles(root)
    is_binary, base, ext = _colmap_paths(files)
    pts = read_colmap_points3d(files.read_bytes(DatasetFiles.join(base, f"sparse/0/points3D.{ext}")), is_binary)
    keys = sorted(pts)
    pos = np.stack([pts[k].xyz for k in keys]).astype(np.float32) if keys else np.zeros((0, 3), np.float32)
    col = (np.array([pts[k].rgb for k in keys], dtype=np.float32) / 255.0) if keys else np.zeros((0, 3), np.float32)
    return pos, col


def splat_init_from_point_cloud(positions: np.ndarray, colors: np.ndarray, sh_degree: int) -> Dict[str, np.ndarray]:
    """Splats::from_point_cloud (crates/brush-render/src/gaussian_splats.rs:71-136): DC colour
    (c - 0.5) / SH_C0, identity rotation (w,x,y,z) = (1,0,0,0), opacity sigmoid^-1(0.1), isotropic
    scale = sqrt(sum of the 3 nearest squared distances, self included) / 3, clamped at 1e-7."""
    from scipy.spatial import cKDTree

    n = positions.shape[0]
    ncoef = (sh_degree + 1) ** 2
    sh = np.zeros((n, ncoef, 3), dtype=np.float32)
    sh[:, 0, :] = (colors.astype(np.float32) - np.float32(0.5)) / np.float32(0.2820947917738781)
    k = min(3, n)
    d, _ = cKDTree(positions.astype(np.float64)).query(positions.astype(np.float64), k=k)
    d = np.asarray(d, dtype=np.float64).reshape(n, k)
    extent = (np.sqrt(np.sum(d * d, axis=1)) / 3.0).astype(np.float32)
    log_scales = np.log(np.clip(extent, 1e-7, np.finfo(np.float32).max))[:, None].repeat(3, 1).astype(np.float32)
    quats = np.zeros((n, 4), dtype=np.float32)
    quats[:, 0] = 1.0
    raw_opac = np.full((n,), math.log(0.1 / 0.9), dtype=np.float32)
    return {"means": positions.astype(np.float32), "sh": sh, "quats": quats, "raw_opac": raw_opac,
            "log_scales": log_scales}
