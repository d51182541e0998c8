"""Camera — mirror of crates/brush-render/src/camera.rs:1-58."""
from __future__ import annotations

import math

import numpy as np


def fov_to_focal(fov_rad: float, pixels: int) -> float:
    """camera.rs:50-52"""
    return 0.5 * float(pixels) / math.tan(fov_rad * 0.5)


def focal_to_fov(focal: float, pixels: int) -> float:
    """camera.rs:55-57"""
    return 2.0 * math.atan(float(pixels) / (2.0 * focal))


def _quat_xyzw_to_mat3(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([
        [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
        [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
        [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
    ], dtype=np.float64)


class Camera:
    """position / rotation are local-to-world (glam Vec3 / Quat in (x,y,z,w) order)."""

    def __init__(self, position, rotation, fov_x: float, fov_y: float, center_uv=(0.5, 0.5)):
        self.position = np.asarray(position, dtype=np.float32)
        self.rotation = np.asarray(rotation, dtype=np.float32)
        self.fov_x = float(fov_x)
        self.fov_y = float(fov_y)
        self.center_uv = (float(center_uv[0]), float(center_uv[1]))

    def focal(self, img_size):
        """camera.rs:28-33"""
        return (np.float32(fov_to_focal(self.fov_x, img_size[0])), np.float32(fov_to_focal(self.fov_y, img_size[1])))

    def center(self, img_size):
        """camera.rs:35-40"""
        return (np.float32(np.float32(self.center_uv[0]) * np.float32(img_size[0])),
                np.float32(np.float32(self.center_uv[1]) * np.float32(img_size[1])))

    def local_to_world(self):
        """camera.rs:42-44"""
        m = np.eye(4, dtype=np.float64)
        m[:3, :3] = _quat_xyzw_to_mat3(self.rotation)
        m[:3, 3] = self.position.astype(np.float64)
        return m

    def world_to_local(self):
        """camera.rs:46-48 (row-major numpy 4x4; the op stores it column-major)."""
        return np.linalg.inv(self.local_to_world()).astype(np.float32)
