"""Splats — the slice of crates/brush-render/src/gaussian_splats.rs the hot path needs:
the parameter container, `render` (quaternion normalisation outside the op, :167-188) and
`from_safetensors` (:208-223).  Initialisation / kd-tree code is out of scope.
"""
from __future__ import annotations

import torch

from .camera import Camera
from .render import render_splats


class Splats(torch.nn.Module):
    def __init__(self, means, sh_coeffs, rotation, raw_opacity, log_scales):
        super().__init__()
        self.means = torch.nn.Parameter(means.detach().clone())
        self.sh_coeffs = torch.nn.Parameter(sh_coeffs.detach().clone())
        self.rotation = torch.nn.Parameter(rotation.detach().clone())
        self.raw_opacity = torch.nn.Parameter(raw_opacity.detach().clone())
        self.log_scales = torch.nn.Parameter(log_scales.detach().clone())
        # carries the screen-space xy gradient (gaussian_splats.rs:157)
        self.xys_dummy = torch.zeros((means.shape[0], 2), dtype=torch.float32, device=means.device,
                                     requires_grad=True)
        # a SplatTrainer with deferred Adam of the SH block registers itself here; sync() applies what is pending
        self.lazy_sh_owner = None

    def sync(self):
        """Brings sh_coeffs up to date when a trainer defers their optimizer steps (SplatTrainer.sync); readers that
        bypass the trainer call this first (render and to_ply below do)."""
        if self.lazy_sh_owner is not None:
            self.lazy_sh_owner.sync(self)

    @classmethod
    def from_safetensors(cls, path_or_dict, device):
        """Key names of gaussian_splats.rs:208-223 (scales are log-scales, opacities raw)."""
        if isinstance(path_or_dict, dict):
            t = path_or_dict
        else:
            from safetensors.numpy import load_file
            t = load_file(path_or_dict)

        def dev(a):
            return torch.as_tensor(a, dtype=torch.float32, device=device)

        return cls(dev(t["means"]), dev(t["coeffs"]), dev(t["quats"]), dev(t["opacities"]), dev(t["scales"]))

    @classmethod
    def from_ply(cls, path_or_bytes, device):
        """crates/brush-dataset/src/splat_import.rs:183-312 (without the streaming updates)."""
        from .ply import load_splat_from_ply

        d = load_splat_from_ply(path_or_bytes)
        t = lambda a: torch.as_tensor(a, dtype=torch.float32, device=device)
        splats = cls(t(d["means"]), t(d["sh_coeffs"]), t(d["rotation"]), t(d["raw_opacity"]), t(d["log_scales"]))
        splats.norm_rotations()  # every import ends with norm_rotations() (splat_import.rs:139,150)
        return splats

    @torch.no_grad()
    def norm_rotations(self):
        """gaussian_splats.rs:190-196: rotation <- rotation / |rotation|, in place."""
        rot = self.rotation
        rot.copy_(rot / torch.sqrt(torch.sum(rot * rot, dim=1, keepdim=True)))

    def to_ply(self) -> bytes:
        """crates/brush-dataset/src/splat_export.rs:67-105"""
        from .ply import splat_to_ply

        self.sync()
        c = lambda p: p.detach().cpu().numpy()
        return splat_to_ply(c(self.means), c(self.log_scales), c(self.rotation), c(self.raw_opacity), c(self.sh_coeffs))

    def num_splats(self) -> int:
        return self.means.shape[0]

    def render(self, camera: Camera, img_size, render_u32_buffer: bool = False, max_intersects=None):
        """gaussian_splats.rs:167-188"""
        self.sync()
        rot = self.rotation
        norm_rot = rot / torch.sqrt(torch.sum(rot * rot, dim=1, keepdim=True))
        return render_splats(camera, img_size, self.means, self.xys_dummy, self.log_scales, norm_rot,
                             self.sh_coeffs, self.raw_opacity, render_u32_buffer, max_intersects)
