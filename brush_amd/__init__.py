"""brush_amd — MI355X-native splat rasterizer behind brush-render's op surface.

Host-side mirror (Python, because no Rust toolchain exists in the build image) of
crates/brush-render's public interface for the forward+backward rasterizer path:

  render_splats / RenderAux  <- Backend::render_splats, RenderAux (src/lib.rs:20-86)
  Camera                     <- camera.rs
  Splats                     <- gaussian_splats.rs (render + from_safetensors only)
  radix_argsort              <- brush-sort/src/lib.rs:32-37
  prefix_sum                 <- brush-prefix-sum/src/lib.rs:17

All compute goes through the C ABI of include/brush_hip.h (libbrush_hip.so, hand-written HIP for
gfx950).  There is no CPU fallback: importing the compute entry points without the built
library raises.
"""
from .camera import Camera, fov_to_focal, focal_to_fov  # noqa: F401
from .render import (RenderAux, render_rgba8, render_splats, rgba8_row_pitch, sh_coeffs_for_degree,  # noqa: F401
                     sh_degree_from_coeffs)
from .sort import radix_argsort  # noqa: F401
from .prefix_sum import prefix_sum  # noqa: F401
from .gaussian_splats import Splats  # noqa: F401
from .train import SplatTrainer, TrainConfig  # noqa: F401
from . import dataset  # noqa: F401

__version__ = "0.4.0"
