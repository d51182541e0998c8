"""Regenerates tests/golden/ from the reference checkout (run in the build container only).

The three files are DATA the reference's own test (`crates/brush-render/src/render.rs:695-833`)
loads: gsplat-generated input/expected tensors and the target image.  They are copied
byte-for-byte; crab.png is additionally decoded to crab_rgb_u8.npy so that the tests need no
PNG decoder.  No reference source text is copied.
"""
import os
import shutil

import numpy as np

REF = "/root/reference/crates/brush-render/test_cases"
HERE = os.path.dirname(os.path.abspath(__file__))

if __name__ == "__main__":
    for f in ("tiny_case.safetensors", "basic_case.safetensors", "crab.png"):
        shutil.copyfile(os.path.join(REF, f), os.path.join(HERE, f))
    from PIL import Image

    img = np.asarray(Image.open(os.path.join(HERE, "crab.png")).convert("RGB"), dtype=np.uint8)
    np.save(os.path.join(HERE, "crab_rgb_u8.npy"), img)
    print("crab", img.shape)
