"""Opt-in per-stage GPU timing (hipEvents recorded inside the C ABI calls).

Counterpart of the reference's tracing spans + sync-span layer (crates/brush-render/src/render.rs
spans, crates/sync-span/src/lib.rs:12-49).  Nothing is recorded unless a StageProfiler is attached.
"""
from __future__ import annotations

import ctypes as C

from . import _lib


class StageProfiler:
    def __init__(self):
        self._l = _lib.lib()
        h = C.c_void_p()
        _lib.check(self._l.brush_profiler_create(C.byref(h)), "brush_profiler_create")
        self._h = h
        self.names = [self._l.brush_stage_name(i).decode() for i in range(_lib.NUM_STAGES)]

    def __enter__(self):
        self._l.brush_profiler_attach(self._h)
        return self

    def __exit__(self, *exc):
        self._l.brush_profiler_attach(None)
        return False

    def read_ms(self) -> dict:
        """Milliseconds per stage of the last recorded forward and backward (synchronises)."""
        buf = (C.c_float * _lib.NUM_STAGES)()
        _lib.check(self._l.brush_profiler_read(self._h, buf), "brush_profiler_read")
        return {n: float(buf[i]) for i, n in enumerate(self.names)}

    def stop_after(self, stage) -> None:
        """Measurement only: while attached, forward / backward end behind `stage` (name or index; None = whole pass)
        and record no events (brush_profiler_stop_after)."""
        idx = -1 if stage is None else (self.names.index(stage) if isinstance(stage, str) else int(stage))
        _lib.check(self._l.brush_profiler_stop_after(self._h, idx), "brush_profiler_stop_after")

    def close(self):
        if self._h:
            self._l.brush_profiler_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
