"""vlatouch — MI355X (gfx950) engine for the VLA-Touch action-refinement path.

`residual_controller/` and `models/` next to this package mirror the reference's Python API
(SURVEY.md §8b); they hold parameters and call the engines in `vlatouch.engine`, which drive
hand-written HIP kernels in libvlatouch_hip.so through the C ABI of include/vlatouch.h.
"""
__all__ = ["synth"]
