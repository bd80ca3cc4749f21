import os
import sys

# the oracle legs of the GPU tests run torch CPU ops on 32 threads: OpenMP workers that sleep between parallel regions instead of spinning make them ~30 % faster on the
# 128-core GPU boxes (bench.py's cpu_baseline sweep, round 6); must be set before torch loads its OpenMP runtime
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("KMP_BLOCKTIME", "0")

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vla-touch_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "needs_reference: needs /root/reference (build container only)")
    config.addinivalue_line("markers", "slow: long oracle episodes; still part of `-m gpu`, but collected LAST so that a time-limited run has finished every other parity test first")


def _gpu_ready() -> bool:
    """A plain `pytest tests` on a GPU-less box skips the gpu-marked tests instead of failing them.  On a GPU box a missing
    extension is NOT a skip: the product raises there (no CPU fallback) and the tests must go red."""
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    have_ref = os.path.isdir("/root/reference/VLA/residual_controller")
    skip_ref = pytest.mark.skip(reason="/root/reference absent (GPU box)")
    have_gpu = _gpu_ready()
    skip_gpu = pytest.mark.skip(reason="needs an MI355X and the built libvlatouch_hip.so (run `-m gpu` on the GPU box)")
    items.sort(key=lambda it: 1 if "slow" in it.keywords else 0)        # stable: the slow sub-marker only moves a test to the end of the run
    for it in items:
        if "needs_reference" in it.keywords and not have_ref:
            it.add_marker(skip_ref)
        if "gpu" in it.keywords and not have_gpu:
            it.add_marker(skip_gpu)
