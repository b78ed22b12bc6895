import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "jpeg-encoder_amd", "python"), os.path.join(ROOT, "oracle"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(autouse=True, scope="session")
def _device_tensors_are_ready():
    """torch fills and uploads device tensors on its own stream; the library works on the stream of
    its context.  The tests hand such tensors straight to the library, with no event in between, so a
    fill that is still running could overwrite what the library has already written (seen once: the
    image generator's output against a late zero fill).  On a GPU box every tensor the tests create
    this way is complete before the call returns."""
    try:
        import torch
    except ImportError:
        yield
        return
    if not torch.cuda.is_available():
        yield
        return

    def finished(fn):
        def call(*args, **kwargs):
            t = fn(*args, **kwargs)
            if getattr(t, "is_cuda", False):
                torch.cuda.synchronize()
            return t
        return call

    names = ("zeros", "ones", "full", "zeros_like", "full_like")
    saved = {n: getattr(torch, n) for n in names}
    saved_cuda = torch.Tensor.cuda
    for n, fn in saved.items():
        setattr(torch, n, finished(fn))
    torch.Tensor.cuda = finished(saved_cuda)
    yield
    for n, fn in saved.items():
        setattr(torch, n, fn)
    torch.Tensor.cuda = saved_cuda
