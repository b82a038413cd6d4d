import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _poison_torch_allocator():
    """Debugging aid, RLN_POISON_TORCH=<GiB>: before the first test, that much device memory is filled with NaN patterns
    and handed back to torch's caching allocator, so every later `torch.empty` block starts out as NaN instead of the
    zeros a freshly mapped page usually holds.  A kernel that reads memory nobody wrote (past the end of an input, a
    tensor's alignment gap) then fails every time rather than on the odd box whose memory was not zero."""
    gib = float(os.environ.get("RLN_POISON_TORCH", "0") or 0)
    if gib > 0:
        import torch
        if torch.cuda.is_available():
            blocks = [torch.full((int(256 * 2 ** 20),), float("nan"), dtype=torch.float32, device="cuda")
                      for _ in range(int(gib))]          # 1 GiB each: reusable for large and (split) small requests
            small = [torch.full((int(2 ** 18),), float("nan"), dtype=torch.float32, device="cuda") for _ in range(64)]
            del blocks, small
    yield
