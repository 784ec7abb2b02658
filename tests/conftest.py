import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the library's tuning / experiment switches (DE265HIP_*; csrc/env.h) are read only in a process that asks for them: the
# schedule variants and fault injections of these tests are such switches (child processes inherit the variable)
os.environ.setdefault("DE265HIP_TUNING", "1")
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


# Processes that use both torch and libde265_hip.so must import torch FIRST: the torch wheel bundles its own ROCm
# runtime under the same sonames as /opt/rocm's; whichever is loaded first serves both, and torch's device
# enumeration only works on top of its own (observed: "No HIP GPUs are available" when libde265_hip.so came first).
import torch  # noqa: E402,F401


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
