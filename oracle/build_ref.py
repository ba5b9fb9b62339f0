#!/usr/bin/env python
"""Builds the part of the reference that compiles from its own sources in this image: the CPU NMS extension of the
mmdetection tree (M/mmdet/ops/nms/src/nms_cpu.cpp: one C++ file against the torch headers).  Output only into
oracle/_ref/ (git-ignored); the sources are compiled from where they lie under /root/reference, nothing is copied.
TEST INFRASTRUCTURE ONLY: tests/test_oracle_ops.py uses it (when present) to check oracle.ops.nms(mode="cpu").
Everything else compiled in the reference is CUDA (nvcc, PyTorch-1.1 THC API) and cannot be built here.
NEVER pass a reference .cu file to torch.utils.cpp_extension.load(): on a ROCm build it hipifies IN PLACE, i.e. writes
*_hip_kernel.hip next to the source inside the read-only /root/reference tree.  Plain .cpp sources are compiled from
where they lie and only oracle/_ref/ is written.

    python oracle/build_ref.py            # no-op when /root/reference is absent (GPU box)
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/SipMask-mmdetection/mmdet/ops/nms/src/nms_cpu.cpp"


def build(verbose=False):
    if not os.path.exists(SRC):
        return None
    out = os.path.join(HERE, "_ref")
    os.makedirs(out, exist_ok=True)
    from torch.utils.cpp_extension import load
    # -w: the file targets the PyTorch 1.1 C++ API (Tensor::type(), data<T>()), still accepted with deprecation warnings
    return load(name="ref_nms_cpu", sources=[SRC], build_directory=out, extra_cflags=["-O2", "-w"], verbose=verbose)


if __name__ == "__main__":
    m = build(verbose=True)
    print("built" if m is not None else "reference not present: nothing to build", getattr(m, "__file__", ""))
