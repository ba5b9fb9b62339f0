#!/usr/bin/env python
"""Builds the parts of the reference that can be compiled from their own sources in this image; output only into
oracle/_ref/ (git-ignored, shipped to the GPU box with the snapshot); the sources are compiled from where they lie
under /root/reference, nothing is copied.  TEST INFRASTRUCTURE ONLY (tests/test_oracle_ops.py, tests/test_ref_pins.py).

1. ref_nms_cpu          M/mmdet/ops/nms/src/nms_cpu.cpp, as it is (one C++ file against the torch headers).
2. the CUDA-only ops, compiled as HOST C++ through oracle/ref_shim/host_shim.h -- the device functions and the host
   algebra of the reference then run on the CPU and pin the oracle's restatements by execution:
     ref_deform_conv      M/mmdet/ops/dcn/src/deform_conv_cuda.cpp + deform_conv_cuda_kernel.cu
                          (deform_conv_forward_cuda, deform_conv_backward_input_cuda, deform_conv_backward_parameters_cuda)
     ref_crop_split       M/mmdet/ops/crop/src/crop_split_cuda.cpp + crop_split_cuda_kernel.cu
     ref_crop_split_gt    M/mmdet/ops/crop/src/crop_split_gt_cuda.cpp + crop_split_gt_cuda_kernel.cu
     ref_focal_loss       M/mmdet/ops/sigmoid_focal_loss/src/sigmoid_focal_loss.cpp + sigmoid_focal_loss_cuda.cu
   Each source is piped through `sed` on its way into g++ to cut the `<<<grid, block, ...>>>` launch syntax (not C++);
   with blockDim = gridDim = 1 in the shim the sources' own grid-stride loops visit every index in one call.

NEVER pass a reference .cu file to torch.utils.cpp_extension.load(): on a ROCm build it hipifies IN PLACE, i.e. writes
*_hip_kernel.hip next to the source inside the read-only /root/reference tree.  This script only READS the tree.

    python oracle/build_ref.py            # no-op when /root/reference is absent (GPU box)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")
OPS = "/root/reference/SipMask-mmdetection/mmdet/ops"
SRC = OPS + "/nms/src/nms_cpu.cpp"
HOST_SHIM_MODULES = {
    "ref_deform_conv": ["dcn/src/deform_conv_cuda.cpp", "dcn/src/deform_conv_cuda_kernel.cu"],
    "ref_crop_split": ["crop/src/crop_split_cuda.cpp", "crop/src/crop_split_cuda_kernel.cu"],
    "ref_crop_split_gt": ["crop/src/crop_split_gt_cuda.cpp", "crop/src/crop_split_gt_cuda_kernel.cu"],
    "ref_focal_loss": ["sigmoid_focal_loss/src/sigmoid_focal_loss.cpp", "sigmoid_focal_loss/src/sigmoid_focal_loss_cuda.cu"],
}


def build(verbose=False):
    if not os.path.exists(SRC):
        return None
    os.makedirs(OUT, exist_ok=True)
    from torch.utils.cpp_extension import load
    # -w: the file targets the PyTorch 1.1 C++ API (Tensor::type(), data<T>()), still accepted with deprecation warnings
    return load(name="ref_nms_cpu", sources=[SRC], build_directory=OUT, extra_cflags=["-O2", "-w"], verbose=verbose)


def build_host_shim(verbose=False, force=False):
    """the CUDA-only ops as host C++ (see the module docstring); returns the list of built .so paths"""
    if not os.path.exists(OPS):
        return []
    import torch
    from torch.utils.cpp_extension import include_paths, library_paths
    os.makedirs(OUT, exist_ok=True)
    shim = os.path.join(HERE, "ref_shim")
    inc = ["-I" + shim] + ["-I" + p for p in include_paths()] + ["-I" + sysconfig.get_paths()["include"]]
    libs = ["-L" + p for p in library_paths()] + ["-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python",
                                                  "-Wl,-rpath," + library_paths()[0]]
    abi = "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)
    built = []
    shim_h = os.path.join(shim, "host_shim.h")
    for name, srcs in HOST_SHIM_MODULES.items():
        so = os.path.join(OUT, name + ".so")
        deps = [os.path.join(OPS, r) for r in srcs] + [shim_h, os.path.abspath(__file__)]
        if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
            built.append(so)           # up to date (each module costs ~1 min of torch-header parsing)
            continue
        objs = []
        for i, rel in enumerate(srcs):
            src = os.path.join(OPS, rel)
            obj = os.path.join(OUT, "%s_%d.o" % (name, i))
            # the launch syntax is cut on the fly; the file itself is only read
            cmd = ("sed -E 's/<<<[^;]*>>>//' '%s' | g++ -x c++ -std=c++17 -O1 -w -fPIC %s -DTORCH_EXTENSION_NAME=%s "
                   "-DTORCH_API_INCLUDE_EXTENSION_H %s -include '%s/host_shim.h' -c - -o '%s'"
                   % (src, abi, name, " ".join(inc), shim, obj))
            if verbose:
                print(cmd)
            subprocess.check_call(["bash", "-o", "pipefail", "-c", cmd])
            objs.append(obj)
        subprocess.check_call(["g++", "-shared", "-o", so] + objs + libs)
        for o in objs:
            os.remove(o)
        built.append(so)
    return built


def load_host_shim(name):
    """import a module built by build_host_shim from oracle/_ref (None if it is not there)"""
    import importlib.util
    so = os.path.join(OUT, name + ".so")
    if not os.path.exists(so):
        return None
    import torch  # noqa: F401  (the module links against libtorch)
    spec = importlib.util.spec_from_file_location(name, so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose=True)
    print("built" if m is not None else "reference not present: nothing to build", getattr(m, "__file__", ""))
    for so in build_host_shim(verbose=False):
        print("built", so)
