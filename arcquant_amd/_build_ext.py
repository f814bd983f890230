"""Build the CPython extension form of the boundary: ``arcquant_amd/lib/agemm.so`` (pybind11 + libtorch, csrc/agemm_ext.cpp) -- the
counterpart of the reference's ``kernels/build/agemm.so`` (CMakeLists.txt:51-64).  Plain g++: the extension contains no device code,
it links ``libarcq_hip.so`` (same directory, rpath $ORIGIN).  In-tree, so that the built file travels to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sysconfig

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "arcquant_amd", "csrc", "agemm_ext.cpp")
LIB_DIR = os.path.join(ROOT, "arcquant_amd", "lib")
OUT = os.path.join(LIB_DIR, "agemm.so")


def build_agemm_extension(force: bool = False) -> str:
    import torch
    from torch.utils import cpp_extension as ce
    deps = [SRC, os.path.join(ROOT, "include", "arcq.h")]       # (libarcq_hip.so is linked dynamically: a rebuilt library needs no relink)
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(d) for d in deps):
        return OUT
    tl = os.path.join(os.path.dirname(torch.__file__), "lib")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = [*ce.include_paths(), os.path.join(rocm, "include"), sysconfig.get_paths()["include"]]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=agemm",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations", *[f"-I{i}" for i in inc], SRC,
           "-L" + tl, "-ltorch", "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip", "-L" + LIB_DIR, "-larcq_hip",
           "-L" + os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + tl, "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", OUT]
    subprocess.check_call(cmd)
    return OUT


def import_agemm_extension():
    """``import agemm`` from arcquant_amd/lib (what a reference checkout does with kernels/build/); raises ImportError if it was not built."""
    import importlib.util
    if not os.path.exists(OUT):
        raise ImportError(f"{OUT} not built: python -c 'import __graft_entry__ as g; g.build()'")
    import torch  # noqa: F401  (libtorch must be loaded first)
    spec = importlib.util.spec_from_file_location("agemm", OUT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
