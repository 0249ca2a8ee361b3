"""Compile dropin_fast.cpp into cdv_slam_amd/_dropin_fast.so (a CPython extension on torch's C++ API; host code only, g++).
Called by the Makefile (`make dropin_fast`, part of `all`); in-tree so that the built module travels with the repository."""
import os
import subprocess
import sys
import sysconfig

import torch
from torch.utils import cpp_extension

here = os.path.dirname(os.path.abspath(__file__))
out = os.path.join(os.path.dirname(here), "_dropin_fast.so")
libdir = os.path.join(os.path.dirname(torch.__file__), "lib")
cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", os.path.join(here, "dropin_fast.cpp"), "-o", out,
       "-DTORCH_EXTENSION_NAME=_dropin_fast", "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI),
       "-I" + sysconfig.get_paths()["include"]] + ["-I" + p for p in cpp_extension.include_paths()] + \
      ["-L" + libdir, "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python", "-Wl,-rpath," + libdir, "-Wall", "-Wno-unused-function"]
print(" ".join(cmd), flush=True)
sys.exit(subprocess.call(cmd))
