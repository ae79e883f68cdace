"""A/B builds of one translation unit: compile SRC with extra flags and link it with the product build's other objects into
radzero_amd/libradzero_hip_<name>.so (git-ignored, travels with gpurun); select it with RZ_LIB_PATH=radzero_amd/libradzero_hip_<name>.so.
  python tools/build_variant.py prio1 attention.hip -DRZ_F32_ATTN_PRIO=1
"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radzero_amd import _lib
name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
_lib.build()                                             # the product objects are current
obj_dir = os.path.join(_lib.PKG_DIR, "build")
vobj = os.path.join(obj_dir, f"variant_{name}_{src.replace('.hip', '.o')}")
base = [f"--offload-arch={_lib.ARCH}", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]
subprocess.run([_lib.HIPCC, *base, *_lib.EXTRA_FLAGS.get(src, []), *flags, "-c", os.path.join(_lib.CSRC, src), "-o", vobj], check=True)
objs = [vobj if s == src else os.path.join(obj_dir, s.replace(".hip", ".o")) for s in _lib.SOURCES]
out = os.path.join(_lib.PKG_DIR, f"libradzero_hip_{name}.so")
subprocess.run([_lib.HIPCC, f"--offload-arch={_lib.ARCH}", "-shared", "-fPIC", "-o", out, *objs], check=True)
print(out)
