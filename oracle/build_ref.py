#!/usr/bin/env python3
"""Build the REFERENCE's own CPU sources (unmodified, where they lie under
/root/reference) into oracle/_ref/.  TEST INFRASTRUCTURE ONLY.

Only runs where /root/reference exists (the build container).  Outputs go to
oracle/_ref/ exclusively (git-ignored, but shipped to the GPU box by gpurun so
that tests and bench.py's cpu_baseline can time the real reference CPU ops).
Nothing from the reference is copied into the repository: the compiler reads
the sources in place.

Modules produced (names follow the reference's pybind modules):
  ref_box_iou_rotated   utils/box_iou_rotated/src/box_iou_rotated_cpu.cpp
  ref_nms_rotated       utils/nms_rotated/src/nms_rotated_cpu.cpp
  ref_ml_nms_rotated    utils/ml_nms_rotated/src/nms_rotated_cpu.cpp
  ref_orn               models/orn/src/{vision.cpp,cpu/*.cpp}
  _polyiou              DOTA_devkit/polyiou/csrc/{polyiou_wrap.cxx,polyiou.cpp} (pre-generated SWIG)
  ref_geom_gpubranch.so the reference geometry header compiled on the host with
                        __CUDACC__ defined (the GPU swap-sort branch), C ABI wrapper
                        written here (oracle/ref_geom_gpubranch.cpp)
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("S2A_REFERENCE", "/root/reference")
OUT = os.path.join(HERE, "_ref")


def have_reference():
    return os.path.isdir(os.path.join(REF, "utils", "box_iou_rotated", "src"))


def _torch_ext(name, sources, extra_cflags=()):
    from torch.utils.cpp_extension import load
    bdir = os.path.join(OUT, name)
    os.makedirs(bdir, exist_ok=True)
    return load(name=name, sources=sources, build_directory=bdir,
                extra_cflags=["-O2", *extra_cflags], verbose=False)


def build_all(verbose=True):
    if not have_reference():
        if verbose:
            print("[build_ref] reference tree absent; using prebuilt oracle/_ref if present")
        return False
    os.makedirs(OUT, exist_ok=True)
    os.environ.setdefault("MAX_JOBS", "4")
    _torch_ext("ref_box_iou_rotated",
               [f"{REF}/utils/box_iou_rotated/src/box_iou_rotated_cpu.cpp"])
    _torch_ext("ref_nms_rotated",
               [f"{REF}/utils/nms_rotated/src/nms_rotated_cpu.cpp"])
    _torch_ext("ref_ml_nms_rotated",
               [f"{REF}/utils/ml_nms_rotated/src/nms_rotated_cpu.cpp"])
    _torch_ext("ref_orn",
               [f"{REF}/models/orn/src/vision.cpp",
                f"{REF}/models/orn/src/cpu/ActiveRotatingFilter_cpu.cpp",
                f"{REF}/models/orn/src/cpu/RotationInvariantEncoding_cpu.cpp"],
               extra_cflags=["-fopenmp"])
    # polyiou: the reference ships the SWIG-generated wrapper; plain g++ builds it.
    pdir = os.path.join(OUT, "polyiou")
    os.makedirs(pdir, exist_ok=True)
    so = os.path.join(pdir, "_polyiou" + sysconfig.get_config_var("EXT_SUFFIX"))
    csrc = f"{REF}/DOTA_devkit/polyiou/csrc"
    if not os.path.exists(so):
        subprocess.check_call(
            ["g++", "-O2", "-fPIC", "-shared", "-w",
             "-I" + sysconfig.get_paths()["include"], "-I" + csrc,
             f"{csrc}/polyiou_wrap.cxx", f"{csrc}/polyiou.cpp", "-o", so])
    # GPU (swap-sort) branch of the reference geometry header, host-compiled.
    gso = os.path.join(OUT, "ref_geom_gpubranch.so")
    src = os.path.join(HERE, "ref_geom_gpubranch.cpp")
    if (not os.path.exists(gso)) or os.path.getmtime(gso) < os.path.getmtime(src):
        subprocess.check_call(
            ["g++", "-O2", "-fPIC", "-shared", "-DNDEBUG", "-ffp-contract=off",
             "-I" + f"{REF}/utils/ml_nms_rotated/src", "-I" + f"{REF}/utils/box_iou_rotated/src",
             src, "-o", gso])
    if verbose:
        print("[build_ref] built reference CPU ops into", OUT)
    return True


if __name__ == "__main__":
    ok = build_all()
    sys.exit(0 if ok else 1)
