"""Loader for the REFERENCE's own CPU ops prebuilt into oracle/_ref/ (see build_ref.py).

TEST INFRASTRUCTURE ONLY.  On the GPU box /root/reference does not exist; the
prebuilt shared objects travel with the snapshot and are loaded from here.  Every
accessor returns None when the corresponding artefact is missing, so callers can
skip (tests) or fall back to the port (bench cpu_baseline kind="port").
"""
import ctypes
import glob
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
_REF = os.path.join(_HERE, "_ref")
_cache = {}


def _load_pyext(name, subdir=None):
    key = ("py", name)
    if key in _cache:
        return _cache[key]
    mod = None
    cands = glob.glob(os.path.join(_REF, subdir or name, name + "*.so"))
    if cands:
        try:
            import torch  # noqa: F401  (the extensions link against libtorch)
            spec = importlib.util.spec_from_file_location(name, cands[0])
            mod = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(mod)
        except Exception as e:  # pragma: no cover - diagnostic only
            sys.stderr.write(f"[oracle.ref] cannot load {cands[0]}: {e}\n")
            mod = None
    _cache[key] = mod
    return mod


def box_iou_rotated():
    """reference box_iou_rotated(boxes1[N,5] f32, boxes2[M,5] f32) -> [N,M] (CPU)"""
    m = _load_pyext("ref_box_iou_rotated")
    return None if m is None else m.box_iou_rotated


def nms_rotated():
    m = _load_pyext("ref_nms_rotated")
    return None if m is None else m.nms_rotated


def ml_nms_rotated():
    m = _load_pyext("ref_ml_nms_rotated")
    return None if m is None else m.ml_nms_rotated


def orn():
    return _load_pyext("ref_orn")


def polyiou():
    """returns f(p8, q8) -> float using the reference's SWIG module, or None"""
    key = ("polyiou",)
    if key in _cache:
        return _cache[key]
    fn = None
    cands = glob.glob(os.path.join(_REF, "polyiou", "_polyiou*.so"))
    if cands:
        try:
            spec = importlib.util.spec_from_file_location("_polyiou", cands[0])
            m = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(m)

            def fn(p8, q8, _m=m):
                vp, vq = _m.new_VectorDouble(), _m.new_VectorDouble()
                try:
                    for v in p8:
                        _m.VectorDouble_push_back(vp, float(v))
                    for v in q8:
                        _m.VectorDouble_push_back(vq, float(v))
                    return _m.iou_poly(vp, vq)
                finally:
                    _m.delete_VectorDouble(vp)
                    _m.delete_VectorDouble(vq)
        except Exception as e:  # pragma: no cover
            sys.stderr.write(f"[oracle.ref] cannot load polyiou: {e}\n")
            fn = None
    _cache[key] = fn
    return fn


def geom_gpubranch():
    """ctypes handle of the reference geometry header host-compiled with __CUDACC__."""
    key = ("gpubranch",)
    if key in _cache:
        return _cache[key]
    L = None
    p = os.path.join(_REF, "ref_geom_gpubranch.so")
    if os.path.exists(p):
        L = ctypes.CDLL(p)
        f32p = ctypes.POINTER(ctypes.c_float)
        L.ref_gpubranch_iou6.restype = ctypes.c_float
        L.ref_gpubranch_iou6.argtypes = [f32p, f32p]
        L.ref_gpubranch_iou6_pairs.restype = None
        L.ref_gpubranch_iou6_pairs.argtypes = [f32p, f32p, ctypes.c_int64, f32p]
    _cache[key] = L
    return L
