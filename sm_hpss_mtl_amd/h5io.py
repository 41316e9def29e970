"""Keras-layout HDF5 weight files through the system's libhdf5, bound with ctypes.

h5py is not installed here, but the HDF5 C library is (conda's libhdf5 1.10): this module writes and reads the file
structure `tensorflow.keras` `Model.save_weights(path.h5)` produces and `load_weights` expects:

    /                      attrs: layer_names (fixed-length byte strings), backend, keras_version
    /<layer>               attrs: weight_names = [b"<weight name>:0", ...]
    /<layer>/<weight name>:0     one float32 dataset per weight (h5py creates the intermediate groups of a name with '/')

A layer is the first component of our canonical tensor names ("tcn/s0_d1/conv/kernel" -> layer "tcn", weight
"tcn/s0_d1/conv/kernel:0"; "S/dense/kernel" -> layer "S").  Files written here open with h5py / HDFView / Keras tooling;
reading a checkpoint written by the reference itself additionally needs the reference's auto-generated layer names
(dense_1, batch_normalization_3, ...) mapped onto ours, which cannot be verified without TensorFlow -- `read_weights`
returns the layers in file order with their names so that such a mapping can be applied on top.
If libhdf5 is not found the callers fall back to `.npz`.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
import sys
from collections import OrderedDict

import numpy as np

_H = None
hid_t, hsize_t, herr_t = C.c_int64, C.c_uint64, C.c_int
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5P_DEFAULT, H5S_ALL = 0, 2, 0, 0
H5T_STR_NULLPAD = 1


def _find():
    cands = []
    name = ctypes.util.find_library("hdf5")
    if name:
        cands.append(name)
    for base in (os.path.join(sys.prefix, "lib"), "/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial"):
        cands += sorted(glob.glob(os.path.join(base, "libhdf5.so*")))
    env = os.environ.get("SMH_LIBHDF5")
    if env:
        cands.insert(0, env)
    for c in cands:
        try:
            return C.CDLL(c)
        except OSError:
            continue
    return None


def lib():
    """The loaded libhdf5 with argument types set, or None."""
    global _H
    if _H is None:
        h = _find()
        if h is None:
            _H = False
            return None
        sig = {
            "H5open": (herr_t, []),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Gclose": (herr_t, [hid_t]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (herr_t, [hid_t, C.c_uint]), "H5Pclose": (herr_t, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sget_simple_extent_npoints": (C.c_int64, [hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tset_strpad": (herr_t, [hid_t, C.c_int]),
            "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aclose": (herr_t, [hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
        }
        for n, (res, args) in sig.items():
            f = getattr(h, n)
            f.restype, f.argtypes = res, args
        h.H5open()
        h.T_F32 = hid_t.in_dll(h, "H5T_NATIVE_FLOAT_g").value
        h.T_F32LE = hid_t.in_dll(h, "H5T_IEEE_F32LE_g").value
        h.T_C_S1 = hid_t.in_dll(h, "H5T_C_S1_g").value
        h.P_LINK_CREATE = hid_t.in_dll(h, "H5P_CLS_LINK_CREATE_ID_g").value
        _H = h
    return _H or None


def available():
    return lib() is not None


def _ok(v, what):
    if v < 0:
        raise IOError("libhdf5: %s failed" % what)
    return v


def _write_str_attr(h, loc, name, values):
    """values: bytes (scalar attribute) or list of bytes (1-D array), stored as fixed-length null-padded strings like h5py does
    for numpy bytes_ data."""
    scalar = isinstance(values, (bytes, bytearray))
    vals = [bytes(values)] if scalar else [bytes(v) for v in values]
    size = max(1, max((len(v) for v in vals), default=1))
    t = _ok(h.H5Tcopy(h.T_C_S1), "H5Tcopy")
    h.H5Tset_size(t, size)
    h.H5Tset_strpad(t, H5T_STR_NULLPAD)
    if scalar:
        sp = _ok(h.H5Screate(0), "H5Screate")  # H5S_SCALAR
    else:
        dims = (hsize_t * 1)(len(vals))
        sp = _ok(h.H5Screate_simple(1, dims, None), "H5Screate_simple")
    buf = b"".join(v.ljust(size, b"\0") for v in vals) or b"\0"
    a = _ok(h.H5Acreate2(loc, name.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2 " + name)
    _ok(h.H5Awrite(a, t, C.c_char_p(buf)), "H5Awrite " + name)
    h.H5Aclose(a), h.H5Sclose(sp), h.H5Tclose(t)


def _read_str_attr(h, loc, name):
    if h.H5Aexists(loc, name.encode()) <= 0:
        return None
    a = _ok(h.H5Aopen(loc, name.encode(), H5P_DEFAULT), "H5Aopen " + name)
    t, sp = h.H5Aget_type(a), h.H5Aget_space(a)
    size, n = h.H5Tget_size(t), h.H5Sget_simple_extent_npoints(sp)
    ndims = h.H5Sget_simple_extent_ndims(sp)
    buf = C.create_string_buffer(max(1, size * n))
    _ok(h.H5Aread(a, t, buf), "H5Aread " + name)
    h.H5Tclose(t), h.H5Sclose(sp), h.H5Aclose(a)
    vals = [buf.raw[i * size:(i + 1) * size].rstrip(b"\0") for i in range(n)]
    return vals[0] if ndims == 0 else vals


def split_name(tensor_name):
    """canonical tensor name -> (layer, Keras weight name)."""
    return tensor_name.split("/", 1)[0], tensor_name + ":0"


def write_weights(path, weights, backend=b"tensorflow", keras_version=b"2.4.0"):
    """weights: ordered mapping canonical tensor name -> float32 array.  Writes a Keras-layout weight file."""
    layers = OrderedDict()
    for name, arr in weights.items():
        layer, wname = split_name(name)
        layers.setdefault(layer, OrderedDict())[wname] = np.ascontiguousarray(arr, dtype=np.float32)
    write_layers(path, layers, backend, keras_version)


def write_layers(path, layers, backend=b"tensorflow", keras_version=b"2.4.0"):
    """layers: ordered mapping layer name -> ordered mapping Keras weight name ('<layer>/kernel:0', ...) -> array; layers
    without weights (activations, dropout, ...) carry an empty mapping, as in the files Keras writes."""
    h = lib()
    if h is None:
        raise RuntimeError("libhdf5 not found (set SMH_LIBHDF5); use the .npz format")
    f = _ok(h.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT), "H5Fcreate " + str(path))
    try:
        lcpl = _ok(h.H5Pcreate(h.P_LINK_CREATE), "H5Pcreate")
        h.H5Pset_create_intermediate_group(lcpl, 1)
        _write_str_attr(h, f, "layer_names", [l.encode() for l in layers])
        _write_str_attr(h, f, "backend", backend)
        _write_str_attr(h, f, "keras_version", keras_version)
        for layer, ws in layers.items():
            g = _ok(h.H5Gcreate2(f, layer.encode(), lcpl, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2 " + layer)
            _write_str_attr(h, g, "weight_names", [w.encode() for w in ws])
            for wname, arr in ws.items():
                arr = np.ascontiguousarray(arr, dtype=np.float32)
                dims = (hsize_t * max(arr.ndim, 1))(*(arr.shape if arr.ndim else (1,)))
                sp = _ok(h.H5Screate_simple(max(arr.ndim, 1), dims, None), "H5Screate_simple")
                d = _ok(h.H5Dcreate2(g, wname.encode(), h.T_F32LE, sp, lcpl, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2 " + wname)
                if arr.size:
                    _ok(h.H5Dwrite(d, h.T_F32, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite " + wname)
                h.H5Dclose(d), h.H5Sclose(sp)
            h.H5Gclose(g)
        h.H5Pclose(lcpl)
    finally:
        h.H5Fclose(f)


def read_weights(path):
    """-> (OrderedDict layer -> OrderedDict weight name (without ':0') -> float32 array, dict of root attributes),
    layers and weights in file order (the order Keras' load_weights consumes them in)."""
    h = lib()
    if h is None:
        raise RuntimeError("libhdf5 not found (set SMH_LIBHDF5); use the .npz format")
    f = h.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
    if f < 0:
        raise IOError("cannot open %s as HDF5" % path)
    out = OrderedDict()
    try:
        layer_names = _read_str_attr(h, f, "layer_names")
        if layer_names is None:
            raise IOError("%s has no 'layer_names' attribute: not a Keras weight file" % path)
        attrs = {"backend": _read_str_attr(h, f, "backend"), "keras_version": _read_str_attr(h, f, "keras_version")}
        for ln in layer_names:
            g = _ok(h.H5Gopen2(f, ln, H5P_DEFAULT), "H5Gopen2 " + ln.decode())
            ws = OrderedDict()
            for wn in _read_str_attr(h, g, "weight_names") or []:
                d = _ok(h.H5Dopen2(g, wn, H5P_DEFAULT), "H5Dopen2 " + wn.decode())
                sp = h.H5Dget_space(d)
                nd = h.H5Sget_simple_extent_ndims(sp)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    h.H5Sget_simple_extent_dims(sp, dims, None)
                shape = tuple(int(x) for x in dims[:nd])
                arr = np.empty(shape, np.float32)
                if arr.size:
                    _ok(h.H5Dread(d, h.T_F32, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)), "H5Dread " + wn.decode())
                h.H5Sclose(sp), h.H5Dclose(d)
                name = wn.decode()
                ws[name[:-2] if name.endswith(":0") else name] = arr
            h.H5Gclose(g)
            out[ln.decode()] = ws
    finally:
        h.H5Fclose(f)
    return out, attrs
