"""A small ctypes binding to the HDF5 C library (libhdf5) -- the part of h5py's surface the self-play path needs:
read Keras model files (model.py:147-157 of the reference loads them through keras -> h5py -> libhdf5) and write / read
the sample files of sgfsave.py:49-79.  h5py itself is not installed in this image, but libhdf5 is (/opt/conda/lib), so
files go through the same C library h5py wraps: what this module writes is what h5py would have written, and what
libhdf5 can read here it can read under the reference's train.py.

    with h5lite.File(path, "w") as f:
        f.create_dataset("board", data=np.zeros((1, 19, 19, 17), np.float32))
        g = f.create_group("model_weights"); g.attrs["layer_names"] = [b"conv2d_1", ...]
    with h5lite.File(path) as f:
        a = f["board"][()]; names = f["model_weights"].attrs["layer_names"]; "x" in f; list(f.keys())

Supported: groups, contiguous datasets of float32 / float64 / int8..int64 / uint8..uint64 (read and written whole),
attributes that are numeric scalars / arrays or strings (fixed-length or variable-length, scalar or 1-D array; returned
as bytes / numpy 'S' arrays like h5py does for Keras files).  `available()` says whether the library could be loaded;
nothing here is a fallback for compute -- it is file format plumbing.

Threads: libhdf5 is usually built without thread safety and ctypes releases the GIL around every call, so two threads must
never be inside the library at once.  `File` used as a context manager holds the module-wide lock `LOCK` from open to close
(h5py serialises the same way with its global lock); code that keeps files open outside a `with` block must take LOCK
itself."""
import ctypes as C
import ctypes.util
import glob
import os

import numpy as np

hid_t = C.c_int64
herr_t = C.c_int
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC = 0, 2
H5P_DEFAULT, H5S_ALL, H5S_SCALAR = 0, 0, 0
H5I_GROUP, H5I_DATASET = 2, 5
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_VARIABLE = C.c_size_t(-1).value

_lib = None
_tried = False
import threading
LOCK = threading.RLock()


def _candidates():
    env = os.environ.get("SGO_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            if "_hl" not in p and "_cpp" not in p and "fortran" not in p:
                yield p


def _load():
    """Opens libhdf5 once per process.  Under LOCK, and `_tried` is set LAST: a writer thread that arrives while another one
    is still inside the search waits for its answer instead of reading a half-made "not available"."""
    if _tried:
        return _lib
    with LOCK:
        return _load_locked()


def _load_locked():
    global _lib, _tried
    if _tried:
        return _lib
    for path in _candidates():
        try:
            lib = C.CDLL(path)
            if lib.H5open() < 0:
                continue
        except OSError:
            continue
        sig = {
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Sclose": (herr_t, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aget_space": (hid_t, [hid_t]), "H5Aget_type": (hid_t, [hid_t]),
            "H5Aclose": (herr_t, [hid_t]), "H5Aget_num_attrs": (C.c_int, [hid_t]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]), "H5Tget_size": (C.c_size_t, [hid_t]),
            "H5Tget_class": (C.c_int, [hid_t]), "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tset_cset": (herr_t, [hid_t, C.c_int]),
            "H5Tclose": (herr_t, [hid_t]), "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Gget_num_objs": (herr_t, [hid_t, C.POINTER(hsize_t)]),
            "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)      # errors come back as return codes; no stack dump on stderr
        lib._path = path
        _lib = lib
        break
    _tried = True
    return _lib


def available():
    return _load() is not None


def libversion():
    a, b, c = C.c_uint(), C.c_uint(), C.c_uint()
    _need().H5get_libversion(C.byref(a), C.byref(b), C.byref(c))
    return (a.value, b.value, c.value)


def _need():
    lib = _load()
    if lib is None:
        raise ImportError("libhdf5 was not found (set SGO_HDF5_LIB to its path); neither h5py nor the HDF5 C library is available")
    return lib


def _tid(name):
    return hid_t.in_dll(_need(), name).value


_NATIVE = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int8": "H5T_NATIVE_INT8_g",
           "int16": "H5T_NATIVE_INT16_g", "int32": "H5T_NATIVE_INT32_g", "int64": "H5T_NATIVE_INT64_g",
           "uint8": "H5T_NATIVE_UINT8_g", "uint16": "H5T_NATIVE_UINT16_g", "uint32": "H5T_NATIVE_UINT32_g",
           "uint64": "H5T_NATIVE_UINT64_g"}


def _native_of(dtype):
    key = np.dtype(dtype).name
    if key not in _NATIVE:
        raise TypeError("h5lite: unsupported dtype %s" % key)
    return _tid(_NATIVE[key])


def _dtype_of(lib, tid):
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == H5T_FLOAT:
        return np.dtype("f%d" % size)
    if cls == H5T_INTEGER:
        return np.dtype(("i%d" if lib.H5Tget_sign(tid) else "u%d") % size)
    raise TypeError("h5lite: unsupported HDF5 datatype class %d" % cls)


def _ck(v, what):
    if v < 0:
        raise OSError("h5lite: %s failed" % what)
    return v


def _shape(lib, sid):
    nd = _ck(lib.H5Sget_simple_extent_ndims(sid), "H5Sget_simple_extent_ndims")
    dims = (hsize_t * max(nd, 1))()
    if nd:
        lib.H5Sget_simple_extent_dims(sid, dims, None)
    return tuple(int(d) for d in dims[:nd])


class _Attrs(object):
    def __init__(self, obj):
        self._o = obj

    def __contains__(self, name):
        return _need().H5Aexists(self._o._id, name.encode()) > 0

    def get(self, name, default=None):
        return self[name] if name in self else default

    def __getitem__(self, name):
        lib = _need()
        if name not in self:
            raise KeyError(name)
        aid = _ck(lib.H5Aopen(self._o._id, name.encode(), H5P_DEFAULT), "H5Aopen")
        tid, sid = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
        try:
            shape = _shape(lib, sid)
            n = int(np.prod(shape)) if shape else 1
            if lib.H5Tget_class(tid) == H5T_STRING:
                if lib.H5Tis_variable_str(tid) > 0:
                    # memory type = the attribute's own type (HDF5 does not convert between character sets)
                    buf = (C.c_void_p * n)()
                    _ck(lib.H5Aread(aid, tid, buf), "H5Aread")
                    vals = [C.string_at(v) if v else b"" for v in buf]
                    lib.H5Dvlen_reclaim(tid, sid, H5P_DEFAULT, buf)
                else:
                    size = lib.H5Tget_size(tid)
                    raw = C.create_string_buffer(size * n)
                    _ck(lib.H5Aread(aid, tid, raw), "H5Aread")
                    vals = [raw.raw[i * size:(i + 1) * size].rstrip(b"\x00") for i in range(n)]
                return vals[0] if not shape else np.array(vals, dtype="S").reshape(shape)
            dt = _dtype_of(lib, tid)
            out = np.empty(shape, dtype=dt)
            _ck(lib.H5Aread(aid, _native_of(dt), out.ctypes.data_as(C.c_void_p)), "H5Aread")
            return out[()] if not shape else out
        finally:
            lib.H5Tclose(tid); lib.H5Sclose(sid); lib.H5Aclose(aid)

    def __setitem__(self, name, value):
        """Like h5py: a Python str becomes a variable-length UTF-8 string scalar, bytes a fixed-length one, an array of
        bytes a fixed-length string array, numbers their native type."""
        lib = _need()
        if isinstance(value, str):
            tid = lib.H5Tcopy(_tid("H5T_C_S1_g"))
            lib.H5Tset_size(tid, H5T_VARIABLE)
            lib.H5Tset_cset(tid, 1)                      # H5T_CSET_UTF8
            sid = lib.H5Screate(H5S_SCALAR)
            aid = _ck(lib.H5Acreate2(self._o._id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
            try:
                buf = (C.c_char_p * 1)(value.encode("utf8"))
                _ck(lib.H5Awrite(aid, tid, buf), "H5Awrite")
            finally:
                lib.H5Aclose(aid); lib.H5Sclose(sid); lib.H5Tclose(tid)
            return
        if isinstance(value, bytes):
            arr, shape = np.array([value], dtype="S%d" % max(1, len(value))), ()
        else:
            arr = np.asarray(value)
            if arr.dtype.kind == "U":
                arr = np.char.encode(arr, "utf8")
            shape = arr.shape
            arr = np.require(arr, requirements="C")
        if arr.dtype.kind == "S":
            tid = lib.H5Tcopy(_tid("H5T_C_S1_g"))
            lib.H5Tset_size(tid, arr.dtype.itemsize)
            mem = tid
        else:
            tid = mem = _native_of(arr.dtype)
        if shape:
            dims = (hsize_t * len(shape))(*shape)
            sid = lib.H5Screate_simple(len(shape), dims, None)
        else:
            sid = lib.H5Screate(H5S_SCALAR)
        aid = _ck(lib.H5Acreate2(self._o._id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT), "H5Acreate2")
        try:
            _ck(lib.H5Awrite(aid, mem, arr.ctypes.data_as(C.c_void_p)), "H5Awrite")
        finally:
            lib.H5Aclose(aid); lib.H5Sclose(sid)
            if arr.dtype.kind == "S":
                lib.H5Tclose(tid)


class _Object(object):
    def __init__(self, oid, owner=None):
        self._id, self._owner = oid, owner        # the owner (file) is kept alive while children exist

    @property
    def attrs(self):
        return _Attrs(self)

    def _close(self):
        if self._id:
            _need().H5Oclose(self._id)
            self._id = 0

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass


class Dataset(_Object):
    @property
    def shape(self):
        lib = _need()
        sid = lib.H5Dget_space(self._id)
        try:
            return _shape(lib, sid)
        finally:
            lib.H5Sclose(sid)

    @property
    def dtype(self):
        lib = _need()
        tid = lib.H5Dget_type(self._id)
        try:
            return _dtype_of(lib, tid)
        finally:
            lib.H5Tclose(tid)

    def __getitem__(self, key):
        lib = _need()
        out = np.empty(self.shape, dtype=self.dtype)
        _ck(lib.H5Dread(self._id, _native_of(out.dtype), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)), "H5Dread")
        return out[key]

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return np.asarray(a, dtype=dtype) if dtype is not None else np.asarray(a)


class Group(_Object):
    def __contains__(self, name):
        lib = _need()
        cur = ""
        for part in name.strip("/").split("/"):          # H5Lexists wants every intermediate link to exist
            cur = part if not cur else cur + "/" + part
            if lib.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                return False
        return True

    def __getitem__(self, name):
        lib = _need()
        if name not in self:
            raise KeyError(name)
        oid = _ck(lib.H5Oopen(self._id, name.encode(), H5P_DEFAULT), "H5Oopen")
        kind = lib.H5Iget_type(oid)
        root = self._owner if self._owner is not None else self
        if kind == H5I_GROUP:
            return Group(oid, root)
        if kind == H5I_DATASET:
            return Dataset(oid, root)
        lib.H5Oclose(oid)
        raise TypeError("h5lite: %r is neither a group nor a dataset" % name)

    def keys(self):
        lib = _need()
        n = hsize_t(0)
        _ck(lib.H5Gget_num_objs(self._id, C.byref(n)), "H5Gget_num_objs")
        out = []
        for i in range(n.value):
            size = lib.H5Gget_objname_by_idx(self._id, i, None, 0)
            buf = C.create_string_buffer(size + 1)
            lib.H5Gget_objname_by_idx(self._id, i, buf, size + 1)
            out.append(buf.value.decode())
        return out

    def __iter__(self):
        return iter(self.keys())

    def create_group(self, name):
        lib = _need()
        gid = _ck(lib.H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Gcreate2")
        return Group(gid, self._owner if self._owner is not None else self)

    def require_group(self, name):
        return self[name] if name in self else self.create_group(name)

    def create_dataset(self, name, data=None, dtype=None):
        """Like h5py: a name with '/' creates the intermediate groups (Keras stores `conv2d_1/kernel:0` that way)."""
        lib = _need()
        parts = name.strip("/").split("/")
        parent = self
        for part in parts[:-1]:
            parent = parent.require_group(part)
        arr = np.require(np.asarray(data, dtype=dtype), requirements="C")     # (ascontiguousarray would make a 0-d array 1-d)
        if arr.shape:
            dims = (hsize_t * arr.ndim)(*arr.shape)
            sid = lib.H5Screate_simple(arr.ndim, dims, None)
        else:
            sid = lib.H5Screate(H5S_SCALAR)
        tid = _native_of(arr.dtype)
        did = _ck(lib.H5Dcreate2(parent._id, parts[-1].encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), "H5Dcreate2")
        try:
            _ck(lib.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)), "H5Dwrite")
        finally:
            lib.H5Sclose(sid)
        return Dataset(did, self._owner if self._owner is not None else self)


class File(Group):
    def __init__(self, path, mode="r"):
        lib = _need()
        if mode == "r":
            fid = lib.H5Fopen(os.fsencode(path), H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "w":
            fid = lib.H5Fcreate(os.fsencode(path), H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("h5lite.File: mode must be 'r' or 'w'")
        if fid < 0:
            raise OSError("h5lite: cannot open %r (mode %s): not an HDF5 file, or not accessible" % (path, mode))
        Group.__init__(self, fid, None)
        self.filename = path

    def close(self):
        if self._id:
            _need().H5Fclose(self._id)
            self._id = 0

    _close = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


class _LockedFile(object):
    """`with h5lite.open(path, mode) as f:` -- the library lock is held for the whole session."""

    def __init__(self, path, mode):
        self.path, self.mode, self.f = path, mode, None

    def __enter__(self):
        LOCK.acquire()
        try:
            self.f = File(self.path, self.mode)
        except Exception:
            LOCK.release()
            raise
        return self.f

    def __exit__(self, *exc):
        try:
            self.f.close()
        finally:
            LOCK.release()
        return False


def open(path, mode="r"):
    return _LockedFile(path, mode)
