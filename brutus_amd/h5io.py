"""Minimal HDF5 reader/writer over the libhdf5 C API (ctypes).

`h5py` is not available in this image, but libhdf5 (1.10) is.  This module
covers what the fit() path needs:

  * `ResultsFile` -- writes `{save_file}.h5` in the reference's layout
    (reference fitting.py:1632-1662 dataset names/dtypes/fill values, rows
    written per object as in fitting.py:1734-1748; `"w-"` semantics: refuse to
    overwrite);
  * `read_dataset` / `list_datasets` -- read back plain and compound datasets
    (used by the tests and to load input catalogues such as the reference's
    `demos/Orion_l204.7_b-19.2.h5`).
"""
import ctypes as C
import ctypes.util
import os

import numpy as np

__all__ = ["ResultsFile", "read_dataset", "list_datasets", "write_datasets",
           "hdf5_available"]

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5P_DEFAULT, H5S_ALL, H5S_SELECT_SET = 0, 0, 0
(H5T_INTEGER, H5T_FLOAT, H5T_STRING, H5T_COMPOUND, H5T_ENUM,
 H5T_ARRAY) = 0, 1, 3, 6, 8, 10

_h5 = None


def _candidates():
    env = os.environ.get("BRUTUS_AMD_HDF5_LIB")
    if env:
        yield env
    for p in ("/opt/conda/lib/libhdf5.so", "/opt/conda/lib/libhdf5.so.103",
              "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so",
              "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"):
        yield p
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found


def _lib():
    global _h5
    if _h5 is not None:
        return _h5
    err = None
    for cand in _candidates():
        try:
            L = C.CDLL(cand)
            break
        except OSError as e:  # try the next location
            err = e
    else:
        raise OSError("libhdf5 not found (set BRUTUS_AMD_HDF5_LIB): %s" % err)
    sig = {
        "H5open": (herr_t, []),
        "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (herr_t, [hid_t]),
        "H5Fflush": (herr_t, [hid_t, C.c_int]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (herr_t, [hid_t]),
        "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t),
                                         C.POINTER(hsize_t), C.POINTER(hsize_t),
                                         C.POINTER(hsize_t)]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t),
                                                C.POINTER(hsize_t)]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dclose": (herr_t, [hid_t]),
        "H5Tcreate": (hid_t, [C.c_int, C.c_size_t]),
        "H5Tinsert": (herr_t, [hid_t, C.c_char_p, C.c_size_t, hid_t]),
        "H5Tarray_create2": (hid_t, [hid_t, C.c_uint, C.POINTER(hsize_t)]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
        "H5Tclose": (herr_t, [hid_t]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tget_nmembers": (C.c_int, [hid_t]),
        "H5Tget_member_name": (C.c_void_p, [hid_t, C.c_uint]),
        "H5Tget_member_offset": (C.c_size_t, [hid_t, C.c_uint]),
        "H5Tget_member_type": (hid_t, [hid_t, C.c_uint]),
        "H5Tget_array_ndims": (C.c_int, [hid_t]),
        "H5Tget_array_dims2": (C.c_int, [hid_t, C.POINTER(hsize_t)]),
        "H5Tget_super": (hid_t, [hid_t]),
        "H5free_memory": (herr_t, [C.c_void_p]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_obj_track_times": (herr_t, [hid_t, C.c_uint]),
        "H5Pset_virtual": (herr_t, [hid_t, hid_t, C.c_char_p, C.c_char_p, hid_t]),
        "H5Sselect_all": (herr_t, [hid_t]),
        "H5Pclose": (herr_t, [hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gclose": (herr_t, [hid_t]),
        "H5Gget_num_objs": (herr_t, [hid_t, C.POINTER(hsize_t)]),
        "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
        "H5Gget_objtype_by_idx": (C.c_int, [hid_t, hsize_t]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    L.H5open()
    L.H5Eset_auto2(0, None, None)   # we raise Python exceptions instead
    # dataset creation properties: no object timestamps (h5py's default `track_times=False`,
    # i.e. what the reference's files look like; it also makes two runs byte-identical)
    L._dcpl = L.H5Pcreate(hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value)
    if L._dcpl < 0 or L.H5Pset_obj_track_times(L._dcpl, 0) < 0:
        L._dcpl = H5P_DEFAULT
    _h5 = L
    return L


def hdf5_available():
    try:
        _lib()
        return True
    except OSError:
        return False


def _native(name):
    return hid_t.in_dll(_lib(), name).value


_NP2H5 = {"f4": "H5T_NATIVE_FLOAT_g", "f8": "H5T_NATIVE_DOUBLE_g",
          "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
          "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g",
          "u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g",
          "u4": "H5T_NATIVE_UINT32_g", "u8": "H5T_NATIVE_UINT64_g",
          "b1": "H5T_NATIVE_INT8_g"}


def _h5type(dt):
    """numpy dtype -> (hid, must_close).  Structured dtypes become compounds,
    sub-array fields become H5T_ARRAY, 'S' becomes fixed-length strings."""
    L = _lib()
    dt = np.dtype(dt)
    if dt.names:
        tid = L.H5Tcreate(H5T_COMPOUND, dt.itemsize)
        for name in dt.names:
            sub, off = dt.fields[name][:2]
            mid, close = _h5type(sub)
            L.H5Tinsert(tid, name.encode(), off, mid)
            if close:
                L.H5Tclose(mid)
        return tid, True
    if dt.subdtype is not None:
        base, shape = dt.subdtype
        bid, close = _h5type(base)
        dims = (hsize_t * len(shape))(*shape)
        tid = L.H5Tarray_create2(bid, len(shape), dims)
        if close:
            L.H5Tclose(bid)
        return tid, True
    if dt.kind == "S":
        tid = L.H5Tcopy(_native("H5T_C_S1_g"))
        L.H5Tset_size(tid, max(1, dt.itemsize))
        return tid, True
    key = dt.kind + str(dt.itemsize)
    if key not in _NP2H5:
        raise TypeError("unsupported dtype for HDF5: %r" % dt)
    return _native(_NP2H5[key]), False


def _nptype(tid):
    """HDF5 datatype -> numpy dtype (native byte order)."""
    L = _lib()
    cls = L.H5Tget_class(tid)
    size = L.H5Tget_size(tid)
    if cls == H5T_INTEGER:
        return np.dtype(("i" if L.H5Tget_sign(tid) else "u") + str(size))
    if cls == H5T_FLOAT:
        return np.dtype("f" + str(size))
    if cls == H5T_STRING:
        return np.dtype("S" + str(size))
    if cls == H5T_ENUM:
        sup = L.H5Tget_super(tid)
        out = _nptype(sup)
        L.H5Tclose(sup)
        return out
    if cls == H5T_ARRAY:
        nd = L.H5Tget_array_ndims(tid)
        dims = (hsize_t * nd)()
        L.H5Tget_array_dims2(tid, dims)
        sup = L.H5Tget_super(tid)
        base = _nptype(sup)
        L.H5Tclose(sup)
        return np.dtype((base, tuple(int(d) for d in dims)))
    if cls == H5T_COMPOUND:
        names, formats, offsets = [], [], []
        for i in range(L.H5Tget_nmembers(tid)):
            p = L.H5Tget_member_name(tid, i)
            names.append(C.string_at(p).decode())
            L.H5free_memory(p)
            mt = L.H5Tget_member_type(tid, i)
            formats.append(_nptype(mt))
            L.H5Tclose(mt)
            offsets.append(L.H5Tget_member_offset(tid, i))
        return np.dtype(dict(names=names, formats=formats, offsets=offsets,
                             itemsize=size))
    raise TypeError("unsupported HDF5 type class %d" % cls)


def _check(h, what):
    if h < 0:
        raise OSError("HDF5 error in %s" % what)
    return h


class _File(object):
    def __init__(self, path, mode):
        L = _lib()
        self.L = L
        b = os.fsencode(path)
        if mode == "r":
            self.fid = L.H5Fopen(b, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode == "r+":
            self.fid = L.H5Fopen(b, H5F_ACC_RDWR, H5P_DEFAULT)
        elif mode == "w-":
            if os.path.exists(path):
                raise OSError("Unable to create file (file exists): %s" % path)
            self.fid = L.H5Fcreate(b, H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        elif mode == "w":
            self.fid = L.H5Fcreate(b, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError(mode)
        if self.fid < 0:
            raise OSError("Unable to open %s (mode %s)" % (path, mode))
        self._dsets = {}

    def create_dataset(self, name, data):
        L = self.L
        data = np.ascontiguousarray(data)
        tid, close = _h5type(data.dtype)
        shape = data.shape if data.ndim else (1,)
        dims = (hsize_t * len(shape))(*shape)
        sid = _check(L.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")
        did = _check(L.H5Dcreate2(self.fid, name.encode(), tid, sid, H5P_DEFAULT,
                                  L._dcpl, H5P_DEFAULT), "H5Dcreate2 " + name)
        if data.size:
            _check(L.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                              data.ctypes.data_as(C.c_void_p)), "H5Dwrite " + name)
        L.H5Sclose(sid)
        self._dsets[name] = (did, tid, close, tuple(shape), data.dtype)

    def create_filled(self, name, shape, dtype, fill, block_rows=4096):
        """Create a dataset holding `fill` everywhere without materialising it in RAM
        (written in blocks of rows)."""
        L = self.L
        dt = np.dtype(dtype)
        tid, close = _h5type(dt)
        dims = (hsize_t * len(shape))(*shape)
        sid = _check(L.H5Screate_simple(len(shape), dims, None), "H5Screate_simple")
        did = _check(L.H5Dcreate2(self.fid, name.encode(), tid, sid, H5P_DEFAULT,
                                  L._dcpl, H5P_DEFAULT), "H5Dcreate2 " + name)
        L.H5Sclose(sid)
        self._dsets[name] = (did, tid, close, tuple(shape), dt)
        blk = np.full((min(block_rows, shape[0]),) + tuple(shape[1:]), fill, dtype=dt)
        for a in range(0, shape[0], block_rows):
            self.write_rows(name, a, blk[:min(block_rows, shape[0] - a)])

    def bind_dataset(self, name, read_first_column=False):
        """Bind an existing dataset for `write_rows` without reading it; optionally
        return its first column (axis 1 index 0) only."""
        L = self.L
        did = _check(L.H5Dopen2(self.fid, name.encode(), H5P_DEFAULT),
                     "H5Dopen2 " + name)
        ftid = L.H5Dget_type(did)
        dt = _nptype(ftid)
        L.H5Tclose(ftid)
        sid = L.H5Dget_space(did)
        nd = L.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sid, dims, None)
        shape = tuple(int(d) for d in dims[:nd])
        tid, close = _h5type(dt)
        self._dsets[name] = (did, tid, close, shape, dt)
        col = None
        if read_first_column:
            st = (hsize_t * nd)(*([0] * nd))
            cnt = (hsize_t * nd)(*([shape[0]] + [1] * (nd - 1)))
            _check(L.H5Sselect_hyperslab(sid, H5S_SELECT_SET, st, None, cnt, None),
                   "H5Sselect_hyperslab")
            msp = L.H5Screate_simple(nd, cnt, None)
            col = np.empty((shape[0],) + (1,) * (nd - 1), dtype=dt)
            if col.size:
                _check(L.H5Dread(did, tid, msp, sid, H5P_DEFAULT,
                                 col.ctypes.data_as(C.c_void_p)), "H5Dread " + name)
            L.H5Sclose(msp)
            col = col.reshape(shape[0])
        L.H5Sclose(sid)
        return shape, col

    def open_dataset(self, name):
        """Bind an existing dataset for `write_rows`; returns its contents."""
        L = self.L
        did = _check(L.H5Dopen2(self.fid, name.encode(), H5P_DEFAULT),
                     "H5Dopen2 " + name)
        ftid = L.H5Dget_type(did)
        dt = _nptype(ftid)
        L.H5Tclose(ftid)
        sid = L.H5Dget_space(did)
        nd = L.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sid, dims, None)
        L.H5Sclose(sid)
        shape = tuple(int(d) for d in dims[:nd])
        tid, close = _h5type(dt)
        out = np.empty(shape, dtype=dt)
        if out.size:
            _check(L.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                             out.ctypes.data_as(C.c_void_p)), "H5Dread " + name)
        self._dsets[name] = (did, tid, close, shape, dt)
        return out

    def write_rows(self, name, start, rows):
        """Overwrite rows [start, start+len(rows)) along axis 0."""
        L = self.L
        did, tid, _, shape, dt = self._dsets[name]
        rows = np.ascontiguousarray(rows, dtype=dt)
        n = rows.shape[0]
        if n == 0:
            return
        fsp = L.H5Dget_space(did)
        nd = len(shape)
        st = (hsize_t * nd)(*([start] + [0] * (nd - 1)))
        cnt = (hsize_t * nd)(*([n] + list(shape[1:])))
        _check(L.H5Sselect_hyperslab(fsp, H5S_SELECT_SET, st, None, cnt, None),
               "H5Sselect_hyperslab")
        msp = L.H5Screate_simple(nd, cnt, None)
        _check(L.H5Dwrite(did, tid, msp, fsp, H5P_DEFAULT,
                          rows.ctypes.data_as(C.c_void_p)), "H5Dwrite rows " + name)
        L.H5Sclose(msp)
        L.H5Sclose(fsp)

    def read_rows(self, name, start, n):
        """Rows [start, start + n) along axis 0 of a bound dataset."""
        L = self.L
        did, tid, _, shape, dt = self._dsets[name]
        nd = len(shape)
        out = np.empty((n,) + tuple(shape[1:]), dtype=dt)
        if out.size == 0:
            return out
        fsp = L.H5Dget_space(did)
        st = (hsize_t * nd)(*([start] + [0] * (nd - 1)))
        cnt = (hsize_t * nd)(*([n] + list(shape[1:])))
        _check(L.H5Sselect_hyperslab(fsp, H5S_SELECT_SET, st, None, cnt, None), "H5Sselect_hyperslab")
        msp = L.H5Screate_simple(nd, cnt, None)
        _check(L.H5Dread(did, tid, msp, fsp, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)),
               "H5Dread rows " + name)
        L.H5Sclose(msp)
        L.H5Sclose(fsp)
        return out

    def flush(self):
        self.L.H5Fflush(self.fid, 1)

    def close(self):
        L = self.L
        for did, tid, close, _, _ in self._dsets.values():
            L.H5Dclose(did)
            if close:
                L.H5Tclose(tid)
        self._dsets = {}
        if self.fid is not None and self.fid >= 0:
            L.H5Fclose(self.fid)
        self.fid = None


def write_datasets(path, arrays, mode="w"):
    """Write a dict of arrays as root-level datasets (test/utility helper)."""
    f = _File(path, mode)
    try:
        for k, v in arrays.items():
            f.create_dataset(k, v)
    finally:
        f.close()


def list_datasets(path, group="/"):
    L = _lib()
    f = _File(path, "r")
    try:
        gid = _check(L.H5Gopen2(f.fid, group.encode(), H5P_DEFAULT), "H5Gopen2")
        n = hsize_t()
        L.H5Gget_num_objs(gid, C.byref(n))
        names = []
        for i in range(n.value):
            buf = C.create_string_buffer(1024)
            L.H5Gget_objname_by_idx(gid, i, buf, 1024)
            names.append(buf.value.decode())
        L.H5Gclose(gid)
        return names
    finally:
        f.close()


def read_dataset(path, name):
    """Read a whole dataset into a numpy array (compound -> structured)."""
    L = _lib()
    f = _File(path, "r")
    try:
        did = _check(L.H5Dopen2(f.fid, name.encode(), H5P_DEFAULT),
                     "H5Dopen2 " + name)
        ftid = L.H5Dget_type(did)
        dt = _nptype(ftid)
        sid = L.H5Dget_space(did)
        nd = L.H5Sget_simple_extent_ndims(sid)
        dims = (hsize_t * max(nd, 1))()
        if nd > 0:
            L.H5Sget_simple_extent_dims(sid, dims, None)
        shape = tuple(int(d) for d in dims[:nd])
        out = np.empty(shape, dtype=dt)
        mtid, close = _h5type(dt)
        if out.size:
            _check(L.H5Dread(did, mtid, H5S_ALL, H5S_ALL, H5P_DEFAULT,
                             out.ctypes.data_as(C.c_void_p)), "H5Dread " + name)
        if close:
            L.H5Tclose(mtid)
        L.H5Sclose(sid)
        L.H5Tclose(ftid)
        L.H5Dclose(did)
        return out
    finally:
        f.close()


def write_virtual_index(path, Ndata, Ndraws, save_dar_draws, data_labels, parts, overwrite=False):
    """`{save_file}.h5` of a `fit_sharded(writer="per_rank")` run: the reference's datasets
    (fitting.py:1635-1662) as HDF5 VIRTUAL datasets whose row range `[lo, hi)` maps onto the
    dataset of the same name in part file `name` -- `parts` = [(lo, hi, file name relative to
    the index)], together covering `[0, Ndata)`.  Nothing is copied: a reader (h5py,
    `read_dataset`) opens this file and libhdf5 fetches the rows from the parts, which have to
    stay beside it.  `labels` is small and written for real."""
    L = _lib()
    self = ResultsFile.__new__(ResultsFile)
    self.Ndraws, self.save_dar_draws = Ndraws, save_dar_draws
    layout = self._layout()
    f = _File(path, "w" if overwrite else "w-")
    try:
        if data_labels is not None:
            f.create_dataset("labels", np.asarray(data_labels))
        for k, (rs, dt, fill, _) in layout.items():
            shape = (Ndata,) + tuple(rs)
            nd = len(shape)
            dcpl = _check(L.H5Pcreate(hid_t.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value), "H5Pcreate")
            L.H5Pset_obj_track_times(dcpl, 0)
            vsp = _check(L.H5Screate_simple(nd, (hsize_t * nd)(*shape), None), "H5Screate_simple")
            for lo, hi, name in parts:
                st = (hsize_t * nd)(*([lo] + [0] * (nd - 1)))
                cnt = (hsize_t * nd)(*([hi - lo] + list(shape[1:])))
                _check(L.H5Sselect_hyperslab(vsp, H5S_SELECT_SET, st, None, cnt, None), "H5Sselect_hyperslab")
                ssp = _check(L.H5Screate_simple(nd, cnt, None), "H5Screate_simple")
                _check(L.H5Pset_virtual(dcpl, vsp, name.encode(), k.encode(), ssp), "H5Pset_virtual " + k)
                L.H5Sclose(ssp)
            L.H5Sselect_all(vsp)
            tid, close = _h5type(np.dtype(dt))
            did = _check(L.H5Dcreate2(f.fid, k.encode(), tid, vsp, H5P_DEFAULT, dcpl, H5P_DEFAULT),
                         "H5Dcreate2 (virtual) " + k)
            L.H5Dclose(did)
            if close:
                L.H5Tclose(tid)
            L.H5Sclose(vsp)
            L.H5Pclose(dcpl)
    finally:
        f.close()


def materialize(index_path, out_path, block_rows=4096):
    """Copy a results file -- in particular the virtual index of a per-rank run -- into ONE plain
    HDF5 file with the same datasets (for moving a result without its part files)."""
    src = _File(index_path, "r")
    dst = _File(out_path, "w-")
    try:
        for k in list_datasets(index_path):
            shape, _ = src.bind_dataset(k)
            did, tid, _, _, dt = src._dsets[k]
            if k == "labels" or len(shape) == 0:
                dst.create_dataset(k, read_dataset(index_path, k))
                continue
            dst.create_filled(k, shape, dt, 0, block_rows=block_rows)
            for a in range(0, shape[0], block_rows):
                n = min(block_rows, shape[0] - a)
                dst.write_rows(k, a, src.read_rows(k, a, n))
    finally:
        src.close()
        dst.close()


class ResultsFile(object):
    """The fit() output file, reference layout (fitting.py:1632-1662).

    `running_io=True` (default): the datasets are created on disk with the
    reference's fill values (model_idx = -99 ...), finished rows are staged in RAM
    and written every `flush_every` objects (and on close).  Memory is bounded by the
    staging -- never by the catalogue -- and rows may arrive in any order (sharded
    runs, resumed runs); an interrupted run keeps everything up to the last flush,
    which is the purpose of the reference's `running_io`, without one tiny HDF5 write
    per dataset per object.  `running_io=False`: everything is kept in RAM and
    written once at the end, like the reference (fitting.py:1784-1798).
    """

    SPEC = (("model_idx", (), "int32", -99), ("ml_scale", (), "float32", 1),
            ("ml_av", (), "float32", 0), ("ml_rv", (), "float32", 0),
            ("ml_cov_sar", (3, 3), "float32", 0), ("obj_log_post", (), "float32", 0))
    SPEC1 = (("obj_log_evid", "float32", 0), ("obj_chi2min", "float32", 0),
             ("obj_Nbands", "int16", 0))
    DAR = ("samps_dist", "samps_red", "samps_dred", "samps_logp")

    def _layout(self):
        """name -> (row shape, dtype, fill, position in the yielded tuple)"""
        nd = (self.Ndraws,)
        lay = {"model_idx": (nd, "int32", -99, 0), "ml_scale": (nd, "float32", 1, 1),
               "ml_av": (nd, "float32", 0, 2), "ml_rv": (nd, "float32", 0, 3),
               "ml_cov_sar": (nd + (3, 3), "float32", 0, 4),
               "obj_Nbands": ((), "int16", 0, 5), "obj_log_post": (nd, "float32", 0, 6),
               "obj_log_evid": ((), "float32", 0, 7), "obj_chi2min": ((), "float32", 0, 8)}
        if self.save_dar_draws:
            for j, k in enumerate(self.DAR):
                lay[k] = (nd, "float32", 1, 9 + j)
        return lay

    @classmethod
    def row_dtype(cls, Ndraws, save_dar_draws):
        """(structured dtype of ONE object's row -- every dataset's slice, in the file's own
        dtypes --, [(dataset, position in the yielded tuple)]): the packed form in which
        `parallel.fit_sharded` ships finished rows between ranks."""
        self = cls.__new__(cls)
        self.Ndraws, self.save_dar_draws = Ndraws, save_dar_draws
        lay = self._layout()
        return (np.dtype([(k, dt, rs) for k, (rs, dt, _, _) in lay.items()]),
                [(k, pos) for k, (_, _, _, pos) in lay.items()])

    def __init__(self, path, Ndata, Ndraws, data_labels, save_dar_draws,
                 running_io=True, flush_every=256, async_io=True):
        self.file = _File(path, "w-")
        self.Ndata, self.Ndraws = Ndata, Ndraws
        self.running_io = running_io
        self.flush_every = max(1, int(flush_every))
        self.save_dar_draws = save_dar_draws
        self.layout = self._layout()
        self.arrays = None
        if data_labels is not None:
            self.file.create_dataset("labels", np.asarray(data_labels))
        if running_io:
            for k, (rs, dt, fill, _) in self.layout.items():
                self.file.create_filled(k, (Ndata,) + rs, dt, fill)
            self.file.flush()
            self._start_writer(async_io)
        else:
            self.arrays = {k: np.full((Ndata,) + rs, fill, dtype=dt)
                           for k, (rs, dt, fill, _) in self.layout.items()}

    # -- staging ring + writer thread ------------------------------------------------------
    # Finished rows are written IN PLACE into one of a few preallocated blocks (one array of
    # `flush_every` rows per dataset, in the file's dtypes: nothing is allocated per row);
    # a full block is handed to a background thread that turns it into hyperslab writes
    # (libhdf5 is the thread-safe build and ctypes drops the GIL around its calls) while the
    # caller fills the next one.  The caller only ever waits when ALL blocks are in flight,
    # i.e. when the disk is slower than the fit.
    NBLOCKS = 3

    class _Block(object):
        def __init__(self, layout, cap):
            self.arr = {k: np.empty((cap,) + rs, dtype=dt) for k, (rs, dt, _, _) in layout.items()}
            self.rows = np.empty(cap, dtype=np.int64)
            self.n = 0

    def _start_writer(self, async_io=True):
        import queue
        import threading
        self._err = None
        self.dropped = []               # [lo, hi) row ranges lost to a writer failure
        self._cur = None
        self._free = queue.Queue()
        self._full = queue.Queue(maxsize=self.NBLOCKS + 4)    # back-pressure on write_block
        self._nblk = 0                  # blocks are allocated when first needed
        self._async = bool(async_io)
        self._thread = None
        if self._async:
            self._thread = threading.Thread(target=self._writer_loop, name="brutus-h5-writer",
                                            daemon=True)
            self._thread.start()

    def _writer_loop(self):
        while True:
            job = self._full.get()
            if job is None:
                return
            try:
                if self._err is None:
                    self._write_job(job)
                else:                           # after a failure nothing more is written:
                    self._note_dropped(job)     # the row ranges are reported with the error
            except BaseException as e:          # surfaces at the next write_row / flush / close
                self._err = e
                self._note_dropped(job)
            finally:
                if isinstance(job, ResultsFile._Block):
                    job.n = 0
                    self._free.put(job)
                self._full.task_done()

    def _note_dropped(self, job):
        if isinstance(job, ResultsFile._Block):
            rows = np.sort(job.rows[:job.n])
            if rows.size:
                cuts = np.flatnonzero(rows[1:] != rows[:-1] + 1) + 1
                for a, b in zip(np.r_[0, cuts], np.r_[cuts, rows.size]):
                    self.dropped.append((int(rows[a]), int(rows[b - 1]) + 1))
        else:
            start, blocks = job
            self.dropped.append((int(start), int(start) + len(next(iter(blocks.values())))))

    #: `resume()` decides from this dataset alone whether a row was fitted (the reference's
    #: -99 sentinel, fitting.py:1635), so it is written LAST, after every other dataset of
    #: the same rows has been flushed: a run killed between two datasets leaves rows whose
    #: sentinel is still in place, never rows that look finished and are not.
    SENTINEL = "model_idx"

    def _write_job(self, job):
        names = [k for k in self.layout if k != self.SENTINEL]
        for part in (names, [self.SENTINEL]):
            if isinstance(job, ResultsFile._Block):
                n = job.n
                rows = job.rows[:n]
                if n and np.all(rows[1:] == rows[:-1] + 1):      # the usual case: one ascending run
                    for k in part:
                        self.file.write_rows(k, int(rows[0]), job.arr[k][:n])
                elif n:
                    order = np.argsort(rows, kind="stable")
                    srt = rows[order]
                    cuts = np.flatnonzero(srt[1:] != srt[:-1] + 1) + 1
                    for a, b in zip(np.r_[0, cuts], np.r_[cuts, n]):
                        for k in part:
                            self.file.write_rows(k, int(srt[a]), job.arr[k][order[a:b]])
            else:                                   # (first row, {dataset: rows}) from write_block
                start, blocks = job
                for k in part:
                    self.file.write_rows(k, start, blocks[k])
            self.file.flush()

    def _check_err(self):
        """A failure of the writer is sticky: every later `write_row` / `write_block` /
        `flush` / `close` raises it again (with the row ranges that never reached the
        file), until the file is closed -- a caller that caught the first one cannot end
        up with a file that has holes and a clean exit."""
        e = getattr(self, "_err", None)
        if e is not None:
            if self.dropped and not getattr(e, "_brutus_noted", False):
                try:
                    e.args = (e.args + ("results rows not written: %s"
                                        % ", ".join("[%d, %d)" % r for r in self.dropped[:8])
                                        + (" ..." if len(self.dropped) > 8 else ""),))
                    e._brutus_noted = True
                except Exception:
                    pass
            raise e

    def _next_block(self):
        if self._nblk < self.NBLOCKS and self._free.empty():
            self._nblk += 1
            return ResultsFile._Block(self.layout, self.flush_every)
        return self._free.get()                 # waits for the writer only when all are in flight

    def _submit(self, job):
        if self._async:
            self._full.put(job)
        else:
            try:
                if self._err is None:
                    self._write_job(job)
                else:
                    self._note_dropped(job)
            except BaseException as e:
                self._err = e
                self._note_dropped(job)
                raise
            finally:
                if isinstance(job, ResultsFile._Block):
                    job.n = 0
                    self._free.put(job)

    @classmethod
    def resume(cls, path, Ndata, Ndraws, save_dar_draws, flush_every=256):
        """Re-open an interrupted `running_io=True` results file.  Rows whose
        `model_idx` still holds the sentinel -99 (reference fitting.py:1635)
        were never fitted; `todo` lists them."""
        self = cls.__new__(cls)
        self.file = _File(path, "r+")
        self.Ndata, self.Ndraws = Ndata, Ndraws
        self.running_io, self.flush_every = True, max(1, int(flush_every))
        self.save_dar_draws = save_dar_draws
        self.layout = self._layout()
        self.arrays = None
        self._start_writer(True)
        try:
            first = None
            for k in self.layout:
                shape, col = self.file.bind_dataset(k, read_first_column=(k == "model_idx"))
                if k == "model_idx":
                    first = col
                    if shape != (Ndata, Ndraws):
                        raise ValueError("existing results file has shape %r, expected %r"
                                         % (shape, (Ndata, Ndraws)))
        except Exception:
            self.file.close()
            raise
        self.todo = np.where(first == -99)[0]
        return self

    def write_row(self, i, results):
        """Row `i` <- one tuple of `BruteForce._fit` (mapping of reference
        fitting.py:1735-1748)."""
        if self.arrays is not None:            # running_io=False: everything in RAM
            with np.errstate(over="ignore"):
                for k, (_, _, _, pos) in self.layout.items():
                    self.arrays[k][i] = results[pos]
            return
        self._check_err()
        blk = self._cur
        if blk is None:
            blk = self._cur = self._next_block()
        n = blk.n
        with np.errstate(over="ignore"):   # -1e300 (out-of-bounds draw) -> -inf in f32, as h5py does
            for k, (_, _, _, pos) in self.layout.items():
                blk.arr[k][n] = results[pos]
        blk.rows[n] = i
        blk.n = n + 1
        if blk.n >= self.flush_every:
            self._cur = None
            self._submit(blk)

    def write_block(self, start, blocks):
        """Rows `start ..` <- `blocks[name]` (n, ...) arrays already in the layout's shapes
        (any float / int dtype; e.g. the packed rows a rank hands over in `fit_sharded`).
        The arrays belong to the writer from here on."""
        if self.arrays is not None:
            with np.errstate(over="ignore"):
                for k in self.layout:
                    self.arrays[k][start:start + len(blocks[k])] = blocks[k]
            return
        self._check_err()
        self._submit((int(start), blocks))

    def flush(self):
        """Hand the partly filled block over and wait until everything is on disk."""
        if self.arrays is not None:
            return
        if self._cur is not None and self._cur.n:
            blk, self._cur = self._cur, None
            self._submit(blk)
        if self._async:
            self._full.join()
        self._check_err()

    def close(self):
        if self.file is None:
            return
        try:
            if self.arrays is None:
                try:
                    self.flush()
                finally:
                    if self._thread is not None:
                        self._full.put(None)
                        self._thread.join()
                        self._thread = None
            else:
                for k, v in self.arrays.items():
                    self.file.create_dataset(k, v)
        finally:
            self.file.close()
            self.file = None
