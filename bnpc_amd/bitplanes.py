"""The data matrix as a bit-plane file.

The reference re-parses its text matrix on every run
(/root/reference/libs/dpmmIO.py:27-98: 250 M entries, 500 MB of text at
config 5) into a float64 array of 2 GB.  What the device needs is 2 bits per
entry: per cell W = ceil(M / 64) pairs of 64-bit words {ones, zeros}
(DESIGN.md section 3).  That form is written ONCE, next to the input, and
memory-mapped afterwards:

    offset  0   8 bytes   magic  b'BNPCBP01'
            8   u64 LE    N      cells (rows of the planes)
           16   u64 LE    M      mutations
           24   u64 LE    W      words per row and plane = ceil(M / 64)
           32   u64 LE    size of the source text file   } a cache file is only
           40   u64 LE    mtime (ns) of the source file  } used for THAT file
           48   u64 LE    1 if the source was transposed on the way in
           56   u64 LE    reserved (0)
           64   N x W x 2 u64 LE   {ones, zeros}: bit b of word w = mutation
                                   64 w + b; neither bit = missing

`BitPlanes` stands in for the float64 matrix wherever the model takes `data`:
`shape`, row gathers `planes[cells]` (float64 with NaN, as the reference's
`self.data[cells]`), and `_lib.Context` uploads the words as they are - no
float64 or int8 matrix of the whole data is built on that path.  Pool workers
are forked, so they share the mapping.
"""
import os
import struct

import numpy as np

MAGIC = b'BNPCBP01'
HEADER = struct.Struct('<8s7Q')
SUFFIX = '.bnpcbits'


class BitPlanes:
    """N x M matrix of 0 | 1 | missing as packed planes (N, W, 2) uint64."""

    def __init__(self, planes, n_muts):
        planes = np.asarray(planes) if not isinstance(planes, np.memmap) \
            else planes
        assert planes.ndim == 3 and planes.shape[2] == 2 \
            and planes.dtype == np.dtype('<u8')
        assert planes.shape[1] == (n_muts + 63) // 64
        self.planes = planes
        self.shape = (planes.shape[0], int(n_muts))
        self.size = self.shape[0] * self.shape[1]
        self.ndim = 2
        self.dtype = np.dtype(np.float64)       # what a row gather returns

    def __len__(self):
        return self.shape[0]

    # -- construction -------------------------------------------------------
    @classmethod
    def from_codes(cls, codes):
        """codes: (N, M) int8 of 0 | 1 | 2 | 3, any strides (a transposed view
        is packed in place, without a copy)."""
        from bnpc_amd import _lib
        codes = np.asarray(codes)
        assert codes.dtype == np.int8 and codes.ndim == 2
        N, M = codes.shape
        planes = np.empty((N, (M + 63) // 64, 2), dtype='<u8')
        _lib.check(_lib.load().bnpc_pack_codes(codes.ctypes.data, N, M,
            codes.strides[0], codes.strides[1], planes.ctypes.data),
            'pack_codes')
        return cls(planes, M)

    @classmethod
    def from_data(cls, data):
        """data: float64 with NaN (the reference's in-memory form)."""
        data = np.asarray(data)
        return cls.from_codes(np.where(np.isnan(data), 3, data)
            .astype(np.int8))

    # -- the matrix, or rows of it ------------------------------------------
    def codes(self, cells=None):
        """int8 codes 0 | 1 | 3 of the rows `cells` (default: all)."""
        from bnpc_amd import _lib
        N, M = self.shape
        planes = np.ascontiguousarray(self.planes)
        if cells is None:
            out = np.empty((N, M), dtype=np.int8)
            ptr, n = None, N
        else:
            cells = _lib.as_i64(np.atleast_1d(cells))
            out = np.empty((cells.size, M), dtype=np.int8)
            ptr, n = cells.ctypes.data, cells.size
        _lib.check(_lib.load().bnpc_unpack_codes(planes.ctypes.data, N, M,
            ptr, n, out.ctypes.data), 'unpack_codes')
        return out

    def to_float64(self, cells=None):
        codes = self.codes(cells)
        out = codes.astype(np.float64)
        out[codes == 3] = np.nan
        return out

    def __getitem__(self, cells):
        """Rows as float64 with NaN - `self.data[cells]` of the reference
        (libs/CRP.py:360, 557-560).  Row indices only, with ndarray
        semantics: integers (negative ones count from the end), integer
        arrays / lists, slices, boolean masks of length N."""
        if isinstance(cells, tuple):
            raise IndexError('BitPlanes takes row indices only')
        N = self.shape[0]
        if isinstance(cells, slice):
            return self.to_float64(np.arange(N)[cells])
        scalar = np.ndim(cells) == 0
        idx = np.atleast_1d(np.asarray(cells))
        if idx.dtype == np.bool_:
            if idx.ndim != 1 or idx.size != N:
                raise IndexError(f'boolean index of length {idx.size} does '
                    f'not match the {N} rows')
            idx = np.flatnonzero(idx)
        elif idx.size == 0:
            idx = idx.astype(np.int64)
        elif idx.dtype.kind not in 'iu' or idx.ndim != 1:
            raise IndexError('BitPlanes rows are indexed by integers, '
                'integer arrays, slices or boolean masks')
        idx = idx.astype(np.int64)
        if idx.size and (idx.min() < -N or idx.max() >= N):
            raise IndexError(f'row index out of range for {N} rows')
        rows = self.to_float64(np.where(idx < 0, idx + N, idx))
        return rows[0] if scalar else rows

    def __array__(self, dtype=None, copy=None):
        out = self.to_float64()
        return out if dtype is None else out.astype(dtype)

    # -- file ---------------------------------------------------------------
    def save(self, path, source=None, transposed=False):
        """Write the bit-plane file (atomically: temporary + rename)."""
        N, M = self.shape
        size = mtime = 0
        if source is not None:
            st = os.stat(source)
            size, mtime = st.st_size, st.st_mtime_ns
        tmp = f'{path}.tmp{os.getpid()}'
        with open(tmp, 'wb') as f:
            f.write(HEADER.pack(MAGIC, N, M, self.planes.shape[1], size, mtime,
                int(bool(transposed)), 0))
            np.ascontiguousarray(self.planes).tofile(f)
        os.replace(tmp, path)
        return path


def load(path, source=None, transposed=None):
    """Memory-map a bit-plane file.  With `source`, the file must have been
    written for that very text file (size and mtime) and orientation;
    ValueError otherwise."""
    with open(path, 'rb') as f:
        head = f.read(HEADER.size)
    if len(head) != HEADER.size:
        raise ValueError(f'{path}: not a bit-plane file')
    magic, N, M, W, size, mtime, was_t, _ = HEADER.unpack(head)
    if magic != MAGIC or W != (M + 63) // 64 or N < 1 or M < 1 \
            or os.path.getsize(path) != HEADER.size + N * W * 16:
        raise ValueError(f'{path}: not a bit-plane file')
    if source is not None:
        st = os.stat(source)
        if (st.st_size, st.st_mtime_ns) != (size, mtime):
            raise ValueError(f'{path} was not written for this {source}')
    if transposed is not None and bool(was_t) != bool(transposed):
        raise ValueError(f'{path} holds the other orientation')
    planes = np.memmap(path, dtype='<u8', mode='r', offset=HEADER.size,
        shape=(N, W, 2))
    return BitPlanes(planes, M)


def load_matrix(in_file, transpose=True, cache=None):
    """The matrix of a text file in the reference's format as BitPlanes: from
    the bit-plane file next to it if that is current, else scanned natively
    (bnpc_parse_matrix), packed and - unless BNPC_BITPLANE_CACHE=0 or the
    directory is read-only - written for the next run."""
    from bnpc_amd import io as bio
    if cache is None:
        cache = os.environ.get('BNPC_BITPLANE_CACHE', '1') != '0'
    side = in_file + SUFFIX
    if cache and os.path.exists(side):
        try:
            return load(side, source=in_file, transposed=transpose)
        except (ValueError, OSError):
            pass        # stale or foreign: rebuilt below
    codes = bio.load_codes_native(in_file, transpose=False)
    planes = BitPlanes.from_codes(codes.T if transpose else codes)
    if cache:
        try:
            planes.save(side, source=in_file, transposed=transpose)
        except OSError:
            pass
    return planes
