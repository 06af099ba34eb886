"""Voxel-grid cache wire format of the reference: one HDF5 file per sample, dataset `voxel_grid`, chunked, compressed
with the HDF5 Blosc filter (id 32001) in `blosc:zstd`, level 1, byte shuffle.

Reference call sites:
  data/utils/generic.py:35-46   _blosc_opts(complevel=1, complib='blosc:zstd', shuffle='byte') -> compression=32001,
                                compression_opts=(0, 0, 0, 0, complevel, shuffle, complib_index)
  data/utils/generic.py:49-55   np_array_to_h5(array, outpath): h5py.File(outpath, 'w').create_dataset('voxel_grid', data=array, ...)
  data/utils/generic.py:58-68   h5_to_np_array(inpath): np.asarray(h5f['voxel_grid']); None (and a message) on OSError
  data/dsec/subsequence/base.py:97-98,205-217    directory `voxel_grids_v{version}_100ms_forward_{num_bins}_bins`, file `{index:06d}.h5`,
                                load if the file exists, else construct and save
  data/multiflow2d/sample.py:100-103             file `voxel_grid_v{version}_{num_bins_total}_bins[_downsampled].h5`

The arithmetic behind that boundary lives in third-party code that is NOT under /root/reference and not importable from the image's Python:
libhdf5 (through h5py; the reference pins no version: environment created from `conda install h5py blosc-hdf5-plugin`, README.md:24-29),
the hdf5-blosc filter plugin (filter revision 2) and c-blosc 1.x with its bundled Zstandard.  This module restates the published
formats:
  * HDF5 File Format Specification 1.x/2.0: superblock version 0/1, version-1 object headers (+ continuation blocks), old-style groups
    (symbol-table message, version-1 B-tree node type 0, SNOD, local heap), dataspace v1/v2, datatype classes 0/1 (fixed / floating
    point), data layout message v3 (compact / contiguous / chunked with a version-1 B-tree node type 1), filter pipeline v1/v2 --
    i.e. what `h5py.File(path, 'w')` writes with its default `libver` -- and reads filters deflate (1), shuffle (2) and Blosc (32001);
  * Blosc 1 frame: 16-byte header (version, versionlz, flags, typesize, nbytes, blocksize, cbytes), `bstarts`, per block either one
    stream or `typesize` split streams, each prefixed by its int32 compressed size (== the raw size: stored), byte un-shuffle per block;
  * Zstandard frames through the system's libzstd.so.1 (ctypes); LZ4 blocks through liblz4.so.1; zlib through Python's zlib.
PARITY PINNED in both directions (tests/test_voxel_cache.py): the system Python of the build image has neither h5py nor blosc, but its
conda interpreter has h5py 3.3.0 / HDF5 1.10.6 and PyTables 3.6.1, which registers the Blosc filter (c-blosc 1.20.1) with libhdf5.
tests/golden/make_voxel_cache_golden.py runs the REFERENCE's own `np_array_to_h5` there and commits the files it writes
(tests/golden/voxel_cache/*.h5): this reader decodes them bit-exactly (also a full-size 15x480x640 file with libhdf5's own two-level
chunk B-tree, checked by hand); and files written by this module are read back by h5py on the real libhdf5 (a test that runs
wherever that interpreter exists).  On top of that the reader is checked against byte streams assembled by hand from the
specifications above (layouts neither writer produces).  Host-side I/O only: nothing here is on the GPU hot path.
"""
from __future__ import annotations

import ctypes
import ctypes.util
import os
import struct
import zlib
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

DATASET_NAME = "voxel_grid"                     # generic.py:54,64
BLOSC_FILTER_ID = 32001                         # generic.py:40
BLOSC_COMPRESSORS = ["blosclz", "lz4", "lz4hc", "snappy", "zlib", "zstd"]   # generic.py:37 (index = cd_values[6])
UNDEF = 0xFFFFFFFFFFFFFFFF
HDF5_SIGNATURE = b"\x89HDF\r\n\x1a\n"


class VoxelCacheError(OSError):
    """A file this reader cannot decode (h5py raises OSError for unreadable files; generic.py:66 catches exactly that)."""


# ------------------------------------------------------------------------------------------------------------------ file naming
def dsec_voxel_grid_dir(ev_dir: Union[str, Path], num_bins: int, extended_voxel_grid: bool = True) -> Path:
    """base.py:93-98: version 1 = boundary-aware window (loads a few events "of the future"), version 0 = strictly causal."""
    return Path(ev_dir) / f"voxel_grids_v{1 if extended_voxel_grid else 0}_100ms_forward_{num_bins}_bins"


def dsec_voxel_grid_file(voxel_grid_dir: Union[str, Path], file_index: int) -> Path:
    assert file_index >= 0                                                      # base.py:206
    return Path(voxel_grid_dir) / (f"{file_index}".zfill(6) + ".h5")            # base.py:209


def multiflow_voxel_grid_file(ev_dir: Union[str, Path], num_bins_total: int, extended_voxel_grid: bool = True, downsample: bool = False) -> Path:
    return Path(ev_dir) / f"voxel_grid_v{1 if extended_voxel_grid else 0}_{num_bins_total}_bins{'_downsampled' if downsample else ''}.h5"   # sample.py:100-102


# ------------------------------------------------------------------------------------------------------------------ codecs
_zstd = None
_lz4 = None


def _libzstd():
    global _zstd
    if _zstd is None:
        name = ctypes.util.find_library("zstd") or "libzstd.so.1"
        try:
            L = ctypes.CDLL(name)
        except OSError as e:
            raise VoxelCacheError(f"libzstd is needed for the blosc:zstd voxel-grid cache and could not be loaded ({e})")
        for f, res, args in (("ZSTD_decompress", ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]),
                             ("ZSTD_compress", ctypes.c_size_t, [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]),
                             ("ZSTD_compressBound", ctypes.c_size_t, [ctypes.c_size_t]), ("ZSTD_isError", ctypes.c_uint, [ctypes.c_size_t])):
            getattr(L, f).restype, getattr(L, f).argtypes = res, args
        _zstd = L
    return _zstd


def zstd_decompress(src: bytes, nbytes: int) -> bytes:
    L = _libzstd()
    dst = ctypes.create_string_buffer(max(nbytes, 1))
    n = L.ZSTD_decompress(dst, nbytes, src, len(src))
    if L.ZSTD_isError(n) or n != nbytes:
        raise VoxelCacheError(f"zstd: corrupt stream (expected {nbytes} bytes)")
    return dst.raw[:nbytes]


def zstd_compress(src: bytes, level: int = 1) -> bytes:
    L = _libzstd()
    cap = L.ZSTD_compressBound(len(src))
    dst = ctypes.create_string_buffer(cap)
    n = L.ZSTD_compress(dst, cap, src, len(src), level)
    if L.ZSTD_isError(n):
        raise VoxelCacheError("zstd: compression failed")
    return dst.raw[:n]


def _lz4_decompress(src: bytes, nbytes: int) -> bytes:
    global _lz4
    if _lz4 is None:
        try:
            _lz4 = ctypes.CDLL(ctypes.util.find_library("lz4") or "liblz4.so.1")
        except OSError as e:
            raise VoxelCacheError(f"liblz4 is needed for this blosc:lz4 file and could not be loaded ({e})")
        _lz4.LZ4_decompress_safe.restype = ctypes.c_int
        _lz4.LZ4_decompress_safe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    dst = ctypes.create_string_buffer(max(nbytes, 1))
    n = _lz4.LZ4_decompress_safe(src, dst, len(src), nbytes)
    if n != nbytes:
        raise VoxelCacheError("lz4: corrupt block")
    return dst.raw[:nbytes]


def byte_shuffle(buf: bytes, typesize: int) -> bytes:
    """Blosc / HDF5 shuffle: byte j of every element gathered into plane j; the bytes after the last whole element are copied."""
    n = len(buf) // typesize
    a = np.frombuffer(buf, dtype=np.uint8)
    return a[:n * typesize].reshape(n, typesize).T.tobytes() + a[n * typesize:].tobytes()


def byte_unshuffle(buf: bytes, typesize: int) -> bytes:
    n = len(buf) // typesize
    a = np.frombuffer(buf, dtype=np.uint8)
    return a[:n * typesize].reshape(typesize, n).T.tobytes() + a[n * typesize:].tobytes()


# Blosc 1 frame --------------------------------------------------------------------------------------------------------------
BLOSC_VERSION_FORMAT = 2
BLOSC_MIN_BUFFERSIZE = 128       # a block is only split into per-byte streams when every stream is at least this long
BLOSC_MAX_SPLITS = 16
_F_SHUFFLE, _F_MEMCPYED, _F_BITSHUFFLE, _F_DONTSPLIT = 0x1, 0x2, 0x4, 0x10


def blosc_decompress(frame: bytes) -> bytes:
    """c-blosc 1.x `blosc_decompress`: header, bstarts, blocks (blosc.c blosc_d / the README_HEADER.rst layout)."""
    if len(frame) < 16:
        raise VoxelCacheError("blosc: frame shorter than its header")
    version, versionlz, flags, typesize = frame[0], frame[1], frame[2], frame[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", frame, 4)
    if version != BLOSC_VERSION_FORMAT:
        raise VoxelCacheError(f"blosc: frame format version {version} (this reader knows {BLOSC_VERSION_FORMAT})")
    if cbytes > len(frame):
        raise VoxelCacheError("blosc: truncated frame")
    if nbytes == 0:
        return b""
    if flags & _F_MEMCPYED:
        return frame[16:16 + nbytes]
    if flags & _F_BITSHUFFLE:
        raise VoxelCacheError("blosc: bit-shuffled frames are not supported (the reference writes shuffle='byte', generic.py:50-55)")
    comp = flags >> 5
    if comp == 4:
        inflate = zstd_decompress
    elif comp == 1:
        inflate = _lz4_decompress
    elif comp == 3:
        inflate = lambda s, n: zlib.decompress(s, bufsize=n)
    else:
        raise VoxelCacheError(f"blosc: compressor format {comp} ({['blosclz', 'lz4', 'snappy', 'zlib', 'zstd'][comp] if comp < 5 else '?'}) is not supported")
    typesize = max(typesize, 1)
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", frame, 16)
    out = bytearray()
    for bi in range(nblocks):
        bsize = min(blocksize, nbytes - bi * blocksize)
        leftover = bsize != blocksize
        split = (not flags & _F_DONTSPLIT) and typesize <= BLOSC_MAX_SPLITS and blocksize // typesize >= BLOSC_MIN_BUFFERSIZE and not leftover
        nsplits = typesize if split else 1
        neblock = bsize // nsplits
        pos = bstarts[bi]
        block = bytearray()
        for _ in range(nsplits):
            (cb,) = struct.unpack_from("<i", frame, pos)
            pos += 4
            if cb < 0 or pos + cb > len(frame):
                raise VoxelCacheError("blosc: corrupt block table")
            block += frame[pos:pos + cb] if cb == neblock else inflate(frame[pos:pos + cb], neblock)
            pos += cb
        out += byte_unshuffle(bytes(block), typesize) if (flags & _F_SHUFFLE) and typesize > 1 else block
    return bytes(out)


def blosc_compress(data: bytes, typesize: int, clevel: int = 1, shuffle: bool = True, blocksize: int = 0, split: Optional[bool] = None) -> bytes:
    """A valid Blosc 1 frame with zstd streams.  `split=None`: one stream per block (flag 0x10, what c-blosc >= 1.15 emits for zstd);
    True reproduces the per-byte-plane streams of older c-blosc builds."""
    nbytes = len(data)
    if blocksize <= 0:
        blocksize = min(max(nbytes, 1), 1 << 18)
    blocksize = max(typesize, blocksize // typesize * typesize)
    flags = (4 << 5) | (_F_SHUFFLE if shuffle and typesize > 1 else 0) | (0 if split else _F_DONTSPLIT)
    nblocks = (nbytes + blocksize - 1) // blocksize
    body = bytearray()
    bstarts = []
    for bi in range(nblocks):
        blk = data[bi * blocksize:(bi + 1) * blocksize]
        bstarts.append(16 + 4 * nblocks + len(body))
        if flags & _F_SHUFFLE:
            blk = byte_shuffle(blk, typesize)
        do_split = bool(split) and typesize <= BLOSC_MAX_SPLITS and blocksize // typesize >= BLOSC_MIN_BUFFERSIZE and len(blk) == blocksize
        ns = typesize if do_split else 1
        ne = len(blk) // ns
        for k in range(ns):
            raw = blk[k * ne:(k + 1) * ne]
            c = zstd_compress(raw, clevel)
            if len(c) >= len(raw):
                c = raw                                              # stored: compressed size == raw size
            body += struct.pack("<i", len(c)) + c
    if 16 + 4 * nblocks + len(body) >= 16 + nbytes and nbytes:      # incompressible: the whole buffer is stored
        return struct.pack("<BBBBIII", BLOSC_VERSION_FORMAT, 1, (flags & ~_F_DONTSPLIT) | _F_MEMCPYED, typesize, nbytes, blocksize, 16 + nbytes) + data
    cbytes = 16 + 4 * nblocks + len(body)
    return struct.pack("<BBBBIII", BLOSC_VERSION_FORMAT, 1, flags, typesize, nbytes, blocksize, cbytes) + struct.pack(f"<{nblocks}i", *bstarts) + bytes(body)


# ------------------------------------------------------------------------------------------------------------------ HDF5 reader
def _dtype_from_message(msg: bytes) -> np.dtype:
    cls, ver = msg[0] & 0x0F, msg[0] >> 4
    bits0 = msg[1]
    (size,) = struct.unpack_from("<I", msg, 4)
    order = ">" if bits0 & 1 else "<"
    if ver not in (1, 2, 3):
        raise VoxelCacheError(f"HDF5: datatype message version {ver}")
    if cls == 0:
        return np.dtype(f"{order}{'i' if bits0 & 0x08 else 'u'}{size}")
    if cls == 1:
        if size not in (2, 4, 8):
            raise VoxelCacheError(f"HDF5: {size}-byte floating point type")
        return np.dtype(f"{order}f{size}")
    raise VoxelCacheError(f"HDF5: datatype class {cls} (only fixed- and floating-point datasets are cached by the reference)")


def _dataspace_from_message(msg: bytes) -> Tuple[int, ...]:
    ver, rank, flags = msg[0], msg[1], msg[2]
    if ver == 1:
        off = 8
    elif ver == 2:
        off = 4
        if msg[3] == 0:
            return ()
        if msg[3] == 2:
            raise VoxelCacheError("HDF5: null dataspace")
    else:
        raise VoxelCacheError(f"HDF5: dataspace message version {ver}")
    return tuple(struct.unpack_from(f"<{rank}Q", msg, off))


class _H5File:
    def __init__(self, buf: bytes):
        self.b = buf
        if buf[:8] != HDF5_SIGNATURE:
            raise VoxelCacheError("not an HDF5 file (signature)")
        ver = buf[8]
        if ver not in (0, 1):
            raise VoxelCacheError(f"HDF5: superblock version {ver}: only the layout h5py writes by default (version 0/1, old-style groups) is read")
        if buf[13] != 8 or buf[14] != 8:
            raise VoxelCacheError("HDF5: offsets / lengths are not 8 bytes")
        off = 24 if ver == 0 else 28            # v1 inserts the indexed-storage K (2) + 2 reserved bytes before the addresses
        self.base, _free, self.eof, _drv = struct.unpack_from("<4Q", buf, off)
        off += 32
        # root group symbol table entry: name offset, object header address, cache type, reserved, scratch pad
        _name, self.root_oh, _cache = struct.unpack_from("<QQI", buf, off)
        if self.eof > len(buf) + self.base:
            raise VoxelCacheError("HDF5: file is truncated (end-of-file address beyond the file)")

    # ---- object headers (version 1) ------------------------------------------------------------------------------------
    def messages(self, addr: int) -> List[Tuple[int, bytes]]:
        b = self.b
        addr += self.base
        if b[addr:addr + 4] == b"OHDR":
            raise VoxelCacheError("HDF5: version-2 object header (file written with libver='latest'): not supported")
        if b[addr] != 1:
            raise VoxelCacheError(f"HDF5: object header version {b[addr]}")
        nmsg, _ref, size = struct.unpack_from("<HII", b, addr + 2)
        blocks = [(addr + 16, size)]
        out = []
        while blocks and len(out) < nmsg:
            pos, left = blocks.pop(0)
            end = pos + left
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize, _mflags = struct.unpack_from("<HHB", b, pos)
                data = b[pos + 8:pos + 8 + msize]
                pos += 8 + msize
                if mtype == 0x0010:                                   # continuation: (address, length) of the next block
                    caddr, clen = struct.unpack_from("<QQ", data, 0)
                    blocks.append((caddr + self.base, clen))
                out.append((mtype, data))
        return out

    # ---- old-style group lookup ----------------------------------------------------------------------------------------
    def _heap_string(self, heap_addr: int, offset: int) -> str:
        b = self.b
        heap_addr += self.base
        if b[heap_addr:heap_addr + 4] != b"HEAP":
            raise VoxelCacheError("HDF5: local heap signature")
        _size, _free, data_addr = struct.unpack_from("<QQQ", b, heap_addr + 8)
        s = data_addr + self.base + offset
        return b[s:b.index(b"\0", s)].decode("utf-8")

    def _group_entries(self, btree: int, heap: int) -> Dict[str, int]:
        b = self.b
        out: Dict[str, int] = {}
        node = btree + self.base
        if b[node:node + 4] == b"SNOD":
            (nsym,) = struct.unpack_from("<H", b, node + 6)
            for i in range(nsym):
                noff, oh = struct.unpack_from("<QQ", b, node + 8 + 40 * i)
                out[self._heap_string(heap, noff)] = oh
            return out
        if b[node:node + 4] != b"TREE" or b[node + 4] != 0:
            raise VoxelCacheError("HDF5: group B-tree node signature")
        (used,) = struct.unpack_from("<H", b, node + 6)
        for i in range(used):                                           # key0 child0 key1 child1 ... : keys and children are 8 bytes each
            (child,) = struct.unpack_from("<Q", b, node + 24 + 8 + 16 * i)
            out.update(self._group_entries(child, heap))
        return out

    def dataset_header(self, name: str) -> int:
        for mtype, data in self.messages(self.root_oh):
            if mtype == 0x0011:
                btree, heap = struct.unpack_from("<QQ", data, 0)
                entries = self._group_entries(btree, heap)
                if name not in entries:
                    raise KeyError(f"no dataset {name!r} in the root group (found {sorted(entries)})")
                return entries[name]
        raise VoxelCacheError("HDF5: the root group has no symbol table (new-style groups, libver='latest'): not supported")

    # ---- chunk index -----------------------------------------------------------------------------------------------------
    def chunks(self, btree: int, rank1: int):
        """Yields (offsets[rank1], address, stored size, filter mask) of a version-1 B-tree of node type 1."""
        b = self.b
        node = btree + self.base
        if b[node:node + 4] != b"TREE" or b[node + 4] != 1:
            raise VoxelCacheError("HDF5: chunk B-tree node signature")
        level = b[node + 5]
        (used,) = struct.unpack_from("<H", b, node + 6)
        ksize = 8 + 8 * rank1
        pos = node + 24
        for _ in range(used):
            csize, mask = struct.unpack_from("<II", b, pos)
            offs = struct.unpack_from(f"<{rank1}Q", b, pos + 8)
            (child,) = struct.unpack_from("<Q", b, pos + ksize)
            pos += ksize + 8
            if level:
                yield from self.chunks(child, rank1)
            else:
                yield offs, child, csize, mask

    # ---- dataset ---------------------------------------------------------------------------------------------------------
    def read_dataset(self, name: str) -> np.ndarray:
        msgs = self.messages(self.dataset_header(name))
        get = lambda t: next((d for mt, d in msgs if mt == t), None)
        sp, dt, lay, flt = get(0x0001), get(0x0003), get(0x0008), get(0x000B)
        if sp is None or dt is None or lay is None:
            raise VoxelCacheError("HDF5: dataset header lacks a dataspace / datatype / layout message")
        shape, dtype = _dataspace_from_message(sp), _dtype_from_message(dt)
        filters = _parse_filters(flt) if flt is not None else []
        if lay[0] != 3:
            raise VoxelCacheError(f"HDF5: data layout message version {lay[0]} (h5py's default writes version 3)")
        cls = lay[1]
        n = int(np.prod(shape, dtype=np.int64)) if shape else 1
        if cls == 0:                                                         # compact
            (sz,) = struct.unpack_from("<H", lay, 2)
            raw = lay[4:4 + sz]
            return np.frombuffer(raw, dtype=dtype, count=n).reshape(shape).astype(dtype.newbyteorder("="))
        if cls == 1:                                                         # contiguous
            addr, sz = struct.unpack_from("<QQ", lay, 2)
            if addr == UNDEF:
                return np.zeros(shape, dtype=dtype.newbyteorder("="))
            return np.frombuffer(self.b, dtype=dtype, count=n, offset=addr + self.base).reshape(shape).astype(dtype.newbyteorder("="))
        if cls != 2:
            raise VoxelCacheError(f"HDF5: layout class {cls}")
        rank1 = lay[2]
        (btree,) = struct.unpack_from("<Q", lay, 3)
        cdims = struct.unpack_from(f"<{rank1}I", lay, 11)
        if rank1 != len(shape) + 1 or cdims[-1] != dtype.itemsize:
            raise VoxelCacheError("HDF5: chunk dimensionality does not match the dataspace")
        cshape = cdims[:-1]
        out = np.zeros(shape, dtype=dtype.newbyteorder("="))
        if btree == UNDEF:
            return out
        cbytes = int(np.prod(cshape, dtype=np.int64)) * dtype.itemsize
        for offs, addr, csize, mask in self.chunks(btree, rank1):
            raw = self.b[addr + self.base:addr + self.base + csize]
            if len(raw) != csize:
                raise VoxelCacheError("HDF5: chunk beyond the end of the file")
            for k in range(len(filters) - 1, -1, -1):                        # the pipeline is undone last filter first
                if not (mask >> k) & 1:
                    raw = filters[k](raw, cbytes, dtype.itemsize)
            if len(raw) != cbytes:
                raise VoxelCacheError(f"HDF5: chunk decodes to {len(raw)} bytes, expected {cbytes}")
            chunk = np.frombuffer(raw, dtype=dtype).reshape(cshape)
            sl_out = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs[:-1], cshape, shape))
            sl_in = tuple(slice(0, s.stop - s.start) for s in sl_out)
            out[sl_out] = chunk[sl_in]
        return out


def _parse_filters(msg: bytes):
    ver, nf = msg[0], msg[1]
    if ver not in (1, 2):
        raise VoxelCacheError(f"HDF5: filter pipeline message version {ver}")
    pos = 8 if ver == 1 else 2
    out = []
    for _ in range(nf):
        (fid,) = struct.unpack_from("<H", msg, pos)
        pos += 2
        nlen = 0
        if ver == 1 or fid >= 256:
            (nlen,) = struct.unpack_from("<H", msg, pos)
            pos += 2
        _flags, ncd = struct.unpack_from("<HH", msg, pos)
        pos += 4
        pos += (nlen + 7) // 8 * 8 if ver == 1 else nlen
        cd = struct.unpack_from(f"<{ncd}I", msg, pos)
        pos += 4 * ncd
        if ver == 1 and ncd % 2:
            pos += 4
        if fid == 1:
            out.append(lambda raw, n, ts: zlib.decompress(raw))
        elif fid == 2:
            out.append(lambda raw, n, ts, k=(cd[0] if cd else 0): byte_unshuffle(raw, k or ts))
        elif fid == BLOSC_FILTER_ID:
            out.append(lambda raw, n, ts: blosc_decompress(raw))
        else:
            raise VoxelCacheError(f"HDF5: filter {fid} is not supported (deflate, shuffle and Blosc 32001 are)")
    return out


def read_h5_dataset(path: Union[str, Path], name: str = DATASET_NAME) -> np.ndarray:
    with open(path, "rb") as f:
        buf = f.read()
    try:
        return _H5File(buf).read_dataset(name)
    except (struct.error, IndexError, ValueError, zlib.error, OverflowError, MemoryError, RecursionError) as e:   # a corrupt structure: h5py reports OSError
        raise VoxelCacheError(f"{path}: damaged HDF5 structure ({type(e).__name__}: {e})")


def h5_to_np_array(inpath: Union[str, Path]) -> Optional[np.ndarray]:
    """generic.py:58-68: the array, or None (with the reference's message) when the file cannot be read."""
    inpath = Path(inpath)
    assert inpath.suffix == ".h5"
    assert inpath.exists()
    try:
        return read_h5_dataset(inpath)
    except OSError:
        print(f"Error loading {inpath}")
    return None


# ------------------------------------------------------------------------------------------------------------------ HDF5 writer
def _msg(mtype: int, data: bytes, flags: int = 0) -> bytes:
    data = data + b"\0" * (-len(data) % 8)
    return struct.pack("<HHB3x", mtype, len(data), flags) + data


def _object_header(msgs: Sequence[bytes]) -> bytes:
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body


def _datatype_message(dtype: np.dtype) -> bytes:
    dtype = np.dtype(dtype)
    be = 1 if dtype.byteorder == ">" else 0
    if dtype.kind == "f":
        props = {2: (0, 16, 10, 5, 0, 10, 15), 4: (0, 32, 23, 8, 0, 23, 127), 8: (0, 64, 52, 11, 0, 52, 1023)}[dtype.itemsize]
        # class 1 version 1; bit field: byte order | mantissa normalisation 2 (implied leading 1) in bits 4-5; sign bit position
        return struct.pack("<BBBBI", 0x11, 0x20 | be, dtype.itemsize * 8 - 1, 0, dtype.itemsize) + struct.pack("<HHBBBBI", *props)
    if dtype.kind in "iu":
        return struct.pack("<BBBBI", 0x10, be | (0x08 if dtype.kind == "i" else 0), 0, 0, dtype.itemsize) + struct.pack("<HH", 0, dtype.itemsize * 8)
    raise TypeError(f"dtype {dtype} cannot be cached (fixed- and floating-point arrays only)")


def default_chunks(shape: Sequence[int], itemsize: int, target: int = 1 << 20) -> Tuple[int, ...]:
    """Chunk = whole trailing planes, leading dimensions halved until a chunk holds at most ~1 MiB (any chunking is valid HDF5; h5py
    would guess its own).  A (C, H, W) fp32 voxel grid becomes one (1, H, W) chunk per time bin at DSEC size."""
    chunks = [max(1, int(s)) for s in shape]
    for d in range(len(chunks)):
        while chunks[d] > 1 and int(np.prod(chunks, dtype=np.int64)) * itemsize > target:
            chunks[d] = (chunks[d] + 1) // 2
    return tuple(chunks)


def _chunk_btree(entries: List[Tuple[Tuple[int, ...], int, int]], rank1: int, end_key: Tuple[int, ...], base_addr: int, K: int = 32) -> Tuple[bytes, int]:
    """Version-1 B-tree (node type 1) over sorted (offsets, address, size) entries, at most 2K per node; returns (bytes, root address)."""
    ksize = 8 + 8 * rank1
    node_size = 24 + 2 * K * 8 + (2 * K + 1) * ksize
    level_items = [(offs, addr, size) for offs, addr, size in entries]      # level 0 children are the chunks
    blob = bytearray()
    level = 0
    while True:
        groups = [level_items[i:i + 2 * K] for i in range(0, len(level_items), 2 * K)]
        next_items = []
        addrs = [base_addr + len(blob) + i * node_size for i in range(len(groups))]
        for gi, grp in enumerate(groups):
            node = bytearray(struct.pack("<4sBBHQQ", b"TREE", 1, level, len(grp), addrs[gi - 1] if gi else UNDEF, addrs[gi + 1] if gi + 1 < len(groups) else UNDEF))
            for offs, addr, size in grp:
                node += struct.pack("<II", size, 0) + struct.pack(f"<{rank1}Q", *offs) + struct.pack("<Q", addr)
            last = groups[gi + 1][0][0] if gi + 1 < len(groups) else end_key
            node += struct.pack("<II", 0, 0) + struct.pack(f"<{rank1}Q", *last)
            node += b"\0" * (node_size - len(node))
            blob += node
            next_items.append((grp[0][0], addrs[gi], grp[0][2]))
        if len(groups) == 1:
            return bytes(blob), addrs[0]
        level_items, level = next_items, level + 1


def write_h5_dataset(path: Union[str, Path], array: np.ndarray, name: str = DATASET_NAME, chunks: Optional[Sequence[int]] = None,
                     complevel: int = 1, shuffle: bool = True, split: Optional[bool] = None) -> None:
    """One dataset in the root group, chunked, filter 32001 with cd_values (2, 2, typesize, chunk bytes, complevel, shuffle, 5 = zstd):
    the file `np_array_to_h5` produces in the reference, structure for structure (superblock 0, old-style root group)."""
    array = np.ascontiguousarray(array)
    if array.ndim == 0:
        raise ValueError("scalar datasets are not chunked")
    dtype, shape, rank = array.dtype, array.shape, array.ndim
    chunks = tuple(int(c) for c in (chunks or default_chunks(shape, dtype.itemsize)))
    assert len(chunks) == rank and all(c >= 1 for c in chunks)
    rank1 = rank + 1
    cbytes = int(np.prod(chunks, dtype=np.int64)) * dtype.itemsize
    grid = [(s + c - 1) // c for s, c in zip(shape, chunks)]
    # ---- chunk payloads (edge chunks are stored full size, zero filled)
    payloads, offsets = [], []
    for idx in np.ndindex(*grid):
        offs = tuple(i * c for i, c in zip(idx, chunks))
        blk = np.zeros(chunks, dtype=dtype)
        src = array[tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunks, shape))]
        blk[tuple(slice(0, n) for n in src.shape)] = src
        payloads.append(blosc_compress(blk.tobytes(), dtype.itemsize, complevel, shuffle, split=split))
        offsets.append(offs + (0,))
    # ---- fixed part: superblock (96) | root object header (40) | local heap (32 + 40) | group B-tree (544) | SNOD (328) | dataset header
    A_ROOT, A_HEAP = 96, 136
    A_HEAPDATA = A_HEAP + 32
    heap_data = b"\0" * 8 + name.encode() + b"\0"
    heap_data += b"\0" * (-len(heap_data) % 8)
    free_off = len(heap_data)
    heap_data += struct.pack("<QQ", 1, 16)                                       # one free block: next = 1 (none), size 16
    A_GTREE = A_HEAPDATA + len(heap_data)
    GK, LK = 16, 4
    gtree_size = 24 + (2 * GK + 1) * 8 + 2 * GK * 8
    A_SNOD = A_GTREE + gtree_size
    snod_size = 8 + 2 * LK * 40
    A_DSET = A_SNOD + snod_size
    layout_placeholder = struct.pack("<BBBQ", 3, 2, rank1, 0) + struct.pack(f"<{rank1}I", *(chunks + (dtype.itemsize,)))
    cd = (2, 2, dtype.itemsize, cbytes, complevel, 1 if shuffle else 0, BLOSC_COMPRESSORS.index("zstd"))
    fname = b"blosc\0\0\0"
    filt = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", BLOSC_FILTER_ID, len(fname), 1, len(cd)) + fname + struct.pack(f"<{len(cd)}I", *cd) + b"\0" * 4

    def dataset_header(btree_addr: int) -> bytes:
        lay = struct.pack("<BBB", 3, 2, rank1) + struct.pack("<Q", btree_addr) + struct.pack(f"<{rank1}I", *(chunks + (dtype.itemsize,)))
        return _object_header([
            _msg(0x0001, struct.pack("<BBB5x", 1, rank, 0) + struct.pack(f"<{rank}Q", *shape)),
            _msg(0x0003, _datatype_message(dtype), flags=1),
            _msg(0x0005, struct.pack("<BBBB", 2, 3, 2, 0)),                       # fill value v2: incremental allocation, write if set, undefined
            _msg(0x000B, filt),
            _msg(0x0008, lay),
        ])

    dset_size = len(dataset_header(0))
    A_CTREE = A_DSET + dset_size
    end_key = (grid[0] * chunks[0],) + (0,) * rank
    tree_probe, _ = _chunk_btree([(o, 0, 0) for o in offsets], rank1, end_key, A_CTREE)
    A_DATA = A_CTREE + len(tree_probe)
    addrs, pos = [], A_DATA
    for p in payloads:
        addrs.append(pos)
        pos += len(p)
    eof = pos
    ctree, ctree_root = _chunk_btree([(o, a, len(p)) for o, a, p in zip(offsets, addrs, payloads)], rank1, end_key, A_CTREE)
    assert len(ctree) == len(tree_probe)
    # ---- assemble
    sb = HDF5_SIGNATURE + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LK, GK, 0) + struct.pack("<QQQQ", 0, UNDEF, eof, UNDEF)
    sb += struct.pack("<QQII", 0, A_ROOT, 1, 0) + struct.pack("<QQ", A_GTREE, A_HEAP)
    assert len(sb) == 96
    root = _object_header([_msg(0x0011, struct.pack("<QQ", A_GTREE, A_HEAP))])
    assert len(root) == 40
    heap = struct.pack("<4sB3xQQQ", b"HEAP", 0, len(heap_data), free_off, A_HEAPDATA) + heap_data
    gtree = struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, UNDEF, UNDEF) + struct.pack("<QQQ", 0, A_SNOD, 8)
    gtree += b"\0" * (gtree_size - len(gtree))
    snod = struct.pack("<4sBxH", b"SNOD", 1, 1) + struct.pack("<QQII16x", 8, A_DSET, 0, 0)
    snod += b"\0" * (snod_size - len(snod))
    blob = sb + root + heap + gtree + snod + dataset_header(ctree_root) + ctree + b"".join(payloads)
    assert len(blob) == eof, (len(blob), eof)
    # atomically: DataLoader workers racing on one index, or an interrupted write, must never leave a truncated file that later passes
    # the exists() check (the reader would return None for it on every epoch)
    tmp = f"{path}.tmp.{os.getpid()}"
    with open(tmp, "wb") as f:
        f.write(blob)
    os.replace(tmp, path)


def np_array_to_h5(array: np.ndarray, outpath: Union[str, Path]) -> None:
    """generic.py:49-55."""
    assert isinstance(array, np.ndarray)
    outpath = Path(outpath)
    assert outpath.suffix == ".h5"
    write_h5_dataset(outpath, array, DATASET_NAME, complevel=1, shuffle=True)


def load_or_construct(voxel_grid_file: Union[str, Path], construct) -> np.ndarray:
    """base.py:205-217 `_load_voxel_grid`: the cached array if the file exists, else `construct()` (an ndarray), saved first."""
    voxel_grid_file = Path(voxel_grid_file)
    if not voxel_grid_file.exists():
        grid = np.asarray(construct())
        np_array_to_h5(grid, voxel_grid_file)
        return grid
    return h5_to_np_array(voxel_grid_file)
