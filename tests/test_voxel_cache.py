"""the voxel-grid cache wire format (HDF5 + Blosc filter 32001, blosc:zstd, byte shuffle) -- host-side I/O, CPU tests.
PINNED against the reference: tests/golden/voxel_cache/*.h5 were written by the reference's own `np_array_to_h5`
(data/utils/generic.py:49-55) on real libhdf5 1.10.6 / h5py 3.3.0 / hdf5-blosc / c-blosc 1.20.1 (tests/golden/make_voxel_cache_golden.py, run
with the build image's conda interpreter); where that interpreter exists the module's WRITER is also read back by libhdf5 itself.
In addition the reader is checked against byte streams assembled here by hand from the published formats (layouts neither writer
produces), and writer -> reader round trips."""
import json
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

from bflow_amd import voxel_cache as VC

U = VC.UNDEF


def _sparse_grid(shape, seed, dtype=np.float32):
    rs = np.random.RandomState(seed)
    g = rs.standard_normal(shape) * (rs.uniform(size=shape) < 0.3)
    return g.astype(dtype)


GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "voxel_cache")
CONDA_PY = "/opt/conda/bin/python3.9"            # the build image's interpreter with h5py + PyTables (Blosc filter); absent on the GPU box


def golden_array(shape, seed):
    """The arrays of tests/golden/make_voxel_cache_golden.py (same code)."""
    rs = np.random.RandomState(seed)
    return (rs.standard_normal(shape) * (rs.uniform(size=shape) < 0.3)).astype(np.float32)


# ------------------------------------------------------------------------------------------------------------------ the reference's own files
def test_reader_on_files_written_by_the_reference():
    """Files produced by the reference's np_array_to_h5 (real libhdf5 + hdf5-blosc + c-blosc/zstd): bit-exact through this module's reader."""
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    assert meta["hdf5"].startswith("1.") and len(meta["cases"]) == 3
    for name, c in meta["cases"].items():
        assert c["filters"] == {"32001": [2, 2, 4, int(np.prod(c["chunks"])) * 4, 1, 1, 5]}     # hdf5-blosc's cd_values (module header)
        a = VC.h5_to_np_array(os.path.join(GOLDEN, name + ".h5"))
        assert a.dtype == np.float32 and a.shape == tuple(c["shape"])
        assert np.array_equal(a, golden_array(tuple(c["shape"]), c["seed"]))
        with pytest.raises(KeyError):
            VC.read_h5_dataset(os.path.join(GOLDEN, name + ".h5"), "flow")


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="needs the build image's conda interpreter (h5py + PyTables' Blosc filter)")
def test_writer_output_is_read_by_libhdf5(tmp_path):
    """The other direction: files written by this module, read by h5py on the real libhdf5 with the real Blosc filter."""
    probe = subprocess.run([CONDA_PY, "-c", "import numpy; numpy.typeDict = numpy.sctypeDict; import tables, h5py; assert h5py.h5z.filter_avail(32001)"],
                           capture_output=True)
    if probe.returncode != 0:
        pytest.skip("h5py / PyTables not importable in the conda interpreter")
    cases = {"a": golden_array((15, 60, 80), 21), "b": golden_array((5, 33, 47), 22), "c": golden_array((9, 480, 640), 23)[:, ::3, ::5].copy()}
    for k, arr in cases.items():
        VC.np_array_to_h5(arr, tmp_path / f"{k}.h5")
        np.save(tmp_path / f"{k}.npy", arr)
    VC.write_h5_dataset(tmp_path / "d.h5", cases["a"], chunks=(1, 8, 8))                     # 600 chunks: a two-level chunk B-tree
    np.save(tmp_path / "d.npy", cases["a"])
    script = (
        "import sys, numpy\n"
        "numpy.typeDict = numpy.sctypeDict\n"
        "import tables, h5py\n"
        "for k in 'abcd':\n"
        "    with h5py.File(sys.argv[1] + '/' + k + '.h5', 'r') as f:\n"
        "        d = f['voxel_grid']\n"
        "        assert list(f.keys()) == ['voxel_grid'] and '32001' in d._filters, d._filters\n"
        "        b = d[...]\n"
        "    a = numpy.load(sys.argv[1] + '/' + k + '.npy')\n"
        "    assert b.dtype == a.dtype and b.shape == a.shape and (a == b).all(), k\n"
        "print('OK')\n")
    r = subprocess.run([CONDA_PY, "-c", script, str(tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0 and "OK" in r.stdout, r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------------ Blosc frames
def _frame(flags, typesize, nbytes, blocksize, blocks):
    """Hand-assembled Blosc-1 frame: 16-byte header | int32 bstarts | blocks (each = its streams, every stream int32-length prefixed)."""
    nb = len(blocks)
    starts, body = [], b""
    for blk in blocks:
        starts.append(16 + 4 * nb + len(body))
        body += b"".join(struct.pack("<i", len(s)) + s for s in blk)
    return struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, 16 + 4 * nb + len(body)) + struct.pack(f"<{nb}i", *starts) + body


def test_blosc_split_shuffled_zstd_frame():
    """What c-blosc emits for typesize 4 with block splitting: every full block = 4 byte-plane streams, the leftover block = 1 stream."""
    data = _sparse_grid((3 * 1024 + 100,), 0).tobytes()
    bs = 4096
    blocks = []
    for i in range(0, len(data), bs):
        blk = VC.byte_shuffle(data[i:i + bs], 4)
        if len(blk) == bs:
            ne = bs // 4
            blocks.append([VC.zstd_compress(blk[k * ne:(k + 1) * ne], 1) for k in range(4)])
        else:
            blocks.append([VC.zstd_compress(blk, 1)])
    frame = _frame((4 << 5) | 0x1, 4, len(data), bs, blocks)
    assert VC.blosc_decompress(frame) == data


def test_blosc_stored_streams_dont_split_and_memcpyed():
    rs = np.random.RandomState(1)
    data = rs.bytes(5000)                                             # incompressible
    # a stream whose length equals the raw length is stored verbatim
    frame = _frame((4 << 5) | 0x10 | 0x1, 2, len(data), 2048, [[VC.byte_shuffle(data[i:i + 2048], 2)] for i in range(0, 5000, 2048)])
    assert VC.blosc_decompress(frame) == data
    memcpyed = struct.pack("<BBBBIII", 2, 1, (4 << 5) | 0x2 | 0x1, 4, len(data), 2048, 16 + len(data)) + data
    assert VC.blosc_decompress(memcpyed) == data
    # zlib streams (compressor format 3), no shuffle
    z = _frame((3 << 5) | 0x10, 1, 3000, 1024, [[zlib.compress(b"ab" * 512)], [zlib.compress(b"cd" * 512)], [zlib.compress(b"e" * 952)]])
    assert VC.blosc_decompress(z) == b"ab" * 512 + b"cd" * 512 + b"e" * 952
    with pytest.raises(VC.VoxelCacheError):
        VC.blosc_decompress(frame[:40])
    with pytest.raises(VC.VoxelCacheError):
        VC.blosc_decompress(struct.pack("<BBBBIII", 2, 1, (0 << 5), 4, 64, 64, 100) + b"\0" * 84)    # blosclz: refused, not mis-decoded


def test_blosc_writer_frames_decode_and_compress():
    g = _sparse_grid((15, 60, 80), 2)
    for split in (None, True):
        f = VC.blosc_compress(g.tobytes(), 4, 1, True, split=split)
        assert f[0] == 2 and f[3] == 4 and struct.unpack_from("<I", f, 4)[0] == g.nbytes and struct.unpack_from("<I", f, 12)[0] == len(f)
        assert (f[2] >> 5) == 4 and (f[2] & 1) and bool(f[2] & 0x10) == (split is None)
        assert len(f) < g.nbytes // 2
        assert VC.blosc_decompress(f) == g.tobytes()
    rnd = np.random.RandomState(3).bytes(1000)
    f = VC.blosc_compress(rnd, 1)
    assert f[2] & 0x2 and VC.blosc_decompress(f) == rnd                # memcpyed
    assert VC.blosc_decompress(VC.blosc_compress(b"", 4)) == b""


# ------------------------------------------------------------------------------------------------------------------ HDF5 by hand
def _oh(msgs, first_block=None):
    body = b"".join(msgs)
    return struct.pack("<BxHII4x", 1, len(msgs) if first_block is None else first_block[0], 1, len(body)) + body


def _m(t, d, flags=0):
    d += b"\0" * (-len(d) % 8)
    return struct.pack("<HHB3x", t, len(d), flags) + d


def _hand_file(array, filters_msg, chunk_bytes_list, chunk_dims, offsets, superblock_version=1, continuation=True, masks=None):
    """A file laid out differently from the module's writer: superblock 1, dataspace with max dims, header continuation block, NIL
    message, modification-time message, SNOD reached through the root header's symbol-table message."""
    rank = array.ndim
    name = b"voxel_grid\0"
    heap_data = (b"\0" * 8 + name + b"\0" * (-len(name) % 8)) + struct.pack("<QQ", 1, 16)
    sb_len = 96 + (4 if superblock_version == 1 else 0)
    A_ROOT = sb_len + 8                                               # deliberately not back to back
    A_HEAP = A_ROOT + 40
    A_HD = A_HEAP + 32
    A_GT = A_HD + len(heap_data)
    A_SN = A_GT + 544
    A_DS = A_SN + 328
    dt = VC._datatype_message(array.dtype)
    space = struct.pack("<BBB5x", 1, rank, 1) + struct.pack(f"<{rank}Q", *array.shape) + struct.pack(f"<{rank}Q", *array.shape)
    first = [_m(0x0001, space), _m(0x0000, b"\0" * 8), _m(0x0003, dt), _m(0x0012, struct.pack("<B3xI", 1, 1700000000))]
    second = [_m(0x0005, struct.pack("<BBBB", 2, 3, 2, 0))]
    if filters_msg is not None:
        second.append(_m(0x000B, filters_msg))
    rank1 = rank + 1
    ksize = 8 + 8 * rank1
    ctree_size = 24 + 64 * 8 + 65 * ksize
    lay_len = len(_m(0x0008, struct.pack("<BBBQ", 3, 2, rank1, 0) + b"\0" * 4 * rank1))
    cont_len = len(b"".join(second)) + lay_len
    first_len = len(b"".join(first)) + 24                            # + the continuation message itself
    A_CONT = A_DS + 16 + first_len + 24                              # a gap after the first block
    A_CT = A_CONT + cont_len + 8
    A_DATA = A_CT + ctree_size
    addrs, pos = [], A_DATA
    for c in chunk_bytes_list:
        addrs.append(pos)
        pos += len(c) + 3                                             # unaligned chunks with gaps
    eof = pos
    node = struct.pack("<4sBBHQQ", b"TREE", 1, 0, len(addrs), U, U)
    for i, (o, a, c) in enumerate(zip(offsets, addrs, chunk_bytes_list)):
        node += struct.pack("<II", len(c), masks[i] if masks else 0) + struct.pack(f"<{rank1}Q", *o, 0) + struct.pack("<Q", a)
    node += struct.pack("<II", 0, 0) + struct.pack(f"<{rank1}Q", array.shape[0] + chunk_dims[0], *([0] * rank))
    node += b"\0" * (ctree_size - len(node))
    lay = _m(0x0008, struct.pack("<BBB", 3, 2, rank1) + struct.pack("<Q", A_CT) + struct.pack(f"<{rank1}I", *chunk_dims, array.dtype.itemsize))
    second.append(lay)
    cont_msg = _m(0x0010, struct.pack("<QQ", A_CONT, cont_len))
    nmsgs = len(first) + 1 + len(second)
    ds = struct.pack("<BxHII4x", 1, nmsgs, 1, first_len) + b"".join(first) + cont_msg
    sb = VC.HDF5_SIGNATURE + struct.pack("<BBBBBBBBHHI", superblock_version, 0, 0, 0, 0, 8, 8, 0, 4, 16, 0)
    if superblock_version == 1:
        sb += struct.pack("<HH", 32, 0)
    sb += struct.pack("<QQQQ", 0, U, eof, U) + struct.pack("<QQII", 0, A_ROOT, 1, 0) + struct.pack("<QQ", A_GT, A_HEAP)
    assert len(sb) == sb_len
    blob = bytearray(eof)
    blob[0:len(sb)] = sb
    root = _oh([_m(0x0011, struct.pack("<QQ", A_GT, A_HEAP))])
    blob[A_ROOT:A_ROOT + len(root)] = root
    heap = struct.pack("<4sB3xQQQ", b"HEAP", 0, len(heap_data), len(heap_data) - 16, A_HD) + heap_data
    blob[A_HEAP:A_HEAP + len(heap)] = heap
    gt = struct.pack("<4sBBHQQ", b"TREE", 0, 0, 1, U, U) + struct.pack("<QQQ", 0, A_SN, 8)
    blob[A_GT:A_GT + len(gt)] = gt
    sn = struct.pack("<4sBxH", b"SNOD", 1, 1) + struct.pack("<QQII16x", 8, A_DS, 0, 0)
    blob[A_SN:A_SN + len(sn)] = sn
    blob[A_DS:A_DS + len(ds)] = ds
    blob[A_CONT:A_CONT + cont_len] = b"".join(second)
    blob[A_CT:A_CT + len(node)] = node
    for a, c in zip(addrs, chunk_bytes_list):
        blob[a:a + len(c)] = c
    return bytes(blob)


def test_reader_on_hand_assembled_hdf5_blosc(tmp_path):
    g = _sparse_grid((5, 20, 24), 4)
    cd = (2, 2, 4, 2 * 16 * 16 * 4, 1, 1, 5)
    fmsg = struct.pack("<BB6x", 1, 1) + struct.pack("<HHHH", 32001, 8, 1, 7) + b"blosc\0\0\0" + struct.pack("<7I", *cd) + b"\0" * 4
    chunks, offsets = [], []
    for c0 in range(0, 5, 2):
        for y in range(0, 20, 16):
            for x in range(0, 24, 16):
                blk = np.zeros((2, 16, 16), np.float32)
                src = g[c0:c0 + 2, y:y + 16, x:x + 16]
                blk[:src.shape[0], :src.shape[1], :src.shape[2]] = src
                sh = VC.byte_shuffle(blk.tobytes(), 4)
                ne = len(sh) // 4                                          # one block, split into the 4 byte planes
                chunks.append(_frame((4 << 5) | 1, 4, blk.nbytes, blk.nbytes, [[VC.zstd_compress(sh[k * ne:(k + 1) * ne]) for k in range(4)]]))
                offsets.append((c0, y, x))
    p = tmp_path / "000012.h5"
    p.write_bytes(_hand_file(g, fmsg, chunks, (2, 16, 16), offsets))
    out = VC.h5_to_np_array(p)
    assert out.dtype == np.float32 and out.shape == g.shape and np.array_equal(out, g)


def test_reader_on_hand_assembled_hdf5_gzip_shuffle_and_filter_mask(tmp_path):
    """HDF5's own shuffle (2) + deflate (1) pipeline, and a chunk whose filter mask says "deflate was skipped"."""
    g = (np.arange(4 * 6 * 8, dtype=np.int16).reshape(4, 6, 8) * 3 - 100).astype(">i2")
    fmsg = struct.pack("<BB6x", 1, 2) + struct.pack("<HHHH", 2, 0, 0, 1) + struct.pack("<I", 2) + b"\0" * 4 + \
        struct.pack("<HHHH", 1, 0, 0, 1) + struct.pack("<I", 6) + b"\0" * 4
    chunks, offsets, masks = [], [], []
    for i, c0 in enumerate(range(0, 4, 2)):
        sh = VC.byte_shuffle(g[c0:c0 + 2].tobytes(), 2)
        skip = i == 1
        chunks.append(sh if skip else zlib.compress(sh, 6))
        masks.append(0b10 if skip else 0)
        offsets.append((c0, 0, 0))
    p = tmp_path / "x.h5"
    p.write_bytes(_hand_file(g, fmsg, chunks, (2, 6, 8), offsets, superblock_version=0, masks=masks))
    out = VC.read_h5_dataset(p)
    assert out.dtype == np.dtype("int16") and np.array_equal(out, g.astype(np.int16))


# ------------------------------------------------------------------------------------------------------------------ writer <-> reader
@pytest.mark.parametrize("shape,chunks,dtype", [
    ((15, 48, 64), (1, 8, 8), np.float32),          # 720 chunks: a two-level chunk B-tree (64 entries per node)
    ((5, 33, 47), (2, 16, 16), np.float32),         # ragged edge chunks
    ((9, 60, 80), None, np.float32),                # default chunking
    ((3, 7, 5), None, np.float16), ((2, 10), (1, 4), np.float64), ((6, 4, 4), (4, 4, 4), np.int16), ((100,), (7,), np.uint8), ((2, 3, 4, 5), None, np.int64),
])
def test_writer_reader_round_trip(tmp_path, shape, chunks, dtype):
    g = _sparse_grid(shape, 5, np.float64)
    g = (g * 50).astype(dtype) if np.issubdtype(dtype, np.integer) else g.astype(dtype)
    p = tmp_path / "000000.h5"
    VC.write_h5_dataset(p, g, chunks=chunks)
    raw = p.read_bytes()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and struct.unpack_from("<Q", raw, 40)[0] == len(raw)      # end-of-file address
    out = VC.h5_to_np_array(p)
    assert out.dtype == np.dtype(dtype) and out.shape == tuple(shape) and np.array_equal(out, g)
    with pytest.raises(KeyError):
        VC.read_h5_dataset(p, "flow")


def test_known_answer_bytes():
    """Constants of the formats, written out by hand."""
    assert VC._datatype_message(np.float32) == bytes([0x11, 0x20, 0x1F, 0x00, 4, 0, 0, 0, 0, 0, 32, 0, 23, 8, 0, 23, 127, 0, 0, 0])
    assert VC._datatype_message(np.dtype("<i2")) == bytes([0x10, 0x08, 0, 0, 2, 0, 0, 0, 0, 0, 16, 0])
    assert VC.byte_shuffle(bytes([1, 2, 3, 4, 5, 6, 7, 8, 9]), 4) == bytes([1, 5, 2, 6, 3, 7, 4, 8, 9])
    assert VC.byte_unshuffle(bytes([1, 5, 2, 6, 3, 7, 4, 8, 9]), 4) == bytes([1, 2, 3, 4, 5, 6, 7, 8, 9])
    assert VC.zstd_compress(b"")[:4] == b"\x28\xb5\x2f\xfd"                                         # Zstandard frame magic
    assert VC.BLOSC_COMPRESSORS.index("zstd") == 5 and VC.BLOSC_FILTER_ID == 32001                   # generic.py:37-41


def test_cache_protocol_and_names(tmp_path, capsys):
    d = VC.dsec_voxel_grid_dir(tmp_path / "events" / "left", 15, True)
    assert d.name == "voxel_grids_v1_100ms_forward_15_bins" and VC.dsec_voxel_grid_dir(tmp_path, 5, False).name == "voxel_grids_v0_100ms_forward_5_bins"
    assert VC.dsec_voxel_grid_file(d, 42).name == "000042.h5"
    assert VC.multiflow_voxel_grid_file(tmp_path, 65).name == "voxel_grid_v1_65_bins.h5"
    assert VC.multiflow_voxel_grid_file(tmp_path, 65, False, True).name == "voxel_grid_v0_65_bins_downsampled.h5"
    os.makedirs(d)
    calls = []
    g = _sparse_grid((5, 16, 24), 6)

    def construct():
        calls.append(1)
        return g
    f = VC.dsec_voxel_grid_file(d, 7)
    a = VC.load_or_construct(f, construct)
    b = VC.load_or_construct(f, construct)
    assert len(calls) == 1 and f.exists() and np.array_equal(a, g) and np.array_equal(b, g)
    # unreadable file: None + the reference's message (generic.py:66-68)
    good = f.read_bytes()
    f.write_bytes(good[:300])
    assert VC.h5_to_np_array(f) is None and "Error loading" in capsys.readouterr().out
    # damage inside the structures (same length): every outcome is the array, or None -- never an unrelated exception
    rs = np.random.RandomState(0)
    for _ in range(200):
        bad = bytearray(good)
        for pos in rs.randint(8, min(len(bad), 1400), size=3):
            bad[pos] = rs.randint(0, 256)
        f.write_bytes(bytes(bad))
        try:
            r = VC.h5_to_np_array(f)
        except KeyError:
            r = None                                          # the dataset name itself was hit: h5py raises KeyError too
        assert r is None or isinstance(r, np.ndarray)
    capsys.readouterr()
