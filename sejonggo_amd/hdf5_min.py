"""Minimal HDF5 writer/reader for the one file shape the self-play path emits: a root group holding a few small,
contiguous, little-endian float32 datasets (`sample.h5`: board, policy_target, value_target -- sgfsave.py:49-79 of
the reference writes them with h5py, train.py:113-119 reads them back).

h5py / libhdf5 are not installed in this image, so the writer follows the published HDF5 File Format Specification
(version 3.0) directly: superblock version 2, version-2 object headers ("OHDR", Jenkins lookup3 checksums), a
new-style root group whose links are stored compactly as Link messages, and per dataset the Dataspace (v2),
Datatype (IEEE float32 LE), Fill Value (v3) and Data Layout (v3, contiguous) messages.  That is the layout libhdf5
itself produces with libver='latest' for such a file.  The reader in this module parses exactly that subset and is
what the tests use; `tests/test_host_logic.py` additionally round-trips through h5py whenever it is importable.
Status: verified by this module's own reader and the lookup3 known-answer vectors; NOT yet verified against libhdf5
(absent here) -- sgfsave falls back to it only when h5py is missing and also writes the .npz twin."""
import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
_UNDEF = 0xFFFFFFFFFFFFFFFF


def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


def lookup3(data, initval=0):
    """Bob Jenkins' lookup3 hashlittle(), the checksum HDF5 uses for version-2 metadata (H5_checksum_lookup3)."""
    length = len(data)
    a = b = c = (0xdeadbeef + length + initval) & 0xFFFFFFFF
    off = 0
    M = 0xFFFFFFFF
    while length > 12:
        a = (a + int.from_bytes(data[off:off + 4], "little")) & M
        b = (b + int.from_bytes(data[off + 4:off + 8], "little")) & M
        c = (c + int.from_bytes(data[off + 8:off + 12], "little")) & M
        a = (a - c) & M; a ^= _rot(c, 4); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 6); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 8); b = (b + a) & M
        a = (a - c) & M; a ^= _rot(c, 16); c = (c + b) & M
        b = (b - a) & M; b ^= _rot(a, 19); a = (a + c) & M
        c = (c - b) & M; c ^= _rot(b, 4); b = (b + a) & M
        off += 12
        length -= 12
    if length == 0:
        return c
    tail = data[off:off + length] + b"\x00" * (12 - length)
    a = (a + int.from_bytes(tail[0:4], "little")) & M
    b = (b + int.from_bytes(tail[4:8], "little")) & M
    c = (c + int.from_bytes(tail[8:12], "little")) & M
    c ^= b; c = (c - _rot(b, 14)) & M
    a ^= c; a = (a - _rot(c, 11)) & M
    b ^= a; b = (b - _rot(a, 25)) & M
    c ^= b; c = (c - _rot(b, 16)) & M
    a ^= c; a = (a - _rot(c, 4)) & M
    b ^= a; b = (b - _rot(a, 14)) & M
    c ^= b; c = (c - _rot(b, 24)) & M
    return c


def _msg(mtype, body, flags=0):
    return struct.pack("<BHB", mtype, len(body), flags) + body


def _ohdr(messages):
    """Version-2 object header, one chunk, 2-byte chunk-size field, no times / creation order."""
    body = b"".join(messages)
    head = b"OHDR" + struct.pack("<BB", 2, 0x01) + struct.pack("<H", len(body))
    blob = head + body
    return blob + struct.pack("<I", lookup3(blob))


def _dataspace(shape):
    if len(shape) == 0:
        return struct.pack("<BBBB", 2, 0, 0, 0)                       # version 2, rank 0, flags 0, type scalar
    return struct.pack("<BBBB", 2, len(shape), 0, 1) + b"".join(struct.pack("<Q", d) for d in shape)


def _datatype_f32le():
    # class 1 (floating point) | version 1; bit field: LE, mantissa normalisation "msb implied", sign bit 31
    return struct.pack("<BBBBI", 0x11, 0x20, 0x1F, 0x00, 4) + struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)


def write_datasets(path, datasets):
    """datasets: ordered mapping name -> array (stored as little-endian float32, contiguous)."""
    names = list(datasets)
    arrays = [np.array(datasets[n], dtype="<f4", order="C") for n in names]      # np.array keeps 0-d scalars 0-d
    sb_size = 48
    # sizes first: object headers are laid out right after the superblock, raw data after them
    link_info = _msg(0x02, struct.pack("<BBQQ", 0, 0, _UNDEF, _UNDEF))
    group_info = _msg(0x0A, struct.pack("<BB", 0, 0))

    def link(name, addr):
        nb = name.encode()
        return _msg(0x06, struct.pack("<BB", 1, 0x00) + struct.pack("<B", len(nb)) + nb + struct.pack("<Q", addr))

    def dset_header(arr, addr):
        return _ohdr([_msg(0x01, _dataspace(arr.shape)), _msg(0x03, _datatype_f32le(), flags=0x01),
                      _msg(0x05, struct.pack("<BB", 3, 0x0A)),
                      _msg(0x08, struct.pack("<BBQQ", 3, 1, addr, arr.nbytes))])

    root_len = len(_ohdr([link_info, group_info] + [link(n, 0) for n in names]))
    dh_len = [len(dset_header(a, 0)) for a in arrays]
    root_addr = sb_size
    dh_addr, pos = [], root_addr + root_len
    for ln in dh_len:
        dh_addr.append(pos)
        pos += ln
    pos = (pos + 7) & ~7
    data_addr = []
    for a in arrays:
        data_addr.append(pos)
        pos += (a.nbytes + 7) & ~7
    eof = pos
    sb = _SIG + struct.pack("<BBBB", 2, 8, 8, 0) + struct.pack("<QQQQ", 0, _UNDEF, eof, root_addr)
    sb += struct.pack("<I", lookup3(sb))
    assert len(sb) == sb_size
    out = bytearray(eof)
    out[0:sb_size] = sb
    root = _ohdr([link_info, group_info] + [link(n, ad) for n, ad in zip(names, dh_addr)])
    out[root_addr:root_addr + len(root)] = root
    for a, ha, da in zip(arrays, dh_addr, data_addr):
        h = dset_header(a, da)
        out[ha:ha + len(h)] = h
        out[da:da + a.nbytes] = a.tobytes()
    with open(path, "wb") as f:
        f.write(bytes(out))


# ---------------------------------------------------------------------------------------------- reader (same subset)
def _parse_ohdr(buf, addr):
    assert buf[addr:addr + 4] == b"OHDR", "not a version-2 object header"
    version, flags = buf[addr + 4], buf[addr + 5]
    assert version == 2
    p = addr + 6
    if flags & 0x20:
        p += 16
    if flags & 0x10:
        p += 4
    szw = 1 << (flags & 3)
    size = int.from_bytes(buf[p:p + szw], "little")
    p += szw
    end = p + size
    stored = int.from_bytes(buf[end:end + 4], "little")
    assert stored == lookup3(bytes(buf[addr:end])), "object header checksum mismatch"
    msgs = []
    while p + 4 <= end:
        mtype, msize, mflags = struct.unpack_from("<BHB", buf, p)
        p += 4
        if flags & 0x04:
            p += 2
        msgs.append((mtype, bytes(buf[p:p + msize])))
        p += msize
    return msgs


def read_datasets(path):
    buf = open(path, "rb").read()
    assert buf[:8] == _SIG and buf[8] == 2 and buf[9] == 8 and buf[10] == 8
    assert int.from_bytes(buf[44:48], "little") == lookup3(buf[:44]), "superblock checksum mismatch"
    root_addr = int.from_bytes(buf[36:44], "little")
    out = {}
    for mtype, body in _parse_ohdr(buf, root_addr):
        if mtype != 0x06:
            continue
        assert body[0] == 1
        lf = body[1]
        p = 2
        if lf & 0x08:
            p += 1
        if lf & 0x04:
            p += 8
        if lf & 0x10:
            p += 1
        lw = 1 << (lf & 3)
        nlen = int.from_bytes(body[p:p + lw], "little")
        p += lw
        name = body[p:p + nlen].decode()
        addr = int.from_bytes(body[p + nlen:p + nlen + 8], "little")
        shape, daddr, dsize = (), None, None
        for t, b in _parse_ohdr(buf, addr):
            if t == 0x01:
                rank = b[1]
                shape = tuple(int.from_bytes(b[4 + 8 * i:12 + 8 * i], "little") for i in range(rank))
            elif t == 0x03:
                assert b[0] == 0x11 and int.from_bytes(b[4:8], "little") == 4, "only float32 is handled"
            elif t == 0x08:
                assert b[0] == 3 and b[1] == 1
                daddr, dsize = int.from_bytes(b[2:10], "little"), int.from_bytes(b[10:18], "little")
        out[name] = np.frombuffer(buf[daddr:daddr + dsize], dtype="<f4").reshape(shape).copy()
    return out
