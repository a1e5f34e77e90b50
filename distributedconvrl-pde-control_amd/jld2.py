"""Minimal pure-Python reader for the subset of JLD2 (HDF5-like) the reference's saved
artifacts use (`FileIO.save(dirpath * "/saves/hook.jld2", "hook", hook)`,
scripts/KS/setup/KSSetup.jl:378-402).  Used by `checkpoint.read_hook` to load reference-trained
actors onto this path and by tests/golden/make_golden.py to extract the
golden vectors committed under tests/golden/*.npz.  Host-side I/O only; no Julia, no h5py.

Format notes: object headers are "OHDR" v2; message 0x01 =
dataspace, 0x03 = datatype, 0x08 = data layout (compact / contiguous); every file address
and every 8-byte object reference is relative to the superblock base address 512; array
dims are stored reversed w.r.t. Julia (memory order is Julia column-major).
"""
import re
import struct

import numpy as np

BASE = 512


class Obj:
    __slots__ = ("off", "dims", "cls", "size", "data_off", "data_size", "sign", "msgs")

    def __repr__(self):
        return f"Obj(off={self.off}, dims={self.dims}, cls={self.cls}, size={self.size}, n={self.data_size})"


def _parse_ohdr(buf, off):
    if buf[off:off + 4] != b"OHDR" or buf[off + 4] != 2:
        return None
    flags = buf[off + 5]
    p = off + 6
    if flags & 0x20:
        p += 16
    if flags & 0x10:
        p += 4
    nsz = 1 << (flags & 3)
    chunk = int.from_bytes(buf[p:p + nsz], "little")
    p += nsz
    end = p + chunk
    o = Obj()
    o.off, o.dims, o.cls, o.size, o.data_off, o.data_size, o.sign = off, None, None, None, None, None, False
    o.msgs = []
    while p + 4 <= end:
        mtype = buf[p]
        msize = struct.unpack_from("<H", buf, p + 1)[0]
        p += 4
        if flags & 0x04:
            p += 2
        pay = buf[p:p + msize]
        o.msgs.append((mtype, p, msize))
        if mtype == 0x01 and msize >= 4:
            rank = pay[1]
            ver = pay[0]
            base = 4 if ver == 2 else 8
            o.dims = tuple(struct.unpack_from("<Q", pay, base + 8 * i)[0] for i in range(rank))
        elif mtype == 0x03 and msize >= 8:
            o.cls = pay[0] & 0xF
            o.sign = bool(pay[1] & 0x08)
            o.size = struct.unpack_from("<I", pay, 4)[0]
        elif mtype == 0x08 and msize >= 2:
            lclass = pay[1]
            if lclass == 0:
                sz = struct.unpack_from("<H", pay, 2)[0]
                o.data_off, o.data_size = p + 4, sz
            elif lclass == 1:
                addr, sz = struct.unpack_from("<QQ", pay, 2)
                if addr != 0xFFFFFFFFFFFFFFFF:
                    o.data_off, o.data_size = addr + BASE, sz
        p += msize
    return o


class JLD2File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        self.objs = {}
        for m in re.finditer(b"OHDR", self.buf):
            o = _parse_ohdr(self.buf, m.start())
            if o is not None:
                self.objs[o.off] = o
        self.order = sorted(self.objs)

    def array(self, o):
        """Return the numeric array of object `o` in Julia shape (column-major semantics
        preserved: result[i, j] == julia[i+1, j+1])."""
        if o.data_off is None or o.cls is None:
            return None
        raw = self.buf[o.data_off:o.data_off + o.data_size]
        if o.cls == 1:
            dt = {4: "<f4", 8: "<f8"}[o.size]
        elif o.cls == 0:
            dt = {1: "i1", 2: "<i2", 4: "<i4", 8: "<i8"}[o.size]
            if not o.sign:
                dt = dt.replace("i", "u")
        elif o.cls == 7:
            dt = "<u8"
        else:
            return None
        a = np.frombuffer(raw, dtype=dt)
        dims = o.dims or ()
        n = int(np.prod(dims)) if dims else a.size
        a = a[:n]
        if dims:
            a = a.reshape(dims)          # on-disk dims = reversed Julia dims, C order
            a = a.T                      # -> Julia shape
        return a

    def deref(self, ref):
        return self.objs.get(int(ref) + BASE)

    def ref_arrays(self):
        """All datasets whose element type is an object reference (class 7)."""
        return [self.objs[k] for k in self.order if self.objs[k].cls == 7 and self.objs[k].dims]

    def numeric(self, cls, size):
        return [self.objs[k] for k in self.order
                if self.objs[k].cls == cls and self.objs[k].size == size and self.objs[k].data_off]


# ---------------------------------------------------------------------------------------------------------------------
# Writer: plain numeric arrays as HDF5 datasets inside a JLD2 container, the subset JLD2.jl reads back as ordinary
# Julia arrays (`load("agent.jld2")` -> Dict{String, Any} of Array{Float32/Float64/Int64}).  Layout mirrors what the
# reference's own files contain (scripts/KS/KS22/saves/hook.jld2): 512-byte text header, superblock v2 at the base
# address 512, object headers "OHDR" v2 with fill-value / dataspace v2 / datatype v3 / layout v4 messages, Jenkins
# lookup3 checksums (verified against the reference file's superblock and root-group checksums, tests/test_host_logic.py).
def _rot(x, k):
    return ((x << k) | (x >> (32 - k))) & 0xFFFFFFFF


def lookup3(data, initval=0):
    """Bob Jenkins' lookup3 hashlittle(): the HDF5 metadata checksum"""
    length = len(data)
    a = b = c = (0xDEADBEEF + length + initval) & 0xFFFFFFFF
    i = 0
    while length > 12:
        a = (a + struct.unpack_from("<I", data, i)[0]) & 0xFFFFFFFF
        b = (b + struct.unpack_from("<I", data, i + 4)[0]) & 0xFFFFFFFF
        c = (c + struct.unpack_from("<I", data, i + 8)[0]) & 0xFFFFFFFF
        a = (a - c) & 0xFFFFFFFF; a ^= _rot(c, 4); c = (c + b) & 0xFFFFFFFF
        b = (b - a) & 0xFFFFFFFF; b ^= _rot(a, 6); a = (a + c) & 0xFFFFFFFF
        c = (c - b) & 0xFFFFFFFF; c ^= _rot(b, 8); b = (b + a) & 0xFFFFFFFF
        a = (a - c) & 0xFFFFFFFF; a ^= _rot(c, 16); c = (c + b) & 0xFFFFFFFF
        b = (b - a) & 0xFFFFFFFF; b ^= _rot(a, 19); a = (a + c) & 0xFFFFFFFF
        c = (c - b) & 0xFFFFFFFF; c ^= _rot(b, 4); b = (b + a) & 0xFFFFFFFF
        i += 12
        length -= 12
    if length == 0:
        return c
    k = struct.unpack("<III", bytes(data[i:]) + b"\0" * (12 - length))
    a = (a + k[0]) & 0xFFFFFFFF
    b = (b + k[1]) & 0xFFFFFFFF
    c = (c + k[2]) & 0xFFFFFFFF
    c ^= b; c = (c - _rot(b, 14)) & 0xFFFFFFFF
    a ^= c; a = (a - _rot(c, 11)) & 0xFFFFFFFF
    b ^= a; b = (b - _rot(a, 25)) & 0xFFFFFFFF
    c ^= b; c = (c - _rot(b, 16)) & 0xFFFFFFFF
    a ^= c; a = (a - _rot(c, 4)) & 0xFFFFFFFF
    b ^= a; b = (b - _rot(a, 14)) & 0xFFFFFFFF
    c ^= b; c = (c - _rot(b, 24)) & 0xFFFFFFFF
    return c


_DT = {   # numpy dtype -> HDF5 datatype message (version 3)
    "float32": bytes.fromhex("31201f000400000000002000170800177f000000"),
    "float64": bytes.fromhex("31203f000800000000004000340b0034ff030000"),
    "int64": bytes.fromhex("3008000008000000" "00004000"),      # fixed-point, little-endian, signed, 64 bits
}


def _msg(mtype, payload):
    return struct.pack("<BHB", mtype, len(payload), 0) + payload


def _ohdr(messages):
    body = b"".join(messages)
    if len(body) < 256:
        head = b"OHDR" + bytes([2, 0]) + struct.pack("<B", len(body))
    else:
        head = b"OHDR" + bytes([2, 1]) + struct.pack("<H", len(body))
    blob = head + body
    return blob + struct.pack("<I", lookup3(blob))


def write_arrays(path, arrays, julia_version="1.9.1"):
    """arrays: {name: numpy array in its JULIA shape} (float32 / float64 / int64).  Element (i, j) of the Julia array is
    a[i, j]; the bytes on disk are Julia's column-major order and the dataspace lists the dims reversed, as JLD2 does."""
    out = bytearray(b"HDF5-based Julia Data Format, version 0.1.1\x00" + f" (Julia {julia_version} 64-bit LE)\x00".encode())
    out += b"\0" * (BASE - len(out))
    out += b"\0" * 48                                   # superblock, filled in at the end
    links = []
    for name, a in arrays.items():
        a = np.asarray(a)
        dt = _DT.get(a.dtype.name)
        if dt is None:
            raise ValueError(f"write_arrays: dtype {a.dtype} of {name!r} is not supported (float32 / float64 / int64)")
        data = np.ascontiguousarray(a.T).tobytes()      # column-major bytes of the Julia-shaped array
        dims = tuple(reversed(a.shape)) if a.ndim else ()
        space = bytes([2, len(dims), 0, 1]) + b"".join(struct.pack("<Q", d) for d in dims)
        msgs = [_msg(0x05, bytes([3, 9])), _msg(0x01, space), _msg(0x03, dt)]
        if len(data) < 60000:
            msgs.append(_msg(0x08, bytes([4, 0]) + struct.pack("<H", len(data)) + data))          # compact layout
            blob = _ohdr(msgs)
            addr = len(out) - BASE
            out += blob
        else:
            daddr = len(out) - BASE                                                                # contiguous layout
            out += data
            msgs.append(_msg(0x08, bytes([4, 1]) + struct.pack("<QQ", daddr, len(data))))
            blob = _ohdr(msgs)
            addr = len(out) - BASE
            out += blob
        links.append((name.encode(), addr))
    # root group: link info, group info, one hard link per dataset
    msgs = [_msg(0x02, bytes([0, 0]) + b"\xff" * 16), _msg(0x0A, bytes([0, 0]))]
    for nm, addr in links:
        if len(nm) > 255:
            raise ValueError("write_arrays: dataset names are limited to 255 bytes")
        msgs.append(_msg(0x06, bytes([1, 0x10, 1, len(nm)]) + nm + struct.pack("<Q", addr)))
    root = len(out) - BASE
    out += _ohdr(msgs)
    sb = b"\x89HDF\r\n\x1a\n" + bytes([2, 8, 8, 0]) + struct.pack("<QQQQ", BASE, 0xFFFFFFFFFFFFFFFF, len(out), root)
    out[BASE:BASE + 48] = sb + struct.pack("<I", lookup3(sb))
    with open(path, "wb") as fh:
        fh.write(out)


def read_arrays(path):
    """{name: array in Julia shape} of a file written by write_arrays (follows the root group's hard links)"""
    f = JLD2File(path)
    b = f.buf
    root = struct.unpack_from("<Q", b, BASE + 36)[0] + BASE
    ro = f.objs[root]
    out = {}
    for mtype, p, msize in ro.msgs:
        if mtype != 0x06:
            continue
        n = b[p + 3]
        name = b[p + 4:p + 4 + n].decode()
        addr = struct.unpack_from("<Q", b, p + 4 + n)[0] + BASE
        o = f.objs[addr]
        a = f.array(o)
        if o.cls == 0 and o.size == 8:
            a = a.astype(np.int64)
        out[name] = a
    return out
