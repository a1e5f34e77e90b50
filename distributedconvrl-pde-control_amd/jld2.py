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
