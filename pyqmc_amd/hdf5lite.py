"""A small read-only HDF5 parser for the files the hot path ingests — SURVEY.md section 8(f4).

Neither image has an HDF5 library, and the reference starts every calculation from a PySCF checkpoint file
(``pyqmc/pyscftools.py:105-191``): the ``mol`` JSON is a variable-length string in the file's global heap and the SCF result
(``scf/mo_coeff``, ``scf/mo_occ``, ``scf/kpts``, k-point lists as ``mo_coeff__from_list__/000000`` ...) are plain contiguous datasets,
complex ones as the compound ``{r, i}`` h5py writes.  This module reads exactly that subset of the format, from the published
HDF5 File Format Specification (version 0/1 superblock, version-1 object headers, symbol-table groups = v1 B-tree + local heap,
contiguous and compact layouts, fixed-point / IEEE float / fixed string / variable-length string / ``{r, i}`` compound types);
anything else (chunked or filtered data, new-style groups) raises ``NotImplementedError`` naming the feature.  ``h5py`` is used
instead wherever it is importable (``chkfile.py``); the test suite pins this parser to the reference's own checkpoint files.

    f = File(path);  f.keys();  f["scf/e_tot"];  f["scf/mo_coeff__from_list__/000003"];  f["mol"]
"""

import struct

import numpy as np

_SIG = b"\x89HDF\r\n\x1a\n"
UNDEF = 0xFFFFFFFFFFFFFFFF


class File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.raw = fh.read()
        b = self.raw
        if b[:8] != _SIG:
            raise ValueError(f"{path}: not an HDF5 file")
        ver = b[8]
        if ver > 1:
            raise NotImplementedError(f"HDF5 superblock version {ver} (files written with libver='latest')")
        self.O, self.L = b[13], b[14]  # sizes of offsets and of lengths
        if (self.O, self.L) != (8, 8):
            raise NotImplementedError("HDF5 files with offsets / lengths other than 8 bytes")
        p = 24 + (4 if ver == 1 else 0)
        self.base = self._u(p, 8)
        root = p + 4 * 8  # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch
        self.root = self._u(root + 8, 8)

    # ---- primitives
    def _u(self, pos, n):
        return int.from_bytes(self.raw[pos : pos + n], "little")

    def _messages(self, addr):
        """(type, body offset, size) of every message of the version-1 object header at ``addr``, continuations followed."""
        b = self.raw
        if b[addr : addr + 4] == b"OHDR":
            raise NotImplementedError("version-2 object headers (libver='latest')")
        if b[addr] != 1:
            raise ValueError(f"object header version {b[addr]} at {addr}")
        nmsg = self._u(addr + 2, 2)
        blocks = [(addr + 16, self._u(addr + 8, 4))]
        out = []
        while blocks and len(out) < nmsg:
            pos, size = blocks.pop(0)
            end = pos + size
            while pos + 8 <= end and len(out) < nmsg:
                mtype, msize = self._u(pos, 2), self._u(pos + 2, 2)
                body = pos + 8
                if mtype == 0x0010:  # continuation
                    blocks.append((self.base + self._u(body, 8), self._u(body + 8, 8)))
                out.append((mtype, body, msize))
                pos = body + msize
        return out

    # ---- groups
    def _group_entries(self, addr):
        """{name: object header address} of the symbol-table group whose header is at ``addr``; None for a dataset."""
        for mtype, body, _ in self._messages(addr):
            if mtype == 0x0011:
                btree, heap = self.base + self._u(body, 8), self.base + self._u(body + 8, 8)
                if self.raw[heap : heap + 4] != b"HEAP":
                    raise ValueError("local heap signature")
                names = self.base + self._u(heap + 24, 8)
                out = {}
                self._walk_btree(btree, names, out)
                return out
            if mtype in (0x0002, 0x0006):
                raise NotImplementedError("new-style (link message) groups")
        return None

    def _walk_btree(self, addr, names, out):
        b = self.raw
        if b[addr : addr + 4] != b"TREE" or b[addr + 4] != 0:
            raise ValueError("group B-tree node")
        level, used = b[addr + 5], self._u(addr + 6, 2)
        pos = addr + 8 + 16
        for k in range(used):
            child = self.base + self._u(pos + 8 + k * 16, 8)  # key_k (8), child_k (8), ...
            if level > 0:
                self._walk_btree(child, names, out)
                continue
            if b[child : child + 4] != b"SNOD":
                raise ValueError("symbol table node")
            for s in range(self._u(child + 6, 2)):
                e = child + 8 + s * 40
                n0 = names + self._u(e, 8)
                out[b[n0 : b.index(b"\0", n0)].decode()] = self.base + self._u(e + 8, 8)

    def _resolve(self, path):
        addr = self.root
        for part in [p for p in path.split("/") if p]:
            ent = self._group_entries(addr)
            if ent is None or part not in ent:
                raise KeyError(path)
            addr = ent[part]
        return addr

    def keys(self, path="/"):
        ent = self._group_entries(self._resolve(path))
        if ent is None:
            raise KeyError(f"{path} is a dataset")
        return sorted(ent)

    def is_group(self, path):
        return self._group_entries(self._resolve(path)) is not None

    def __contains__(self, path):
        try:
            self._resolve(path)
            return True
        except KeyError:
            return False

    # ---- datasets
    def _dtype(self, pos):
        """(numpy dtype or ("vlen-str",) marker, bytes consumed) of the datatype message at ``pos``."""
        b = self.raw
        cls, ver = b[pos] & 0x0F, b[pos] >> 4
        bits = self._u(pos + 1, 3)
        size = self._u(pos + 4, 4)
        order = ">" if bits & 1 else "<"
        if cls == 0:
            return np.dtype(f"{order}{'i' if bits & 8 else 'u'}{size}"), 8 + 4
        if cls == 1:
            if size not in (4, 8):
                raise NotImplementedError(f"{size}-byte floating-point datasets")
            return np.dtype(f"{order}f{size}"), 8 + 12
        if cls == 3:
            return np.dtype(f"S{size}"), 8
        if cls == 9:
            if bits & 0x0F != 1:
                raise NotImplementedError("variable-length sequences (only variable-length strings are read)")
            _, n = self._dtype(pos + 8)
            return ("vlen-str",), 8 + n
        if cls == 6:
            nmem = bits & 0xFFFF
            p, fields = pos + 8, []
            for _ in range(nmem):
                end = b.index(b"\0", p)
                name = b[p:end].decode()
                if ver < 3:
                    p += (end - p + 8) // 8 * 8  # name, null-terminated, padded to a multiple of 8
                    off = self._u(p, 4)
                    p += 4 + (28 if ver == 1 else 0)  # v1: dimensionality, reserved, permutation, reserved, four dimension sizes
                else:
                    p = end + 1
                    nb = max(1, (max(size - 1, 1).bit_length() + 7) // 8)
                    off = self._u(p, nb)
                    p += nb
                dt, n = self._dtype(p)
                p += n
                fields.append((name, dt, off))
            if [f[0] for f in fields] == ["r", "i"] and fields[0][1] == fields[1][1] and fields[0][1].kind == "f":
                return np.dtype(f"{fields[0][1].byteorder.replace('=', '<')}c{size}"), p - pos
            return np.dtype({"names": [f[0] for f in fields], "formats": [f[1] for f in fields], "offsets": [f[2] for f in fields], "itemsize": size}), p - pos
        raise NotImplementedError(f"HDF5 datatype class {cls}")

    def __getitem__(self, path):
        addr = self._resolve(path)
        shape, dtype, data = None, None, None
        for mtype, body, msize in self._messages(addr):
            b = self.raw
            if mtype == 0x0001:
                ver, rank = b[body], b[body + 1]
                p = body + (8 if ver == 1 else 4)
                shape = tuple(self._u(p + 8 * k, 8) for k in range(rank))
            elif mtype == 0x0003:
                dtype, _ = self._dtype(body)
            elif mtype == 0x000B:
                raise NotImplementedError(f"{path}: filtered (compressed) datasets")
            elif mtype == 0x0008:
                ver = b[body]
                if ver == 3:
                    cls = b[body + 1]
                    if cls == 0:
                        n = self._u(body + 2, 2)
                        data = (body + 4, n)
                    elif cls == 1:
                        a = self._u(body + 2, 8)
                        data = (None if a == UNDEF else self.base + a, self._u(body + 10, 8))
                    else:
                        raise NotImplementedError(f"{path}: chunked datasets (resizable block files need h5py)")
                elif ver in (1, 2):
                    rank, cls = b[body + 1], b[body + 2]
                    if cls == 1:
                        a = self._u(body + 8, 8)
                        data = (None if a == UNDEF else self.base + a, None)
                    elif cls == 0:
                        p = body + 8 + 4 * rank
                        data = (p + 4, self._u(p, 4))
                    else:
                        raise NotImplementedError(f"{path}: chunked datasets (resizable block files need h5py)")
                else:
                    raise NotImplementedError(f"data layout message version {ver}")
        if shape is None or dtype is None or data is None:
            if self._group_entries(addr) is not None:
                raise KeyError(f"{path} is a group: {self.keys(path)}")
            raise ValueError(f"{path}: incomplete dataset header")
        count = int(np.prod(shape)) if shape else 1
        if isinstance(dtype, tuple):  # variable-length strings: (length, global heap collection, index) per element
            if data[0] is None:
                return "" if not shape else np.full(shape, "", dtype=object)
            vals = [self._heap_object(self.base + self._u(data[0] + 16 * k + 4, 8), self._u(data[0] + 16 * k + 12, 4), self._u(data[0] + 16 * k, 4)) for k in range(count)]
            return vals[0] if not shape else np.array(vals, dtype=object).reshape(shape)
        if data[0] is None:  # never written: the fill value (zero)
            return np.zeros(shape, dtype=dtype)[()] if shape else dtype.type(0)
        arr = np.frombuffer(self.raw, dtype=dtype, count=count, offset=data[0]).reshape(shape)
        arr = arr.astype(dtype.newbyteorder("="))
        return arr[()] if not shape else arr

    def _heap_object(self, coll, index, length):
        b = self.raw
        if b[coll : coll + 4] != b"GCOL":
            raise ValueError("global heap collection signature")
        end = coll + self._u(coll + 8, 8)
        p = coll + 16
        while p + 16 <= end:
            idx, size = self._u(p, 2), self._u(p + 8, 8)
            if idx == index:
                return b[p + 16 : p + 16 + length].decode("utf-8")
            if idx == 0:
                break
            p += 16 + (size + 7) // 8 * 8
        raise KeyError(f"global heap object {index}")

    def walk(self, path="/"):
        """All dataset paths below ``path``."""
        out = []
        for k in self.keys(path):
            q = path.rstrip("/") + "/" + k
            if self.is_group(q):
                out += self.walk(q)
            else:
                out.append(q)
        return out
