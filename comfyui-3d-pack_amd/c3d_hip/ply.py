"""Minimal PLY reader/writer with the slice of the `plyfile` API the 3DGS wire format needs (plyfile is not installed and
not vendored by the reference: `from plyfile import PlyData, PlyElement`, main_3DGS_renderer.py:5).

Supported: one or more elements of scalar properties (float/double/int/uint/short/ushort/char/uchar), formats
binary_little_endian 1.0 and ascii 1.0; list properties are read and skipped for mesh files we do not need here.
API subset: PlyElement.describe(structured_array, name), PlyData([elements]).write(path), PlyData.read(path),
plydata.elements[i][prop], plydata.elements[i].properties[j].name, plydata['vertex']."""
import numpy as np

_T = {"float": "f4", "float32": "f4", "double": "f8", "float64": "f8", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
      "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2", "char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1"}
_N = {"f4": "float", "f8": "double", "i4": "int", "u4": "uint", "i2": "short", "u2": "ushort", "i1": "char", "u1": "uchar"}


class PlyProperty:
    def __init__(self, name, dtype):
        self.name, self.dtype = name, dtype

    def __repr__(self):
        return "PlyProperty(%r, %r)" % (self.name, self.dtype)


class PlyElement:
    def __init__(self, name, data):
        self.name, self.data = name, data
        self.properties = tuple(PlyProperty(n, data.dtype[n].str.lstrip("<|=")) for n in data.dtype.names)

    @staticmethod
    def describe(data, name):
        return PlyElement(name, np.asarray(data))

    @property
    def count(self):
        return self.data.shape[0]

    def __getitem__(self, key):
        return self.data[key]

    def __len__(self):
        return self.data.shape[0]


class PlyData:
    def __init__(self, elements=(), text=False):
        self.elements, self.text = list(elements), text

    def __getitem__(self, name):
        for e in self.elements:
            if e.name == name:
                return e
        raise KeyError(name)

    def write(self, stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "wb") if own else stream
        try:
            hdr = ["ply", "format %s 1.0" % ("ascii" if self.text else "binary_little_endian")]
            for e in self.elements:
                hdr.append("element %s %d" % (e.name, e.count))
                hdr += ["property %s %s" % (_N[p.dtype], p.name) for p in e.properties]
            hdr.append("end_header")
            f.write(("\n".join(hdr) + "\n").encode("ascii"))
            for e in self.elements:
                if self.text:
                    np.savetxt(f, np.stack([e.data[n] for n in e.data.dtype.names], 1), fmt="%.9g")
                else:
                    f.write(e.data.astype(e.data.dtype.newbyteorder("<"), copy=False).tobytes())
        finally:
            if own:
                f.close()

    @staticmethod
    def read(stream):
        own = isinstance(stream, (str, bytes)) or hasattr(stream, "__fspath__")
        f = open(stream, "rb") if own else stream
        try:
            if f.readline().strip() != b"ply":
                raise ValueError("not a PLY file")
            fmt, elems = None, []
            while True:
                line = f.readline()
                if not line:
                    raise ValueError("unexpected end of PLY header")
                tok = line.decode("ascii").split()
                if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                    continue
                if tok[0] == "format":
                    fmt = tok[1]
                elif tok[0] == "element":
                    elems.append([tok[1], int(tok[2]), []])
                elif tok[0] == "property":
                    if tok[1] == "list":
                        elems[-1][2].append((tok[4], ("list", _T[tok[2]], _T[tok[3]])))
                    else:
                        elems[-1][2].append((tok[2], _T[tok[1]]))
                elif tok[0] == "end_header":
                    break
            out = []
            for name, count, props in elems:
                if any(isinstance(t, tuple) for _, t in props):
                    if fmt == "ascii":
                        for _ in range(count):
                            f.readline()
                    else:   # variable-length rows: walk them
                        for _ in range(count):
                            for _, t in props:
                                if isinstance(t, tuple):
                                    n = int(np.frombuffer(f.read(np.dtype(t[1]).itemsize), dtype="<" + t[1])[0])
                                    f.read(n * np.dtype(t[2]).itemsize)
                                else:
                                    f.read(np.dtype(t).itemsize)
                    continue
                dt = np.dtype([(n, "<" + t) for n, t in props])
                if fmt == "ascii":
                    rows = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
                    data = np.zeros(count, dtype=dt)
                    for j, (n, _) in enumerate(props):
                        data[n] = rows[:, j]
                elif fmt == "binary_little_endian":
                    data = np.frombuffer(f.read(dt.itemsize * count), dtype=dt, count=count).copy()
                else:
                    raise ValueError("unsupported PLY format %r" % fmt)
                out.append(PlyElement(name, data))
            return PlyData(out, text=(fmt == "ascii"))
        finally:
            if own:
                f.close()
