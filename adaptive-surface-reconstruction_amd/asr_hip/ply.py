"""Minimal PLY reader / writer for the `asrtool` command line (the reference reads its input with
cpp/bin/plyreader.h: vertex properties x, y, z, nx, ny, nz and an optional per-point radius named `value` or
`radius`, cpp/bin/main.cpp:27-112; it writes the mesh through Open3D, main.cpp:164-174)."""
import numpy as np

_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2",
          "ushort": "u2", "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4",
          "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_points(path):
    """-> (points f32[N,3], normals f32[N,3], radii f32[N] or f32[0]).  ValueError if a required property is
    missing (the reference returns empty arrays and then fails with "points is null!")."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elements = None, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: truncated PLY header" % path)
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment" or tok[0] == "obj_info":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                elements.append({"name": tok[1], "count": int(tok[2]), "props": []})
            elif tok[0] == "property":
                if tok[1] == "list":
                    elements[-1]["props"].append((tok[4], "list", tok[2], tok[3]))
                else:
                    if tok[1] not in _TYPES:
                        raise ValueError("%s: unknown property type %s" % (path, tok[1]))
                    elements[-1]["props"].append((tok[2], _TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: unsupported PLY format %r" % (path, fmt))
        data = None
        for el in elements:
            has_list = any(p[1] == "list" for p in el["props"])
            if el["name"] != "vertex":
                if has_list and fmt != "ascii":
                    raise ValueError("%s: a list element before 'vertex' cannot be skipped in a binary file" % path)
                if fmt == "ascii":
                    for _ in range(el["count"]):
                        f.readline()
                else:
                    f.seek(el["count"] * sum(np.dtype(p[1]).itemsize for p in el["props"]), 1)
                continue
            if has_list:
                raise ValueError("%s: list properties in the vertex element are not supported" % path)
            names = [p[0] for p in el["props"]]
            if fmt == "ascii":
                rows = np.loadtxt(f, dtype=np.float64, max_rows=el["count"], ndmin=2) if el["count"] else np.zeros((0, len(names)))
                data = {n: rows[:, i] for i, n in enumerate(names)}
            else:
                order = "<" if fmt == "binary_little_endian" else ">"
                dt = np.dtype([(n, order + t) for n, t in el["props"]])
                rec = np.frombuffer(f.read(dt.itemsize * el["count"]), dtype=dt, count=el["count"])
                data = {n: rec[n] for n in names}
            break
    if data is None:
        raise ValueError("%s: no vertex element" % path)
    for req in ("x", "y", "z", "nx", "ny", "nz"):
        if req not in data:
            raise ValueError("%s: vertex property %s is missing (needed: x y z nx ny nz)" % (path, req))
    points = np.stack([data["x"], data["y"], data["z"]], 1).astype(np.float32)
    normals = np.stack([data["nx"], data["ny"], data["nz"]], 1).astype(np.float32)
    radii = np.zeros(0, np.float32)
    for name in ("value", "radius"):  # main.cpp:103-106
        if name in data:
            radii = np.asarray(data[name], np.float32)
            break
    return points, normals, radii


def write_points(path, points, normals, radii=None, binary=True):
    """point cloud in the layout read_points takes (used by the tests and to export synthetic scans)"""
    points, normals = np.asarray(points, np.float32), np.asarray(normals, np.float32)
    cols = [points, normals] + ([np.asarray(radii, np.float32)[:, None]] if radii is not None and len(radii) else [])
    table = np.concatenate(cols, 1)
    props = ["x", "y", "z", "nx", "ny", "nz"] + (["radius"] if table.shape[1] == 7 else [])
    with open(path, "wb") as f:
        f.write(("ply\nformat %s 1.0\nelement vertex %d\n" % ("binary_little_endian" if binary else "ascii", len(table))).encode())
        f.write("".join("property float %s\n" % p for p in props).encode())
        f.write(b"end_header\n")
        if binary:
            f.write(table.astype("<f4").tobytes())
        else:
            np.savetxt(f, table, fmt="%.9g")


def write_mesh(path, vertices, triangles, binary=True):
    """triangle mesh: vertices f32[M,3], triangles i32[T,3] (the result dict of reconstruct_surface)"""
    v = np.asarray(vertices, np.float32).reshape(-1, 3)
    t = np.asarray(triangles, np.int32).reshape(-1, 3)
    with open(path, "wb") as f:
        f.write(("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                 "element face %d\nproperty list uchar int vertex_indices\nend_header\n"
                 % ("binary_little_endian" if binary else "ascii", len(v), len(t))).encode())
        if binary:
            f.write(v.astype("<f4").tobytes())
            rec = np.empty(len(t), dtype=[("n", "u1"), ("i", "<i4", 3)])
            rec["n"] = 3
            rec["i"] = t
            f.write(rec.tobytes())
        else:
            np.savetxt(f, v, fmt="%.9g")
            np.savetxt(f, np.concatenate([np.full((len(t), 1), 3), t], 1), fmt="%d")


def read_mesh(path):
    """inverse of write_mesh (tests)"""
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        nv = int([h for h in header if h.startswith("element vertex")][0].split()[2])
        nt = int([h for h in header if h.startswith("element face")][0].split()[2])
        if "format ascii 1.0" in header:
            v = np.loadtxt(f, dtype=np.float32, max_rows=nv, ndmin=2).reshape(-1, 3)
            t = np.loadtxt(f, dtype=np.int32, max_rows=nt, ndmin=2).reshape(-1, 4)[:, 1:] if nt else np.zeros((0, 3), np.int32)
        else:
            v = np.frombuffer(f.read(12 * nv), "<f4").reshape(-1, 3)
            rec = np.frombuffer(f.read(13 * nt), dtype=[("n", "u1"), ("i", "<i4", 3)])
            t = rec["i"].astype(np.int32)
    return v.astype(np.float32), t
