"""Minimal PCD v0.7 reader (ascii / binary) -- enough for the reference's fixture
`wave_matching/tests/data/testscan.pcd` (binary, fields `x y z _ intensity ring _`,
32-byte stride; loaded by the reference tests with pcl::io::loadPCDFile,
wave_matching/tests/icp_tests.cpp:9,26).  Returns float32 XYZ (n, 3)."""
import numpy as np

_NP = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4",
       ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}


def load_pcd_xyz(path):
    with open(path, "rb") as f:
        raw = f.read()
    hdr = {}
    pos = 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(fields))]
    npts = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0].lower()
    if mode == "binary":
        dt = []
        for i, (f_, s, t, c) in enumerate(zip(fields, sizes, types, counts)):
            name = f_ if f_ != "_" else "_pad%d" % i
            dt.append((name, _NP[(t, s)], (c,)) if c > 1 else (name, _NP[(t, s)]))
        dt = np.dtype(dt)
        rec = np.frombuffer(raw, dtype=dt, count=npts, offset=pos)
        xyz = np.stack([rec["x"], rec["y"], rec["z"]], axis=1)
    elif mode == "ascii":
        cols = []
        for f_, c in zip(fields, counts):
            cols += [f_] * c
        arr = np.loadtxt(raw[pos:].decode("ascii").splitlines(), ndmin=2)
        xyz = np.stack([arr[:, cols.index(a)] for a in "xyz"], axis=1)
    else:
        raise ValueError("unsupported PCD DATA mode: %s" % mode)
    return np.ascontiguousarray(xyz, dtype=np.float32)
