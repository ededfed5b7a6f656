"""Data / IO edges of the evaluation driver (the last "next" row):
`resample_pcd` (data_util.py:8-13), PCD read/write without open3d (io_util.py:7-15) and the
`id,cd,emd` results table of recon_test.py:42-44,68 (whose `emd` column holds fidelity_loss,
recon_test.py:28,65).  Host-side numpy only; nothing here is on the kernel path.
"""
import csv
import os

import numpy as np


def resample_pcd(pcd, n, rng=None):
    """Drop or duplicate points so that pcd has exactly n points: the first n if there are
    enough, else all of them followed by uniformly drawn duplicates (which is where the exact
    ties in the operators' inputs come from)."""
    pcd = np.asarray(pcd)
    idx = np.arange(pcd.shape[0])
    if idx.shape[0] < n:
        draw = (rng if rng is not None else np.random).randint(pcd.shape[0], size=n - pcd.shape[0])
        idx = np.concatenate([idx, draw])
    return pcd[idx[:n]]


def save_pcd(filename, points, binary=True):
    """Writes an unorganised x/y/z float32 PCD v0.7 (what open3d's write_point_cloud emits for a
    cloud without colours/normals; open3d's default is binary)."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32).reshape(-1, 3))
    n = pts.shape[0]
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z\nSIZE 4 4 4\n"
              "TYPE F F F\nCOUNT 1 1 1\nWIDTH %d\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\n"
              "DATA %s\n" % (n, n, "binary" if binary else "ascii"))
    with open(filename, "wb") as f:
        f.write(header.encode("ascii"))
        if binary:
            f.write(pts.tobytes())
        else:
            for p in pts:
                f.write(("%.9g %.9g %.9g\n" % (p[0], p[1], p[2])).encode("ascii"))


def read_pcd(filename):
    """Reads the x, y, z fields of an ascii or binary PCD file -> (n,3) float64 like
    np.array(pcd.points) in the reference."""
    with open(filename, "rb") as f:
        raw = f.read()
    fields, sizes, types, counts, npoints, data_kind, pos = None, None, None, None, None, None, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, rest = line.partition(" ")
        vals = rest.split()
        if key == "FIELDS":
            fields = vals
        elif key == "SIZE":
            sizes = [int(v) for v in vals]
        elif key == "TYPE":
            types = vals
        elif key == "COUNT":
            counts = [int(v) for v in vals]
        elif key == "POINTS":
            npoints = int(vals[0])
        elif key == "DATA":
            data_kind = vals[0]
            break
    if counts is None:
        counts = [1] * len(fields)
    if data_kind == "ascii":
        arr = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2)
        col, cols = 0, {}
        for name, c in zip(fields, counts):
            cols[name] = col
            col += c
        return arr[:npoints][:, [cols["x"], cols["y"], cols["z"]]]
    if data_kind == "binary":
        kinds = {"F": "f", "I": "i", "U": "u"}
        dt = np.dtype([(name, "<%s%d" % (kinds[t], s), (c,) if c > 1 else ())
                       for name, s, t, c in zip(fields, sizes, types, counts)])
        rec = np.frombuffer(raw, dtype=dt, count=npoints, offset=pos)
        return np.stack([rec["x"], rec["y"], rec["z"]], -1).astype(np.float64)
    raise ValueError("PCD DATA %r is not supported (ascii and binary are)" % data_kind)


def write_results_csv(path, rows):
    """rows: iterable of (model_id, cd, emd) -> the reference's results.csv (header id,cd,emd)."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["id", "cd", "emd"])
        for r in rows:
            w.writerow(list(r))


def read_results_csv(path):
    with open(path, newline="") as f:
        rd = csv.reader(f)
        header = next(rd)
        assert header == ["id", "cd", "emd"], header
        return [(r[0], float(r[1]), float(r[2])) for r in rd]


def per_category_means(rows):
    """{synset_id: (mean cd, mean emd)} as recon_test.py:70-77,95-100 prints them."""
    acc = {}
    for model_id, cd, emd in rows:
        acc.setdefault(model_id.split("/")[0], []).append((cd, emd))
    return {k: tuple(np.mean(np.asarray(v), 0)) for k, v in acc.items()}
