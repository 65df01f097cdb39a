"""Keyframe directories on disk — the format KeyFrame::save / KeyFrame::load read and write
(src/hdl_graph_slam/keyframe.cpp:21-58, :60-145; directories named "%06d" by the save/load services,
apps/hdl_graph_slam_nodelet.cpp:818-923, 932-974): a text file `data` with whitespace-separated tokens

    stamp <sec> <nsec> | estimate <4x4 row-major> | odom <4x4 row-major> | accum_distance <d>
    [floor_coeffs a b c d] [utm_coord x y z] [acceleration x y z] [orientation w x y z] id <node id>

and `cloud.pcd`, a binary PCD v0.7 of pcl::PointXYZI (FIELDS x y z intensity, 16 bytes per point).
"Next" row f4: with this, loop-closure candidates can be loaded straight into resident hgs_clouds (RegistrationHIP.upload),
whose search index / covariances then persist on the device across detections.
"""
from __future__ import annotations

import dataclasses
import os
from typing import Optional

import numpy as np

from . import synth


@dataclasses.dataclass
class KeyFrameRecord:
    stamp: tuple                      # (sec, nsec)
    estimate: np.ndarray              # node->estimate(), 4x4 float64
    odom: np.ndarray                  # 4x4 float64
    accum_distance: float
    cloud: np.ndarray                 # PointXYZI records
    node_id: int = -1
    floor_coeffs: Optional[np.ndarray] = None
    utm_coord: Optional[np.ndarray] = None
    acceleration: Optional[np.ndarray] = None
    orientation: Optional[np.ndarray] = None   # quaternion w, x, y, z


def _fmt(v: float) -> str:
    return f"{v:.6g}"   # Eigen's operator<< default: 6 significant digits (keyframe.cpp:30,33)


def _matrix_text(M: np.ndarray, shape=(4, 4)) -> str:
    """Eigen's operator<< (default IOFormat): every coefficient right-aligned to the width of the widest one, " " between columns, "\\n"
    between rows.  Pinned against the reference's own KeyFrame::save run here (tests/test_reference_code_pins.py)."""
    cells = [[_fmt(float(x)) for x in row] for row in np.asarray(M, np.float64).reshape(shape)]
    width = max(len(c) for row in cells for c in row)
    return "\n".join(" ".join(c.rjust(width) for c in row) for row in cells)


def write_pcd_binary(path: str, cloud: np.ndarray) -> None:
    """pcl::io::savePCDFileBinary for pcl::PointXYZI (keyframe.cpp:57)."""
    n = len(cloud)
    body = np.zeros((n, 4), "<f4")
    if n:
        body[:, 0], body[:, 1], body[:, 2] = cloud["x"], cloud["y"], cloud["z"]
        body[:, 3] = cloud["intensity"]
    header = ("# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
              f"WIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {n}\nDATA binary\n")
    with open(path, "wb") as fh:
        fh.write(header.encode("ascii"))
        fh.write(body.tobytes())


def read_pcd(path: str) -> np.ndarray:
    """pcl::io::loadPCDFile into pcl::PointXYZI (keyframe.cpp:139-141): ascii and binary PCD, any field order; fields other
    than x, y, z, intensity are ignored, a missing intensity reads as 0."""
    with open(path, "rb") as fh:
        raw = fh.read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, rest = line.partition(" ")
        hdr[key.upper()] = rest.split()
        if key.upper() == "DATA":
            break
    fields = hdr["FIELDS"]
    sizes = [int(s) for s in hdr["SIZE"]]
    types = hdr["TYPE"]
    counts = [int(c) for c in hdr.get("COUNT", ["1"] * len(fields))]
    n = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    mode = hdr["DATA"][0].lower()
    out = np.zeros(n, dtype=synth.POINT_XYZI_DTYPE)
    out["w"] = 1.0
    if mode == "binary":
        code = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1", ("I", 2): "<i2", ("I", 4): "<i4"}
        dt = np.dtype([(f"{name}_{i}", code[(t.upper(), s)], (c,)) for i, (name, s, t, c) in enumerate(zip(fields, sizes, types, counts))])
        rec = np.frombuffer(raw, dtype=dt, count=n, offset=pos)
        for i, name in enumerate(fields):
            if name in ("x", "y", "z", "intensity"):
                out[name] = rec[f"{name}_{i}"][:, 0].astype(np.float32)
    elif mode == "ascii":
        tab = np.loadtxt(raw[pos:].decode("ascii").splitlines(), ndmin=2) if n else np.zeros((0, sum(counts)))
        col = 0
        for name, c in zip(fields, counts):
            if name in ("x", "y", "z", "intensity"):
                out[name] = tab[:, col].astype(np.float32)
            col += c
    else:
        raise ValueError(f"unsupported PCD DATA mode '{mode}' (pcl writes keyframes with savePCDFileBinary)")
    return out


def save_keyframe(directory: str, kf: KeyFrameRecord) -> None:
    """KeyFrame::save (keyframe.cpp:21-58)."""
    os.makedirs(directory, exist_ok=True)
    lines = [f"stamp {int(kf.stamp[0])} {int(kf.stamp[1])}", "estimate", _matrix_text(kf.estimate), "odom", _matrix_text(kf.odom),
             f"accum_distance {_fmt(kf.accum_distance)}"]
    for name in ("floor_coeffs", "utm_coord", "acceleration", "orientation"):
        v = getattr(kf, name)
        if v is None:
            continue
        v = np.asarray(v, np.float64).ravel()
        if name == "orientation":       # four scalars streamed one by one (keyframe.cpp:49)
            lines.append(name + " " + " ".join(_fmt(float(x)) for x in v))
        else:                           # `->transpose()` of an Eigen vector: a 1 x N matrix, column-aligned like the 4x4s (keyframe.cpp:37-46)
            lines.append(name + " " + _matrix_text(v, (1, len(v))))
    if kf.node_id >= 0:
        lines.append(f"id {kf.node_id}")
    with open(os.path.join(directory, "data"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    write_pcd_binary(os.path.join(directory, "cloud.pcd"), kf.cloud)


def load_keyframe(directory: str) -> KeyFrameRecord:
    """KeyFrame::load (keyframe.cpp:60-145) without the g2o vertex lookup: the node id is returned, not resolved."""
    with open(os.path.join(directory, "data")) as fh:
        tok = fh.read().split()
    kf = KeyFrameRecord((0, 0), np.eye(4), np.eye(4), -1.0, np.zeros(0, synth.POINT_XYZI_DTYPE))
    i = 0

    def take(k):
        nonlocal i
        vals = [float(t) for t in tok[i:i + k]]
        i += k
        return vals

    while i < len(tok):
        t = tok[i]
        i += 1
        if t == "stamp":
            s = take(2)
            kf.stamp = (int(s[0]), int(s[1]))
        elif t == "estimate":
            M = np.array(take(16)).reshape(4, 4)
            kf.estimate = np.eye(4)
            kf.estimate[:3, :3], kf.estimate[:3, 3] = M[:3, :3], M[:3, 3]      # linear() and translation() only (:83-85)
        elif t == "odom":
            M = np.array(take(16)).reshape(4, 4)
            kf.odom = np.eye(4)
            kf.odom[:3, :3], kf.odom[:3, 3] = M[:3, :3], M[:3, 3]
        elif t == "accum_distance":
            kf.accum_distance = take(1)[0]
        elif t == "floor_coeffs":
            kf.floor_coeffs = np.array(take(4))
        elif t == "utm_coord":
            kf.utm_coord = np.array(take(3))
        elif t == "acceleration":
            kf.acceleration = np.array(take(3))
        elif t == "orientation":
            kf.orientation = np.array(take(4))
        elif t == "id":
            kf.node_id = int(take(1)[0])
    if kf.node_id < 0:
        raise ValueError(f"invalid node id in {directory}")                     # :121-125
    kf.cloud = read_pcd(os.path.join(directory, "cloud.pcd"))
    return kf
