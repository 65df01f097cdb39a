"""Keyframe directory format ("next" row f4; src/hdl_graph_slam/keyframe.cpp:21-145): round trip, a hand-written file in
the reference's layout (Eigen's default matrix formatting), PCD variants."""
import os

import numpy as np
import pytest

from hdl_graph_slam_amd import synth
from hdl_graph_slam_amd.keyframe_io import KeyFrameRecord, load_keyframe, read_pcd, save_keyframe, write_pcd_binary


def _cloud(n=500, seed=0):
    rng = np.random.default_rng(seed)
    return synth.to_xyzi(rng.normal(0, 10, (n, 3)).astype(np.float32), rng.uniform(0, 255, n).astype(np.float32))


def test_round_trip(tmp_path):
    T = synth.pose_matrix([12.5, -3.25, 0.5], [0.01, -0.02, 1.2])
    kf = KeyFrameRecord((1700000000, 123456789), T, synth.pose_matrix([1, 2, 3], [0, 0, 0.1]), 42.5, _cloud(), node_id=17,
                        floor_coeffs=np.array([0.0, 0.0, 1.0, 1.7]), orientation=np.array([1.0, 0.0, 0.0, 0.0]))
    d = str(tmp_path / "000017")
    save_keyframe(d, kf)
    assert sorted(os.listdir(d)) == ["cloud.pcd", "data"]
    back = load_keyframe(d)
    assert back.stamp == kf.stamp and back.node_id == 17 and back.accum_distance == 42.5
    assert np.allclose(back.estimate, T, rtol=1e-5, atol=1e-5)          # 6 significant digits, like Eigen's operator<<
    assert np.allclose(back.floor_coeffs, kf.floor_coeffs) and back.utm_coord is None
    for f in ("x", "y", "z", "intensity"):
        assert np.array_equal(back.cloud[f], kf.cloud[f])               # the cloud itself is binary: exact


def test_reads_the_reference_layout(tmp_path):
    d = tmp_path / "000003"
    d.mkdir()
    (d / "data").write_text("stamp 1528183372 706177000\nestimate\n   0.999  -0.0436        0    1.25\n  0.0436    0.999        0   -0.5\n"
                            "       0        0        1       0\n       0        0        0       1\nodom\n1 0 0 1.2\n0 1 0 -0.4\n0 0 1 0\n0 0 0 1\n"
                            "accum_distance 1.34\nacceleration 0.1 0.2 9.8\nid 3\n")
    write_pcd_binary(str(d / "cloud.pcd"), _cloud(10))
    kf = load_keyframe(str(d))
    assert kf.node_id == 3 and kf.stamp == (1528183372, 706177000) and abs(kf.accum_distance - 1.34) < 1e-12
    assert kf.estimate[0, 3] == 1.25 and kf.estimate[1, 0] == 0.0436 and np.allclose(kf.acceleration, [0.1, 0.2, 9.8])
    assert len(kf.cloud) == 10
    (d / "data").write_text("stamp 1 2\naccum_distance 0\n")
    with pytest.raises(ValueError):
        load_keyframe(str(d))                                           # no id -> "invalid node id!!" (keyframe.cpp:121-125)


def test_pcd_ascii_and_extra_fields(tmp_path):
    p = tmp_path / "a.pcd"
    p.write_text("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z ring intensity\nSIZE 4 4 4 2 4\nTYPE F F F U F\nCOUNT 1 1 1 1 1\nWIDTH 2\nHEIGHT 1\n"
                 "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS 2\nDATA ascii\n1 2 3 7 0.5\n4 5 6 8 0.25\n")
    c = read_pcd(str(p))
    assert list(c["x"]) == [1, 4] and list(c["intensity"]) == [0.5, 0.25]
    # binary with a different field order and an extra uint16 field
    dt = np.dtype([("intensity", "<f4"), ("ring", "<u2"), ("x", "<f4"), ("y", "<f4"), ("z", "<f4")])
    rec = np.zeros(3, dt)
    rec["x"], rec["intensity"], rec["ring"] = [1, 2, 3], [9, 8, 7], [1, 2, 3]
    q = tmp_path / "b.pcd"
    q.write_bytes(b"VERSION 0.7\nFIELDS intensity ring x y z\nSIZE 4 2 4 4 4\nTYPE F U F F F\nCOUNT 1 1 1 1 1\nWIDTH 3\nHEIGHT 1\nPOINTS 3\nDATA binary\n" + rec.tobytes())
    c = read_pcd(str(q))
    assert list(c["x"]) == [1, 2, 3] and list(c["intensity"]) == [9, 8, 7]


@pytest.mark.gpu
def test_loaded_keyframes_become_resident_candidates(tmp_path):
    """load -> upload once -> the cloud's index and covariances persist on the device across detections."""
    from hdl_graph_slam_amd import _lib as L
    from hdl_graph_slam_amd.registration import RegistrationHIP
    tgt, src, T = synth.make_pair("VLP-16", 1, downsample=0.3)
    save_keyframe(str(tmp_path / "000001"), KeyFrameRecord((1, 0), np.eye(4), np.eye(4), 0.0, src, node_id=1))
    kf = load_keyframe(str(tmp_path / "000001"))
    reg = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
    cand = reg.upload(kf.cloud)
    reg.setInputTarget(tgt)
    rec1, best1 = reg.loop_match_batch([cand], [np.eye(4, dtype=np.float32)])
    reg.profile_enable(True)
    reg.profile_read(True)
    rec2, best2 = reg.loop_match_batch([cand], [np.eye(4, dtype=np.float32)])
    prof = reg.profile_read(True)
    assert prof["covariance"][1] == 0 and prof["index"][1] == 0          # nothing rebuilt for the resident candidate
    assert bytes(rec1["final_transformation"]) == bytes(rec2["final_transformation"]) and best1 == best2 == 0
    dt, dr = synth.pose_error(np.array(rec1[0]["final_transformation"]).reshape(4, 4).T, T)
    assert dt < 0.2
    reg.close()
