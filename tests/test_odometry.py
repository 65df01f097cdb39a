"""ScanMatchingOdometry (mirror of scan_matching_odometry_nodelet.cpp:165-262): keyframe logic against a scripted
registration, and a short synthetic VLP-16 stream through the CPU oracle engine (config 1 of BASELINE.json)."""
import numpy as np

import oracle as O
from hdl_graph_slam_amd import synth, workloads
from hdl_graph_slam_amd.odometry import ScanMatchingOdometry


class _Scripted:
    """Registration stub: align() returns guess * step (a constant per-sweep motion) and records the call sequence."""
    def __init__(self, step, converge=True):
        self.step, self.converge, self.log = step, converge, []

    def setInputTarget(self, c):
        self.log.append(("target", id(c)))

    def setInputSource(self, c):
        self.log.append(("source", id(c)))

    def align(self, guess):
        self.log.append(("align", np.array(guess, np.float32)))
        T = (np.asarray(guess, np.float64) @ self.step).astype(np.float32)

        class R:
            converged = self.converge

            def matrix(self_inner):
                return T
        return R()


def test_keyframe_switch_and_guess_chain():
    step = synth.pose_matrix([0.1, 0, 0], [0, 0, 0])
    reg = _Scripted(step)
    od = ScanMatchingOdometry(reg, keyframe_delta_trans=0.25, keyframe_delta_angle=10.0, keyframe_delta_time=1e9)
    clouds = [np.zeros((4, 4), np.float32) for _ in range(6)]
    odoms = [od.matching(0.1 * i, c) for i, c in enumerate(clouds)]
    assert np.allclose(odoms[0], np.eye(4))
    # every sweep advances 0.1 m; the keyframe switches when the translation EXCEEDS 0.25 m, i.e. at the third match
    xs = [o[0, 3] for o in odoms]
    assert np.allclose(xs, [0.0, 0.1, 0.2, 0.3, 0.4, 0.5], atol=1e-6)
    aligns = [e for e in reg.log if e[0] == "align"]
    # guesses: I, 0.1, 0.2 against keyframe 0; after the switch at 0.3 the guess restarts from identity
    assert np.allclose([a[1][0, 3] for a in aligns], [0.0, 0.1, 0.2, 0.0, 0.1], atol=1e-6)
    assert od.num_keyframes == 2
    targets = [e for e in reg.log if e[0] == "target"]
    assert targets[1][1] == id(clouds[3])     # the sweep that crossed the threshold becomes the new keyframe


def test_non_converged_frame_is_ignored():
    reg = _Scripted(synth.pose_matrix([0.1, 0, 0], [0, 0, 0]), converge=False)
    od = ScanMatchingOdometry(reg)
    od.matching(0.0, np.zeros((4, 4), np.float32))
    out = od.matching(0.1, np.zeros((4, 4), np.float32))
    assert np.allclose(out, np.eye(4)) and np.allclose(od.prev_trans, np.eye(4))


def test_vlp16_stream_with_oracle_engine_tracks_ground_truth():
    stream = workloads.make_odometry_stream("VLP-16", scene_seed=1, n_scans=5, speed=4.0, downsample=0.25)
    p = O.default_params(O.HGS_FAST_GICP)
    od = ScanMatchingOdometry(O.OracleRegistration(p), keyframe_delta_trans=0.5, keyframe_delta_angle=0.15, keyframe_delta_time=1e9)
    est = [od.matching(t, c) for t, c in zip(stream.stamps, stream.scans)]
    gt0 = np.linalg.inv(stream.poses[0])
    # sparse 16-beam sweeps slide a little along the ground plane: ~10 % of each 0.4 m step (the CPU oracle engine's own
    # accuracy on this scene, not the wrapper's); what is asserted is that the chain of guesses and keyframes tracks
    drift = [synth.pose_error(T_est, gt0 @ pose) for T_est, pose in zip(est, stream.poses)]
    assert all(dt < 0.05 * (i + 1) and dr < 0.005 for i, (dt, dr) in enumerate(drift)), drift
    assert np.linalg.norm(est[-1][:3, 3]) > 1.3      # 4 steps of 0.4 m
    assert od.num_keyframes >= 2


import pytest


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["FAST_GICP", "FAST_VGICP", "NDT_OMP"])
def test_hip_stream_follows_the_oracle_stream(method):
    """The odometry caller on the HIP engine against the same caller on the CPU oracle: same keyframes, same poses."""
    from hdl_graph_slam_amd.registrations import select_registration_method
    stream = workloads.make_odometry_stream("VLP-16", scene_seed=1, n_scans=5, speed=4.0, downsample=0.25)
    reg = select_registration_method({"registration_method": method, "reg_resolution": 1.0}, device_id=0)
    p = O.HgsParams()
    for name, _ in O.HgsParams._fields_:
        setattr(p, name, getattr(reg.params, name))
    kf = dict(keyframe_delta_trans=0.5, keyframe_delta_angle=0.15, keyframe_delta_time=1e9)
    a, b = ScanMatchingOdometry(reg, **kf), ScanMatchingOdometry(O.OracleRegistration(p), **kf)
    for t, c in zip(stream.stamps, stream.scans):
        Ta, Tb = a.matching(t, c), b.matching(t, c)
        dt, dr = synth.pose_error(Ta, Tb)
        assert dt < 1e-5 and dr < 1e-5, (method, t, dt, dr)
    assert a.num_keyframes == b.num_keyframes
    reg.close()


@pytest.mark.gpu
@pytest.mark.parametrize("filter_name,method_id", [("VOXELGRID", 1), ("APPROX_VOXELGRID", 2)])
def test_hip_stream_with_device_voxelgrid_downsample(filter_name, method_id):
    """Row a3: the nodelet's VOXELGRID / APPROX_VOXELGRID downsample on the device (resident result registered in place) gives the
    same odometry as downsampling with the oracle's pcl::VoxelGrid / pcl::ApproximateVoxelGrid restatement on the host and
    uploading the result."""
    from hdl_graph_slam_amd.odometry import make_downsample
    from hdl_graph_slam_amd.registrations import select_registration_method
    stream = workloads.make_odometry_stream("VLP-16", scene_seed=1, n_scans=4, speed=4.0)
    kf = dict(keyframe_delta_trans=0.5, keyframe_delta_angle=0.15, keyframe_delta_time=1e9)
    reg = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
    a = ScanMatchingOdometry(reg, downsample=make_downsample(reg, filter_name, 0.25), **kf)
    pf = O.default_prefilter_params()
    pf.use_distance_filter, pf.outlier_removal_method, pf.downsample_resolution, pf.downsample_method = 0, 0, 0.25, method_id
    reg2 = select_registration_method({"registration_method": "FAST_GICP"}, device_id=0)
    b = ScanMatchingOdometry(reg2, downsample=lambda c: synth.to_xyzi(O.prefilter(c, pf)[:, :3]), **kf)
    for t, c in zip(stream.stamps, stream.scans):
        Ta, Tb = a.matching(t, c), b.matching(t, c)
        assert np.array_equal(Ta, Tb)
    reg.close(), reg2.close()


@pytest.mark.gpu
def test_hdl64_ndt_stream_with_the_kitti_keyframe_rule():
    """Config 3: a 50-sweep HDL-64E stream (KITTI prefilter: 0.25 m voxel grid, hdl_graph_slam_kitti.launch:27-28) through the
    odometry caller with the reference's NDT and the keyframe rule of hdl_graph_slam_kitti.launch:41-43 (5 m / 2 rad /
    10000 s), every align to convergence.  The HIP stream equals the stream on the oracle's exact-sum mode bit for bit —
    poses, keyframe switches — and stays within 1e-3 m / 1e-3 rad of the stream on the oracle's serial (ndt_omp) sum."""
    import parity_checks as PC
    from hdl_graph_slam_amd.registrations import select_registration_method
    stream = workloads.make_odometry_stream("HDL-64E", scene_seed=2, n_scans=50, speed=4.0, downsample=0.25)
    reg = select_registration_method({"registration_method": "NDT_OMP", "reg_resolution": 1.0}, device_id=0)
    p = O.HgsParams()
    for name, _ in O.HgsParams._fields_:
        setattr(p, name, getattr(reg.params, name))
    kf = dict(keyframe_delta_trans=5.0, keyframe_delta_angle=2.0, keyframe_delta_time=10000.0)
    hip = ScanMatchingOdometry(reg, **kf)
    exact = ScanMatchingOdometry(O.OracleRegistration(p).set_ndt_sum_mode(1), **kf)
    serial = ScanMatchingOdometry(O.OracleRegistration(p), **kf)
    worst, iters = (0.0, 0.0), []
    for t, c in zip(stream.stamps, stream.scans):
        Ta, Tb, Tc = hip.matching(t, c), exact.matching(t, c), serial.matching(t, c)
        assert np.array_equal(Ta, Tb), (t, synth.pose_error(Ta, Tb))
        dt, dr = synth.pose_error(Ta, Tc)
        worst = (max(worst[0], dt), max(worst[1], dr))
        if hip.last_result is not None:
            iters.append(int(hip.last_result.iterations))
    assert worst[0] <= 1e-3 and worst[1] <= 1e-3, worst
    assert hip.num_keyframes == exact.num_keyframes == serial.num_keyframes and hip.num_keyframes >= 2
    gt = np.linalg.inv(stream.poses[0]) @ stream.poses[-1]
    dt, dr = synth.pose_error(hip.keyframe_pose @ hip.prev_trans, gt)
    PC.write_report("ndt_stream_hdl64_kitti.json", {"sweeps": len(stream.scans), "points_per_sweep": int(np.mean([len(c) for c in stream.scans])),
                                                     "keyframes": hip.num_keyframes, "iterations_mean": float(np.mean(iters)), "iterations_max": int(max(iters)),
                                                     "max_diff_vs_serial_m": worst[0], "max_diff_vs_serial_rad": worst[1],
                                                     "drift_vs_ground_truth_m": float(dt), "drift_vs_ground_truth_rad": float(dr)})
    reg.close()
