"""The on-disk cache of ray-cast revolutions (hdl_graph_slam_amd/synth.py, HGS_SCAN_CACHE): a cached scan is the scan, the key covers every input."""
import numpy as np

from hdl_graph_slam_amd import synth


def test_cached_scan_is_the_scan_and_the_key_covers_the_inputs(tmp_path, monkeypatch):
    scene = synth.make_scene(3)
    pose = synth.pose_matrix([1.0, 0.2, 0.0], [0.0, 0.0, 0.1])
    monkeypatch.setenv("HGS_SCAN_CACHE", "")                      # off: the reference result
    ref = synth.scan(scene, "VLP-16", pose, 11)
    monkeypatch.setenv("HGS_SCAN_CACHE", str(tmp_path))
    first = synth.scan(scene, "VLP-16", pose, 11)                 # cast and stored
    files = sorted(p.name for p in tmp_path.iterdir())
    assert len(files) == 1 and files[0].endswith(".npy")
    again = synth.scan(scene, "VLP-16", pose, 11)                 # read back
    assert first.dtype == ref.dtype and np.array_equal(first, ref) and np.array_equal(again, ref)
    other_seed = synth.scan(scene, "VLP-16", pose, 12)            # another noise seed, another pose, another scene: other entries
    pose2 = pose.copy()
    pose2[0, 3] += 0.5
    other_pose = synth.scan(scene, "VLP-16", pose2, 11)
    other_scene = synth.scan(synth.make_scene(4), "VLP-16", pose, 11)
    assert len(list(tmp_path.iterdir())) == 4
    assert not np.array_equal(other_seed, ref) and len(other_pose) != 0 and len(other_scene) != 0
    (tmp_path / files[0]).write_bytes(b"truncated")               # a damaged entry is cast again, not trusted
    assert np.array_equal(synth.scan(scene, "VLP-16", pose, 11), ref)
