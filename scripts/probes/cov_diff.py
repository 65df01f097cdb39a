import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O
from hdl_graph_slam_amd import synth, _lib as L
from hdl_graph_slam_amd.registration import RegistrationHIP
tgt, src, T = synth.make_pair("HDL-32E", 5)
e = RegistrationHIP(L.default_params(L.HGS_FAST_GICP))
e.setInputTarget(tgt)
got = e.target_covariances(len(tgt)).astype(np.float64)
ref = O.covariances(tgt, 20)
rel = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
bad = np.flatnonzero(rel > 5e-6)
print("n", len(tgt), "bad", len(bad), bad[:40])
xyz = synth.xyz_of(tgt)
from scipy.spatial import cKDTree
tree = cKDTree(xyz.astype(np.float64))
d, idx = tree.query(xyz[bad[:5]].astype(np.float64), k=22)
for b, dd in zip(bad[:5], d):
    print(b, "rel", rel[b], "d19..21", dd[18:22], "pt", xyz[b])
