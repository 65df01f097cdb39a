import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O
from hdl_graph_slam_amd import synth, _lib as L
from hdl_graph_slam_amd.registration import RegistrationHIP
np.set_printoptions(precision=17, linewidth=250)
tgt, src, T = synth.make_pair("VLP-16", 2, downsample=0.1)
p = O.default_params(O.HGS_NDT_OMP); p.resolution = 1.0
def hip(params):
    q = L.HgsParams()
    for name, _ in L.HgsParams._fields_: setattr(q, name, getattr(params, name))
    return RegistrationHIP(q)
p6 = np.array([T[0, 3] + 0.1, T[1, 3], T[2, 3], 3.14, 3.13, 3.1])
for search in (O.HGS_DIRECT1, O.HGS_DIRECT7):
    p.neighbor_search = search
    e, o = hip(p), O.OracleRegistration(p)
    s = src[10225:10226]
    for r in (e, o): r.setInputTarget(tgt); r.setInputSource(s)
    se, ge, He = e.ndt_derivatives(p6); so, go, Ho = o.ndt_derivatives(p6)
    print("search", search, "score", se.hex() if hasattr(se,'hex') else float(se).hex(), float(so).hex())
    print("g hip", [float(v).hex() for v in ge]); print("g ora", [float(v).hex() for v in go])
    print("H diff rel"); print((He - Ho) / np.abs(Ho).max())
