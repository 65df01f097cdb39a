import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle as O
from hdl_graph_slam_amd import synth, _lib as L
from hdl_graph_slam_amd.registration import RegistrationHIP
np.set_printoptions(precision=3, linewidth=200)
tgt, src, T = synth.make_pair("VLP-16", 2, downsample=0.1)
p = O.default_params(O.HGS_NDT_OMP); p.resolution = 1.0
def hip(params):
    q = L.HgsParams()
    for name, _ in L.HgsParams._fields_: setattr(q, name, getattr(params, name))
    return RegistrationHIP(q)
e, o = hip(p), O.OracleRegistration(p)
p6 = np.array([T[0, 3] + 0.1, T[1, 3], T[2, 3], 3.14, 3.13, 3.1])
def run(s):
    for r in (e, o): r.setInputTarget(tgt); r.setInputSource(s)
    se, ge, He = e.ndt_derivatives(p6); so, go, Ho = o.ndt_derivatives(p6)
    return (se - so) / abs(so), (ge - go) / np.abs(go).max(), (He - Ho) / np.abs(Ho).max()
ds, dg, dH = run(src)
print("score", ds); print("g", dg); print("H"); print(dH)
# bisect to the offending points
lo, hi = 0, len(src)
while hi - lo > 1:
    mid = (lo + hi) // 2
    _, dg, _ = run(src[lo:mid])
    if np.abs(dg).max() > 1e-12: hi = mid
    else: lo = mid
    print(lo, hi, np.abs(dg).max())
print("point", lo, src[lo])
