import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
from hdl_graph_slam_amd import _lib as L, synth
from hdl_graph_slam_amd.registration import RegistrationHIP
import ctypes as C
rng = np.random.default_rng(0)
clouds = [synth.to_xyzi(rng.normal(0, 20, (119000, 3)).astype(np.float32), intensity=rng.random(119000).astype(np.float32)) for _ in range(40)]
reg = RegistrationHIP(L.default_params(L.HGS_NDT_OMP))
lib = L.lib()
for name, seq in (("same cloud (hot)", [0] * 30), ("rotating clouds (cold)", list(range(40)))):
    ts = []
    for i in seq:
        arr, n, stride = L.cloud_args(clouds[i])
        t = time.perf_counter()
        rc = lib.hgs_set_source(reg._h, arr.ctypes.data_as(C.c_void_p), n, stride)
        ts.append((time.perf_counter() - t) * 1e3)
        assert rc == 0
        reg.synchronize()
    print(name, "hgs_set_source p50 ms", round(float(np.median(ts[5:])), 4), flush=True)
reg.close()
