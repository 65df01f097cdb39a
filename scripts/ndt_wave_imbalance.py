"""How unequal are the four waves of a k_ndt_pass block?  CPU-only estimate on the bench clouds (HDL-64E pair at the ground-truth pose, DIRECT7,
resolution 1.0): per wave and tile the cell loop runs max-over-lanes(valid cells) times; the waves of a block meet at the barriers of a queue grab
(1 to 8 tiles).  Output quoted in DESIGN.md section 9."""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__file__), '..'))
from hdl_graph_slam_amd import workloads, synth
S = workloads.make_loop_closure_set("HDL-64E", 0, 4, n_distinct=4)
tgt = np.asarray(S.target)
def xyz(c):
    c = np.asarray(c)
    if c.dtype.names: return np.stack([c['x'], c['y'], c['z']], 1).astype(np.float64)
    return c[:, :3].astype(np.float64)
T = xyz(tgt); res = 1.0
cell = np.floor(T / res).astype(np.int64)
keys, inv, cnt = np.unique(cell, axis=0, return_inverse=True, return_counts=True)
valid = {tuple(k) for k, n in zip(keys, cnt) if n >= 6}
print("target cells", len(keys), "valid", len(valid))
offs = [(0,0,0),(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)]
def morton(p):
    q = np.floor((p - p.min(0)) / 0.05).astype(np.uint64)
    def spread(v):
        v &= np.uint64(0x1fffff)
        v = (v | (v << np.uint64(32))) & np.uint64(0x1f00000000ffff)
        v = (v | (v << np.uint64(16))) & np.uint64(0x1f0000ff0000ff)
        v = (v | (v << np.uint64(8))) & np.uint64(0x100f00f00f00f00f)
        v = (v | (v << np.uint64(4))) & np.uint64(0x10c30c30c30c30c3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x1249249249249249)
        return v
    return spread(q[:,0]) | (spread(q[:,1]) << np.uint64(1)) | (spread(q[:,2]) << np.uint64(2))
for ci in range(2):
    P = xyz(S.candidates[ci]); Tg = S.T_gt[ci]
    order = np.argsort(morton(P)); P = P[order]
    X = P @ Tg[:3,:3].T + Tg[:3,3]
    c0 = np.floor(X / res).astype(np.int64)
    npts = len(X)
    counts = np.zeros(npts, int)
    for o in offs:
        cc = c0 + np.array(o)
        counts += np.fromiter(((tuple(k) in valid) for k in cc), bool, npts)
    pad = (-npts) % 256
    cw = np.concatenate([counts, np.zeros(pad, int)]).reshape(-1, 4, 64)   # tile, wave, lane
    trips = cw.max(2)                                                   # per tile per wave
    anyc = trips > 0
    CELL, FIXED, RED = 377, 500, 1100
    cost = FIXED + trips * CELL + anyc * RED
    print(f"cand {ci}: mean cells/pt {counts.mean():.2f}  mean trips/wave {trips.mean():.2f}  lane efficiency {counts.sum()/ (trips.sum()*64):.2f}")
    for chunk in (1, 2, 4, 8):
        nt = (cost.shape[0] // chunk) * chunk
        cc = cost[:nt].reshape(-1, chunk, 4).sum(1)
        print(f"  chunk {chunk}: barrier idle {(1 - cc.mean(1).sum() / cc.max(1).sum()) * 100:.1f} %")
