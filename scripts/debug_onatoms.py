import sys, os
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..'), os.path.join(here, '..', 'tests')]
import numpy as np
from ceg_hip import workloads as W
from ceg_hip.plan import GridPlan
from util import synthetic_probes, grid_points
from oracle import oracle as O
np.set_printoptions(precision=6, linewidth=200)
L = 30.0; mat = np.diag([L, L, L]); cset = W.grid_setup_with_dims(mat, (15, 15, 15))
pos = np.array([[4.0, 6.0, 8.0], [10.0, 10.0, 10.0], [20.0, 2.0, 28.0], [11.3, 17.7, 5.1], [0.0, 0.0, 0.0]])
kinds = np.array([1, 2, 4, 2, 1]); q = np.array([1.0, -1.0, 0.5, 0.0, -0.7])
pv, pc = synthetic_probes(mat, pos, kinds, q)
pts = grid_points(cset)
ref = O.points_vdw(pv, pts)
plan = GridPlan(cset, pv, pc, 0.265)
print("images", plan.num_images)
got = plan.eval_points("vdw", pts, 2)
bad = np.where(~np.isclose(got[:, 0], ref[:, 0], rtol=1e-9, atol=1e-12, equal_nan=True))[0]
print(len(bad), "bad points; indices", bad[:40])
print("tiles", sorted(set(bad // 64)))
for b in bad[:5]:
    print(b, pts[b], "got", got[b, 0], "ref", ref[b, 0])
    one = plan.eval_points("vdw", pts[b:b + 1], 2)
    print("   single-point culled:", one[0, 0])
# whole tile alone
t = bad[0] // 64
tile = pts[64 * t:64 * t + 64]
alone = plan.eval_points("vdw", tile, 2)
print("tile alone mismatches:", int((~np.isclose(alone[:, 0], ref[64 * t:64 * t + 64, 0], rtol=1e-9, atol=1e-12, equal_nan=True)).sum()))
print("tile box", tile.min(0), tile.max(0))
