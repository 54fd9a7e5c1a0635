"""Per-term table of the Na-guest literals of test/runtests.jl that this repo meets only to 1e-4...1e-5:
every term of the energy separately, the interpolated grid terms beside the exact (non-interpolated) pair sums at the same
position, and the sensitivity of the total to each ingredient.  CPU only (oracle + host mirror)."""
import json, math, os, sys
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import grids as G
from ceg_hip.hostmirror.probes import ProbeSystem
from oracle import oracle as O
ceg.setdir_RASPA(os.path.join(here, '..', 'golden', 'raspa'))
FF = "BoulfelfelSholl2021"
ff = ceg.parse_forcefield_RASPA(FF)


def corners(cset, probe, point, alpha=None):
    nx, ny, nz = cset.npoints
    p0, p1, r = G.interpolation_stencil(cset, (nx, ny, nz), point)
    pts = np.array([ceg.abc_to_xyz(cset, x - 1, y - 1, z - 1) for z in (p0[2], p1[2]) for y in (p0[1], p1[1]) for x in (p0[0], p1[0])])
    raw = O.points_vdw(probe, pts) if alpha is None else O.points_coulomb(probe, alpha, pts)
    lam, thr = G.vdw_scaling() if alpha is None else G.coulomb_scaling()
    return raw, pts, r, lam, thr


def interp(cset, probe, point, alpha=None, mutate=None, f32=True):
    raw, pts, r, lam, thr = corners(cset, probe, point, alpha)
    if mutate is not None:
        raw = mutate(raw.copy())
    stored = O.set_gridpoints(raw, cset.delta, lam, thr)
    if f32:
        stored = (stored.astype(np.float64) * ceg.GRID_TO_KELVIN).astype(np.float32).astype(np.float64)
    else:
        d = np.asarray(cset.delta)
        sc = np.array([1, d[0], d[1], d[2], d[0] * d[1], d[0] * d[2], d[1] * d[2], d[0] * d[1] * d[2]])
        stored = raw * sc[None, :] * lam * ceg.GRID_TO_KELVIN
    X = stored.T.reshape(64)
    return G.interpolate_from_corners(X, r, alpha is None)


def exact(probe, point, alpha=None):
    raw = O.points_vdw(probe, np.array([point])) if alpha is None else O.points_coulomb(probe, alpha, np.array([point]))
    lam = G.vdw_scaling()[0] if alpha is None else G.coulomb_scaling()[0]
    return raw[0, 0] * lam * ceg.GRID_TO_KELVIN


def report(name, fwname, pos, literal, supercell=None):
    fw = ceg.load_framework_RASPA(fwname, FF)
    cset = ceg.GridCoordinatesSetup.from_cell(fw.mat, 0.15)
    pv = ProbeSystem.build(fw, ff, "Na")
    pc = ProbeSystem.build(fw, ff)
    from ceg_hip.hostmirror.utils import find_supercell
    sc = tuple(find_supercell(fw.mat, 12.0))
    ew = ceg.initialize_ewald(fw, sc)
    na = ceg.load_molecule_RASPA("Na", "TraPPE", FF, fw)
    q = na.atomic_charge[0]
    pos = np.asarray(pos, dtype=np.float64)
    vdw = interp(cset, pv, pos)
    direct = q * interp(cset, pc, pos, ew.alpha)
    recip = ceg.compute_ewald(ew, ((na.with_positions([pos]),),))
    vdw_x, direct_x = exact(pv, pos), q * exact(pc, pos, ew.alpha)
    print(f"\n== {name}: {fwname}, Na at {pos.tolist()}  (supercell {sc}, {len(ew.kfactors)} k-vectors)")
    print(f"   framework VdW   interpolated {vdw:18.6f}   exact pair sum {vdw_x:18.6f}   interpolation error {vdw - vdw_x:10.4f}")
    print(f"   framework direct interpolated {direct:17.6f}   exact pair sum {direct_x:18.6f}   interpolation error {direct - direct_x:10.4f}")
    print(f"   reciprocal (incl. net-charge and self terms) {recip:18.6f}")
    return vdw, direct, recip, vdw_x, direct_x, (cset, pv, pc, ew, q, pos)


if __name__ == "__main__":
    from ceg_hip.hostmirror import montecarlo as M
    # --- Na in CIT-7 (runtests.jl:222-228)
    solo = [-4.728415488310421, 32.03533696753957, 2.943765448968882]
    v, d, r, vx, dx, ctx = report("baseSolo", "CIT-7", solo, -21375.116833457894)
    tail = -70.44772635984882
    total = v + d + r + tail
    print(f"   tail correction {tail:.6f} (literal, met to 1e-8)   total {total:.6f}   literal -21375.116833   residue {total + 21375.116833457894:+.4f} K")
    print(f"   with exact (non-interpolated) framework terms the total would be {vx + dx + r + tail:.6f}  ({vx + dx + r + tail + 21375.116833457894:+.4f} from the literal)")
    cset, pv, pc, ew, q, pos = ctx
    # sensitivity: zero one channel group of the Na VdW grid / the Coulomb grid at the 8 corners
    for label, cols in (("d1", [1, 2, 3]), ("d2", [4, 5, 6]), ("d3", [7])):
        def mut(raw, cols=cols):
            raw[:, cols] = 0.0
            return raw
        print(f"   zeroing {label} of the VdW corners moves VdW by {interp(cset, pv, pos, None, mut) - v:+9.4f} K;  of the Coulomb corners moves direct by "
              f"{q * interp(cset, pc, pos, ew.alpha, mut) - d:+9.4f} K")
    print(f"   Float32 storage of the corners: VdW {interp(cset, pv, pos, None, None, False) - v:+.5f} K, direct {q * interp(cset, pc, pos, ew.alpha, None, False) - d:+.5f} K")
    nxt = [-5.036, 31.876, 3.117]
    v2, d2, r2, vx2, dx2, _ = report("baseSoloNext", "CIT-7", nxt, -21795.8765195143)
    t2 = v2 + d2 + r2 + tail
    print(f"   total {t2:.6f}   literal -21795.876520   residue {t2 + 21795.8765195143:+.4f} K;  with exact framework terms {vx2 + dx2 + r2 + tail + 21795.8765195143:+.4f} K")
    # --- Na in bare CHA: origin (met to 1e-9) and the energy_grid minimum (4e-5)
    fw = ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", FF)
    a, b, c = fw.mat[:, 0], fw.mat[:, 1], fw.mat[:, 2]
    num = [int(np.floor(np.linalg.norm(x) / 0.3)) + 1 for x in (a, b, c)]
    pmin = 28 * a / num[0] + 59 * b / num[1] + 59 * c / num[2]
    v0, d0, r0, *_ = report("Na/CHA origin (runtests.jl:44-45)", "CHA_1.4_3b4eeb96", [0.0, 0.0, 0.0], None)
    print(f"   vdw {v0:.8f} (literal -11083.13758653269)   coulomb {d0 + r0:.6f} (literal -1850940.2225092095)")
    vm, dm, rm, vxm, dxm, _ = report("Na/CHA energy_grid minimum (29,60,60) (runtests.jl:49)", "CHA_1.4_3b4eeb96", pmin, -1927894.4364761321)
    print(f"   total {vm + dm + rm:.6f}   literal -1927894.436476   residue {vm + dm + rm + 1927894.4364761321:+.4f} K;  with exact framework terms "
          f"{vxm + dxm + rm + 1927894.4364761321:+.4f} K")
    # --- Ewald convergence of the restatement: exact direct sum + reciprocal sum for precision 1e-6 ... 1e-12
    from ceg_hip.hostmirror.utils import find_supercell
    fwc = ceg.load_framework_RASPA("CIT-7", FF)
    pcc = ProbeSystem.build(fwc, ff)
    nac = ceg.load_molecule_RASPA("Na", "TraPPE", FF, fwc)
    for name, p, lit, other in (("baseSolo", solo, -21375.116833457894, v + tail), ("baseSoloNext", nxt, -21795.8765195143, v2 + tail)):
        print(f"\n== {name}: Coulomb part implied by the literal (literal - VdW - tail) = {lit - other:.5f}")
        for prec in (1e-6, 1e-8, 1e-10, 1e-12):
            ewp = ceg.initialize_ewald(fwc, tuple(find_supercell(fwc.mat, 12.0)), prec)
            dd = nac.atomic_charge[0] * exact(pcc, np.asarray(p, dtype=np.float64), ewp.alpha)
            rr = ceg.compute_ewald(ewp, ((nac.with_positions([p]),),))
            print(f"   precision {prec:g}: alpha {ewp.alpha:.5f}, {len(ewp.kfactors):6d} k-vectors: direct (exact) {dd:14.5f} + reciprocal {rr:14.5f} = {dd + rr:14.5f}")
