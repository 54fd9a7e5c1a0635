"""Start-up of a GPU consumer from a CACHED grid (the reference's common case, "Retrieved ... grid": raspa.jl:426-438):
  (a) round 2: parse_grid (file -> host array, scaled on the host) + GridInterpolator(g) (upload + node-major copy)
  (b) round 3: GridInterpolator.from_file (file -> pinned ring -> device, scaled and transposed there)
on the 0.15 A Ar / CHA grid (218 x 204 x 190 points, 270 MB) in the page cache; and the plan-creation share of repeated one-shot
calls with the image cache (CEG_HIP_TRACE=1 stamps go to stderr)."""
import os, sys, time, tempfile
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import workloads as W, grids as G
from ceg_hip.interp import GridInterpolator

W.use_fixture_dir()
ff = ceg.parse_forcefield_RASPA("BoulfelfelSholl2021")
fw = ceg.load_framework_RASPA("CHA_1.4_3b4eeb96", "BoulfelfelSholl2021")
tmp = tempfile.mkdtemp(dir=os.environ.get("TMPDIR", "/tmp"))
path = os.path.join(tmp, "ar.grid")
ceg.create_grid_vdw(path, fw, ff, 0.15, "Ar")
print(f"grid file {os.path.getsize(path) / 1e6:.1f} MB")
pts = np.random.default_rng(0).uniform(0, 25, (1000, 3))
for rep in range(3):
    t0 = time.perf_counter(); eg = ceg.parse_grid(path, False, fw.mat); t1 = time.perf_counter(); it = GridInterpolator(eg); a = it(pts); t2 = time.perf_counter()
    it.close(); del eg
    t3 = time.perf_counter(); it2 = GridInterpolator.from_file(path, False, mat=fw.mat); b = it2(pts); t4 = time.perf_counter()
    it2.close()
    assert np.array_equal(a, b)
    print(f"rep {rep}: parse_grid {1e3*(t1-t0):7.1f} ms + GridInterpolator(g) {1e3*(t2-t1):7.1f} ms = {1e3*(t2-t0):7.1f} ms    |    "
          f"from_file {1e3*(t4-t3):7.1f} ms   ({(t2-t0)/(t4-t3):.1f}x)", flush=True)
os.remove(path)
# plan creation inside repeated one-shot calls: the K + 1 grids of a setup on the roofline framework
import ctypes as C
lib = ceg._abi.load_library()
ws = {a: W.roofline_workload(a, 63) for a in ("C_co2", "O_co2", "Ar")}
def stats():
    h, m, e = C.c_int64(), C.c_int64(), C.c_int64(); lib.ceg_image_cache_stats(C.byref(h), C.byref(m), C.byref(e)); return h.value, m.value
lib.ceg_release_cached_buffers()
for label, env in (("image cache on", None), ("image cache off", "0")):
    if env is None: os.environ.pop("CEG_HIP_IMAGE_CACHE", None)
    else: os.environ["CEG_HIP_IMAGE_CACHE"] = env
    for rnd in range(2):
        for a in ("C_co2", "O_co2", "Ar"):
            w = ws[a]
            t = time.perf_counter(); G.build_vdw_array(w.probe_vdw, w.cset); dt = time.perf_counter() - t
            print(f"{label}, round {rnd}: one-shot ceg_grid_vdw {a:6s} 64^3 x 11664 atoms {1e3*dt:6.2f} ms wall   cache (hits, misses) = {stats()}", flush=True)
        t = time.perf_counter(); G.build_coulomb_array(w.probe_coulomb, w.alpha, w.cset); dt = time.perf_counter() - t
        print(f"{label}, round {rnd}: one-shot ceg_grid_coulomb        64^3 x 11664 atoms {1e3*dt:6.2f} ms wall   cache (hits, misses) = {stats()}", flush=True)
