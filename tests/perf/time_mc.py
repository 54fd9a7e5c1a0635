"""BASELINE config 5 at the granularity the reference runs it: latency of ONE movement_energy evaluation on the
device-resident Monte-Carlo state (ceg_mc_trial: before + after in one launch), and of batches of 16 and 1024 trial
placements, with the device-side update (ceg_mc_accept) between moves; beside it the CPU oracle (C, one thread) doing the
same three sums for one placement.  CHA + Na framework (1107 atoms, 0.15 A grids built by the HIP kernels), 64 CO2 guests."""
import os, sys, time, tempfile
here = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.join(here, '..', '..', 'crystalenergygrids.jl_amd'), os.path.join(here, '..', '..')]
import numpy as np
import ceg_hip as ceg
from ceg_hip import workloads as W
from ceg_hip.hostmirror import montecarlo as M
from ceg_hip.energy import DeviceMonteCarlo
from oracle import oracle as O

golden = os.path.join(here, '..', 'golden', 'raspa')
tmp = tempfile.mkdtemp(prefix="ceg_mc_")
os.makedirs(os.path.join(tmp, "raspa"))
for sub in ("forcefield", "molecules", "structures"):
    os.symlink(os.path.join(golden, sub), os.path.join(tmp, "raspa", sub))
ceg.setdir_RASPA(os.path.join(tmp, "raspa"))
FF = "BoulfelfelSholl2021"
nguest = int(sys.argv[1]) if len(sys.argv) > 1 else 64
co2 = ceg.load_molecule_RASPA("CO2", "TraPPE", FF)
base = np.asarray(co2.position, dtype=np.float64).reshape(-1, 3)
fw = ceg.load_framework_RASPA("CHA_1.4_3b4eeb96_Na_11812", FF)
rng = np.random.default_rng(0)
centers = (W._random_atoms_min_sep(nguest, 1.0, 0.14, rng)) @ fw.mat.T        # fractional, >= ~4 A apart
t0 = time.perf_counter()
mc = M.setup_montecarlo("CHA_1.4_3b4eeb96_Na_11812", FF, [co2.with_positions(c + base) for c in centers])
print(f"# setup_montecarlo (3 grids of {tuple(int(d) + 1 for d in mc.coulomb.csetup.dims)} points built on the GPU, written, parsed): {time.perf_counter() - t0:.1f} s")
M.baseline_energy(mc) if nguest <= 16 else M.compute_ewald_mc(mc)
dev = DeviceMonteCarlo(mc)
mols = [(i, j) for i, kind in enumerate(mc.positions) for j in range(len(kind))]
natoms = sum(len(ids) for *_x, ids, _p in mc.molecules())
print(f"# {len(mols)} molecules, {natoms} guest atoms, {len(mc.ewald.kfactors)} k-vectors, MC cell = 1 x 1 x 1 CHA cell")

def bench(nbatch, reps):
    idx = mols[7 % len(mols)]
    cur = mc.positions[idx[0]][idx[1]]
    trial = cur[None] + rng.uniform(-0.5, 0.5, (nbatch, 1, 3))
    dev.trial(idx, trial)
    t = time.perf_counter()
    for k in range(reps):
        e = dev.trial(idx, trial)
        if k % 2 == 0:
            dev.accept(idx, trial[0])
    dt = (time.perf_counter() - t) / reps
    return dt, e

def bench_device(nbatch, reps, own_stream=True, sync_each=True):
    """ceg_mc_trial_device: trials and rows resident on the GPU, on a stream of the caller; completion by one stream synchronisation
    per call (sync_each) or one at the end of ``reps`` enqueued calls"""
    import torch
    idx = mols[7 % len(mols)]
    cur = mc.positions[idx[0]][idx[1]]
    d_trial = torch.tensor(cur[None] + rng.uniform(-0.5, 0.5, (nbatch, 1, 3)), dtype=torch.float64, device="cuda")
    d_rows = torch.empty((nbatch + 1, 4), dtype=torch.float64, device="cuda")
    st = torch.cuda.Stream() if own_stream else torch.cuda.default_stream()
    torch.cuda.synchronize()
    for _ in range(3):
        dev.trial_device(idx, d_trial.data_ptr(), nbatch, d_rows.data_ptr(), st.cuda_stream)
    st.synchronize()
    t = time.perf_counter()
    for k in range(reps):
        dev.trial_device(idx, d_trial.data_ptr(), nbatch, d_rows.data_ptr(), st.cuda_stream)
        if sync_each:
            st.synchronize()
    st.synchronize()
    return (time.perf_counter() - t) / reps

if os.environ.get("CEG_TIME_MC_ONLY_BIG"):          # profiling runs: the large batch alone
    dt = bench(65536, 20)[0]
    print(f"GPU  batch  65536: {dt * 1e6:9.1f} us per call")
    dtd = bench_device(65536, 20)
    print(f"   (the same on the null stream: {bench_device(65536, 20, own_stream=False) * 1e6:9.1f} us; 20 calls enqueued, one synchronisation: {bench_device(65536, 20, sync_each=False) * 1e6:9.1f} us per call)")
    fl_ = 65537 * (3 * 1368 * 16.0 + 1368 * 10.0 + 3 * 51 * 40.0 + 3 * 189 * 47.0 + 60 * 72.0 + 1500.0)
    print(f"GPU  batch  65536, trials and rows resident on the device (ceg_mc_trial_device): {dtd * 1e6:9.1f} us per call = {fl_ / dtd / 78.6e12:.3f} of the FP64 vector peak")
    dev.close()
    sys.exit(0)
for nbatch, reps in ((1, 2000), (16, 1000), (1024, 200), (65536, 10)):
    dt, e = bench(nbatch, reps)
    print(f"GPU  batch {nbatch:6d}: {dt * 1e6:9.1f} us per call (trial launch + every second call an accept) = {dt * 1e6 / nbatch:9.3f} us per trial placement")

dtd = bench_device(65536, 10)
print(f"GPU  batch  65536, trials and rows resident on the device (ceg_mc_trial_device): {dtd * 1e6:9.1f} us per call = {dtd * 1e6 / 65536:9.4f} us per trial placement")

# where the wave-per-placement kernels (three launches: k_mcw_frame / k_mcw_ewald / k_mcw_pairs) take over from the
# workgroup-per-placement kernel (one launch: k_mc_trial): both forced on the same batches
print("# batch: workgroup-per-placement kernel | wave-per-placement kernels (us per call)")
for nbatch, reps in ((64, 400), (128, 400), (256, 400), (512, 300), (1024, 200), (2048, 100), (4096, 100), (16384, 30), (65536, 10)):
    os.environ["CEG_HIP_MC_WAVE_MIN"] = "1000000000"
    t_group = bench(nbatch, reps)[0]
    os.environ["CEG_HIP_MC_WAVE_MIN"] = "0"
    t_wave = bench(nbatch, reps)[0]
    del os.environ["CEG_HIP_MC_WAVE_MIN"]
    print(f"GPU  batch {nbatch:6d}: {t_group * 1e6:9.1f} | {t_wave * 1e6:9.1f}")

# roofline.  Algorithmic work of ONE placement of a 3-atom CO2 among 64 CO2 (192 guest atoms), 1368 k-vectors:
# interpolation 3 atoms x 2 grids x 256 B gathered = 1.5 KB and ~1500 flops; pair sum 3 x 189 tests x 47 + in-cutoff rules ~ 30 kflop;
# reciprocal 3 x 1368 x 16 + 1368 x 10 + tables ~ 85 kflop: ~0.12 Mflop and 2 KB per placement.  At batch 1 the launch is TWO
# workgroups (current position + the trial) on 256 CUs: 15 us of kernel + 14 us of launch / completion latency against a
# speed-of-light of 0.12 Mflop / 78.6 TFLOP/s = 1.5 ns -- latency bound by four orders of magnitude, which is why the per-placement
# cost keeps falling up to batches of ~10^4 placements (the figure to read is the large-batch one).
dt_big = bench(65536, 10)[0]
fl = 65537 * (3 * 1368 * 16.0 + 1368 * 10.0 + 3 * 51 * 40.0 + 3 * 189 * 47.0 + 60 * 72.0 + 1500.0)
print(f"roofline movement_energy, batch 65536, PER CALL (host arrays in and out: 4.7 MB H2D + 2.1 MB D2H + three launches inside the call): "
      f"~{fl / 65537 / 1e3:.0f} kflop/placement -> {fl / dt_big / 1e12:.2f} TFLOP/s = {fl / dt_big / 78.6e12:.3f} of the FP64 vector peak; "
      f"batch 1: latency bound (2 workgroups on 256 CUs)")
print(f"roofline movement_energy, batch 65536, KERNELS (inputs resident): the same {fl / 1e9:.2f} Gflop / (sum of the average durations of k_mcw_frame + "
      f"k_mcw_ewald + k_mcw_pairs in a run of the large batch alone: scripts/profile_mc_big.sh) / 78.6 TFLOP/s")

# CPU: the oracle's three sums for the same molecule, one thread, amortised over 2000 placements (no per-call overhead)
idx = mols[7 % len(mols)]
cur = mc.positions[idx[0]][idx[1]]
n = 2000
trial = cur[None] + rng.uniform(-0.5, 0.5, (n, 1, 3))
ids = mc.ffidx[idx[0]]
t = time.perf_counter()
pair = O.single_contribution_vdw(mc, idx, trial, nthreads=1)
t_pair = time.perf_counter() - t
t = time.perf_counter()
for a, ix in enumerate(ids):
    O.interpolate_points(mc.grids[ix - 1], trial[:, a], nthreads=1)
    O.interpolate_points(mc.coulomb, trial[:, a], nthreads=1)
t_int = time.perf_counter() - t
t = time.perf_counter()
O.reciprocal_energies(mc.ewald, co2, trial, nthreads=1)
t_rec = time.perf_counter() - t
print(f"CPU oracle, 1 thread, per placement: pairs {t_pair / n * 1e6:.1f} us + interpolation {t_int / n * 1e6:.1f} us + reciprocal {t_rec / n * 1e6:.1f} us "
      f"= {(t_pair + t_int + t_rec) / n * 1e6:.1f} us")
t = time.perf_counter()
for k in range(20):
    M.movement_energy(mc, idx, trial[k])
print(f"Python host mirror ceg_hip.hostmirror.montecarlo.movement_energy: {(time.perf_counter() - t) / 20 * 1e6:.0f} us per placement")
dev.close()
