"""Register / scratch / LDS use of the kernels of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel:
    python scripts/resources.py crystalenergygrids.jl_amd/csrc/ceg_pairs.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-c", "-o", "/dev/null", src,
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
demangle = lambda n: subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+(\w[\w ]*\w)(?: \[[\w/]+\])?: (\S+)", line)
    if m and cur is not None:
        cur[m.group(1)] = m.group(2)
for r in rows:
    name = re.sub(r"\(.*", "", demangle(r["name"]).replace("(anonymous namespace)::", "").replace("void ", ""))
    if flt and flt not in name:
        continue
    print(f"{name:48s} VGPRs {r.get('VGPRs', '?'):>4} AGPRs {r.get('AGPRs', '?'):>3} SGPRs {r.get('TotalSGPRs', r.get('SGPRs', '?')):>4} scratch {r.get('ScratchSize', '?'):>5} "
          f"occupancy {r.get('Occupancy', '?'):>2} LDS {r.get('LDS Size', '?'):>6} spills S {r.get('SGPRs Spill', '?')} V {r.get('VGPRs Spill', '?')}")
