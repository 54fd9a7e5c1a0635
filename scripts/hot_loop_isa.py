#!/usr/bin/env python3
"""Instruction mix of the hot loop (the innermost loop that holds the candidate LDS read + the accumulations) of a k_culled
variant, from the ISA hipcc emits:  python scripts/hot_loop_isa.py /tmp/kern.s MODE VDWK EWK [NP] [--dump]
(produce the .s with hipcc <Makefile flags> -S --cuda-device-only -o /tmp/kern.s ceg_kernels.hip)"""
import re, sys
path, mode, vdwk, ewk = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
lines = open(path).read().split("\n")
np_ = int(sys.argv[5]) if len(sys.argv) > 5 and sys.argv[5].isdigit() else 1
start = [k for k, l in enumerate(lines) if l.startswith(f"_ZN3ceg8k_culledILi{mode}ELb0ELi{vdwk}ELi{ewk}ELi{np_}EE")][0]
end = start
while "s_endpgm" not in lines[end]:
    end += 1
body = lines[start:end]
# the hot loop = the Depth=3 loop whose header block reads the candidate record (ds_read_b128 x2 right after the header label)
hdr = [k for k, l in enumerate(body) if "This Inner Loop Header: Depth=3" in l]
loops = []
for h in hdr:
    name = None
    for back in range(1, 4):
        m = re.match(r"(\.LBB\d+_\d+):", body[h - back].strip())
        if m:
            name = m.group(1); break
    tag = "Header=" + name[2:] + " "
    blocks = [k for k, l in enumerate(body) if tag in l]
    lo, hi = min(blocks + [h]), max(blocks + [h])
    k = hi + 1
    while k < len(body) and not re.match(r"\.LBB\d+_\d+:", body[k].strip()):
        k += 1
    loops.append((lo - 3, k))
for lo, hi in loops:
    seg = [l.strip() for l in body[lo:hi] if l.strip() and not l.strip().startswith(";")]
    ins = [l.split()[0] for l in seg if not l.endswith(":") and not l.startswith(".")]
    f64 = [i for i in ins if i.startswith("v_") and "f64" in i]
    valu = [i for i in ins if i.startswith("v_")]
    if len(f64) < 20 or any(i in ("v_div_scale_f64", "v_div_fmas_f64") for i in ins):
        continue            # staging / exact-path loops
    print(f"k_culled<mode {mode}, VDWK {vdwk}, EWK {ewk}, NP {np_}> hot loop: {len(ins)} instructions; VALU {len(valu)} (FP64 {len(f64)}, other {len(valu) - len(f64)}), "
          f"SALU {sum(1 for i in ins if i.startswith('s_'))}, LDS {sum(1 for i in ins if i.startswith('ds_'))}, "
          f"scratch {sum(1 for i in ins if i.startswith('scratch_'))}, global {sum(1 for i in ins if i.startswith('global_'))}")
    if "--dump" in sys.argv:
        print("\n".join(seg))
