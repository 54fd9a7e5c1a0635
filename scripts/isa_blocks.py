"""Basic blocks of one kernel in a device-only assembly listing (hipcc --cuda-device-only -S), with instruction counts per class:
    python scripts/isa_blocks.py /tmp/pairs.s <substring of the mangled kernel name> [--dump LBBx_y]"""
import re, sys
lines = open(sys.argv[1]).read().splitlines()
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] or ".Lfunc_end" in lines[i])
body = lines[start + 1:end + 1]
dump = sys.argv[4] if len(sys.argv) > 4 and sys.argv[3] == "--dump" else None
blk = "entry"
stats = {blk: {}}
order = [blk]
for l in body:
    mm = re.match(r"^(\.LBB\d+_\d+):", l)
    if mm:
        blk = mm.group(1); stats[blk] = {}; order.append(blk); continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    if dump == blk:
        print(l)
    st = stats[blk]
    def inc(k): st[k] = st.get(k, 0) + 1
    inc("n")
    op = t.split()[0]
    if op.startswith("scratch_"): inc("scratch")
    elif op.startswith("v_readlane") or op.startswith("v_writelane"): inc("lane")
    elif op.startswith("v_"):
        inc("valu")
        if "f64" in op: inc("f64")
    elif op.startswith("s_swappc"): inc("call")
    elif op.startswith("s_cbranch") or op.startswith("s_branch"): inc("br")
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"): inc("wait")
    elif op.startswith("s_"): inc("salu")
    elif op.startswith("ds_"): inc("ds")
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"): inc("vmem")
if not dump:
    for b in order:
        if stats[b].get("n"):
            print(b, stats[b])
