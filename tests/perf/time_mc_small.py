"""time_mc.py's set-up and its batch-1 / batch-16 loops only (latency experiments)"""
import os, sys
here = os.path.dirname(os.path.abspath(__file__))
src = open(os.path.join(here, "time_mc.py")).read()
src = src.split("dtd = bench_device(65536, 10)")[0].replace("for nbatch, reps in ((1, 2000), (16, 1000), (1024, 200), (65536, 10)):", "for nbatch, reps in ((1, 3000), (16, 1000)):")
src = src.replace("os.path.abspath(__file__)", repr(os.path.join(here, "time_mc.py")))
exec(compile(src, "time_mc_small", "exec"))
dev.close()
