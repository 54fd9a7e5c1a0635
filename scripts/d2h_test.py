import ctypes, time, numpy as np, torch
hip = ctypes.CDLL("libamdhip64.so")
n = 8*256**3
dev = torch.empty(n, dtype=torch.float32, device="cuda").fill_(1.0)
torch.cuda.synchronize()
host = np.empty(n, dtype=np.float32)
host[:] = 0   # touch pages
def t(f, name):
    t0=time.perf_counter(); f(); torch.cuda.synchronize(); dt=time.perf_counter()-t0; print(f"{name}: {dt*1e3:.1f} ms ({n*4/dt/1e9:.1f} GB/s)"); return dt
hip.hipMemcpy.argtypes=[ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
t(lambda: hip.hipMemcpy(host.ctypes.data, dev.data_ptr(), n*4, 2), "hipMemcpy D2H pageable")
t(lambda: hip.hipMemcpy(host.ctypes.data, dev.data_ptr(), n*4, 2), "hipMemcpy D2H pageable (2nd)")
hip.hipHostRegister.argtypes=[ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint]
t(lambda: print("rc", hip.hipHostRegister(host.ctypes.data, n*4, 0)), "hipHostRegister 537MB")
t(lambda: hip.hipMemcpy(host.ctypes.data, dev.data_ptr(), n*4, 2), "hipMemcpy D2H registered")
t(lambda: hip.hipMemcpy(host.ctypes.data, dev.data_ptr(), n*4, 2), "hipMemcpy D2H registered (2nd)")
hip.hipHostUnregister.argtypes=[ctypes.c_void_p]
t(lambda: hip.hipHostUnregister(host.ctypes.data), "hipHostUnregister")
pinned = torch.empty(n, dtype=torch.float32, pin_memory=True)
t(lambda: pinned.copy_(dev), "D2H into torch pinned")
t(lambda: pinned.copy_(dev), "D2H into torch pinned (2nd)")
hp = pinned.numpy()
t(lambda: np.copyto(host, hp), "host memcpy pinned->pageable 1 thread")
