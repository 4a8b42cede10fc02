"""hipMalloc / hipFree / hipMemset cost against size (what a one-shot update_div_restarts call pays for its arena)."""
import ctypes as C, time
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipDeviceSynchronize()
p = C.c_void_p()
hip.hipMalloc(C.byref(p), 1 << 20); hip.hipFree(p)
for rep in range(2):
    for mb in (8, 32, 60, 64, 65, 80, 100, 110, 128, 200, 256, 512):
        t0 = time.perf_counter(); hip.hipMalloc(C.byref(p), mb << 20); t1 = time.perf_counter()
        hip.hipMemset(p, 0, mb << 20); hip.hipDeviceSynchronize(); t2 = time.perf_counter()
        hip.hipFree(p); t3 = time.perf_counter()
        print(f"{mb:4d} MiB: hipMalloc {1e3 * (t1 - t0):7.3f} ms, memset+sync {1e3 * (t2 - t1):7.3f} ms, hipFree {1e3 * (t3 - t2):7.3f} ms", flush=True)
