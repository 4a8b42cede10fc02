import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, nmf_gpu_amd as ng
hip = C.CDLL("libamdhip64.so")
def bench(fn, reps=10):
    for _ in range(4): fn()          # clocks ramp over the first few launches
    hip.hipDeviceSynchronize()
    t0=time.perf_counter()
    for _ in range(reps): fn()
    hip.hipDeviceSynchronize()
    return (time.perf_counter()-t0)/reps
rng = np.random.default_rng(0)
for (M,N,K) in ((4096,65536,256),(8192,16384,512),(4096,4096,4096)):
    W = ng.Matrix(rng.random((M,K),dtype=np.float32)).to_device()
    H = ng.Matrix(rng.random((K,N),dtype=np.float32)).to_device()
    Z = ng.Matrix(rows=M, cols=N).to_device()
    WtZ = ng.Matrix(rows=K, cols=N).to_device()
    ZHt = ng.Matrix(rows=M, cols=K).to_device()
    t = bench(lambda: ng.matrix_multiply(W,H,Z));      print(f"({M},{N},{K}) W*H      NN: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.1f} TF")
    t = bench(lambda: ng.matrix_multiply_AtB(W,Z,WtZ)); print(f"({M},{N},{K}) W'*Z     TN: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.1f} TF")
    t = bench(lambda: ng.matrix_multiply_ABt(Z,H,ZHt)); print(f"({M},{N},{K}) Z*H'     NT: {t*1e3:8.3f} ms  {2*M*N*K/t/1e12:6.1f} TF")
    del W,H,Z,WtZ,ZHt
M,N,K=8192,16384,512
for path in (ng.PATH_UNFUSED,):
    s = ng.Solver(M,N,K,path=path)
    s.upload(np.asfortranarray(rng.random((M,K),dtype=np.float32)), np.asfortranarray(rng.random((K,N),dtype=np.float32)), np.asfortranarray(rng.random((M,N),dtype=np.float32)))
    s.iterate(2); s.sync(); t0=time.perf_counter(); s.iterate(10); s.sync(); dt=(time.perf_counter()-t0)/10
    extra = f" kernels H/W {s.time_piece(2,3):.3f}/{s.time_piece(3,3):.3f} ms" if path == ng.PATH_FUSED else ""
    print(f"K=512 path {s.path} iteration ({M},{N},{K}): {dt*1e3:.2f} ms  {8*M*N*K/dt/1e12:.1f} TF effective{extra}")
    s.close()
