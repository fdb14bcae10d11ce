"""How long does rocSOLVER take to invert a dense f64 matrix of the coarse-space sizes (getrf + getri)?"""
import ctypes as C, sys, time
import torch
rs = C.CDLL("librocsolver.so.0")
rb = C.CDLL("librocblas.so")
h = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(h)) == 0
for n in [int(a) for a in sys.argv[1:]] or [2000, 4000, 8000, 16000]:
    A = torch.rand((n, n), dtype=torch.float64, device="cuda") + n * torch.eye(n, dtype=torch.float64, device="cuda")
    A0 = A.clone()
    ipiv = torch.zeros(n, dtype=torch.int32, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    for rep in range(2):
        A.copy_(A0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r1 = rs.rocsolver_dgetrf(h, n, n, C.c_void_p(A.data_ptr()), n, C.c_void_p(ipiv.data_ptr()), C.c_void_p(info.data_ptr()))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        r2 = rs.rocsolver_dgetri(h, n, C.c_void_p(A.data_ptr()), n, C.c_void_p(ipiv.data_ptr()), C.c_void_p(info.data_ptr()))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        err = float((A @ A0 - torch.eye(n, dtype=torch.float64, device="cuda")).abs().max()) if n <= 8000 else -1
        print(f"n={n} rep={rep}: getrf {1e3*(t1-t0):.1f} ms (rc {r1}), getri {1e3*(t2-t1):.1f} ms (rc {r2}), info {int(info)}, |A^-1 A - I| {err:.1e}", flush=True)
