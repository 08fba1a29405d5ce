import torch, time
dev = torch.device("cuda:0")
x = torch.randn(160000, 16, device=dev)
lin = torch.nn.Linear(16, 10).to(dev)
for lib in ("default", "cublas", "cublaslt"):
    if lib != "default":
        torch.backends.cuda.preferred_blas_library(lib)
    for _ in range(5):
        y = lin(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        y = lin(x)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(lib, "host us/call", (t1 - t0) / 200 * 1e6, "total us/call", (t2 - t0) / 200 * 1e6)
