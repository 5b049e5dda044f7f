"""what does the vendor GEMM (torch -> hipBLASLt / rocBLAS) reach on the hot shapes?  (diagnostic only, not the product path)"""
import torch, time
dev = torch.device("cuda")
shapes = [(12608, 2304, 768), (12608, 3072, 768), (12608, 768, 3072), (12608, 768, 768), (8192, 768, 768), (8192, 3072, 768),
          (8192, 30528, 768), (8192, 8192, 8192)]
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    w = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    for _ in range(3): torch.nn.functional.linear(a, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): torch.nn.functional.linear(a, w)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"vendor linear M={M} N={N} K={K}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)
