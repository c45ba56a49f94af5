"""What the vendor library reaches on this chip at the engine's big GEMM shape, on the same kind of random data
(dev tool: calibrates 'how far from a tuned kernel', cdna_hip_programming.md rule 10/25).  f16 in, f32 accumulate."""
import torch
dev = torch.device('cuda:0')
for M, N, K in ((149100, 512, 2304), (149100, 256, 2304), (58800, 256, 2304), (9500, 256, 2304), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    b = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3):
        c = a @ b.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        c = a @ b.t()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print('torch.matmul f16 (hipBLASLt) M=%d N=%d K=%d: %.3f ms, %.0f TFLOP/s executed (= %.0f TF/s algorithmic if it were one of the 3 split products)'
          % (M, N, K, ms, 2.0 * M * N * K / ms / 1e9, 2.0 * M * N * K / ms / 1e9 / 3), flush=True)
