# what the vendor library reaches at the encoder's GEMM shapes (reference point only; not used by the product)
import torch, time
torch.manual_seed(0)
dev = "cuda:0"
shapes = [("enc qkv", 384000, 3840, 1280), ("enc out", 384000, 1280, 1280), ("enc fc1", 384000, 5120, 1280), ("enc fc2", 384000, 1280, 5120),
          ("cross kv", 384000, 2560, 1280), ("dec qkv", 1280, 3840, 1280), ("dec DxD", 1280, 1280, 1280), ("dec fc1", 1280, 5120, 1280), ("dec fc2", 1280, 1280, 5120)]
for name, M, N, K in shapes:
    a = torch.randn(M, K, device=dev, dtype=torch.float16)
    w = torch.randn(N, K, device=dev, dtype=torch.float16)
    for _ in range(3): c = a @ w.t()
    torch.cuda.synchronize()
    n = 10 if M > 10000 else 200
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): c = a @ w.t()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / n
    print(f"{name:9s} M={M} N={N} K={K}: {us:9.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s", flush=True)
    del a, w, c
