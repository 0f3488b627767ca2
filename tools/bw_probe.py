#!/usr/bin/env python3
"""Write / copy bandwidth at the sizes of the encoder's intermediates (roofline of the GEMM epilogues). MI355X only."""
import torch

dev = torch.device("cuda:0")


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for mb in (38, 75, 151, 302, 604):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device=dev).bfloat16()
    y = torch.empty_like(x)
    t_fill = timeit(lambda: y.fill_(1.0))
    t_copy = timeit(lambda: y.copy_(x))
    t_gelu = timeit(lambda: torch.nn.functional.gelu(x, approximate="none"))
    print(f"{mb:4d} MiB bf16: fill {t_fill:7.1f} us = {mb * 1.048576 / t_fill * 1e3:6.0f} GB/s written | "
          f"copy {t_copy:7.1f} us = {2 * mb * 1.048576 / t_copy * 1e3:6.0f} GB/s moved | "
          f"torch gelu {t_gelu:7.1f} us = {2 * mb * 1.048576 / t_gelu * 1e3:6.0f} GB/s moved", flush=True)
