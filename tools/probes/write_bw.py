"""Streaming-write ceiling of the box: torch fill / copy of 1 GiB buffers (HIP events)."""
import torch
x = torch.empty(1 << 28, dtype=torch.float32, device="cuda")   # 1 GiB
y = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn, nbytes in (("fill 1 GiB", lambda: x.fill_(1.0), 1 << 30), ("copy 1 GiB -> 1 GiB (read + write)", lambda: y.copy_(x), 2 << 30),
                         ("sum 1 GiB (read)", lambda: x.sum(), 1 << 30)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(10):
        fn()
    ev1.record()
    ev1.synchronize()
    ms = ev0.elapsed_time(ev1) / 10
    print(f"{name}: {ms * 1e3:.1f} us, {nbytes / ms / 1e6:.0f} GB/s")
