"""How fast can this GPU write?  (fill and copy of buffers far larger than the 256 MiB Infinity Cache)"""
import torch, time
dev = torch.device("cuda", 0)
for mb in (1.6, 6.8, 82, 160, 820, 4096):
    n = int(mb * 1e6) // 8
    a = torch.empty(n, dtype=torch.float64, device=dev)
    b = torch.empty(n, dtype=torch.float64, device=dev)
    for name, fn, bytes_ in (("fill (write only)", lambda: a.fill_(1.5), 8 * n), ("copy (read + write)", lambda: b.copy_(a), 16 * n)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 200 if mb < 100 else 20
        blk = torch.empty((4096, 4096), dtype=torch.float64, device=dev).normal_()
        torch.cuda.synchronize()
        torch.mm(blk, blk)            # the launches queue behind this, so the host's launch rate does not show
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        print(f"{mb:7.1f} MB {name:20s} {ms * 1e3:9.1f} us  {bytes_ / ms / 1e6:8.1f} GB/s")
