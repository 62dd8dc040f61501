"""Pinned host -> device copy rate of this box (DIAGNOSTIC): one stream and several streams, step-sized and large buffers."""
import time, torch
dev = torch.device("cuda", 0)
for mb in (3.7, 16, 64, 256):
    n = int(mb * 2**20 / 4)
    for S in (1, 2, 4):
        hs = [torch.empty(n, dtype=torch.float32).pin_memory() for _ in range(S)]
        ds = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(S)]
        st = [torch.cuda.Stream() for _ in range(S)]
        reps = max(4, int(400 / mb))
        for _ in range(2):
            for k in range(S):
                with torch.cuda.stream(st[k]): ds[k].copy_(hs[k], non_blocking=True)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            for k in range(S):
                with torch.cuda.stream(st[k]): ds[k].copy_(hs[k], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"{mb:6.1f} MB x {S} stream(s): {reps * S * n * 4 / dt / 1e9:6.1f} GB/s  ({dt / (reps * S) * 1e6:.0f} us per copy)")
