"""How fast can ANY kernel gather 512-byte rows at random from a table of a given size?  (torch.index_select: the
read side is the gather, the write side a stream; both counted.)  Context for the k = 128 item half of a C5 rank,
which gathers from a 51.2 GB replica: DESIGN.md section 6, round 3."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
k = 128
g = torch.Generator(device=dev)
g.manual_seed(1)
for n_rows in (1_000_000, 10_000_000, 100_000_000):
    X = torch.empty(n_rows, k, dtype=torch.float32, device=dev).normal_(generator=g)
    n_idx = 50_000_000
    idx = torch.randint(0, n_rows, (n_idx,), generator=g, device=dev)
    out = torch.empty(n_idx // 10, k, dtype=torch.float32, device=dev)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c in range(10):
            torch.index_select(X, 0, idx[c * (n_idx // 10):(c + 1) * (n_idx // 10)], out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    read = n_idx * k * 4
    print("table %6.1f GB: %d random 512-byte rows in %.1f ms: gather read %.2f TB/s (+ the same bytes written: %.2f TB/s total)"
          % (n_rows * k * 4 / 1e9, n_idx, dt * 1e3, read / dt / 1e12, 2 * read / dt / 1e12))
    del X, idx, out
    torch.cuda.empty_cache()
