#!/usr/bin/env python3
"""Round 5: what the memory system gives plain streams on this MI355X (no engine code): fill, copy, read -- the ceilings the
write-heavy Gym-rows legs are read against."""
import torch
dev = torch.device("cuda:0")
n = 1 << 29                                     # 4 GiB of doubles
x = torch.empty(n, dtype=torch.float64, device=dev)
y = torch.empty(n, dtype=torch.float64, device=dev)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / reps


t = timed(lambda: x.zero_());             print(f"fill   (write 4 GiB)          {n * 8 / t / 1e12:.2f} TB/s written")
t = timed(lambda: x.fill_(1.5));          print(f"fill_  (write 4 GiB)          {n * 8 / t / 1e12:.2f} TB/s written")
t = timed(lambda: y.copy_(x));            print(f"copy   (read 4 + write 4 GiB) {2 * n * 8 / t / 1e12:.2f} TB/s moved")
t = timed(lambda: x.sum());               print(f"sum    (read 4 GiB)           {n * 8 / t / 1e12:.2f} TB/s read")
t = timed(lambda: torch.add(x, 1.0, out=y)); print(f"add    (read 4 + write 4 GiB) {2 * n * 8 / t / 1e12:.2f} TB/s moved")
