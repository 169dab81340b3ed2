#!/usr/bin/env python3
"""The practically attainable HBM ceiling on this box (SURVEY.md §8d asks for it next to the 8 TB/s spec): device-to-device
copy and read-only / write-only streams with torch's own kernels, 1 GiB buffers (4x the 256 MiB Infinity Cache)."""
import json
import time

import torch


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    n = 1 << 28  # floats: 1 GiB
    a = torch.rand(n, device="cuda")
    b = torch.empty_like(a)
    out = {"buffer_GiB": 1.0}
    t = timed(lambda: b.copy_(a))
    out["copy_read_plus_write_TBps"] = round(2 * 4 * n / t / 1e12, 3)
    t = timed(lambda: b.fill_(1.0))
    out["write_only_TBps"] = round(4 * n / t / 1e12, 3)
    t = timed(lambda: a.sum())
    out["read_only_TBps"] = round(4 * n / t / 1e12, 3)
    t = timed(lambda: torch.add(a, b, out=b))
    out["triad_like_2r1w_TBps"] = round(3 * 4 * n / t / 1e12, 3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
