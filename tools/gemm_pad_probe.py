#!/usr/bin/env python3
"""Development probe: the channel mixer's two GEMMs (bias epilogue) + GELU at RecNeXt-A3's hidden widths (1.875 x dim: 120 / 240 / 480 / 960) against
the same with the hidden width zero-padded to the next multiple of 64 / 128 (a function-preserving inference transform: gelu(0) = 0 meets zero weights)."""
import torch

dev = torch.device("cuda:0")


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for dim, hw in ((64, 56), (128, 28), (256, 14), (512, 7)):
    m = 256 * hw * hw
    x = torch.randn(m, dim, device=dev, dtype=torch.bfloat16)
    row = {"dim": dim, "M": m}
    for hid in sorted({int(dim * 1.875), -(-int(dim * 1.875) // 64) * 64, dim * 2}):
        w1 = torch.randn(hid, dim, device=dev, dtype=torch.bfloat16) * 0.1
        b1 = torch.randn(hid, device=dev, dtype=torch.bfloat16)
        w2 = torch.randn(dim, hid, device=dev, dtype=torch.bfloat16) * 0.1
        b2 = torch.randn(dim, device=dev, dtype=torch.bfloat16)
        f = lambda: torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(x, w1, b1)), w2, b2)
        row[f"hidden_{hid}_us"] = round(timed(f), 1)
    print(row)
