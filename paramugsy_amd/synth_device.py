"""Synthetic profile-pair batches generated on the GPU (torch), for benches and full-size tests.

dp.synth_batch (numpy) needs about six minutes of host time for BASELINE.json's headline batch (100 000 pairs of 8 rows x 4 096
columns: 2 x 410 M packed columns); this module makes a batch of the same construction -- A's consensus uniform ACGT, B's
consensus A's resampled to lb columns, cyclically shifted for a fraction of the pairs, with substitutions; every row copies
its consensus or a random base, or is a gap -- with torch's generator on the device, in pieces of a few million columns, and
copies the packed columns to host arrays (the C ABI takes host pointers).  Seeded and repeatable on one machine; NOT the same
bytes as dp.synth_batch draws for the same seed (another generator), which is fine: every check compares the HIP path and the
oracle on the same arrays.  Nothing here is on the product's path.
"""
from __future__ import annotations

import numpy as np

from .dp import DpInputs


def synth_batch_device(seed: int, la, lb, rows_a: int, rows_b: int, device="cuda", sub_rate: float = 0.08, row_noise: float = 0.1,
                       gap_col_rate: float = 0.05, shift_rate: float = 0.3, piece_columns: int = 1 << 25) -> DpInputs:
    import torch
    la = np.asarray(la, dtype=np.int64)
    lb = np.asarray(lb, dtype=np.int64)
    n = len(la)
    off_a = np.concatenate([[0], np.cumsum(la)]).astype(np.int64)
    off_b = np.concatenate([[0], np.cumsum(lb)]).astype(np.int64)
    cols_a = np.empty((int(off_a[-1]), 8), dtype=np.uint8)
    cols_b = np.empty((int(off_b[-1]), 8), dtype=np.uint8)
    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(int(seed) & ((1 << 62) - 1))

    def counts(cons, rows):
        acc = torch.zeros(cons.numel(), dtype=torch.int64, device=dev)
        for _ in range(rows):
            base = torch.randint(0, 4, cons.shape, dtype=torch.int64, device=dev, generator=gen)
            u = torch.rand(cons.shape, device=dev, generator=gen)
            r = torch.where(u < row_noise, base, cons)
            # the top of the same draw decides the gap (independent of the noise decision for rates well below 1)
            r = torch.where(u > 1.0 - gap_col_rate, torch.full_like(r, 4), r)
            acc += torch.ones_like(acc) << (r << 3)
        return acc.view(torch.uint8).reshape(-1, 8)

    k0 = 0
    while k0 < n:
        k1 = k0 + 1
        while k1 < n and max(off_a[k1 + 1] - off_a[k0], off_b[k1 + 1] - off_b[k0]) <= piece_columns:
            k1 += 1
        a0, a1, b0, b1 = int(off_a[k0]), int(off_a[k1]), int(off_b[k0]), int(off_b[k1])
        A, B, m = a1 - a0, b1 - b0, k1 - k0
        la_t = torch.from_numpy(la[k0:k1]).to(dev)
        lb_t = torch.from_numpy(lb[k0:k1]).to(dev)
        cons_a = torch.randint(0, 4, (A,), dtype=torch.int64, device=dev, generator=gen)
        if B > 0:
            pid = torch.repeat_interleave(torch.arange(m, device=dev), lb_t)
            start_b = torch.cumsum(lb_t, 0) - lb_t
            start_a = torch.cumsum(la_t, 0) - la_t
            pos = torch.arange(B, device=dev) - start_b[pid]
            shift = torch.where(torch.rand(m, device=dev, generator=gen) < shift_rate,
                                torch.randint(1, 6, (m,), device=dev, generator=gen), torch.zeros(m, dtype=torch.int64, device=dev))
            lbp = torch.clamp(lb_t, min=1)[pid]
            pos = (pos + shift[pid]) % lbp
            lap = la_t[pid]
            src = start_a[pid] + torch.minimum((pos * lap) // lbp, torch.clamp(lap - 1, min=0))
            if A > 0:
                cons_b = cons_a[torch.clamp(src, max=A - 1)]
            else:
                cons_b = torch.randint(0, 4, (B,), dtype=torch.int64, device=dev, generator=gen)
            del pid, pos, lbp, lap, src
            subs = torch.rand(B, device=dev, generator=gen) < sub_rate
            cons_b = torch.where(subs, torch.randint(0, 4, (B,), dtype=torch.int64, device=dev, generator=gen), cons_b)
            cols_b[b0:b1] = counts(cons_b, rows_b).cpu().numpy()
            del cons_b, subs
        if A > 0:
            cols_a[a0:a1] = counts(cons_a, rows_a).cpu().numpy()
        k0 = k1
    return DpInputs(cols_a, off_a, cols_b, off_b)
