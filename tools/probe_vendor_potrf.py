"""Yardstick for the few-fits regime: what AMD's own libraries need for the factor + inverse of
ONE (or eight) float64 matrices - `torch.linalg.cholesky` + `torch.cholesky_inverse` (rocSOLVER potrf / potri through
PyTorch; nothing in the product links them) - at N = 2048 / 4096, batch 1 and 8, next to this build's whole
value+gradient evaluation at the same sizes (tools/probe_latency.py: Gram + factor + inverse + K_y^-1 + trace)."""
import sys
import time

import torch

dev = torch.device("cuda", 0)
torch.manual_seed(0)
print(f"torch {torch.__version__}, {torch.cuda.get_device_name(0)}")
print(f"{'N':>6} {'batch':>5} | cholesky ms | cholesky_inverse ms | both ms   (float64, median of 7 after 2 warm-up calls)")
for N in (2048, 4096):
    for nb in (1, 8):
        A = torch.randn(nb, N, N, dtype=torch.float64, device=dev)
        K = A @ A.transpose(1, 2) / N + torch.eye(N, dtype=torch.float64, device=dev)
        t_c, t_i = [], []
        for it in range(9):
            torch.cuda.synchronize()
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            L = torch.linalg.cholesky(K)
            e1.record()
            Kinv = torch.cholesky_inverse(L)
            e2.record()
            torch.cuda.synchronize()
            if it >= 2:
                t_c.append(e0.elapsed_time(e1))
                t_i.append(e1.elapsed_time(e2))
        t_c.sort(); t_i.sort()
        c, i = t_c[len(t_c) // 2], t_i[len(t_i) // 2]
        err = float((Kinv[0] @ K[0] - torch.eye(N, dtype=torch.float64, device=dev)).abs().max())
        print(f"{N:6d} {nb:5d} | {c:11.3f} | {i:19.3f} | {c + i:8.3f}   (max |K^-1 K - I| = {err:.1e}; N^3 FLOP at {nb * N**3 / ((c + i) * 1e-3) / 1e12:.2f} TFLOP/s)", flush=True)
