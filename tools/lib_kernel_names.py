"""Yardstick, not product: which hipBLASLt (Tensile) kernels torch.mm / addmm pick on the hot shapes.  Run under
rocprofv3 --kernel-trace --stats (tools/prof_script.sh); the kernel names encode macro tile, depthU, wave tile, LDS buffering.
    python tools/lib_kernel_names.py [M]"""
import sys
import torch

M = int(sys.argv[1]) if len(sys.argv) > 1 else 295424
for N, K in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    b = torch.rand(N, device='cuda').to(torch.bfloat16)
    o = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    for _ in range(6):
        torch.addmm(b, a, w.t(), out=o)
    torch.cuda.synchronize()
