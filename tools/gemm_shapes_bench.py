"""The four encoder GEMM shapes through the default dispatch (persistent 256x256 kernel) under the current environment
(e.g. VITCAP_GEMM_GROUP_N); used by tools/group_sweep.sh (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.gemm_bench import bench
from vitcap_amd import _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 36928
row = []
for name, N, K, act, f32, res in (('qkv', 2304, 768, L.ACT_NONE, 0, False), ('fc1', 3072, 768, L.ACT_GELU_ERF, 0, False),
                                  ('proj', 768, 768, L.ACT_NONE, 1, True), ('fc2', 768, 3072, L.ACT_NONE, 1, True)):
    ms = min(bench(M, N, K, act, f32, res, 0, iters=30)[0] for _ in range(2))
    row.append('%s %.1f' % (name, ms * 1e3))
print('GROUP_N=%-4s M=%d us: %s' % (os.environ.get('VITCAP_GEMM_GROUP_N', '-'), M, ' | '.join(row)), flush=True)
