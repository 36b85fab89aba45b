import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nitorch_fastmath_amd import reduce as R
dev = torch.device('cuda:0')
x = torch.randn(1 << 22, int(sys.argv[1]) if len(sys.argv) > 1 else 256, device=dev)
for _ in range(6):
    R.median(x, dim=1)
torch.cuda.synchronize()
