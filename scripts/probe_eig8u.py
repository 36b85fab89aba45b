import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
g = torch.Generator(device=dev).manual_seed(1)
n = 1953125
A = torch.randn(n, 8, 8, device=dev, generator=g)
S = A + A.transpose(-1, -2)
for _ in range(30):
    N.eig_sym(S, compute_u=True, check_finite=False)
    N.eig_sym(S, compute_u=False, check_finite=False)
torch.cuda.synchronize()
