import os, sys, time, torch, numpy as np
sys.path.insert(0, os.getcwd())
import nitorch_fastmath_amd as N
dev = torch.device('cuda:0')
def cost(S):
    torch.cuda.synchronize(); t = time.perf_counter()
    N.eig_sym(S, check_finite=False, max_iter=1024); torch.cuda.synchronize()
    return time.perf_counter() - t
g = torch.Generator(device=dev).manual_seed(0)
n = 4096
a8 = torch.randn(n, 8, 8, device=dev, generator=g, dtype=torch.float64) + 8 * torch.eye(8, device=dev, dtype=torch.float64)
S = (a8[:, :3, :3] + a8[:, :3, :3].transpose(-1, -2)).contiguous()
cost(S[:64])
slow = []
for w in range(0, n, 64):
    if cost(S[w:w + 64]) > 5e-4:
        for i in range(w, w + 64):
            if cost(S[i:i + 1]) > 5e-4:
                slow.append(i)
print('slow lanes', len(slow), 'of', n, slow[:20])
np.save('gpurun_out/eig_slow.npy', S[slow].cpu().numpy())
np.save('gpurun_out/eig_all.npy', S.cpu().numpy())
for i in slow[:3]:
    for mi in (6, 8, 10, 12, 1024):
        print(i, mi, N.eig_sym(S[i:i + 1], check_finite=False, max_iter=mi).cpu().numpy())
