import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from covo_mpc_amd.controllers._core import SamplingCore
N = 65536
core = SamplingCore(N, 32, 0.01, 1.0, device="cuda:0")
A = np.random.default_rng(0).normal(size=(128, 128))
L = torch.from_numpy(np.linalg.cholesky(A @ A.T / 128 + 0.05 * np.eye(128)).astype(np.float32)).cuda()
am = torch.zeros(128, device="cuda")
for _ in range(5):
    core.noise_gemm_philox(L, am, (1, 2))
torch.cuda.synchronize()
d = core.a.view(-1)[:1024].view(torch.int32).cpu().numpy().reshape(512, 2)
t0 = d[:, 0].min()
st, en = (d[:, 0] - t0) * 10, (d[:, 1] - t0) * 10   # ns
print("start ns: min %d max %d ; end ns: min %d max %d ; median dur %d" % (st.min(), st.max(), en.min(), en.max(), np.median(en - st)))
import collections
print("start histogram (us):", np.histogram(st / 1e3, bins=8)[0], np.histogram(st / 1e3, bins=8)[1].round(1))
