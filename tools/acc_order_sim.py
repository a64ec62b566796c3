import numpy as np, torch
rng = np.random.default_rng(0)
def run(K, nterm, M=2048, N=129, relu=True):
    A = rng.standard_normal((M, nterm*K)).astype(np.float32)
    if relu: A = np.maximum(A, 0)
    W = (rng.uniform(-1, 1, (nterm*K, N)) / np.sqrt(K)).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    scale = np.abs(ref).max()
    # (a) one sequential fp32 fma chain over all k (the MFMA 32x32x2 order: k ascending, 2 k's per instruction = chain of pairs; model: strictly sequential)
    acc = np.zeros((M, N), np.float32)
    for k in range(nterm*K):
        acc = (acc.astype(np.float64) + A[:, k:k+1].astype(np.float64) * W[k:k+1, :].astype(np.float64)).astype(np.float32)   # fma: one rounding
    # (b) per-term chains, then terms added in order
    tot = np.zeros((M, N), np.float32)
    for t in range(nterm):
        a = np.zeros((M, N), np.float32)
        for k in range(t*K, (t+1)*K):
            a = (a.astype(np.float64) + A[:, k:k+1].astype(np.float64) * W[k:k+1, :].astype(np.float64)).astype(np.float32)
        tot = tot + a
    # (c) torch CPU matmul per term + add (the oracle's dataflow)
    tt = torch.zeros(M, N)
    for t in range(nterm):
        tt = tt + torch.from_numpy(A[:, t*K:(t+1)*K]) @ torch.from_numpy(W[t*K:(t+1)*K])
    # (d) chains of 8 chunks interleaved? (two accumulators: even / odd 8-wide chunks)
    a0 = np.zeros((M, N), np.float32); a1 = np.zeros((M, N), np.float32)
    for k in range(nterm*K):
        tgt = a0 if (k // 8) % 2 == 0 else a1
        tgt[...] = (tgt.astype(np.float64) + A[:, k:k+1].astype(np.float64) * W[k:k+1, :].astype(np.float64)).astype(np.float32)
    two = a0 + a1
    def e(x):
        d = np.abs(x.astype(np.float64) - ref)
        return d.max() / scale, np.sqrt((d**2).mean()) / scale
    print(f"K={K} x {nterm} terms: one chain max {e(acc)[0]:.2e} rms {e(acc)[1]:.2e} | per-term chains {e(tot)[0]:.2e} {e(tot)[1]:.2e} | torch per-term {e(tt.numpy())[0]:.2e} {e(tt.numpy())[1]:.2e} | two interleaved chains {e(two)[0]:.2e} {e(two)[1]:.2e}")
run(129, 1); run(129, 4); run(512, 4, M=512, N=128); run(260, 1)
