"""CPU check of the index arithmetic of tools/ubench_ring_gemm.hip (one tile, numpy; run: python tools/sim_ring_gemm.py): DMA piece mapping -> LDS image -> per-lane fragment reads -> MFMA lane semantics ->
epilogue scatter; compares with X @ W^T.  Mirrors the kernel's formulas literally."""
import numpy as np
BM, BN, BK = 256, 320, 32
MT, NT = 8, 5
A_BYTES, W_BYTES = BM * BK * 2, BN * BK * 2
NIA, NIW = A_BYTES // 8192, (W_BYTES + 8191) // 8192
SLOT = A_BYTES + NIW * 8192
K = 320
rng = np.random.default_rng(0)
X = rng.standard_normal((BM, K)).astype(np.float32)
W = rng.standard_normal((BN, K)).astype(np.float32)
Y = np.zeros((BM, BN), np.float32)
acc = np.zeros((8, NT, MT, 64, 4), np.float64)     # wave, j, i, lane, reg
for kc in range(K // BK):
    lds = np.full(SLOT // 2, np.nan, np.float32)   # halves
    for wave in range(8):
        for lane in range(64):
            for t in range(NIA):
                p = (wave + 8 * t) * 64 + lane; row = p >> 2; c = (p & 3) ^ ((row >> 1) & 2)
                dst = (wave * 1024 + t * 8192 + lane * 16) // 2
                lds[dst:dst + 8] = X[row, kc * BK + c * 8: kc * BK + c * 8 + 8]
            for t in range(NIW):
                p = (wave + 8 * t) * 64 + lane; row = p >> 2; c = (p & 3) ^ ((row >> 1) & 2)
                dst = (A_BYTES + wave * 1024 + t * 8192 + lane * 16) // 2
                lds[dst:dst + 8] = W[row, kc * BK + c * 8: kc * BK + c * 8 + 8] if row < BN else 0.0
    for wave in range(8):
        wm, wn = wave >> 2, wave & 3
        fx = np.zeros((MT, 64, 8), np.float32); fw = np.zeros((NT, 64, 8), np.float32)
        for lane in range(64):
            frow = lane & 15; fslot = (lane >> 4) ^ ((frow >> 1) & 2)
            xbase = ((wm * 128 + frow) * 4 + fslot) * 16
            wbase = A_BYTES + ((wn * 80 + frow) * 4 + fslot) * 16
            for i in range(MT):
                a = (xbase + i * 16 * 64) // 2; fx[i, lane] = lds[a:a + 8]
            for j in range(NT):
                a = (wbase + j * 16 * 64) // 2; fw[j, lane] = lds[a:a + 8]
        assert not np.isnan(fx).any() and not np.isnan(fw).any()
        # mfma(a = fw[j], b = fx[i]): A[i_][k] = a[lane = i_ + 16 * (k // 8)][k % 8]; B[k][n_] = b[lane = n_ + 16 * (k // 8)][k % 8]; D[lane][r] = D[i_ = (lane >> 4) * 4 + r][n_ = lane & 15]
        for j in range(NT):
            Am = np.zeros((16, 32), np.float32)
            for l in range(64):
                Am[l & 15, (l >> 4) * 8:(l >> 4) * 8 + 8] = fw[j, l]
            for i in range(MT):
                Bm = np.zeros((32, 16), np.float32)
                for l in range(64):
                    Bm[(l >> 4) * 8:(l >> 4) * 8 + 8, l & 15] = fx[i, l]
                D = Am.astype(np.float64) @ Bm.astype(np.float64)
                for l in range(64):
                    for r in range(4):
                        acc[wave, j, i, l, r] += D[(l >> 4) * 4 + r, l & 15]
for wave in range(8):
    wm, wn = wave >> 2, wave & 3
    for lane in range(64):
        m0 = wm * 128 + (lane & 15); n0 = wn * 80 + (lane >> 4) * 4
        for i in range(MT):
            for j in range(NT):
                Y[m0 + i * 16, n0 + j * 16:n0 + j * 16 + 4] = acc[wave, j, i, lane]
ref = X.astype(np.float64) @ W.astype(np.float64).T
print("max abs err", np.abs(Y - ref).max(), "ref scale", np.abs(ref).max())
# bank conflicts of the fragment reads under ds_read_b128's lane groups
groups = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)), list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
          list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)), list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
worst = 0
for wm in range(2):
    for i in range(MT):
        for g in groups:
            quads = {}
            for lane in g:
                frow = lane & 15; fslot = (lane >> 4) ^ ((frow >> 1) & 2)
                a = ((wm * 128 + frow) * 4 + fslot) * 16 + i * 1024
                quads.setdefault((a // 16) % 16, set()).add(a)
            worst = max(worst, max(len(v) for v in quads.values()))
print("worst b128 conflict degree", worst)
