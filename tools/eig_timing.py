"""Phase timing of k_eig_update from a -DWISKI_EIG_TIMING build (WISKI_HIP_SO=build/libwiski_eigtiming.so): subspace iteration, H, Jacobi,
V U, T_q, residual."""
import os, sys, ctypes, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import _hip
lib = _hip.lib()
d, g, kw, kuse, kref = 3, 50, 16, 10, 18
h = 2.2 / (g - 1)
def cols(e): return [np.exp(-0.5 * (np.arange(g) * h / e[q]) ** 2) for q in range(d)]
def eig(cs):
    out = []
    for c in cs:
        idx = np.abs(np.arange(g)[:, None] - np.arange(g)[None, :]); w, V = np.linalg.eigh(c[idx]); out.append(V[:, ::-1].copy())
    return out
e0 = np.array([0.69, 0.69, 0.69]); V0 = eig(cols(e0)); c1 = cols(e0 * 1.003)
dev = "cuda"
Vin = torch.as_tensor(np.concatenate([V[:, :kw].reshape(-1) for V in V0])).to(dev)
Vref = torch.as_tensor(np.concatenate([V[:, :kref].reshape(-1) for V in V0])).to(dev)
tcol = torch.as_tensor(np.concatenate(c1)).to(dev)
gd = torch.tensor([g] * d, dtype=torch.int32, device=dev)
Vout = torch.empty_like(Vin); ev = torch.empty(d * kw, dtype=torch.float64, device=dev); resid = torch.zeros(32, dtype=torch.float64, device=dev)
Tq = torch.empty(d * 32 * 32, dtype=torch.float64, device=dev)
P = lambda t: ctypes.c_void_p(t.data_ptr())
for rep in range(3):
    rc = lib.wiski_basis_eig_update(ctypes.c_int32(d), P(gd), P(tcol), P(Vin), ctypes.c_int32(kw), ctypes.c_int32(kuse), P(Vout), P(ev), P(resid), P(Vref), ctypes.c_int32(kref), P(Tq),
                                    ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    st = resid[8:16].cpu().numpy()
    names = ["2x (applyK + MGS)", "applyK + H", "Jacobi", "rank + V U + store", "Tq", "applyK (resid)"]
    print(rc, " | ".join(f"{n} {(st[i + 1] - st[i]) / 100:.1f} us" for i, n in enumerate(names)), "| total %.1f" % ((st[6] - st[0]) / 100))
