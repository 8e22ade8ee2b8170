"""Phase stamps of k_eig_update (block 0) from a -DWISKI_EIG_TIMING build of csrc/spectral_basis.hip:
   hipcc ... -DWISKI_EIG_TIMING -> WISKI_HIP_SO=<that .so> python tools/eig_phases.py [drift]
Stamps: 0 start, 1 after the subspace iterations, 2 after H = V^T K V, 3 after Jacobi, 4 after V <- V U, 6 after the residual products, 5 end."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from online_gp_amd import _hip
drift = float(sys.argv[1]) if len(sys.argv) > 1 else 1.001
d, g, kw, kuse = 3, 50, 22, 16
h = 2.2 / (g - 1)
idx = np.abs(np.arange(g)[:, None] - np.arange(g)[None, :])
cols = lambda e: [np.exp(-0.5 * (np.arange(g) * h / e[q]) ** 2) for q in range(d)]
ells = np.array([0.69, 0.69, 0.69])
V0 = [np.linalg.eigh(c[idx])[1][:, ::-1][:, :kw] for c in cols(ells)]
c1 = cols(ells * drift)
dev = "cuda"
Vin = torch.as_tensor(np.concatenate([V.reshape(-1) for V in V0])).to(dev)
tc = torch.as_tensor(np.concatenate(c1)).to(dev)
gd = torch.tensor([g] * d, dtype=torch.int32, device=dev)
Vout = torch.empty_like(Vin); ev = torch.empty((d, kw), dtype=torch.float64, device=dev); resid = torch.zeros(32, dtype=torch.float64, device=dev)
lib = _hip.lib()
for name, args in (("two iterations", None), ("adaptive (resid_ok 1.25e-10)", (0, 1.25e-10))):
    for _ in range(3):
        base = [ctypes.c_int32(d), _hip.dptr(gd), _hip.dptr(tc), _hip.dptr(Vin), ctypes.c_int32(kw), ctypes.c_int32(kuse), _hip.dptr(Vout), _hip.dptr(ev), _hip.dptr(resid), None, ctypes.c_int32(0), None]
        if args is None:
            lib.wiski_basis_eig_update(*base, _hip.stream_ptr("cuda:0"))
        else:
            lib.wiski_basis_eig_update_adaptive(*base, ctypes.c_int32(args[0]), ctypes.c_double(args[1]), _hip.stream_ptr("cuda:0"))
    torch.cuda.synchronize()
    s = resid.cpu().numpy()
    st = s[8:15]
    print(name, "resid", s[:3], "us from start:", [round(float(v - st[0]) / 100.0, 1) for v in st], "Jacobi sweeps (pass 0, 1):", s[16:18], "off2/dg2 at the last check:", s[18:20])
