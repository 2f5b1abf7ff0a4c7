import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from sapr_amd import _lib
lib = _lib.load()
dev = _lib.require_gpu()
x = torch.randn(100, 13, device=dev)
offs = torch.tensor([0, 40, 100], dtype=torch.int64, device=dev)
out = torch.zeros(13, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
rc = lib.sapr_custom_global_sum(_lib.ptr(x), _lib.ptr(offs), 2, 13, _lib.ptr(out), _lib.current_stream())
print("rc", rc, lib.sapr_last_error())
torch.cuda.synchronize()
print(out.cpu().numpy()[:3], x[:, :3].sum(0).cpu().numpy())
n = C.c_size_t(0)
print("pack bytes rc", lib.sapr_diag_pack_bytes(1, 10, 13, C.byref(n)), n.value)
