import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes, os
print("avail", torch.cuda.is_available())
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
def maps():
    return sorted(set(l.split()[-1] for l in open("/proc/self/maps") if "amdhip" in l or "hsa-runtime" in l))
print("before", maps())
from dicp_amd import _lib
lib = _lib.load()
print("after", maps())
hip = ctypes.CDLL("libamdhip64.so.7")
print("hip handle maps", maps())
hip.hipGetLastError.restype = ctypes.c_int
print("last err", hip.hipGetLastError())
n = ctypes.c_int(0); print("count rc", hip.hipGetDeviceCount(ctypes.byref(n)), n.value)
from dicp_amd import _ops
y = torch.rand(2, 20, 6, device="cuda")
print("stale err before:", hip.hipGetLastError())
out = torch.empty(2, 32, 4, device="cuda")
rc = lib.dicp_pack_target(0, ctypes.c_void_p(y.data_ptr()), 2, 20, 6, ctypes.c_void_p(out.data_ptr()), 32, None)
print("pack rc", rc)
torch.cuda.synchronize()
print("out", out[0, :2], out[0, 20:22])
