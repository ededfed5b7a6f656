import ctypes, os, subprocess, sys
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
CHILD = r'''
import ctypes, os, sys
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
if order == "lib_first":
    lib = ctypes.CDLL(os.path.join(%r, "rfnet_amd", "librfops.so"))
    print("after librfops:", maps())
    import torch
    print("after torch:", maps())
else:
    import torch
    print("after torch:", maps())
    lib = ctypes.CDLL(os.path.join(%r, "rfnet_amd", "librfops.so"))
    print("after librfops:", maps())
print("cuda available:", torch.cuda.is_available())
try:
    x = torch.zeros(4, device="cuda"); print("tensor ok")
except Exception as e:
    print("tensor failed:", e)
print("rf_device_check:", lib.rf_device_check())
''' % (ROOT, ROOT)
for order in ("torch_first", "lib_first"):
    print("=====", order)
    r = subprocess.run([sys.executable, "-c", CHILD, order], capture_output=True, text=True)
    print(r.stdout, r.stderr[-300:])
