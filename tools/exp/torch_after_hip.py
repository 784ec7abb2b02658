import sys
sys.path[:0]=['/root/repo','/root/repo/oracle','/root/repo/tools']
import torch
print("count", torch.cuda.device_count())
from libde265_amd import backend
print("hip devices", backend.device_count())
d = backend.Decoder()
try:
    print("avail", torch.cuda.is_available())
    torch.cuda.init()
    x = torch.zeros(4, device="cuda")
    print("ok", x.device)
except Exception as e:
    print("ERR", e)
import os
print({k:v for k,v in os.environ.items() if "VISIBLE" in k or "HIP" in k or "ROC" in k})
