"""qconc inside a Python process, before / after torch has initialised the device (does torch change how many streams
of the process run concurrently?)"""
import ctypes, os, sys
here = os.path.dirname(os.path.abspath(__file__))
L = ctypes.CDLL(os.path.join(here, "libqconc.so"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
    x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
    print("torch initialised", flush=True)
L.qconc_main()
