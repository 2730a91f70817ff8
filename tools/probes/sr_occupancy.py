import ctypes as C, sys
import torch
torch.zeros(1, device="cuda")
lib = C.CDLL(sys.argv[1])
for b in (68608, 65536, 40960, 35000, 32768):
    print(b, lib.loc_debug_sr_occupancy(b))
