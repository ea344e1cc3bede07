import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open('/proc/self/maps') if 'amdhip' in l or 'libhsa' in l or 'librccl' in l})
if order == 'torch_first':
    import torch
    print('avail', torch.cuda.is_available(), torch.cuda.device_count())
    if torch.cuda.is_available():
        x = torch.ones(4, device='cuda'); print(x.sum().item())
    from lightcurver_amd import _lib
    c = _lib.Context(0); print('ctx ok', c.device_info())
else:
    from lightcurver_amd import _lib
    c = _lib.Context(0); print('ctx ok', c.device_info())
    import torch
    print('avail', torch.cuda.is_available(), torch.cuda.device_count())
    if torch.cuda.is_available():
        x = torch.ones(4, device='cuda'); print(x.sum().item())
print(maps())
print({k: v for k, v in os.environ.items() if 'VISIBLE' in k or 'HSA' in k or 'HIP' in k})
