"""Which engine moves a D2H copy when the call comes from torch vs. straight hipMemcpyAsync? (diagnostic)"""
import ctypes, sys, torch
hip = ctypes.CDLL('libamdhip64.so')
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
mode = sys.argv[1]
n = 25 << 20
dev = torch.arange(n, dtype=torch.int32, device='cuda')
pin = torch.empty(n, dtype=torch.int32, pin_memory=True)
raw = ctypes.c_void_p()
assert hip.hipHostMalloc(ctypes.byref(raw), ctypes.c_size_t(n * 4), ctypes.c_uint(0)) == 0
own = ctypes.c_void_p()
assert hip.hipStreamCreateWithFlags(ctypes.byref(own), ctypes.c_uint(1)) == 0
side = torch.cuda.Stream()
torch.cuda.synchronize()
for rep in range(3):
    dev.add_(1)
    ev = torch.cuda.Event(); ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        if mode == 'torch':
            pin.copy_(dev, non_blocking=True)
        elif mode == 'hip':
            rc = hip.hipMemcpyAsync(pin.data_ptr(), dev.data_ptr(), n * 4, 2, side.cuda_stream)
            assert rc == 0, rc
        elif mode == 'hostmalloc':
            rc = hip.hipMemcpyAsync(raw, dev.data_ptr(), n * 4, 2, side.cuda_stream)
            assert rc == 0, rc
        elif mode == 'ownstream':
            torch.cuda.synchronize()
            rc = hip.hipMemcpyAsync(raw, dev.data_ptr(), n * 4, 2, own)
            assert rc == 0, rc
            hip.hipStreamSynchronize(own)
    side.synchronize()
    print(mode, rep, int(pin[12345]), ctypes.cast(raw, ctypes.POINTER(ctypes.c_int))[12345])
