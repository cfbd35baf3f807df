"""Does hipExtStreamCreateWithCUMask confine kernels on this stack?  Times a 1 GiB elementwise pass on masked streams."""
import ctypes as C, torch
hip = C.CDLL('libamdhip64.so')
x = torch.zeros(256 << 20, device='cuda')
def mk(every):
    if every <= 1:
        return torch.cuda.Stream()
    words = (C.c_uint32 * 8)()
    for cu in range(256):
        if cu % every == 0:
            words[cu // 32] |= 1 << (cu % 32)
    h = C.c_void_p()
    assert hip.hipExtStreamCreateWithCUMask(C.byref(h), 8, words) == 0
    return torch.cuda.ExternalStream(h.value)
for every in (1, 2, 4, 8, 32):
    s = mk(every)
    with torch.cuda.stream(s):
        for _ in range(3): x.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(10): x.add_(1.0)
        e1.record(s)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print('every %d: %.3f ms  %.2f TB/s' % (every, ms, 2 * x.numel() * 4 / ms / 1e9))
