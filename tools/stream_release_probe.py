"""Latency of releasing a parked stream: hipLaunchHostFunc + condition variable vs hipStreamWaitValue32 on signal memory / pinned memory."""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tscd_amd import _host, ops
hip = C.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
x = torch.zeros(1024, device=dev)
stream = torch.cuda.current_stream()
raw = ops._stream()
attr = C.c_int(0)
hip.hipDeviceGetAttribute.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
# find the enum value of hipDeviceAttributeCanUseStreamWaitValue by scanning the header order is fragile: just try the call
hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipStreamWaitValue32.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint, C.c_uint32]
hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
hip.hipHostGetDevicePointer.argtypes = [C.POINTER(C.c_void_p), C.c_void_p, C.c_uint]

def run(kind, reps=20):
    lat = []
    if kind == "signal":
        p = C.c_void_p()
        st = hip.hipExtMallocWithFlags(C.byref(p), 8, 0x2)       # hipMallocSignalMemory
        print("hipExtMallocWithFlags(signal) ->", st, hex(p.value or 0))
        if st != 0: return
        word = C.cast(p, C.POINTER(C.c_uint32))
        dptr = p
    elif kind == "pinned":
        p = C.c_void_p()
        st = hip.hipHostMalloc(C.byref(p), 64, 0)
        print("hipHostMalloc ->", st, hex(p.value or 0))
        word = C.cast(p, C.POINTER(C.c_uint32))
        dptr = p
    latch = _host.Latch()
    for i in range(1, reps + 1):
        torch.cuda.synchronize()
        if kind == "hostfunc":
            g = latch.new_generation()
            ops.call("tscd_stream_host_func", raw, _host.LATCH_WAIT_CB, latch.ticket(g))
        else:
            st = hip.hipStreamWaitValue32(raw, dptr, i, 0, 0xFFFFFFFF)     # flags 0 = Gte
            if st != 0:
                print("hipStreamWaitValue32 ->", st); return
        x.add_(1)
        ev = torch.cuda.Event(); ev.record()
        time.sleep(0.004)                                   # the stream is parked by now
        assert not ev.query()
        t0 = time.perf_counter()
        if kind == "hostfunc":
            latch.signal(g)
        else:
            word[0] = i
        while not ev.query():
            pass
        lat.append((time.perf_counter() - t0) * 1e6)
    lat.sort()
    print(f"{kind:9s}: release -> kernel done, us: min {lat[0]:.1f} median {lat[len(lat)//2]:.1f} max {lat[-1]:.1f}")

for k in ("hostfunc", "signal", "pinned"):
    try:
        run(k)
    except Exception as ex:
        print(k, "failed:", repr(ex))
