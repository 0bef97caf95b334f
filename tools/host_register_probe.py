"""Does pinning the caller's pageable buffer in place (hipHostRegister) beat staging it through pinned buffers?  488 MB numpy array:
register / async copy / unregister times.  usage: host_register_probe.py"""
import ctypes
import time

import numpy as np

hip = ctypes.CDLL("libamdhip64.so")
n, d = 1_000_000, 128
x = np.random.RandomState(0).standard_normal((n, d)).astype(np.float32)
nbytes = x.nbytes
dev = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(dev), ctypes.c_size_t(nbytes)) == 0
hip.hipDeviceSynchronize()
for flags, name in ((0, "hipHostRegisterDefault"), (1, "hipHostRegisterPortable"), (0x8, "hipHostRegisterReadOnly (if supported)")):
    for rep in range(3):
        t0 = time.perf_counter()
        rc = hip.hipHostRegister(ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), ctypes.c_uint(flags))
        t1 = time.perf_counter()
        if rc != 0:
            print(name, "register failed rc", rc)
            hip.hipGetLastError()
            break
        rc2 = hip.hipMemcpyAsync(dev, ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), ctypes.c_int(1), ctypes.c_void_p(0))
        hip.hipDeviceSynchronize()
        t2 = time.perf_counter()
        hip.hipHostUnregister(ctypes.c_void_p(x.ctypes.data))
        t3 = time.perf_counter()
        print("%-40s register %.2f ms  copy %.2f ms (rc %d)  unregister %.2f ms  total %.2f ms" % (
            name, (t1 - t0) * 1e3, (t2 - t1) * 1e3, rc2, (t3 - t2) * 1e3, (t3 - t0) * 1e3), flush=True)
t0 = time.perf_counter()
hip.hipMemcpy(dev, ctypes.c_void_p(x.ctypes.data), ctypes.c_size_t(nbytes), ctypes.c_int(1))
print("plain hipMemcpy of the pageable buffer: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
