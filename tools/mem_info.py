import ctypes
hip = ctypes.CDLL("libamdhip64.so")
f, t = ctypes.c_size_t(), ctypes.c_size_t()
hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
print("hipMemGetInfo free %.1f GB of %.1f GB" % (f.value / 1e9, t.value / 1e9))
