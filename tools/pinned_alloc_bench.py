"""Diagnostic (GPU host): what 256 MB of page-locked memory cost — hipHostMalloc, mmap(MAP_POPULATE) + hipHostRegister, transparent huge
pages — and the device-to-host rate into each (fk_host_alloc uses the second).  usage: python tools/pinned_alloc_bench.py"""
import ctypes as C, mmap, time, threading, sys
hip = C.CDLL("libamdhip64.so")
hip.hipSetDevice(0); hip.hipFree(None)
N = 256 << 20
def t(f):
    t0 = time.perf_counter(); r = f(); return (time.perf_counter() - t0) * 1e3, r
def host_malloc():
    p = C.c_void_p(); rc = hip.hipHostMalloc(C.byref(p), C.c_size_t(N), C.c_uint(0)); assert rc == 0; return p
for i in range(3):
    ms, p = t(host_malloc); print("hipHostMalloc 256MB: %.1f ms" % ms); hip.hipHostFree(p)
libc = C.CDLL("libc.so.6", use_errno=True)
libc.mmap.restype = C.c_void_p; libc.mmap.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_long]
MAP_PRIVATE, MAP_ANON, MAP_POPULATE, MAP_HUGETLB = 2, 0x20, 0x8000, 0x40000
def mm(flags):
    p = libc.mmap(None, N, 3, MAP_PRIVATE | MAP_ANON | flags, -1, 0); assert p not in (None, C.c_void_p(-1).value), C.get_errno(); return p
for i in range(3):
    ms, p = t(lambda: mm(MAP_POPULATE)); print("mmap MAP_POPULATE 256MB: %.1f ms" % ms)
    ms2, rc = t(lambda: hip.hipHostRegister(C.c_void_p(p), C.c_size_t(N), C.c_uint(0))); print("  hipHostRegister: %.1f ms rc=%d" % (ms2, rc))
    # D2H bandwidth into it
    d = C.c_void_p(); hip.hipMalloc(C.byref(d), C.c_size_t(N))
    hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(N), C.c_int(2))
    ms3, _ = t(lambda: hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(N), C.c_int(2))); print("  D2H into registered: %.1f ms = %.1f GB/s" % (ms3, N / ms3 / 1e6))
    hip.hipFree(d)
    hip.hipHostUnregister(C.c_void_p(p)); libc.munmap(C.c_void_p(p), C.c_size_t(N))
# madvise hugepage variant
MADV_HUGEPAGE = 14
p = libc.mmap(None, N, 3, MAP_PRIVATE | MAP_ANON, -1, 0)
libc.madvise(C.c_void_p(p), C.c_size_t(N), MADV_HUGEPAGE)
ms, _ = t(lambda: C.memset(C.c_void_p(p), 0, N)); print("THP madvise + memset: %.1f ms" % ms)
ms2, rc = t(lambda: hip.hipHostRegister(C.c_void_p(p), C.c_size_t(N), C.c_uint(0))); print("  hipHostRegister (THP): %.1f ms rc=%d" % (ms2, rc))
d = C.c_void_p(); hip.hipMalloc(C.byref(d), C.c_size_t(N)); hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(N), C.c_int(2))
ms3, _ = t(lambda: hip.hipMemcpy(C.c_void_p(p), d, C.c_size_t(N), C.c_int(2))); print("  D2H into registered THP: %.1f ms = %.1f GB/s" % (ms3, N / ms3 / 1e6))
# pinned hipHostMalloc D2H for reference
ms, q = t(host_malloc); hip.hipMemcpy(q, d, C.c_size_t(N), C.c_int(2))
ms3, _ = t(lambda: hip.hipMemcpy(q, d, C.c_size_t(N), C.c_int(2))); print("D2H into hipHostMalloc: %.1f ms = %.1f GB/s" % (ms3, N / ms3 / 1e6))
# two hipHostMalloc in parallel threads
res = []
def w():
    ms, p = t(host_malloc); res.append(ms)
ths = [threading.Thread(target=w) for _ in range(2)]
t0 = time.perf_counter(); [x.start() for x in ths]; [x.join() for x in ths]
print("two parallel hipHostMalloc: each", [round(r, 1) for r in res], "total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
