"""pgv_host_pool_shutdown with sleepers on every futex (no GPU needed: a pool segment that no server has joined).
Clients that wait for a lane to come under a leader, and the shutdown that tells them to stop waiting."""
import ctypes as C
import mmap
import threading
import time

from pgvector_amd import _host


def test_shutdown_wakes_every_sleeper():
    lib = _host.lib
    lib.pgv_host_pool_shm_bytes.restype = C.c_size_t
    lib.pgv_host_pool_shm_bytes.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pgv_host_pool_shm_init.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pgv_host_pool_attach.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    lib.pgv_host_pool_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.pgv_host_pool_shutdown.argtypes = [C.c_void_p]
    lib.pgv_host_pool_detach.argtypes = [C.c_void_p]
    dim, k = 16, 4
    nbytes = lib.pgv_host_pool_shm_bytes(0, dim, k, 8, 2)
    seg = mmap.mmap(-1, nbytes)
    base = C.addressof(C.c_char.from_buffer(seg))
    assert lib.pgv_host_pool_shm_init(base, nbytes, 0, dim, 3, k, 8, 100, 2) == 0
    pool = C.c_void_p()
    assert lib.pgv_host_pool_attach(base, nbytes, C.byref(pool)) == 0
    results = []

    def client():
        q = (C.c_float * dim)()
        tid = (C.c_uint64 * k)()
        dist = (C.c_float * k)()
        t0 = time.time()
        rc = lib.pgv_host_pool_search(pool, q, tid, dist)
        results.append((rc, time.time() - t0))
    threads = [threading.Thread(target=client) for _ in range(12)]
    for t in threads:
        t.start()
    time.sleep(0.4)                       # all twelve sleep: no lane has a leader
    assert not results
    t0 = time.time()
    lib.pgv_host_pool_shutdown(pool)
    for t in threads:
        t.join(timeout=5)
    assert all(not t.is_alive() for t in threads), "a client slept through the shutdown"
    assert time.time() - t0 < 2.0
    assert len(results) == 12 and all(rc != 0 for rc, _ in results)
    # and a pool nobody ever serves tells its clients so by itself, after five seconds
    seg2 = mmap.mmap(-1, nbytes)
    base2 = C.addressof(C.c_char.from_buffer(seg2))
    assert lib.pgv_host_pool_shm_init(base2, nbytes, 0, dim, 3, k, 8, 100, 2) == 0
    pool2 = C.c_void_p()
    assert lib.pgv_host_pool_attach(base2, nbytes, C.byref(pool2)) == 0
    q = (C.c_float * dim)()
    t0 = time.time()
    assert lib.pgv_host_pool_search(pool2, q, (C.c_uint64 * k)(), (C.c_float * k)()) != 0
    assert 4.0 < time.time() - t0 < 8.0
    lib.pgv_host_pool_detach(pool2)
    lib.pgv_host_pool_detach(pool)
    del base, base2
