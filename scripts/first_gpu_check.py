"""Quick GPU sanity run of libipx (host-pointer API) against a float64 numpy top-k."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, '.')
from mural_amd import _lib
L = _lib.lib()
print(L.ipx_version().decode(), 'devices', L.ipx_device_count(), flush=True)

def run(N, d, nq, k, seed=0, adds=1):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((N, d), dtype=np.float32)
    Q = rng.standard_normal((nq, d), dtype=np.float32)
    h = C.c_void_p()
    _lib.check(L.ipx_create(d, 0, C.byref(h)))
    step = (N + adds - 1) // adds
    for a in range(0, N, step):
        xs = np.ascontiguousarray(X[a:a + step])
        _lib.check(L.ipx_add(h, xs.ctypes.data, xs.shape[0], 0, 0, None))
    D = np.empty((nq, k), np.float32); I = np.empty((nq, k), np.int64)
    t = time.time()
    _lib.check(L.ipx_search(h, Q.ctypes.data, nq, k, D.ctypes.data, I.ctypes.data, 0, None))
    dt = time.time() - t
    S = Q.astype(np.float64) @ X.astype(np.float64).T
    kk = min(k, N)
    ref = np.argsort(-S, axis=1, kind='stable')[:, :kk]
    refD = np.take_along_axis(S, ref, axis=1)
    ok_ids = (I[:, :kk] == ref)
    bad = (~ok_ids).sum()
    maxerr = np.abs(D[:, :kk] - refD).max() if kk else 0
    pad_ok = (I[:, kk:] == -1).all() and (D[:, kk:] == -np.finfo(np.float32).max).all()
    g = [C.c_int(), C.c_int(), C.c_int()]
    L.ipx_last_geometry(h, *[C.byref(x) for x in g])
    print(f'N={N} d={d} nq={nq} k={k} adds={adds}: id mismatches={bad}/{ok_ids.size} maxerr={maxerr:.3e} pad_ok={pad_ok} '
          f'geom={[x.value for x in g]} t={dt*1e3:.1f}ms', flush=True)
    if bad:
        r, c = np.argwhere(~ok_ids)[0]
        print('  first mismatch q', r, 'rank', c, 'got', I[r, c], D[r, c], 'want', ref[r, c], refD[r, c], flush=True)
    L.ipx_destroy(h)
    return bad == 0 and pad_ok

ok = True
for args in [(1000, 64, 10, 5), (5000, 768, 300, 10), (5000, 768, 300, 100), (100, 768, 7, 100),
             (20000, 100, 513, 100, 1, 3), (40000, 768, 1000, 1000), (300000, 768, 2000, 100)]:
    ok &= run(*args)
print('ALL OK' if ok else 'FAILURES', flush=True)
sys.exit(0 if ok else 1)
