"""Diagnostic: per-wave cycle sums (vmcnt wait, barrier wait, epilogue) from the -DIPX_STAMPS
build (mural_amd/libipx_diag.so). Shares, not run time, are what to read."""
import os, sys, ctypes as C
import numpy as np
import torch
sys.path.insert(0, '.')
from mural_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), 'libipx_diag.so')
from mural_amd.faiss_retriever.retriever import FlatIPIndex
n, d, nq, k = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000, 768, 6980, 100
dev = torch.device('cuda:0')
ix = FlatIPIndex(d, 0); ix.reserve(n)
g = torch.Generator(device=dev); g.manual_seed(1)
for lo in range(0, n, 200_000):
    ix.add(torch.randn((min(200_000, n - lo), d), generator=g, device=dev))
Q = torch.randn((nq, d), generator=g, device=dev)
ix.set_param('trace', 1); ix.set_timing(True)
for it in range(2):
    ix.search_device(Q, k); torch.cuda.synchronize()
print(ix.last_timing(), ix.last_geometry())
G = ix.last_geometry()['workgroups']
out = np.zeros((G * 20,), dtype=np.uint64)
_lib.check(_lib.lib().ipx_read_trace(ix._h, C.c_void_p(out.ctypes.data), G * 5))
wg = out[:G * 4].reshape(G, 4).astype(np.int64)
wv = out[G * 4:].reshape(G, 4, 4).astype(np.float64)
dur = (wg[:, 1] - wg[:, 0]) / 100.0
lone = dur < 0.7 * np.median(dur)
for name, sel in (('paired', ~lone), ('lone', lone)):
    w = wv[sel]
    tot = w[:, :, 0]
    print(name, 'wgs', sel.sum(), 'dur ms %.1f' % (dur[sel].mean() / 1e3), 'cycles/wave %.3e' % tot.mean(),
          'filter %.1f%%' % (100 * (w[:, :, 1] / tot).mean()), 'barrier %.1f%%' % (100 * (w[:, :, 2] / tot).mean()),
          'epilogue %.1f%%' % (100 * (w[:, :, 3] / tot).mean()))
    print('   per wave (mean over wgs): vm', (100 * w[:, :, 1] / tot).mean(0).round(1), 'bar', (100 * w[:, :, 2] / tot).mean(0).round(1))
