"""Per-workgroup timeline of one scan launch (profiling aid; run on the GPU box)."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
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
tr = ix.read_trace().astype(np.int64)
t0 = tr[:, 0].min()
start, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0   # us
dur = end - start
hw = tr[:, 2]
print('start us: min %.1f max %.1f' % (start.min(), start.max()))
print('end   us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f' % (end.min(), *np.percentile(end, [10, 50, 90]), end.max()))
print('dur   us: min %.1f p10 %.1f p50 %.1f p90 %.1f max %.1f' % (dur.min(), *np.percentile(dur, [10, 50, 90]), dur.max()))
nc, nt = tr[:, 3] & 0xffffffff, tr[:, 3] >> 32
print('compactions: min %d max %d; tiles per wg: min %d p50 %d max %d' % (nc.min(), nc.max(), nt.min(), np.median(nt), nt.max()))
# HW_ID fields (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = se * 32 + sh * 16 + cu
print('distinct (se,sh,cu) keys:', len(set(key.tolist())), ' hist of wgs/key:', np.bincount(np.bincount(key)))
order = np.argsort(end)
for i in list(order[:5]) + list(order[-5:]):
    print('wg %4d key %3d start %.1f end %.1f dur %.1f' % (i, key[i], start[i], end[i], dur[i]))
