"""Condense a gpurun_out/prof_<tag>/ directory (scripts/profile_bench.sh) into profiles/<name>/."""
import collections, csv, glob, hashlib, json, os, shutil, subprocess, sys
tag, name = sys.argv[1], sys.argv[2]
kern = sys.argv[3] if len(sys.argv) > 3 else 'scan_f32'
src, dst = f'gpurun_out/prof_{tag}', f'profiles/{name}'
os.makedirs(dst, exist_ok=True)
shutil.copy(glob.glob(f'{src}/stats/vm/*kernel_stats.csv')[0], f'{dst}/kernel_stats.csv')
out = {}
for t in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
    fs = glob.glob(f'{src}/{t}/vm/*counter_collection.csv')
    if not fs:
        continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if kern in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            out['dispatch'] = {k: r[k] for k in ('Grid_Size', 'Workgroup_Size', 'LDS_Block_Size', 'VGPR_Count', 'SGPR_Count')}
    out[t] = {k: sum(v) / len(v) for k, v in agg.items()}
f, w = out.get('pmc_fetch', {}).get('FETCH_SIZE'), out.get('pmc_write', {}).get('WRITE_SIZE')
if f is not None and w is not None:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half
    # the bytes of a wide (16 B/lane, LDS-DMA included) streaming read -> doubled; WRITE_SIZE exact.
    out['traffic_bytes_per_launch'] = f * 1024 * 2 + w * 1024
    out['traffic_note'] = 'memory-side (L2 fabric) bytes: FETCH_SIZE*1024*2 + WRITE_SIZE*1024; Infinity-Cache hits are included'
# what was profiled: bench.py quotes traffic_bytes_per_launch only while ipx_kernels.h still has this hash
out['kernel_source_sha256'] = hashlib.sha256(open('mural_amd/csrc/ipx_kernels.h', 'rb').read()).hexdigest()
out['git_head'] = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()
import re
safe = re.sub(r'[^A-Za-z0-9]+', '_', kern).strip('_')      # scan_bf16_kernel<1, 2> -> scan_bf16_kernel_1_2
out['kernel'] = kern
json.dump(out, open(f'{dst}/{safe}_pmc.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
