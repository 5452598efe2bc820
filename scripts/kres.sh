#!/bin/bash
# kernel resource usage summary for a .hip file: name VGPR AGPR scratch occupancy
f=${1:-/root/repo/emap_amd/csrc/udf_mlp.hip}
mkdir -p /tmp/t && cd /tmp/t && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o /tmp/t/kres.o -save-temps -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}
    for k in ['VGPRs:','AGPRs:','ScratchSize [bytes/lane]:','Occupancy [waves/SIMD]:','SGPRs Spill:','LDS Size']:
        if k in l and cur: d[k]=l.split(k)[1].split()[0]
    if 'LDS Size' in l and cur:
        print(cur.replace('_ZN4emap14udf_mlp_kernel','mlp')[:64].ljust(64), 'V',d.get('VGPRs:'),'A',d.get('AGPRs:'),'scr',d.get('ScratchSize [bytes/lane]:'),'occ',d.get('Occupancy [waves/SIMD]:'),'sspill',d.get('SGPRs Spill:'))
"
