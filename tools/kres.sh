#!/bin/bash
# kernel resource usage of one .hip file: tools/kres.sh fprop_roll.hip [filter]
cd "$(dirname "$0")/../segmentation-networks-benchmark_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-gpu-rdc --cuda-device-only -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+([A-Za-z \[\]/]+): (\d+)',l)
    if m and cur:
        d[m.group(1).strip()]=m.group(2)
        if m.group(1).startswith('LDS'):
            print(cur[:90], 'VGPR',d.get('VGPRs'),'AGPR',d.get('AGPRs'),'spill',d.get('VGPRs Spill'),'SGPR',d.get('TotalSGPRs'),'occ',d.get('Occupancy [waves/SIMD]'),'LDS',d.get('LDS Size [bytes/block]'))
" | grep "${2:-.}"
