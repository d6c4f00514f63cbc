#!/bin/bash
# dev helper: compile rowfft.hip and list kernels with their VGPR count / spills
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -I/opt/rocm/include $RF_DEFS -c rowfft.hip -o build/rowfft.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None
for l in sys.stdin:
    if 'error' in l: print(l)
    m=re.search(r'Function Name: _ZN6pfbhip\d+(k_\w+?)INS_7RfShapeILi(\d+)ELi(\d+)',l)
    if m: name='%s<%s,%s>'%m.groups(); d={}
    for k in ['VGPRs','VGPR Spill','ScratchSize','Occupancy']:
        m=re.search(r' '+k+r'[^:]*: (\d+)',l)
        if m and name: d[k]=m.group(1)
    if name and 'LDS Size' in l: print(name,d)
"
