#!/bin/bash
# tools/regs.sh [TUNE flags] — VGPRs / spills / LDS of every kernel in qm2_hip.hip
cd "$(dirname "$0")/../quick-mer2_amd" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null csrc/qm2_hip.hip 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows={}
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}
    for k in ('VGPRs','VGPRs Spill','SGPRs Spill','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]'):
        m=re.search(r'remark:\s+'+k+r': (\d+)',l)
        if m and cur: rows[cur][k]=m.group(1)
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0]
    print('%-60s' % name[:60], ' '.join('%s=%s'%(a.split(' [')[0].replace('\\\\',''),b) for a,b in v.items()))
"
