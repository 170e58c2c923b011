#!/bin/bash
# kernel resource usage of one .hip file: name, VGPRs, scratch, occupancy, spills (hipcc -Rpass-analysis=kernel-resource-usage)
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | \
python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        if 'error' in line: print(line.rstrip())
        continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        cur={'name':t.split(': ',1)[1]}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
for r in rows:
    n=subprocess.run(['/usr/bin/c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    print('%-110s V=%s S=%s scratch=%s occ=%s spillV=%s LDS=%s'%(n[:110],r.get('VGPRs'),r.get('SGPRs'),r.get('ScratchSize [bytes/lane]'),r.get('Occupancy [waves/SIMD]'),r.get('VGPRs Spill'),r.get('LDS Size [bytes/block]')))
"
