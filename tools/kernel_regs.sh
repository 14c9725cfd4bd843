#!/bin/bash
# Register / LDS / occupancy table of every kernel of a HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
#   tools/kernel_regs.sh glomeruli_segmentation_amd/csrc/espnet.hip [extra hipcc flags]
src=$1; shift
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; rows.append(cur)
    for k,tag in [('VGPRs:','v'),('AGPRs:','a'),('VGPR Spill:','vspill'),('Occupancy [waves/SIMD]:','occ'),('LDS Size [bytes/block]:','lds'),('SGPRs:','s'),('ScratchSize [bytes/lane]:','scratch')]:
        m=re.search(r'remark: .*?'+re.escape(k)+r' *(\d+)',l)
        if m and cur is not None: cur[tag]=m.group(1)
names=subprocess.run(['c++filt']+[r['name'] for r in rows],capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=n.replace('void gs::','').replace('(gs::ConvArgs)','')
    print('%-72s'%n[:72],' '.join('%s=%s'%(k,v) for k,v in r.items() if k!='name'))
"
