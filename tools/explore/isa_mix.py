"""Instruction mix (v_mfma / VALU / SALU / s_waitcnt / s_nop / LDS / VMEM) of every loop block with >= 8 MFMAs in a kernel's ISA text
(hipcc -S --cuda-device-only, one kernel cut out with awk): tools/explore/isa_mix.py kernel.s [...]"""
import re,collections,sys
def analyse(path):
    lines=[l.strip() for l in open(path) if l.strip() and not l.strip().startswith(';')]
    # find the innermost loop with most mfma: split by labels
    blocks=[];cur=[];name='entry'
    for l in lines:
        if re.match(r'^\.LBB\d+_\d+:',l):
            blocks.append((name,cur));cur=[];name=l.split(':')[0]
        else: cur.append(l)
    blocks.append((name,cur))
    out=[]
    for name,b in blocks:
        c=collections.Counter()
        for l in b:
            op=l.split()[0]
            if op.startswith('v_mfma'): c['mfma']+=1
            elif op.startswith('v_'): c['valu']+=1
            elif op.startswith('s_waitcnt'): c['waitcnt']+=1
            elif op.startswith('s_nop'): c['nop']+=1
            elif op.startswith('s_'): c['salu']+=1
            elif op.startswith('ds_'): c['lds']+=1
            elif op.startswith('buffer_') or op.startswith('global_'): c['vmem']+=1
        if c['mfma']>=8: out.append((name,dict(c)))
    return out
for p in sys.argv[1:]:
    print(p)
    for name,c in analyse(p): print("   ",name,c)
