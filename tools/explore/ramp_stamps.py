import numpy as np, sys
raw=np.loadtxt(sys.argv[1]); st=raw[:,1:]
t0=st[:,0]
us=lambda x:(x-t0)/100.0
pre=us(st[:,126]); dma=us(st[:,127]); staged=us(st[:,1])
first_chunk=us(st[:,2])
for n,v in (("start -> before DMA (kernargs, task decode, first ring loads issued)",pre),("-> DMA issued",dma),("-> staged (wait + barrier)",staged),("-> end of first chunk",first_chunk)):
    print("%-75s median %.2f us  p10 %.2f  p90 %.2f"%(n,np.median(v),np.percentile(v,10),np.percentile(v,90)))
print("launch-relative start spread: p90-p10 %.2f us"%((np.percentile(t0,90)-np.percentile(t0,10))/100))
