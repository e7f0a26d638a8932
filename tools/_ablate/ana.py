import numpy as np, sys, re
us={}
for l in open('gpurun_out/gemm_timeline_p.txt'):
    m=re.match(r"(.*) \((\d), (\d)\) mask (\d+): launch ([\d.]+) us",l)
    if m: us[(m.group(4),m.group(2)+m.group(3))]=float(m.group(5))
def ana(mask,form):
    d=np.load('gpurun_out/gemm_tl_%s_%s.npy'%(mask,form))
    d=d[d[:,0]>0]
    t=d[:,:6].astype(np.float64); hw=d[:,6]; xcc=d[:,7]&0xf
    key=(xcc<<16)|(hw&0xff00)|((hw>>13)&7)
    spans=[]; gaps=[]
    for k in np.unique(key):
        tt=t[key==k]; tt=tt[np.argsort(tt[:,0])]
        spans.append(tt[-1,5]-tt[0,0]); gaps.append(tt[1:,0]-tt[:-1,5])
    spans=np.array(spans); g=np.concatenate(gaps)
    tk=np.median(spans)/us[(mask,form)]
    print(form,'launch %.1f us, tiles'%us[(mask,form)],len(d),'-> %.0f ticks/us'%tk)
    for i,n in ((2,'k loop (0->2)'),(3,'epi half0'),(4,'epi half1'),(5,'stats/end')):
        j = 0 if i==2 else i-1
        dt=t[:,i]-t[:,j]; print('   %-14s mean %6.2f us p10 %6.2f p90 %6.2f'%(n,dt.mean()/tk,np.quantile(dt,.1)/tk,np.quantile(dt,.9)/tk))
    print('   tile %.2f us; gap %.2f us'%((t[:,5]-t[:,0]).mean()/tk,g.mean()/tk))
import os
for mk in os.environ.get('MASKS','16').split(','):
    print('### mask',mk)
    for f in ('21','00','02'): ana(mk,f)
