import sys; sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))); sys.path.insert(0,'.')
import numpy as np, kpop_amd
from kpop_amd import api
from oracle import oracle as O
kpop_amd.init(0)
rng=np.random.RandomState(6)
api.tune("summary2",0)
for r1 in (5000, 20000, 66000, 200003):
    dm=np.round(np.abs(rng.normal(1.0,0.2,size=(1,r1))),2)
    st,n,idx,d,z=kpop_amd.summarize_distances(dm,keep_at_most=2,max_neighbours=8)
    so=O.summarize_row(dm[0],2)[0]
    t=np.abs(dm[0]-so[2]); v=st[0,3]
    print(r1, 'median',st[0,2],so[2],'mad got',v,'want',so[3],' rank range of got: [%d,%d) n/2=%d'%((t<v).sum(),(t<=v).sum(),r1//2), 'max t', t.max(), 'groups near want:', [(x,(t==x).sum()) for x in np.unique(t) if abs(x-so[3])<0.011])
