import sys, numpy as np, torch
sys.path.insert(0,'.')
import gpu_matrix_inversion_amd as g
def gate(n,seed):
    rng=np.random.default_rng(seed); a=rng.uniform(-1,1,(n,n))+np.sqrt(n)*np.eye(n); return a[rng.permutation(n)].astype(np.float32)
n=40
a=gate(n,77); a[5,5]=np.nan
for algo in sys.argv[1:]:
    inv=g.Inverter(algo=algo)
    print('running',algo,flush=True)
    x,st=inv.inv(torch.from_numpy(a).cuda()); torch.cuda.synchronize()
    print(algo,'status',st.item(), 'nan count', int(torch.isnan(x).sum()),flush=True)
