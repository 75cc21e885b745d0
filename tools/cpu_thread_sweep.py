import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from oracle import stgcn_ref as R
gargs=dict(layout='ntu-rgb+d', strategy='spatial_3')
m = R.RefModel('st_gcn_msgcn', 3, 60, gargs, True, dropout=0.5); R.weights_init_(m, seed=0); opt=R.make_optimizer(m)
x=torch.randn(4,3,300,25,2); y=torch.randint(0,60,(4,))
for th in (8,16,32,64):
    torch.set_num_threads(th)
    R.train_step(m,opt,x,y)
    t=time.time(); R.train_step(m,opt,x,y); dt=time.time()-t
    print('threads',th,'s/step(B=4)',round(dt,2),'clips/s',round(4/dt,3), flush=True)
