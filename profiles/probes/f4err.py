import numpy as np, os, sys
sys.path.insert(0,'.')
import rocket_path_amd as rp
t=np.load('tests/golden/f4_steps.npz')
n=len(t['state_in'])
for dt,name in ((rp.DTYPE_F64,'f64'),(rp.DTYPE_F32,'f32')):
    with rp.Batch(n, rp.VARIANT_F4, dt) as b:
        b.set_state(t['state_in']); b.step(1); st=b.get_state()
    err=np.abs(st[:,:3]-t['state_out'][:,:3])/np.maximum(np.abs(t['state_out'][:,:3]),1.0)
    e=err.max(axis=1)
    print(name,'quantiles 50/90/99/99.9/max', [float('%.3g'%np.quantile(e,q)) for q in (.5,.9,.99,.999,1.0)])
    worst=np.argsort(e)[-5:]
    for i in worst: print('   i',i,'presteps',t['presteps'][i],'err',e[i],'in',t['state_in'][i,:3],'out ref',t['state_out'][i,:3],'gpu',st[i,:3])
