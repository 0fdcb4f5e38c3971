import sys, os
root = os.environ.get('GRAFT_REPO_ROOT', '/root/repo')
sys.path[:0]=[root, root+'/segmentation-networks-benchmark_amd', root+'/tests']
import numpy as np, torch, time
import model_checks as mc, abi_replay
from lib.losses import BCEWithLogitsLossAndSmoothJaccard
g=np.load(root+'/tests/golden/tiramisu_small.npz')
_,_,x,y=mc.make_tiramisu(g)
t=time.time()
n,rep=abi_replay.replay(lambda: mc.make_tiramisu(g)[0], x,y,BCEWithLogitsLossAndSmoothJaccard(),'f32')
print(n, time.time()-t)
print('\n'.join(rep))
