import sys; sys.path.insert(0,'.')
import numpy as np, torch
from tests import pipeline_runner as R
gold = np.load('tests/golden/glue_step.npz')
out = R.run_train_step(gold, torch.device('cuda:0'))
res=[]
for name, norm, maxabs in zip(gold['grad_names'], gold['grad_norms'], gold['grad_maxabs']):
    key='train_grad/'+str(name)
    if key in gold.files and maxabs>1e-5:
        res.append((np.abs(out['grads'][str(name)]-gold[key]).max()/maxabs, str(name)))
    else:
        res.append((abs(np.linalg.norm(out['grads'][str(name)].astype(np.float64))-norm)/max(norm,1e-9), 'N:'+str(name)))
res.sort(reverse=True)
for r in res[:25]: print(f"{r[0]:.2e} {r[1]}")
print("median", np.median([r[0] for r in res]))
