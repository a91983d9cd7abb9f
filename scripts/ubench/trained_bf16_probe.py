import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle
sys.path.insert(0,'tests')
from test_trained_weights import recording
from voice_activity_detection_amd import SelfAttentiveVAD
z=np.load('tests/golden/trained.npz'); state={k[6:]:z[k] for k in z.files if k.startswith('state/')}
m=SelfAttentiveVAD(80,3,128,0.5); m.load_state_dict({k:torch.from_numpy(v) for k,v in state.items()}); m=m.cuda().eval()
_,feat,_=recording(0)
x=np.ascontiguousarray(feat[:6400].reshape(8,800,80))
ref=oracle.forward(state,x,threads=8)
for mode in (0,1,5):
    m.precision='bf16'; m.row_mode=mode
    with torch.no_grad(): y=m(features=torch.from_numpy(x).cuda()).cpu().numpy()
    m.precision='fp32'; m.row_mode=0
    lg=y[...,1]-y[...,0]; lr=ref[...,1]-ref[...,0]
    print(mode,'max|dlogp|',np.abs(y-ref).max(),'max|dp|',np.abs(np.exp(y)-np.exp(ref)).max(),'max|dlogit|',np.abs(lg-lr).max(),'rel',(np.abs(lg-lr)/(1+np.abs(lr))).max(),'agree',((lg>0)==(lr>0)).mean(), 'logit range', lr.min(), lr.max())
xw=feat[np.arange(19,len(feat)-19)[:,None]+np.array([-19,-10,-1,0,1,10,19])[None,:]]
ref=oracle.forward(state,xw,threads=8)
m.precision='bf16'
with torch.no_grad(): y=m(features=torch.from_numpy(xw).cuda()).cpu().numpy()
m.precision='fp32'
lg=y[...,1]-y[...,0]; lr=ref[...,1]-ref[...,0]
print('T7 max|dlogp|',np.abs(y-ref).max(),'max|dp|',np.abs(np.exp(y)-np.exp(ref)).max(),'rel',(np.abs(lg-lr)/(1+np.abs(lr))).max(),'agree',((lg>0)==(lr>0)).mean())
