import sys, time, torch, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import hip_helpers as hh
from crossscore_amd import synth
from crossscore_amd.config import model_config
from crossscore_amd.model import CrossScoreNet
DEV='cuda'
dh, heads, Lq, Lk = 64, 1, 160, 200
Q = torch.zeros((1, Lq, dh), device=DEV); K = torch.zeros((1, Lk, dh), device=DEV)
sel = (torch.arange(Lq, device=DEV) * 37 + 11) % Lk
def code(i):
    bits = ((i[:, None] >> torch.arange(8, device=DEV)[None]) & 1).float() * 2 - 1
    return bits * 16
K[0, :, :8] = code(torch.arange(Lk, device=DEV)); Q[0, :, :8] = code(sel)
V = torch.arange(Lk * dh, device=DEV).float().view(1, Lk, dh) % 251 - 125
O = hh.attention(Q.bfloat16(), K.bfloat16(), V.bfloat16(), heads, dh).float()[0]
ref = V[0][sel]
bad = (O != ref).nonzero()
print('mismatches', bad.shape[0], 'of', O.numel())
for b in bad[:10]:
    q,d=b.tolist(); print(q,d,O[q,d].item(),ref[q,d].item(), 'sel',sel[q].item())
# timing of full forward cfg-2
net = CrossScoreNet(model_config()); net.load_numpy_state_dict(synth.make_state_dict(net.arch,1)); net=net.cuda()
q,r = synth.make_inputs(8,5,518,518,1); tq=torch.from_numpy(q).cuda(); tr=torch.from_numpy(r).cuda()
for chunk in (0, 8, 24, 48):
    net.enc_chunk_images=chunk; net._mark_dirty()
    for _ in range(2): net(tq,tr,False,0,False)
    torch.cuda.synchronize(); t=time.time()
    for _ in range(5): net(tq,tr,False,0,False)
    torch.cuda.synchronize(); dt=(time.time()-t)/5
    print(f'chunk {chunk}: {dt*1e3:.2f} ms/batch -> {8/dt:.1f} q/s')
net.profile_enable(True); net(tq,tr,False,0,False); 
for fam,name in enumerate(['gemm','attn','misc']):
    ms,n,fl=net.profile_read(fam); print(name, f'{ms:.3f} ms', n, 'launches', f'{fl/ms/1e9 if ms else 0:.1f} TFLOP/s')
net.profile_enable(False)
