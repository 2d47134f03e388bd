import sys, os
R="/root/repo"
v=os.environ.get("CS_VARIANT","-")
sys.path.insert(0,R); sys.path.insert(0,R+"/tests")
if v!="-": sys.path.insert(0,R+"/tools/_var/"+v)
import torch, hip_helpers as hh
from crossscore_amd import _lib
from test_hip_panel import _make
lib=_lib.load(); lib.cs_debug_panel_impl(1)
dev=torch.device("cuda:0")
for M in (300, 65760):
    x,o,w=_make(M,5,dev)
    img=hh.panel_pack(w["wo"],w["ls1"],w["w1"],w["g2"],w["w2"],w["ls2"])
    outs=[]
    for r in range(5):
        xa=x.clone(); ua=hh.encoder_panel(xa,o,img,w["bo"],w["b1"],w["b2"]); torch.cuda.synchronize(); outs.append((xa,ua))
    neq=[int((outs[0][0]!=v_[0]).sum()) for v_ in outs[1:]]
    print(v, "M",M,"differing x elements vs run 0:", neq, "u:", [int((outs[0][1]!=v_[1]).sum()) for v_ in outs[1:]])
