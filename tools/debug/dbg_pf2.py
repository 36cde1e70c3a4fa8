import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from segclip_amd import ops
for (B,T,H,hd,causal) in [(1,77,1,64,True),(2,170,1,64,True),(2,196,2,64,False)]:
    D=H*hd
    g=torch.Generator(device="cuda").manual_seed(5)
    qkv=torch.randn(B*T,3*D,device="cuda",generator=g).to(torch.bfloat16)
    o=torch.full((B*T,D),float("nan"),dtype=torch.bfloat16,device="cuda")
    s3=(T*3*D,3*D)
    d=ops._attn_desc(qkv,qkv,qkv,o,B,H,T,T,hd,s3,s3,s3,(T*D,D),1/math.sqrt(hd),causal,0,D,2*D)
    st=ops.p_attn_fwd(d,qkv)
    q,k,v=(qkv.float().view(B,T,3,D)[:,:,i] for i in range(3))
    s=q@k.transpose(-1,-2)/math.sqrt(hd)
    if causal: s=s+torch.full((T,T),float("-inf"),device="cuda").triu_(1)
    ref=torch.softmax(s,-1)@v
    err=(o.float().view(B,T,D)-ref).abs().amax(-1)
    print(os.environ.get("SEGCLIP_ATTN_PF_DBG"),B,T,causal," max err per 32-row tile:", [round(float(err[0,i:i+32].max()),4) for i in range(0,T,32)])
