import math, sys, os
sys.path.insert(0, os.getcwd())
import torch
from segclip_amd import ops
B,T,H,hd,causal=1,77,1,64,True
D=H*hd
g=torch.Generator(device="cuda").manual_seed(5)
qkv=torch.randn(B*T,3*D,device="cuda",generator=g).to(torch.bfloat16)
o=torch.full((B*T,D),float("nan"),dtype=torch.bfloat16,device="cuda")
s3=(T*3*D,3*D)
d=ops._attn_desc(qkv,qkv,qkv,o,B,H,T,T,hd,s3,s3,s3,(T*D,D),1/math.sqrt(hd),causal,0,D,2*D)
st=ops.p_attn_fwd(d,qkv)
q,k,v=(qkv.float().view(B,T,3,D)[0,:,i] for i in range(3))
s=q@k.T/math.sqrt(hd)
o=o.float()
def att(mask):
    ss=s.clone(); ss[~mask]=float("-inf"); return torch.softmax(ss,-1)@v
idx=torch.arange(T,device="cuda")
full=idx[None,:]<=idx[:,None]
hyps={"causal":full,
 "only_first_tile": (idx[None,:]<32).expand(T,T),
 "only_diag_tile": full & (idx[None,:]>=(idx[:,None]//32*32)),
 "no_mask_in_diag": idx[None,:] < (idx[:,None]//32*32+32),
 "strict": idx[None,:]<idx[:,None],
 "lh0_only_diag": full & ~((idx[None,:]>=(idx[:,None]//32*32)) & ((idx[None,:]&4)!=0)),
}
for r in (32,33,40,36):
    print("row",r,"lse",float(st[r]), "ref lse", float(torch.logsumexp(s[r,:r+1],0)))
    for n,mk in hyps.items():
        print("   ",n, float((o[r]-att(mk)[r]).abs().max()))
