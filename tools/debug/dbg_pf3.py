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
for r in (32,33,35,36,40,41,44,48,64,65,68,72):
    lse=torch.logsumexp(s[r,:r+1],0)
    p=torch.exp(s[r]-lse); p[r+1:]=0
    t0=(r//32)*32
    U0=p[:t0]@v[:t0]; U1=p[t0:]@v[t0:]
    A=torch.stack([U0,U1],1)
    sol=torch.linalg.lstsq(A,o[r][:,None]).solution[:,0]
    res=(A@sol-o[r]).abs().max()
    # per-key coefficients for the diag tile: o - U0 = sum_k c_k v_k, solve least squares over diag keys (<=32 unknowns, 64 eqs)
    Vd=v[t0:t0+32].T  # 64 x 32
    c=torch.linalg.lstsq(Vd,(o[r]-U0)[:,None]).solution[:,0]
    want=p[t0:t0+32]
    bad=[(int(i),round(float(c[i]),3),round(float(want[i]),3)) for i in range(min(32,T-t0)) if abs(float(c[i]-want[i]))>0.01]
    print(f"row {r}: a={float(sol[0]):.3f} b={float(sol[1]):.3f} resid {float(res):.3f}; diag-key coeffs off (key, got, want): {bad[:12]}")
