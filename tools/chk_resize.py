import torch, numpy as np, torch.nn.functional as F, sys
sys.path.insert(0,'/root/repo/galerkin-transformer_amd')
torch.manual_seed(0)
def mine(x, no, nq, dt=np.float32):
    def ax(ni,no):
        scale=(np.array((ni-1),dtype=dt)/np.array(max(no-1,1),dtype=dt)) if no>1 else dt(0)
        o=np.arange(no).astype(dt); src=(scale*o).astype(dt)
        i0=np.minimum(src.astype(np.int64),ni-1); i1=i0+(i0<ni-1)
        l1=(src-i0.astype(dt)).astype(dt); l0=(1-l1).astype(dt); return i0,i1,l0,l1
    X=x.numpy().astype(dt); _,_,hi,wi=X.shape
    i0,i1,l0,l1=ax(hi,no); j0,j1,m0,m1=ax(wi,nq)
    a=X[:,:,i0,:]; b=X[:,:,i1,:]
    top=m0*a[:,:,:,j0]+m1*a[:,:,:,j1]; bot=m0*b[:,:,:,j0]+m1*b[:,:,:,j1]
    return torch.from_numpy(l0[None,None,:,None]*top+l1[None,None,:,None]*bot)
def rel(a,b): return float((a.double().cpu()-b.double().cpu()).norm()/b.double().cpu().norm())
for (ni,nj,no,nq) in [(141,144,78,76),(78,81,43,41),(43,46,77,75)]:
    x=torch.randn(1,4,ni,nj)
    y=F.interpolate(x,size=(no,nq),mode='bilinear',align_corners=True)
    m=mine(x,no,nq)
    line=f"{ni}x{nj}->{no}x{nq}: numpy32 vs torch32 {rel(m,y):.2e}"
    if torch.cuda.is_available():
        from galerkin_transformer import _hip as H
        g=H.bilinear2d_fwd(x.cuda(),(no,nq),False,False,0)
        line+=f"  hip vs torch32 {rel(g,y):.2e}  hip vs numpy32 {rel(g,m):.2e}"
        yg=F.interpolate(x.cuda(),size=(no,nq),mode='bilinear',align_corners=True)
        line+=f"  torchGPU vs torch32cpu {rel(yg,y):.2e}"
        d=(g.cpu()-y).abs(); idx=np.unravel_index(int(d.argmax()),d.shape); line+=f" maxabs {float(d.max()):.2e} at {idx}"
    print(line)
