"""CPU twin of csrc/conv_winograd.hip: the kernels' tile decode (dilation as sub-grids, odd maps), input / output-gradient /
filter transforms and the 16 contractions restated line by line in numpy loops (fp64), checked against F.conv2d for the forward
pass, the data gradient (180-degree rotated [C,3,3,K] filter) and the filter gradient.  This is what was run BEFORE any GPU
time was spent on the kernels: every printed number must be ~1e-15."""
import numpy as np, torch, torch.nn.functional as F
def cdiv(a,b): return (a+b-1)//b
def twin(x, w, dil, flip=False):
    # x [N,H,W,C] NHWC ; w [O,3,3,I] ; returns y [N,H,W,O]
    N,H,W,C = x.shape; O = w.shape[0]
    d = dil; th=(cdiv(H,d)+1)//2; tw=(cdiv(W,d)+1)//2; T=N*d*d*th*tw
    U=np.zeros((16,O,C),np.float64)
    for o in range(O):
        for i in range(C):
            g=np.array([[w[o,(2-r if flip else r),(2-s if flip else s),i] for s in range(3)] for r in range(3)])
            t=np.zeros((4,3))
            for s in range(3):
                t[0,s]=g[0,s]; t[1,s]=.5*(g[0,s]+g[1,s]+g[2,s]); t[2,s]=.5*(g[0,s]-g[1,s]+g[2,s]); t[3,s]=g[2,s]
            for a in range(4):
                U[a*4+0,o,i]=t[a,0]; U[a*4+1,o,i]=.5*(t[a,0]+t[a,1]+t[a,2]); U[a*4+2,o,i]=.5*(t[a,0]-t[a,1]+t[a,2]); U[a*4+3,o,i]=t[a,2]
    V=np.zeros((16,T,C))
    def tile_of(t):
        tx=t%tw; t//=tw; ty=t%th; t//=th; j=t%d; t//=d; i=t%d; n=t//d
        return n,i,j,ty,tx
    for t in range(T):
        n,i,j,ty,tx=tile_of(t)
        r=np.zeros((4,4,C))
        for v in range(4):
            sx=2*tx-1+v; ww=sx*d+j; wok = sx>=0 and ww<W
            dcol=np.zeros((4,C))
            for u in range(4):
                sy=2*ty-1+u; h=sy*d+i
                if wok and sy>=0 and h<H: dcol[u]=x[n,h,ww]
            r[0,v]=dcol[0]-dcol[2]; r[1,v]=dcol[1]+dcol[2]; r[2,v]=dcol[2]-dcol[1]; r[3,v]=dcol[1]-dcol[3]
        for u in range(4):
            V[u*4+0,t]=r[u,0]-r[u,2]; V[u*4+1,t]=r[u,1]+r[u,2]; V[u*4+2,t]=r[u,2]-r[u,1]; V[u*4+3,t]=r[u,1]-r[u,3]
    M=np.einsum('xtc,xoc->xto',V,U)
    y=np.zeros((N,H,W,O))
    for t in range(T):
        n,i,j,ty,tx=tile_of(t)
        s=np.zeros((2,4,O))
        for v in range(4):
            m0,m1,m2,m3=M[0*4+v,t],M[1*4+v,t],M[2*4+v,t],M[3*4+v,t]
            s[0,v]=m0+m1+m2; s[1,v]=m1-m2-m3
        for a in range(2):
            h=(2*ty+a)*d+i
            if h>=H: continue
            o0=s[a,0]+s[a,1]+s[a,2]; o1=s[a,1]-s[a,2]-s[a,3]
            for b in range(2):
                ww=(2*tx+b)*d+j
                if ww>=W: continue
                y[n,h,ww]=o1 if b else o0
    return y
rng=np.random.default_rng(0)
for (N,C,H,W,K,d) in [(1,3,5,7,4,1),(2,2,6,6,3,2),(1,2,9,5,2,3),(1,2,6,6,2,6),(1,3,7,8,2,4)]:
    x=rng.standard_normal((N,C,H,W)); w=rng.standard_normal((K,C,3,3))
    xt=torch.tensor(x,requires_grad=True); wt=torch.tensor(w)
    yr=F.conv2d(xt,wt,None,1,d,d)
    y=twin(x.transpose(0,2,3,1), w.transpose(0,2,3,1), d)
    e1=np.abs(y.transpose(0,3,1,2)-yr.detach().numpy()).max()
    gy=rng.standard_normal(yr.shape); yr.backward(torch.tensor(gy))
    # dgrad: input dy NHWC [N,H,W,K], filter crsk [C,3,3,K] flipped
    wcrsk=w.transpose(1,2,3,0)
    dx=twin(gy.transpose(0,2,3,1), wcrsk, d, flip=True)
    e2=np.abs(dx.transpose(0,3,1,2)-xt.grad.numpy()).max()
    print((N,C,H,W,K,d), e1, e2)

# ---- filter gradient twin (mirrors wino_input_kernel / wino_dy_kernel / wino_filter_grad_kernel)
def twin_wgrad(x, dy, dil):
    # x [N,H,W,C], dy [N,H,W,K] -> dg [K,3,3,C]
    N,H,W,C=x.shape; K=dy.shape[3]; d=dil
    th=(cdiv(H,d)+1)//2; tw=(cdiv(W,d)+1)//2; T=N*d*d*th*tw
    def tile_of(t):
        tx=t%tw; t//=tw; ty=t%th; t//=th; j=t%d; t//=d; i=t%d; n=t//d
        return n,i,j,ty,tx
    V=np.zeros((16,T,C)); Wt=np.zeros((16,T,K))
    for t in range(T):
        n,i,j,ty,tx=tile_of(t)
        r=np.zeros((4,4,C))
        for v in range(4):
            sx=2*tx-1+v; ww=sx*d+j; wok = sx>=0 and ww<W
            dcol=np.zeros((4,C))
            for u in range(4):
                sy=2*ty-1+u; h=sy*d+i
                if wok and sy>=0 and h<H: dcol[u]=x[n,h,ww]
            r[0,v]=dcol[0]-dcol[2]; r[1,v]=dcol[1]+dcol[2]; r[2,v]=dcol[2]-dcol[1]; r[3,v]=dcol[1]-dcol[3]
        for u in range(4):
            V[u*4+0,t]=r[u,0]-r[u,2]; V[u*4+1,t]=r[u,1]+r[u,2]; V[u*4+2,t]=r[u,2]-r[u,1]; V[u*4+3,t]=r[u,1]-r[u,3]
        e=np.zeros((2,2,K))
        for a in range(2):
            for b in range(2):
                h=(2*ty+a)*d+i; ww=(2*tx+b)*d+j
                if h<H and ww<W: e[a,b]=dy[n,h,ww]
        rr=np.zeros((4,2,K))
        for b in range(2):
            rr[0,b]=e[0,b]; rr[1,b]=e[0,b]+e[1,b]; rr[2,b]=e[0,b]-e[1,b]; rr[3,b]=-e[1,b]
        for u in range(4):
            Wt[u*4+0,t]=rr[u,0]; Wt[u*4+1,t]=rr[u,0]+rr[u,1]; Wt[u*4+2,t]=rr[u,0]-rr[u,1]; Wt[u*4+3,t]=-rr[u,1]
    dU=np.einsum('xtk,xtc->xkc',Wt,V)
    dg=np.zeros((K,3,3,C))
    for k in range(K):
        for c in range(C):
            u=dU[:,k,c].reshape(4,4)
            t=np.zeros((3,4))
            for b in range(4):
                t[0,b]=u[0,b]+.5*(u[1,b]+u[2,b]); t[1,b]=.5*(u[1,b]-u[2,b]); t[2,b]=.5*(u[1,b]+u[2,b])+u[3,b]
            for r_ in range(3):
                dg[k,r_,0,c]=t[r_,0]+.5*(t[r_,1]+t[r_,2]); dg[k,r_,1,c]=.5*(t[r_,1]-t[r_,2]); dg[k,r_,2,c]=.5*(t[r_,1]+t[r_,2])+t[r_,3]
    return dg
print("filter gradient twin")
for (N,C,H,W,K,d) in [(1,3,5,7,4,1),(2,2,6,6,3,2),(1,2,9,5,2,3),(1,2,6,6,2,6),(1,3,7,8,2,4)]:
    x=rng.standard_normal((N,C,H,W)); w=rng.standard_normal((K,C,3,3))
    xt=torch.tensor(x); wt=torch.tensor(w,requires_grad=True)
    yr=F.conv2d(xt,wt,None,1,d,d); gy=rng.standard_normal(yr.shape); yr.backward(torch.tensor(gy))
    dg=twin_wgrad(x.transpose(0,2,3,1), gy.transpose(0,2,3,1), d)
    print((N,C,H,W,K,d), np.abs(dg.transpose(0,3,1,2)-wt.grad.numpy()).max())
