"""fp32 rounding error of the Winograd forms this library uses against an fp64 direct convolution (numpy model: transforms, channel-by-channel
fp32 accumulation and output transform in fp32; random post-ReLU-like inputs, 128 and 512 channels): direct sum, F(2x2,3x3), F(2x4,3x3), F(4x4,3x3).
    python tools/winograd_accuracy.py        (CPU only, a minute)"""
import numpy as np
rng=np.random.default_rng(0)
BT6=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],dtype=np.float64)
G6=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]],dtype=np.float64)
AT6=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],dtype=np.float64)
BT4=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],dtype=np.float64)
G4=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],dtype=np.float64)
AT4=np.array([[1,1,1,0],[0,1,-1,-1]],dtype=np.float64)
def wino(x,w,BTr,Gr,ATr,BTc,Gc,ATc,dt):
    # x [C,H,W] one tile patch set ; do for a set of tiles: x: [T,C,ph,pw], w [N,C,3,3]
    x=x.astype(dt); w=w.astype(dt)
    BTr=BTr.astype(dt);Gr=Gr.astype(dt);ATr=ATr.astype(dt);BTc=BTc.astype(dt);Gc=Gc.astype(dt);ATc=ATc.astype(dt)
    V=np.einsum('ia,tcab,jb->tcij',BTr,x,BTc).astype(dt)
    U=np.einsum('ia,ncab,jb->ncij',Gr,w,Gc).astype(dt)
    M=np.zeros((x.shape[0],w.shape[0])+V.shape[2:],dtype=dt)
    # accumulate over c sequentially in dt (chunks) to mimic fp32 accumulation
    for c in range(x.shape[1]):
        M+= (V[:,None,c]*U[None,:,c]).astype(dt)
    Y=np.einsum('pi,tnij,qj->tnpq',ATr,M,ATc).astype(dt)
    return Y
def direct(x,w,oh,ow):
    T,C=x.shape[:2];N=w.shape[0]
    y=np.zeros((T,N,oh,ow))
    for a in range(3):
        for b in range(3):
            y+=np.einsum('tcpq,nc->tnpq',x[:,:,a:a+oh,b:b+ow],w[:,:,a,b])
    return y
for C in (128,512):
    T,N=64,32
    x=np.maximum(rng.standard_normal((T,C,6,6)),0)*1.0+0.1*rng.standard_normal((T,C,6,6))
    w=rng.standard_normal((N,C,3,3))/np.sqrt(9*C)
    ref=direct(x,w,4,4)
    y44=wino(x,w,BT6,G6,AT6,BT6,G6,AT6,np.float32)
    e44=np.linalg.norm(y44-ref)/np.linalg.norm(ref)
    # F(2x4): rows 4 (first 4 rows of patch -> 2 output rows)
    y24=wino(x[:,:,:4],w,BT4,G4,AT4,BT6,G6,AT6,np.float32)
    e24=np.linalg.norm(y24-ref[:,:,:2])/np.linalg.norm(ref[:,:,:2])
    y22=wino(x[:,:,:4,:4],w,BT4,G4,AT4,BT4,G4,AT4,np.float32)
    e22=np.linalg.norm(y22-ref[:,:,:2,:2])/np.linalg.norm(ref[:,:,:2,:2])
    # direct fp32
    yd=np.zeros((T,N,4,4),dtype=np.float32)
    xf=x.astype(np.float32);wf=w.astype(np.float32)
    for c in range(C):
        for a in range(3):
            for b in range(3):
                yd+=xf[:,None,c,a:a+4,b:b+4]*wf[None,:,c,a,b,None,None]
    ed=np.linalg.norm(yd-ref)/np.linalg.norm(ref)
    print(C,'direct',ed,'F22',e22,'F24',e24,'F44',e44)
