import sys, numpy as np
sys.path.insert(0,'/root/repo')
from envidr_amd import scenes
F=np.float32
R=800
sc=scenes.toaster_scene(arrays=True) if False else None
bit = scenes.occupancy_bitfield(scenes.shell())
H=128
# unpack morton bitfield -> linear occupancy [z][y][x]
ax=np.arange(H); ix,iy,iz=np.meshgrid(ax,ax,ax,indexing='ij')
code=scenes.morton3d(ix.ravel(),iy.ravel(),iz.ravel()).astype(np.int64)
occ_m=((bit[code>>3]>>(code&7))&1).astype(bool).reshape(H,H,H)   # [x][y][z]
ro,rd=scenes.camera_rays(R,R)
N=ro.shape[0]
bound=F(1); dt=F(2*1.7320508075688772/1024)
rdi=(F(1)/rd).astype(F)
# near/far slab with aabb [-1,1] and min_near .2
t0=((-bound-ro)*rdi); t1=((bound-ro)*rdi)
near=np.maximum.reduce(np.minimum(t0,t1),axis=1); far=np.minimum.reduce(np.maximum(t0,t1),axis=1)
near=np.maximum(near,F(0.2)).astype(F); far=far.astype(F)
miss=far<near
# occupied box clip
occ_idx=np.argwhere(occ_m); lo=occ_idx.min(0); hi=occ_idx.max(0)
cs=F(2.0/H)
blo=(-1+(lo-1)*cs).astype(F); bhi=(-1+(hi+2)*cs).astype(F)
t0=((blo-ro)*rdi); t1=((bhi-ro)*rdi)
n2=np.maximum.reduce(np.minimum(t0,t1),axis=1); f2=np.minimum.reduce(np.maximum(t0,t1),axis=1)
n2=np.maximum(n2,0); miss2=f2<n2
far=np.where(miss2|miss, F(0), np.minimum(far,f2)).astype(F)
t=near.copy()
active=(t<far)
CH=16
want=np.zeros(N,int); visits_first=np.zeros(N,int)
# per-ray log of (outer index at each visit) for nested cost: inner[ray, outer] = visits in that outer iteration
inner=np.zeros((N,CH+1),int)
going=active.copy()
it=0
while going.any():
    idx=np.nonzero(going)[0]
    tt=t[idx]
    p=np.clip(ro[idx]+tt[:,None]*rd[idx],-1,1).astype(F)
    c=np.clip(((p*F(1)+F(1))*F(H/2)),0,H-1).astype(int)
    o=occ_m[c[:,0],c[:,1],c[:,2]]
    inner[idx,want[idx]]+=1
    # occupied: sample
    s=idx[o]; want[s]+=1; t[s]=t[s]+dt
    # empty: skip
    e=idx[~o]
    if e.size:
        pe=p[~o]; ce=c[~o]
        ex=(((ce+0.5+0.5*np.sign(rd[e]))*F(1.0/H)*2-1)*bound-pe)*rdi[e]
        tte=t[e]+np.maximum(0,ex.min(1)).astype(F)
        k=np.ceil((tte-t[e])/dt); k=np.maximum(k,1)
        t[e]=(t[e]+k*dt).astype(F)
    going[idx]= (t[idx]<far[idx]) & (want[idx]<CH)
    it+=1
print("iterations",it,"hit rays",(want>0).sum(),"samples",want.sum())
B=inner.reshape(-1,64,CH+1)
# nested cost per block: first-hit loop = max over lanes of inner[:,0] (visits before first sample incl. the sample visit);
# then outer iterations 1..15: sum over outer of max over lanes
first=B[:,:,0].max(1)
nested=first+B[:,:,1:CH].max(1).sum(1)
flat=B.sum(2).max(1)
hitb=(B.sum((1,2))>0)
print("blocks with work",hitb.sum())
print("nested: mean visits per block (blocks with work)",nested[hitb].mean(),"max",nested.max(), " p90",np.percentile(nested[hitb],90))
print("flat  : mean",flat[hitb].mean(),"max",flat.max()," p90",np.percentile(flat[hitb],90))
print("first-hit part mean",first[hitb].mean())
