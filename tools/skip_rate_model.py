"""Model of a pair-kernel idea before writing it: if the list builder recorded, per row of 64 j atoms, which
groups of 8 i atoms of the 32-atom block have a partner inside the list cutoff, how many (row, group) pairs could the pair kernel
skip?  TIP3P box, Morton-sorted 32-atom blocks, list cutoff 1.08 nm.  Result: 94 % of the groups are occupied -- a 6 % saving.
    python tools/skip_rate_model.py"""
import numpy as np, sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from scipy.spatial import cKDTree
from openmm_amd import testsystems as T
w=T.water_box(20, seed=1)   # 24000 atoms
L=w.box[0,0]; pos=np.mod(w.positions, L)
# sort: Hilbert-ish -> use Morton on 0.3 nm bins of molecule... per atom like the code
cell=np.floor(pos/0.3).astype(np.int64)
key=np.zeros(len(pos),np.int64)
for bit in range(7):
    for d in range(3):
        key|=((cell[:,d]>>bit)&1)<<(3*bit+d)
order=np.argsort(key,kind='stable'); p=pos[order]
n=len(p); nb=n//32
rl=1.08; rc=0.9
tree=cKDTree(p,boxsize=L)
tot_groups=0; occ_groups=0; evals=0; useful=0; rows=0
rng=np.random.default_rng(0)
for X in rng.choice(nb,150,replace=False):
    ii=np.arange(X*32,X*32+32)
    nbrs=tree.query_ball_point(p[ii], rl)
    js=np.unique(np.concatenate(nbrs))
    js=js[js//32>=X]           # Y>=X rule
    # order by block (candidate order) then slot
    js=np.sort(js)
    d=p[js][:,None,:]-p[ii][None,:,:]; d-=np.round(d/L)*L
    r=np.linalg.norm(d,axis=2)    # [j, i]
    near=r<rl; cut=r<rc
    useful+=cut.sum()
    for r0 in range(0,len(js),64):
        blk=near[r0:r0+64]
        g=blk.reshape(len(blk),4,8).any(axis=2).any(axis=0)   # group occupied by any j in row
        tot_groups+=4; occ_groups+=g.sum(); rows+=1
        evals+=64*32
print("rows/block %.1f  group occupancy %.3f  evals/useful %.2f -> with skip %.2f" % (rows/150, occ_groups/tot_groups, evals/useful, evals*occ_groups/tot_groups/useful))
