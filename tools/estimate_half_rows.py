"""What half rows could save: for 120 i-blocks of the equilibrated water tile (blocks of 32 atoms along a Morton curve), the j atoms within the list
cutoff of the block, and how many of them come near atoms of one 16-atom half (or one 8-atom quarter) of the block only.  The idea was built
and measured in round 4 (docs/EXPERIMENTS.md, profiles/r09a_*, r09b_*): sorting the j atoms by that class costs the pair kernel more than the
skipped evaluations give back."""
import numpy as np, sys
sys.path.insert(0,'/root/repo')
from openmm_amd import testsystems as T
from scipy.spatial import cKDTree
d = np.load('/root/repo/tests/golden/water_tile_36501_equilibrated.npz')
pos = d['positions']; L = float(d['box'])
n = len(pos)
# sort molecules along a Morton curve, blocks of 32 atoms
mol = pos.reshape(-1,3,3)[:,0,:] % L
cells = np.floor(mol / L * 64).astype(np.int64)
def part(x):
    x = x & 0x3ff
    x = (x | (x << 16)) & 0x30000ff
    x = (x | (x << 8)) & 0x300f00f
    x = (x | (x << 4)) & 0x30c30c3
    x = (x | (x << 2)) & 0x9249249
    return x
key = part(cells[:,0]) | (part(cells[:,1]) << 1) | (part(cells[:,2]) << 2)
order = np.argsort(key)
P = (pos.reshape(-1,3,3)[order]).reshape(-1,3) % L
nb = n // 32
rc = 0.9; pad = 0.1
tree = cKDTree(P, boxsize=L)
rng = np.random.default_rng(0)
tot_j = tot_inside = tot_evalA = 0; single_half = 0; quarter_evals = 0
for b in rng.choice(nb, 120, replace=False):
    I = P[32*b:32*b+32]
    c0 = I[0]
    Irel = (I - c0 + L/2) % L - L/2
    lo, hi = Irel.min(0), Irel.max(0)
    ctr = (lo+hi)/2 + c0
    cand = tree.query_ball_point(ctr % L, np.linalg.norm((hi-lo)/2) + rc + pad)
    J = P[cand]; Jrel = (J - c0 + L/2) % L - L/2
    def boxdist(lo, hi):
        dd = np.maximum(0, np.maximum(lo - Jrel, Jrel - hi)); return np.sqrt((dd**2).sum(1))
    sel = boxdist(lo, hi) < rc + pad
    Js = Jrel[sel]
    # exclude own block atoms roughly: keep all
    dist = np.sqrt(((Js[:,None,:] - Irel[None,:,:])**2).sum(2))
    inside = (dist < rc).sum()
    tot_j += sel.sum(); tot_inside += inside
    # per-j selection by min distance to atoms (individually selected j: within rc+pad of ANY i atom) -- closer to the real builder?
    near_any = dist.min(1) < rc + pad
    tot_evalA += near_any.sum()*32
    # halves: atoms 0-15, 16-31
    hA = dist[:, :16].min(1) < rc + pad; hB = dist[:, 16:].min(1) < rc + pad
    single_half += ((hA ^ hB) & near_any).sum()
    q = sum(((dist[:, 8*k:8*k+8].min(1) < rc + pad) & near_any).sum()*8 for k in range(4))
    quarter_evals += q
print("box-selected j per block", tot_j/120, " atom-selected j per block", tot_evalA/32/120)
print("evals/useful (box sel)", tot_j*32/tot_inside, " (atom sel)", tot_evalA/tot_inside)
print("fraction of atom-selected j that touch one half only", single_half/(tot_evalA/32), " -> evals/useful with halves", (tot_evalA - single_half*16)/tot_inside)
print("with quarters", quarter_evals/tot_inside)
