"""numpy model of the contact scan's block culling (not product code): fraction of 16-vertex blocks a task of WAVE (64) Morton-sorted object
points has to score, for block sizes 8 / 16 / 32, on the synthetic body (default or `coherent`) posed by ground-truth and by noisy poses.
The block size, the seed and the two-level hierarchy of csrc/correction.hip were chosen with it.    python tools/cull_sim.py [coherent]"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from interdiff_amd import synthetic as syn
from oracle.smpl import smpl_forward
from oracle import rotations as R
torch.set_grad_enabled(False)

WAVE = 64          # points per task (csrc/correction.hip TASK)

def morton(p, bits=10):
    lo, hi = p.min(0), p.max(0)
    q = np.clip(((p - lo) / np.maximum(hi - lo, 1e-12) * (2**bits - 1)).astype(np.int64), 0, 2**bits-1)
    code = np.zeros(len(p), dtype=np.int64)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3*b + a)
    return np.argsort(code, kind='stable')

def sim(model, pose, betas, trans, objR, objT, pts, CB=16, coherent_pts=True, seedmode='rep', order='scan', SBLK=8):
    verts = smpl_forward(model, pose, betas, trans)[0].numpy().astype(np.float32)   # [N,V,3]
    vt = model['v_template'].numpy()
    vord = morton(vt)
    N, V = verts.shape[:2]
    nCB = (V + CB - 1)//CB
    tot_exec = 0; tot = 0; lane_need=0
    for n in range(N):
        v = verts[n][vord]
        q = (pts @ objR[n].T + objT[n]).astype(np.float32)
        if coherent_pts:
            q = q[morton(pts)]
        pad = nCB*CB - V
        vp = np.concatenate([v, np.full((pad,3), 3e18, np.float32)]) if pad else v
        blk = vp.reshape(nCB, CB, 3)
        real = (np.arange(nCB*CB) < V).reshape(nCB, CB)
        bmin = np.where(real[...,None], blk, np.inf).min(1); bmax = np.where(real[...,None], blk, -np.inf).max(1)
        # lower bounds [P,nCB]
        ex = np.maximum(np.maximum(bmin[None]-q[:,None], q[:,None]-bmax[None]), 0)
        lb = (ex**2).sum(-1)
        d2 = ((q[:,None,:]-v[None])**2).sum(-1)     # [P,V]
        d2p = np.concatenate([d2, np.full((len(q),pad), np.inf)],1).reshape(len(q), nCB, CB)
        bm = d2p.min(2)                               # block minima
        if seedmode=='rep':
            best0 = d2p[:,:,0].min(1)
        elif seedmode=='perfect':                     # the true nearest distance as the seed: the floor of what a better seed could reach
            best0 = d2.min(1)
        else:
            best0 = np.full(len(q), np.inf)
        # sequential sim per task of WAVE points
        for w in range(len(q)//128):
            sl = slice(128*w, 128*w+128)
            best = best0[sl].copy()
            ne = 0
            visit = range(nCB)
            if order != 'scan':                        # start at the super-block of the seed nearest to the task's closest point and walk outwards (super-block granularity, round 6)
                seeds = d2p[sl][:, ::4, 0]                                    # the kernel's seed records: first record of every 4th block
                lane = seeds.min(1).argmin()
                sb0 = (int(seeds[lane].argmin()) * 4) // SBLK
                nSB = (nCB + SBLK - 1) // SBLK
                sbs = [sb0]
                for k in range(1, nSB):
                    for c in (sb0 + k, sb0 - k):
                        if 0 <= c < nSB:
                            sbs.append(c)
                visit = [cb for sb in sbs for cb in range(sb * SBLK, min((sb + 1) * SBLK, nCB))]
            for cb in visit:
                need = lb[sl, cb] <= best
                if need.any():
                    ne += 1; lane_need += need.sum()
                    best = np.minimum(best, bm[sl, cb])
            tot_exec += ne; tot += nCB
    return tot_exec/tot, lane_need/(tot*WAVE)

if __name__ == '__main__':
    coherent = 'coherent' in sys.argv[1:]
    SEED = 'perfect' if 'perfect' in sys.argv[1:] else 'rep'
    ORDER = 'outward' if 'outward' in sys.argv[1:] else 'scan'
    kw = dict(coherent=True) if coherent else {}
    model = {k: torch.from_numpy(v) for k, v in syn.smplh_model(7, **kw).items()}
    bt = syn.make_clip_batch(seed=233, B=4, T=100, n_points=2048)
    gt = torch.from_numpy(bt['gt'])[:, 0].permute(2, 0, 1)    # [T,B,144]
    T, B = gt.shape[:2]
    for mode in ('gt', 'noisy'):
        x = gt.clone()
        if mode == 'noisy':
            x = x + 0.3*torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
        fr = [(t, b) for t in (10, 50, 99) for b in range(B)]
        body6 = torch.stack([x[t, b, :132] for t, b in fr]).reshape(-1, 22, 6)
        aa = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(body6)).reshape(len(fr), 66)
        hp = torch.from_numpy(bt['hand_pose'])
        pose = torch.cat([aa, torch.stack([hp[t, b] for t, b in fr])], 1)
        betas = torch.stack([torch.from_numpy(bt['beta'])[t, b] for t, b in fr])
        trans = torch.stack([x[t, b, 132:135] for t, b in fr])
        objR = R.rotation_6d_to_matrix(torch.stack([x[t, b, 135:141] for t, b in fr])).numpy()
        objT = torch.stack([x[t, b, 141:144] for t, b in fr]).numpy()
        for CB in ((16,) if ORDER != 'scan' else (8, 16, 32)):
            fs = []
            for i, (t, b) in enumerate(fr):
                f = sim(model, pose[i:i+1], betas[i:i+1], trans[i:i+1], objR[i:i+1], objT[i:i+1], bt['obj_points'][b], CB=CB, seedmode=SEED, order=ORDER)
                fs.append(f)
            fs = np.array(fs)
            print(mode, 'coherent' if coherent else 'default', 'CB', CB, 'exec frac mean %.3f min %.3f max %.3f | lane-need %.3f' % (fs[:,0].mean(), fs[:,0].min(), fs[:,0].max(), fs[:,1].mean()))
