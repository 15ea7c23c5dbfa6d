"""numpy model of three candidate changes to the contact scan (csrc/correction.hip corr_contact_kernel), round 6 -- not product code:
(a) 8-vertex blocks, (b) a flat box-vs-box prefilter (lanes = boxes, the task's bounding box against the largest seeded minimum of its 64 points) in front of the per-point tests,
(c) re-running that prefilter every `refilter` survivors.  Prints tests / executed blocks per 64-point task; the instruction model next to it is in profiles/r06_contact_bound.txt.
    python tools/contact_prefilter_sim.py"""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools'))
src = open(os.path.join(ROOT, 'tools', 'cull_sim.py')).read().split("if __name__")[0]
exec(src)

def sim2(model, pose, betas, trans, objR, objT, pts, G=64, CB=16, refilter=0):
    verts = smpl_forward(model, pose, betas, trans)[0].numpy().astype(np.float32)
    vt = model['v_template'].numpy(); vord = morton(vt)
    N, V = verts.shape[:2]; nCB = (V + CB - 1)//CB
    res = dict(tests_now=0, exec=0, surv=0, tests_pref=0, tasks=0, nCB=nCB)
    for n in range(N):
        v = verts[n][vord]
        q = (pts @ objR[n].T + objT[n]).astype(np.float32); q = q[morton(pts)]
        pad = nCB*CB - V
        vp = np.concatenate([v, np.full((pad,3), 3e18, np.float32)]) if pad else v
        blk = vp.reshape(nCB, CB, 3); real = (np.arange(nCB*CB) < V).reshape(nCB, CB)
        bmin = np.where(real[...,None], blk, np.inf).min(1); bmax = np.where(real[...,None], blk, -np.inf).max(1)
        ex = np.maximum(np.maximum(bmin[None]-q[:,None], q[:,None]-bmax[None]), 0); lb = (ex**2).sum(-1)
        d2 = ((q[:,None,:]-v[None])**2).sum(-1)
        d2p = np.concatenate([d2, np.full((len(q),pad), np.inf)],1).reshape(len(q), nCB, CB)
        bm = d2p.min(2)
        step = max(1, (nCB + 107)//108)
        best0 = d2p[:, ::step, 0].min(1)
        SB = 8
        nSB = (nCB + SB - 1)//SB
        sbmin = np.stack([bmin[s*SB:(s+1)*SB].min(0) for s in range(nSB)]); sbmax = np.stack([bmax[s*SB:(s+1)*SB].max(0) for s in range(nSB)])
        exs = np.maximum(np.maximum(sbmin[None]-q[:,None], q[:,None]-sbmax[None]), 0); lbs = (exs**2).sum(-1)
        for w in range(len(q)//G):
            sl = slice(G*w, G*w+G)
            # current kernel: two-level, evolving best
            best = best0[sl].copy(); tests = 0; ne = 0
            for sb in range(nSB):
                tests += 1
                if (lbs[sl, sb] <= best).any():
                    for cb in range(sb*SB, min((sb+1)*SB, nCB)):
                        tests += 1
                        if (lb[sl, cb] <= best).any():
                            ne += 1; best = np.minimum(best, bm[sl, cb])
            res['tests_now'] += tests; res['exec'] += ne
            # prefilter: box-box bound against the wave's max seeded best
            pmin = q[sl].min(0); pmax = q[sl].max(0)
            e2 = np.maximum(np.maximum(bmin - pmax[None], pmin[None] - bmax), 0); lbb = (e2**2).sum(-1)
            best = best0[sl].copy()
            surv = np.nonzero(lbb <= best.max())[0]
            res['surv'] += len(surv)
            tests = 0; ne2 = 0; cnt = 0
            alive = lbb <= best.max()
            for cb in range(nCB):
                if not alive[cb]: continue
                if refilter and cnt and cnt % refilter == 0:
                    alive = alive & (lbb <= best.max())
                    if not alive[cb]: continue
                cnt += 1
                tests += 1
                if (lb[sl, cb] <= best).any():
                    ne2 += 1; best = np.minimum(best, bm[sl, cb])
            assert ne2 == ne, (ne2, ne)
            res['tests_pref'] += tests; res['tasks'] += 1
    return res

model = {k: torch.from_numpy(v) for k, v in syn.smplh_model(7).items()}
bt = syn.make_clip_batch(seed=233, B=4, T=100, n_points=2048)
gt = torch.from_numpy(bt['gt'])[:, 0].permute(2, 0, 1)
T,B=gt.shape[:2]
for mode in ('gt','noisy'):
    x=gt.clone()
    if mode=='noisy': x = x + 0.05*torch.randn(x.shape, generator=torch.Generator().manual_seed(1))
    fr=[(t,b) for t in (10,50,99) for b in range(B)]
    body6 = torch.stack([x[t, b, :132] for t, b in fr]).reshape(-1, 22, 6)
    aa = R.matrix_to_axis_angle(R.rotation_6d_to_matrix(body6)).reshape(len(fr), 66)
    hp = torch.from_numpy(bt['hand_pose'])
    pose = torch.cat([aa, torch.stack([hp[t, b] for t, b in fr])], 1)
    betas = torch.stack([torch.from_numpy(bt['beta'])[t, b] for t, b in fr])
    trans = torch.stack([x[t, b, 132:135] for t, b in fr])
    objR = R.rotation_6d_to_matrix(torch.stack([x[t, b, 135:141] for t, b in fr])).numpy()
    objT = torch.stack([x[t, b, 141:144] for t, b in fr]).numpy()
    for CB in (16, 8):
      for rf in (0, 16):
        tot = None
        for i,(t,b) in enumerate(fr):
            r = sim2(model, pose[i:i+1], betas[i:i+1], trans[i:i+1], objR[i:i+1], objT[i:i+1], bt['obj_points'][b], CB=CB, refilter=rf)
            tot = r if tot is None else {k: (tot[k]+r[k] if k!='nCB' else r[k]) for k in r}
        n = tot['tasks']
        print(mode, 'CB', CB, 'refilter', rf, 'nCB', tot['nCB'], 'per task: tests now %.1f exec %.1f | prefilter survivors %.1f, per-point tests %.1f' % (tot['tests_now']/n, tot['exec']/n, tot['surv']/n, tot['tests_pref']/n))
