import sys; sys.path.insert(0,'.')
import numpy as np, torch
import oracle
from oracle import pipeline, grids
from evolutionary_illusion_generator_amd import synth, weights, genome as gm
from evolutionary_illusion_generator_amd.engine import Engine
w,h,ch,structure=64,64,[3,8,16,32],2
cfg = synth.make_config(2,3); pop = synth.make_population(6,cfg,seed=21)
grid = grids.create_grid(structure,w,h,10)
wts = weights.synthetic_prednet_weights(ch,w,h,seed=7)
e = Engine(w,h,ch,6); e.set_weights(wts); e.set_grid([grid['x_mat'],grid['y_mat']])
imgs = np.stack([pipeline.render_chw(g,cfg,grid,3,w,h) for _,g in pop])
d_img = torch.from_numpy(imgs).cuda()
d_fr = torch.zeros((6,22,3,h,w),dtype=torch.uint8,device='cuda')
e.prednet_rollout(d_img,6,22,0,d_fr); torch.cuda.synchronize()
fr = d_fr.cpu().numpy()
dv = torch.zeros((6,e.K,4),device='cuda'); dc = torch.zeros(6,dtype=torch.int32,device='cuda')
f21 = d_fr[:,21].contiguous()
e.flow(d_img, 3*h*w, f21, 3*h*w, 6, dv, dc); torch.cuda.synchronize()
corners, nc, nxt, st = e.debug_corners(6)
v = dv.cpu().numpy(); n = dc.cpu().numpy()
for b in range(6):
    ref_fr = oracle.prednet_rollout(wts,ch,w,h,imgs[b])
    print(b,'frames equal', np.array_equal(ref_fr, fr[b]))
    g0 = oracle.gray(imgs[b]); g1 = oracle.gray(fr[b,21])
    pts = oracle.good_features(g0)
    rn, rs = oracle.pyr_lk(g0,g1,pts)
    print('  corners equal', nc[b]==len(pts) and np.array_equal(corners[b,:nc[b]],pts), 'status equal', np.array_equal(st[b,:nc[b]], rs))
    d = np.abs(nxt[b,:nc[b]]-rn).max(1)
    for i in np.flatnonzero(d>0):
        print('   feat',i,pts[i],'gpu',nxt[b,i],'ref',rn[i],'st',st[b,i],rs[i])
