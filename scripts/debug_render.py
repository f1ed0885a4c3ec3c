import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import pipeline, grids, cppn
from evolutionary_illusion_generator_amd import synth, genome as gm
from evolutionary_illusion_generator_amd.engine import Engine
for (w,h,structure,seed,n) in [(64,64,2,21,6),(256,256,1,0,32),(160,120,2,5,32)]:
    cfg = synth.make_config(2,3); pop = synth.make_population(n,cfg,seed=seed)
    grid = grids.create_grid(structure,w,h,10)
    e = Engine(w,h,[3,8,16],n); e.set_grid([grid['x_mat'],grid['y_mat']])
    gb = gm.GenomeBatch([g for _,g in pop],cfg,3)
    d = torch.zeros((n,3,h,w),dtype=torch.uint8,device='cuda')
    e.render_cppn(gb,d); torch.cuda.synchronize(); got = d.cpu().numpy()
    x = grid['x_mat'].reshape(-1); y = grid['y_mat'].reshape(-1)
    tot=0
    for i,(_,g) in enumerate(pop):
        ref = pipeline.render_chw(g,cfg,grid,3,w,h)
        bad = np.argwhere(ref!=got[i])
        tot+=len(bad)
        if len(bad):
            planes = cppn.render_planes(g,cfg,[x,y])
            for b in bad[:5]:
                c,yy,xx = b
                v = np.asarray(planes[c],dtype=np.float64)[yy*w+xx]
                print('  genome',i,'px',b,'ref',ref[tuple(b)],'gpu',got[i][tuple(b)],'v*255=%r'%(v*255.0), 'act', g.nodes[c].activation)
    print((w,h,structure,seed), 'bytes differing', tot, 'of', got.size)
