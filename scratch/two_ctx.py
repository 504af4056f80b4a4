import sys, time, threading
sys.path.insert(0, '.')
import numpy as np
import rust_pathtracer_amd as pt
w,h,spp,depth=1920,1080,256,50
nctx=int(sys.argv[1]); bpc=int(sys.argv[2])
scene=pt.Scene("CornellBox").set_seeds(1,1,1).set_image(w,h,spp,depth,(0,0,0)).build()
cam=scene.camera(0.0)
ctxs=[]
for i in range(nctx):
    c=pt.Context(0); c.set_option("blocks_per_cu",bpc); c.set_option("paths_per_pass", (1<<27)//nctx); c.upload(scene); c.accum_reset(w,h); ctxs.append(c)
def run(i, out):
    b=spp*i//nctx; e=spp*(i+1)//nctx
    out[i]=ctxs[i].render_accumulate(cam,w,h,(0,0,0),b,e-b,depth,1)
for rep in range(3):
    out=[None]*nctx
    t0=time.perf_counter()
    th=[threading.Thread(target=run,args=(i,out)) for i in range(nctx)]
    [t.start() for t in th]; [t.join() for t in th]
    dt=time.perf_counter()-t0
    rays=sum(o.rays for o in out)
    print(nctx,bpc,"ms",round(dt*1e3,1),"Grays/s",round(rays/dt/1e9,2), [round(o.ms_total,1) for o in out])
