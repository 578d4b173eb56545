import sys, os, time, threading, json
ROOT='/root/repo'; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'tests'))
import svo_loader, util
svo = svo_loader.load()
L, R = util.urban_pair()
p = svo.elas_default_params(0)
for T in (int(a) for a in (sys.argv[1:] or ["1", "2", "4", "8", "16"])):
    ctxs=[svo.Svo(util.KITTI_W, util.KITTI_H) for _ in range(T)]
    for c in ctxs: c.elas_process(L,R,p)
    iters=30
    def w(c):
        for _ in range(iters): c.elas_process(L,R,p)
    th=[threading.Thread(target=w,args=(c,)) for c in ctxs]
    t0=time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    dt=time.perf_counter()-t0
    print(T, round(T*iters/dt,1), "pairs/s")
    for c in ctxs: c.close()
