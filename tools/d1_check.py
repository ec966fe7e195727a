import numpy as np, sys
sys.path.insert(0,'/root/repo')
from oracle import pyoracle as po
import pathtracer_amd as pt
from tests._util import psnr, mean_rel_error
O=po.Oracle(); R=po.Ref()
z=np.load('/root/repo/tests/golden/converged_images.npz')
for key,sid in (("scene1",1),):
    ref=z[key+"_img"]; w,h,spp,depth=int(z[key+"_w"]),int(z[key+"_h"]),int(z[key+"_spp"]),int(z[key+"_depth"])
    hr=R.scene_builtin(sid,w/h); nn=R.build_bvh(hr,0); boxes,links=R.dump_bvh(hr,nn); sa=R.scene_arrays(hr)
    oh=O.scene(sa); O.scene_set_bvh(oh,boxes,links)
    for q,name in ((po.QUERY_CLOSEST,"closest"),(po.QUERY_BVH_QUIRK,"quirk")):
        s,_=O.render_philox(oh,w,h,0,spp,depth,seed=123,query=q,nthreads=8)
        img=pt.resolve(s,spp).reshape(h,w,4)
        d=np.abs(img[...,:3]-ref[...,:3]).max(axis=2)
        print(key,name,"PSNR",psnr(img[...,:3],ref[...,:3]),"mre",mean_rel_error(img,ref),"max diff",d.max(),"pixels>0.02:",(d>0.02).sum(), np.argwhere(d>0.05)[:10].tolist())
