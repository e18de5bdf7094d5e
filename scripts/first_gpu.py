import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import hala_renderer_amd as H
import oracle_lib as O
from hala_renderer_amd import scenes
import __graft_entry__ as g
scene = scenes.cornell_box()
w=h=128
r = H.HalaRenderer("t", w, h, 5, 3, False, False, False, 0)
r.set_scene(scene); r.commit()
bi = r.bvh_info(); print("bvh nodes", bi.node_count, "tris", bi.triangle_count, "depth", bi.max_depth, "lds", bi.lds_node_count, list(bi.scene_min), list(bi.scene_max))
osc = O.OracleScene(scene)
nodes, tris = r.download_bvh()
print("validate", O.validate_bvh(nodes, tris, osc.triangles()))
rays = osc.camera_rays(w, h, 0)
hg, cg = r.trace_rays_host(rays, 0, count_steps=True)
ho = osc.trace(rays, 0, brute=True)
print("closest prim equal", (hg['prim']==ho['prim']).mean(), "t equal", (hg['t']==ho['t']).mean(), "uv", (hg['u']==ho['u']).mean())
hb, cb = O.trace_on_bvh(nodes, tris, rays, 0)
print("counters gpu", cg, "oracle-on-gpu-bvh", cb, "hits equal", (hb['prim']==hg['prim']).all())
g.smoke()
for f in range(4): r.update(); r.render()
img = r.read_image(0)
ref, st = osc.render(w, h, frames=4)
d = np.abs(img[...,:3]-ref[0][...,:3])
print("render maxdiff", d.max(), "exact frac", (d==0).all(-1).mean(), "stats", r.statistics().rays_total, st.rays_closest+st.rays_shadow)
np.save("gpurun_out/first_img.npy", img)
