import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import hala_renderer_amd as H
from hala_renderer_amd import workloads
cfg = workloads.baseline_config(4)
r = H.HalaRenderer("probe", cfg["width"], cfg["height"], cfg["max_depth"], cfg["rr_depth"], False, False, False, 0)
r.set_tile_shard(0, 8, 32)
r.set_envmap(cfg["env"], 0.0); r.set_scene(cfg["scene"]); r.commit()
r.set_launch_timing_period(0)
for _ in range(4):
    r.reset_accumulation(); r.update_batch(cfg["spp"]); r.render()
r.wait_idle(); r.close()
