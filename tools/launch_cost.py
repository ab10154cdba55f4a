import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from raytracer_amd.renderer import HipRayTracer
model, camera, W, H, tex = bench.build_workload("atrium_1080p")
sc = model.scene_arrays(); dev = torch.device("cuda", 0)
tr = HipRayTracer(0); tr.upload_scene(sc)
for slot, t in enumerate(tex): tr.upload_texture(slot, t)
n = W * H
d_dir = torch.empty((n, 3), dtype=torch.float32, device=dev); tr.generate_rays_device(W, H, camera.inv_proj_view(), d_dir.data_ptr()); tr.synchronize()
d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n, 1).contiguous()
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
timing = int(os.environ.get("LAUNCH_TIMING", "1"))
tr.set_option("launch_timing", timing)
rgba8 = os.environ.get("RGBA8") == "1"   # also write the display-format pixels (rtamd_frame_out.rgba8)
blocks = int(os.environ.get("BLOCKS", "0"))
tr.set_option("max_blocks_per_cu", blocks)
print("launch_timing", timing, "max_blocks_per_cu", blocks)
for world in (1, 2, 4, 8):
    per = tr.tiles_per_rank(W, H, world) * 1024
    outs = [torch.zeros((per if world > 1 else n, 4), dtype=torch.float32, device=dev) for _ in range(8)]
    outs8 = [torch.zeros((per if world > 1 else n,), dtype=torch.int32, device=dev) for _ in range(8)]
    for S in ((1, 2, 4, 6, 8) if not os.environ.get('STREAMS') else tuple(int(x) for x in os.environ['STREAMS'].split(','))):
        K = 400
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(K):
                tr.set_stream(streams[k % S].cuda_stream)
                tr.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=True, has_normals=True, tile_rank=3 % world, tile_world=world, rgba_ptr=outs[k % S].data_ptr(),
                                 rgba8_ptr=outs8[k % S].data_ptr() if rgba8 else 0)
            t_cpu = time.perf_counter() - t0
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"tile_world {world} streams {S}: {dt / K * 1e6:.1f} us per frame (CPU issue {t_cpu / K * 1e6:.1f} us), last kernel {tr.last_kernel_ms()*1e3 if timing else -1:.1f} us", flush=True)
