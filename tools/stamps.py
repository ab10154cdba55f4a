#!/usr/bin/env python3
"""Diagnostic (GPU box): wave lifetime histogram of k_persistent from the
-DRTAMD_EXP_STAMPS build.  RTAMD_DEVICE_LIB must point at that build:
    hipcc <HIP_FLAGS of raytracer_amd/csrc/Makefile> -DRTAMD_EXP_STAMPS -Iinclude -shared \
          -o raytracer_amd/lib/exp_stamps.so raytracer_amd/csrc/device/rtamd_api.hip
    RTAMD_DEVICE_LIB=$PWD/raytracer_amd/lib/exp_stamps.so python tools/stamps.py [workload [leaf_thresh]]
The per-item records hold 65 536 items (a 1080p frame has 32 400): use it on 1080p workloads."""
import ctypes, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch, bench
from raytracer_amd.renderer import HipRayTracer
model, camera, W, H, tex = bench.build_workload(sys.argv[1] if len(sys.argv) > 1 else "atrium_1080p")
sc = model.scene_arrays(); dev = torch.device("cuda", 0)
tr = HipRayTracer(0); tr.upload_scene(sc)
for slot, t in enumerate(tex):
    if t is not None: tr.upload_texture(slot, t)
n = W * H
d_dir = torch.empty((n, 3), dtype=torch.float32, device=dev); tr.generate_rays_device(W, H, camera.inv_proj_view(), d_dir.data_ptr()); tr.synchronize()
d_org = torch.tensor(np.asarray(camera.position, np.float32), device=dev).repeat(n, 1).contiguous()
d_rgba = torch.zeros((n, 4), dtype=torch.float32, device=dev)
tr.set_option('leaf_thresh', int(sys.argv[2]) if len(sys.argv) > 2 else 0)     # 0 = the defaults (by mode); 8 = round 2 / early round 3
prev_dur = prev_ok = None
for th, nq, ad, pr in ((1, 8, 0, 0),):
    tr.set_option("dyn_thresh", th); tr.set_option("num_queues", nq)
    for _ in range(3):
        tr.render_device(W, H, d_dir.data_ptr(), d_org.data_ptr(), has_tex_coords=model.has_tex_coords(), has_normals=model.has_normals(), rgba_ptr=d_rgba.data_ptr())
    ms = tr.last_kernel_ms()
    waves = 8192
    buf = np.zeros((waves, 16), np.uint64)
    tr._lib.rtamd_exp_read_stamps.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    tr._check(tr._lib.rtamd_exp_read_stamps(tr._h, buf.ctypes.data, waves))
    b = buf[buf[:, 1] > 0].copy()
    worst = (b[:, 2] >> np.uint64(32)).astype(np.float64) / 100.0
    worst_item = (b[:, 3] >> np.uint64(8)).astype(np.int64)
    b[:, 2] &= np.uint64(0xFFFFFFFF); b[:, 3] &= np.uint64(0xFF)
    alive_us = (b[:, 1] - b[:, 0]) / 100.0
    uniform = (b[:, 6] >> np.uint64(32)).astype(np.float64); b[:, 6] &= np.uint64(0xFFFFFFFF)
    inner, inner_l, leaf, leaf_l = (b[:, k].astype(np.float64) for k in (4, 5, 6, 7))
    print(f'   wave-uniform inner iterations: {uniform.sum()/inner.sum():.3f}')
    print(f"   wave-level iterations per wave: inner mean {inner.mean():.0f}, leaf {leaf.mean():.0f}; lane util inner {inner_l.sum()/(64*inner.sum()):.3f} leaf {leaf_l.sum()/(64*leaf.sum()):.3f}; "
          f"us per inner iteration: mean {(alive_us/inner).mean():.3f} p10 {np.percentile(alive_us/inner,10):.3f} p90 {np.percentile(alive_us/inner,90):.3f}")
    tot = b[:, 14].astype(np.float64).sum()
    print(f'   shader-cycle shares: inner loop {b[:,12].astype(np.float64).sum()/tot:.3f}, leaf {b[:,13].astype(np.float64).sum()/tot:.3f}, rest (fetch/refill/shade/retire/idle at end) {1-(b[:,12].astype(np.float64).sum()+b[:,13].astype(np.float64).sum())/tot:.3f}; cycles per inner iteration {b[:,12].astype(np.float64).sum()/inner.sum():.0f}, per leaf iteration {b[:,13].astype(np.float64).sum()/leaf.sum():.0f}; clock {tot/ (alive_us.sum()):.0f} MHz')
    last_us = b[:, 15].astype(np.float64) / 100.0
    late = np.argsort(-(b[:,1].astype(np.float64)))[:8]
    print('   last-item us p50/p90/p99/max', np.round(np.percentile(last_us, [50, 90, 99, 100])), '; the 8 latest-ending waves: last item us', np.round(last_us[late]), 'items', (b[late,2] & 0xffffffff))
    n_items = ((W + 31) // 32) * ((H + 31) // 32) * 16
    ib = np.zeros((n_items, 4), np.uint64)
    tr._lib.rtamd_exp_read_items.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
    tr._check(tr._lib.rtamd_exp_read_items(tr._h, ib.ctypes.data, n_items))
    ok = ib[:, 1] > 0
    it0 = ib[ok, 0].min()
    ist = (ib[ok, 0] - it0) / 100.0; idur = (ib[ok, 1] - ib[ok, 0]) / 100.0
    iin = (ib[ok, 2] & np.uint64(0xFFFFFFFF)).astype(np.float64); ilf = (ib[ok, 2] >> np.uint64(32)).astype(np.float64)
    print(f'   items recorded {ok.sum()} of {n_items}; duration us p1/p10/p50/p90/p99/max', np.round(np.percentile(idur, [1, 10, 50, 90, 99, 100])),
          f'; corr(duration, inner+2*leaf) {np.corrcoef(idur, iin + 2 * ilf)[0, 1]:.3f}; us per iteration p10/p50/p90', np.round(np.percentile(idur / np.maximum(1, iin + ilf), [10, 50, 90]), 2))
    clk = tot / alive_us.sum()  # shader cycles per us
    itmi = (ib[ok, 3] & np.uint64(0xFFFFFFFF)).astype(np.float64) * 16 / clk; itml = (ib[ok, 3] >> np.uint64(32)).astype(np.float64) * 16 / clk
    dec = np.argsort(ist); parts = np.array_split(dec, 10)
    print('   by start-time decile: us in inner loop', [int(itmi[q].mean()) for q in parts], 'in leaf', [int(itml[q].mean()) for q in parts], 'rest', [int((idur - itmi - itml)[q].mean()) for q in parts], 'inner iters', [int(iin[q].mean()) for q in parts], 'leaf iters', [int(ilf[q].mean()) for q in parts])
    print('   by start-time decile: start us', [int(ist[q].mean()) for q in parts], 'mean duration us', [int(idur[q].mean()) for q in parts], 'mean iterations', [int((iin + ilf)[q].mean()) for q in parts], 'us/iteration', [round(float(idur[q].sum() / (iin + ilf)[q].sum()), 2) for q in parts])
    if prev_dur is not None and prev_ok is not None:
        both = ok & prev_ok
        d_now = np.zeros(n_items); d_now[ok] = idur
        print(f'   corr of item duration with previous configuration: {np.corrcoef(d_now[both], prev_dur[both])[0, 1]:.3f}')
    prev_dur = np.zeros(n_items); prev_dur[ok] = idur; prev_ok = ok
    slow = np.argsort(-alive_us)[:5]
    for o in slow:
        print(f"   slowest wave: alive {alive_us[o]:.0f} us items {b[o,2]} inner {inner[o]:.0f} leaf {leaf[o]:.0f} util {inner_l[o]/(64*inner[o]):.2f} us/inner {alive_us[o]/inner[o]:.3f}")
    if th == 1:
        order = np.argsort(-worst)[:12]
        tiles_x = (W + 31) // 32
        for o in order:
            it = worst_item[o]; j, sub = it >> 4, it & 15
            x = ((j % tiles_x) * 4 + (sub & 3)) * 8; y = ((j // tiles_x) * 4 + (sub >> 2)) * 8
            print(f"   slow item {it} at px ({x},{y}) took {worst[o]:.0f} us: inner {b[o,8]} leaf {b[o,9]} uniform {b[o,10]} lane-util {b[o,11]/max(1,64*b[o,8]):.2f} -> {worst[o]/max(1,b[o,8]+b[o,9]):.2f} us/iteration")
        print("   worst-item-per-wave us: p50/p90/p99/max", np.percentile(worst, [50, 90, 99, 100]).round(0))
    t0 = b[:, 0].min(); st = (b[:, 0] - t0) / 100.0; en = (b[:, 1] - t0) / 100.0  # us
    print(f"thresh={th} queues={nq} adaptive={ad}: kernel {ms*1e3:.0f} us, waves {len(b)}, start p50/p99/max {np.percentile(st,50):.1f}/{np.percentile(st,99):.1f}/{st.max():.1f} us, "
          f"end min/p10/p50/p90/max {en.min():.0f}/{np.percentile(en,10):.0f}/{np.percentile(en,50):.0f}/{np.percentile(en,90):.0f}/{en.max():.0f} us, "
          f"items/wave min/mean/max {b[:,2].min()}/{b[:,2].mean():.1f}/{b[:,2].max()}, mean alive {np.mean(en-st):.0f} us")
    for x in range(8):
        m = b[:, 3] == x
        print(f"   xcc {x}: waves {m.sum()}, items {int(b[m,2].sum())}, end p50 {np.percentile(en[m],50):.0f} max {en[m].max():.0f}")
