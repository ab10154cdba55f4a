#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference and oracle/_ref): oracle A against the reference itself beyond the committed
fixtures.  The adversarial scene / ray generators of tests/test_gpu_fuzz.py (exact ties, flat boxes, duplicated and degenerate
triangles, 1e-30 .. 1e6 coordinates, denormals, zero direction components, origins on box faces, NO_RAY entries) feed the
UNMODIFIED ray_trace.frag and ray_trace_and_shade.frag on Mesa llvmpipe and oracle/oracle_a.c: hit records must agree bit for
bit, colours within 1/255.

One class of colour differences is the executor's, not the shader's, and is counted apart ("llvmpipe and
rays through the coordinate origin"; tests/golden/llvmpipe_origin_artifact.py shows it without any oracle): when a primary ray
passes EXACTLY through (0, 0, 0) -- its slab distances to that point are equal on all three axes -- the fused shader's second
traversal on llvmpipe ends after its first node, for that pixel and for pixels that share its SIMD group, so a shadowed pixel
comes out lit.  A pixel whose colour differs is therefore rendered once more alone in a 1x1 draw: it must then agree, or its
own ray must pass through the origin.  Anything else fails the soak.
Usage: python tests/golden/soak_oracle_vs_llvmpipe.py [scenes per kind] [seed]"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from common import bits, rgb_err_255  # noqa: E402
from oracles import OracleA, OracleB  # noqa: E402
from test_gpu_fuzz import _rays, _scene  # noqa: E402



def through_origin(org, d):
    """The slab test of ray_trace_and_shade.frag:53-61 against a box of zeros with t = 1000000, in float32; rays with a zero
    direction component (0 * inf) count as through the origin when the other axes allow it."""
    with np.errstate(all="ignore"):
        tx = ((np.float32(0) - org) * (np.float32(1) / d)).astype(np.float32)
        tmin, tmax = np.nanmax(tx, axis=1), np.nanmin(tx, axis=1)
        return np.isnan(tx).all(axis=1) | ((tmax >= tmin) & (tmax > 0))


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
A, B = OracleA(), OracleB()
A.set_flush_denormals(1)
print("reference:", B.renderer)
bad = n_origin = n_group = 0
for kind in ("grid", "dup", "degenerate", "scales", "soup"):
    rng = np.random.default_rng({"grid": 11, "dup": 12, "degenerate": 13, "scales": 14, "soup": 15}[kind] + 1000 * seed)
    n_rays = n_hit = 0
    worst = 0.0
    for rep in range(reps):
        sc, tri_pos = _scene(rng, kind)
        n = int(rng.integers(1, 3000))
        org, d = _rays(rng, tri_pos, n)
        o3, d3 = org.reshape(n, 1, 3), d.reshape(n, 1, 3)
        ref, _ = B.trace(sc, d3, o3)
        got = A.trace(sc, org, d)["hits"]
        ref = ref.reshape(-1, 4)
        if not np.array_equal(bits(got), bits(ref)):
            bad += 1
            k = np.where((bits(got) != bits(ref)).any(axis=1))[0]
            print(f"MISMATCH trace {kind} rep {rep}: {len(k)} of {n} rays, first {k[0]}: oracle {got[k[0]]} reference {ref[k[0]]} org {org[k[0]]} dir {d[k[0]]}")
        d2 = d.copy()
        d2[~d.any(axis=1)] = np.float32([0, 0, 1])
        fused, _ = B.fused(sc, d2.reshape(n, 1, 3), o3, False, False, None, None)
        a = A.render(sc, d2, org, 0, 0)
        fused = fused.reshape(-1, 4)
        ok = np.isfinite(fused).all(axis=1) & np.isfinite(a["rgba"]).all(axis=1)
        err = np.where(ok, rgb_err_255(a["rgba"], np.where(ok[:, None], fused, a["rgba"])), 0.0)
        origin = through_origin(org, d2)
        for i in np.where(err > 1.0)[0]:
            alone, _ = B.fused(sc, d2[i].reshape(1, 1, 3), org[i].reshape(1, 1, 3), False, False, None, None)
            e1 = rgb_err_255(a["rgba"][i:i + 1], alone.reshape(1, 4))[0]
            lit_instead_of_shadow = a["shadow"][i] == 1
            if e1 <= 1.0 and lit_instead_of_shadow:
                n_group += 1        # the neighbour's artefact: alone, the pixel agrees
                err[i] = e1
            elif origin[i] and lit_instead_of_shadow:
                n_origin += 1
                err[i] = 0.0
        e = err.max(initial=0.0)
        worst = max(worst, float(e))
        if e > 1.0 or not np.array_equal(np.isfinite(fused).all(axis=1), np.isfinite(a["rgba"]).all(axis=1)):
            bad += 1
            k = int(err.argmax())
            print(f"MISMATCH colour {kind} rep {rep}: max {e:.3f}/255 at ray {k}: oracle {a['rgba'][k]} reference {fused[k]} org {org[k]} dir {d2[k]}")
        n_rays += n
        n_hit += int((ref[:, 0] < 1e30).sum())
    print(f"{kind}: {reps} scenes, {n_rays} rays, {n_hit} hits, worst colour difference {worst:.4f}/255")
print(f"llvmpipe origin artefact (not counted): {n_origin} pixels whose own ray passes through (0,0,0), {n_group} pixels that agree when drawn alone")
print("FAILED" if bad else "ok", bad)
sys.exit(1 if bad else 0)
