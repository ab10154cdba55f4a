#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference and oracle/_ref).  Shows, with no oracle of ours involved, that Mesa llvmpipe
(23.2.1 / LLVM 15 here) executes ray_trace_and_shade.frag wrongly for one class of pixels: when the PRIMARY ray passes exactly
through the coordinate origin, the SECOND call of traverseBVH (the shadow ray, :218) leaves its `while (stack.idx > 0)` loop
after one node with stack.idx still 1, and reports no occluder.

The shader text is read from /root/reference at run time and patched in a temporary directory (nothing of it is kept here):
  variant 1  the shadow ray's origin is replaced by one CONSTANT point P, so every input of the second traversal is the same
             for all pixels, and the shader writes shadow_t.  GLSL gives one value for all pixels; llvmpipe gives the occluder's
             distance for ordinary primary rays and 1000000 (nothing found) for primary rays through (0,0,0).
  variant 2  additionally records stack.idx when the second traversal's loop is left: 0 for ordinary rays, 1 (loop condition
             still true) for rays through the origin.
The slab test of a box of zeros is what correlates (400 of 400 rays in the sweep that found it): llvmpipe runs both sides of
`if (node.is_leaf)` under masks, loads of masked-off lanes return zeros, and a ray through (0,0,0) "hits" that all-zero box --
the likely trigger; the cause inside llvmpipe was not traced.
Moving a 1e-3 off the origin removes the effect.  Consequence for parity: DESIGN.md section 3; the soak
(soak_oracle_vs_llvmpipe.py) counts such pixels apart.  Usage: python tests/golden/llvmpipe_origin_artifact.py"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))

from oracles import REFERENCE_SHADERS, OracleB  # noqa: E402
from test_gpu_fuzz import _scene  # noqa: E402

SUN = np.frombuffer(np.uint32([0x3ED105EB, 0x3F5105EB, 0x3ED105EB]).tobytes(), dtype=np.float32)   # normalize(vec3(1, 2, 1))


def patched(edits):
    root = tempfile.mkdtemp(prefix="rtamd_shaders_")
    shutil.rmtree(root)
    shutil.copytree(REFERENCE_SHADERS, root)
    path = os.path.join(root, "ray_trace_and_shade.frag")
    text = open(path).read()
    for old, new in edits:
        assert text.count(old) == 1, old
        text = text.replace(old, new)
    open(path, "w").write(text)
    return root


def main():
    rng = np.random.default_rng(5)
    scene, tri_pos = _scene(rng, "soup")
    plain = OracleB()
    # a constant point that is in shadow according to the reference's own ray_trace.frag on llvmpipe
    for _ in range(1000):
        p = rng.uniform(-2, 2, 3).astype(np.float32)
        t_occluder = plain.trace(scene, SUN.reshape(1, 1, 3), p.reshape(1, 1, 3))[0].reshape(-1)[0]
        if t_occluder < 1000:
            break
    print(f"{plain.renderer}\nconstant shadow origin P = {p}, ray_trace.frag from P towards the sun: t = {t_occluder}")
    const_origin = ("org + dir * t + normal * 0.0001,",
                    f"vec3(uintBitsToFloat({p.view(np.uint32)[0]}u), uintBitsToFloat({p.view(np.uint32)[1]}u), uintBitsToFloat({p.view(np.uint32)[2]}u)),")
    v1 = patched([const_origin, ("fragCol = vec4(ambient + diffuse + specular, 1);", "fragCol = vec4(shadow_t);")])
    v2 = patched([
        const_origin,
        ("float intersectAABB(", "uint idx_at_exit = 77;\nfloat intersectAABB("),
        ("            }\n        }\n    }\n}", "            }\n        }\n    }\n    idx_at_exit = stack.idx;\n}"),
        ("fragCol = vec4(ambient + diffuse + specular, 1);", "fragCol = vec4(shadow_t, float(idx_at_exit), 0, 0);")])

    def fused_with(root, o):        # the harness keeps ONE shader root per process: switch it for the call
        assert plain.L.glslref_init(root.encode()) == 0
        try:
            return plain.fused(scene, d.reshape(n, 1, 3), o.reshape(n, 1, 3), False, False, None, None)[0].reshape(-1, 4)
        finally:
            assert plain.L.glslref_init(REFERENCE_SHADERS.encode()) == 0

    n = 2000
    d = (rng.choice([-1, 1], (n, 3)) * 2.0 ** rng.integers(-3, 1, (n, 3))).astype(np.float32)     # powers of two: org = -s * d is exact
    org = (-(2.0 ** rng.integers(1, 4, (n, 1))) * d).astype(np.float32)
    shifted = org.copy()
    shifted[:, 0] += np.float32(1e-3)
    failed = False
    for label, o, expect_artefact in (("primary rays through (0,0,0)", org, True), ("the same rays, origin moved 1e-3", shifted, False)):
        hit = plain.trace(scene, d.reshape(n, 1, 3), o.reshape(n, 1, 3))[0].reshape(-1, 4)[:, 0] < 1000     # pixels that reach the shading branch
        s1 = fused_with(v1, o)[hit, 0]
        s2 = fused_with(v2, o)[hit]
        found = int((s1 == t_occluder).sum())
        nothing = int((s1 == 1000000).sum())
        print(f"{label}: {int(hit.sum())} shaded pixels; second traversal from P: {found} find t = {t_occluder}, {nothing} find nothing; "
              f"stack.idx when its loop is left: { {int(k): int(c) for k, c in zip(*np.unique(s2[:, 1].astype(int), return_counts=True))} }")
        failed |= (nothing > 0) != expect_artefact or found + nothing != int(hit.sum())
    print("UNEXPECTED: this Mesa does not show the artefact as described" if failed else "as described")
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
