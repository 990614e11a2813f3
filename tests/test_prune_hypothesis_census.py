"""Why cross-mesh pruning is opt-in, reproduced on the CPU with the oracle's census of every triangle
hit against its leaf box (oracle/shader_oracle.cpp: census_hit) on the directed scenes of tools/prune_directed.py:

 * axis-aligned grazing configurations: the leaf-box entry distance is never more than a few ulps beyond the hit's t -- the
   hypothesis "a hit at t is not found under a box entered far beyond t" holds with six orders of magnitude to spare, as on
   every random scene (profiles/r04_prune_census.txt);
 * the same strips in a GENERIC orientation, the ray origin within 1e-5 .. 1e-3 of their planes: the shader's own
   t = dot(ao, n) / det (wgsl:266,273) is a quotient of two cancelling sums and comes out up to 1.5 x smaller than the distance at
   which the ray enters the triangle's leaf box -- beyond the pruning's 12.5 % of slack.  That is the measured reason for
   rt_set_option("cross_prune") defaulting to 0 (tests/test_gpu_prune_directed.py: 1 texel of 2 M differs with it on)."""
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.mark.slow
def test_leaf_box_entry_against_reported_t(rt, oracle):
    import prune_directed as pd
    p = rt.make_params(960, 540, 1, 1, skybox=1, frames=0)
    oracle.census(True)
    oracle.render(p, pd.blades())
    c = oracle.census(False)
    assert c["hits"] > 100_000 and c["gt_1e-6"] == 0 and c["max_ratio"] < 1.0 + 1e-6
    R = pd._rotation((0.3, 0.5, 0.8), 0.9)
    wide = [1e-4, -1e-4, 3e-5, -3e-5, 3e-4, -3e-4, 1e-5, -1e-5, 1e-3, -1e-3, 3e-6, 0.0]
    oracle.census(True)
    oracle.render(p, pd.blades(rot=R, pencil=4e-5, segments=32, betas=wide, n_blades=12))
    c = oracle.census(False)
    assert c["hits"] > 100_000 and c["gt_12.5pct"] >= 1 and c["max_ratio"] > 1.125, c
