#!/usr/bin/env python3
"""Writes tests/golden/sponza/: sponza_standin.mtl and textures/ -- DATA taken from the reference checkout (build container
only: /root/reference/assets/sponza.mtl and 25 of the sponza textures present under assets/textures, at native resolution).
sponza.obj itself and four curtain textures are missing from the checkout (/root/reference/.MISSING_LARGE_BLOBS); the
geometry of the config 4 stand-in is procedural (ray_tracer_2_amd/scenes.py: sponza_hetero).

    python tests/golden/sponza/make_fixture.py
"""
import os
import re
import shutil

SRC = "/root/reference/assets/"
DST = os.path.dirname(os.path.abspath(__file__)) + "/"
KEEP_DDN = {"textures/chain_texture_ddn.png", "textures/sponza_thorn_ddn.png", "textures/vase_round_ddn.png", "textures/sponza_arch_ddn.png"}
SUBST = {"textures/sponza_curtain_diff.png": "textures/sponza_fabric_diff.png",
         "textures/sponza_curtain_green_diff.png": "textures/sponza_fabric_green_diff.png",
         "textures/sponza_curtain_blue_diff.png": "textures/sponza_fabric_blue_diff.png"}
HEADER = ["# sponza_standin.mtl -- fixture derived from the reference's assets/sponza.mtl (data): its 25 materials with every value",
          "# verbatim; `map_Disp` kept for the four normal maps committed beside it and dropped otherwise (the shader only samples them in a",
          "# debug view, wgsl:542); the three curtain materials, whose textures are missing from the reference checkout itself",
          "# (.MISSING_LARGE_BLOBS), point at the fabric textures of the same colour.  tests/golden/sponza/make_fixture.py wrote it."]


def main():
    out, copied = [], set()
    for line in open(SRC + "sponza.mtl").read().split("\n"):
        m = re.match(r"\s*(map_Kd|map_Disp|map_Ka)\s+(\S+)", line)
        if m:
            kind, p = m.groups()
            if kind in ("map_Kd", "map_Ka"):
                line = line.replace(p, SUBST.get(p, p))
                if kind == "map_Kd":
                    copied.add(SUBST.get(p, p))
            elif p in KEEP_DDN:
                copied.add(p)
            else:
                continue
        out.append(line)
    open(DST + "sponza_standin.mtl", "w").write("\n".join(HEADER + out))
    os.makedirs(DST + "textures", exist_ok=True)
    for p in sorted(copied):
        shutil.copyfile(SRC + p, DST + p)
        os.chmod(DST + p, 0o644)
    print(len(copied), "textures")


if __name__ == "__main__":
    main()
