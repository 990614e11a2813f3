#!/bin/bash
# Registers / scratch / LDS of every render kernel, from the gfx950 ISA (no GPU needed):
#   bash tools/kernel_resources.sh [extra hipcc flags]
REPO=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize \
  -I "$REPO/include" "$@" --cuda-device-only -S "$REPO/ray_tracer_2_amd/csrc/rt_kernel.hip" -o "$T/k.s" || exit 1
python3 - "$T/k.s" <<'PY'
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", s, re.S):
    pass
# metadata block: one entry per kernel
md = s[s.index("amdhsa.kernels:"):]
for ent in md.split("  - .agpr_count:")[1:]:
    g = lambda k: re.search(rf"\.{k}:\s+(\S+)", ent).group(1)
    name = g("name")
    dem = name
    print(f"{g('vgpr_count'):>4s} vgpr {g('sgpr_count'):>4s} sgpr {g('private_segment_fixed_size'):>5s} B scratch  {name[:100]}")
print("v_mfma:", len(re.findall(r"\bv_mfma", s)), " flat_:", len(re.findall(r"\bflat_(load|store)", s)), " v_pk_:", len(re.findall(r"\bv_pk_", s)))
PY
rm -rf "$T"
