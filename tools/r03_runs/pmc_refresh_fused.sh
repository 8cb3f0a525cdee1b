#!/bin/bash
# Counter passes for the shapes whose extend kernel changed with the fused pass (variant 9 default for scenes of more than one object).
set -e
bash tools/pmc_passes.sh r3_sphere_3x750000 --config sphere --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_sphere_1x2000000 --config sphere --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_1x2000000 --config instanced --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_4x1500000 --config instanced --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_ext_1x2000000 --config instanced_ext --pools 1 --paths 2000000 --steps 1 --warmup 1
bash tools/pmc_passes.sh r3_instanced_ext_4x1500000 --config instanced_ext --steps 1 --warmup 1
